// blob_post.cpp -- blob list post-processing: sorting by score, discarding blobs
// whose centres are masked out, greedy non-max suppression of overlapping blobs.
//
// These are short sequential host algorithms in the reference as well (lib/visfd/feature.hpp:519-616,
// :720-913, :924-969; sphere overlap lib/visfd/visfd_utils.hpp:95-118); the greedy suppression's
// result depends on the visiting order and on WHICH earlier blobs a blob is compared with, so both
// are reproduced exactly:
//   * order: blobs are ranked by (key, original index) tuples, key = score or |score|, ascending,
//     or the exact reverse of that ranking for descending (so ties then go to the LARGER index);
//   * candidates: a kept blob k is compared with blob i only if the two meet in a coarse occupancy
//     grid (cell = `scale` voxels; a blob occupies the cells within ceil(r/scale)+1 of its centre
//     cell).  The grid origin/extents come from integer-truncated bounds of all blobs.
// Arithmetic types follow the reference expression by expression (float unless a double constant
// such as M_PI enters; see each line).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

#include "common.hpp"

namespace {

using vh::i64;

// lib/visfd/visfd_utils.hpp:95-118 (Scalar = float; constants with M_PI are double)
float sphere_overlap(float rij, float Ri, float Rj) {
  if (Ri > Rj) std::swap(Ri, Rj);
  if (rij <= Ri) return (float)((4 * M_PI / 3) * Ri * Ri * Ri);  // (double * float) * float * float
  const float xi = (float)(0.5 * (1.0 / rij) * (rij * rij + Ri * Ri - Rj * Rj));
  const float xj = (float)(0.5 * (1.0 / rij) * (rij * rij + Rj * Rj - Ri * Ri));
  const float qi = xi / Ri, qj = xj / Rj;
  const float ti = Ri * Ri * Ri * (2 - qi * (3 - qi * qi));
  const float tj = Rj * Rj * Rj * (2 - qj * (3 - qj * qj));
  return (float)((M_PI / 3) * (ti + tj));
}

// Ranking of lib/visfd/feature.hpp:519-616: returns order[] with order[rank] = original index.
void rank_blobs(const float* scores, i64 n, int criteria, bool ascending_order, std::vector<i64>* order) {
  bool ascending = ascending_order, magnitude = false;
  switch (criteria) {
    case VISFD_HIP_SORT_DECREASING: break;
    case VISFD_HIP_SORT_INCREASING: ascending = !ascending; break;
    case VISFD_HIP_SORT_DECREASING_MAGNITUDE: magnitude = true; break;
    case VISFD_HIP_SORT_INCREASING_MAGNITUDE: magnitude = true; ascending = !ascending; break;
    default: break;
  }
  std::vector<std::pair<float, i64> > key((size_t)n);
  for (i64 i = 0; i < n; i++) key[(size_t)i] = std::make_pair(magnitude ? std::fabs(scores[i]) : scores[i], i);
  std::sort(key.begin(), key.end());                  // (key, index) lexicographic
  if (!ascending) std::reverse(key.begin(), key.end());  // == sorting through reverse iterators
  order->resize((size_t)n);
  for (i64 i = 0; i < n; i++) (*order)[(size_t)i] = key[(size_t)i].second;
}

void permute(const std::vector<i64>& order, float* crds, float* diameters, float* scores) {
  const size_t n = order.size();
  std::vector<float> c(crds, crds + 3 * n), d(diameters, diameters + n), s(scores, scores + n);
  for (size_t i = 0; i < n; i++) {
    const size_t j = (size_t)order[i];
    crds[3 * i] = c[3 * j]; crds[3 * i + 1] = c[3 * j + 1]; crds[3 * i + 2] = c[3 * j + 2];
    diameters[i] = d[j];
    scores[i] = s[j];
  }
}

}  // namespace

extern "C" {

float visfd_hip_sphere_overlap(float rij, float ri, float rj) { return sphere_overlap(rij, ri, rj); }

int visfd_hip_sort_blobs(float* crds, float* diameters, float* scores, int64_t n, int sort_criteria,
                         int ascending_order, uint64_t* permutation) {
  if (n < 0 || (n > 0 && (!crds || !diameters || !scores))) return vh::fail(VISFD_HIP_EINVAL, "sort_blobs: null list");
  if (sort_criteria == VISFD_HIP_DO_NOT_SORT) {   // the reference's dispatcher falls through: nothing happens
    return VISFD_HIP_OK;
  }
  if (sort_criteria < 0 || sort_criteria > VISFD_HIP_SORT_INCREASING_MAGNITUDE)
    return vh::fail(VISFD_HIP_EINVAL, "sort_blobs: unknown sort criteria");
  if (n == 0) return VISFD_HIP_OK;
  std::vector<i64> order;
  rank_blobs(scores, n, sort_criteria, ascending_order != 0, &order);
  if (permutation)
    for (i64 i = 0; i < n; i++) permutation[i] = (uint64_t)order[(size_t)i];
  permute(order, crds, diameters, scores);
  return VISFD_HIP_OK;
}

int visfd_hip_discard_masked_blobs(float* crds, float* diameters, float* scores, int64_t* n_inout,
                                   const float* mask, int64_t nx, int64_t ny, int64_t nz) {
  if (!n_inout || *n_inout < 0) return vh::fail(VISFD_HIP_EINVAL, "discard_masked_blobs: bad count");
  if (!mask) return VISFD_HIP_OK;   // feature.hpp:951: nothing is discarded without a mask
  const i64 n = *n_inout;
  i64 kept = 0;
  for (i64 i = 0; i < n; i++) {
    // centre voxel = floor(x + 0.5) evaluated in double (feature.hpp:948-950)
    const i64 ix = (i64)std::floor((double)crds[3 * i] + 0.5);
    const i64 iy = (i64)std::floor((double)crds[3 * i + 1] + 0.5);
    const i64 iz = (i64)std::floor((double)crds[3 * i + 2] + 0.5);
    if (ix < 0 || ix >= nx || iy < 0 || iy >= ny || iz < 0 || iz >= nz)
      return vh::fail(VISFD_HIP_EINVAL, "discard_masked_blobs: a blob centre lies outside the mask image");
    if (mask[(iz * ny + iy) * nx + ix] == 0.0f) continue;
    if (kept != i) {
      std::memmove(crds + 3 * kept, crds + 3 * i, 3 * sizeof(float));
      diameters[kept] = diameters[i];
      scores[kept] = scores[i];
    }
    kept++;
  }
  *n_inout = kept;
  return VISFD_HIP_OK;
}

int visfd_hip_discard_overlapping_blobs(float* crds, float* diameters, float* scores, int64_t* n_inout,
                                        float min_radial_separation_ratio, float max_volume_overlap_large,
                                        float max_volume_overlap_small, int sort_criteria, int scale) {
  if (!n_inout || *n_inout < 0) return vh::fail(VISFD_HIP_EINVAL, "discard_overlapping_blobs: bad count");
  if (scale < 1) return vh::fail(VISFD_HIP_EINVAL, "discard_overlapping_blobs: scale must be >= 1");
  const i64 n = *n_inout;
  if (n == 0) return VISFD_HIP_OK;
  // 1. best blobs first (feature.hpp:737-743: ascending_order = false)
  VH_TRY(visfd_hip_sort_blobs(crds, diameters, scores, n, sort_criteria, 0, nullptr));

  // 2. integer bounds of all blobs (feature.hpp:756-768); float -> int conversions truncate
  int bmin[3] = {0, 0, 0}, bmax[3] = {-1, -1, -1};
  for (i64 i = 0; i < n; i++)
    for (int d = 0; d < 3; d++) {
      const float reff = std::ceil(diameters[i] / 2);
      const float c = crds[3 * i + d];
      if ((c - reff < (float)bmin[d]) || (bmin[d] > bmax[d])) bmin[d] = (int)(c - reff);
      if ((c + reff > (float)bmax[d]) || (bmin[d] > bmax[d])) bmax[d] = (int)(c + reff);
    }
  int tsz[3];
  for (int d = 0; d < 3; d++) tsz[d] = (1 + bmax[d] - bmin[d]) / scale;
  const bool has_grid = tsz[0] > 0 && tsz[1] > 0 && tsz[2] > 0;

  // 3. occupancy grid: for every cell the kept blobs (sorted positions) that reach it
  std::vector<std::vector<uint32_t> > cells(has_grid ? (size_t)tsz[0] * tsz[1] * tsz[2] : 0);
  std::vector<i64> keep;
  keep.reserve((size_t)n);
  for (i64 i = 0; i < n; i++) {
    const float reff_ = diameters[i] / 2;
    const float Reff_ = reff_ / scale;
    const int Reff = (int)std::ceil(Reff_) + 1;
    const int Reffsq = Reff * Reff;
    const float ix = crds[3 * i], iy = crds[3 * i + 1], iz = crds[3 * i + 2];
    const int C[3] = {(int)std::floor((ix - bmin[0]) / scale), (int)std::floor((iy - bmin[1]) / scale),
                      (int)std::floor((iz - bmin[2]) / scale)};
    bool discard = false;
    auto for_cells = [&](auto&& fn) {   // the blob's cells inside the table, fn returns false to stop
      if (!has_grid) return;
      for (int Jz = -Reff; Jz <= Reff; Jz++) {
        const int z = C[2] + Jz;
        if (z < 0 || z >= tsz[2]) continue;
        for (int Jy = -Reff; Jy <= Reff; Jy++) {
          const int y = C[1] + Jy;
          if (y < 0 || y >= tsz[1]) continue;
          for (int Jx = -Reff; Jx <= Reff; Jx++) {
            const int x = C[0] + Jx;
            if (x < 0 || x >= tsz[0]) continue;
            if (Jx * Jx + Jy * Jy + Jz * Jz > Reffsq) continue;
            if (!fn(cells[((size_t)z * tsz[1] + y) * tsz[0] + x])) return;
          }
        }
      }
    };
    for_cells([&](std::vector<uint32_t>& cell) {
      for (uint32_t k : cell) {
        const float kx = crds[3 * k], ky = crds[3 * k + 1], kz = crds[3 * k + 2];
        const float rik = std::sqrt((ix - kx) * (ix - kx) + (iy - ky) * (iy - ky) + (iz - kz) * (iz - kz));
        const float ri = diameters[i] / 2, rk = diameters[k] / 2;
        const float vol = sphere_overlap(rik, ri, rk);
        if (rik < (ri + rk) * min_radial_separation_ratio) discard = true;
        const float vi = (float)((4 * M_PI / 3) * (ri * ri * ri));
        const float vk = (float)((4 * M_PI / 3) * (rk * rk * rk));
        float v_large = vi, v_small = vk;
        if (vk > vi) { v_large = vk; v_small = vi; }
        if ((vol / v_small > max_volume_overlap_small) || (vol / v_large > max_volume_overlap_large)) discard = true;
      }
      return !discard;
    });
    if (discard) continue;
    keep.push_back(i);
    for_cells([&](std::vector<uint32_t>& cell) { cell.push_back((uint32_t)i); return true; });
  }
  // 4. compact the kept blobs (ascending positions: in place is safe)
  for (size_t j = 0; j < keep.size(); j++) {
    const i64 i = keep[j];
    if ((i64)j != i) {
      std::memmove(crds + 3 * j, crds + 3 * i, 3 * sizeof(float));
      diameters[j] = diameters[i];
      scores[j] = scores[i];
    }
  }
  *n_inout = (int64_t)keep.size();
  return VISFD_HIP_OK;
}

}  // extern "C"
