// connect.cpp -- LabelConnected (reference lib/visfd/connect.hpp:168-1427 with
// lib/visfd/morphology_implementation.hpp:57-515 for the seeds): agglomerative clustering of bright,
// mutually compatible voxels into "islands", the consumer of the tensor-voting output
// (bin/filter_mrc/handlers.cpp:1925-2035).
//
// This is a sequential priority-flood (Meyer) in the reference and stays on the host here: every
// step depends on the labels written by the steps before it.  What is reproduced exactly, because the
// labels depend on it:
//   * seeds: local maxima (or minima) of the saliency including plateaus, found in scan order,
//     kept if they pass the threshold, ranked by (score, scan rank) -- descending = the exact reverse
//     of the ascending ranking;
//   * flood order: a max-heap of (key, basin, (x,y,z)) tuples compared lexicographically, key =
//     saliency for maxima (-saliency for minima): a strict total order, so any heap pops alike;
//   * per-voxel admission tests in float: the finite-difference Hessian of the SALIENCY (clamped one
//     voxel inwards at the faces) against the vote tensor and against the direction, and the
//     neighbour-to-neighbour tests, with the reference's operand order;
//   * merges: the cluster with the smaller id absorbs the other; polarity bookkeeping for
//     sign-less directions; final ids by size (descending, ties to the larger provisional id).
// One quirk is load-bearing: the reference's TraceProductSym3 (lin3_utils.hpp:502-529) indexes the
// 6x2 "linear -> (i,j)" table as if it were the 3x3 "(i,j) -> linear" table, so the value it returns
// is  a0*b0 + a0*b1 + a1*b2 + a1*b0 + a1*b1 + a2*b2 + a2*b1 + a2*b2 + a0*b0  (only the diagonal
// entries enter).  That effective formula -- pinned against the compiled reference by
// tests/test_connect.py -- is what is evaluated here, since the thresholds act on it.
// Must-link constraints (connect.hpp:829-1045) and voxel weights (:1154-1290) are honoured by the _ex entry.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <queue>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "common.hpp"
#include "eigen3.hpp"

namespace {

using vh::i64;

struct Grid {
  int nx, ny, nz;
  i64 at(int x, int y, int z) const { return ((i64)z * ny + y) * nx + x; }
};

std::vector<std::array<int, 3> > neighbour_offsets(int connectivity) {   // connect.hpp:219-250
  const int r = (int)std::floor(std::sqrt((double)connectivity));
  std::vector<std::array<int, 3> > v;
  for (int jz = -r; jz <= r; jz++)
    for (int jy = -r; jy <= r; jy++)
      for (int jx = -r; jx <= r; jx++) {
        if (jx == 0 && jy == 0 && jz == 0) continue;
        if (jx * jx + jy * jy + jz * jz > connectivity) continue;
        v.push_back({{jx, jy, jz}});
      }
  return v;
}

// Seeds (morphology_implementation.hpp:57-515, one kind of extremum, allow_borders = true).
void find_extrema(const Grid& g, const float* S, const float* M, bool seek_minima, float threshold,
                  const std::vector<std::array<int, 3> >& nb, std::vector<i64>* index, std::vector<float>* score) {
  if (!seek_minima && threshold == std::numeric_limits<float>::infinity())   // :774-775
    threshold = -std::numeric_limits<float>::infinity();
  const i64 n = (i64)g.nx * g.ny * g.nz;
  std::vector<unsigned char> state((size_t)n, 0);   // 0 undefined, 1 queued, 2 done
  std::vector<i64> cand;
  std::vector<float> cscore;
  std::vector<std::array<int, 3> > plateau;
  for (int z0 = 0; z0 < g.nz; z0++)
    for (int y0 = 0; y0 < g.ny; y0++)
      for (int x0 = 0; x0 < g.nx; x0++) {
        const i64 c0 = g.at(x0, y0, z0);
        if (M && M[c0] == 0.0f) continue;
        if (state[c0] != 0) continue;
        bool is_min = true, is_max = true;
        plateau.clear();
        plateau.push_back({{x0, y0, z0}});
        state[c0] = 1;
        for (size_t head = 0; head < plateau.size(); head++) {   // breadth-first over the equal-valued region
          const int x = plateau[head][0], y = plateau[head][1], z = plateau[head][2];
          const float v = S[g.at(x, y, z)];
          for (size_t j = 0; j < nb.size(); j++) {
            const int xx = x + nb[j][0], yy = y + nb[j][1], zz = z + nb[j][2];
            if (zz < 0 || zz >= g.nz || yy < 0 || yy >= g.ny || xx < 0 || xx >= g.nx) continue;
            const i64 cj = g.at(xx, yy, zz);
            if (M && M[cj] == 0.0f) continue;
            const float vj = S[cj];
            if (vj == v) {
              if (state[cj] == 0) { plateau.push_back({{xx, yy, zz}}); state[cj] = 1; }
            } else if (vj < v) is_min = false;
            else if (vj > v) is_max = false;
          }
        }
        for (size_t k = 0; k < plateau.size(); k++) state[g.at(plateau[k][0], plateau[k][1], plateau[k][2])] = 2;
        const float v0 = S[c0];
        if (seek_minima ? (is_min && v0 <= threshold) : (is_max && v0 >= threshold)) {
          cand.push_back(c0);
          cscore.push_back(v0);
        }
      }
  // ranking: minima ascending, maxima descending = the reverse of ascending (score, rank) (:404-470)
  std::vector<std::pair<float, i64> > key(cand.size());
  for (size_t i = 0; i < cand.size(); i++) key[i] = std::make_pair(cscore[i], (i64)i);
  std::sort(key.begin(), key.end());
  if (!seek_minima) std::reverse(key.begin(), key.end());
  index->resize(cand.size());
  score->resize(cand.size());
  for (size_t i = 0; i < cand.size(); i++) {
    (*index)[i] = cand[(size_t)key[i].second];
    (*score)[i] = cscore[(size_t)key[i].second];
  }
}

// lib/visfd/visfd_utils.hpp:528-616: finite-difference Hessian, stencil centre moved one voxel inwards at a face
void hessian_fd(const Grid& g, const float* S, int ix, int iy, int iz, float H[3][3]) {
  if (ix == 0) ix++; else if (ix == g.nx - 1) ix--;
  if (iy == 0) iy++; else if (iy == g.ny - 1) iy--;
  if (iz == 0) iz++; else if (iz == g.nz - 1) iz--;
  auto s = [&](int dx, int dy, int dz) { return S[g.at(ix + dx, iy + dy, iz + dz)]; };
  H[0][0] = (s(1, 0, 0) + s(-1, 0, 0) - 2 * s(0, 0, 0));
  H[1][1] = (s(0, 1, 0) + s(0, -1, 0) - 2 * s(0, 0, 0));
  H[2][2] = (s(0, 0, 1) + s(0, 0, -1) - 2 * s(0, 0, 0));
  H[0][1] = (float)(0.25 * (s(1, 1, 0) + s(-1, -1, 0) - s(1, -1, 0) - s(-1, 1, 0)));
  H[1][0] = H[0][1];
  H[1][2] = (float)(0.25 * (s(0, 1, 1) + s(0, -1, -1) - s(0, 1, -1) - s(0, -1, 1)));
  H[2][1] = H[1][2];
  H[2][0] = (float)(0.25 * (s(1, 0, 1) + s(-1, 0, -1) - s(-1, 0, 1) - s(1, 0, -1)));
  H[0][2] = H[2][0];
}

// the value the reference's TraceProductSym3 actually returns (see the header comment)
inline float trace_product_sym3(const float a[6], const float b[6]) {
  return a[0] * b[0] + a[0] * b[1] + a[1] * b[2] + a[1] * b[0] + a[1] * b[1] + a[2] * b[2] + a[2] * b[1] +
         a[2] * b[2] + a[0] * b[0];
}
inline float frobenius_sym3(const float a[6]) { return std::sqrt(trace_product_sym3(a, a)); }
inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline float sqr(float x) { return x * x; }

}  // namespace

extern "C" int visfd_hip_label_connected_ex(const float* saliency, int64_t* labels, const float* mask, int64_t nx64,
                                            int64_t ny64, int64_t nz64, float threshold_saliency, float* direction,
                                            float threshold_vector_saliency, float threshold_vector_neighbor,
                                            int consider_dot_product_sign, const float* tensor,
                                            float threshold_tensor_saliency, float threshold_tensor_neighbor,
                                            int tensor_is_positive_definite_near_target, int connectivity,
                                            int64_t label_undefined, int sort_by_size, int standardize_directions,
                                            int start_from_saliency_maxima, int64_t* n_clusters_out,
                                            float* cluster_maxima, float* cluster_sizes, float* cluster_saliencies,
                                            int64_t cluster_capacity, const float* voxel_weights,
                                            const float* must_link_crds, const int64_t* must_link_group_sizes,
                                            int64_t must_link_ngroups, const int* must_link_directions) {
  if (!saliency || !labels) return vh::fail(VISFD_HIP_EINVAL, "label_connected: null image");
  VH_TRY(vh::check_dims(nx64, ny64, nz64));
  if (nx64 < 3 || ny64 < 3 || nz64 < 3)   // visfd_utils.hpp:585-587 (asserted there)
    return vh::fail(VISFD_HIP_EINVAL, "label_connected: the image must be at least 3 voxels wide in every direction");
  if (nx64 * ny64 * nz64 >= (1LL << 31)) return vh::fail(VISFD_HIP_EINVAL, "label_connected: image too large");
  if (connectivity < 1) return vh::fail(VISFD_HIP_EINVAL, "label_connected: connectivity must be >= 1");
  const Grid g = {(int)nx64, (int)ny64, (int)nz64};
  const i64 n = (i64)g.nx * g.ny * g.nz;
  const float* S = saliency;
  const float* M = mask;
  float* V = direction;
  const float* T = tensor;
  const bool from_max = start_from_saliency_maxima != 0;
  const bool signs = consider_dot_product_sign != 0;
  const bool standardize = V && standardize_directions && !signs;
  const int order = from_max ? VISFD_HIP_DECREASING_EIVALS : VISFD_HIP_INCREASING_EIVALS;   // connect.hpp:197-201
  if (!signs) {   // connect.hpp:207-216: a negative threshold means "do not compare"
    if (threshold_vector_saliency < 0) threshold_vector_saliency = 0.0f;
    if (threshold_vector_neighbor < 0) threshold_vector_neighbor = 0.0f;
  }
  const std::vector<std::array<int, 3> > nb = neighbour_offsets(connectivity);
  const float SIGN = from_max ? -1.0f : 1.0f;

  std::vector<i64> seed;
  std::vector<float> seed_score;
  find_extrema(g, S, M, !from_max, threshold_saliency, nb, &seed, &seed_score);
  const i64 nseeds = (i64)seed.size();
  const i64 UNDEFINED = nseeds + 1, QUEUED = nseeds + 2;
  for (i64 i = 0; i < n; i++) labels[i] = UNDEFINED;

  typedef std::tuple<float, i64, float, float, float> Entry;   // key, basin, x, y, z (coordinates compare as floats)
  std::priority_queue<Entry> q;
  for (i64 i = 0; i < nseeds; i++) {
    const i64 c = seed[(size_t)i];
    const int x = (int)(c % g.nx), y = (int)((c / g.nx) % g.ny), z = (int)(c / ((i64)g.nx * g.ny));
    const float sc = seed_score[(size_t)i] * SIGN;
    q.push(Entry(-sc, i, (float)x, (float)y, (float)z));
    labels[c] = QUEUED;
  }
  std::vector<i64> basin2cluster((size_t)nseeds);
  std::vector<std::vector<i64> > cluster2basins((size_t)nseeds);
  for (i64 i = 0; i < nseeds; i++) { basin2cluster[(size_t)i] = i; cluster2basins[(size_t)i].assign(1, i); }
  std::vector<signed char> polarity((size_t)nseeds, 1);

  while (!q.empty()) {
    const Entry p = q.top();
    q.pop();
    const float i_score = -std::get<0>(p);
    const i64 basin = std::get<1>(p);
    const int ix = (int)std::get<2>(p), iy = (int)std::get<3>(p), iz = (int)std::get<4>(p);
    const i64 c = g.at(ix, iy, iz);
    if (i_score > threshold_saliency * SIGN) { labels[c] = UNDEFINED; continue; }
    if (M && M[c] == 0.0f) { labels[c] = UNDEFINED; continue; }
    {   // is the saliency's own shape here compatible with the tensor / direction? (connect.hpp:455-553)
      float H3[3][3];
      hessian_fd(g, S, ix, iy, iz, H3);
      if ((tensor_is_positive_definite_near_target != 0) == from_max)
        for (int a = 0; a < 3; a++)
          for (int b = a; b < 3; b++) H3[a][b] *= -1.0f;
      const float H[6] = {H3[0][0], H3[1][1], H3[2][2], H3[0][1], H3[1][2], H3[0][2]};   // lin3_utils.hpp:400-403
      bool discard = false;
      if (T) {
        const float tp = trace_product_sym3(H, T + 6 * c);
        const float fs = frobenius_sym3(H), ft = frobenius_sym3(T + 6 * c);
        if (tp < threshold_tensor_saliency * fs * ft) discard = true;
      }
      if (V) {
        float d6[6], e0[3];
        vh::eig::diagonalize_flat(H, order, d6);      // ConvertFlatSym2Evects3: principal eigenvector = row 0
        vh::eig::shoemake_row0(d6 + 3, e0);
        const float* v = V + 3 * c;
        bool exceeded = false;
        if (signs) {
          if (dot3(e0, v) < (threshold_vector_saliency * std::sqrt(dot3(e0, e0)) * std::sqrt(dot3(v, v)))) exceeded = true;
        } else {
          if (sqr(dot3(e0, v)) < (sqr(threshold_vector_saliency) * dot3(e0, e0) * dot3(v, v))) exceeded = true;
        }
        if (exceeded) discard = true;
      }
      if (discard) {
        labels[c] = UNDEFINED;
        if (c == seed[(size_t)basin]) basin2cluster[(size_t)basin] = -1;   // the whole basin is dropped
        continue;
      }
    }
    labels[c] = basin;
    for (size_t j = 0; j < nb.size(); j++) {
      const int xx = ix + nb[j][0], yy = iy + nb[j][1], zz = iz + nb[j][2];
      if (zz < 0 || zz >= g.nz || yy < 0 || yy >= g.ny || xx < 0 || xx >= g.nx) continue;
      const i64 cj = g.at(xx, yy, zz);
      if (M && M[cj] == 0.0f) continue;
      if (T) {   // neighbour compatibility (connect.hpp:625-672; both tests sit under "if tensor" there)
        const float* ti = T + 6 * c;
        const float* tj = T + 6 * cj;
        if (trace_product_sym3(ti, tj) < (threshold_tensor_neighbor * frobenius_sym3(ti) * frobenius_sym3(tj))) continue;
        if (V) {
          const float* vi = V + 3 * c;
          const float* vj = V + 3 * cj;
          if (signs) {
            if (dot3(vi, vj) < (threshold_tensor_neighbor * std::sqrt(dot3(vi, vi)) * std::sqrt(dot3(vj, vj)))) continue;
          } else {
            if (sqr(dot3(vi, vj)) < (sqr(threshold_vector_neighbor) * dot3(vi, vi) * dot3(vj, vj))) continue;
          }
        }
      }
      if (labels[cj] == QUEUED) continue;
      if (labels[cj] == UNDEFINED) {
        labels[cj] = QUEUED;
        const float ns = S[cj] * SIGN;
        q.push(Entry(-ns, basin, (float)xx, (float)yy, (float)zz));
        if (standardize && dot3(V + 3 * c, V + 3 * cj) < 0.0f) {   // newcomers follow the voxel that reached them
          V[3 * cj] *= -1.0f; V[3 * cj + 1] *= -1.0f; V[3 * cj + 2] *= -1.0f;
        }
        continue;
      }
      // the neighbour already belongs to a basin: same cluster, or merge
      const i64 basin_i = basin, basin_j = labels[cj];
      const i64 ci = basin2cluster[(size_t)basin_i], cjl = basin2cluster[(size_t)basin_j];
      bool polarity_match = true;
      if (standardize && (float)(dot3(V + 3 * c, V + 3 * cj) * polarity[(size_t)basin_i] * polarity[(size_t)basin_j]) < 0.0f)
        polarity_match = false;
      if (ci == cjl) continue;   // (a polarity mismatch inside one cluster only skips this neighbour)
      const i64 keep = std::min(ci, cjl), gone = std::max(ci, cjl);
      if (keep < 0) {   // a basin whose seed was discarded carries cluster -1 (connect.hpp:549); the reference indexes
        continue;       // cluster2basins[-1] there (undefined behaviour) -- such contacts are ignored here
      }
      for (size_t k = 0; k < cluster2basins[(size_t)gone].size(); k++) {
        const i64 b = cluster2basins[(size_t)gone][k];
        cluster2basins[(size_t)keep].push_back(b);
        basin2cluster[(size_t)b] = keep;
        if (standardize && !polarity_match) polarity[(size_t)b] = (signed char)-polarity[(size_t)b];
      }
      cluster2basins[(size_t)gone].clear();
    }
  }

  // ---- must-link constraints (connect.hpp:829-1045): for every group of locations, the clusters of the clustered
  // voxels nearest to consecutive locations are merged (the smaller id absorbs the other), with the polarity of the
  // absorbed basins flipped when the two contact voxels face incompatibly (given per location, or inferred from the
  // angles their normals make with the line joining them).
  if (must_link_crds && must_link_ngroups > 0) {
    if (!must_link_group_sizes) return vh::fail(VISFD_HIP_EINVAL, "label_connected: must-link group sizes missing");
    i64 at = 0;
    for (i64 grp = 0; grp < must_link_ngroups; grp++) {
      i64 basin_j = -1;
      int rj[3] = {-1, -1, -1};
      for (i64 loc = 0; loc < must_link_group_sizes[grp]; loc++, at++) {
        const int want[3] = {(int)std::floor(must_link_crds[3 * at] + 0.5), (int)std::floor(must_link_crds[3 * at + 1] + 0.5),
                             (int)std::floor(must_link_crds[3 * at + 2] + 0.5)};
        // FindNearestVoxel (visfd_utils.hpp:147-186): scan order, strictly closer wins, distances in double
        int ri[3] = {-1, -1, -1};
        double r_min_sq = -1.0;
        for (int z = 0; z < g.nz; z++)
          for (int y = 0; y < g.ny; y++)
            for (int x = 0; x < g.nx; x++) {
              const i64 c = g.at(x, y, z);
              if (M && M[c] == 0.0f) continue;
              if (labels[c] == UNDEFINED) continue;
              const double dx = want[0] - x, dy = want[1] - y, dz = want[2] - z;
              const double r_sq = dx * dx + dy * dy + dz * dz;
              if (r_min_sq == -1.0 || r_sq < r_min_sq) { r_min_sq = r_sq; ri[0] = x; ri[1] = y; ri[2] = z; }
            }
        if (ri[0] == -1)
          return vh::fail(VISFD_HIP_EINVAL, "Error: No voxels clustered. Empty image. Your cluster criteria are too strict.\n"
                                            "       (Attempting the find the nearest voxel from an empty set.)\n");
        const i64 basin_i = labels[g.at(ri[0], ri[1], ri[2])];
        if (basin_j != -1 && basin_i != basin_j) {
          const i64 ci = basin2cluster[(size_t)basin_i], cj = basin2cluster[(size_t)basin_j];
          if (ci != cj && ci >= 0 && cj >= 0) {
            const i64 keep = std::min(ci, cj), gone = std::max(ci, cj);
            bool flip = false;
            if (V) {
              const float* n_i = V + 3 * g.at(ri[0], ri[1], ri[2]);
              const float* n_j = V + 3 * g.at(rj[0], rj[1], rj[2]);
              float r_ij[3] = {(float)(ri[0] - rj[0]), (float)(ri[1] - rj[1]), (float)(ri[2] - rj[2])};
              {   // Normalize3 (lin3_utils.hpp:143-155)
                const float len = std::sqrt(r_ij[0] * r_ij[0] + r_ij[1] * r_ij[1] + r_ij[2] * r_ij[2]);
                if (len > 0) { const float inv = (float)(1.0 / (double)len); for (int d = 0; d < 3; d++) r_ij[d] *= inv; }
                else { r_ij[0] = 1.0f; r_ij[1] = 0.0f; r_ij[2] = 0.0f; }
              }
              bool polarity_match;
              const int how = must_link_directions ? must_link_directions[at] : 2;   // 0 same, 1 opposite, 2 auto
              if (how == 0) polarity_match = dot3(n_i, n_j) > 0;
              else if (how == 1) polarity_match = dot3(n_i, n_j) < 0;
              else {
                const float ni_dot_rij = dot3(n_i, r_ij), nj_dot_rij = dot3(n_j, r_ij);
                const float theta0 = (float)(M_PI / 4);
                const float theta_i = std::asin(std::abs(ni_dot_rij)), theta_j = std::asin(std::abs(nj_dot_rij));
                if (theta_i < theta0 && theta_j < theta0) polarity_match = dot3(n_i, n_j) > 0;
                else polarity_match = (ni_dot_rij * nj_dot_rij <= 0);
              }
              flip = polarity_match != (polarity[(size_t)basin_i] == polarity[(size_t)basin_j]);
            }
            for (size_t k = 0; k < cluster2basins[(size_t)gone].size(); k++) {
              const i64 b = cluster2basins[(size_t)gone][k];
              cluster2basins[(size_t)keep].push_back(b);
              basin2cluster[(size_t)b] = keep;
              if (standardize && flip) polarity[(size_t)b] = (signed char)-polarity[(size_t)b];
            }
            cluster2basins[(size_t)gone].clear();
          }
        }
        basin_j = basin_i;
        for (int d = 0; d < 3; d++) rj[d] = ri[d];
      }
    }
  }

  // provisional cluster numbers 0..n_clusters-1 in basin order (connect.hpp:1049-1075)
  std::vector<i64> old2new((size_t)nseeds), deepest;
  i64 n_clusters = 0;
  for (i64 i = 0; i < nseeds; i++) {
    old2new[(size_t)i] = n_clusters;
    if (basin2cluster[(size_t)i] == i) { deepest.push_back(i); n_clusters++; }
  }
  for (i64 i = 0; i < nseeds; i++)
    if (basin2cluster[(size_t)i] >= 0) basin2cluster[(size_t)i] = old2new[(size_t)basin2cluster[(size_t)i]];
  auto live = [&](i64 c) { return !(M && M[c] == 0.0f) && labels[c] != UNDEFINED; };
  if (standardize)
    for (i64 c = 0; c < n; c++)
      if (live(c)) {
        const float pol = (float)polarity[(size_t)labels[c]];
        V[3 * c] *= pol; V[3 * c + 1] *= pol; V[3 * c + 2] *= pol;
      }
  for (i64 c = 0; c < n; c++)
    if (live(c)) labels[c] = basin2cluster[(size_t)labels[c]];
  std::vector<long double> sizes((size_t)n_clusters, 0.0L);
  const float* W = voxel_weights;   // connect.hpp:1154-1183: the "size" of a cluster is then the sum of its voxels' weights
  for (i64 c = 0; c < n; c++)
    if (live(c)) sizes[(size_t)labels[c]] += W ? (long double)W[c] : 1.0L;
  if (standardize) {   // outward orientation by the sign of sum (r - r_com).n (connect.hpp:1192-1290)
    std::vector<std::array<long double, 3> > com((size_t)n_clusters, std::array<long double, 3>{{0.0L, 0.0L, 0.0L}});
    for (int z = 0; z < g.nz; z++)
      for (int y = 0; y < g.ny; y++)
        for (int x = 0; x < g.nx; x++) {
          const i64 c = g.at(x, y, z);
          if (!live(c)) continue;
          if (W) {   // connect.hpp:1211-1224: int * float products in float, summed in long double
            com[(size_t)labels[c]][0] += x * W[c]; com[(size_t)labels[c]][1] += y * W[c]; com[(size_t)labels[c]][2] += z * W[c];
          } else {
            com[(size_t)labels[c]][0] += x; com[(size_t)labels[c]][1] += y; com[(size_t)labels[c]][2] += z;
          }
        }
    for (i64 k = 0; k < n_clusters; k++)
      for (int d = 0; d < 3; d++) com[(size_t)k][d] /= sizes[(size_t)k];
    std::vector<long double> sum_dot((size_t)n_clusters, 0.0L);
    for (int z = 0; z < g.nz; z++)
      for (int y = 0; y < g.ny; y++)
        for (int x = 0; x < g.nx; x++) {
          const i64 c = g.at(x, y, z);
          if (!live(c)) continue;
          const i64 k = labels[c];
          const float r[3] = {(float)(x - com[(size_t)k][0]), (float)(y - com[(size_t)k][1]), (float)(z - com[(size_t)k][2])};
          long double delta_sum = (long double)dot3(r, V + 3 * c);
          if (W) delta_sum *= W[c];
          sum_dot[(size_t)k] += delta_sum;
        }
    for (i64 c = 0; c < n; c++)
      if (live(c) && sum_dot[(size_t)labels[c]] < 0.0L) { V[3 * c] *= -1.0f; V[3 * c + 1] *= -1.0f; V[3 * c + 2] *= -1.0f; }
  }
  // per-cluster outputs in provisional order, then the final order
  std::vector<i64> perm((size_t)n_clusters);
  for (i64 k = 0; k < n_clusters; k++) perm[(size_t)k] = k;
  if (sort_by_size && n_clusters > 0) {   // connect.hpp:1322-1365: (float size, id) descending = reverse of ascending
    std::vector<std::pair<float, i64> > key((size_t)n_clusters);
    for (i64 k = 0; k < n_clusters; k++) key[(size_t)k] = std::make_pair((float)sizes[(size_t)k], k);
    std::sort(key.begin(), key.end());
    std::reverse(key.begin(), key.end());
    std::vector<i64> inv((size_t)n_clusters);
    for (i64 k = 0; k < n_clusters; k++) { perm[(size_t)k] = key[(size_t)k].second; inv[(size_t)key[(size_t)k].second] = k; }
    for (i64 c = 0; c < n; c++)
      if (live(c)) labels[c] = inv[(size_t)labels[c]];
  }
  for (i64 c = 0; c < n; c++) {
    if (M && M[c] == 0.0f) continue;             // masked voxels keep the provisional "undefined" code (connect.hpp:1401-1403)
    if (labels[c] == UNDEFINED) labels[c] = label_undefined;
    else labels[c] += 1;
  }
  if (n_clusters_out) *n_clusters_out = n_clusters;
  if (cluster_maxima || cluster_sizes || cluster_saliencies) {
    if (cluster_capacity < n_clusters) return vh::fail(VISFD_HIP_ECAPACITY, "label_connected: cluster arrays too small");
    for (i64 k = 0; k < n_clusters; k++) {
      // quirk kept: only the seed list follows the size ordering; sizes and saliencies stay in provisional
      // order (connect.hpp:1294-1315 fill all three before the sort, :1347-1348 permutes the maxima only)
      const i64 cm = seed[(size_t)deepest[(size_t)perm[(size_t)k]]];
      if (cluster_maxima) {
        cluster_maxima[3 * k] = (float)(cm % g.nx);
        cluster_maxima[3 * k + 1] = (float)((cm / g.nx) % g.ny);
        cluster_maxima[3 * k + 2] = (float)(cm / ((i64)g.nx * g.ny));
      }
      if (cluster_sizes) cluster_sizes[k] = (float)sizes[(size_t)k];
      if (cluster_saliencies) cluster_saliencies[k] = S[seed[(size_t)deepest[(size_t)k]]];
    }
  }
  return VISFD_HIP_OK;
}

extern "C" int visfd_hip_label_connected(const float* saliency, int64_t* labels, const float* mask, int64_t nx64,
                                         int64_t ny64, int64_t nz64, float threshold_saliency, float* direction,
                                         float threshold_vector_saliency, float threshold_vector_neighbor,
                                         int consider_dot_product_sign, const float* tensor,
                                         float threshold_tensor_saliency, float threshold_tensor_neighbor,
                                         int tensor_is_positive_definite_near_target, int connectivity,
                                         int64_t label_undefined, int sort_by_size, int standardize_directions,
                                         int start_from_saliency_maxima, int64_t* n_clusters_out,
                                         float* cluster_maxima, float* cluster_sizes, float* cluster_saliencies,
                                         int64_t cluster_capacity) {
  return visfd_hip_label_connected_ex(saliency, labels, mask, nx64, ny64, nz64, threshold_saliency, direction,
                                      threshold_vector_saliency, threshold_vector_neighbor, consider_dot_product_sign, tensor,
                                      threshold_tensor_saliency, threshold_tensor_neighbor,
                                      tensor_is_positive_definite_near_target, connectivity, label_undefined, sort_by_size,
                                      standardize_directions, start_from_saliency_maxima, n_clusters_out, cluster_maxima,
                                      cluster_sizes, cluster_saliencies, cluster_capacity, nullptr, nullptr, nullptr, 0, nullptr);
}

namespace {
template <typename F>
void parallel_voxels(int64_t nvox, F work) {
  unsigned nt = std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if ((int64_t)nt > nvox / 4096 + 1) nt = (unsigned)(nvox / 4096 + 1);
  std::vector<std::thread> pool;
  const int64_t chunk = (nvox + nt - 1) / nt;
  for (unsigned t = 1; t < nt; t++)
    pool.emplace_back(work, std::min<int64_t>(nvox, t * chunk), std::min<int64_t>(nvox, (t + 1) * chunk));
  work(0, std::min<int64_t>(nvox, chunk));
  for (size_t t = 0; t < pool.size(); t++) pool[t].join();
}
}  // namespace

// Post-vote score lambda0 - lambda1 (handlers.cpp:1868-1888) of nvox flat tensors [nvox][6], on the HOST in the
// reference's arithmetic: the clustering that follows orders and thresholds voxels by this number, so the
// command-line path recomputes it here instead of using the device kernel's value (equal to ~1e-7 only).
extern "C" int visfd_hip_tensor_saliency_host(const float* tensor, const float* mask, int64_t nvox, int order,
                                              float* saliency) {
  if (!tensor || !saliency || nvox < 0) return vh::fail(VISFD_HIP_EINVAL, "tensor_saliency_host: bad argument");
  parallel_voxels(nvox, [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; i++) {
      if (mask && mask[i] == 0.0f) continue;
      double lam[3];
      vh::eig::D3 E[3];
      vh::eig::eig_sym3(tensor + 6 * i, order, lam, E, true);   // the reference diagonalises fully (same eigenvalues)
      const double l1 = (float)lam[0], l2 = (float)lam[1];      // stored as float, re-read as double
      saliency[i] = (float)(l1 - l2);
    }
  });
  return VISFD_HIP_OK;
}

// Principal eigenvector (row 0 of ConvertFlatSym2Evects3) of every voxel's tensor, on the HOST with glibc
// arithmetic -- bit-identical to the reference's loop at bin/filter_mrc/handlers.cpp:1935-1952, which matters
// because LabelConnected thresholds act on these directions.  Voxels with mask == 0 are left untouched.
// The voxels are independent, so the loop is split over the host's cores.
extern "C" int visfd_hip_principal_directions_host(const float* tensor, const float* mask, int64_t nvox, int order,
                                                   float* direction) {
  if (!tensor || !direction || nvox < 0) return vh::fail(VISFD_HIP_EINVAL, "principal_directions_host: bad argument");
  parallel_voxels(nvox, [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; i++) {
      if (mask && mask[i] == 0.0f) continue;
      float d6[6];
      vh::eig::diagonalize_flat(tensor + 6 * i, order, d6);
      vh::eig::shoemake_row0(d6 + 3, direction + 3 * i);
    }
  });
  return VISFD_HIP_OK;
}

// DiagonalizeFlatSym3 (eigen3_simple.hpp:271-342) of n flat matrices on the HOST, bit-identical to the reference
// (the per-voxel call a library user makes inside their own loops; the device batch is visfd_hip_diagonalize_flat_sym3).
extern "C" int visfd_hip_diagonalize_flat_sym3_host(const float* m6, float* out6, int64_t n, int order) {
  if (!m6 || !out6 || n < 0 || order < 0 || order > 1) return vh::fail(VISFD_HIP_EINVAL, "diagonalize_flat_sym3_host: bad argument");
  parallel_voxels(n, [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; i++) {
      float d6[6];
      vh::eig::diagonalize_flat(m6 + 6 * i, order, d6);
      for (int c = 0; c < 6; c++) out6[6 * i + c] = d6[c];
    }
  });
  return VISFD_HIP_OK;
}

// ConvertFlatSym2Evects3<float> (eigen3_simple.hpp:392-405): flat symmetric matrix -> eigenvalues and the
// eigenvectors as rows, through the same float Shoemake round trip as the reference (host).
extern "C" int visfd_hip_convert_flat_sym2_evects3_host(const float* m6, int order, float* eivals3, float* eivects9) {
  if (!m6 || !eivals3 || !eivects9 || order < 0 || order > 1) return vh::fail(VISFD_HIP_EINVAL, "convert_flat_sym2_evects3_host: bad argument");
  float d6[6], M[3][3];
  vh::eig::diagonalize_flat(m6, order, d6);
  vh::eig::shoemake_frame(d6 + 3, M);
  for (int c = 0; c < 3; c++) eivals3[c] = d6[c];
  for (int c = 0; c < 9; c++) eivects9[c] = M[c / 3][c % 3];
  return VISFD_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Oriented surface points (the -normals-file tail of HandleTV, bin/filter_mrc/handlers.cpp:2039-2309).
// For every unmasked voxel of the selected cluster: follow the (standardized) normal direction forwards and
// backwards in steps of `curve_ds` while staying inside the cluster, take the saliency-weighted mean arc
// length as the position of the sheet along that normal, optionally snap it to the ridge of the saliency
// (one Newton step along the principal Hessian direction), and emit position and normal (normal scaled by the
// voxel's saliency).  All arithmetic is the reference's float arithmetic, including its 3x3 eigen solver
// instantiated for float: double literals promote exactly where the reference's expressions have them
// (eigen3_simple.hpp:47-266), which the tests pin against the compiled reference.
// ---------------------------------------------------------------------------------------------------------
namespace {

template <typename R>
void eig_roots3(const R m[3][3], R roots[3]) {   // eigen3_simple.hpp:47-82
  const R s_inv3 = (R)(1.0 / 3.0);
  const R s_sqrt3 = (R)std::sqrt(3.0);
  const R c0 = (R)(m[0][0] * m[1][1] * m[2][2] + 2.0 * m[1][0] * m[2][0] * m[2][1] - m[0][0] * m[2][1] * m[2][1] -
                   m[1][1] * m[2][0] * m[2][0] - m[2][2] * m[1][0] * m[1][0]);
  const R c1 = m[0][0] * m[1][1] - m[1][0] * m[1][0] + m[0][0] * m[2][2] - m[2][0] * m[2][0] + m[1][1] * m[2][2] -
               m[2][1] * m[2][1];
  const R c2 = m[0][0] + m[1][1] + m[2][2];
  const R c2_over_3 = c2 * s_inv3;
  R a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
  a_over_3 = std::max(a_over_3, (R)0.0);
  const R half_b = (R)(0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1)));
  R q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
  q = std::max(q, (R)0.0);
  const R rho = std::sqrt(a_over_3);
  const R theta = std::atan2(std::sqrt(q), half_b) * s_inv3;
  const R cos_theta = std::cos(theta), sin_theta = std::sin(theta);
  roots[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  roots[2] = (R)(c2_over_3 + 2.0 * rho * cos_theta);
}

template <typename R>
void cross3r(const R a[3], const R b[3], R d[3]) {
  d[2] = a[0] * b[1] - a[1] * b[0];
  d[0] = a[1] * b[2] - a[2] * b[1];
  d[1] = a[2] * b[0] - a[0] * b[2];
}
template <typename R>
void normalize3r(R a[3]) {   // lin3_utils.hpp:143-155
  R L = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  if (L > 0.0) {
    L = (R)(1.0 / L);
    for (int d = 0; d < 3; d++) a[d] *= L;
  } else { a[0] = 1.0; a[1] = 0.0; a[2] = 0.0; }
}

template <typename R>
void eig_kernel3(const R mat[3][3], R res[3], R rep[3]) {   // eigen3_simple.hpp:86-133
  int i0 = 0;
  R max_diag = std::abs(mat[0][0]);
  for (int d = 1; d < 3; d++)
    if (std::abs(mat[d][d]) > max_diag) { i0 = d; max_diag = std::abs(mat[d][d]); }
  R col[3][3];
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) col[c][r] = mat[r][c];
  for (int d = 0; d < 3; d++) rep[d] = col[i0][d];
  R c0[3], c1[3];
  cross3r(rep, col[(i0 + 1) % 3], c0);
  cross3r(rep, col[(i0 + 2) % 3], c1);
  const R n0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2], n1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
  if (n0 > n1) {
    const R s = (R)(1.0 / std::sqrt(n0));
    for (int d = 0; d < 3; d++) res[d] = c0[d] * s;
  } else {
    const R s = (R)(1.0 / std::sqrt(n1));
    for (int d = 0; d < 3; d++) res[d] = c1[d] * s;
  }
}

enum { EIG_INCREASING = 0, EIG_DECREASING = 1, EIG_INCREASING_ABS = 2, EIG_DECREASING_ABS = 3 };   // eigen3_simple.hpp:36-43

template <typename R>
void diagonalize_sym3(const R mat[3][3], R eivals[3], R eivects[3][3], int order) {   // eigen3_simple.hpp:137-266
  const R EPS = std::numeric_limits<R>::epsilon();
  const R shift = (R)((mat[0][0] + mat[1][1] + mat[2][2]) / 3.0);
  R S[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) S[i][j] = mat[i][j];
  for (int d = 0; d < 3; d++) S[d][d] -= shift;
  R scale = -1.0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      if (std::abs(S[i][j]) > scale) scale = std::abs(S[i][j]);
  if (scale > 0) {
    const R inv = (R)(1.0 / scale);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) S[i][j] *= inv;
  }
  eig_roots3(S, eivals);
  if ((eivals[2] - eivals[0]) <= EPS) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) eivects[i][j] = (i == j) ? (R)1.0 : (R)0.0;
  } else {
    R d0 = eivals[2] - eivals[1];
    const R d1 = eivals[1] - eivals[0];
    int k = 0, l = 2;
    if (d0 > d1) { d0 = d1; std::swap(k, l); }
    R T[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) T[i][j] = S[i][j];
    for (int d = 0; d < 3; d++) T[d][d] -= eivals[k];
    eig_kernel3(T, eivects[k], eivects[l]);
    if (d0 <= 2 * EPS * d1) {
      const R kl = eivects[k][0] * eivects[l][0] + eivects[k][1] * eivects[l][1] + eivects[k][2] * eivects[l][2];
      for (int d = 0; d < 3; d++) eivects[l][d] -= kl * eivects[l][d];   // (the reference updates with E[l] on both sides)
      normalize3r(eivects[l]);
    } else {
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[i][j] = S[i][j];
      for (int d = 0; d < 3; d++) T[d][d] -= eivals[l];
      R dummy[3];
      eig_kernel3(T, eivects[l], dummy);
    }
    cross3r(eivects[2], eivects[0], eivects[1]);
    normalize3r(eivects[1]);
  }
  for (int d = 0; d < 3; d++) { eivals[d] *= scale; eivals[d] += shift; }
  const bool swap = (order == EIG_INCREASING && eivals[0] > eivals[2]) || (order == EIG_DECREASING && eivals[0] < eivals[2]) ||
                    (order == EIG_INCREASING_ABS && std::abs(eivals[0]) > std::abs(eivals[2])) ||
                    (order == EIG_DECREASING_ABS && std::abs(eivals[0]) < std::abs(eivals[2]));
  if (swap) {
    std::swap(eivals[0], eivals[2]);
    for (int d = 0; d < 3; d++) std::swap(eivects[0][d], eivects[2][d]);
  }
}

inline float length3f(const float* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }   // visfd_utils.hpp:41-43

}  // namespace

// DiagonalizeSym3<float> (eigen3_simple.hpp:137-266): m9 = 3x3 row-major, eivects9 = eigenvectors as rows;
// order 0..3 = INCREASING, DECREASING, INCREASING_ABS, DECREASING_ABS eigenvalues.
extern "C" int visfd_hip_diagonalize_sym3_f32_host(const float* m9, int order, float* eivals3, float* eivects9) {
  if (!m9 || !eivals3 || !eivects9 || order < 0 || order > 3) return vh::fail(VISFD_HIP_EINVAL, "diagonalize_sym3: bad argument");
  float M[3][3], E[3][3];
  for (int i = 0; i < 9; i++) M[i / 3][i % 3] = m9[i];
  diagonalize_sym3<float>(M, eivals3, E, order);
  for (int i = 0; i < 9; i++) eivects9[i] = E[i / 3][i % 3];
  return VISFD_HIP_OK;
}

extern "C" int visfd_hip_surface_points(const float* saliency, const float* voxel2cluster, const float* direction,
                                        const float* mask, int64_t nx64, int64_t ny64, int64_t nz64, int select_cluster,
                                        const float voxel_width[3], float curve_ds, int find_ridge,
                                        float max_distance_to_feature, float* crds, float* norms, int64_t capacity,
                                        int64_t* n_points) {
  if (!saliency || !direction || !voxel_width || !n_points) return vh::fail(VISFD_HIP_EINVAL, "surface_points: null argument");
  VH_TRY(vh::check_dims(nx64, ny64, nz64));
  if (nx64 < 3 || ny64 < 3 || nz64 < 3) return vh::fail(VISFD_HIP_EINVAL, "surface_points: the image must be at least 3 voxels wide");
  if (nx64 * ny64 * nz64 >= (1LL << 31)) return vh::fail(VISFD_HIP_EINVAL, "surface_points: image too large");
  const Grid g = {(int)nx64, (int)ny64, (int)nz64};
  const float* S = saliency;
  const float* V = direction;
  const float* C = voxel2cluster;   // null: no clustering was done, every unmasked voxel is exported as is
  const int size[3] = {g.nx, g.ny, g.nz};
  int64_t count = 0;
  bool overflow = false;
  auto emit = [&](const float xyz[3], const float nrm[3]) {
    if (crds && norms && count < capacity)
      for (int d = 0; d < 3; d++) { crds[3 * count + d] = xyz[d]; norms[3 * count + d] = nrm[d]; }
    else if (crds || norms) overflow = true;
    count++;
  };
  auto inside = [&](const int p[3]) { return p[0] >= 0 && p[0] < g.nx && p[1] >= 0 && p[1] < g.ny && p[2] >= 0 && p[2] < g.nz; };
  std::vector<float> vS, vW, bS, bW;
  std::vector<std::array<float, 3> > vR, bR;
  for (int iz = 0; iz < g.nz; iz++)
    for (int iy = 0; iy < g.ny; iy++)
      for (int ix = 0; ix < g.nx; ix++) {
        const i64 c = g.at(ix, iy, iz);
        if (mask && mask[c] == 0.0f) continue;
        float xyz[3], normal[3];
        if (!C) {   // handlers.cpp:2056-2068
          xyz[0] = ix * voxel_width[0]; xyz[1] = iy * voxel_width[1]; xyz[2] = iz * voxel_width[2];
          for (int d = 0; d < 3; d++) normal[d] = V[3 * c + d];
          emit(xyz, normal);
          continue;
        }
        if ((float)select_cluster != C[c]) continue;
        xyz[0] = (float)ix; xyz[1] = (float)iy; xyz[2] = (float)iz;
        float norm = length3f(V + 3 * c);
        for (int d = 0; d < 3; d++) { normal[d] = V[3 * c + d] / norm; normal[d] *= S[c]; }
        if (curve_ds > 0.0f) {   // handlers.cpp:2098-2217
          const float ds = curve_ds;
          vS.clear(); vW.clear(); vR.clear(); bS.clear(); bW.clear(); bR.clear();
          float s = 0.0f, drds[3];
          std::array<float, 3> r = {{(float)ix, (float)iy, (float)iz}};
          int p[3] = {ix, iy, iz};
          while (inside(p) && (!mask || mask[g.at(p[0], p[1], p[2])] != 0.0f) && C[g.at(p[0], p[1], p[2])] == C[c]) {
            const i64 cp = g.at(p[0], p[1], p[2]);
            vS.push_back(s); vR.push_back(r); vW.push_back(S[cp]);
            norm = length3f(V + 3 * cp);
            for (int d = 0; d < 3; d++) drds[d] = V[3 * cp + d] / norm;
            s += ds;
            for (int d = 0; d < 3; d++) { r[d] += ds * drds[d]; p[d] = (int)std::round(r[d]); }
          }
          r = {{(float)ix, (float)iy, (float)iz}};
          p[0] = ix; p[1] = iy; p[2] = iz;
          s = 0.0f;
          for (;;) {
            const i64 cp = g.at(p[0], p[1], p[2]);
            norm = length3f(V + 3 * cp);
            for (int d = 0; d < 3; d++) drds[d] = V[3 * cp + d] / norm;
            s -= ds;
            for (int d = 0; d < 3; d++) { r[d] -= ds * drds[d]; p[d] = (int)std::round(r[d]); }
            if (!inside(p)) break;
            const i64 cq = g.at(p[0], p[1], p[2]);
            if (mask && mask[cq] == 0.0f) break;
            if (C[cq] != C[c]) break;
            bS.push_back(s); bR.push_back(r); bW.push_back(S[cq]);
          }
          vS.insert(vS.begin(), bS.rbegin(), bS.rend());
          vR.insert(vR.begin(), bR.rbegin(), bR.rend());
          vW.insert(vW.begin(), bW.rbegin(), bW.rend());
          float sum_s = 0.0f, sum_w = 0.0f;
          for (size_t i = 0; i < vS.size(); i++) { sum_s += vW[i] * vS[i]; sum_w += vW[i]; }
          const float ave_s = sum_s / sum_w;
          size_t i = 0;
          while (i + 1 < vS.size()) {
            i++;
            if (vS[i - 1] <= ave_s && ave_s <= vS[i]) break;
          }
          for (int d = 0; d < 3; d++) p[d] = (int)std::round(vR[i][d]);
          const i64 cp = g.at(p[0], p[1], p[2]);
          norm = length3f(V + 3 * cp);
          for (int d = 0; d < 3; d++) normal[d] = V[3 * cp + d] / norm;
          for (int d = 0; d < 3; d++) {
            if (i + 1 < vS.size()) xyz[d] = (vR[i][d] + (vR[i + 1][d] - vR[i][d]) * ((ave_s - vS[i]) / (vS[i + 1] - vS[i])));
            else xyz[d] = vR[i][d];
            normal[d] *= S[c];
          }
        }
        if (find_ridge) {   // handlers.cpp:2227-2296
          const int ix0 = (int)std::round(xyz[0]), iy0 = (int)std::round(xyz[1]), iz0 = (int)std::round(xyz[2]);
          float H[3][3], grad[3];
          hessian_fd(g, S, ix0, iy0, iz0, H);
          {   // visfd_utils.hpp:631-669: clamped central differences
            int gx = ix0, gy = iy0, gz = iz0;
            if (gx == 0) gx++; else if (gx == g.nx - 1) gx--;
            if (gy == 0) gy++; else if (gy == g.ny - 1) gy--;
            if (gz == 0) gz++; else if (gz == g.nz - 1) gz--;
            grad[0] = (float)(0.5 * (S[g.at(gx + 1, gy, gz)] - S[g.at(gx - 1, gy, gz)]));
            grad[1] = (float)(0.5 * (S[g.at(gx, gy + 1, gz)] - S[g.at(gx, gy - 1, gz)]));
            grad[2] = (float)(0.5 * (S[g.at(gx, gy, gz + 1)] - S[g.at(gx, gy, gz - 1)]));
          }
          float ev[3], E[3][3];
          diagonalize_sym3<float>(H, ev, E, EIG_DECREASING_ABS);
          float g1 = dot3(grad, E[0]);
          if (g1 < 0.0f) { g1 = -g1; for (int d = 0; d < 3; d++) E[0][d] = -E[0][d]; }
          else if (g1 == 0.0f) continue;
          const float dist = (ev[0] != 0) ? g1 / ev[0] : std::numeric_limits<float>::infinity();
          if (max_distance_to_feature > 0.0f && std::abs(dist) > max_distance_to_feature) continue;
          xyz[0] = ix0 - dist * E[0][0];
          xyz[1] = iy0 - dist * E[0][1];
          xyz[2] = iz0 - dist * E[0][2];
          if (xyz[0] < 0.0f || size[0] < xyz[0] || xyz[1] < 0.0f || size[1] < xyz[1] || xyz[2] < 0.0f || size[2] < xyz[2]) continue;
          for (int d = 0; d < 3; d++) xyz[d] *= voxel_width[d];
        }
        emit(xyz, normal);
      }
  *n_points = count;
  if (overflow) return vh::fail(VISFD_HIP_ECAPACITY, "surface_points: output arrays too small");
  return VISFD_HIP_OK;
}
