// blob.hip -- 4-D (x,y,z,scale) strict non-max / non-min scan of the scale-space blob detector
// (reference lib/visfd/feature.hpp:212-346).
//
// A voxel of the middle LoG volume is a minimum iff it is strictly smaller than all 80
// neighbours in (x,y,z,scale), all 26 spatial neighbours are inside the image and unmasked, the
// voxel itself is unmasked, its score is negative and passes the (absolute) threshold; maxima are
// symmetric (feature.hpp:231-304).  Float comparisons are exact, so indices match the reference
// bit-for-bit whenever the three LoG volumes do.
#include <algorithm>

#include "common.hpp"

namespace vh {

namespace {

constexpr int BLOCK = 256;

struct Cand {
  int ix, iy, iz;
  int kind;  // 0 = minimum, 1 = maximum
  float score;
};

__global__ void __launch_bounds__(BLOCK)
blob_scan_kernel(const float* __restrict__ lo, const float* __restrict__ mid,
                 const float* __restrict__ hi, const float* __restrict__ mask, int nx, int ny, int nz,
                 float min_thr, float max_thr, int want_min, int want_max,
                 Cand* __restrict__ out, unsigned long long capacity,
                 unsigned long long* __restrict__ counter) {
  // interior voxels only: a voxel on a face has an out-of-bounds neighbour (feature.hpp:245-252)
  const int wx = nx - 2;
  const int xblocks = (wx + BLOCK - 1) / BLOCK;
  unsigned b = blockIdx.x;
  const int bx = b % xblocks;
  b /= xblocks;
  const int iy = 1 + (int)(b % (unsigned)(ny - 2));
  const int iz = 1 + (int)(b / (unsigned)(ny - 2));
  const int ix = 1 + bx * BLOCK + (int)threadIdx.x;
  if (ix > nx - 2) return;
  const i64 plane = (i64)nx * ny;
  const i64 c = (i64)iz * plane + (i64)iy * nx + ix;
  if (mask && mask[c] == 0.0f) return;
  const float e = mid[c];
  bool is_min = want_min && (e < 0.0f) && (e < min_thr);
  bool is_max = want_max && (e > 0.0f) && (e > max_thr);
  if (!is_min && !is_max) return;
  const float* vol[3] = {mid, lo, hi};  // same-scale neighbours first: they reject most voxels
  for (int r = 0; r < 3 && (is_min || is_max); r++) {
    const float* v = vol[r];
    for (int jz = -1; jz <= 1; jz++)
      for (int jy = -1; jy <= 1; jy++) {
        const i64 row = c + (i64)jz * plane + (i64)jy * nx;
#pragma unroll
        for (int jx = -1; jx <= 1; jx++) {
          if (r == 0 && jz == 0 && jy == 0 && jx == 0) continue;
          const i64 q = row + jx;
          if (mask && r == 0 && mask[q] == 0.0f) { is_min = false; is_max = false; }
          const float nb = v[q];
          if (nb <= e) is_min = false;
          if (nb >= e) is_max = false;
        }
      }
  }
  if (is_min || is_max) {
    const unsigned long long slot = atomicAdd(counter, 1ULL);
    if (slot < capacity) {
      Cand cd;
      cd.ix = ix; cd.iy = iy; cd.iz = iz;
      cd.kind = is_min ? 0 : 1;
      cd.score = e;
      out[slot] = cd;
    }
  }
}

}  // namespace

int dev_blob_scan(visfd_hip_ctx* ctx, const float* lo, const float* mid, const float* hi,
                  const float* mask, i64 nx, i64 ny, i64 nz, int scale_index, float sigma,
                  float min_thr, float max_thr, bool want_min, bool want_max,
                  std::vector<visfd_hip_blob>* minima, std::vector<visfd_hip_blob>* maxima) {
  if (nx < 3 || ny < 3 || nz < 3) return VISFD_HIP_OK;  // no interior voxels: nothing can be a blob
  if (nx >= (1LL << 31) || ny >= (1LL << 31) || nz >= (1LL << 31))
    return fail(VISFD_HIP_EINVAL, "dimension too large");
  hipStream_t st = ctx->stream;
  unsigned long long* counter = nullptr;
  VH_TRY(ws(ctx, WS_COUNTER, 1, &counter));
  size_t capacity = ctx->slot_bytes[WS_CAND] / sizeof(Cand);
  if (capacity < (1u << 20)) capacity = 1u << 20;
  const i64 xblocks = (nx - 2 + BLOCK - 1) / BLOCK;
  const i64 nblocks = xblocks * (ny - 2) * (nz - 2);
  if (nblocks > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  for (int attempt = 0; attempt < 2; attempt++) {
    Cand* cand = nullptr;
    VH_TRY(ws(ctx, WS_CAND, capacity, &cand));
    VH_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned long long), st));
    blob_scan_kernel<<<dim3((unsigned)nblocks), dim3(BLOCK), 0, st>>>(
        lo, mid, hi, mask, (int)nx, (int)ny, (int)nz, min_thr, max_thr, want_min ? 1 : 0,
        want_max ? 1 : 0, cand, (unsigned long long)capacity, counter);
    VH_HIP(hipGetLastError());
    unsigned long long count = 0;
    VH_HIP(hipMemcpyAsync(&count, counter, sizeof(count), hipMemcpyDeviceToHost, st));
    VH_HIP(hipStreamSynchronize(st));
    if (count > capacity) {  // rare: grow once to the exact size and rescan
      capacity = (size_t)count;
      continue;
    }
    std::vector<Cand> h((size_t)count);
    if (count) {
      VH_HIP(hipMemcpyAsync(h.data(), cand, sizeof(Cand) * (size_t)count, hipMemcpyDeviceToHost, st));
      VH_HIP(hipStreamSynchronize(st));
    }
    // deterministic order (the device appends in arrival order): by (iz, iy, ix)
    std::sort(h.begin(), h.end(), [](const Cand& a, const Cand& b) {
      if (a.iz != b.iz) return a.iz < b.iz;
      if (a.iy != b.iy) return a.iy < b.iy;
      return a.ix < b.ix;
    });
    for (const Cand& cd : h) {
      visfd_hip_blob bl;
      bl.ix = cd.ix; bl.iy = cd.iy; bl.iz = cd.iz;
      bl.scale = scale_index;
      bl.sigma = sigma;
      bl.score = cd.score;
      (cd.kind == 0 ? minima : maxima)->push_back(bl);
    }
    return VISFD_HIP_OK;
  }
  return fail(VISFD_HIP_EDEVICE, "blob candidate list kept overflowing");
}

}  // namespace vh
