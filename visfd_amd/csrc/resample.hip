// resample.hip -- binning / un-binning of volumes (reference lib/visfd/resample.hpp:53-166).
//
// BinArray3D: every destination voxel is the float sum of its bx*by*bz source voxels accumulated in
// dz, dy, dx order from 0.0f, divided by the (integer) bin volume converted to float -- the same
// adds in the same order, one IEEE divide, so results are bit-identical.  Source voxels beyond
// size_dest*bin are dropped.  UnbinArray3D is a nearest-lower gather with clamping.
// Both are pure HBM streaming kernels: bin reads 4 B per source voxel (each lane reads bx
// consecutive floats per row, so wavefront reads stay contiguous), unbin writes 4 B per voxel.
#include "common.hpp"

namespace vh {

namespace {

constexpr int BLOCK = 256;

template <int B>   // B > 0: cubic bins of that width with a fully unrolled sum; 0: general
__global__ void __launch_bounds__(BLOCK)
bin_kernel(const float* __restrict__ src, float* __restrict__ dst, int snx, int sny, int dnx, int dny, int dnz,
           int bx, int by, int bz, int ox, int oy, int oz) {
  const i64 n = (i64)dnx * dny * dnz;
  const float denom = (float)(B > 0 ? B * B * B : bx * by * bz);
  for (i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) {
    const int Ix = (int)(i % dnx);
    const i64 r = i / dnx;
    const int Iy = (int)(r % dny), Iz = (int)(r / dny);
    float sum = 0.0f;
    if (B > 0) {
      const float* p = src + ((i64)(Iz * B + oz) * sny + (Iy * B + oy)) * snx + (Ix * B + ox);
#pragma unroll
      for (int dz = 0; dz < B; dz++)
#pragma unroll
        for (int dy = 0; dy < B; dy++)
#pragma unroll
          for (int dx = 0; dx < B; dx++) sum = sum + p[((i64)dz * sny + dy) * snx + dx];
    } else {
      const float* p = src + ((i64)(Iz * bz + oz) * sny + (Iy * by + oy)) * snx + (Ix * bx + ox);
      for (int dz = 0; dz < bz; dz++)
        for (int dy = 0; dy < by; dy++)
          for (int dx = 0; dx < bx; dx++) sum = sum + p[((i64)dz * sny + dy) * snx + dx];
    }
    dst[i] = sum / denom;
  }
}

__global__ void __launch_bounds__(BLOCK)
unbin_kernel(const float* __restrict__ src, float* __restrict__ dst, int snx, int sny, int snz, int dnx, int dny,
             int dnz, int bx, int by, int bz, int ox, int oy, int oz) {
  const i64 n = (i64)dnx * dny * dnz;
  for (i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) {
    const int Ix = (int)(i % dnx);
    const i64 r = i / dnx;
    const int Iy = (int)(r % dny), Iz = (int)(r / dny);
    // C++ integer division truncates toward zero (resample.hpp:150-152), then clamp (:153-158)
    int ix = (Ix - ox) / bx, iy = (Iy - oy) / by, iz = (Iz - oz) / bz;
    ix = ix < 0 ? 0 : (ix >= snx ? snx - 1 : ix);
    iy = iy < 0 ? 0 : (iy >= sny ? sny - 1 : iy);
    iz = iz < 0 ? 0 : (iz >= snz ? snz - 1 : iz);
    dst[i] = src[((i64)iz * sny + iy) * snx + ix];
  }
}

int bins_of(const int64_t big[3], const int64_t small_[3], const int* offset, int bin[3], int off[3],
            bool reads_big) {
  for (int d = 0; d < 3; d++) {
    if (big[d] <= 0 || small_[d] <= 0) return fail(VISFD_HIP_EINVAL, "image dimensions must be positive");
    if (big[d] >= (1LL << 31) || small_[d] >= (1LL << 31)) return fail(VISFD_HIP_EINVAL, "image dimension too large");
    bin[d] = (int)(big[d] / small_[d]);
    if (bin[d] < 1) return fail(VISFD_HIP_EINVAL, "the binned image cannot be larger than the full-size image");
    off[d] = offset ? offset[d] : 0;
    if (off[d] < 0 || off[d] >= bin[d])   // resample.hpp:63-69, :129-135
      return fail(VISFD_HIP_EINVAL, "bin offset must lie between 0 and floor(size_big / size_small) - 1");
    // the reference only asserts this (resample.hpp:85-87); reading past the source is refused here
    if (reads_big && small_[d] * bin[d] + off[d] > big[d])
      return fail(VISFD_HIP_EINVAL, "bin offset moves the binning window outside the source image");
  }
  return VISFD_HIP_OK;
}

}  // namespace

int dev_bin_array3d(visfd_hip_ctx* ctx, const float* src, const int64_t ssz[3], float* dst, const int64_t dsz[3],
                    const int* offset) {
  int b[3], o[3];
  VH_TRY(bins_of(ssz, dsz, offset, b, o, true));
  const i64 n = dsz[0] * dsz[1] * dsz[2];
  const unsigned g = grid_for(n, BLOCK, (i64)ctx->num_cus * 64);
#define VH_BIN(BB)                                                                                         \
  bin_kernel<BB><<<dim3(g), dim3(BLOCK), 0, ctx->stream>>>(src, dst, (int)ssz[0], (int)ssz[1], (int)dsz[0], \
                                                          (int)dsz[1], (int)dsz[2], b[0], b[1], b[2], o[0], o[1], o[2])
  const bool cubic = b[0] == b[1] && b[1] == b[2];
  if (cubic && b[0] == 1) VH_BIN(1);
  else if (cubic && b[0] == 2) VH_BIN(2);
  else if (cubic && b[0] == 3) VH_BIN(3);
  else if (cubic && b[0] == 4) VH_BIN(4);
  else VH_BIN(0);
#undef VH_BIN
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

int dev_unbin_array3d(visfd_hip_ctx* ctx, const float* src, const int64_t ssz[3], float* dst, const int64_t dsz[3],
                      const int* offset) {
  int b[3], o[3];
  VH_TRY(bins_of(dsz, ssz, offset, b, o, false));
  const i64 n = dsz[0] * dsz[1] * dsz[2];
  const unsigned g = grid_for(n, BLOCK, (i64)ctx->num_cus * 64);
  unbin_kernel<<<dim3(g), dim3(BLOCK), 0, ctx->stream>>>(src, dst, (int)ssz[0], (int)ssz[1], (int)ssz[2], (int)dsz[0],
                                                         (int)dsz[1], (int)dsz[2], b[0], b[1], b[2], o[0], o[1], o[2]);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

}  // namespace vh
