// gauss.hip -- separable 3-D filter (reference lib/visfd/filter3d.hpp:686-1050, ApplySeparable)
// and the DoG/LoG element-wise epilogue, for gfx950.
//
// Arithmetic contract, kept by every kernel in this file:
//   * pass order Z -> Y -> X, each pass rounded to float before the next;
//   * per output:  acc = 0; for j = -h..+h (ascending): if 0 <= i-j < n: acc += t[j]*f[i-j];
//     separate multiply and add (no FMA: the file is compiled with -ffp-contract=off);
//   * masked Z pass: w = t[j]*mask; acc += w*f; den += w;
//   * normalisation: unmasked  dst /= (Dx[ix]*Dy[iy])*Dz[iz];  masked  dst /= den where den > 0.
// The reference's "sparse input" shortcut (filter1d.hpp:59-94) writes exactly 0 when every
// source sample under the window is zero; the plain sum gives the same +0.0 for finite data, so
// only the masked Z pass (where the shortcut is keyed on the mask, not the data) restates it.
#include <cstdlib>

#include "common.hpp"

namespace vh {

namespace {

constexpr int BLOCK = 256;

// ---------------------------------------------------------------------------------------------
// Generic single-axis pass: one thread per output voxel, x-contiguous threads (coalesced for all
// three axes).  Used for the masked path, for very wide filters and as the fallback of the fused
// kernel below.  Re-reads of the 2h+1 neighbours are served by L1/L2.
// ---------------------------------------------------------------------------------------------
enum { NORM_NONE = 0, NORM_BOX = 1, NORM_DEN = 2 };

template <int AXIS, bool MASKED, int NORM>
__global__ void __launch_bounds__(BLOCK)
conv_axis_kernel(const float* __restrict__ in, float* __restrict__ out,
                 const float* __restrict__ mask, float* __restrict__ den_out,
                 const float* __restrict__ den_in,  // NORM_DEN: final denominator volume
                 const float* __restrict__ Dx, const float* __restrict__ Dy,
                 const float* __restrict__ Dz, i64 dz_offset,
                 Taps taps, i64 nx, i64 ny, i64 nz) {
  const i64 xblocks = (nx + BLOCK - 1) / BLOCK;
  i64 b = blockIdx.x;
  const i64 bx = b % xblocks;
  b /= xblocks;
  const i64 iy = b % ny;
  const i64 iz = b / ny;
  const i64 ix = bx * BLOCK + threadIdx.x;
  if (ix >= nx) return;
  const i64 plane = nx * ny;
  const i64 c = iz * plane + iy * nx + ix;
  const int h = taps.h;
  const i64 n = (AXIS == 0) ? nx : (AXIS == 1) ? ny : nz;
  const i64 i = (AXIS == 0) ? ix : (AXIS == 1) ? iy : iz;
  const i64 stride = (AXIS == 0) ? 1 : (AXIS == 1) ? nx : plane;
  float acc = 0.0f;
  float den = 0.0f;
  bool any = false;
  // j ascending  <=>  source index k = i - j descending
  int jlo = -h, jhi = h;
  if (i - jlo > n - 1) jlo = (int)(i - (n - 1));  // k <= n-1
  if (i - jhi < 0) jhi = (int)i;                  // k >= 0
  for (int j = jlo; j <= jhi; j++) {
    const i64 k = c - (i64)j * stride;
    float w = taps.t[j + h];
    if (MASKED) {
      const float m = mask[k];
      any = any || (m != 0.0f);
      w = w * m;
      den = den + w;
    }
    const float term = w * in[k];
    acc = acc + term;
  }
  if (MASKED) {
    if (!any) { acc = 0.0f; den = 0.0f; }
    if (den_out) den_out[c] = den;
  }
  if (NORM == NORM_BOX) {
    const float d = (Dx[ix] * Dy[iy]) * Dz[iz + dz_offset];
    acc = acc / d;
  } else if (NORM == NORM_DEN) {
    const float d = den_in[c];
    if (d > 0.0f) acc = acc / d;
  }
  out[c] = acc;
}

__global__ void __launch_bounds__(BLOCK)
sub_scale_kernel(float* __restrict__ a, const float* __restrict__ b, i64 n, float scale, int do_scale) {
  i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x;
  const i64 step = (i64)gridDim.x * BLOCK;
  for (; i < n; i += step) {
    float d = a[i] - b[i];
    if (do_scale) d = d * scale;
    a[i] = d;
  }
}

int fill_taps(Taps* T, const float* t, int h) {
  if (h < 0 || h > MAX_HALFWIDTH)
    return fail(VISFD_HIP_EINVAL, "filter halfwidth must be in [0, 64]");
  std::memset(T, 0, sizeof(Taps));
  T->h = h;
  for (int k = 0; k < 2 * h + 1; k++) T->t[k] = t[k];
  return VISFD_HIP_OK;
}

}  // namespace

// One translation unit per window half-width (gauss_fused.hip compiled with -DVH_FUSED_H=h).
#define VH_DECL_FUSED(HH)                                                                          \
  int launch_gauss_fused_h##HH(visfd_hip_ctx* ctx, const float* src, float* dst, i64 nx, i64 ny,   \
                               i64 nz, const Taps& tx, const Taps& ty, const Taps& tz,            \
                               const float* Dx, const float* Dy, const float* Dz, i64 dz_offset,  \
                               bool normalize, int cfg, const float* minuend, float log_scale);
VH_DECL_FUSED(1) VH_DECL_FUSED(2) VH_DECL_FUSED(3) VH_DECL_FUSED(4) VH_DECL_FUSED(5)
VH_DECL_FUSED(6) VH_DECL_FUSED(7) VH_DECL_FUSED(8) VH_DECL_FUSED(9) VH_DECL_FUSED(10)
#undef VH_DECL_FUSED

// The single-sweep kernel covers the unmasked case with equal half-widths 1..10 on the three axes
// (any sigma per axis), nx a multiple of 4, and planes below 2 GiB; everything else takes the 3-pass path.
static int dev_gauss_fused(visfd_hip_ctx* ctx, const float* src, float* dst, i64 nx, i64 ny, i64 nz,
                           const Taps& tx, const Taps& ty, const Taps& tz, const float* Dx,
                           const float* Dy, const float* Dz, i64 dz_offset, bool normalize,
                           const float* minuend, float log_scale, bool* handled) {
  *handled = false;
  const int H = tx.h;
  if (ty.h != H || tz.h != H || H < 1 || H > 10) return VISFD_HIP_OK;
  if ((nx & 3) || nx * ny >= (1LL << 29) || nz >= (1LL << 31)) return VISFD_HIP_OK;
  if (src == dst) return VISFD_HIP_OK;  // in place: 3-pass path through scratch volumes
  const char* force = getenv("VISFD_HIP_GAUSS_3PASS");
  if (force && force[0] == '1') return VISFD_HIP_OK;
  int cfg = 0;
  if (const char* e = getenv("VISFD_HIP_GAUSS_CFG")) cfg = atoi(e);
  *handled = true;
  switch (H) {
#define VH_CASE(HH) case HH: return launch_gauss_fused_h##HH(ctx, src, dst, nx, ny, nz, tx, ty, tz, Dx, Dy, Dz, dz_offset, normalize, cfg, minuend, log_scale);
    VH_CASE(1) VH_CASE(2) VH_CASE(3) VH_CASE(4) VH_CASE(5) VH_CASE(6) VH_CASE(7) VH_CASE(8) VH_CASE(9) VH_CASE(10)
#undef VH_CASE
  }
  *handled = false;
  return VISFD_HIP_OK;
}

int dev_separable3d(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask, i64 nx,
                    i64 ny, i64 nz, const float* tx, int hx, const float* ty, int hy,
                    const float* tz, int hz, bool normalize, SlabInfo slab, float* A_out,
                    const float* minuend, float log_scale, bool* epilogue_done) {
  if (epilogue_done) *epilogue_done = false;
  VH_TRY(check_dims(nx, ny, nz));
  Taps Tx, Ty, Tz;
  VH_TRY(fill_taps(&Tx, tx, hx));
  VH_TRY(fill_taps(&Ty, ty, hy));
  VH_TRY(fill_taps(&Tz, tz, hz));
  if (A_out) *A_out = (tx[hx] * ty[hy]) * tz[hz];  // filter3d.hpp:1044-1046
  const i64 n = nx * ny * nz;
  hipStream_t st = ctx->stream;
  const i64 xblocks = (nx + BLOCK - 1) / BLOCK;
  const i64 nblocks = xblocks * ny * nz;
  if (nblocks > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  const dim3 grid((unsigned)nblocks), block(BLOCK);

  // boundary normaliser lines (unmasked case): host arithmetic, a few KB
  float *Dx = nullptr, *Dy = nullptr, *Dz = nullptr;
  if (normalize && !mask) {
    float* D = nullptr;
    const i64 total = nx + ny + slab.nz_global;
    VH_TRY(ws(ctx, WS_NORM, (size_t)total, &D));
    std::vector<float> hD((size_t)total);
    host_conv_ones(nx, tx, hx, hD.data());
    host_conv_ones(ny, ty, hy, hD.data() + nx);
    host_conv_ones(slab.nz_global, tz, hz, hD.data() + nx + ny);
    // synchronous copy from pageable memory: the host vector may die right after this call
    VH_HIP(hipMemcpyAsync(D, hD.data(), sizeof(float) * (size_t)total, hipMemcpyHostToDevice, st));
    VH_HIP(hipStreamSynchronize(st));
    Dx = D; Dy = D + nx; Dz = D + nx + ny;
  }

  if (!mask) {
    bool handled = false;
    VH_TRY(dev_gauss_fused(ctx, src, dst, nx, ny, nz, Tx, Ty, Tz, Dx, Dy, Dz, slab.z_lo, normalize,
                           minuend, log_scale, &handled));
    if (handled) {
      if (epilogue_done) *epilogue_done = (minuend != nullptr);
      return VISFD_HIP_OK;
    }
  }
  // the DoG/LoG epilogue exists only in the single-sweep kernel: when it does not apply, nothing is
  // computed here (dst may alias the minuend) and the caller takes the two-volume route
  if (minuend) {
    if (!epilogue_done) return fail(VISFD_HIP_EINVAL, "internal: unfused epilogue request");
    return VISFD_HIP_OK;
  }

  float *A = nullptr, *B = nullptr;
  VH_TRY(ws(ctx, WS_A, (size_t)n, &A));
  VH_TRY(ws(ctx, WS_B, (size_t)n, &B));
  if (!mask) {
    conv_axis_kernel<2, false, NORM_NONE><<<grid, block, 0, st>>>(
        src, A, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, Tz, nx, ny, nz);
    conv_axis_kernel<1, false, NORM_NONE><<<grid, block, 0, st>>>(
        A, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, Ty, nx, ny, nz);
    if (normalize)
      conv_axis_kernel<0, false, NORM_BOX><<<grid, block, 0, st>>>(
          B, dst, nullptr, nullptr, nullptr, Dx, Dy, Dz, slab.z_lo, Tx, nx, ny, nz);
    else
      conv_axis_kernel<0, false, NORM_NONE><<<grid, block, 0, st>>>(
          B, dst, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, Tx, nx, ny, nz);
  } else if (!normalize) {
    conv_axis_kernel<2, true, NORM_NONE><<<grid, block, 0, st>>>(
        src, A, mask, nullptr, nullptr, nullptr, nullptr, nullptr, 0, Tz, nx, ny, nz);
    conv_axis_kernel<1, false, NORM_NONE><<<grid, block, 0, st>>>(
        A, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, Ty, nx, ny, nz);
    conv_axis_kernel<0, false, NORM_NONE><<<grid, block, 0, st>>>(
        B, dst, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, Tx, nx, ny, nz);
  } else {
    float *DA = nullptr, *DB = nullptr;
    VH_TRY(ws(ctx, WS_DEN_A, (size_t)n, &DA));
    VH_TRY(ws(ctx, WS_DEN_B, (size_t)n, &DB));
    conv_axis_kernel<2, true, NORM_NONE><<<grid, block, 0, st>>>(
        src, A, mask, DA, nullptr, nullptr, nullptr, nullptr, 0, Tz, nx, ny, nz);
    conv_axis_kernel<1, false, NORM_NONE><<<grid, block, 0, st>>>(
        A, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, Ty, nx, ny, nz);
    conv_axis_kernel<1, false, NORM_NONE><<<grid, block, 0, st>>>(
        DA, DB, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, Ty, nx, ny, nz);
    conv_axis_kernel<0, false, NORM_NONE><<<grid, block, 0, st>>>(
        DB, DA, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, Tx, nx, ny, nz);
    conv_axis_kernel<0, false, NORM_DEN><<<grid, block, 0, st>>>(
        B, dst, nullptr, nullptr, DA, nullptr, nullptr, nullptr, 0, Tx, nx, ny, nz);
  }
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

int dev_sub_scale(visfd_hip_ctx* ctx, float* a, const float* b, i64 n, float scale, bool do_scale) {
  const unsigned g = grid_for(n, BLOCK, (i64)ctx->num_cus * 16);
  sub_scale_kernel<<<dim3(g), dim3(BLOCK), 0, ctx->stream>>>(a, b, n, scale, do_scale ? 1 : 0);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

}  // namespace vh
