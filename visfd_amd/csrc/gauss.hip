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

// Outputs that this kernel does not read again are stored non-temporally: stores that allocate in L2 push out the rows
// neighbouring workgroups share (measured on the single sweep: csrc/gauss_fused.hip).
#define VH_STREAM_STORE(v, p) __builtin_nontemporal_store((v), (p))

namespace vh {

namespace {

constexpr int BLOCK = 256;

enum { NORM_NONE = 0, NORM_BOX = 1, NORM_DEN = 2 };

// ---------------------------------------------------------------------------------------------
// Tuned single-axis passes (one read + one write of the volume each, 8 B/voxel): the source
// window is staged in LDS, so HBM sees every input once per tile (halo overhead (T+2h)/T) and the
// 2h+1 re-reads per output are LDS reads.  HT > 0 fixes the half-width at compile time (taps as
// SGPR operands, fully unrolled sums); HT == 0 takes any half-width up to MAX_HALFWIDTH.
//
// conv_march_kernel: the Y or Z pass.  A block of 256 threads = 64 x-columns x 4 groups handles a
// tile of 64 columns x MARCH_T outputs along the filtered ("march") axis; LDS holds
// (MARCH_T + 2h) x 64 source values (rows outside the image are zeros: zero extension).  The
// masked form (Z pass of the masked filter, filter1d.hpp:204-295) also stages the mask, forms
// w = t*mask first, then w*f, accumulates the denominator, and restates the "no unmasked sample
// under the window" shortcut.
// ---------------------------------------------------------------------------------------------
constexpr int MARCH_T = 128;   // outputs per block along the filtered axis
constexpr int MARCH_X = 64;    // x-columns per block

template <int HT>
struct TapsK {
  float t[2 * (HT > 0 ? HT : 1) + 1];
};

template <int HT, bool MASKED>
__global__ void __launch_bounds__(BLOCK)
conv_march_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ mask,
                  float* __restrict__ den_out, TapsK<HT> tk, Taps taps_rt, int nx, int n_march, int n_other,
                  i64 stride_march, i64 stride_other, int xblocks, int mblocks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int h = HT > 0 ? HT : taps_rt.h;
  const int W = 2 * h + 1;
  const int rows = MARCH_T + 2 * h;
  float* sF = lds;
  float* sM = lds + rows * MARCH_X;   // masked form only
  unsigned b = blockIdx.x;
  const int bx = b % xblocks;
  b /= xblocks;
  const int bm = b % mblocks;
  const int o = b / mblocks;
  const int tx = threadIdx.x & (MARCH_X - 1), g = threadIdx.x / MARCH_X;
  const int x = bx * MARCH_X + tx;
  const int m0 = bm * MARCH_T;
  const bool xin = x < nx;
  const i64 base = (i64)o * stride_other + x;
  // stage rows m0-h .. m0+MARCH_T+h-1 (LDS row r <-> m = m0 - h + r): 16 threads x float4 per row, 16 rows per
  // sweep, all loads of a batch issued before their LDS stores (the loop is latency-bound otherwise)
  {
    const int c4 = (threadIdx.x & 15) * 4, rr = threadIdx.x >> 4;
    const int xs = bx * MARCH_X + c4;
    const bool vec_ok = ((nx & 3) == 0) && xs + 3 < nx;   // whole float4 inside the row and 16-byte aligned
    constexpr int SWEEP = BLOCK / 16;                     // rows per sweep
    constexpr int BATCH = 5;                              // sweeps in flight
    for (int r0 = 0; r0 < rows; r0 += SWEEP * BATCH) {
      float4 vf[BATCH], vm[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; u++) {
        const int r = r0 + u * SWEEP + rr;
        const int m = m0 - h + r;
        const bool ok = r < rows && m >= 0 && m < n_march;
        const i64 a = (i64)o * stride_other + (i64)m * stride_march + xs;
        vf[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        vm[u] = vf[u];
        if (ok && vec_ok) {
          vf[u] = *reinterpret_cast<const float4*>(in + a);
          if (MASKED) vm[u] = *reinterpret_cast<const float4*>(mask + a);
        } else if (ok) {
          float* pf = &vf[u].x;
          float* pm = &vm[u].x;
          for (int e = 0; e < 4; e++)
            if (xs + e < nx) { pf[e] = in[a + e]; if (MASKED) pm[e] = mask[a + e]; }
        }
      }
#pragma unroll
      for (int u = 0; u < BATCH; u++) {
        const int r = r0 + u * SWEEP + rr;
        if (r < rows) {
          *reinterpret_cast<float4*>(&sF[r * MARCH_X + c4]) = vf[u];
          if (MASKED) *reinterpret_cast<float4*>(&sM[r * MARCH_X + c4]) = vm[u];
        }
      }
    }
  }
  __syncthreads();
  if (!xin) return;
  constexpr int PER = MARCH_T / (BLOCK / MARCH_X);   // consecutive outputs per thread
  for (int q = 0; q < PER; q++) {
    const int k = g * PER + q;
    const int m = m0 + k;
    if (m >= n_march) break;
    // j ascending <=> source row descending: r = k + 2h - jj
    const float* pf = sF + (k + 2 * h) * MARCH_X + tx;
    const float* pm = sM + (k + 2 * h) * MARCH_X + tx;
    float acc = 0.0f, den = 0.0f;
    bool any = false;
    if (HT > 0) {
#pragma unroll
      for (int jj = 0; jj < 2 * HT + 1; jj++) {
        float w = tk.t[jj];
        if (MASKED) {
          const float mv = pm[-jj * MARCH_X];
          any = any || (mv != 0.0f);
          w = w * mv;
          den = den + w;
        }
        const float term = w * pf[-jj * MARCH_X];
        acc = acc + term;
      }
    } else {
      for (int jj = 0; jj < W; jj++) {
        float w = taps_rt.t[jj];
        if (MASKED) {
          const float mv = pm[-jj * MARCH_X];
          any = any || (mv != 0.0f);
          w = w * mv;
          den = den + w;
        }
        const float term = w * pf[-jj * MARCH_X];
        acc = acc + term;
      }
    }
    const i64 c = base + (i64)m * stride_march;
    if (MASKED) {
      if (!any) { acc = 0.0f; den = 0.0f; }
      if (den_out) den_out[c] = den;
    }
    VH_STREAM_STORE(acc, &out[c]);
  }
}

// conv_row_kernel: the X pass.  A block handles ROW_X consecutive x of ROW_Y rows (same z); LDS
// holds the ROW_Y x (ROW_X + 2h) source values; one thread per x computes the ROW_Y outputs of
// its column position.  NORM_BOX / NORM_DEN fold the normalisation of filter3d.hpp:986-1026 in, and an
// optional minuend turns the store into the DoG/LoG result (minuend - G) * scale.
constexpr int ROW_X = 256;
constexpr int ROW_Y = 8;
static_assert(ROW_X == BLOCK && 2 * MAX_HALFWIDTH <= BLOCK, "the row tile is staged with two loads per thread");

template <int HT, int NORM>
__global__ void __launch_bounds__(BLOCK)
conv_row_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ den_in,
                const float* __restrict__ Dx, const float* __restrict__ Dy, const float* __restrict__ Dz,
                i64 dz_offset, TapsK<HT> tk, Taps taps_rt, int nx, int ny, int nz, int xblocks, int yblocks,
                const float* __restrict__ minuend, float log_scale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int h = HT > 0 ? HT : taps_rt.h;
  const int W = 2 * h + 1;
  const int span = ROW_X + 2 * h;      // LDS row length; LDS index s <-> x = x0 - h + s
  unsigned b = blockIdx.x;
  const int bx = b % xblocks;
  b /= xblocks;
  const int by = b % yblocks;
  const int iz = b / yblocks;
  const int x0 = bx * ROW_X, y0 = by * ROW_Y;
  const int t = threadIdx.x;
  const i64 plane = (i64)nx * ny;
  {  // all loads of the tile issued before the LDS stores: per row a main element and (first 2h threads) a tail one
    float va[ROW_Y], vb[ROW_Y];
#pragma unroll
    for (int r = 0; r < ROW_Y; r++) {
      const int y = y0 + r;
      const float* row = in + (i64)iz * plane + (i64)y * nx;
      const int xa = x0 - h + t, xb = xa + BLOCK;
      va[r] = (y < ny && xa >= 0 && xa < nx) ? row[xa] : 0.0f;
      vb[r] = (t < 2 * h && y < ny && xb >= 0 && xb < nx) ? row[xb] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < ROW_Y; r++) {
      lds[r * span + t] = va[r];
      if (t < 2 * h) lds[r * span + t + BLOCK] = vb[r];
    }
  }
  __syncthreads();
  const int x = x0 + t;
  if (x >= nx) return;
  float dxz = 1.0f;
  if (NORM == NORM_BOX) dxz = Dx[x];
  for (int r = 0; r < ROW_Y; r++) {
    const int y = y0 + r;
    if (y >= ny) break;
    const float* p = lds + r * span + t + 2 * h;   // j ascending <=> s descending: s = t + 2h - jj
    float acc = 0.0f;
    if (HT > 0) {
#pragma unroll
      for (int jj = 0; jj < 2 * HT + 1; jj++) {
        const float term = tk.t[jj] * p[-jj];
        acc = acc + term;
      }
    } else {
      for (int jj = 0; jj < W; jj++) {
        const float term = taps_rt.t[jj] * p[-jj];
        acc = acc + term;
      }
    }
    const i64 c = (i64)iz * plane + (i64)y * nx + x;
    if (NORM == NORM_BOX) {
      const float d = (dxz * Dy[y]) * Dz[iz + dz_offset];
      acc = acc / d;
    } else if (NORM == NORM_DEN) {
      const float d = den_in[c];
      if (d > 0.0f) acc = acc / d;
    }
    if (minuend) {   // DoG/LoG epilogue (filter3d.hpp:1387-1390, :1495-1498): two roundings; out may alias minuend
      const float dd = minuend[c] - acc;
      acc = dd * log_scale;
    }
    VH_STREAM_STORE(acc, &out[c]);
  }
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

// LocalFluctuations, filter3d.hpp:1776-1790: P = source - average; P *= P (two roundings)
__global__ void __launch_bounds__(BLOCK)
sub_square_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, i64 n) {
  i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x;
  const i64 step = (i64)gridDim.x * BLOCK;
  for (; i < n; i += step) {
    const float d = a[i] - b[i];
    out[i] = d * d;
  }
}

// LocalFluctuations, filter3d.hpp:1819-1846: variance *= wpeak; negative -> 0; sqrt (correctly rounded)
__global__ void __launch_bounds__(BLOCK)
scale_clamp_sqrt_kernel(float* __restrict__ a, i64 n, float scale) {
  i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x;
  const i64 step = (i64)gridDim.x * BLOCK;
  for (; i < n; i += step) {
    float v = a[i] * scale;
    if (v < 0.0f) v = 0.0f;
    a[i] = sqrtf(v);   // correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt; __fsqrt_rn is not)
  }
}

// Boundary normaliser lines Dx | Dy | Dz (filter3d.hpp:1004-1021): each axis filter applied to a line of ones of the
// global length -- acc = 0; acc += t[j] * 1 for the taps whose sample lies inside, j ascending (host_conv_ones).
__global__ void __launch_bounds__(BLOCK)
norm_lines_kernel(float* __restrict__ D, i64 nx, i64 ny, i64 nz, Taps tx, Taps ty, Taps tz) {
  const i64 g = (i64)blockIdx.x * BLOCK + threadIdx.x;
  if (g >= nx + ny + nz) return;
  const Taps& T = g < nx ? tx : (g < nx + ny ? ty : tz);
  const i64 n = g < nx ? nx : (g < nx + ny ? ny : nz);
  const i64 i = g < nx ? g : (g < nx + ny ? g - nx : g - nx - ny);
  float acc = 0.0f;
  for (int j = -T.h; j <= T.h; j++) {
    const i64 k = i - j;
    if (k < 0 || k >= n) continue;
    acc += T.t[j + T.h] * 1.0f;
  }
  D[g] = acc;
}

int fill_taps(Taps* T, const float* t, int h) {
  if (h < 0 || h > MAX_HALFWIDTH)
    return fail(VISFD_HIP_EINVAL, "filter halfwidth must be in [0, 64]");
  std::memset(T, 0, sizeof(Taps));
  T->h = h;
  for (int k = 0; k < 2 * h + 1; k++) T->t[k] = t[k];
  return VISFD_HIP_OK;
}

template <int HT>
TapsK<HT> taps_k(const Taps& T) {
  TapsK<HT> k;
  for (int i = 0; i < 2 * (HT > 0 ? HT : 1) + 1; i++) k.t[i] = HT > 0 ? T.t[i] : 0.0f;
  return k;
}

#define VH_FOR_H(h, ...)                                                                            \
  switch (h) {                                                                                      \
    case 1: { constexpr int HT = 1; __VA_ARGS__; } break;   case 2: { constexpr int HT = 2; __VA_ARGS__; } break;   \
    case 3: { constexpr int HT = 3; __VA_ARGS__; } break;   case 4: { constexpr int HT = 4; __VA_ARGS__; } break;   \
    case 5: { constexpr int HT = 5; __VA_ARGS__; } break;   case 6: { constexpr int HT = 6; __VA_ARGS__; } break;   \
    case 7: { constexpr int HT = 7; __VA_ARGS__; } break;   case 8: { constexpr int HT = 8; __VA_ARGS__; } break;   \
    case 9: { constexpr int HT = 9; __VA_ARGS__; } break;   case 10: { constexpr int HT = 10; __VA_ARGS__; } break; \
    default: { constexpr int HT = 0; __VA_ARGS__; } break;                                                 \
  }

// Y pass (axis 1) or Z pass (axis 2) of a [nz][ny][nx] volume
template <bool MASKED>
int launch_march(visfd_hip_ctx* ctx, int axis, const float* in, float* out, const float* mask, float* den_out,
                 const Taps& T, i64 nx, i64 ny, i64 nz) {
  const i64 plane = nx * ny;
  const int n_march = (int)(axis == 1 ? ny : nz), n_other = (int)(axis == 1 ? nz : ny);
  const i64 stride_march = axis == 1 ? nx : plane, stride_other = axis == 1 ? plane : nx;
  const int xblocks = (int)((nx + MARCH_X - 1) / MARCH_X), mblocks = (n_march + MARCH_T - 1) / MARCH_T;
  const i64 nblk = (i64)xblocks * mblocks * n_other;
  if (nblk > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  const size_t lds = sizeof(float) * (size_t)(MARCH_T + 2 * T.h) * MARCH_X * (MASKED ? 2 : 1);
  VH_FOR_H(T.h, {
    auto kern = conv_march_kernel<HT, MASKED>;
    if (lds > 48 * 1024)
      VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3((unsigned)nblk), dim3(BLOCK), lds, ctx->stream>>>(in, out, mask, den_out, taps_k<HT>(T), T, (int)nx, n_march,
                                                                  n_other, stride_march, stride_other, xblocks, mblocks);
  })
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

template <int NORM>
int launch_row(visfd_hip_ctx* ctx, const float* in, float* out, const float* den_in, const float* Dx, const float* Dy,
               const float* Dz, i64 dz_offset, const Taps& T, i64 nx, i64 ny, i64 nz, const float* minuend = nullptr,
               float log_scale = 1.0f) {
  const int xblocks = (int)((nx + ROW_X - 1) / ROW_X), yblocks = (int)((ny + ROW_Y - 1) / ROW_Y);
  const i64 nblk = (i64)xblocks * yblocks * nz;
  if (nblk > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  const size_t lds = sizeof(float) * (size_t)ROW_Y * (ROW_X + 2 * T.h);
  VH_FOR_H(T.h, {
    conv_row_kernel<HT, NORM><<<dim3((unsigned)nblk), dim3(BLOCK), lds, ctx->stream>>>(
        in, out, den_in, Dx, Dy, Dz, dz_offset, taps_k<HT>(T), T, (int)nx, (int)ny, (int)nz, xblocks, yblocks, minuend,
        log_scale);
  })
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

}  // namespace

// One translation unit per window half-width (gauss_fused.hip compiled with -DVH_FUSED_H=h).
#define VH_DECL_FUSED(HH)                                                                          \
  int launch_gauss_fused_h##HH(visfd_hip_ctx* ctx, const float* src, float* dst, i64 nx, i64 ny,   \
                               i64 nz, const Taps& tx, const Taps& ty, const Taps& tz,            \
                               const float* Dx, const float* Dy, const float* Dz, i64 dz_offset,  \
                               bool normalize, int cfg, const float* minuend, float log_scale, bool fma);
VH_DECL_FUSED(1) VH_DECL_FUSED(2) VH_DECL_FUSED(3) VH_DECL_FUSED(4) VH_DECL_FUSED(5)
VH_DECL_FUSED(6) VH_DECL_FUSED(7) VH_DECL_FUSED(8)
#undef VH_DECL_FUSED
#define VH_DECL_FUSED_YX(HH)                                                                       \
  int launch_gauss_fused_yx_h##HH(visfd_hip_ctx* ctx, const float* src, float* dst, i64 nx, i64 ny, \
                                  i64 nz, const Taps& tx, const Taps& ty, const float* numer,       \
                                  const float* minuend, float log_scale);
VH_DECL_FUSED_YX(1) VH_DECL_FUSED_YX(2) VH_DECL_FUSED_YX(3) VH_DECL_FUSED_YX(4) VH_DECL_FUSED_YX(5)
VH_DECL_FUSED_YX(6) VH_DECL_FUSED_YX(7) VH_DECL_FUSED_YX(8)
#undef VH_DECL_FUSED_YX

// The single-sweep kernel covers the unmasked case with equal half-widths 1..8 on the three axes (any sigma
// per axis) and planes below 2 GiB; everything else takes the 3-pass path.  (Beyond h = 8
// the register ring no longer fits 128 VGPRs: the spilling single-sweep kernels ran at 10 and 16 ms for
// h = 9 and 10 at 1024^3, three bandwidth-bound passes take 5.3 ms.)
static int dev_gauss_fused(visfd_hip_ctx* ctx, const float* src, float* dst, i64 nx, i64 ny, i64 nz,
                           const Taps& tx, const Taps& ty, const Taps& tz, const float* Dx,
                           const float* Dy, const float* Dz, i64 dz_offset, bool normalize,
                           const float* minuend, float log_scale, bool fma, bool* handled) {
  *handled = false;
  const int H = tx.h;
  if (ty.h != H || tz.h != H || H < 1 || H > 8) return VISFD_HIP_OK;
  if (nx * ny >= (1LL << 29) || nz >= (1LL << 31)) return VISFD_HIP_OK;
  if (src == dst) return VISFD_HIP_OK;  // in place: 3-pass path through scratch volumes
  // the single-sweep kernel's Z pass shares the product of a sample with the taps +j and -j, and all three passes
  // keep only the taps 0..H (in VGPRs): the taps must be symmetric bit for bit on every axis (every Gaussian is;
  // arbitrary filters take the 3-pass path)
  for (int j = 1; j <= H; j++)
    if (std::memcmp(&tz.t[H + j], &tz.t[H - j], sizeof(float)) != 0 || std::memcmp(&ty.t[H + j], &ty.t[H - j], sizeof(float)) != 0 ||
        std::memcmp(&tx.t[H + j], &tx.t[H - j], sizeof(float)) != 0)
      return VISFD_HIP_OK;
  if (ctx->opt.gauss_3pass) return VISFD_HIP_OK;
  const int cfg = ctx->opt.gauss_cfg;
  *handled = true;
  switch (H) {
#define VH_CASE(HH) case HH: return launch_gauss_fused_h##HH(ctx, src, dst, nx, ny, nz, tx, ty, tz, Dx, Dy, Dz, dz_offset, normalize, cfg, minuend, log_scale, fma);
    VH_CASE(1) VH_CASE(2) VH_CASE(3) VH_CASE(4) VH_CASE(5) VH_CASE(6) VH_CASE(7) VH_CASE(8)
#undef VH_CASE
  }
  *handled = false;
  return VISFD_HIP_OK;
}

// Y and X passes in one sweep (the masked filter after its Z pass): equal half-widths 1..8 in x and y, planes
// below 2 GiB; numer != null selects the masked-normalisation epilogue.
static int dev_gauss_fused_yx(visfd_hip_ctx* ctx, const float* src, float* dst, i64 nx, i64 ny, i64 nz, const Taps& tx,
                              const Taps& ty, const float* numer, const float* minuend, float log_scale,
                              bool* handled) {
  *handled = false;
  const int H = tx.h;
  if (ty.h != H || H < 1 || H > 8) return VISFD_HIP_OK;
  if (nx * ny >= (1LL << 29) || nz >= (1LL << 31)) return VISFD_HIP_OK;
  if (src == dst || numer == dst) return VISFD_HIP_OK;
  for (int j = 1; j <= H; j++)   // the kernel keeps the taps 0..H only
    if (std::memcmp(&ty.t[H + j], &ty.t[H - j], sizeof(float)) != 0 || std::memcmp(&tx.t[H + j], &tx.t[H - j], sizeof(float)) != 0)
      return VISFD_HIP_OK;
  if (ctx->opt.gauss_3pass) return VISFD_HIP_OK;
  *handled = true;
  switch (H) {
#define VH_CASE(HH) case HH: return launch_gauss_fused_yx_h##HH(ctx, src, dst, nx, ny, nz, tx, ty, numer, minuend, log_scale);
    VH_CASE(1) VH_CASE(2) VH_CASE(3) VH_CASE(4) VH_CASE(5) VH_CASE(6) VH_CASE(7) VH_CASE(8)
#undef VH_CASE
  }
  *handled = false;
  return VISFD_HIP_OK;
}

int dev_separable3d(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask, i64 nx,
                    i64 ny, i64 nz, const float* tx, int hx, const float* ty, int hy,
                    const float* tz, int hz, bool normalize, SlabInfo slab, float* A_out,
                    const float* minuend, float log_scale, bool* epilogue_done, bool fma) {
  if (epilogue_done) *epilogue_done = false;
  VH_TRY(check_dims(nx, ny, nz));
  Taps Tx, Ty, Tz;
  VH_TRY(fill_taps(&Tx, tx, hx));
  VH_TRY(fill_taps(&Ty, ty, hy));
  VH_TRY(fill_taps(&Tz, tz, hz));
  if (A_out) *A_out = (tx[hx] * ty[hy]) * tz[hz];  // filter3d.hpp:1044-1046
  const i64 n = nx * ny * nz;
  hipStream_t st = ctx->stream;

  // boundary normaliser lines (unmasked case): host arithmetic, a few KB
  float *Dx = nullptr, *Dy = nullptr, *Dz = nullptr;
  if (normalize && !mask) {
    // filled on the device (same float sums as host_conv_ones): no host copy, no stream synchronisation --
    // a dozen Gaussians in a row (blob detection) stay queued back to back
    float* D = nullptr;
    const i64 total = nx + ny + slab.nz_global;
    VH_TRY(ws(ctx, WS_NORM, (size_t)total, &D));
    norm_lines_kernel<<<dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st>>>(D, nx, ny, slab.nz_global,
                                                                                            Tx, Ty, Tz);
    VH_HIP(hipGetLastError());
    Dx = D; Dy = D + nx; Dz = D + nx + ny;
  }

  if (!mask) {
    bool handled = false;
    VH_TRY(dev_gauss_fused(ctx, src, dst, nx, ny, nz, Tx, Ty, Tz, Dx, Dy, Dz, slab.z_lo, normalize,
                           minuend, log_scale, fma && !minuend, &handled));
    if (handled) {
      if (epilogue_done) *epilogue_done = (minuend != nullptr);
      return VISFD_HIP_OK;
    }
  }
  // the DoG/LoG epilogue is also folded into the X pass of the three-pass route (the intermediate volumes are
  // workspace slots, so dst may alias the minuend here as well)
  if (minuend && !epilogue_done) return fail(VISFD_HIP_EINVAL, "internal: epilogue request without a result flag");
  if (minuend) *epilogue_done = true;

  float *A = nullptr, *B = nullptr;
  VH_TRY(ws(ctx, WS_A, (size_t)n, &A));
  VH_TRY(ws(ctx, WS_B, (size_t)n, &B));
  if (nx >= (1LL << 31) || ny >= (1LL << 31) || nz >= (1LL << 31)) return fail(VISFD_HIP_EINVAL, "image dimension too large");
  if (!mask) {
    VH_TRY(launch_march<false>(ctx, 2, src, A, nullptr, nullptr, Tz, nx, ny, nz));
    VH_TRY(launch_march<false>(ctx, 1, A, B, nullptr, nullptr, Ty, nx, ny, nz));
    if (normalize) VH_TRY(launch_row<NORM_BOX>(ctx, B, dst, nullptr, Dx, Dy, Dz, slab.z_lo, Tx, nx, ny, nz, minuend, log_scale));
    else VH_TRY(launch_row<NORM_NONE>(ctx, B, dst, nullptr, nullptr, nullptr, nullptr, 0, Tx, nx, ny, nz, minuend, log_scale));
  } else if (!normalize) {
    // masked: the Z pass applies the mask (one bandwidth-bound kernel); Y and X then run as one sweep where the
    // single-sweep kernel applies (36 instead of 52 bytes per voxel for the normalised filter)
    VH_TRY(launch_march<true>(ctx, 2, src, A, mask, nullptr, Tz, nx, ny, nz));
    bool yx = false;
    VH_TRY(dev_gauss_fused_yx(ctx, A, dst, nx, ny, nz, Tx, Ty, nullptr, minuend, log_scale, &yx));
    if (!yx) {
      VH_TRY(launch_march<false>(ctx, 1, A, B, nullptr, nullptr, Ty, nx, ny, nz));
      VH_TRY(launch_row<NORM_NONE>(ctx, B, dst, nullptr, nullptr, nullptr, nullptr, 0, Tx, nx, ny, nz, minuend, log_scale));
    }
  } else {
    float *DA = nullptr, *DB = nullptr;
    VH_TRY(ws(ctx, WS_DEN_A, (size_t)n, &DA));
    VH_TRY(ws(ctx, WS_DEN_B, (size_t)n, &DB));
    VH_TRY(launch_march<true>(ctx, 2, src, A, mask, DA, Tz, nx, ny, nz));
    bool yx = false;
    VH_TRY(dev_gauss_fused_yx(ctx, A, B, nx, ny, nz, Tx, Ty, nullptr, nullptr, 1.0f, &yx));       // numerator
    if (yx) {
      // denominator through Y and X, divided into the numerator (and the DoG/LoG epilogue) as it is stored
      VH_TRY(dev_gauss_fused_yx(ctx, DA, dst, nx, ny, nz, Tx, Ty, B, minuend, log_scale, &yx));
      if (!yx) return fail(VISFD_HIP_EDEVICE, "internal: single-sweep Y/X pass refused its second call");
    } else {
      VH_TRY(launch_march<false>(ctx, 1, A, B, nullptr, nullptr, Ty, nx, ny, nz));
      VH_TRY(launch_march<false>(ctx, 1, DA, DB, nullptr, nullptr, Ty, nx, ny, nz));
      VH_TRY(launch_row<NORM_NONE>(ctx, DB, DA, nullptr, nullptr, nullptr, nullptr, 0, Tx, nx, ny, nz));
      VH_TRY(launch_row<NORM_DEN>(ctx, B, dst, DA, nullptr, nullptr, nullptr, 0, Tx, nx, ny, nz, minuend, log_scale));
    }
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

int dev_sub_square(visfd_hip_ctx* ctx, const float* a, const float* b, float* out, i64 n) {
  const unsigned g = grid_for(n, BLOCK, (i64)ctx->num_cus * 16);
  sub_square_kernel<<<dim3(g), dim3(BLOCK), 0, ctx->stream>>>(a, b, out, n);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

int dev_scale_clamp_sqrt(visfd_hip_ctx* ctx, float* a, i64 n, float scale) {
  const unsigned g = grid_for(n, BLOCK, (i64)ctx->num_cus * 16);
  scale_clamp_sqrt_kernel<<<dim3(g), dim3(BLOCK), 0, ctx->stream>>>(a, n, scale);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

}  // namespace vh
