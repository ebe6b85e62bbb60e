// tv.hip -- dense stick tensor voting (reference lib/visfd/feature.hpp:1914-2037, :2217-2384).
//
// Receiver-centric gather, one thread per receiver voxel, so that each receiver accumulates its
// votes in exactly the reference's order (jz, jy, jx ascending over the window, sender = r - j):
//   u   = (rhat_x*n_x + rhat_y*n_y) + rhat_z*n_z          rhat = unit vector sender -> receiver
//   ang = 1 - u*u  (surfaces)   |   u*u (curves)
//   dec = ang (exponent 2) | ang*ang (exponent 4) | pow(ang, exponent/2) otherwise
//   m   = 2u*rhat - n  (surfaces)  |  n - 2u*rhat (curves)
//   T_ab += (((sal * (w*mask_src)) * dec) * m_a) * m_b      for a <= b
// all in float with separate multiplies and adds (-ffp-contract=off).
//
// Device layout: direction is 3 planes, tensor is 6 planes (xx,yy,zz,xy,yz,xz), each nvox floats.
#include <vector>

#include "common.hpp"

namespace vh {

namespace {

constexpr int BLOCK = 256;

struct TvParams {
  int nx, ny, nz;        // local array extent
  int z_out0, z_out1;    // receiver planes [z_out0, z_out1)
  int h;
  int exponent;
  int curves;
  int weights_only;      // 1: no tensors -- the sum of the vote weights of every receiver into ONE plane (feature.hpp:2376-2377)
};

__device__ __forceinline__ float decay_of(float ang, int exponent) {
  if (exponent == 2) return ang;
  if (exponent == 4) return ang * ang;
  return (float)pow((double)ang, 0.5 * (double)exponent);
}

// One vote, accumulated into T[6] (xx,yy,zz,xy,yz,xz).
__device__ __forceinline__ void add_vote(float T[6], float sal, float fv, float r0, float r1, float r2,
                                         float n0, float n1, float n2, int exponent, int curves) {
  const float u = (r0 * n0 + r1 * n1) + r2 * n2;
  const float ux2 = u * 2.0f;
  const float u2 = u * u;
  const float c2 = 1.0f - u2;
  const float ang = curves ? u2 : c2;
  const float dec = decay_of(ang, exponent);
  float m0, m1, m2;
  if (curves) {
    m0 = n0 - ux2 * r0; m1 = n1 - ux2 * r1; m2 = n2 - ux2 * r2;
  } else {
    m0 = ux2 * r0 - n0; m1 = ux2 * r1 - n1; m2 = ux2 * r2 - n2;
  }
  const float base = (sal * fv) * dec;
  const float b0 = base * m0, b1 = base * m1, b2 = base * m2;
  T[0] = T[0] + b0 * m0;
  T[3] = T[3] + b0 * m1;
  T[5] = T[5] + b0 * m2;
  T[1] = T[1] + b1 * m1;
  T[4] = T[4] + b1 * m2;
  T[2] = T[2] + b2 * m2;
}

// ---------------------------------------------------------------------------------------------
// Baseline kernel: every thread walks the whole window.  Lanes of a wave are consecutive x, so
// the tap (jz,jy,jx) -- and with it w and rhat -- is wave-uniform; zero-weight taps are skipped
// for the whole wave.  Used for windows whose lookup table does not fit the tiled kernel and as
// an in-library cross-check (tests compare both against the CPU oracle).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK)
tv_dense_kernel(const float* __restrict__ sal, const float* __restrict__ dir,
                float* __restrict__ ten, const float* __restrict__ mask_src,
                const float* __restrict__ mask_dst, const float4* __restrict__ table /* w,rx,ry,rz */,
                TvParams p) {
  const int xblocks = (p.nx + BLOCK - 1) / BLOCK;
  unsigned b = blockIdx.x;
  const int bx = b % xblocks;
  b /= xblocks;
  const int iy = b % p.ny;
  const int iz = p.z_out0 + (int)(b / p.ny);
  const int ix = bx * BLOCK + (int)threadIdx.x;
  if (ix >= p.nx) return;
  const i64 plane = (i64)p.nx * p.ny;
  const i64 nvox = plane * p.nz;
  const i64 c = (i64)iz * plane + (i64)iy * p.nx + ix;
  if (mask_dst && mask_dst[c] == 0.0f) return;
  float T[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  const int h = p.h, n = 2 * h + 1;
  for (int jz = -h; jz <= h; jz++) {
    const int sz = iz - jz;
    if (sz < 0 || sz >= p.nz) continue;
    for (int jy = -h; jy <= h; jy++) {
      const int sy = iy - jy;
      if (sy < 0 || sy >= p.ny) continue;
      const i64 row = (i64)sz * plane + (i64)sy * p.nx;
      const float4* trow = table + ((i64)(jz + h) * n + (jy + h)) * n + h;
      for (int jx = -h; jx <= h; jx++) {
        const float4 t = trow[jx];
        if (t.x == 0.0f) continue;  // wave-uniform
        const int sx = ix - jx;
        if (sx < 0 || sx >= p.nx) continue;
        const i64 s = row + sx;
        float fv = t.x;
        if (mask_src) {
          const float mv = mask_src[s];
          if (mv == 0.0f) continue;
          fv = fv * mv;
        }
        const float sv = sal[s];
        if (sv == 0.0f) continue;
        if (fv == 0.0f) continue;
        if (p.weights_only) { T[0] = T[0] + fv; continue; }     // "denominator += filter_val"
        add_vote(T, sv, fv, t.y, t.z, t.w, dir[s], dir[nvox + s], dir[2 * nvox + s], p.exponent,
                 p.curves);
      }
    }
  }
  if (p.weights_only) { ten[c] = T[0]; return; }
#pragma unroll
  for (int k = 0; k < 6; k++) ten[k * nvox + c] = T[k];
}

}  // namespace

// declared in tv_tiled.hip
int dev_tv_tiled(visfd_hip_ctx* ctx, const float* sal, const float* dir, float* ten,
                 const float* mask_src, const float* mask_dst, i64 nx, i64 ny, i64 nz, i64 z_out0,
                 i64 z_out1, int h, const float4* dtab, int exponent, bool curves, bool weights_only, bool* handled);

// declared in tv_box.hip
int dev_tv_box(visfd_hip_ctx* ctx, const float* sal, const float* dir, float* ten, const float* mask_src,
               const float* mask_dst, i64 nx, i64 ny, i64 nz, i64 z_out0, i64 z_out1, int h, const float4* dtab_box,
               int exponent, bool* handled, bool exact);


// The vote table of (sigma_tv, cutoff) on the device: float4 {w, rhat_x, rhat_y, rhat_z} per offset j in z, y, x order
// (filter3d.hpp:563-578, feature.hpp:2470-2478).  Built on the host once and kept in the context: a launch with the same
// parameters queues no copy and never waits for the stream.
static int tv_table_device(visfd_hip_ctx* ctx, float sigma_tv, float cutoff, int h, const float4** out) {
  const size_t n = 2 * (size_t)h + 1, m = n * n * n;
  if (ctx->tv_table_dev && ctx->tv_table_h == h && ctx->tv_table_key[0] == sigma_tv && ctx->tv_table_key[1] == cutoff) {
    *out = reinterpret_cast<const float4*>(ctx->tv_table_dev);
    return VISFD_HIP_OK;
  }
  // kernels of an earlier call may still read the old table, and the copy below reads host memory of this call
  VH_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<float> w(m), rh(3 * m);
  host_tv_tables(sigma_tv, h, w.data(), rh.data());
  // three tables: [0, m) the reference's {w, rhat}, packed (baseline kernel); then the same with rows padded to
  // tv_padded_row(h) entries (tiled kernel: LDS banks; pad entries are never read); then the tolerance mode's
  // {w, sqrt(2) rhat} in the slice layout of tv_box.hip (zero rows and zero row tails, which ARE read: common.hpp)
  const size_t sp = (size_t)tv_padded_row(h), m2 = n * n * sp;
  // ... and the reference's {w, rhat} once more in that slice layout (the exact form of tv_box.hip)
  const size_t spb = (size_t)tv_box_row(h), nslb = (size_t)tv_box_slice(h), m3 = (n + 1) * nslb;   // (+ one slice of zeros)
  std::vector<float4> tab(m + m2 + 2 * m3, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
  const float rt2 = 1.41421356237309504880f;
  for (size_t k = 0; k < m; k++) {
    tab[k] = make_float4(w[k], rh[3 * k], rh[3 * k + 1], rh[3 * k + 2]);
    tab[m + (k / n) * sp + (k % n)] = tab[k];
    const size_t kb = (k / (n * n)) * nslb + 4 + ((k / n) % n + 3) * spb + (k % n);
    tab[m + m2 + kb] = make_float4(w[k], rt2 * rh[3 * k], rt2 * rh[3 * k + 1], rt2 * rh[3 * k + 2]);
    tab[m + m2 + m3 + kb] = tab[k];
  }
  float4* dtab = nullptr;
  ctx->tv_table_dev = nullptr;
  VH_TRY(ws(ctx, WS_TVTAB, m + m2 + 2 * m3, &dtab));
  VH_HIP(hipMemcpyAsync(dtab, tab.data(), sizeof(float4) * (m + m2 + 2 * m3), hipMemcpyHostToDevice, ctx->stream));
  VH_HIP(hipStreamSynchronize(ctx->stream));
  ctx->tv_table_dev = reinterpret_cast<float*>(dtab);
  ctx->tv_table_h = h;
  ctx->tv_table_key[0] = sigma_tv;
  ctx->tv_table_key[1] = cutoff;
  *out = dtab;
  return VISFD_HIP_OK;
}

int dev_tv_dense_stick(visfd_hip_ctx* ctx, const float* sal, const float* dir, float* ten,
                       const float* mask_src, const float* mask_dst, i64 nx, i64 ny, i64 nz,
                       i64 z_out0, i64 z_out1, float sigma_tv, int exponent, float cutoff, bool curves) {
  VH_TRY(check_dims(nx, ny, nz));
  if (nx >= (1LL << 31) || ny >= (1LL << 31) || nz >= (1LL << 31))
    return fail(VISFD_HIP_EINVAL, "dimension too large");
  VH_REQUIRE(z_out0 >= 0 && z_out1 <= nz && z_out0 <= z_out1, "bad receiver plane range");
  if (z_out0 == z_out1) return VISFD_HIP_OK;
  const int h = host_tv_halfwidth(sigma_tv, cutoff);
  VH_REQUIRE(h >= 0 && h <= 255, "tensor-voting window halfwidth out of range");
  const float4* dtab = nullptr;
  VH_TRY(tv_table_device(ctx, sigma_tv, cutoff, h, &dtab));

  const size_t m_packed = (size_t)(2 * h + 1) * (2 * h + 1) * (2 * h + 1);
  const size_t m_padded = (size_t)(2 * h + 1) * (2 * h + 1) * (size_t)tv_padded_row(h);
  bool handled = false;
  // tolerance mode (option tv_fma): fused multiply-adds, sub-patches with two sender streams, box-tested hit lists
  // (tv_box.hip); windows and vote forms it does not take fall through to the exact kernels
  if (!ctx->opt.tv_dense && ctx->opt.tv_fma && !curves)
    VH_TRY(dev_tv_box(ctx, sal, dir, ten, mask_src, mask_dst, nx, ny, nz, z_out0, z_out1, h, dtab + m_packed + m_padded, exponent,
                      &handled, false));
  if (handled) return VISFD_HIP_OK;
  // exact arithmetic in the same kernel structure (surfaces, exponent 2 or 4, source mask absent or binary, finite saliencies; option
  // tv_exact_tiled = 1 keeps the round-2 kernel)
  if (!ctx->opt.tv_dense && !ctx->opt.tv_exact_tiled && !curves)
    VH_TRY(dev_tv_box(ctx, sal, dir, ten, mask_src, mask_dst, nx, ny, nz, z_out0, z_out1, h,
                      dtab + m_packed + m_padded + (size_t)(2 * h + 2) * (size_t)tv_box_slice(h), exponent, &handled, true));
  if (handled) return VISFD_HIP_OK;
  if (!ctx->opt.tv_dense)
    VH_TRY(dev_tv_tiled(ctx, sal, dir, ten, mask_src, mask_dst, nx, ny, nz, z_out0, z_out1, h, dtab + m_packed, exponent, curves,
                        false, &handled));
  if (handled) return VISFD_HIP_OK;

  hipStream_t st = ctx->stream;
  TvParams p;
  p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
  p.z_out0 = (int)z_out0; p.z_out1 = (int)z_out1;
  p.h = h; p.exponent = exponent; p.curves = curves ? 1 : 0; p.weights_only = 0;
  const i64 nb = ((nx + BLOCK - 1) / BLOCK) * ny * (z_out1 - z_out0);
  if (nb > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  tv_dense_kernel<<<dim3((unsigned)nb), dim3(BLOCK), 0, st>>>(sal, dir, ten, mask_src, mask_dst, dtab, p);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

// The sum of the weights of the votes every receiver takes (one plane-sized volume): the "denominator" TVDenseStick
// accumulates for its normalisation (feature.hpp:1761-1822, 2376-2382) -- w(j) * mask_src(sender) over the in-bounds
// senders with non-zero saliency, source mask and weight, added in vote order.
int dev_tv_weight_sum(visfd_hip_ctx* ctx, const float* sal, float* den, const float* mask_src, const float* mask_dst, i64 nx,
                      i64 ny, i64 nz, float sigma_tv, float cutoff) {
  VH_TRY(check_dims(nx, ny, nz));
  if (nx >= (1LL << 31) || ny >= (1LL << 31) || nz >= (1LL << 31)) return fail(VISFD_HIP_EINVAL, "dimension too large");
  const int h = host_tv_halfwidth(sigma_tv, cutoff);
  VH_REQUIRE(h >= 0 && h <= 255, "tensor-voting window halfwidth out of range");
  const float4* dtab = nullptr;
  VH_TRY(tv_table_device(ctx, sigma_tv, cutoff, h, &dtab));
  bool handled = false;
  if (!ctx->opt.tv_dense)
    VH_TRY(dev_tv_tiled(ctx, sal, nullptr, den, mask_src, mask_dst, nx, ny, nz, 0, nz, h,
                        dtab + (size_t)(2 * h + 1) * (2 * h + 1) * (2 * h + 1), 4, false, true, &handled));
  if (handled) return VISFD_HIP_OK;
  // windows the tiled kernel declines (h = 0, h > 40, slices beyond LDS) and the tv_dense option: the baseline kernel in
  // its weights-only form, same order of accumulation
  TvParams p;
  p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
  p.z_out0 = 0; p.z_out1 = (int)nz;
  p.h = h; p.exponent = 4; p.curves = 0; p.weights_only = 1;
  const i64 nb = ((nx + BLOCK - 1) / BLOCK) * ny * nz;
  if (nb > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  tv_dense_kernel<<<dim3((unsigned)nb), dim3(BLOCK), 0, ctx->stream>>>(sal, nullptr, den, mask_src, mask_dst, dtab, p);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

}  // namespace vh
