// ridge.hip -- Hessian by finite differences, per-voxel eigen-decomposition, ridge saliency and
// the post-voting score (reference lib/visfd/feature.hpp:1271-1345, lib/visfd/visfd_utils.hpp:528-669,
// bin/filter_mrc/handlers.cpp:1640-1746 and :1870-1892).
//
// Device layout: multi-channel fields are channel-planar, field[c*nvox + v].
#include "common.hpp"
#include "eigen3.hpp"

namespace vh {

namespace {

constexpr int BLOCK = 256;

struct Stencil {
  const float* S;
  i64 c;       // centre (after clamping to the interior)
  i64 sx, sy, sz;
  __device__ __forceinline__ float at(int dx, int dy, int dz) const {
    return S[c + dx * sx + dy * sy + dz * sz];
  }
};

// 19-point Hessian, operand order of visfd_utils.hpp:538-564, then *= sigma*sigma (feature.hpp:1331-1333)
__device__ __forceinline__ void hessian_at(const Stencil& f, float s2, float h6[6]) {
  const float f0 = f.at(0, 0, 0);
  const float two_f0 = 2 * f0;
  const float hxx = (f.at(1, 0, 0) + f.at(-1, 0, 0)) - two_f0;
  const float hyy = (f.at(0, 1, 0) + f.at(0, -1, 0)) - two_f0;
  const float hzz = (f.at(0, 0, 1) + f.at(0, 0, -1)) - two_f0;
  const float hxy = 0.25f * (((f.at(1, 1, 0) + f.at(-1, -1, 0)) - f.at(1, -1, 0)) - f.at(-1, 1, 0));
  const float hyz = 0.25f * (((f.at(0, 1, 1) + f.at(0, -1, -1)) - f.at(0, 1, -1)) - f.at(0, -1, 1));
  const float hzx = 0.25f * (((f.at(1, 0, 1) + f.at(-1, 0, -1)) - f.at(-1, 0, 1)) - f.at(1, 0, -1));
  h6[0] = hxx * s2; h6[1] = hyy * s2; h6[2] = hzz * s2;
  h6[3] = hxy * s2; h6[4] = hyz * s2; h6[5] = hzx * s2;
}

__device__ __forceinline__ bool voxel_of_block(int nx, int ny, int& ix, int& iy, int& iz) {
  const int xblocks = (nx + BLOCK - 1) / BLOCK;
  unsigned b = blockIdx.x;
  const int bx = b % xblocks;
  b /= xblocks;
  iy = b % ny;
  iz = b / ny;
  ix = bx * BLOCK + threadIdx.x;
  return ix < nx;
}

__device__ __forceinline__ Stencil clamped_stencil(const float* S, int ix, int iy, int iz, int nx, int ny,
                                                   int nz) {
  // faces use the stencil of the neighbouring interior voxel (visfd_utils.hpp:597-610)
  const int x = min(max(ix, 1), nx - 2), y = min(max(iy, 1), ny - 2), z = min(max(iz, 1), nz - 2);
  Stencil f;
  f.S = S;
  f.sx = 1; f.sy = nx; f.sz = (i64)nx * ny;
  f.c = (i64)z * f.sz + (i64)y * nx + x;
  return f;
}

// saliency + principal direction from a flat Hessian (handlers.cpp:1653-1740)
template <bool F32>
__device__ __forceinline__ void saliency_dir(const float h6[6], int order, float& sal, float dir[3]) {
  float d6[6];
  eig::diagonalize_flat<F32>(h6, order, d6);
  const double l1 = d6[0], l2 = d6[1];
  double N = l1 * l1 - l2 * l2;
  N *= N;
  sal = (float)N;
  eig::shoemake_row0(d6 + 3, dir);
}

__global__ void __launch_bounds__(BLOCK)
hessian_kernel(const float* __restrict__ S, const float* __restrict__ mask, int nx, int ny, int nz,
               float sigma, float* __restrict__ grad, float* __restrict__ hess) {
  int ix, iy, iz;
  if (!voxel_of_block(nx, ny, ix, iy, iz)) return;
  const i64 nvox = (i64)nx * ny * nz;
  const i64 v = ((i64)iz * ny + iy) * nx + ix;
  if (mask && mask[v] == 0.0f) return;
  const Stencil f = clamped_stencil(S, ix, iy, iz, nx, ny, nz);
  if (grad) {
    const float g0 = 0.5f * (f.at(1, 0, 0) - f.at(-1, 0, 0));
    const float g1 = 0.5f * (f.at(0, 1, 0) - f.at(0, -1, 0));
    const float g2 = 0.5f * (f.at(0, 0, 1) - f.at(0, 0, -1));
    grad[v] = g0 * sigma;
    grad[nvox + v] = g1 * sigma;
    grad[2 * nvox + v] = g2 * sigma;
  }
  if (hess) {
    float h6[6];
    hessian_at(f, sigma * sigma, h6);
#pragma unroll
    for (int c = 0; c < 6; c++) hess[c * nvox + v] = h6[c];
  }
}

template <bool F32>
__global__ void __launch_bounds__(BLOCK)
hessian_saliency_kernel(const float* __restrict__ hess, const float* __restrict__ mask, i64 nvox,
                        int order, float* __restrict__ sal, float* __restrict__ dir) {
  const i64 v = (i64)blockIdx.x * BLOCK + threadIdx.x;
  if (v >= nvox) return;
  if (mask && mask[v] == 0.0f) { sal[v] = 0.0f; return; }
  float h6[6];
#pragma unroll
  for (int c = 0; c < 6; c++) h6[c] = hess[c * nvox + v];
  float s, d[3];
  saliency_dir<F32>(h6, order, s, d);
  sal[v] = s;
  dir[v] = d[0];
  dir[nvox + v] = d[1];
  dir[2 * nvox + v] = d[2];
}

template <bool F32>
__global__ void __launch_bounds__(BLOCK)
ridge_fused_kernel(const float* __restrict__ S, const float* __restrict__ mask, int nx, int ny, int nz,
                   float sigma, int order, float* __restrict__ sal, float* __restrict__ dir) {
  int ix, iy, iz;
  if (!voxel_of_block(nx, ny, ix, iy, iz)) return;
  const i64 nvox = (i64)nx * ny * nz;
  const i64 v = ((i64)iz * ny + iy) * nx + ix;
  if (mask && mask[v] == 0.0f) { sal[v] = 0.0f; return; }
  const Stencil f = clamped_stencil(S, ix, iy, iz, nx, ny, nz);
  float h6[6];
  hessian_at(f, sigma * sigma, h6);
  float s, d[3];
  saliency_dir<F32>(h6, order, s, d);
  sal[v] = s;
  dir[v] = d[0];
  dir[nvox + v] = d[1];
  dir[2 * nvox + v] = d[2];
}

// The two halves of ridge_fused_kernel for pipelines that threshold the saliency before they need directions
// (handlers.cpp:1751-1797 zeroes 95 % of the voxels; tensor voting reads the direction of the others only):
// eigenvalues and score for every voxel ...
template <bool F32>
__global__ void __launch_bounds__(BLOCK)
ridge_score_kernel(const float* __restrict__ S, const float* __restrict__ mask, int nx, int ny, int nz,
                   float sigma, int order, float* __restrict__ sal,
                   // optional peak-height factor (handlers.cpp:1698-1702): score *= image - background
                   const float* __restrict__ peak_img, const float* __restrict__ peak_bg) {
  int ix, iy, iz;
  if (!voxel_of_block(nx, ny, ix, iy, iz)) return;
  const i64 v = ((i64)iz * ny + iy) * nx + ix;
  if (mask && mask[v] == 0.0f) { sal[v] = 0.0f; return; }
  const Stencil f = clamped_stencil(S, ix, iy, iz, nx, ny, nz);
  float h6[6];
  hessian_at(f, sigma * sigma, h6);
  double lam[3];
  eig::D3 E[3];
  eig::eig_sym3<F32>(h6, order, lam, E, false);      // the eigenvalues do not depend on the eigenvector branch
  const double l1 = (float)lam[0], l2 = (float)lam[1];   // stored as float by DiagonalizeFlatSym3, re-read as double
  double N = l1 * l1 - l2 * l2;
  N *= N;
  float score = (float)N;
  if (peak_img) score *= peak_img[v] - peak_bg[v];
  __builtin_nontemporal_store(score, &sal[v]);
}

// ... and the principal direction of the voxels whose saliency is non-zero.  A workgroup scans DIR_CHUNK
// consecutive voxels, compacts the survivors into an LDS list, and spends the eigenvector work (fp64 null
// vectors, quaternion, Shoemake round trip) on full lanes only: with 5 % survivors one wave-round instead of 16.
constexpr int DIR_PER = 16;               // voxels scanned per thread: ~5 % survive, so ~200 per workgroup -- the
constexpr int DIR_CHUNK = DIR_PER * BLOCK;  // eigenvector loop then keeps 3-4 of the 4 waves busy (4 per thread: 8.3 ms)
template <bool F32>
__global__ void __launch_bounds__(BLOCK)
ridge_directions_kernel(const float* __restrict__ S, const float* __restrict__ sal, int nx, int ny, int nz,
                        float sigma, int order, float* __restrict__ dir) {
  __shared__ unsigned short list[DIR_CHUNK];
  __shared__ int wave_tot[BLOCK / 64];
  const i64 nvox = (i64)nx * ny * nz;
  const i64 base = (i64)blockIdx.x * DIR_CHUNK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned keep = 0u;   // bit k: voxel base + k * BLOCK + tid survives
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < DIR_PER; k++) {
    const i64 v = base + k * BLOCK + tid;
    const bool kp = v < nvox && sal[v] != 0.0f;
    keep |= (kp ? 1u : 0u) << k;
    cnt += kp ? 1 : 0;
  }
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int off = incl - cnt, total = 0;
#pragma unroll
  for (int w = 0; w < BLOCK / 64; w++) {
    const int t = wave_tot[w];
    off += (w < wave) ? t : 0;
    total += t;
  }
#pragma unroll
  for (int k = 0; k < DIR_PER; k++)
    if ((keep >> k) & 1u) list[off++] = (unsigned short)(k * BLOCK + tid);
  __syncthreads();
  const i64 plane = (i64)nx * ny;
  for (int i = tid; i < total; i += BLOCK) {
    const i64 v = base + list[i];
    const int iz = (int)(v / plane);
    const i64 r = v - (i64)iz * plane;
    const int iy = (int)(r / nx), ix = (int)(r - (i64)iy * nx);
    const Stencil f = clamped_stencil(S, ix, iy, iz, nx, ny, nz);
    float h6[6];
    hessian_at(f, sigma * sigma, h6);
    float s, d[3];
    saliency_dir<F32>(h6, order, s, d);
    dir[v] = d[0];
    dir[nvox + v] = d[1];
    dir[2 * nvox + v] = d[2];
  }
}

template <bool F32>
__global__ void __launch_bounds__(BLOCK)
diagonalize_kernel(const float* __restrict__ m, float* __restrict__ out, i64 n, int order) {
  const i64 v = (i64)blockIdx.x * BLOCK + threadIdx.x;
  if (v >= n) return;
  float h6[6], d6[6];
#pragma unroll
  for (int c = 0; c < 6; c++) h6[c] = m[c * n + v];
  eig::diagonalize_flat<F32>(h6, order, d6);
#pragma unroll
  for (int c = 0; c < 6; c++) out[c * n + v] = d6[c];
}

// handlers.cpp:1873-1888: score = lambda0 - lambda1 of the diagonalised tensor
template <bool F32>
__global__ void __launch_bounds__(BLOCK)
tensor_saliency_kernel(const float* __restrict__ ten, const float* __restrict__ mask, i64 nvox, int order,
                       float* __restrict__ sal,
                       // optional peak-height factor (handlers.cpp:1883-1887)
                       const float* __restrict__ peak_img, const float* __restrict__ peak_bg) {
  const i64 v = (i64)blockIdx.x * BLOCK + threadIdx.x;
  if (v >= nvox) return;
  if (mask && mask[v] == 0.0f) return;
  float t6[6];
#pragma unroll
  for (int c = 0; c < 6; c++) t6[c] = ten[c * nvox + v];
  double lam[3];
  eig::D3 E[3];
  eig::eig_sym3<F32>(t6, order, lam, E, false);
  const double l1 = (float)lam[0], l2 = (float)lam[1];  // stored as float, re-read as double
  float score = (float)(l1 - l2);
  if (peak_img) score *= peak_img[v] - peak_bg[v];
  __builtin_nontemporal_store(score, &sal[v]);
}

__global__ void __launch_bounds__(BLOCK)
aos_to_planar_kernel(const float* __restrict__ aos, float* __restrict__ planar, i64 n, int ch) {
  const i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x;  // index into the interleaved array
  if (i >= n * ch) return;
  const i64 v = i / ch;
  const int c = (int)(i - v * ch);
  planar[c * n + v] = aos[i];
}

__global__ void __launch_bounds__(BLOCK)
planar_to_aos_kernel(const float* __restrict__ planar, float* __restrict__ aos, i64 n, int ch,
                     const float* __restrict__ mask) {
  const i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n * ch) return;
  const i64 v = i / ch;
  const int c = (int)(i - v * ch);
  if (mask && mask[v] == 0.0f) return;
  aos[i] = planar[c * n + v];
}

// the eigen kernels exist in two forms (eigen3.hpp: eig_sym3<F32TRIG>); the context option eig_f32 chooses
#define VH_EIG_LAUNCH(kernel, grid, ...)                                                          \
  do {                                                                                            \
    if (ctx->opt.eig_f32) kernel<true><<<grid, dim3(BLOCK), 0, ctx->stream>>>(__VA_ARGS__);       \
    else kernel<false><<<grid, dim3(BLOCK), 0, ctx->stream>>>(__VA_ARGS__);                       \
  } while (0)

int voxel_grid(i64 nx, i64 ny, i64 nz, unsigned* g) {
  if (nx >= (1LL << 31) || ny >= (1LL << 31) || nz >= (1LL << 31))
    return fail(VISFD_HIP_EINVAL, "dimension too large");
  const i64 nb = ((nx + BLOCK - 1) / BLOCK) * ny * nz;
  if (nb > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  *g = (unsigned)nb;
  return VISFD_HIP_OK;
}

int linear_grid(i64 n, unsigned* g) {
  const i64 nb = (n + BLOCK - 1) / BLOCK;
  if (nb > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "array too large for one launch");
  *g = (unsigned)(nb < 1 ? 1 : nb);
  return VISFD_HIP_OK;
}

}  // namespace

int dev_hessian(visfd_hip_ctx* ctx, const float* S, const float* mask, i64 nx, i64 ny, i64 nz,
                float sigma, float* grad, float* hess) {
  if (nx < 3 || ny < 3 || nz < 3)
    return fail(VISFD_HIP_EINVAL, "CalcHessian requires an image at least 3 voxels wide in x,y,z");
  unsigned g;
  VH_TRY(voxel_grid(nx, ny, nz, &g));
  hessian_kernel<<<dim3(g), dim3(BLOCK), 0, ctx->stream>>>(S, mask, (int)nx, (int)ny, (int)nz, sigma,
                                                          grad, hess);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

int dev_hessian_saliency(visfd_hip_ctx* ctx, const float* hess, const float* mask, i64 nvox, int order,
                         float* sal, float* dir) {
  unsigned g;
  VH_TRY(linear_grid(nvox, &g));
  VH_EIG_LAUNCH(hessian_saliency_kernel, dim3(g), hess, mask, nvox, order, sal, dir);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

int dev_ridge_saliency_fused(visfd_hip_ctx* ctx, const float* S, const float* mask, i64 nx, i64 ny,
                             i64 nz, float sigma, int order, float* sal, float* dir) {
  if (nx < 3 || ny < 3 || nz < 3)
    return fail(VISFD_HIP_EINVAL, "ridge detection requires an image at least 3 voxels wide in x,y,z");
  unsigned g;
  VH_TRY(voxel_grid(nx, ny, nz, &g));
  VH_EIG_LAUNCH(ridge_fused_kernel, dim3(g), S, mask, (int)nx, (int)ny, (int)nz, sigma, order, sal, dir);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

int dev_ridge_score(visfd_hip_ctx* ctx, const float* S, const float* mask, i64 nx, i64 ny, i64 nz, float sigma,
                    int order, float* sal, const float* peak_img, const float* peak_bg) {
  if (nx < 3 || ny < 3 || nz < 3)
    return fail(VISFD_HIP_EINVAL, "ridge detection requires an image at least 3 voxels wide in x,y,z");
  unsigned g;
  VH_TRY(voxel_grid(nx, ny, nz, &g));
  VH_EIG_LAUNCH(ridge_score_kernel, dim3(g), S, mask, (int)nx, (int)ny, (int)nz, sigma, order, sal, peak_img, peak_bg);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

int dev_ridge_directions(visfd_hip_ctx* ctx, const float* S, const float* sal, i64 nx, i64 ny, i64 nz, float sigma,
                         int order, float* dir) {
  if (nx < 3 || ny < 3 || nz < 3)
    return fail(VISFD_HIP_EINVAL, "ridge detection requires an image at least 3 voxels wide in x,y,z");
  if (nx >= (1LL << 31) || ny >= (1LL << 31) || nz >= (1LL << 31)) return fail(VISFD_HIP_EINVAL, "dimension too large");
  const i64 nb = (nx * ny * nz + DIR_CHUNK - 1) / DIR_CHUNK;
  if (nb > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  VH_EIG_LAUNCH(ridge_directions_kernel, dim3((unsigned)nb), S, sal, (int)nx, (int)ny, (int)nz, sigma, order, dir);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

int dev_diagonalize(visfd_hip_ctx* ctx, const float* m, float* out, i64 n, int order) {
  unsigned g;
  VH_TRY(linear_grid(n, &g));
  VH_EIG_LAUNCH(diagonalize_kernel, dim3(g), m, out, n, order);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

int dev_tensor_saliency(visfd_hip_ctx* ctx, const float* ten, const float* mask, i64 nvox, int order,
                        float* sal, const float* peak_img, const float* peak_bg) {
  unsigned g;
  VH_TRY(linear_grid(nvox, &g));
  VH_EIG_LAUNCH(tensor_saliency_kernel, dim3(g), ten, mask, nvox, order, sal, peak_img, peak_bg);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

int dev_interleaved_to_planar(visfd_hip_ctx* ctx, const float* aos, float* planar, i64 n, int ch) {
  unsigned g;
  VH_TRY(linear_grid(n * ch, &g));
  aos_to_planar_kernel<<<dim3(g), dim3(BLOCK), 0, ctx->stream>>>(aos, planar, n, ch);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

int dev_planar_to_interleaved(visfd_hip_ctx* ctx, const float* planar, float* aos, i64 n, int ch,
                              const float* mask) {
  unsigned g;
  VH_TRY(linear_grid(n * ch, &g));
  planar_to_aos_kernel<<<dim3(g), dim3(BLOCK), 0, ctx->stream>>>(planar, aos, n, ch, mask);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

}  // namespace vh
