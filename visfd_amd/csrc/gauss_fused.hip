// gauss_fused.hip -- single-sweep separable 3-D Gaussian for gfx950 (unmasked case).
//
// One kernel reads the source once and writes the result once (algorithmic 8 B/voxel), while
// keeping the reference's value-level order (lib/visfd/filter3d.hpp:741-981): convolve along Z,
// round to float, convolve along Y, round, convolve along X, round, then divide by the boundary
// normaliser (filter3d.hpp:997-1022).  Every sum is "acc = 0; acc += t[j]*f[i-j]" for j ascending
// with separate multiply and add (-ffp-contract=off), so the output is bit-identical to the
// three-pass form.
//
// Work decomposition (per workgroup of NT threads):
//   * an output tile of TX x TY voxels in the XY plane, marched along Z over a chunk of planes;
//   * Z pass: each thread owns NC columns of the (TX+2H) x (TY+2H) haloed tile and keeps the last
//     2H+1 source planes of each column in a REGISTER RING (the march is unrolled 2H+1 times so the
//     ring is statically indexed); one new plane is fetched per step, x-contiguous across lanes;
//   * the Z-filtered haloed plane goes to LDS; Y pass: two adjacent x per lane (ds_read_b64 down a
//     column), result rows to LDS; X pass: two adjacent outputs per lane from one ds_read_b64
//     window, normalise, float2 store (x-contiguous across lanes).
// Zero extension: samples outside the image are fed as 0.0f, which adds an exact +0.0 to the
// accumulator (the reference skips those terms, filter1d.hpp:98-99; same bits for finite data).
#include "common.hpp"

namespace vh {

namespace {

template <int H, int TX, int TY, int NT>
struct FusedCfg {
  static constexpr int W = 2 * H + 1;
  static constexpr int HX = TX + 2 * H;           // haloed tile width (even: TX even)
  static constexpr int HY = TY + 2 * H;
  static constexpr int NCOL = HX * HY;
  static constexpr int NC = (NCOL + NT - 1) / NT;  // ring columns per thread
  static constexpr int SX = HX + 2;                // LDS row stride in floats (even, 8-byte rows)
  static constexpr int YTASKS = (HX / 2) * TY;     // (x pair, y) outputs of the Y pass
  static constexpr int YROUNDS = (YTASKS + NT - 1) / NT;
  static constexpr int XTASKS = (TX / 2) * TY;
  static constexpr int XROUNDS = (XTASKS + NT - 1) / NT;
  static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)SX * (HY + TY);
};

template <int H>
struct TapsH {
  float t[2 * H + 1];
};

template <int H, int TX, int TY, int NT, bool NORMALIZE>
__global__ void __launch_bounds__(NT)
gauss_fused_kernel(const float* __restrict__ src, float* __restrict__ dst, TapsH<H> tz, TapsH<H> ty,
                   TapsH<H> tx, const float* __restrict__ Dx, const float* __restrict__ Dy,
                   const float* __restrict__ Dz, i64 dz_offset, int nx, int ny, int nz, int zchunk,
                   int tiles_x, int tiles_y, int nchunks) {
  typedef FusedCfg<H, TX, TY, NT> C;
  constexpr int W = C::W;
  __shared__ __attribute__((aligned(16))) float sZ[C::HY * C::SX];
  __shared__ __attribute__((aligned(16))) float sY[TY * C::SX];

  // XCD-aware block order: consecutive logical tiles (neighbours in x, then y) share an XCD/L2.
  const unsigned nblk = gridDim.x;
  unsigned b = blockIdx.x;
  {
    const unsigned per = nblk / 8;
    if (per * 8 == nblk) b = (b % 8) * per + b / 8;
  }
  const int tile_x = b % tiles_x;
  const int tile_y = (b / tiles_x) % tiles_y;
  const int chunk = b / (tiles_x * tiles_y);
  const int x0 = tile_x * TX, y0 = tile_y * TY;
  const int zs = chunk * zchunk;
  const int ze = min(zs + zchunk, nz);
  const int tid = threadIdx.x;
  const i64 plane = (i64)nx * ny;

  // ---- ring columns owned by this thread ---------------------------------------------------
  float ring[C::NC][W];
  i64 col_off[C::NC];   // offset of the column inside a plane, or -1 when outside the image
  int lds_off[C::NC];   // where its Z-filtered value goes in sZ, or -1 for padding slots
#pragma unroll
  for (int c = 0; c < C::NC; c++) {
    const int id = tid + c * NT;
    const int cy = id / C::HX, cx = id - cy * C::HX;
    const int gx = x0 - H + cx, gy = y0 - H + cy;
    const bool slot = id < C::NCOL;
    const bool inside = slot && gx >= 0 && gx < nx && gy >= 0 && gy < ny;
    col_off[c] = inside ? ((i64)gy * nx + gx) : -1;
    lds_off[c] = slot ? (cy * C::SX + cx) : -1;
  }
  // preload planes zs-H .. zs+H-1 into ring slots 1..W-1 (slot k holds plane zs-H+k-1 ... see below)
#pragma unroll
  for (int k = 0; k < W - 1; k++) {
    const int z = zs - H + k;
    const bool zin = (z >= 0) && (z < nz);
#pragma unroll
    for (int c = 0; c < C::NC; c++)
      ring[c][k + 1] = (zin && col_off[c] >= 0) ? src[(i64)z * plane + col_off[c]] : 0.0f;
  }

  // ---- march along z, unrolled W times so that ring indices are compile-time ----------------
  // At unrolled step u (output plane z): plane z+H is written to slot u, and slot (u+1+m) % W
  // holds plane z-H+m for m = 0..W-2.
  for (int zbase = zs; zbase < ze; zbase += W) {
#pragma unroll
    for (int u = 0; u < W; u++) {
      const int z = zbase + u;
      if (z < ze) {  // uniform across the workgroup
        // newest plane
        {
          const int zn = z + H;
          const bool zin = zn < nz;
#pragma unroll
          for (int c = 0; c < C::NC; c++)
            ring[c][u] = (zin && col_off[c] >= 0) ? src[(i64)zn * plane + col_off[c]] : 0.0f;
        }
        // Z pass: j ascending <=> plane z-j descending: newest (slot u) first
#pragma unroll
        for (int c = 0; c < C::NC; c++) {
          float acc = 0.0f;
#pragma unroll
          for (int jj = 0; jj < W; jj++) {
            // j = jj - H ; plane z - j = z + H - jj ; m = W-1-jj ; slot = (u+1+m) % W, for jj=0: slot u
            constexpr int dummy = 0;
            (void)dummy;
            const int slot = (jj == 0) ? u : ((u + 1 + (W - 1 - jj)) % W);
            const float term = tz.t[jj] * ring[c][slot];
            acc = acc + term;
          }
          if (lds_off[c] >= 0) sZ[lds_off[c]] = acc;
        }
        __syncthreads();
        // Y pass: outputs (xp, y): two adjacent x (haloed coords 2xp, 2xp+1), y in [0,TY)
#pragma unroll
        for (int r = 0; r < C::YROUNDS; r++) {
          const int task = tid + r * NT;
          if (task < C::YTASKS) {
            const int y = task / (C::HX / 2);
            const int xp = task - y * (C::HX / 2);
            const float* base = &sZ[y * C::SX + 2 * xp];  // haloed row y+H-j, j=-H: row y+2H
            float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int jj = 0; jj < W; jj++) {
              const float2 v = *reinterpret_cast<const float2*>(base + (2 * H - jj) * C::SX);
              const float t = ty.t[jj];
              const float p0 = t * v.x;
              const float p1 = t * v.y;
              a0 = a0 + p0;
              a1 = a1 + p1;
            }
            *reinterpret_cast<float2*>(&sY[y * C::SX + 2 * xp]) = make_float2(a0, a1);
          }
        }
        __syncthreads();
        // X pass: outputs (xp, y): x = 2xp, 2xp+1 in tile coords
#pragma unroll
        for (int r = 0; r < C::XROUNDS; r++) {
          const int task = tid + r * NT;
          if (task < C::XTASKS) {
            const int y = task / (TX / 2);
            const int xp = task - y * (TX / 2);
            const float* base = &sY[y * C::SX + 2 * xp];
            float v[2 * H + 2];
#pragma unroll
            for (int k = 0; k < H + 1; k++) {
              const float2 q = *reinterpret_cast<const float2*>(base + 2 * k);
              v[2 * k] = q.x;
              v[2 * k + 1] = q.y;
            }
            float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int jj = 0; jj < W; jj++) {
              // j = jj-H: haloed index x + H - j = x + 2H - jj
              const float t = tx.t[jj];
              const float p0 = t * v[2 * H - jj];
              const float p1 = t * v[2 * H - jj + 1];
              a0 = a0 + p0;
              a1 = a1 + p1;
            }
            const int gx = x0 + 2 * xp, gy = y0 + y;
            if (gy < ny && gx < nx) {
              if (NORMALIZE) {
                const float dyz = Dy[gy];
                const float dz = Dz[z + dz_offset];
                const float d0 = (Dx[gx] * dyz) * dz;
                a0 = a0 / d0;
                if (gx + 1 < nx) {
                  const float d1 = (Dx[gx + 1] * dyz) * dz;
                  a1 = a1 / d1;
                }
              }
              float* o = dst + (i64)z * plane + (i64)gy * nx + gx;
              if (gx + 1 < nx && ((nx & 1) == 0)) {
                *reinterpret_cast<float2*>(o) = make_float2(a0, a1);
              } else {
                o[0] = a0;
                if (gx + 1 < nx) o[1] = a1;
              }
            }
          }
        }
        // no barrier needed here: the next step's Z pass writes sZ only after every thread has
        // passed the barrier that follows the Y pass (all sZ reads done), and the next Y pass
        // writes sY only after the barrier that follows the next Z pass (all sY reads done).
      }
    }
  }
}

template <int H, int TX, int TY, int NT>
int launch_fused(visfd_hip_ctx* ctx, const float* src, float* dst, i64 nx, i64 ny, i64 nz,
                 const Taps& tx, const Taps& ty, const Taps& tz, const float* Dx, const float* Dy,
                 const float* Dz, i64 dz_offset, bool normalize) {
  TapsH<H> a, b, c;
  for (int k = 0; k < 2 * H + 1; k++) { a.t[k] = tz.t[k]; b.t[k] = ty.t[k]; c.t[k] = tx.t[k]; }
  const int tiles_x = (int)((nx + TX - 1) / TX), tiles_y = (int)((ny + TY - 1) / TY);
  // z chunks: enough workgroups to fill the chip several times over, but long enough marches
  // to amortise the 2H-plane ring warm-up
  const i64 tiles = (i64)tiles_x * tiles_y;
  i64 want_chunks = ((i64)ctx->num_cus * 8 + tiles - 1) / tiles;
  if (want_chunks < 1) want_chunks = 1;
  i64 zchunk = (nz + want_chunks - 1) / want_chunks;
  const i64 min_chunk = 16 * (2 * H + 1) > 64 ? 64 : 16 * (2 * H + 1);
  if (zchunk < min_chunk) zchunk = min_chunk;
  if (zchunk > nz) zchunk = nz;
  i64 nchunks = (nz + zchunk - 1) / zchunk;
  i64 nblk = tiles * nchunks;
  if (nblk > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  dim3 grid((unsigned)nblk), block(NT);
  if (normalize)
    gauss_fused_kernel<H, TX, TY, NT, true><<<grid, block, 0, ctx->stream>>>(
        src, dst, a, b, c, Dx, Dy, Dz, dz_offset, (int)nx, (int)ny, (int)nz, (int)zchunk, tiles_x,
        tiles_y, (int)nchunks);
  else
    gauss_fused_kernel<H, TX, TY, NT, false><<<grid, block, 0, ctx->stream>>>(
        src, dst, a, b, c, Dx, Dy, Dz, dz_offset, (int)nx, (int)ny, (int)nz, (int)zchunk, tiles_x,
        tiles_y, (int)nchunks);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

}  // namespace

int dev_gauss_fused(visfd_hip_ctx* ctx, const float* src, float* dst, i64 nx, i64 ny, i64 nz,
                    const Taps& tx, const Taps& ty, const Taps& tz, const float* Dx, const float* Dy,
                    const float* Dz, i64 dz_offset, bool normalize, bool* handled) {
  *handled = false;
  const int H = tx.h;
  if (ty.h != H || tz.h != H) return VISFD_HIP_OK;          // anisotropic window: 3-pass path
  if (H < 1 || H > 10) return VISFD_HIP_OK;
  if (nx >= (1LL << 31) || ny >= (1LL << 31) || nz >= (1LL << 31)) return VISFD_HIP_OK;
  if (src == dst) return VISFD_HIP_OK;                        // in place: 3-pass path via scratch
  *handled = true;
#define VH_FUSED_CASE(HH, TXX, TYY, NTT) \
  case HH: return launch_fused<HH, TXX, TYY, NTT>(ctx, src, dst, nx, ny, nz, tx, ty, tz, Dx, Dy, Dz, dz_offset, normalize);
  switch (H) {
    VH_FUSED_CASE(1, 64, 16, 256)
    VH_FUSED_CASE(2, 64, 16, 256)
    VH_FUSED_CASE(3, 64, 16, 256)
    VH_FUSED_CASE(4, 64, 16, 256)
    VH_FUSED_CASE(5, 64, 16, 256)
    VH_FUSED_CASE(6, 64, 16, 256)
    VH_FUSED_CASE(7, 64, 16, 512)
    VH_FUSED_CASE(8, 64, 16, 512)
    VH_FUSED_CASE(9, 64, 16, 512)
    VH_FUSED_CASE(10, 64, 16, 512)
  }
#undef VH_FUSED_CASE
  *handled = false;
  return VISFD_HIP_OK;
}

}  // namespace vh
