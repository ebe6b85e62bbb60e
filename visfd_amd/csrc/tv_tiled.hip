// tv_tiled.hip -- LDS-tiled dense stick tensor voting for gfx950
// (reference lib/visfd/feature.hpp:1914-2037 and :2217-2384).
//
// The reference walks, for every receiver voxel, the whole (2h+1)^3 window and skips senders whose
// saliency is zero (typically 95 % of them, feature.hpp:1704-1709).  Here the skipping is done
// once per workgroup instead of once per receiver:
//
//   * a workgroup owns a 16 x 16 x 1 tile of receivers (one per thread; each wave an 8 x 8 patch);
//   * sender planes are visited from z+h down to z-h (= jz ascending).  For each plane the
//     workgroup reads the (16+2h)^2 region of saliencies around the tile (next plane prefetched in
//     registers), and compacts the salient, unmasked senders -- position, saliency, normal, mask
//     value -- into an LDS list in scan order (ordered block-wide prefix sum, so deterministic);
//   * the list is consumed from its end in chunks of 64.  Phase A: every lane tests the 64 senders
//     against its own receiver with integer arithmetic (jx^2+jy^2+jz^2 <= h^2, an exact superset of
//     the table's spherical support) and records the hits as a 64-bit mask in registers.
//     Phase B: every lane pops ITS OWN hits in order and accumulates the votes, so lanes are
//     busy with real votes instead of idling under a sparse exec mask;
//   * weights and unit displacements come from the |jz| slice of a (h+1)^3 octant table held in
//     LDS: w(j) depends on (|jx|,|jy|,|jz|) only and rhat(-j) = -rhat(j) exactly.
//
// Order of accumulation per receiver: jz ascending (plane order), then jy, jx ascending
// (= list order reversed), exactly the reference's, and each vote is the same chain of float
// multiplies and adds (no FMA), so tensors are bit-identical to the CPU path for exponent 2 and 4.
#include <vector>

#include "common.hpp"

namespace vh {

namespace {

constexpr int NT = 256;
constexpr int TILE = 16;
constexpr int VPT = 7;               // region voxels per thread per band
constexpr int BAND_CAP = NT * VPT;   // 1792 voxels (and list entries) per band
constexpr int CHUNK = 64;

struct alignas(8) Entry {  // 24 bytes
  float sal, n0, n1, n2;
  float mv;              // source-mask value (1 when unmasked)
  unsigned pos;          // (ey << 8) | ex, region-relative
};

struct TiledParams {
  int nx, ny, nz;
  int z_out0;
  int h, hp1;            // halfwidth, h+1
  int rw, rh;            // region width/height = TILE + 2h
  int band_rows, nbands;
  int tiles_x, tiles_y;
  int exponent, curves;
};

__device__ __forceinline__ float decay_of(float ang, int exponent) {
  if (exponent == 4) return ang * ang;
  if (exponent == 2) return ang;
  return (float)pow((double)ang, 0.5 * (double)exponent);
}

template <bool MASKED_SRC>
__global__ void __launch_bounds__(NT)
tv_tiled_kernel(const float* __restrict__ sal, const float* __restrict__ dir, float* __restrict__ ten,
                const float* __restrict__ mask_src, const float* __restrict__ mask_dst,
                const float4* __restrict__ octant /* [(h+1)^3] : w, |rx|, |ry|, |rz| */, TiledParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Entry* list = reinterpret_cast<Entry*>(smem);                                  // BAND_CAP entries
  float4* slice = reinterpret_cast<float4*>(smem + sizeof(Entry) * BAND_CAP);    // (h+1)^2
  __shared__ int wave_tot[NT / 64];
  __shared__ int list_len;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  unsigned b = blockIdx.x;
  const int tile_x = b % p.tiles_x;
  b /= p.tiles_x;
  const int tile_y = b % p.tiles_y;
  const int rz = p.z_out0 + (int)(b / p.tiles_y);
  const int x0 = tile_x * TILE, y0 = tile_y * TILE;
  const int h = p.h;
  const i64 plane = (i64)p.nx * p.ny;
  const i64 nvox = plane * p.nz;

  // receiver of this thread: wave w owns the 8x8 patch (w&1, w>>1)
  const int lx = (wave & 1) * 8 + (lane & 7), ly = (wave >> 1) * 8 + (lane >> 3);
  const int rx = x0 + lx, ry = y0 + ly;
  const int rxr = lx + h, ryr = ly + h;   // region-relative receiver coordinates
  const bool r_in = rx < p.nx && ry < p.ny;
  const i64 rc = (i64)rz * plane + (i64)ry * p.nx + rx;
  const bool r_live = r_in && !(mask_dst && mask_dst[r_in ? rc : 0] == 0.0f);
  const int h2 = h * h;

  float T[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

  const int sz_hi = min(rz + h, p.nz - 1), sz_lo = max(rz - h, 0);
  const int nsteps = (sz_hi - sz_lo + 1) * p.nbands;

  // region voxels owned by this thread inside a band: VPT consecutive positions (row er, column ec)
  int er_[VPT], ec_[VPT];
#pragma unroll
  for (int v = 0; v < VPT; v++) {
    const int q = tid * VPT + v;
    er_[v] = q / p.rw;
    ec_[v] = q - er_[v] * p.rw;
  }
  float pre[VPT];
  auto fetch = [&](int step, float out[VPT]) {
    const int sz = sz_hi - step / p.nbands;
    const int band = p.nbands - 1 - (step % p.nbands);   // bands visited from the last rows down
    const int row0 = band * p.band_rows;
    const int rows = min(p.band_rows, p.rh - row0);
#pragma unroll
    for (int v = 0; v < VPT; v++) {
      float s = 0.0f;
      if (er_[v] < rows) {
        const int er = er_[v], ec = ec_[v];
        const int sx = x0 - h + ec, sy = y0 - h + row0 + er;
        if (sx >= 0 && sx < p.nx && sy >= 0 && sy < p.ny) {
          const i64 s_idx = (i64)sz * plane + (i64)sy * p.nx + sx;
          s = sal[s_idx];
          if (MASKED_SRC && s != 0.0f && mask_src[s_idx] == 0.0f) s = 0.0f;
        }
      }
      out[v] = s;
    }
  };

  if (nsteps > 0) fetch(0, pre);
  for (int step = 0; step < nsteps; step++) {
    const int sz = sz_hi - step / p.nbands;
    const int band = p.nbands - 1 - (step % p.nbands);
    const int row0 = band * p.band_rows;
    const int jz = rz - sz;
    const int az = jz < 0 ? -jz : jz;
    float cur[VPT];
#pragma unroll
    for (int v = 0; v < VPT; v++) cur[v] = pre[v];
    if (step + 1 < nsteps) fetch(step + 1, pre);   // in flight while this band is processed

    // ---- ordered compaction of the band's salient senders into the LDS list ------------------
    int cnt = 0;
#pragma unroll
    for (int v = 0; v < VPT; v++) cnt += (cur[v] != 0.0f) ? 1 : 0;
    int incl = cnt;   // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    // the |jz| slice of the octant table (only when the plane changes)
    if (step % p.nbands == 0) {
      const int nsl = p.hp1 * p.hp1;
      for (int i = tid; i < nsl; i += NT) slice[i] = octant[(i64)az * nsl + i];
    }
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) base += (w < wave) ? wave_tot[w] : 0;
    if (tid == NT - 1) list_len = base + incl;
    int at = base + incl - cnt;
#pragma unroll
    for (int v = 0; v < VPT; v++) {
      if (cur[v] != 0.0f) {
        const int er = er_[v], ec = ec_[v];
        const int sx = x0 - h + ec, sy = y0 - h + row0 + er;
        const i64 s_idx = (i64)sz * plane + (i64)sy * p.nx + sx;
        Entry e;
        e.sal = cur[v];
        e.n0 = dir[s_idx];
        e.n1 = dir[nvox + s_idx];
        e.n2 = dir[2 * nvox + s_idx];
        e.mv = MASKED_SRC ? mask_src[s_idx] : 1.0f;
        e.pos = ((unsigned)(row0 + er) << 8) | (unsigned)ec;
        list[at++] = e;
      }
    }
    __syncthreads();
    const int n_list = list_len;

    // ---- consume the list from its end (descending position = ascending (jy, jx)) ------------
    for (int hi = n_list; hi > 0; hi -= CHUNK) {
      const int n_c = min(CHUNK, hi);
      // phase A: bit k of `hits` <=> entry hi-1-k votes at this receiver
      unsigned long long hits = 0ULL;
      if (r_live) {
        for (int k = 0; k < n_c; k++) {
          const unsigned pos = list[hi - 1 - k].pos;   // same address in every lane: LDS broadcast
          const int jx = rxr - (int)(pos & 0xffu);
          const int jy = ryr - (int)(pos >> 8);
          const int d2 = jx * jx + jy * jy + jz * jz;
          if (d2 <= h2) hits |= (1ULL << k);
        }
      }
      // phase B: every lane drains its own hits in order
      while (hits) {
        const int k = __ffsll((long long)hits) - 1;
        hits &= hits - 1;
        const Entry e = list[hi - 1 - k];
        const int jx = rxr - (int)(e.pos & 0xffu);
        const int jy = ryr - (int)(e.pos >> 8);
        const int ax = jx < 0 ? -jx : jx, ay = jy < 0 ? -jy : jy;
        const float4 t = slice[ay * p.hp1 + ax];
        float fv = t.x;
        if (MASKED_SRC) fv = fv * e.mv;
        if (fv != 0.0f) {
          const float r0 = jx < 0 ? -t.y : t.y;
          const float r1 = jy < 0 ? -t.z : t.z;
          const float r2 = jz < 0 ? -t.w : t.w;
          const float u = (r0 * e.n0 + r1 * e.n1) + r2 * e.n2;
          const float ux2 = u * 2.0f;
          const float u2 = u * u;
          const float c2 = 1.0f - u2;
          const float ang = p.curves ? u2 : c2;
          const float dec = decay_of(ang, p.exponent);
          float m0, m1, m2;
          if (p.curves) {
            m0 = e.n0 - ux2 * r0; m1 = e.n1 - ux2 * r1; m2 = e.n2 - ux2 * r2;
          } else {
            m0 = ux2 * r0 - e.n0; m1 = ux2 * r1 - e.n1; m2 = ux2 * r2 - e.n2;
          }
          const float bse = (e.sal * fv) * dec;
          const float b0 = bse * m0, b1 = bse * m1, b2 = bse * m2;
          T[0] = T[0] + b0 * m0;
          T[3] = T[3] + b0 * m1;
          T[5] = T[5] + b0 * m2;
          T[1] = T[1] + b1 * m1;
          T[4] = T[4] + b1 * m2;
          T[2] = T[2] + b2 * m2;
        }
      }
    }
    __syncthreads();   // the list, the slice and list_len are rewritten by the next step
  }

  if (r_live) {
#pragma unroll
    for (int k = 0; k < 6; k++) ten[k * nvox + rc] = T[k];
  }
}

}  // namespace

int dev_tv_tiled(visfd_hip_ctx* ctx, const float* sal, const float* dir, float* ten,
                 const float* mask_src, const float* mask_dst, i64 nx, i64 ny, i64 nz, i64 z_out0,
                 i64 z_out1, int h, const float* w, const float* rhat, int exponent, bool curves,
                 bool* handled) {
  *handled = false;
  if (h < 1 || h > 40) return VISFD_HIP_OK;  // octant slice + coordinates packing limits
  const int hp1 = h + 1, n = 2 * h + 1;
  // octant table: entry (az, ay, ax) = values at j = (+ax, +ay, +az); rhat(-j) = -rhat(j) and w is
  // even in every component (filter3d.hpp:569-573, feature.hpp:2473-2478)
  std::vector<float4> oct((size_t)hp1 * hp1 * hp1);
  for (int az = 0; az <= h; az++)
    for (int ay = 0; ay <= h; ay++)
      for (int ax = 0; ax <= h; ax++) {
        const size_t k = ((size_t)(az + h) * n + (ay + h)) * n + (ax + h);
        oct[((size_t)az * hp1 + ay) * hp1 + ax] = make_float4(w[k], rhat[3 * k], rhat[3 * k + 1], rhat[3 * k + 2]);
      }
  float4* doct = nullptr;
  VH_TRY(ws(ctx, WS_TVTAB, oct.size(), &doct));
  hipStream_t st = ctx->stream;
  VH_HIP(hipMemcpyAsync(doct, oct.data(), sizeof(float4) * oct.size(), hipMemcpyHostToDevice, st));
  VH_HIP(hipStreamSynchronize(st));

  TiledParams p;
  p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
  p.z_out0 = (int)z_out0;
  p.h = h; p.hp1 = hp1;
  p.rw = TILE + 2 * h; p.rh = TILE + 2 * h;
  p.band_rows = BAND_CAP / p.rw;
  if (p.band_rows > p.rh) p.band_rows = p.rh;
  p.nbands = (p.rh + p.band_rows - 1) / p.band_rows;
  p.tiles_x = (int)((nx + TILE - 1) / TILE);
  p.tiles_y = (int)((ny + TILE - 1) / TILE);
  p.exponent = exponent;
  p.curves = curves ? 1 : 0;
  const i64 nblk = (i64)p.tiles_x * p.tiles_y * (z_out1 - z_out0);
  if (nblk > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  const size_t lds = sizeof(Entry) * BAND_CAP + sizeof(float4) * (size_t)hp1 * hp1;
  if (mask_src) {
    VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tv_tiled_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    tv_tiled_kernel<true><<<dim3((unsigned)nblk), dim3(NT), lds, st>>>(sal, dir, ten, mask_src, mask_dst,
                                                                      doct, p);
  } else {
    VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tv_tiled_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    tv_tiled_kernel<false><<<dim3((unsigned)nblk), dim3(NT), lds, st>>>(sal, dir, ten, mask_src, mask_dst,
                                                                       doct, p);
  }
  VH_HIP(hipGetLastError());
  *handled = true;
  return VISFD_HIP_OK;
}

}  // namespace vh
