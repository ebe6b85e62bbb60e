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
constexpr int BAND_CAP = NT * VPT;   // 1792 region voxels per band
constexpr int CAP = 512;             // list entries held in LDS at a time (a band may need several passes)
constexpr int CHUNK = 64;
constexpr unsigned OOB = 0x7ffffff0u;  // byte offset beyond any plane descriptor: reads give 0

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, 0));
}

struct TiledParams {
  int nx, ny, nz;
  int z_out0;
  int h, hp1;            // halfwidth, h+1
  int rw, rh;            // region width/height = TILE + 2h
  int band_rows, nbands;
  int tiles_x, tiles_y;
  int exponent, curves;
};

// MODE 0: surfaces with angular exponent 4 (the CLI default, settings.cpp:154); MODE 1: general.
template <int MODE>
__device__ __forceinline__ void vote(float T[6], float sal, float fv, float r0, float r1, float r2,
                                     float n0, float n1, float n2, int exponent, int curves) {
  const float u = (r0 * n0 + r1 * n1) + r2 * n2;
  const float ux2 = u * 2.0f;
  const float u2 = u * u;
  const float c2 = 1.0f - u2;
  float dec, m0, m1, m2;
  if (MODE == 0) {
    dec = c2 * c2;
    m0 = ux2 * r0 - n0; m1 = ux2 * r1 - n1; m2 = ux2 * r2 - n2;
  } else {
    const float ang = curves ? u2 : c2;
    if (exponent == 4) dec = ang * ang;
    else if (exponent == 2) dec = ang;
    else dec = (float)pow((double)ang, 0.5 * (double)exponent);
    if (curves) { m0 = n0 - ux2 * r0; m1 = n1 - ux2 * r1; m2 = n2 - ux2 * r2; }
    else        { m0 = ux2 * r0 - n0; m1 = ux2 * r1 - n1; m2 = ux2 * r2 - n2; }
  }
  const float bse = (sal * fv) * dec;
  const float b0 = bse * m0, b1 = bse * m1, b2 = bse * m2;
  T[0] = T[0] + b0 * m0;
  T[3] = T[3] + b0 * m1;
  T[5] = T[5] + b0 * m2;
  T[1] = T[1] + b1 * m1;
  T[4] = T[4] + b1 * m2;
  T[2] = T[2] + b2 * m2;
}

template <bool MASKED_SRC, int MODE>
__global__ void __launch_bounds__(NT)
tv_tiled_kernel(const float* __restrict__ sal, const float* __restrict__ dir, float* __restrict__ ten,
                const float* __restrict__ mask_src, const float* __restrict__ mask_dst,
                const float4* __restrict__ octant /* [(h+1)^3] : w, |rx|, |ry|, |rz| */, TiledParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS: sender list of the current pass (structure of arrays) + two |jz| slices of the octant table
  float4* l_dat = reinterpret_cast<float4*>(smem);                         // sal, n0, n1, n2
  unsigned* l_pos = reinterpret_cast<unsigned*>(smem + 16 * CAP);          // (ey << 8) | ex
  float* l_mv = reinterpret_cast<float*>(smem + 20 * CAP);                 // source-mask value
  float4* slices = reinterpret_cast<float4*>(smem + 24 * CAP);             // 2 x (h+1)^2
  __shared__ int wave_tot[2][NT / 64];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  unsigned b = blockIdx.x;
  const int tile_x = b % p.tiles_x;
  b /= p.tiles_x;
  const int tile_y = b % p.tiles_y;
  const int rz = p.z_out0 + (int)(b / p.tiles_y);
  const int x0 = tile_x * TILE, y0 = tile_y * TILE;
  const int h = p.h;
  const int nsl = p.hp1 * p.hp1;
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

  // region voxels owned by this thread inside a band: VPT consecutive positions, (row << 8) | column
  int rc_[VPT];
#pragma unroll
  for (int v = 0; v < VPT; v++) {
    const int q = tid * VPT + v;
    const int er = q / p.rw;
    rc_[v] = (er << 8) | (q - er * p.rw);
  }
  // Region reads are buffer loads with hardware range checking: a per-plane descriptor (scalar) plus
  // a 32-bit byte offset per voxel; voxels outside the image or the band use an out-of-range offset
  // and read as 0.0f (= not salient) without branches.
  const int plane_bytes = (int)(plane * 4);
  auto plane_rsrc = [&](const float* base, int sz) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (i64)sz * plane), 0, plane_bytes, 0x00020000);
  };
  auto voff_of = [&](int v, int row0, int rows) -> unsigned {
    const int er = rc_[v] >> 8, ec = rc_[v] & 0xff;
    const int sx = x0 - h + ec, sy = y0 - h + row0 + er;
    const bool ok = er < rows && sx >= 0 && sx < p.nx && sy >= 0 && sy < p.ny;
    return ok ? (unsigned)(sy * p.nx + sx) * 4u : OOB;
  };
  float pre[VPT];
  auto fetch = [&](int step, float out[VPT]) {
    const int sz = sz_hi - step / p.nbands;
    const int band = p.nbands - 1 - (step % p.nbands);   // bands visited from the last rows down
    const int row0 = band * p.band_rows;
    const int rows = min(p.band_rows, p.rh - row0);
    const __amdgpu_buffer_rsrc_t rs = plane_rsrc(sal, sz);
#pragma unroll
    for (int v = 0; v < VPT; v++) {
      const unsigned off = voff_of(v, row0, rows);
      float s = buf_load(rs, off);
      if (MASKED_SRC) {
        const float m = buf_load(plane_rsrc(mask_src, sz), off);
        if (m == 0.0f) s = 0.0f;
      }
      out[v] = s;
    }
  };

  if (nsteps > 0) fetch(0, pre);
  for (int step = 0; step < nsteps; step++) {
    const int par = step & 1;
    const int sz = sz_hi - step / p.nbands;
    const int band = p.nbands - 1 - (step % p.nbands);
    const int row0 = band * p.band_rows;
    const int jz = rz - sz;
    const int az = jz < 0 ? -jz : jz;
    const float4* slice = slices + ((step / p.nbands) & 1) * nsl;
    float cur[VPT];
#pragma unroll
    for (int v = 0; v < VPT; v++) cur[v] = pre[v];
    if (step + 1 < nsteps) fetch(step + 1, pre);   // in flight while this band is processed

    // ---- ordered compaction of the band's salient senders ------------------------------------
    int cnt = 0;
#pragma unroll
    for (int v = 0; v < VPT; v++) cnt += (cur[v] != 0.0f) ? 1 : 0;
    int incl = cnt;   // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[par][wave] = incl;
    // the |jz| slice of the octant table (when the plane changes); double-buffered like wave_tot so
    // that waves still consuming the previous step are not disturbed
    if (step % p.nbands == 0) {
      float4* dst = slices + ((step / p.nbands) & 1) * nsl;
      for (int i = tid; i < nsl; i += NT) dst[i] = octant[(i64)az * nsl + i];
    }
    __syncthreads();   // (1) every wave has finished consuming the previous step's list
    int base = 0, len = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
      const int t = wave_tot[par][w];
      base += (w < wave) ? t : 0;
      len += t;
    }
    const int first_rank = base + incl - cnt;   // rank of this thread's first sender in the band list

    // The band's list (ranks 0..len-1, ascending position) is walked from its end in passes of at
    // most CAP entries: one pass unless more than CAP senders of the region are salient.
    for (int pass_hi = len; pass_hi > 0; pass_hi -= CAP) {
      const int pass_lo = max(0, pass_hi - CAP);
      if (pass_hi != len) __syncthreads();  // previous pass fully consumed before the list is rewritten
      {
        const int rows = min(p.band_rows, p.rh - row0);
        const __amdgpu_buffer_rsrc_t rd0 = plane_rsrc(dir, sz);
        const __amdgpu_buffer_rsrc_t rd1 = plane_rsrc(dir + nvox, sz);
        const __amdgpu_buffer_rsrc_t rd2 = plane_rsrc(dir + 2 * nvox, sz);
        int r = first_rank;
#pragma unroll
        for (int v = 0; v < VPT; v++) {
          if (cur[v] != 0.0f) {
            if (r >= pass_lo && r < pass_hi) {
              const unsigned off = voff_of(v, row0, rows);
              const int slot = r - pass_lo;
              l_dat[slot] = make_float4(cur[v], buf_load(rd0, off), buf_load(rd1, off), buf_load(rd2, off));
              l_pos[slot] = ((unsigned)(row0 + (rc_[v] >> 8)) << 8) | (unsigned)(rc_[v] & 0xff);
              if (MASKED_SRC) l_mv[slot] = buf_load(plane_rsrc(mask_src, sz), off);
            }
            r++;
          }
        }
      }
      __syncthreads();   // (2) list visible
      const int n_pass = pass_hi - pass_lo;   // entries 0..n_pass-1, ascending position

      // ---- consume from the end (descending position = ascending (jy, jx)) -------------------
      for (int hi = n_pass; hi > 0; hi -= CHUNK) {
        const int n_c = min(CHUNK, hi);
        // phase A: lane l fetches the position of chunk entry l; every lane then tests all n_c
        // senders (broadcast through v_readlane) against its own receiver.  bit k <=> entry hi-1-k.
        const unsigned mypos = (lane < n_c) ? l_pos[hi - 1 - lane] : 0u;
        unsigned hits_lo = 0u, hits_hi = 0u;
        const int lim2 = h2 - jz * jz;   // jx^2 + jy^2 <= h^2 - jz^2
#pragma unroll 1
        for (int k0 = 0; k0 < n_c; k0 += 8) {   // wave-uniform trip count; 8 tests per trip
          unsigned bits = 0u;
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const unsigned ps = (unsigned)__builtin_amdgcn_readlane((int)mypos, k0 + j);  // lanes >= n_c hold 0
            const int jx = rxr - (int)(ps & 0xffu);
            const int jy = ryr - (int)(ps >> 8);
            const int d2 = jx * jx + jy * jy;
            bits |= (d2 <= lim2) ? (1u << j) : 0u;
          }
          // entries beyond n_c (k0+j >= n_c) decode position 0, which may look like a hit: mask them
          const int valid = n_c - k0;
          if (valid < 8) bits &= (1u << valid) - 1u;
          if (k0 < 32) hits_lo |= bits << (k0 & 31); else hits_hi |= bits << (k0 & 31);
        }
        unsigned long long hits = r_live ? (((unsigned long long)hits_hi << 32) | hits_lo) : 0ULL;
        // phase B: every lane drains its own hits in order, loads of the next hit in flight
        // while the current vote is accumulated
        float4 d_c, t_c; float mv_c = 1.0f; int jx_c = 0, jy_c = 0;
        auto load_hit = [&](int k, float4& d, float4& t, float& mvv, int& jx, int& jy) {
          const int idx = hi - 1 - k;
          const unsigned ps = l_pos[idx];
          d = l_dat[idx];
          if (MASKED_SRC) mvv = l_mv[idx];
          jx = rxr - (int)(ps & 0xffu);
          jy = ryr - (int)(ps >> 8);
          const int ax = jx < 0 ? -jx : jx, ay = jy < 0 ? -jy : jy;
          t = slice[ay * p.hp1 + ax];
        };
        bool have = hits != 0ULL;
        if (have) {
          const int k = __ffsll((long long)hits) - 1;
          hits &= hits - 1;
          load_hit(k, d_c, t_c, mv_c, jx_c, jy_c);
        }
        while (have) {
          float4 d_n, t_n; float mv_n = 1.0f; int jx_n = 0, jy_n = 0;
          const bool more = hits != 0ULL;
          if (more) {
            const int k = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            load_hit(k, d_n, t_n, mv_n, jx_n, jy_n);
          }
          float fv = t_c.x;
          if (MASKED_SRC) fv = fv * mv_c;
          if (fv != 0.0f) {
            const float r0 = jx_c < 0 ? -t_c.y : t_c.y;
            const float r1 = jy_c < 0 ? -t_c.z : t_c.z;
            const float r2 = jz < 0 ? -t_c.w : t_c.w;
            vote<MODE>(T, d_c.x, fv, r0, r1, r2, d_c.y, d_c.z, d_c.w, p.exponent, p.curves);
          }
          have = more;
          d_c = d_n; t_c = t_n; mv_c = mv_n; jx_c = jx_n; jy_c = jy_n;
        }
      }
    }
    // no barrier here: wave_tot and the slices are double-buffered, and barrier (1) of the next step
    // orders every wave's consumption of this list before it is rewritten
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
  if (nx * ny >= (1LL << 29)) return VISFD_HIP_OK;  // plane descriptors are 32-bit
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
  const size_t lds = (size_t)24 * CAP + 2 * sizeof(float4) * (size_t)hp1 * hp1;
  const int mode = (exponent == 4 && !curves) ? 0 : 1;
#define VH_TV_LAUNCH(MSK, MD)                                                                        \
  do {                                                                                               \
    VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tv_tiled_kernel<MSK, MD>),             \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
    tv_tiled_kernel<MSK, MD><<<dim3((unsigned)nblk), dim3(NT), lds, st>>>(sal, dir, ten, mask_src,   \
                                                                         mask_dst, doct, p);         \
  } while (0)
  if (mask_src) { if (mode == 0) VH_TV_LAUNCH(true, 0); else VH_TV_LAUNCH(true, 1); }
  else          { if (mode == 0) VH_TV_LAUNCH(false, 0); else VH_TV_LAUNCH(false, 1); }
#undef VH_TV_LAUNCH
  VH_HIP(hipGetLastError());
  *handled = true;
  return VISFD_HIP_OK;
}

}  // namespace vh
