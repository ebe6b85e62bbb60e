// blob.hip -- 4-D (x,y,z,scale) strict non-max / non-min scan of the scale-space blob detector
// (reference lib/visfd/feature.hpp:212-346).
//
// A voxel of the middle LoG volume is a minimum iff it is strictly smaller than all 80
// neighbours in (x,y,z,scale), all 26 spatial neighbours are inside the image and unmasked, the
// voxel itself is unmasked, its score is negative and passes the (absolute) threshold; maxima are
// symmetric (feature.hpp:231-304).  Float comparisons are exact, so indices match the reference
// bit-for-bit whenever the three LoG volumes do.
//
// Two kernels:
//   1. candidates: an HBM sweep of the MIDDLE volume only (4 B/voxel).  A workgroup owns a 64 x 8
//      column tile and marches along z four planes at a time (the next four already requested); each
//      plane goes through an LDS tile with a one-voxel halo,
//      each thread takes the min and max of its 3x3 neighbourhood, and a three-plane register ring
//      gives the 3x3x3 box min/max.  A voxel is a candidate when it equals the box min (or max)
//      and passes the sign/threshold tests -- a non-strict superset of the 26-neighbour rule.
//   2. verify: one thread per candidate repeats the reference's exact test on all 80 neighbours
//      (strict comparisons, masks) and appends the survivors.
// LoG volumes are smooth, so candidates are a small fraction of the voxels and kernel 2 is cheap.
#include <algorithm>
#include <cstdlib>
#include <limits>

#include "common.hpp"

namespace vh {

namespace {

constexpr int BLOCK = 512;
constexpr int TX = 64, TY = 8;          // one voxel column per thread
constexpr int PZ = 4;                   // planes staged per step (and between flushes of the candidate buffer)
constexpr int BUFCAP = 2 * PZ * BLOCK;  // candidate buffer (32-bit tile-relative codes): flushed when a step might not fit
constexpr int LW = TX + 2, LH = TY + 2; // LDS tile with halo

static_assert(BLOCK == 512 && TX == 64, "candidate codes pack the thread index in 9 bits");

struct Cand {
  int ix, iy, iz;
  int kind;  // 0 = minimum, 1 = maximum
  float score;
};

__global__ void __launch_bounds__(BLOCK)
blob_candidates_kernel(const float* __restrict__ mid, const float* __restrict__ mask, int nx, int ny, int nz,
                       float min_thr, float max_thr, int zchunk, int tiles_x, int tiles_y,
                       unsigned long long* __restrict__ out, unsigned long long capacity,
                       unsigned long long* __restrict__ counter) {
  constexpr int NLD = (LH * LW + BLOCK - 1) / BLOCK;
  __shared__ float tile[PZ][NLD * BLOCK];   // (LH * LW used; padded so that every thread's stores are unconditional)
  // candidates are collected per workgroup and written out with ONE global atomic per flush
  __shared__ unsigned int buf[BUFCAP];   // (z - zs) << 9 | thread index: one voxel of this workgroup's column
  __shared__ unsigned int buf_n;
  __shared__ unsigned long long buf_base;
  if (threadIdx.x == 0) buf_n = 0;
  unsigned b = blockIdx.x;
  const int tile_x = b % tiles_x;
  b /= tiles_x;
  const int tile_y = b % tiles_y;
  const int chunk = b / tiles_y;
  const int x0 = tile_x * TX, y0 = tile_y * TY;
  const int zs = max(1, chunk * zchunk), ze = min(nz - 1, (chunk + 1) * zchunk);  // interior planes only
  if (zs >= ze) return;
  const int tid = threadIdx.x;
  const int lx = tid & (TX - 1), ly = tid >> 6;
  const int gx = x0 + lx, gy = y0 + ly;
  const i64 plane = (i64)nx * ny;
  const bool interior = gx >= 1 && gx <= nx - 2 && gy >= 1 && gy <= ny - 2;

  // cooperative load of one haloed plane (values outside the image are irrelevant: they only influence
  // face voxels, which can never be blobs, feature.hpp:245-252).  Split in two so that the global loads
  // of plane z+2 are in flight while plane z+1 is being compared (the march is latency-bound otherwise):
  // fetch() reads this thread's (up to NLD) tile elements into registers, stash() writes them to LDS.
  // Raw buffer loads through one descriptor per plane (planes below 2 GiB: the host checks): an element outside the image
  // or the tile has an out-of-range offset, a plane outside the volume a zero-length descriptor -- the hardware returns
  // 0.0 and the loads need no branches.  (Conditional loads are each closed by a full s_waitcnt at their join: the eight
  // loads of a step then went out in three or four dependent groups, and the sweep ran at memory latency.)
  unsigned ld_off[NLD];   // BYTE offset inside a plane
#pragma unroll
  for (int k = 0; k < NLD; k++) {
    const int i = tid + k * BLOCK;
    const int r = i / LW, c = i - r * LW;
    const int sx = x0 - 1 + c, sy = y0 - 1 + r;
    ld_off[k] = (i < LH * LW && sx >= 0 && sx < nx && sy >= 0 && sy < ny) ? (unsigned)(sy * nx + sx) * 4u : 0x7ffffff0u;
  }
  const int plane_bytes = (int)(plane * 4);
  auto fetch = [&](int z, float v[NLD]) {
    const bool zin = z < nz;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(mid + (zin ? (i64)z * plane : 0)), 0,
                                                                        zin ? plane_bytes : 0, 0x00020000);
#pragma unroll
    for (int k = 0; k < NLD; k++) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)ld_off[k], 0, 0));
  };
  auto stash = [&](const float v[NLD], float* dst) {
#pragma unroll
    for (int k = 0; k < NLD; k++) dst[tid + k * BLOCK] = v[k];
  };
  // min and max over the 3x3 neighbourhood of (lx, ly) in an LDS plane, plus the centre value
  auto minmax9 = [&](const float* t, float& mn, float& mx, float& centre) {
    const float* p0 = t + ly * LW + lx;  // top-left of the 3x3 window (halo offset 1 folded in)
    float a = p0[0], bb = p0[1], c = p0[2];
    mn = fminf(fminf(a, bb), c); mx = fmaxf(fmaxf(a, bb), c);
    a = p0[LW]; bb = p0[LW + 1]; c = p0[LW + 2];
    centre = bb;
    mn = fminf(mn, fminf(fminf(a, bb), c)); mx = fmaxf(mx, fmaxf(fmaxf(a, bb), c));
    a = p0[2 * LW]; bb = p0[2 * LW + 1]; c = p0[2 * LW + 2];
    mn = fminf(mn, fminf(fminf(a, bb), c)); mx = fmaxf(mx, fmaxf(fmaxf(a, bb), c));
  };

  // March in steps of PZ planes: the PZ planes after the current one are staged in LDS together (one barrier
  // per step instead of one per plane) while the global loads of the following PZ planes are already in
  // flight in registers, so the HBM latency is covered by PZ planes of comparisons.
  float mn_prev, mx_prev, c_prev, mn_cur, mx_cur, c_cur;
  // TWO steps of planes in flight, in two register sets used alternately (a step's loads are requested two steps before
  // they are stashed: one step of comparisons does not cover a loaded HBM round trip)
  float preA[PZ][NLD], preB[PZ][NLD];
  fetch(zs - 1, preA[0]);
  fetch(zs, preA[1]);
  stash(preA[0], tile[0]);
  stash(preA[1], tile[1]);
#pragma unroll
  for (int k = 0; k < PZ; k++) fetch(zs + 1 + k, preA[k]);
#pragma unroll
  for (int k = 0; k < PZ; k++) fetch(zs + PZ + 1 + k, preB[k]);
  __syncthreads();
  minmax9(tile[0], mn_prev, mx_prev, c_prev);
  minmax9(tile[1], mn_cur, mx_cur, c_cur);
  __syncthreads();   // both tiles free again
  // one step: planes z0 .. z0+PZ-1 are compared; `pre` holds planes z0+1 .. z0+PZ and is refilled with z0+2PZ+1 ..
  auto step = [&](float (&pre)[PZ][NLD], int z0) {
#pragma unroll
    for (int k = 0; k < PZ; k++) stash(pre[k], tile[k]);                      // planes z0+1 .. z0+PZ
#pragma unroll
    for (int k = 0; k < PZ; k++) fetch(z0 + 2 * PZ + 1 + k, pre[k]);          // the planes of the step after the next
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PZ; k++) {
      const int z = z0 + k;
      if (z < ze) {   // uniform
        float mn_nxt, mx_nxt, c_nxt;
        minmax9(tile[k], mn_nxt, mx_nxt, c_nxt);
        const float e = c_cur;
        const float bmin = fminf(fminf(mn_prev, mn_cur), mn_nxt);
        const float bmax = fmaxf(fmaxf(mx_prev, mx_cur), mx_nxt);
        const bool is_min = (e == bmin) && (e < 0.0f) && (e < min_thr);
        const bool is_max = (e == bmax) && (e > 0.0f) && (e > max_thr);
        {  // append with ONE LDS atomic per wave (ballot + prefix count) instead of one per candidate
          const i64 v = (i64)z * plane + (i64)gy * nx + gx;
          bool cand = interior && (is_min || is_max);
          if (cand && mask && mask[v] == 0.0f) cand = false;
          const unsigned long long bal = __ballot(cand);
          if (bal != 0ull) {
            const int lane = tid & 63;
            const int leader = __ffsll((long long)bal) - 1;
            unsigned int base = 0;
            if (lane == leader) base = atomicAdd(&buf_n, (unsigned int)__popcll(bal));
            base = (unsigned int)__shfl((int)base, leader);
            if (cand) buf[base + (unsigned int)__popcll(bal & ((1ull << lane) - 1ull))] = ((unsigned int)(z - zs) << 9) | (unsigned int)tid;
          }
        }
        mn_prev = mn_cur; mx_prev = mx_cur;
        mn_cur = mn_nxt; mx_cur = mx_nxt; c_cur = c_nxt;
      }
    }
    // Flush the workgroup's candidates with ONE global atomic -- and only when the next step might not fit:
    // every workgroup of the grid adds to the same counter, and atomics on one address are served one
    // at a time by L2 (flushing every step made this kernel 3x slower than its memory traffic).
    // The first barrier also keeps the next step's stash from overwriting tiles that are still being read.
    __syncthreads();
    const unsigned int n = buf_n;
    const bool last = z0 + PZ >= ze;
    if (n > (unsigned int)(BUFCAP - PZ * BLOCK) || (last && n)) {   // uniform
      if (tid == 0) buf_base = atomicAdd(counter, (unsigned long long)n);
      __syncthreads();
      const unsigned long long base = buf_base;
      for (unsigned int i = tid; i < n; i += BLOCK)
        if (base + i < capacity) {
          const unsigned int code = buf[i];
          const int t = (int)(code & 511u), zz = zs + (int)(code >> 9);
          out[base + i] = (unsigned long long)((i64)zz * plane + (i64)(y0 + (t >> 6)) * nx + (x0 + (t & (TX - 1))));
        }
      __syncthreads();
      if (tid == 0) buf_n = 0;   // ordered before the next appends by the next step's barrier
    }
  };
  for (int z0 = zs; z0 < ze; z0 += 2 * PZ) {
    step(preA, z0);
    step(preB, z0 + PZ);   // (unconditional -- a second step past the end compares nothing: with a branch here the wait counts
                           //  at the loop head must assume the shorter path and drain the younger set's loads as well)
  }
}

__global__ void __launch_bounds__(BLOCK)
blob_verify_kernel(const unsigned long long* __restrict__ cand_idx, const unsigned long long* __restrict__ n_cand_ptr,
                   unsigned long long cand_capacity,
                   const float* __restrict__ lo, const float* __restrict__ mid, const float* __restrict__ hi,
                   const float* __restrict__ mask, int nx, int ny, int nz, float min_thr, float max_thr,
                   Cand* __restrict__ out, unsigned long long capacity, unsigned long long* __restrict__ counter) {
  // the number of candidates is read on the device (no host round trip between the two kernels); an overflowed
  // candidate list is cut at its capacity here and reported by the host afterwards
  unsigned long long n_cand = *n_cand_ptr;
  if (n_cand > cand_capacity) n_cand = cand_capacity;
  for (unsigned long long t = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; t < n_cand;
       t += (unsigned long long)gridDim.x * BLOCK) {
  const i64 c = (i64)cand_idx[t];
  const i64 plane = (i64)nx * ny;
  const int iz = (int)(c / plane);
  const int rem = (int)(c - (i64)iz * plane);
  const int iy = rem / nx, ix = rem - iy * nx;
  const float e = mid[c];
  bool is_min = (e < 0.0f) && (e < min_thr);
  bool is_max = (e > 0.0f) && (e > max_thr);
  // the two other-scale values at the same position decide most candidates: test them first
  {
    const float a = lo[c], b = hi[c];
    if (a <= e || b <= e) is_min = false;
    if (a >= e || b >= e) is_max = false;
  }
  const float* vol[3] = {mid, lo, hi};
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
  }   // next candidate of this thread
}

}  // namespace

// The scan of one scale in two halves, so that a caller can queue the next scale's filters before it waits for
// this scale's list (BlobDog runs a dozen scales back to back): blob_scan_launch only enqueues the two kernels on
// the context's stream (buffer set 0 or 1: candidate codes, survivors, counters) and records an event;
// blob_scan_collect waits for that event on an auxiliary stream, copies the survivors to the host there (the main
// stream keeps running), sorts them and appends them to the lists.  Returns 1 when a buffer overflowed (the caller
// then repeats the scale with dev_blob_scan, which grows the buffers).
struct ScanBufs {
  unsigned long long* idx = nullptr;
  Cand* cand = nullptr;
  unsigned long long* counters = nullptr;   // [0]: candidates, [1]: verified
  size_t cap_idx = 0, cap_out = 0;
};

static int scan_bufs(visfd_hip_ctx* ctx, int set, ScanBufs* B, bool pipelined, i64 nvox) {
  // at least 4 M candidates / 1 M survivors, and room for 1 voxel in 32 / 128 (noise volumes at 1024^3 give ~1 in
  // 100 / 1 in 4000 per scale): the pipelined scan then does not overflow on its first call
  constexpr size_t NSET = 3;   // buffer sets of the pipelined scan (api.hip: blob_dog_dev)
  size_t cap_idx = ctx->slot_bytes[WS_TVAUX] / sizeof(unsigned long long) / NSET;
  if (cap_idx < (1u << 22)) cap_idx = 1u << 22;
  if (cap_idx < (size_t)(nvox / 32)) cap_idx = (size_t)(nvox / 32);
  size_t cap_out = ctx->slot_bytes[WS_CAND] / sizeof(Cand) / NSET;
  if (cap_out < (1u << 20)) cap_out = 1u << 20;
  if (cap_out < (size_t)(nvox / 128)) cap_out = (size_t)(nvox / 128);
  unsigned long long* idx = nullptr;
  Cand* cand = nullptr;
  unsigned long long* counters = nullptr;
  VH_TRY(ws(ctx, WS_TVAUX, NSET * cap_idx, &idx));
  VH_TRY(ws(ctx, WS_CAND, NSET * cap_out, &cand));
  VH_TRY(ws(ctx, WS_SCANCNT, 8, &counters));   // (2 words per set)
  B->idx = idx + (size_t)set * cap_idx;
  B->cand = cand + (size_t)set * cap_out;
  B->counters = counters + 2 * set;
  B->cap_idx = cap_idx;
  B->cap_out = cap_out;
  if (pipelined) {   // test hook: pretend the buffers of the pipelined form are tiny, so that its overflow path runs
    if (ctx->opt.blob_test_cap > 0) {
      const size_t v = (size_t)ctx->opt.blob_test_cap;
      if (v < B->cap_idx) B->cap_idx = v;
      if (v < B->cap_out) B->cap_out = v;
    }
  }
  return VISFD_HIP_OK;
}

static void sort_and_append(std::vector<Cand>& h, i64 nx, i64 ny, int scale_index, float sigma,
                            std::vector<visfd_hip_blob>* minima, std::vector<visfd_hip_blob>* maxima) {
  // deterministic order (the device appends in arrival order): by (iz, iy, ix), i.e. by linear voxel index.
  // LSD radix sort of (index, position) pairs: lists reach 250 k entries per scale at 1024^3, where a
  // comparison sort of the records cost 12 ms.
  const size_t m = h.size();
  std::vector<unsigned long long> key(m), key2(m);
  std::vector<unsigned int> pos(m), pos2(m);
  unsigned long long maxkey = 0;
  for (size_t i = 0; i < m; i++) {
    key[i] = (unsigned long long)(((i64)h[i].iz * ny + h[i].iy) * nx + h[i].ix);
    pos[i] = (unsigned int)i;
    if (key[i] > maxkey) maxkey = key[i];
  }
  constexpr int RB = 11;
  for (int shift = 0; shift < 64 && (maxkey >> shift) != 0; shift += RB) {
    size_t hist[(1 << RB) + 1] = {0};
    for (size_t i = 0; i < m; i++) hist[((key[i] >> shift) & ((1u << RB) - 1)) + 1]++;
    for (int b = 0; b < (1 << RB); b++) hist[b + 1] += hist[b];
    for (size_t i = 0; i < m; i++) {
      const size_t d = hist[(key[i] >> shift) & ((1u << RB) - 1)]++;
      key2[d] = key[i];
      pos2[d] = pos[i];
    }
    key.swap(key2);
    pos.swap(pos2);
  }
  for (size_t i = 0; i < m; i++) {
    const Cand& cd = h[pos[i]];
    visfd_hip_blob bl;
    bl.ix = cd.ix; bl.iy = cd.iy; bl.iz = cd.iz;
    bl.scale = scale_index;
    bl.sigma = sigma;
    bl.score = cd.score;
    (cd.kind == 0 ? minima : maxima)->push_back(bl);
  }
}

static int scan_enqueue(visfd_hip_ctx* ctx, const ScanBufs& B, const float* lo, const float* mid, const float* hi,
                        const float* mask, i64 nx, i64 ny, i64 nz, float min_thr, float max_thr) {
  hipStream_t st = ctx->stream;
  if (nx * ny >= (1LL << 29)) return fail(VISFD_HIP_EINVAL, "planes of 2 GiB and more are not supported by the blob scan");
  const int tiles_x = (int)((nx + TX - 1) / TX), tiles_y = (int)((ny + TY - 1) / TY);
  const i64 tiles = (i64)tiles_x * tiles_y;
  i64 want_chunks = ((i64)ctx->num_cus * 16 + tiles - 1) / tiles;
  if (want_chunks < 1) want_chunks = 1;
  i64 zchunk = (nz + want_chunks - 1) / want_chunks;
  if (zchunk < 16) zchunk = 16;
  const i64 nchunks = (nz + zchunk - 1) / zchunk;
  const i64 nblocks = tiles * nchunks;
  if (nblocks > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  VH_HIP(hipMemsetAsync(B.counters, 0, 2 * sizeof(unsigned long long), st));
  blob_candidates_kernel<<<dim3((unsigned)nblocks), dim3(BLOCK), 0, st>>>(
      mid, mask, (int)nx, (int)ny, (int)nz, min_thr, max_thr, (int)zchunk, tiles_x, tiles_y, B.idx,
      (unsigned long long)B.cap_idx, B.counters);
  VH_HIP(hipGetLastError());
  blob_verify_kernel<<<dim3((unsigned)(ctx->num_cus * 8)), dim3(BLOCK), 0, st>>>(
      B.idx, B.counters, (unsigned long long)B.cap_idx, lo, mid, hi, mask, (int)nx, (int)ny, (int)nz, min_thr, max_thr,
      B.cand, (unsigned long long)B.cap_out, B.counters + 1);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

int blob_scan_launch(visfd_hip_ctx* ctx, int set, hipEvent_t done, const float* lo, const float* mid, const float* hi,
                     const float* mask, i64 nx, i64 ny, i64 nz, float min_thr, float max_thr) {
  ScanBufs B;
  VH_TRY(scan_bufs(ctx, set, &B, true, nx * ny * nz));
  VH_TRY(scan_enqueue(ctx, B, lo, mid, hi, mask, nx, ny, nz, min_thr, max_thr));
  VH_HIP(hipEventRecord(done, ctx->stream));
  return VISFD_HIP_OK;
}

int blob_scan_collect(visfd_hip_ctx* ctx, int set, hipEvent_t done, hipStream_t aux, i64 nx, i64 ny, i64 nvox,
                      int scale_index,
                      float sigma, std::vector<visfd_hip_blob>* minima, std::vector<visfd_hip_blob>* maxima,
                      bool* overflow) {
  ScanBufs B;
  VH_TRY(scan_bufs(ctx, set, &B, true, nvox));   // (sizes unchanged since the launch: same pointers)
  *overflow = false;
  VH_HIP(hipStreamWaitEvent(aux, done, 0));
  unsigned long long c2[2] = {0, 0};
  VH_HIP(hipMemcpyAsync(c2, B.counters, sizeof(c2), hipMemcpyDeviceToHost, aux));
  VH_HIP(hipStreamSynchronize(aux));
  if (ctx->opt.debug) fprintf(stderr, "[blob scan] scale %d: %llu candidates, %llu blobs\n", scale_index, c2[0], c2[1]);
  if (c2[0] > B.cap_idx || c2[1] > B.cap_out) { *overflow = true; return VISFD_HIP_OK; }
  std::vector<Cand> h((size_t)c2[1]);
  if (c2[1]) {
    VH_HIP(hipMemcpyAsync(h.data(), B.cand, sizeof(Cand) * (size_t)c2[1], hipMemcpyDeviceToHost, aux));
    VH_HIP(hipStreamSynchronize(aux));
  }
  sort_and_append(h, nx, ny, scale_index, sigma, minima, maxima);
  return VISFD_HIP_OK;
}

// One scale, synchronously (also the fallback of the pipelined form: grows its buffers until the lists fit).
int dev_blob_scan(visfd_hip_ctx* ctx, const float* lo, const float* mid, const float* hi,
                  const float* mask, i64 nx, i64 ny, i64 nz, int scale_index, float sigma,
                  float min_thr, float max_thr, bool want_min, bool want_max,
                  std::vector<visfd_hip_blob>* minima, std::vector<visfd_hip_blob>* maxima) {
  if (nx < 3 || ny < 3 || nz < 3) return VISFD_HIP_OK;  // no interior voxels: nothing can be a blob
  if (nx >= (1LL << 31) || ny >= (1LL << 31) || nz >= (1LL << 31))
    return fail(VISFD_HIP_EINVAL, "dimension too large");
  const float inf = std::numeric_limits<float>::infinity();
  if (!want_min) min_thr = -inf;   // nothing is < -inf
  if (!want_max) max_thr = inf;
  hipStream_t st = ctx->stream;
  for (int attempt = 0; attempt < 4; attempt++) {
    ScanBufs B;
    VH_TRY(scan_bufs(ctx, 0, &B, false, nx * ny * nz));
    VH_TRY(scan_enqueue(ctx, B, lo, mid, hi, mask, nx, ny, nz, min_thr, max_thr));
    unsigned long long c2[2] = {0, 0};
    VH_HIP(hipMemcpyAsync(c2, B.counters, sizeof(c2), hipMemcpyDeviceToHost, st));
    VH_HIP(hipStreamSynchronize(st));
    if (ctx->opt.debug) fprintf(stderr, "[blob scan] scale %d: %llu candidates\n", scale_index, c2[0]);
    if (c2[0] > B.cap_idx) {   // rare: grow and rescan
      unsigned long long* p = nullptr;
      VH_TRY(ws(ctx, WS_TVAUX, 3 * (size_t)c2[0] + 16, &p));   // (a third of the slot per buffer set: scan_bufs)
      continue;
    }
    if (c2[1] > B.cap_out) {
      Cand* p = nullptr;
      VH_TRY(ws(ctx, WS_CAND, 3 * (size_t)c2[1] + 16, &p));
      continue;
    }
    std::vector<Cand> h((size_t)c2[1]);
    if (c2[1]) {
      VH_HIP(hipMemcpyAsync(h.data(), B.cand, sizeof(Cand) * (size_t)c2[1], hipMemcpyDeviceToHost, st));
      VH_HIP(hipStreamSynchronize(st));
    }
    sort_and_append(h, nx, ny, scale_index, sigma, minima, maxima);
    return VISFD_HIP_OK;
  }
  return fail(VISFD_HIP_EDEVICE, "blob candidate list kept overflowing");
}

}  // namespace vh
