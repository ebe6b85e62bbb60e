// tv_box.hip -- dense stick tensor voting in TOLERANCE MODE (context option tv_fma) for gfx950
// (reference lib/visfd/feature.hpp:1914-2037 and :2217-2384; surfaces with angular exponent 2 or 4).
//
// BASELINE.json's north_star asks for vote tensors within 1e-5 relative, not for the reference's bits.  Giving up the
// bits buys fused multiply-adds (a vote is 19 vector instructions, vote_fma below) and a free order of accumulation.
// Round 3's kernel (tv_pair) used that freedom for MIRROR-PAIRED SENDER PLANES -- the sender planes z + d and z + 1 - d
// (d = 1..h+1) see the receiver planes (z, z+1) through the same two table slices |jz| = d-1 and d, so one barrier
// interval serves both -- and kept the exact kernel's sweep: every wave tests every listed sender against its 64
// receivers (8 x 4 x 2) with a v_dot4 + v_cmp and votes under the execution mask of the lanes it reaches.  That sweep
// used 49 % of its lanes (ball of radius h against an 8 x 4 x 2 patch) at ~27 issue slots per vote step.
//
// This kernel keeps the skeleton (persistent workgroups claiming units from a global counter, per-workgroup rings of
// listed sender planes in global memory, packed lists, mirror-paired planes, two receiver pairs per pass) and
// replaces the sweep:
//
//   * SUB-PATCHES OF 32 RECEIVERS, TWO SENDER STREAMS PER WAVE.  A wave owns, per receiver pair, two sub-patches of
//     4 x 4 x 2 receivers (the left and right half of its 8 x 4 rows).  Lanes 0-31 and lanes 32-63 hold the SAME 32
//     receivers and take DIFFERENT senders: a vote step serves two senders, and the two partial sums of a receiver are
//     added when the pass stores.  A ball of radius 12 covers 67 % of the lanes of a 4 x 4 x 2 patch it touches (56 % of
//     an 8 x 4 x 2 one): a sixth fewer vote steps for the same votes.
//   * NO PER-LANE REACH TEST.  A sender is a HIT of a sub-patch if it reaches at least one of its receivers -- its
//     distance to the box of receivers is at most the slice's radius: a handful of float instructions that test 64
//     listed senders at once (one per lane) against both sub-patches.  The hits are compacted (ballot / mbcnt) into a
//     per-wave hit list in LDS, dealt alternately to the two streams.  A hit votes on ALL 32 lanes: the table slices in
//     LDS carry 3 zero rows above and below and >= 3 zero entries between rows, so that a receiver the sender does not
//     reach reads a zero weight (its vote adds 0) instead of being masked off.  The vote loop is branch-free:
//     1 address subtraction + 19 vote instructions per step, its LDS reads (hit entries, sender, table) requested
//     ahead of their use.
//   * 6 WAVES PER SIMD, 80 VGPRs, no spills: the 24 sums of a lane (2 pairs x 2 sub-patches x 6) and the read-ahead
//     registers of the vote loop stay in registers.  3 workgroups per CU, ~51 KB of LDS each at h = 12.
//
// Results differ from the reference's in the last bits (tests/test_tolerance_modes.py: within 1e-5 of the field's
// scale on every case the exact kernel is tested on, including crops of the 1024^3 bench volume).  One documented
// difference in kind: a NON-FINITE saliency spreads to the (up to 3 voxel wide) rim of zero-weight receivers around
// its ball (0 * inf), where the reference -- and the exact kernel, which is the default -- leave finite values.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace vh {

namespace {

constexpr int NT = 512;
constexpr int NW = NT / 64;
constexpr int TX = 8, TY = 4 * NW;     // a workgroup's tile of receivers: 8 x 32 (wave w: rows 4w..4w+3), NP pairs of planes
constexpr int NP = 2;                  // receiver pairs (z, z+1), (z+2, z+3) per pass: they need the same two slices at step d
constexpr int NLIST = 2 * NP;          // lists per interval: (A, B) of pair 0, (A, B) of pair 1
constexpr int NSUB = 2;                // sub-patches per wave and pair: x 0..3 and x 4..7
constexpr int NCH_MAX = 4;             // chunks of the region per wave the two-plane lister handles (h <= 12)
constexpr int LSLOTS = NT;             // LDS entry slots of an interval: one per thread
constexpr int HCAP = 36;               // hit entries per (wave, sub-patch, stream): 64 tests per chunk -> <= 32, + null + read-ahead
constexpr int YPAD = 3;                // zero rows above and below a table slice (a 4-row sub-patch overhangs by 3)
constexpr unsigned OOB = 0x7ffffff0u;  // byte offset beyond any plane descriptor: reads give 0

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, 0));
}

typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(3))) T* lds_ptr(unsigned a) {
  return (const __attribute__((address_space(3))) T*)(uintptr_t)a;
}

#ifdef VH_TV_STAMPS   // development build (tools/build_variant.py; cross-compiled by tests/test_abi.py): where a wave's time goes
__device__ unsigned long long g_box_stamps[8];
#define VH_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += (unsigned)(t_ - st_last); st_last = t_; } while (0)
#else
#define VH_STAMP(i) do {} while (0)
#endif
#ifdef VH_TV_COUNT    // development build: list entries tested, hits, vote steps
__device__ unsigned long long g_box_counts[8];
#endif

struct BoxParams {
  int nx, ny, nz;
  int z_out0, z_out1;    // receiver planes [z_out0, z_out1)
  int h;
  int rw, rh;            // region width = TX + 2h, height = TY + 2h
  int rw_magic;          // q / rw == (q * rw_magic) >> 20 for every region position q (checked by the launcher)
  int nchunk;            // 64-voxel chunks of the region per wave
  int tiles_x, tiles_y;
  int zrun;              // receiver planes per unit of work
  int relist;            // 1: list every sender plane again for every pass (option tv_no_replay; tests)
  int sp;                // row stride of a table slice in float4 entries (tv_box_row)
  int nsl;               // float4 entries of a table slice (tv_box_slice)
};

__device__ __forceinline__ void fmacc(float& t, float a, float b) {
  asm("v_fmac_f32 %0, %1, %2" : "+v"(t) : "v"(a), "v"(b));
}

// One vote.  The table holds R = sqrt(2) rhat (tv.hip: tv_table_device), so that t = R.n = sqrt(2) u, t R - n = 2 u rhat - n = m
// and 2 - t^2 = 2 (1 - u^2); the factor 1/2 (exponent 2) or 1/4 (exponent 4) the decay then lacks is applied to the sender's
// saliency when it is listed (an exact scaling), as is the sender's mask value.  ZNEG: the slice in LDS is the one of -jz:
// rhat_z has the opposite sign.
template <int MODE, bool ZNEG>
__device__ __forceinline__ void vote_fma(float (&T)[6], const f4v& snd /* sal, n */, const f4v& tw /* w, R */) {
  const float Rz = ZNEG ? -tw.w : tw.w;    // (a source modifier of the instructions below)
  const float t = __builtin_fmaf(Rz, snd.w, __builtin_fmaf(tw.z, snd.z, tw.y * snd.y));
  const float q = __builtin_fmaf(-t, t, 2.0f);
  const float m0 = __builtin_fmaf(t, tw.y, -snd.y);
  const float m1 = __builtin_fmaf(t, tw.z, -snd.z);
  const float m2 = __builtin_fmaf(t, Rz, -snd.w);
  const float sw = snd.x * tw.x;
  const float bse = (MODE == 0) ? (sw * q) * q : sw * q;
  const float b0 = bse * m0, b1 = bse * m1, b2 = bse * m2;
  fmacc(T[0], b0, m0);
  fmacc(T[3], b0, m1);
  fmacc(T[5], b0, m2);
  fmacc(T[1], b1, m1);
  fmacc(T[4], b1, m2);
  fmacc(T[2], b2, m2);
}

// (volatile: the reads of the vote loop stay in program order -- requested a step ahead of their use -- and are neither
// paired into ds_read2 forms nor sunk behind the loop's exits)
__device__ __forceinline__ f4v lds_f4(unsigned a) { return *(const volatile __attribute__((address_space(3))) f4v*)(uintptr_t)a; }
__device__ __forceinline__ void lds_store_u2(unsigned a, unsigned x, unsigned y) {
  u2v v = {x, y};
  *(__attribute__((address_space(3))) u2v*)(uintptr_t)a = v;
}
__device__ __forceinline__ uint2 lds_u2(unsigned a) {
  const u2v v = *(const volatile __attribute__((address_space(3))) u2v*)(uintptr_t)a;
  return make_uint2(v.x, v.y);
}

// The votes of one hit list: nst steps, each serving one hit per stream.  hp: LDS address of this lane's stream's entries
// {LDS address of the sender's {saliency, normal}, byte offset E of the sender in a table slice}; r16: LDS address of this
// lane's table entry for a sender at E = 0.  Software pipeline over two register sets, four steps per trip: two entries
// come with one ds_read_b128, requested two to four steps ahead; a step's sender and table reads are requested one step
// ahead of its vote.  Entries behind a stream's last one are stale or null, never invalid addresses: the reads run ahead
// of the votes.
__device__ __forceinline__ u4v lds_u4(unsigned a) { return *(const volatile __attribute__((address_space(3))) u4v*)(uintptr_t)a; }

template <int MODE, bool ZNEG, int OFF>
__device__ __forceinline__ void vote_hits(float (&T)[6], unsigned hp, int nst, unsigned r16) {
  u4v h0 = lds_u4(hp + (unsigned)OFF), h1 = lds_u4(hp + (unsigned)(OFF + 16));
  f4v sa = lds_f4(h0.x), ta = lds_f4(r16 - h0.y);
  f4v sb, tb;
  int k = 0;
  for (;;) {   // uniform
    sb = lds_f4(h0.z);
    tb = lds_f4(r16 - h0.w);
    vote_fma<MODE, ZNEG>(T, sa, ta);
    if (++k >= nst) break;
    sa = lds_f4(h1.x);
    ta = lds_f4(r16 - h1.y);
    h0 = lds_u4(hp + (unsigned)(OFF + 32));
    vote_fma<MODE, ZNEG>(T, sb, tb);
    if (++k >= nst) break;
    sb = lds_f4(h1.z);
    tb = lds_f4(r16 - h1.w);
    vote_fma<MODE, ZNEG>(T, sa, ta);
    if (++k >= nst) break;
    sa = lds_f4(h0.x);
    ta = lds_f4(r16 - h0.y);
    h1 = lds_u4(hp + (unsigned)(OFF + 48));
    hp += 32u;
    vote_fma<MODE, ZNEG>(T, sb, tb);
    if (++k >= nst) break;
  }
}

template <bool MASKED_SRC, int MODE>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(6, 6)))
tv_box_kernel(const float* __restrict__ sal, const float* __restrict__ dir, float* __restrict__ ten,
              const float* __restrict__ mask_src, const float* __restrict__ mask_dst,
              const float4* __restrict__ table /* [2h+1] slices of nsl entries: w, sqrt(2) rhat at j, zero padding */,
              BoxParams p, unsigned* __restrict__ tile_counter, unsigned ntiles,
              unsigned char* __restrict__ scratch /* per-workgroup rings of compacted sender planes */) {
  // l_ent[e]  float4 {saliency (scaled), n0, n1, n2} of the interval's e-th entry; l_ent[LSLOTS]: the null sender (zeros)
  // l_pos[e]  {region position bytes (ex, ey), byte offset of the sender in a table slice: 16 (ey SP + ex)}
  // l_hit     per wave and sub-patch, the hits of the current chunk of 64 tested entries, dealt alternately to two streams
  __shared__ __attribute__((aligned(16))) float4 l_ent[LSLOTS + 1];
  __shared__ __attribute__((aligned(16))) uint2 l_pos[LSLOTS];
  __shared__ __attribute__((aligned(16))) uint2 l_hit[NW][NSUB][2][HCAP];
  __shared__ int wave_tot[2][2][NW];
  __shared__ unsigned claimed_tile;
  __shared__ int plane_cnt[88];              // entries per ring slot, [2h + 2 NP] (h <= 40)
  __shared__ int rho_tab[44];                // floor(sqrt(h^2 - j^2)), j = 0..h: the radius of slice j
  extern __shared__ __attribute__((aligned(16))) unsigned char slices[];   // two table slices: S_j (jz = +j) in slot j & 1

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = p.h;
  const int S = 2 * h + 1;
  // LDS rows of a table slice are SP float4 apart, SP = 4 mod 8 and >= S + 3 (tv_box_row): row offsets of 64 or 192 bytes
  // modulo the 256 bytes of the 64 banks.  ds_read_b128 serves a half wave as two groups of 16 lanes, {0-3, 12-15, 20-27}
  // and the rest (MI355X_MICROARCH.md); the lanes of a half wave are dealt to their 4 x 4 x 2 receivers so that each group
  // is the four rows of ONE receiver plane: four 64-byte segments on different banks, whatever the sender's offset.
  const int SP = p.sp;
  const int nsl = p.nsl;
  const int R = p.rw * p.rh;
  const i64 plane = (i64)p.nx * p.ny;
  const i64 nvox = plane * p.nz;
  const int plane_bytes = (int)(plane * 4);
  constexpr int ENT_BYTES = 20;
  const size_t plane_stride = (size_t)R * ENT_BYTES;   // a ring slot: float4 ent[R]; unsigned pos[R]
  const int P = S + 2 * NP - 1;   // sender planes the receiver planes of a pass reach = slots of the ring
  unsigned char* const ring = scratch + (size_t)blockIdx.x * plane_stride * P;
  int npar = 0;
  int slot_has[2] = {-1, -1};                // which slice S_j each LDS slot holds (uniform)
  bool up = false;                           // direction of d for the next pass (flips after every pass)
  float4* const sl4 = reinterpret_cast<float4*>(slices);
  const unsigned ent_base = lds_addr(l_ent);
  const unsigned null_ent = ent_base + 16u * (unsigned)LSLOTS;
#ifdef VH_TV_STAMPS
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
#ifdef VH_TV_COUNT
  unsigned cnt_tested = 0, cnt_hits = 0, cnt_steps = 0;
#endif

  // lane -> receiver of a sub-patch: stream = lane >> 5; quads of lanes 0, 3, 5, 6 (lanes 0-3, 12-15, 20-27: one group of
  // ds_read_b128) are rows 0-3 of the lower receiver plane, quads 1, 2, 4, 7 rows 0-3 of the upper one
  const int strm = lane >> 5;
  const int qd = (lane & 31) >> 2;
  const int lcol = lane & 3, lrow = qd >> 1, lpl = (0x96 >> qd) & 1;

  // every stale hit entry must be a valid pair of LDS addresses: the vote loop reads ahead of its hits
  if (tid == 0) l_ent[LSLOTS] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (tid <= h) {
    const int r2 = h * h - tid * tid;
    int rho = (int)__builtin_sqrtf((float)r2);
    while (rho * rho > r2) rho--;
    while ((rho + 1) * (rho + 1) <= r2) rho++;
    rho_tab[tid] = rho;
  }
  for (int i = tid; i < NW * NSUB * 2 * HCAP; i += NT) (&l_hit[0][0][0][0])[i] = make_uint2(null_ent, 0u);

  for (;;) {
    if (tid == 0) claimed_tile = atomicAdd(tile_counter, 1u);
    __syncthreads();
    unsigned b = claimed_tile;
    __syncthreads();
    if (b >= ntiles) break;
    const int tile_x = b % p.tiles_x;
    b /= p.tiles_x;
    const int tile_y = b % p.tiles_y;
    const int z_run0 = p.z_out0 + (int)(b / p.tiles_y) * p.zrun;
    const int z_run1 = min(z_run0 + p.zrun, p.z_out1);
    const int x0 = tile_x * TX, y0 = tile_y * TY;

    // MIRRORED ROW BLOCKS.  A wave's receivers of pair pp are the four rows of block w for the first pair, NW - 1 - w for
    // the second: the sweep of an interval lasts as long as its slowest wave, and which rows are heavy -- those near a
    // membrane -- is much the same for the two pairs of a pass.
    auto row_block = [&](int pp) -> int { return (pp & 1) ? NW - 1 - wave : wave; };
    // The per-lane constants of a phase are RECOMPUTED from the lane number where the phase starts (the empty asm hides the
    // number's origin from the compiler): hoisted out of the step loop they stay live across the vote loops and are spilled.
    auto fresh_lane = [&]() -> unsigned {
      unsigned ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(ln));
      return ln;
    };

    // ---- LISTING, two planes at a time (window regions of <= 4 chunks per wave: h <= 12).  Every load of a phase is in
    // flight at once -- the saliencies of both planes (kept in registers across the barrier: one read per voxel), then the
    // normals of a plane's salient voxels -- and both planes share one barrier.  A plane index < 0 means "no plane".
    // Entries are written in DESCENDING region position (row order, which the row-range culling needs).
    auto list_two = [&](int sz0, int sz1) {
      constexpr int NCH = NCH_MAX;
      const int q0 = wave * p.nchunk * 64 + lane;
      unsigned off[NCH];
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int q = q0 + 64 * j;
        const int ey = (int)(((unsigned)q * (unsigned)p.rw_magic) >> 20);
        const int ex = q - ey * p.rw;
        const int sx = x0 - h + ex, sy = y0 - h + ey;
        const bool ok = j < p.nchunk && q < R && sx >= 0 && sx < p.nx && sy >= 0 && sy < p.ny;
        off[j] = ok ? (unsigned)(sy * p.nx + sx) * 4u : OOB;
      }
      float sv[2][NCH];
      int cnt[2] = {0, 0};
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int sz = k ? sz1 : sz0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(sal + (i64)(sz < 0 ? 0 : sz) * plane), 0,
                                                                            sz < 0 ? 0 : plane_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < NCH; j++) sv[k][j] = buf_load(rs, off[j]);
        if (MASKED_SRC) {
          const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)(mask_src + (i64)(sz < 0 ? 0 : sz) * plane), 0,
                                                                              sz < 0 ? 0 : plane_bytes, 0x00020000);
#pragma unroll
          for (int j = 0; j < NCH; j++)
            if (buf_load(rm, off[j]) == 0.0f) sv[k][j] = 0.0f;
        }
      }
#pragma unroll
      for (int k = 0; k < 2; k++)
#pragma unroll
        for (int j = 0; j < NCH; j++) cnt[k] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(sv[k][j] != 0.0f));
      const int par = (npar++) & 1;
      if (lane == 0) { wave_tot[par][0][wave] = cnt[0]; wave_tot[par][1][wave] = cnt[1]; }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int sz = k ? sz1 : sz0;
        if (sz < 0) continue;   // uniform
        int running = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
          const int t = wave_tot[par][k][w];
          running += (w > wave) ? t : 0;
          total += t;
        }
        running = __builtin_amdgcn_readfirstlane(running);
        const int slot = sz % P;
        unsigned char* const ring_plane = ring + (size_t)slot * plane_stride;
        if (tid == 0) plane_cnt[slot] = total;
        if (cnt[k] == 0) continue;   // uniform
        const __amdgpu_buffer_rsrc_t rd0 =
            __builtin_amdgcn_make_buffer_rsrc((void*)(dir + (i64)sz * plane), 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd1 =
            __builtin_amdgcn_make_buffer_rsrc((void*)(dir + nvox + (i64)sz * plane), 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd2 =
            __builtin_amdgcn_make_buffer_rsrc((void*)(dir + 2 * nvox + (i64)sz * plane), 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((MASKED_SRC ? mask_src : sal) + (i64)sz * plane), 0, plane_bytes, 0x00020000);
        float n0[NCH], n1[NCH], n2[NCH], mvv[NCH];
#pragma unroll
        for (int j = 0; j < NCH; j++) {   // the normals of the salient voxels only, all chunks requested before the first use
          n0[j] = n1[j] = n2[j] = 0.0f;
          mvv[j] = 1.0f;
          if (sv[k][j] != 0.0f) {
            n0[j] = buf_load(rd0, off[j]);
            n1[j] = buf_load(rd1, off[j]);
            n2[j] = buf_load(rd2, off[j]);
            if (MASKED_SRC) mvv[j] = buf_load(rm, off[j]);
          }
        }
#pragma unroll
        for (int j = NCH - 1; j >= 0; j--) {
          const bool f = sv[k][j] != 0.0f;
          const unsigned long long bal = __builtin_amdgcn_ballot_w64(f);
          const int tb = __builtin_popcountll(bal);
          const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
          if (f) {
            const int idx = running + (tb - below - 1);
            const int q = q0 + 64 * j;
            const int ey = (int)(((unsigned)q * (unsigned)p.rw_magic) >> 20);
            const int ex = q - ey * p.rw;
            float s = sv[k][j] * (MODE == 0 ? 0.25f : 0.5f);
            if (MASKED_SRC) s = s * mvv[j];
            reinterpret_cast<float4*>(ring_plane)[idx] = make_float4(s, n0[j], n1[j], n2[j]);
            reinterpret_cast<unsigned*>(ring_plane + (size_t)R * 16)[idx] = (unsigned)ex | ((unsigned)ey << 8);
          }
          running += tb;
        }
      }
    };

    // ---- LISTING, one plane, any window: sender plane sz of this tile's region into its ring slot ------------------------
    auto list_plane = [&](int sz) {
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(sal + (i64)sz * plane), 0, plane_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((MASKED_SRC ? mask_src : sal) + (i64)sz * plane), 0, plane_bytes, 0x00020000);
      const int q0 = wave * p.nchunk * 64 + lane;
      auto voff_of = [&](int q, int& ex, int& ey) -> unsigned {
        ey = (int)(((unsigned)q * (unsigned)p.rw_magic) >> 20);
        ex = q - ey * p.rw;
        const int sx = x0 - h + ex, sy = y0 - h + ey;
        const bool ok = q < R && sx >= 0 && sx < p.nx && sy >= 0 && sy < p.ny;
        return ok ? (unsigned)(sy * p.nx + sx) * 4u : OOB;
      };
      auto salient = [&](unsigned off) -> float {
        float s = buf_load(rs, off);
        if (MASKED_SRC) {
          if (buf_load(rm, off) == 0.0f) s = 0.0f;
        }
        return s;
      };
      int cnt = 0;
#pragma unroll 1
      for (int j = 0; j < p.nchunk; j++) {
        int ex, ey;
        const float s = salient(voff_of(q0 + 64 * j, ex, ey));
        cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(s != 0.0f));
      }
      const int par = (npar++) & 1;
      if (lane == 0) wave_tot[par][0][wave] = cnt;
      __syncthreads();
      int running = 0, total = 0;
#pragma unroll
      for (int w = 0; w < NW; w++) {
        const int t = wave_tot[par][0][w];
        running += (w > wave) ? t : 0;
        total += t;
      }
      running = __builtin_amdgcn_readfirstlane(running);
      const int slot = sz % P;
      unsigned char* const ring_plane = ring + (size_t)slot * plane_stride;
      if (cnt > 0) {
        const __amdgpu_buffer_rsrc_t rd0 =
            __builtin_amdgcn_make_buffer_rsrc((void*)(dir + (i64)sz * plane), 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd1 =
            __builtin_amdgcn_make_buffer_rsrc((void*)(dir + nvox + (i64)sz * plane), 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd2 =
            __builtin_amdgcn_make_buffer_rsrc((void*)(dir + 2 * nvox + (i64)sz * plane), 0, plane_bytes, 0x00020000);
#pragma unroll 1
        for (int j = p.nchunk - 1; j >= 0; j--) {
          int ex, ey;
          const unsigned off = voff_of(q0 + 64 * j, ex, ey);
          const float s = salient(off);
          const bool f = s != 0.0f;
          const unsigned long long bal = __builtin_amdgcn_ballot_w64(f);
          if (bal == 0ull) continue;   // uniform
          const int tb = __builtin_popcountll(bal);
          const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
          if (f) {
            const int idx = running + (tb - below - 1);
            float sc = s * (MODE == 0 ? 0.25f : 0.5f);
            if (MASKED_SRC) sc = sc * buf_load(rm, off);
            reinterpret_cast<float4*>(ring_plane)[idx] = make_float4(sc, buf_load(rd0, off), buf_load(rd1, off), buf_load(rd2, off));
            reinterpret_cast<unsigned*>(ring_plane + (size_t)R * 16)[idx] = (unsigned)ex | ((unsigned)ey << 8);
          }
          running += tb;
        }
      }
      if (tid == 0) plane_cnt[slot] = total;
    };

    float TT[NP][NSUB][6];

    // ---- TEST + VOTE: entries [i0, i1) of one list's share of the interval (first LDS slot `base`), 64 at a time.  Lane l
    // tests entry i0 + 64 c + l against the boxes of both sub-patches: with the sender at region position (ex, ey) and a
    // sub-patch's receivers at x in [bx, bx + 3], y in [by, by + 3], the nearest receiver is max(|ex - (bx + 1.5)| - 1.5, 0)
    // columns and as many rows (with by) away; it is reached if dx^2 + dy^2 <= rr = h^2 - (d-1)^2 (the nearer of the two
    // receiver planes is d - 1 planes from the sender plane).  All small integers: exact in float.
    auto test_vote = [&](auto ZN, auto PP, int base, int i0, int i1, unsigned r16, float cy, float rr, unsigned null_e16) {
      constexpr bool ZNEG = decltype(ZN)::value;
      constexpr int pp = decltype(PP)::value;
      const float cx0 = (float)h + 1.5f;
      // this wave's hit entries: sub-patch s, stream t at hb + (2 s + t) * HCAP * 8 (the constants are offset fields)
      const unsigned hb = lds_addr(&l_hit[wave][0][0][0]);
      for (int c = i0; c < i1; c += 64) {   // uniform
        const int ln = (int)fresh_lane();
        const int e = c + ln;
        uint2 pw = make_uint2(0xffffffffu, 0u);           // lanes without an entry: far from every box
        if (e < i1) pw = l_pos[base + e];
        const float exf = (float)(pw.x & 0xffu), eyf = (float)((pw.x >> 8) & 0xffu);
        const float dy = fmaxf(__builtin_fabsf(eyf - cy) - 1.5f, 0.0f);
        const float dy2 = dy * dy;
        const unsigned ent = ent_base + 16u * (unsigned)(base + e);
        int nh[NSUB];
#pragma unroll
        for (int s = 0; s < NSUB; s++) {
          const float dx = fmaxf(__builtin_fabsf(exf - (cx0 + 4.0f * (float)s)) - 1.5f, 0.0f);
          const bool hit = __builtin_fmaf(dx, dx, dy2) <= rr;
          const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
          nh[s] = __builtin_popcountll(bal);
          const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
          // (sub-patch 1's table entries are 4 columns = 64 bytes further: taken off the sender's offset here, so that the
          // vote loop has one table base for both sub-patches)
          if (hit) lds_store_u2(hb + (unsigned)(2 * s * HCAP * 8) + (rank & 1u) * (unsigned)(HCAP * 8) + (rank >> 1) * 8u, ent, pw.y - 64u * (unsigned)s);
          // the second stream's last step when the count is odd: the null sender (zero saliency, zero normal), placed on the
          // sub-patch's first receiver so that every lane reads a table entry of the slice (finite; times 0)
          if (ln == 0) lds_store_u2(hb + (unsigned)((2 * s + 1) * HCAP * 8) + (unsigned)(nh[s] >> 1) * 8u, null_ent, null_e16);
        }
#ifdef VH_TV_COUNT
        cnt_tested += (unsigned)min(64, i1 - c);
        cnt_hits += (unsigned)(nh[0] + nh[1]);
        cnt_steps += (unsigned)(((nh[0] + 1) >> 1) + ((nh[1] + 1) >> 1));
#endif
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        const unsigned hp = hb + (fresh_lane() >> 5) * (unsigned)(HCAP * 8);
        __builtin_amdgcn_s_setprio(1);   // a voting wave is on its workgroup's critical path; waves that list or fill are not (337 -> 333 ms)
        if (nh[0] > 0) vote_hits<MODE, ZNEG, 0>(TT[pp][0], hp, (nh[0] + 1) >> 1, r16);              // (uniform)
        if (nh[1] > 0) vote_hits<MODE, ZNEG, 2 * HCAP * 8>(TT[pp][1], hp, (nh[1] + 1) >> 1, r16);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
    };

    int cached_lo = 1, cached_hi = 0;
    for (int rz = z_run0; rz < z_run1; rz += 2 * NP) {
      // sender planes that reach the LIVE receivers of this pass (a run may end inside a pass: nothing above the last
      // live receiver + h is needed -- or, in a slab run, complete -- then)
      const int sz_hi = min(min(rz + 2 * NP - 1, z_run1 - 1) + h, p.nz - 1), sz_lo = max(rz - h, 0);
      if (p.nchunk <= NCH_MAX) {
        int pend = -1;
        for (int sz = sz_hi; sz >= sz_lo; sz--) {   // uniform
          if (!(p.relist || sz < cached_lo || sz > cached_hi)) continue;
          if (pend < 0) { pend = sz; continue; }
          list_two(pend, sz);
          pend = -1;
        }
        if (pend >= 0) list_two(pend, -1);
      } else {
        for (int sz = sz_hi; sz >= sz_lo; sz--)   // uniform
          if (p.relist || sz < cached_lo || sz > cached_hi) list_plane(sz);
      }
      cached_lo = sz_lo;
      cached_hi = sz_hi;

#pragma unroll
      for (int pp = 0; pp < NP; pp++)
#pragma unroll
        for (int s = 0; s < NSUB; s++)
#pragma unroll
          for (int k = 0; k < 6; k++) TT[pp][s][k] = 0.0f;
      __syncthreads();   // ring entries and counts of this pass are visible
      VH_STAMP(0);

      // d = 1 .. h+1.  Pair pp (receiver planes z = rz + 2 pp and z + 1): sender planes A = z + d (above: jz = -d for the
      // lower receiver plane, 1-d for the upper one) and B = z + 1 - d (below: jz = d-1 and d).  All of them need the slices
      // S_(d-1) and S_d; the direction of d alternates from pass to pass, so that every step -- the first of a pass
      // included -- finds one of its two slices in LDS already.
      const int rzs = rz % P;            // ring slot of plane rz; the planes of a pass are within (-P, 2P) of it
      auto ring_slot = [&](int sz) -> int {
        int sl = rzs + (sz - rz);
        sl = sl < 0 ? sl + P : sl;
        return sl >= P ? sl - P : sl;
      };
      for (int step = 0; step <= h; step++) {
        const int d = up ? step + 1 : h + 1 - step;
        int lsz[NLIST], lcnt[NLIST];     // list 2 pp: plane A of pair pp; list 2 pp + 1: its plane B
        int cmax = 0;
#pragma unroll
        for (int pp = 0; pp < NP; pp++) {
          const int z = rz + 2 * pp;
          const bool pair_live = z < z_run1;                   // (uniform) a pair beyond the end of the run takes no votes
          lsz[2 * pp] = z + d;
          lsz[2 * pp + 1] = z + 1 - d;
          lcnt[2 * pp] = (pair_live && lsz[2 * pp] <= sz_hi) ? __builtin_amdgcn_readfirstlane(plane_cnt[ring_slot(lsz[2 * pp])]) : 0;
          lcnt[2 * pp + 1] = (pair_live && lsz[2 * pp + 1] >= sz_lo) ? __builtin_amdgcn_readfirstlane(plane_cnt[ring_slot(lsz[2 * pp + 1])]) : 0;
          cmax = max(cmax, max(lcnt[2 * pp], lcnt[2 * pp + 1]));
        }
        if (cmax == 0) continue;   // uniform
        // slices S_(d-1) and S_d (S_(h+1), which the receiver plane h+1 planes from the sender plane reads at the step
        // d = h+1, is a slice of zeros); lists and slices are free: every interval ends with a barrier
        int need[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const int j = d - 1 + k;
          need[k] = (slot_has[j & 1] != j) ? j : -1;
          if (need[k] >= 0) slot_has[j & 1] = j;
        }
        // rows a wave can reach: the nearer of its two receiver planes is |jz| = d-1 away from either sender plane
        const int rho = __builtin_amdgcn_readfirstlane(rho_tab[d - 1]);
        const float rr = (float)(h * h - (d - 1) * (d - 1));
        int pre[NLIST + 1];                                    // (uniform) first position of list k in the step's sequence
        pre[0] = 0;
#pragma unroll
        for (int k = 0; k < NLIST; k++) pre[k + 1] = pre[k] + lcnt[k];
        const int total = pre[NLIST];
        int pl[NLIST];                                         // (uniform) ring slot of list k's plane
#pragma unroll
        for (int k = 0; k < NLIST; k++) pl[k] = ring_slot(lsz[k]);
        for (int done = 0; done < total; done += NT) {   // uniform
          // PACKED LISTS: the lists of a step are dealt to the threads as ONE sequence; list k's share of this interval:
          // sequence positions = LDS slots [c[k], c[k] + len[k])
          int c[NLIST], len[NLIST];
#pragma unroll
          for (int k = 0; k < NLIST; k++) {
            const int lo = min(max(pre[k], done), done + NT), hi = min(pre[k + 1], done + NT);
            c[k] = lo - done;
            len[k] = max(hi - lo, 0);
          }
          const int g = done + tid;
          int k_me = 0;
#pragma unroll
          for (int k = 1; k < NLIST; k++) k_me += (g >= pre[k]) ? 1 : 0;
          const bool have = g < total;
          int idx = g, pl_me = pl[0];
#pragma unroll
          for (int k = 1; k < NLIST; k++)
            if (k_me == k) { idx = g - pre[k]; pl_me = pl[k]; }
          const unsigned char* ring_plane = ring + (size_t)pl_me * plane_stride;
          float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          unsigned m = 0u;
          if (have) {
            a = reinterpret_cast<const float4*>(ring_plane)[idx];
            m = reinterpret_cast<const unsigned*>(ring_plane + (size_t)R * 16)[idx];
          }
          if (done == 0) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
              if (need[k] < 0) continue;   // uniform
              const int j = need[k];
              const float4* src4 = table + (i64)(j + h) * nsl;
              float4* dst4 = sl4 + (j & 1) * nsl;
              if (j <= h) for (int i = tid; i < nsl; i += NT) dst4[i] = src4[i];
              else for (int i = tid; i < nsl; i += NT) dst4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
          }
          if (have) {
            l_ent[tid] = a;
            l_pos[tid] = make_uint2(m, 16u * (((m >> 8) & 0xffu) * (unsigned)SP + (m & 0xffu)));
          }
          VH_STAMP(1);
          __syncthreads();   // lists (and slices) complete
          VH_STAMP(2);
          // entries are in descending row order: of list k, this wave needs those from the first one at or below region row
          // 4 wv + h + 3 + rho to the last one at or above row 4 wv + h - rho.  Every wave counts both kinds itself, from the
          // row bytes of the position words in LDS, 64 entries at a time.
          int i0[NLIST], i1[NLIST];
#pragma unroll
          for (int k = 0; k < NLIST; k++) {
            const int wv = row_block(k >> 1);
            const int hi_row = 4 * wv + h + 3 + rho, lo_row = 4 * wv + h - rho;
            int above = 0, upto = 0;
            for (int j = 0; j < len[k]; j += 64) {   // uniform
              int ey = -1;
              if (j + lane < len[k]) ey = (int)((l_pos[c[k] + j + lane].x >> 8) & 0xffu);
              above += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ey > hi_row));
              upto += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ey >= lo_row));
            }
            i0[k] = above;
            i1[k] = upto;
          }
          auto pair_votes = [&](auto PP) {
            constexpr int pp = decltype(PP)::value;
            if (i1[2 * pp] <= i0[2 * pp] && i1[2 * pp + 1] <= i0[2 * pp + 1]) return;   // uniform
            const int rb = row_block(pp);
            const float cy = (float)(4 * rb + h) + 1.5f;
            const unsigned ln = fresh_lane();
            const int fq = (int)((ln & 31u) >> 2);
            const int fcol = (int)(ln & 3u), frow = fq >> 1, fpl = (0x96 >> fq) & 1;
            // this lane's table entry of a sender at region position (0, 0), sub-patch 0, in slice slot 0:
            // 4 guard entries, then row (jy + h + YPAD), column (jx + h) with jy = 4 rb + frow + h - ey, jx = fcol + h - ex
            const unsigned r16_0 = lds_addr(slices) + 16u * (unsigned)(4 + (4 * rb + frow + 2 * h + YPAD) * SP + fcol + 2 * h);
            const unsigned null_e16 = 16u * (unsigned)((4 * rb + h) * SP + h);
            // plane A (above): the lower receiver plane sees it at jz = -d (slice S_d, rhat_z negated), the upper one at 1-d
            if (i1[2 * pp] > i0[2 * pp]) {
              const int js = fpl ? d - 1 : d;
              test_vote(std::true_type{}, PP, c[2 * pp], i0[2 * pp], i1[2 * pp], r16_0 + (unsigned)(16 * nsl) * (unsigned)(js & 1), cy, rr, null_e16);
            }
            // plane B (below): jz = d-1 for the lower plane (S_(d-1)), d for the upper one (S_d)
            if (i1[2 * pp + 1] > i0[2 * pp + 1]) {
              const int js = fpl ? d : d - 1;
              test_vote(std::false_type{}, PP, c[2 * pp + 1], i0[2 * pp + 1], i1[2 * pp + 1], r16_0 + (unsigned)(16 * nsl) * (unsigned)(js & 1), cy, rr, null_e16);
            }
          };
          pair_votes(std::integral_constant<int, 0>{});
          pair_votes(std::integral_constant<int, 1>{});
          VH_STAMP(3);
          __syncthreads();   // everyone done reading before the lists or the slices are refilled
          VH_STAMP(4);
        }
      }
      up = !up;

      // ---- the pass's sums: a receiver's two streams are added; lanes 0-31 store sub-patch 0, lanes 32-63 sub-patch 1 ----
#pragma unroll
      for (int pp = 0; pp < NP; pp++) {
        const int rb = row_block(pp);
        const int rx = x0 + 4 * strm + lcol, ry = y0 + 4 * rb + lrow, rzl = rz + 2 * pp + lpl;
        const bool in = rx < p.nx && ry < p.ny && rzl < z_run1;
        const i64 rc = (i64)rzl * plane + (i64)ry * p.nx + rx;
        const bool live = in && !(mask_dst && mask_dst[in ? rc : 0] == 0.0f);
        float v[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const float a = TT[pp][0][k] + __shfl_xor(TT[pp][0][k], 32);
          const float bq = TT[pp][1][k] + __shfl_xor(TT[pp][1][k], 32);
          v[k] = strm ? bq : a;
        }
        if (live) {
#pragma unroll
          for (int k = 0; k < 6; k++) __builtin_nontemporal_store(v[k], &ten[k * nvox + rc]);   // written once, not read here
        }
      }
    }   // next pass of the run
  }   // next unit
#ifdef VH_TV_STAMPS
  VH_STAMP(5);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; i++) atomicAdd(&g_box_stamps[i], st_acc[i]);
  }
#endif
#ifdef VH_TV_COUNT
  if (lane == 0) {
    atomicAdd(&g_box_counts[0], (unsigned long long)cnt_tested);
    atomicAdd(&g_box_counts[1], (unsigned long long)cnt_hits);
    atomicAdd(&g_box_counts[2], (unsigned long long)cnt_steps);
  }
#endif
}

// Test aid (context option tv_poison): fills every CU's LDS with NaN bit patterns before the voting kernel runs, so that a
// vote that uses LDS (or ring memory, or an output voxel) the kernel has not written shows up as NaN on every box -- not
// only on one whose previous tenant happened to leave such bits behind.
__global__ void __launch_bounds__(256) lds_poison_kernel(unsigned* sink) {
  extern __shared__ unsigned pz[];
  for (int i = threadIdx.x; i < 160 * 256; i += 256) pz[i] = 0xffffffffu;
  __syncthreads();
  if (pz[(threadIdx.x * 37) % (160 * 256)] == 1u) sink[0] = 1u;
}

}  // namespace

// Tolerance-mode tensor voting (surfaces, exponent 2 or 4).  dtab_box: the {w, sqrt(2) rhat} table on the device in this
// kernel's slice layout (tv.hip: tv_table_device).
int dev_tv_box(visfd_hip_ctx* ctx, const float* sal, const float* dir, float* ten, const float* mask_src,
               const float* mask_dst, i64 nx, i64 ny, i64 nz, i64 z_out0, i64 z_out1, int h, const float4* dtab_box,
               int exponent, bool* handled) {
  *handled = false;
  if (exponent != 2 && exponent != 4) return VISFD_HIP_OK;
  if (h < 1 || h > 40) return VISFD_HIP_OK;
  if (nx * ny >= (1LL << 29)) return VISFD_HIP_OK;
  const int n = 2 * h + 1;
  hipStream_t st = ctx->stream;
  BoxParams p;
  p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
  p.z_out0 = (int)z_out0; p.z_out1 = (int)z_out1;
  p.h = h;
  p.rw = TX + 2 * h;
  p.rh = TY + 2 * h;
  const int R = p.rw * p.rh;
  p.nchunk = (R + NT - 1) / NT;
  p.rw_magic = ((1 << 20) + p.rw - 1) / p.rw;
  for (int q = 0; q < p.nchunk * NT; q++)
    if ((int)(((unsigned)q * (unsigned)p.rw_magic) >> 20) != q / p.rw) return fail(VISFD_HIP_EINVAL, "tv_box: region index division");
  if (p.rw > 255 || p.rh > 255) return VISFD_HIP_OK;   // region positions travel as bytes
  p.sp = tv_box_row(h);
  p.nsl = tv_box_slice(h);
  const size_t slice_bytes = sizeof(float4) * (size_t)p.nsl;
  p.tiles_x = (int)((nx + TX - 1) / TX);
  p.tiles_y = (int)((ny + TY - 1) / TY);
  p.relist = ctx->opt.tv_no_replay ? 1 : 0;
  p.zrun = 32;
  if (ctx->opt.tv_zrun >= 1 && ctx->opt.tv_zrun <= 4096) p.zrun = ctx->opt.tv_zrun;
  if ((i64)p.zrun > z_out1 - z_out0) p.zrun = (int)(z_out1 - z_out0);
  if (p.zrun < 1) p.zrun = 1;
  const i64 nruns = (z_out1 - z_out0 + p.zrun - 1) / p.zrun;
  const i64 nblk = (i64)p.tiles_x * p.tiles_y * nruns;
  if (nblk > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  if (nblk <= 0) { *handled = true; return VISFD_HIP_OK; }
  const size_t lds = 2 * slice_bytes;
  const size_t lds_static = sizeof(float4) * (LSLOTS + 1) + sizeof(uint2) * LSLOTS + sizeof(uint2) * NW * NSUB * 2 * HCAP + 1024;
  if (lds + lds_static > 150 * 1024) return VISFD_HIP_OK;   // window too wide: the caller falls back
  unsigned* counter = nullptr;
  VH_TRY(ws(ctx, WS_COUNTER, 16, &counter));
  VH_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned), st));
  size_t wg_per_cu = (160 * 1024) / (lds + lds_static);
  if (wg_per_cu > 3) wg_per_cu = 3;   // 6 waves per SIMD (80 VGPRs)
  if (wg_per_cu < 1) wg_per_cu = 1;
  i64 ngrid = (i64)ctx->num_cus * (i64)wg_per_cu;
  // slab runs: workgroup slots left free for the transport's kernels while a halo is in flight (slab.hip) -- counted
  // against THIS kernel's own chip-filling grid
  if (ctx->opt.tv_reserve_wg > 0) ngrid = std::max<i64>(ngrid - ctx->opt.tv_reserve_wg, 1);
  if (ctx->opt.tv_max_wg > 0 && ngrid > ctx->opt.tv_max_wg) ngrid = ctx->opt.tv_max_wg;
  if (ngrid > nblk) ngrid = nblk;
  unsigned char* scratch = nullptr;
  const size_t per_wg = (size_t)(n + 2 * NP - 1) * R * 20;
  if ((size_t)ngrid * per_wg > ((size_t)16 << 30)) ngrid = (i64)(((size_t)16 << 30) / per_wg);
  for (; ngrid >= 1; ngrid /= 2) {
    if (ws(ctx, WS_TVSCRATCH, per_wg * (size_t)ngrid, &scratch) == VISFD_HIP_OK) break;
    scratch = nullptr;
    set_error("");
    (void)hipGetLastError();
  }
  if (!scratch) return VISFD_HIP_OK;
  if (ctx->opt.tv_poison) {   // tests: everything the kernel may read without having written it becomes NaN
    VH_HIP(hipMemsetAsync(scratch, 0xff, per_wg * (size_t)ngrid, st));
    VH_HIP(hipMemsetAsync(ten, 0xff, sizeof(float) * 6 * (size_t)(nx * ny * nz), st));
    VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    lds_poison_kernel<<<dim3(1024), dim3(256), 160 * 1024, st>>>(counter + 2);
  }
#define VH_BOX_LAUNCH(MSK, MD)                                                                        \
  do {                                                                                               \
    VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tv_box_kernel<MSK, MD>),               \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
    tv_box_kernel<MSK, MD><<<dim3((unsigned)ngrid), dim3(NT), lds, st>>>(sal, dir, ten, mask_src,     \
                                                                        mask_dst, dtab_box, p, counter, \
                                                                        (unsigned)nblk, scratch);    \
  } while (0)
  if (mask_src) { if (exponent == 4) VH_BOX_LAUNCH(true, 0); else VH_BOX_LAUNCH(true, 2); }
  else          { if (exponent == 4) VH_BOX_LAUNCH(false, 0); else VH_BOX_LAUNCH(false, 2); }
#undef VH_BOX_LAUNCH
  VH_HIP(hipGetLastError());
#ifdef VH_TV_COUNT
  {
    unsigned long long c4[8], z4[8] = {};
    VH_HIP(hipStreamSynchronize(st));
    VH_HIP(hipMemcpyFromSymbol(c4, HIP_SYMBOL(g_box_counts), sizeof(c4)));
    fprintf(stderr, "[tv_box counts] list entries tested %.4g  hits (sender, sub-patch) %.4g  vote steps %.4g  -> hits per tested entry %.3f, "
            "stream fill %.3f\n", (double)c4[0], (double)c4[1], (double)c4[2], (double)c4[1] / (double)c4[0], (double)c4[1] / (2.0 * (double)c4[2]));
    VH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_box_counts), z4, sizeof(z4)));
  }
#endif
#ifdef VH_TV_STAMPS
  {
    unsigned long long st8[8], z8[8] = {};
    VH_HIP(hipStreamSynchronize(st));
    VH_HIP(hipMemcpyFromSymbol(st8, HIP_SYMBOL(g_box_stamps), sizeof(st8)));
    double tot = 0;
    for (int i = 0; i < 6; i++) tot += (double)st8[i];
    fprintf(stderr, "[tv_box stamps] share of wave time: listing+claim %.3f | fill %.3f | barrier before sweep %.3f | test+vote %.3f | "
            "barrier after sweep %.3f | stores+rest %.3f  (total %.3g ticks over %lld waves)\n", st8[0] / tot, st8[1] / tot,
            st8[2] / tot, st8[3] / tot, st8[4] / tot, st8[5] / tot, tot, (long long)ngrid * NW);
    VH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_box_stamps), z8, sizeof(z8)));
  }
#endif
  *handled = true;
  return VISFD_HIP_OK;
}

}  // namespace vh
