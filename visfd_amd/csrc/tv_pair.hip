// tv_pair.hip -- dense stick tensor voting in TOLERANCE MODE (context option tv_fma) for gfx950
// (reference lib/visfd/feature.hpp:1914-2037 and :2217-2384; surfaces with angular exponent 2 or 4).
//
// BASELINE.json's north_star asks for vote tensors within 1e-5 relative, not for the reference's bits.  Giving up the
// bits buys two things the exact kernel (tv_tiled.hip) cannot have:
//
//   * FUSED MULTIPLY-ADDS: a vote is 19 vector instructions instead of 32 (vote_fma below);
//   * FREE ORDER OF ACCUMULATION, used here for MIRROR-PAIRED SENDER PLANES.  The exact kernel must visit the sender
//     planes of a pair of receiver planes (z, z+1) with jz ascending: 2h+2 barrier intervals, each bringing one plane's
//     list and one (2h+1)^2 slice of the vote table into LDS, and the skew of eight waves at the two barriers of an
//     interval is what that kernel loses most of its time to (profiles/r02_tv_design.txt, profiles/r03_tv_*).  The table
//     is symmetric under jz -> -jz up to the sign of rhat_z, and the sender planes z + d and z + 1 - d (d = 1..h+1) see the
//     receiver pair at jz = (-d, 1-d) and (d-1, d): the SAME two slices |jz| = d-1 and d, with the halves of the wave
//     swapped and rhat_z negated.  So one interval serves both planes: h+1 intervals instead of 2h+2, every thread brings
//     one list entry, one new slice per interval, and because the direction of d may alternate from one pass to the next,
//     never a slice reload at a turn.  Two receiver pairs go through the steps together (NP below): four lists of 128
//     entries per interval.
//
// Everything else follows tv_tiled.hip: persistent workgroups claiming units (an 8 x 32 tile of receivers over a run of
// receiver planes) from a global counter; sender planes LISTED once per unit into a per-workgroup ring in global memory
// (here 20 bytes per sender: saliency and normal as one float4, position bytes + e'x^2+e'y^2 as one word; the table
// offset is recomputed at replay) and replayed from there; a wave = 8 x 4 x 2 receivers; senders tested against the wave
// with one v_dot4_i32_i8 and voted under the execution mask; row-range culling of the list per wave.
//
// Results differ from the reference's in the last bits (tests/test_tolerance_modes.py: within 1e-5 of the field's scale on
// every case the exact kernel is tested on, including crops of the 1024^3 bench volume).
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace vh {

namespace {

#ifndef VH_PAIR_NT
#define VH_PAIR_NT 512
#endif
constexpr int NT = VH_PAIR_NT;
constexpr int NW = NT / 64;
// RECEIVERS PER LANE (build parameter).  A wave's patch is 8 x 4 receivers (x 2 mirrored planes).  With VH_PAIR_TX = 16 every
// lane owns the receivers at x and x + 8 of a 16 x 4 patch (NS = 2 sub-patches, one receiver pair per pass): both see the
// same list entries, so a sender's position word and {saliency, normal} are read from LDS once for two votes and a tile
// lists 4.4 instead of 7.0 region voxels per receiver column.  Measured at 1024^3: 400 ms against 395 ms for the 8-wide
// tile with two receiver pairs per pass (the listing share falls from 0.09 to 0.06 of a wave's time, the sweep grows by as
// much: 1.78 instead of 1.44 tests per vote step) -- kept as a parameter, not the default (profiles/r03_tv_design.txt).
#ifndef VH_PAIR_TX
#define VH_PAIR_TX 8
#endif
// WAVES SIDE BY SIDE (build parameter VH_PAIR_WX = 2): a 16 x 16 tile, the waves' 8 x 4 patches in two columns.  The waves'
// reach windows overlap more than in the 8 x 32 tile (their vote counts scatter less), at the price of more failed tests.
#ifndef VH_PAIR_WX
#define VH_PAIR_WX 1
#endif
constexpr int WX = VH_PAIR_WX;
constexpr int NS = (WX == 1) ? VH_PAIR_TX / 8 : 1;   // receivers per lane and plane pair
constexpr int TX = 8 * NS * WX, TY = 4 * NW / WX;
static_assert((TX == 8 || TX == 16) && (WX == 1 || WX == 2), "one or two 8-column sub-patches per wave, or two waves side by side");
// RECEIVER PAIRS PER PASS.  The sender planes of the receiver pairs (z, z+1) and (z+2, z+3) at step d are four different
// planes, but they need the SAME two table slices S_(d-1), S_d: with NP = 2 a workgroup takes both pairs through the
// steps together -- four lists and four sweeps per barrier interval (each wave: 2 x 6 sums), the same two slices, half as
// many intervals, fills and barriers per vote.  LDS is unchanged: 4 lists of 128 entries instead of 2 of 256.
#ifndef VH_PAIR_REMAT
#define VH_PAIR_REMAT 1
#endif
#ifndef VH_PAIR_MIRROR
#define VH_PAIR_MIRROR (VH_PAIR_PACK ? 1 : 0)
#endif
#ifndef VH_PAIR_NP
#define VH_PAIR_NP (VH_PAIR_TX == 8 ? 2 : 1)
#endif
constexpr int NP = VH_PAIR_NP;
constexpr int NLIST = 2 * NP;          // lists per interval: (A, B) of pair 0, (A, B) of pair 1
constexpr int CAPH = NT / NLIST;       // list entries per sender plane held in LDS per sweep: one per thread of its share of the workgroup
constexpr int WPL = NW / NLIST;        // waves that bring (and count the row ranges of) one list
static_assert(CAPH % 64 == 0 && WPL >= 1, "a list is brought by whole waves");
constexpr int NCH_MAX = (NT >= 512 && (TX == 8 || WX == 2)) ? 4 : 5;   // chunks of the region per wave the two-plane lister handles
constexpr int LSTRIDE = CAPH + 8;      // LDS entries per list (8 never-hit entries of slack behind each list)
// PACKED LISTS.  The lists of a step share the NT entry slots of LDS: their entries are dealt to the threads as ONE sequence
// (list 0, then list 1, ...), each list's share of an interval lands contiguously (an even start, 8 never-hit entries of slack
// behind it), and an interval ends when NT entries are in -- not when the longest list has had CAPH.  At 1024^3 a step brings
// 349 entries on average but its longest list often more than 128: 1.34 intervals per step with fixed quarters, 1.0x packed.
#ifndef VH_PAIR_PACK
#define VH_PAIR_PACK 1
#endif
constexpr int LSLOTS = VH_PAIR_PACK ? NT + 12 * NLIST : NLIST * LSTRIDE;   // LDS entry slots
constexpr unsigned OOB = 0x7ffffff0u;  // byte offset beyond any plane descriptor: reads give 0

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, 0));
}

typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(3))) T* lds_ptr(unsigned a) {
  return (const __attribute__((address_space(3))) T*)(uintptr_t)a;
}

#ifdef VH_TV_STAMPS   // development build (tools/build_variant.py): where a wave's time goes
__device__ unsigned long long g_pair_stamps[8];
#define VH_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += (unsigned)(t_ - st_last); st_last = t_; } while (0)
#else
#define VH_STAMP(i) do {} while (0)
#endif

#ifdef VH_TV_IMBAL   // development build: the slowest wave's sweep time against the mean, per barrier interval
__device__ unsigned long long g_pair_imbal[4];
#endif

#ifdef VH_TV_COUNT   // development build: how many senders the waves test, how many of those vote, how many lanes they reach
__device__ unsigned long long g_pair_counts[8];   // tested, voted, lanes, -, steps with senders, barrier intervals, list entries, -
#endif

struct PairParams {
  int nx, ny, nz;
  int z_out0, z_out1;    // receiver planes [z_out0, z_out1)
  int h;
  int rw, rh;            // region width = TX + 2h, height = TY + 2h
  int rw_magic;          // q / rw == (q * rw_magic) >> 20 for every region position q (checked by the launcher)
  int nchunk;            // 64-voxel chunks of the region per wave
  int tiles_x, tiles_y;
  int zrun;              // receiver planes per unit of work
  int relist;            // 1: list every sender plane again for every receiver pair (option tv_no_replay; tests)
  int sp;                // row stride of a table slice in float4 entries (>= 2h+1)
};

__device__ __forceinline__ void fmacc(float& t, float a, float b) {
  asm("v_fmac_f32 %0, %1, %2" : "+v"(t) : "v"(a), "v"(b));
}

// One vote.  The table holds R = sqrt(2) rhat (tv.hip: tv_table_device), so that t = R.n = sqrt(2) u, t R - n = 2 u rhat - n = m
// and 2 - t^2 = 2 (1 - u^2); the factor 1/2 (exponent 2) or 1/4 (exponent 4) the decay then lacks is applied to the sender's
// saliency when it is listed (an exact scaling).  ZNEG: the slice in LDS is the one of -jz: rhat_z has the opposite sign.
template <int MODE, bool ZNEG>
__device__ __forceinline__ void vote_fma(float T[6], float sal, float fv, float R0, float R1, float R2, float n0, float n1, float n2) {
  const float Rz = ZNEG ? -R2 : R2;    // (a source modifier of the instructions below)
  const float t = __builtin_fmaf(Rz, n2, __builtin_fmaf(R1, n1, R0 * n0));
  const float q = __builtin_fmaf(-t, t, 2.0f);
  const float m0 = __builtin_fmaf(t, R0, -n0);
  const float m1 = __builtin_fmaf(t, R1, -n1);
  const float m2 = __builtin_fmaf(t, Rz, -n2);
  const float sw = sal * fv;
  const float bse = (MODE == 0) ? (sw * q) * q : sw * q;
  const float b0 = bse * m0, b1 = bse * m1, b2 = bse * m2;
  fmacc(T[0], b0, m0);
  fmacc(T[3], b0, m1);
  fmacc(T[5], b0, m2);
  fmacc(T[1], b1, m1);
  fmacc(T[4], b1, m2);
  fmacc(T[2], b2, m2);
}

template <bool MASKED_SRC, int MODE>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8)))
tv_pair_kernel(const float* __restrict__ sal, const float* __restrict__ dir, float* __restrict__ ten,
               const float* __restrict__ mask_src, const float* __restrict__ mask_dst,
               const float4* __restrict__ table /* [(2h+1)^3] : w, sqrt(2) rhat at j */,
               PairParams p, unsigned* __restrict__ tile_counter, unsigned ntiles,
               unsigned char* __restrict__ scratch /* per-workgroup rings of compacted sender planes */) {
  // l_ent[e]  float4 {sal (scaled), n0, n1, n2} of list entry e; list 2 pp (the sender plane ABOVE receiver pair pp) in
  //           [2 pp LSTRIDE, 2 pp LSTRIDE + CAPH), list 2 pp + 1 (the plane below it) one LSTRIDE further
  // l_pos[e]  {distance-test operand, table offset E}: packed signed bytes (e'x, e'y, -(|e'|^2 >> 7), |e'|^2 & 127) with e' =
  //           sender position relative to the tile centre and the LOWER receiver plane; 8 never-hit entries behind each list
  __shared__ __attribute__((aligned(16))) float4 l_ent[LSLOTS];
  __shared__ __attribute__((aligned(16))) uint2 l_pos[LSLOTS];
  __shared__ float l_mv[MASKED_SRC ? LSLOTS : 1];
  __shared__ int wave_tot[2][2][NW];
#if !VH_PAIR_PACK
  __shared__ int cull[NW][2 * NW];           // per wave holding entries (waves WPL k .. WPL k + WPL - 1: list k): entries above / not below each wave's rows
#endif
  __shared__ unsigned claimed_tile;
#ifdef VH_TV_IMBAL
  __shared__ unsigned imb_max, imb_sum;
#endif
  __shared__ int plane_cnt[88];              // entries per ring slot, [2h + 2 NP] (h <= 40)
  extern __shared__ __attribute__((aligned(16))) unsigned char slices[];   // two table slices: S_j (jz = +j) in slot j & 1

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = p.h;
  const int S = 2 * h + 1;
  // LDS rows of a table slice are SP float4 apart, SP = S rounded up to 4 mod 8, and the lanes of a half wave are dealt to
  // its 8 x 4 receivers so that each of ds_read_b128's two 16-lane groups ({0-3, 12-15, 20-27} and the rest) is one
  // 4-column block of all four rows: with row offsets of 64 or 192 bytes modulo the 256 bytes of the 64 banks the four
  // 64-byte segments of a group fall on different banks -- 2 LDS cycles per half wave, where rows of 2h+1 entries with lanes
  // in row order take 4 (tools/lds_bank_model.py).  The table in global memory has the same padded rows (tv.hip).
  const int SP = p.sp;
  const int nsl = S * SP;
  const int R = p.rw * p.rh;
  const i64 plane = (i64)p.nx * p.ny;
  const i64 nvox = plane * p.nz;
  const int plane_bytes = (int)(plane * 4);
  constexpr int ENT_BYTES = MASKED_SRC ? 24 : 20;
  const size_t plane_stride = (size_t)R * ENT_BYTES;   // a ring slot: float4 ent[R]; unsigned pos[R]; (float mv[R])
  const int P = S + 2 * NP - 1;   // sender planes the receiver planes of a pass reach = slots of the ring
  unsigned char* const ring = scratch + (size_t)blockIdx.x * plane_stride * P;
  int npar = 0;
  int slot_has[2] = {-1, -1};                // which slice S_j each LDS slot holds (uniform)
  bool up = false;                           // direction of d for the next receiver pair (flips after every pair)
  float4* const sl4 = reinterpret_cast<float4*>(slices);
#ifdef VH_TV_STAMPS
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif

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

    // lane -> receiver inside the wave's 8 x 4 patch (see SP above): lanes 0-3, 12-15, 20-23, 24-27 are columns 0-3 of rows
    // 0, 1, 2, 3; lanes 4-7, 8-11, 16-19, 28-31 columns 4-7 of rows 0, 1, 2, 3.  half: 0 = receiver plane rz, 1 = plane rz + 1.
    // distance test as one dot product (tv_tiled.hip):  |r'-e'|^2 - h^2 - 1 = (-2r'x, -2r'y, -128, 1).(e'x, e'y, -q, m) + (|r'|^2 - h^2 - 1)
    struct LaneConst {
      int half, lx, ly;
      unsigned recv4[NS];
      int recv_c[NS];
      unsigned r16_0;     // LDS address of this lane's table entry of j = 0 in slice slot 0
    };
    // MIRRORED ROW BLOCKS.  A wave's receivers of pair pp are the four rows of block wv(pp): its own number for the first pair,
    // NW - 1 - (its number) for the second (VH_PAIR_MIRROR = 2: half a tile further, cyclically -- the same effect).  The
    // sweep of an interval lasts as long as its slowest wave (measured: 1.36 x the mean wave), and which rows are heavy -- those
    // near a membrane -- is much the same for the two pairs of a pass: a wave that is heavy for one pair is lighter for the
    // other.  Measured: max / mean 1.36 -> 1.31, 381 -> 376 ms; what remains is the scatter of eight waves' vote counts.
    auto row_block = [&](int pp) -> int {
      if (VH_PAIR_MIRROR == 1 && (pp & 1)) return NW - 1 - wave;
      if (VH_PAIR_MIRROR == 2 && (pp & 1)) return (wave + NW / 2) % NW;
      return wave;
    };
    auto lane_consts = [&](unsigned ln, int wv) -> LaneConst {
      LaneConst c;
      c.half = (int)(ln >> 5);
      const int l5 = (int)(ln & 31);
      const int lrow = l5 >> 3;
      const int lcol = (l5 & 3) + (((0xc33cu >> (l5 >> 1)) & 1u) ? 4 : 0);   // lane pairs 2-5, 8-9, 14-15 (lanes 4-11, 16-19, 28-31): the right block
      c.lx = (wv % WX) * 8 + lcol;
      c.ly = (wv / WX) * 4 + lrow;
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const int rpx = c.lx + 8 * s - TX / 2, rpy = c.ly - TY / 2;
        c.recv4[s] = (unsigned)((-2 * rpx) & 0xff) | ((unsigned)((-2 * rpy) & 0xff) << 8) | (0x80u << 16) | (1u << 24);
        c.recv_c[s] = rpx * rpx + rpy * rpy - h * h - 1;
      }
      c.r16_0 = lds_addr(slices) + (unsigned)(16 * ((c.ly + 2 * h) * SP + c.lx + 2 * h));
      return c;
    };
    // The constants are RECOMPUTED from the lane number wherever a phase needs them (pass start, every step): kept live
    // across the sweeps they are spilled, and a step would begin with their reloads from scratch memory in front of its ring
    // loads.  (The empty asm hides the lane number's origin, so that the compiler cannot share one computation.)
    auto fresh_lane = [&]() -> unsigned {
      unsigned ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#if VH_PAIR_REMAT
      asm volatile("" : "+v"(ln));
#endif
      return ln;
    };
    constexpr unsigned NEVER_HIT = 0x009c0000u;
    const unsigned ent_base = lds_addr(l_ent);

    // position word of a list entry = the sender's operand of the distance test: signed bytes (e'x, e'y, -q, m) with
    // e'x^2 + e'y^2 = 128 q + m.  It does not depend on the receiver planes: e'z^2 is added to the receivers' accumulator
    // operand instead, so the word is built once, when the sender is listed.
    auto pos_word = [&](int epx, int epy) -> unsigned {
      const int e2 = epx * epx + epy * epy;
      return (unsigned)(epx & 0xff) | ((unsigned)(epy & 0xff) << 8) | ((unsigned)((-(e2 >> 7)) & 0xff) << 16) | ((unsigned)(e2 & 127) << 24);
    };

    // ---- LISTING, two planes at a time (window regions of <= 4 chunks per wave: h <= 12).  Every load of a phase is in
    // flight at once -- the saliencies of both planes (kept in registers across the barrier: one read per voxel), then the
    // normals of a plane's salient voxels -- and both planes share one barrier: the listing of the two sender planes that
    // enter the window with every pair of receiver planes costs two memory round trips instead of sixteen.
    // A plane index < 0 means "no plane" (zero-length descriptors: nothing is salient).
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
          n0[j] = n1[j] = n2[j] = mvv[j] = 0.0f;
          if (sv[k][j] != 0.0f) {
            n0[j] = buf_load(rd0, off[j]);
            n1[j] = buf_load(rd1, off[j]);
            n2[j] = buf_load(rd2, off[j]);
            if (MASKED_SRC) mvv[j] = buf_load(rm, off[j]);
          }
        }
#pragma unroll
        for (int j = NCH - 1; j >= 0; j--) {   // descending region position = row order, which the culling needs
          const bool f = sv[k][j] != 0.0f;
          const unsigned long long bal = __builtin_amdgcn_ballot_w64(f);
          const int tb = __builtin_popcountll(bal);
          const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
          if (f) {
            const int idx = running + (tb - below - 1);
            const int q = q0 + 64 * j;
            const int ey = (int)(((unsigned)q * (unsigned)p.rw_magic) >> 20);
            const int ex = q - ey * p.rw;
            reinterpret_cast<float4*>(ring_plane)[idx] = make_float4(sv[k][j] * (MODE == 0 ? 0.25f : 0.5f), n0[j], n1[j], n2[j]);
            reinterpret_cast<unsigned*>(ring_plane + (size_t)R * 16)[idx] = pos_word(ex - h - TX / 2, ey - h - TY / 2);
            if (MASKED_SRC) reinterpret_cast<float*>(ring_plane + (size_t)R * 20)[idx] = mvv[j];
          }
          running += tb;
        }
      }
    };

    // ---- LISTING, one plane, any window (as tv_tiled.hip): sender plane sz of this tile's region into its ring slot ------
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
            const int idx = running + (tb - below - 1);      // descending region position: row order, which the culling needs
            const float4 a = make_float4(s * (MODE == 0 ? 0.25f : 0.5f), buf_load(rd0, off), buf_load(rd1, off), buf_load(rd2, off));
            const int epx = ex - h - TX / 2, epy = ey - h - TY / 2;
            reinterpret_cast<float4*>(ring_plane)[idx] = a;
            reinterpret_cast<unsigned*>(ring_plane + (size_t)R * 16)[idx] = pos_word(epx, epy);
            if (MASKED_SRC) reinterpret_cast<float*>(ring_plane + (size_t)R * 20)[idx] = buf_load(rm, off);
          }
          running += tb;
        }
      }
      if (tid == 0) plane_cnt[slot] = total;
    };

    float TT[NP][NS][6];
#ifdef VH_TV_COUNT
    unsigned cnt_tested = 0, cnt_voted = 0, cnt_lanes = 0;
#endif

    // ---- the SWEEP over list entries [i0, i1) of one list (base = its first LDS entry), in list order ------------------
    // r16: this lane's table base in its slice slot (sub-patch s: 128 bytes further); rcl[s]: its accumulator operand of the
    // distance test (large: never hit)
    auto sweep = [&](auto ZN, float (&T)[NS][6], int base, int i0, int i1, unsigned r16, const int (&rcl)[NS],
                     const unsigned (&recv4)[NS]) {
      constexpr bool ZNEG = decltype(ZN)::value;
      auto vote_sub = [&](auto SUB, const f4v& d, int s, unsigned e16) {
        constexpr int sub = decltype(SUB)::value;
#if defined(VH_PAIR_ABLATE) && (VH_PAIR_ABLATE & 1)   // development: no table read (wrong results): what do the LDS reads of a vote cost?
        const float e16f = __builtin_bit_cast(float, e16 + r16);
        const f4v tw = {e16f, d.y, e16f, d.z};
#else
        const f4v tw = *lds_ptr<f4v>(r16 - e16 + 128u * sub);   // (the constant is the instruction's offset field)
#endif
        float fv = tw.x;
        if (MASKED_SRC) fv = fv * l_mv[s];
        vote_fma<MODE, ZNEG>(T[sub], d.x, fv, tw.y, tw.z, tw.w, d.y, d.z, d.w);
      };
      auto batch = [&](const uint4& ca, const uint4& cb, unsigned ent, int s0) {
        if constexpr (NS == 1) {
          int d0, d1, d2, d3;
          asm("v_dot4_i32_i8 %0, %4, %6, %5\n\t"
              "v_dot4_i32_i8 %1, %4, %7, %5\n\t"
              "v_dot4_i32_i8 %2, %4, %8, %5\n\t"
              "v_dot4_i32_i8 %3, %4, %9, %5\n\t"
            "s_nop 2"
              : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
              : "v"(recv4[0]), "v"(rcl[0]), "v"(ca.x), "v"(ca.z), "v"(cb.x), "v"(cb.z));
          auto vote_one = [&](int k, int s, unsigned e16) {
#if defined(VH_PAIR_ABLATE) && (VH_PAIR_ABLATE & 2)   // development: no entry read (wrong results)
            const float ef = __builtin_bit_cast(float, e16);
            const f4v d = {ef, ef, ef, ef};
#else
            const f4v d = *lds_ptr<f4v>(ent + 16u * (unsigned)k);
#endif
            vote_sub(std::integral_constant<int, 0>{}, d, s, e16);
          };
          if (__builtin_expect(d0 < 0, 1)) vote_one(0, s0, ca.y);
          if (__builtin_expect(d1 < 0, 1)) vote_one(1, s0 + 1, ca.w);
          if (__builtin_expect(d2 < 0, 1)) vote_one(2, s0 + 2, cb.y);
          if (__builtin_expect(d3 < 0, 1)) vote_one(3, s0 + 3, cb.w);
#ifdef VH_TV_COUNT
          {
            const int dd[4] = {d0, d1, d2, d3};
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const unsigned long long b = __builtin_amdgcn_ballot_w64(dd[k] < 0);
              cnt_tested += 1;
              cnt_voted += b ? 1 : 0;
              cnt_lanes += __builtin_popcountll(b);
            }
          }
#endif
        } else {
          // two receivers per lane: both distance tests, then ONE read of the sender for the lanes either of them reaches
          auto entry = [&](int k, int s, unsigned posw, unsigned e16) {
            int da, db;
            // (a dot result may be read by another kind of vector instruction three wait states later at the earliest,
            // and the compiler does not see hazards inside an asm block: the s_nop is part of it)
            asm("v_dot4_i32_i8 %0, %2, %6, %3\n\t"
                "v_dot4_i32_i8 %1, %4, %6, %5\n\t"
                "s_nop 2"
                : "=&v"(da), "=&v"(db)
                : "v"(recv4[0]), "v"(rcl[0]), "v"(recv4[NS - 1]), "v"(rcl[NS - 1]), "v"(posw));
            if (__builtin_expect((da | db) < 0, 1)) {
              const f4v d = *lds_ptr<f4v>(ent + 16u * (unsigned)k);
              if (da < 0) vote_sub(std::integral_constant<int, 0>{}, d, s, e16);
              if (db < 0) vote_sub(std::integral_constant<int, NS - 1>{}, d, s, e16);
            }
#ifdef VH_TV_COUNT
            {
              const unsigned long long ba = __builtin_amdgcn_ballot_w64(da < 0), bb = __builtin_amdgcn_ballot_w64(db < 0);
              cnt_tested += 2;
              cnt_voted += (ba ? 1 : 0) + (bb ? 1 : 0);
              cnt_lanes += __builtin_popcountll(ba) + __builtin_popcountll(bb);
            }
#endif
          };
          entry(0, s0, ca.x, ca.y);
          entry(1, s0 + 1, ca.z, ca.w);
          entry(2, s0 + 2, cb.x, cb.y);
          entry(3, s0 + 3, cb.z, cb.w);
        }
      };
      int s0 = base + (i0 & ~1);
      const int send = base + i1;
      const uint4* pq = reinterpret_cast<const uint4*>(l_pos) + (s0 >> 1);
      unsigned ent = ent_base + 16u * (unsigned)s0;
      asm volatile("" : "+v"(ent));
      uint4 a0 = pq[0], a1 = pq[1];
      while (s0 < send) {   // uniform
        const uint4 b0 = pq[2], b1 = pq[3];
        batch(a0, a1, ent, s0);
        if (s0 + 4 >= send) break;
        a0 = pq[4];
        a1 = pq[5];
        batch(b0, b1, ent + 64u, s0 + 4);
        pq += 4;
        ent += 128u;
        asm volatile("" : "+v"(ent));
        s0 += 8;
      }
    };

    int cached_lo = 1, cached_hi = 0;
    for (int rz = z_run0; rz < z_run1; rz += 2 * NP) {
      // sender planes that reach the LIVE receivers of this pass (a run may end inside a pass: nothing above the last
      // live receiver + h is needed -- or, in a slab run, complete -- then)
      const int sz_hi = min(min(rz + 2 * NP - 1, z_run1 - 1) + h, p.nz - 1), sz_lo = max(rz - h, 0);
#ifndef VH_PAIR_LIST2
#define VH_PAIR_LIST2 1
#endif
      if (VH_PAIR_LIST2 && p.nchunk <= NCH_MAX) {
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

      // this lane's receivers: plane rz + 2 pp + half of pair pp
      i64 rc[NP];                // (sub-patch s: rc + 8 s)
      bool r_live[NP][NS];
#pragma unroll
      for (int pp = 0; pp < NP; pp++) {
        const LaneConst pc = lane_consts(fresh_lane(), row_block(pp));
        const int rx = x0 + pc.lx, ry = y0 + pc.ly;             // (sub-patch s: column rx + 8 s)
        const int rzl = rz + 2 * pp + pc.half;
        const bool z_in = rzl < z_run1;
        rc[pp] = (i64)rzl * plane + (i64)ry * p.nx + rx;
#pragma unroll
        for (int s = 0; s < NS; s++) {
          const bool in = rx + 8 * s < p.nx && ry < p.ny && z_in;
          r_live[pp][s] = in && !(mask_dst && mask_dst[in ? rc[pp] + 8 * s : 0] == 0.0f);
#pragma unroll
          for (int k = 0; k < 6; k++) TT[pp][s][k] = 0.0f;
        }
      }
      __syncthreads();   // ring entries and counts of this pass are visible
      VH_STAMP(0);

      // d = 1 .. h+1.  Pair pp (receiver planes z = rz + 2 pp and z + 1): sender planes A = z + d (above: jz = -d for the
      // lower receiver plane, 1-d for the upper one) and B = z + 1 - d (below: jz = d-1 and d).  All of them need the slices
      // S_(d-1) and S_d; the direction of d alternates from pass to pass, so that every step -- the first of a pass
      // included -- finds one of its two slices in LDS already.
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
          lcnt[2 * pp] = (pair_live && lsz[2 * pp] <= sz_hi) ? __builtin_amdgcn_readfirstlane(plane_cnt[lsz[2 * pp] % P]) : 0;
          lcnt[2 * pp + 1] = (pair_live && lsz[2 * pp + 1] >= sz_lo) ? __builtin_amdgcn_readfirstlane(plane_cnt[lsz[2 * pp + 1] % P]) : 0;
          cmax = max(cmax, max(lcnt[2 * pp], lcnt[2 * pp + 1]));
        }
        if (cmax == 0) continue;   // uniform
        // slices S_(d-1) and S_d (S_(h+1) does not exist: its lanes never hit); lists and slices are free: every sweep ends
        // with a barrier
        int need[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const int j = d - 1 + k;
          need[k] = (j <= h && slot_has[j & 1] != j) ? j : -1;
          if (need[k] >= 0) slot_has[j & 1] = j;
        }
        // rows a wave can reach: the nearer of its two receiver planes is |jz| = d-1 away from either sender plane
        const int jn = (d - 1) * (d - 1);
        int rho = (int)__builtin_sqrtf((float)(h * h - jn));
        while (rho * rho > h * h - jn) rho--;
        while ((rho + 1) * (rho + 1) <= h * h - jn) rho++;
#if VH_PAIR_PACK
        const int half = (int)(fresh_lane() >> 5);
        const int t512 = wave * 64 + (int)fresh_lane();        // this thread's place in the interval's entry sequence
        int pre[NLIST + 1];                                    // (uniform) first position of list k in the step's sequence
        pre[0] = 0;
#pragma unroll
        for (int k = 0; k < NLIST; k++) pre[k + 1] = pre[k] + lcnt[k];
        const int total = pre[NLIST];
        int pl[NLIST];                                         // (uniform) ring slot of list k's plane
#pragma unroll
        for (int k = 0; k < NLIST; k++) pl[k] = __builtin_amdgcn_readfirstlane(((lsz[k] % P) + P) % P);
#ifdef VH_TV_COUNT
        if (tid == 0) {
          atomicAdd(&g_pair_counts[4], 1ull);
          atomicAdd(&g_pair_counts[5], (unsigned long long)((total + NT - 1) / NT));
          atomicAdd(&g_pair_counts[6], (unsigned long long)total);
        }
#endif
        for (int done = 0; done < total; done += NT) {   // uniform
          // list k's share of this interval: sequence positions [c[k], c[k] + len[k]) of the NT, LDS slots from S[k]
          int c[NLIST], len[NLIST], S[NLIST];
#pragma unroll
          for (int k = 0; k < NLIST; k++) {
            const int lo = min(max(pre[k], done), done + NT), hi = min(pre[k + 1], done + NT);
            c[k] = lo - done;
            len[k] = max(hi - lo, 0);
            S[k] = ((c[k] + 1) & ~1) + 10 * k;
          }
          const int g = done + t512;
          int k_me = 0;
#pragma unroll
          for (int k = 1; k < NLIST; k++) k_me += (g >= pre[k]) ? 1 : 0;
          const bool have = g < total;
          int idx = g, slot = t512, pl_me = pl[0];
#pragma unroll
          for (int k = 0; k < NLIST; k++)
            if (k_me == k) { idx = g - pre[k]; slot = S[k] + (t512 - c[k]); pl_me = pl[k]; }
          const unsigned char* ring_plane = ring + (size_t)pl_me * plane_stride;
          int epy = -128;                                      // threads without an entry: below every range
          float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          unsigned m = 0u;
          float mvv = 0.0f;
          if (have) {
            a = reinterpret_cast<const float4*>(ring_plane)[idx];
            m = reinterpret_cast<const unsigned*>(ring_plane + (size_t)R * 16)[idx];
            if (MASKED_SRC) mvv = reinterpret_cast<const float*>(ring_plane + (size_t)R * 20)[idx];
          }
          if (done == 0) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
              if (need[k] < 0) continue;   // uniform
              const int j = need[k];
              const float4* src4 = table + (i64)(j + h) * nsl;
              float4* dst4 = sl4 + (j & 1) * nsl;
              for (int i = tid; i < nsl; i += NT) dst4[i] = src4[i];
            }
          }
          if (have) {
            l_ent[slot] = a;
            const int epx = (int)(signed char)(m & 0xff);
            epy = (int)(signed char)((m >> 8) & 0xff);
            const unsigned e16 = (unsigned)(16 * ((epy + h + TY / 2) * SP + (epx + h + TX / 2)));
            l_pos[slot] = make_uint2(m, e16);
            if (MASKED_SRC) l_mv[slot] = mvv;
          }
          if (t512 < 8 * NLIST) {                               // 8 never-hit entries behind every (non-empty) list's share
            int sk = S[0] + len[0], lk = len[0];
#pragma unroll
            for (int k = 1; k < NLIST; k++)
              if ((t512 >> 3) == k) { sk = S[k] + len[k]; lk = len[k]; }
            if (lk > 0) l_pos[sk + (t512 & 7)] = make_uint2(NEVER_HIT, 0u);
          }
          VH_STAMP(1);
          __syncthreads();   // lists (and slices) complete
          VH_STAMP(2);
#ifdef VH_TV_IMBAL
          const unsigned long long imb_t0 = __builtin_amdgcn_s_memtime();
          if (tid == 0) { imb_max = 0u; imb_sum = 0u; }
#endif
          // entries are in descending row order: of list k, this wave needs those from the first one at or below row
          // 4w-13+rho to the last one at or above row 4w-16-rho.  Every wave counts both kinds itself, from the row bytes of
          // the position words in LDS, 64 entries at a time.
          int i0[NLIST], i1[NLIST];
          {
            const int ln = (int)fresh_lane();
#pragma unroll
            for (int k = 0; k < NLIST; k++) {
              const int wv = row_block(k >> 1) / WX;
              const int hi_row = 4 * wv - (TY / 2 - 3) + rho, lo_row = 4 * wv - TY / 2 - rho;
              int above = 0, upto = 0;
              for (int j = 0; j < len[k]; j += 64) {   // uniform
                int ey = -128;
                if (j + ln < len[k]) ey = (int)(signed char)((l_pos[S[k] + j + ln].x >> 8) & 0xff);
                above += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ey > hi_row));
                upto += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ey >= lo_row));
              }
              i0[k] = above;
              i1[k] = upto;
            }
          }
#pragma unroll
          for (int pp = 0; pp < NP; pp++) {
            if (i1[2 * pp] <= i0[2 * pp] && i1[2 * pp + 1] <= i0[2 * pp + 1]) continue;   // uniform
            const LaneConst lc = lane_consts(fresh_lane(), row_block(pp));
            const unsigned r16_0 = lc.r16_0;
            // plane A (above): the lower receiver plane sees it at jz = -d (slice S_d, rhat_z negated), the upper one at
            // 1-d (S_(d-1)); |r-e|^2 of the upper plane's receivers differs by 1 - 2 e'z
            if (i1[2 * pp] > i0[2 * pp]) {
              const int jl = d, ju = d - 1;
              const bool zok = half ? ju <= h : jl <= h;
              const unsigned r16 = r16_0 + (unsigned)(16 * nsl) * (unsigned)((half ? ju : jl) & 1);
              int rcl[NS];
#pragma unroll
              for (int s = 0; s < NS; s++) rcl[s] = (r_live[pp][s] && zok) ? lc.recv_c[s] + d * d + (half ? 1 - 2 * d : 0) : 0x100000;
              sweep(std::true_type{}, TT[pp], S[2 * pp], i0[2 * pp], i1[2 * pp], r16, rcl, lc.recv4);
            }
            // plane B (below): jz = d-1 for the lower plane (S_(d-1)), d for the upper one (S_d)
            if (i1[2 * pp + 1] > i0[2 * pp + 1]) {
              const int jl = d - 1, ju = d;
              const bool zok = half ? ju <= h : jl <= h;
              const unsigned r16 = r16_0 + (unsigned)(16 * nsl) * (unsigned)((half ? ju : jl) & 1);
              int rcl[NS];
#pragma unroll
              for (int s = 0; s < NS; s++)
                rcl[s] = (r_live[pp][s] && zok) ? lc.recv_c[s] + (1 - d) * (1 - d) + (half ? 1 - 2 * (1 - d) : 0) : 0x100000;
              sweep(std::false_type{}, TT[pp], S[2 * pp + 1], i0[2 * pp + 1], i1[2 * pp + 1], r16, rcl, lc.recv4);
            }
          }
          VH_STAMP(3);
#ifdef VH_TV_IMBAL
          if (lane == 0) {
            const unsigned dt = (unsigned)(__builtin_amdgcn_s_memtime() - imb_t0);
            atomicMax(&imb_max, dt);
            atomicAdd(&imb_sum, dt);
          }
#endif
          __syncthreads();   // everyone done reading before the lists or the slices are refilled
#ifdef VH_TV_IMBAL
          if (tid == 0) { atomicAdd(&g_pair_imbal[0], (unsigned long long)imb_max); atomicAdd(&g_pair_imbal[1], (unsigned long long)imb_sum); atomicAdd(&g_pair_imbal[2], 1ull); }
#endif
          VH_STAMP(4);
        }
#else
        const int li = wave / WPL;                             // (uniform) the list this thread brings entries of
        int my_sz = lsz[0], my_cnt = lcnt[0];
#pragma unroll
        for (int k = 1; k < NLIST; k++)
          if (li == k) { my_sz = lsz[k]; my_cnt = lcnt[k]; }
        const unsigned char* ring_plane = ring + (size_t)(((my_sz % P) + P) % P) * plane_stride;
        const LaneConst lc = lane_consts(fresh_lane(), wave);
        const int half = lc.half;
        const unsigned r16_0 = lc.r16_0;
        const int ltid = VH_PAIR_REMAT ? (wave % WPL) * 64 + (int)(fresh_lane()) : (tid & (CAPH - 1));
        const int lbase = li * LSTRIDE;
#ifdef VH_TV_COUNT
        if (tid == 0) {
          atomicAdd(&g_pair_counts[4], 1ull);
          atomicAdd(&g_pair_counts[5], (unsigned long long)((cmax + CAPH - 1) / CAPH));
          unsigned long long tot = 0;
          for (int k = 0; k < NLIST; k++) tot += (unsigned long long)lcnt[k];
          atomicAdd(&g_pair_counts[6], tot);
        }
#endif
        for (int done = 0; done < cmax; done += CAPH) {   // uniform
          const int take = min(CAPH, max(my_cnt - done, 0));
          int epy = -128;                                      // threads without an entry: below every range
          float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          unsigned m = 0u;
          float mvv = 0.0f;
          if (ltid < take) {
            const int idx = done + ltid;
            a = reinterpret_cast<const float4*>(ring_plane)[idx];
            m = reinterpret_cast<const unsigned*>(ring_plane + (size_t)R * 16)[idx];
            if (MASKED_SRC) mvv = reinterpret_cast<const float*>(ring_plane + (size_t)R * 20)[idx];
          }
          if (done == 0) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
              if (need[k] < 0) continue;   // uniform
              const int j = need[k];
              const float4* src4 = table + (i64)(j + h) * nsl;
              float4* dst4 = sl4 + (j & 1) * nsl;
              for (int i = tid; i < nsl; i += NT) dst4[i] = src4[i];
            }
          }
          if (ltid < take) {
            l_ent[lbase + ltid] = a;
            const int epx = (int)(signed char)(m & 0xff);
            epy = (int)(signed char)((m >> 8) & 0xff);
            const unsigned e16 = (unsigned)(16 * ((epy + h + TY / 2) * SP + (epx + h + TX / 2)));
            l_pos[lbase + ltid] = make_uint2(m, e16);
            if (MASKED_SRC) l_mv[lbase + ltid] = mvv;
          }
          if (ltid < 8) l_pos[lbase + take + ltid] = make_uint2(NEVER_HIT, 0u);
          // entries are in descending row order: wave w needs those from the first one at or below row 4w-13+rho to the
          // last one at or above row 4w-16-rho
#pragma unroll
          for (int w = 0; w < NW; w++) {
            const int above = __builtin_popcountll(__builtin_amdgcn_ballot_w64(epy > 4 * w - (TY / 2 - 3) + rho));
            const int upto = __builtin_popcountll(__builtin_amdgcn_ballot_w64(epy >= 4 * w - TY / 2 - rho));
            if (lane == 0) { cull[wave][2 * w] = above; cull[wave][2 * w + 1] = upto; }
          }
          VH_STAMP(1);
          __syncthreads();   // lists (and slices) complete
          VH_STAMP(2);
          int i0[NLIST], i1[NLIST];
#pragma unroll
          for (int k = 0; k < NLIST; k++) {
            int s0 = 0, s1 = 0;
#pragma unroll
            for (int w = 0; w < WPL; w++) { s0 += cull[k * WPL + w][2 * wave]; s1 += cull[k * WPL + w][2 * wave + 1]; }
            i0[k] = __builtin_amdgcn_readfirstlane(s0);
            i1[k] = __builtin_amdgcn_readfirstlane(s1);
          }
#pragma unroll
          for (int pp = 0; pp < NP; pp++) {
            // plane A (above): the lower receiver plane sees it at jz = -d (slice S_d, rhat_z negated), the upper one at
            // 1-d (S_(d-1)); |r-e|^2 of the upper plane's receivers differs by 1 - 2 e'z
            if (i1[2 * pp] > i0[2 * pp]) {
              const int jl = d, ju = d - 1;
              const bool zok = half ? ju <= h : jl <= h;
              const unsigned r16 = r16_0 + (unsigned)(16 * nsl) * (unsigned)((half ? ju : jl) & 1);
              int rcl[NS];
#pragma unroll
              for (int s = 0; s < NS; s++) rcl[s] = (r_live[pp][s] && zok) ? lc.recv_c[s] + d * d + (half ? 1 - 2 * d : 0) : 0x100000;
              sweep(std::true_type{}, TT[pp], (2 * pp) * LSTRIDE, i0[2 * pp], i1[2 * pp], r16, rcl, lc.recv4);
            }
            // plane B (below): jz = d-1 for the lower plane (S_(d-1)), d for the upper one (S_d)
            if (i1[2 * pp + 1] > i0[2 * pp + 1]) {
              const int jl = d - 1, ju = d;
              const bool zok = half ? ju <= h : jl <= h;
              const unsigned r16 = r16_0 + (unsigned)(16 * nsl) * (unsigned)((half ? ju : jl) & 1);
              int rcl[NS];
#pragma unroll
              for (int s = 0; s < NS; s++)
                rcl[s] = (r_live[pp][s] && zok) ? lc.recv_c[s] + (1 - d) * (1 - d) + (half ? 1 - 2 * (1 - d) : 0) : 0x100000;
              sweep(std::false_type{}, TT[pp], (2 * pp + 1) * LSTRIDE, i0[2 * pp + 1], i1[2 * pp + 1], r16, rcl, lc.recv4);
            }
          }
          VH_STAMP(3);
#if !(defined(VH_PAIR_ABLATE) && (VH_PAIR_ABLATE & 4))   // (development: without it lists are refilled under the readers -- wrong results)
          __syncthreads();   // everyone done reading before the lists or the slices are refilled
#endif
          VH_STAMP(4);
        }
#endif
      }
      up = !up;
#ifdef VH_TV_COUNT
      if (lane == 0) {
        atomicAdd(&g_pair_counts[0], (unsigned long long)cnt_tested);
        atomicAdd(&g_pair_counts[1], (unsigned long long)cnt_voted);
        atomicAdd(&g_pair_counts[2], (unsigned long long)cnt_lanes);
      }
      cnt_tested = cnt_voted = cnt_lanes = 0;
#endif

      {
#pragma unroll
        for (int pp = 0; pp < NP; pp++) {
          const LaneConst sc = lane_consts(fresh_lane(), row_block(pp));   // (the receiver index again, rather than a value live since the pass began)
#if VH_PAIR_REMAT
          const i64 rcs = (i64)(rz + 2 * pp + sc.half) * plane + (i64)(y0 + sc.ly) * p.nx + (x0 + sc.lx);
#else
          const i64 rcs = rc[pp];
#endif
#pragma unroll
          for (int s = 0; s < NS; s++)
            if (r_live[pp][s]) {
#pragma unroll
              for (int k = 0; k < 6; k++) __builtin_nontemporal_store(TT[pp][s][k], &ten[k * nvox + rcs + 8 * s]);   // written once, not read here
            }
        }
      }
    }   // next pass of the run
  }   // next unit
#ifdef VH_TV_STAMPS
  VH_STAMP(5);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; i++) atomicAdd(&g_pair_stamps[i], st_acc[i]);
  }
#endif
}

}  // namespace

// Tolerance-mode tensor voting (surfaces, exponent 2 or 4).  dtab_fma: the {w, sqrt(2) rhat} table on the device.
int dev_tv_pair(visfd_hip_ctx* ctx, const float* sal, const float* dir, float* ten, const float* mask_src,
                const float* mask_dst, i64 nx, i64 ny, i64 nz, i64 z_out0, i64 z_out1, int h, const float4* dtab_fma,
                int exponent, bool* handled) {
  *handled = false;
  if (exponent != 2 && exponent != 4) return VISFD_HIP_OK;
  if (h < 1 || h > 40) return VISFD_HIP_OK;
  if (nx * ny >= (1LL << 29)) return VISFD_HIP_OK;
  const int n = 2 * h + 1;
  hipStream_t st = ctx->stream;
  PairParams p;
  p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
  p.z_out0 = (int)z_out0; p.z_out1 = (int)z_out1;
  p.h = h;
  p.rw = TX + 2 * h;
  p.rh = TY + 2 * h;
  const int R = p.rw * p.rh;
  p.nchunk = (R + NT - 1) / NT;
  p.rw_magic = ((1 << 20) + p.rw - 1) / p.rw;
  for (int q = 0; q < p.nchunk * NT; q++)
    if ((int)(((unsigned)q * (unsigned)p.rw_magic) >> 20) != q / p.rw) return fail(VISFD_HIP_EINVAL, "tv_pair: region index division");
  // e'x^2 + e'y^2 travels in 16 bits of the position word
  if ((h + TX / 2) * (h + TX / 2) + (h + TY / 2) * (h + TY / 2) >= (1 << 16)) return VISFD_HIP_OK;
  const size_t slice_bytes = sizeof(float4) * (size_t)n * tv_padded_row(h);
  p.tiles_x = (int)((nx + TX - 1) / TX);
  p.tiles_y = (int)((ny + TY - 1) / TY);
  p.relist = ctx->opt.tv_no_replay ? 1 : 0;
  p.sp = tv_padded_row(h);
  p.zrun = 32;
  if (ctx->opt.tv_zrun >= 1 && ctx->opt.tv_zrun <= 4096) p.zrun = ctx->opt.tv_zrun;
  if ((i64)p.zrun > z_out1 - z_out0) p.zrun = (int)(z_out1 - z_out0);
  if (p.zrun < 1) p.zrun = 1;
  const i64 nruns = (z_out1 - z_out0 + p.zrun - 1) / p.zrun;
  const i64 nblk = (i64)p.tiles_x * p.tiles_y * nruns;
  if (nblk > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  if (nblk <= 0) { *handled = true; return VISFD_HIP_OK; }
  const size_t lds = 2 * slice_bytes;
  const size_t lds_static = (sizeof(float4) + sizeof(uint2) + (mask_src ? sizeof(float) : 0)) * LSLOTS + 2560;
  if (lds + lds_static > 150 * 1024) return VISFD_HIP_OK;   // window too wide: the caller falls back
  unsigned* counter = nullptr;
  VH_TRY(ws(ctx, WS_COUNTER, 16, &counter));
  VH_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned), st));
  size_t wg_per_cu = (160 * 1024) / (lds + lds_static);
  if (wg_per_cu > 2048 / NT) wg_per_cu = 2048 / NT;
  if (wg_per_cu < 1) wg_per_cu = 1;
  i64 ngrid = (i64)ctx->num_cus * (i64)wg_per_cu;
  if (ctx->opt.tv_max_wg > 0 && ngrid > ctx->opt.tv_max_wg) ngrid = ctx->opt.tv_max_wg;
  if (ngrid > nblk) ngrid = nblk;
  unsigned char* scratch = nullptr;
  const size_t per_wg = (size_t)(n + 2 * NP - 1) * R * (mask_src ? 24 : 20);
  if ((size_t)ngrid * per_wg > ((size_t)16 << 30)) ngrid = (i64)(((size_t)16 << 30) / per_wg);
  for (; ngrid >= 1; ngrid /= 2) {
    if (ws(ctx, WS_TVSCRATCH, per_wg * (size_t)ngrid, &scratch) == VISFD_HIP_OK) break;
    scratch = nullptr;
    set_error("");
    (void)hipGetLastError();
  }
  if (!scratch) return VISFD_HIP_OK;
#define VH_PAIR_LAUNCH(MSK, MD)                                                                       \
  do {                                                                                               \
    VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tv_pair_kernel<MSK, MD>),              \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
    tv_pair_kernel<MSK, MD><<<dim3((unsigned)ngrid), dim3(NT), lds, st>>>(sal, dir, ten, mask_src,    \
                                                                         mask_dst, dtab_fma, p, counter, \
                                                                         (unsigned)nblk, scratch);   \
  } while (0)
  if (mask_src) { if (exponent == 4) VH_PAIR_LAUNCH(true, 0); else VH_PAIR_LAUNCH(true, 2); }
  else          { if (exponent == 4) VH_PAIR_LAUNCH(false, 0); else VH_PAIR_LAUNCH(false, 2); }
#undef VH_PAIR_LAUNCH
  VH_HIP(hipGetLastError());
#ifdef VH_TV_COUNT
  {
    unsigned long long c4[8], z4[8] = {};
    VH_HIP(hipStreamSynchronize(st));
    VH_HIP(hipMemcpyFromSymbol(c4, HIP_SYMBOL(g_pair_counts), sizeof(c4)));
    fprintf(stderr, "[tv_pair counts] tested (incl. batch padding) %.4g  voted wave-steps %.4g  votes (lane hits) %.4g  -> lane use of a vote step %.3f, "
            "tested / voted %.3f\n", (double)c4[0], (double)c4[1], (double)c4[2], (double)c4[2] / (64.0 * (double)c4[1]), (double)c4[0] / (double)c4[1]);
    fprintf(stderr, "[tv_pair counts] steps with senders %.4g  barrier intervals %.4g (%.3f per step)  list entries per step %.1f (capacity %d)\n",
            (double)c4[4], (double)c4[5], (double)c4[5] / (double)c4[4], (double)c4[6] / (double)c4[4], NLIST * CAPH);
    VH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pair_counts), z4, sizeof(z4)));
  }
#endif
#ifdef VH_TV_IMBAL
  {
    unsigned long long c4[4], z4[4] = {};
    VH_HIP(hipStreamSynchronize(st));
    VH_HIP(hipMemcpyFromSymbol(c4, HIP_SYMBOL(g_pair_imbal), sizeof(c4)));
    fprintf(stderr, "[tv_pair imbalance] intervals %.4g: slowest wave's sweep %.0f ticks on average, mean wave %.0f  -> max / mean %.3f\n",
            (double)c4[2], (double)c4[0] / (double)c4[2], (double)c4[1] / (8.0 * (double)c4[2]), 8.0 * (double)c4[0] / (double)c4[1]);
    VH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pair_imbal), z4, sizeof(z4)));
  }
#endif
#ifdef VH_TV_STAMPS
  {
    unsigned long long st8[8], z8[8] = {};
    VH_HIP(hipStreamSynchronize(st));
    VH_HIP(hipMemcpyFromSymbol(st8, HIP_SYMBOL(g_pair_stamps), sizeof(st8)));
    double tot = 0;
    for (int i = 0; i < 6; i++) tot += (double)st8[i];
    fprintf(stderr, "[tv_pair stamps] share of wave time: listing+claim %.3f | fill %.3f | barrier before sweep %.3f | sweep %.3f | "
            "barrier after sweep %.3f | stores+rest %.3f  (total %.3g ticks over %lld waves)\n", st8[0] / tot, st8[1] / tot,
            st8[2] / tot, st8[3] / tot, st8[4] / tot, st8[5] / tot, tot, (long long)ngrid * NW);
    VH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pair_stamps), z8, sizeof(z8)));
  }
#endif
  *handled = true;
  return VISFD_HIP_OK;
}

}  // namespace vh
