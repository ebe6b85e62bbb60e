// tv_box.hip -- dense stick tensor voting in TOLERANCE MODE (context option tv_fma) for gfx950
// (reference lib/visfd/feature.hpp:1914-2037 and :2217-2384; surfaces with angular exponent 2 or 4) -- and, further down,
// the EXACT form of the same kernel structure (tv_boxx_kernel: bit-identical to tv_tiled.hip, the default exact route for
// surfaces without a source mask).
//
// BASELINE.json's north_star asks for vote tensors within 1e-5 relative, not for the reference's bits.  Giving up the
// bits buys fused multiply-adds (a vote is 19 vector instructions, vote_fma below) and a free order of accumulation.
// Round 3's kernel (tv_pair) used that freedom for MIRROR-PAIRED SENDER PLANES -- the sender planes z + d and z + 1 - d
// (d = 1..h+1) see the receiver planes (z, z+1) through the same two table slices |jz| = d-1 and d, so one barrier
// interval serves both -- and kept the exact kernel's sweep: every wave tests every listed sender against its 64
// receivers (8 x 4 x 2) with a v_dot4 + v_cmp and votes under the execution mask of the lanes it reaches.  That sweep
// used 49 % of its lanes (ball of radius h against an 8 x 4 x 2 patch) at ~27 issue slots per vote step.
//
// This kernel keeps the skeleton (persistent workgroups claiming units from a global counter, packed lists, mirror-paired
// planes), takes a 16 x 32 tile of ONE receiver pair per pass (two lists per step instead of the four of two pairs over an
// 8-wide tile: 224 instead of 358 list entries per step, so a step is one barrier interval; 313 -> 302 ms), lists the
// senders ONCE for the whole launch (tvl_* kernels below: the round-3
// kernels listed every sender plane again in every workgroup that reached it -- 7 region voxels read per receiver column,
// 12-15 % of a wave's time) and replaces the sweep:
//
//   * SUB-PATCHES OF 32 RECEIVERS, TWO SENDER STREAMS PER WAVE.  A wave owns four sub-patches of 4 x 4 x 2 receivers (four rows
//     of each 8-column half of the tile, two sub-patches per half).  Lanes 0-31 and lanes 32-63 hold the SAME 32
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
//   * 6 WAVES PER SIMD, 80 VGPRs: the 24 sums of a lane (2 halves x 2 sub-patches x 6) and the read-ahead
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
constexpr int TX = 16, TY = 4 * NW;    // a workgroup's tile of receivers: 16 x 32 on ONE pair of planes (z, z+1) per pass
constexpr int NH = TX / 8;             // 8-column halves of the tile: a wave owns four rows of each (two sub-patches per half)
constexpr int NLIST = 2;               // lists per interval: the sender plane above (A) and the one below (B) the pair
constexpr int NSUB = 2;                // sub-patches per wave and pair: x 0..3 and x 4..7
constexpr int LSLOTS = NT;             // LDS entry slots of an interval: one per thread
constexpr int HCAP = 36;               // hit entries per (wave, sub-patch, stream): 64 tests per chunk -> <= 32, + null + read-ahead
constexpr int YPAD = 3;                // zero rows above and below a table slice (a 4-row sub-patch overhangs by 3)

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
  int tiles_x, tiles_y;
  int zrun;              // receiver planes per unit of work
  int zl0, nzl;          // sender planes [zl0, zl0 + nzl) are listed (tvl_* kernels below)
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
//
// FOLD (every listed saliency is positive -- the usual case: the ridge score is a square): the saliency is folded into the
// listed normal, n' = a n with a^6 = s (exponent 4; a^4 = s for exponent 2), and the record carries c = 2 a^2 in the
// saliency's place: t' = R.n' = a t, q' = c - t'^2 = a^2 q, m' = t' R - n' = a m, so that w q'^2 m' m'^T = s w q^2 m m^T
// without the multiplication by the saliency: 18 instructions.  (Round 3 measured no gain from this form -- that kernel was
// bound by the latency of a vote's LDS reads; this one issues vector instructions 75 % of the time.)
template <int MODE, bool ZNEG, bool FOLD>
__device__ __forceinline__ void vote_fma(float (&T)[6], const f4v& snd /* sal, n  or  c, n' */, const f4v& tw /* w, R */) {
  const float Rz = ZNEG ? -tw.w : tw.w;    // (a source modifier of the instructions below)
  const float t = __builtin_fmaf(Rz, snd.w, __builtin_fmaf(tw.z, snd.z, tw.y * snd.y));
  const float q = __builtin_fmaf(-t, t, FOLD ? snd.x : 2.0f);
  const float m0 = __builtin_fmaf(t, tw.y, -snd.y);
  const float m1 = __builtin_fmaf(t, tw.z, -snd.z);
  const float m2 = __builtin_fmaf(t, Rz, -snd.w);
  const float sw = FOLD ? tw.x : snd.x * tw.x;
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

template <int MODE, bool ZNEG, bool FOLD, int OFF>
__device__ __forceinline__ void vote_hits(float (&T)[6], unsigned hp, int nst, unsigned r16) {
  u4v h0 = lds_u4(hp + (unsigned)OFF), h1 = lds_u4(hp + (unsigned)(OFF + 16));
  f4v sa = lds_f4(h0.x), ta = lds_f4(r16 - h0.y);
  f4v sb, tb;
  int k = 0;
  for (;;) {   // uniform
    sb = lds_f4(h0.z);
    tb = lds_f4(r16 - h0.w);
    vote_fma<MODE, ZNEG, FOLD>(T, sa, ta);
    if (++k >= nst) break;
    sa = lds_f4(h1.x);
    ta = lds_f4(r16 - h1.y);
    h0 = lds_u4(hp + (unsigned)(OFF + 32));
    vote_fma<MODE, ZNEG, FOLD>(T, sb, tb);
    if (++k >= nst) break;
    sb = lds_f4(h1.z);
    tb = lds_f4(r16 - h1.w);
    vote_fma<MODE, ZNEG, FOLD>(T, sa, ta);
    if (++k >= nst) break;
    sa = lds_f4(h0.x);
    ta = lds_f4(r16 - h0.y);
    h1 = lds_u4(hp + (unsigned)(OFF + 48));
    hp += 32u;
    vote_fma<MODE, ZNEG, FOLD>(T, sb, tb);
    if (++k >= nst) break;
  }
}

template <int MODE, bool FOLD>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(6, 6)))
tv_box_kernel(float* __restrict__ ten, const float* __restrict__ mask_dst,
              const float4* __restrict__ table /* [2h+1] slices of nsl entries: w, sqrt(2) rhat at j, zero padding */,
              BoxParams p, unsigned* __restrict__ tile_counter, unsigned ntiles,
              // the sender lists of tvl_write_kernel: for every listed plane and every 8-column tile column, the salient senders
              // of the columns the tile column can reach, in descending (y, x); rows[(zl (ny+1) + y) ntx + tx] = index of
              // the first entry of the rows below y (0 for y = ny, the list's length... offsets are absolute)
              const float4* __restrict__ lst_ent, const unsigned* __restrict__ lst_pos, const unsigned* __restrict__ lst_rows) {
  // l_ent[e]  float4 {saliency (scaled), n0, n1, n2} of the interval's e-th entry; l_ent[LSLOTS]: the null sender (zeros)
  // l_pos[e]  {region position bytes (ex, ey), byte offset of the sender in a table slice: 16 (ey SP + ex)}
  // l_hit     per wave and sub-patch, the hits of the current chunk of 64 tested entries, dealt alternately to two streams
  __shared__ __attribute__((aligned(16))) float4 l_ent[LSLOTS + 1];
  __shared__ __attribute__((aligned(16))) uint2 l_pos[LSLOTS];
  __shared__ __attribute__((aligned(16))) uint2 l_hit[NW][NSUB][2][HCAP];
  __shared__ unsigned claimed_tile;
  __shared__ unsigned plane_beg[88];         // per sender plane of a pass (slot = plane - (rz - h), < 2h + 2 <= 82): first entry ...
  __shared__ int plane_cnt[88];              // ... and number of entries of this tile's rows in the plane's list
  __shared__ int rho_tab[44];                // floor(sqrt(h^2 - j^2)), j = 0..h: the radius of slice j
  extern __shared__ __attribute__((aligned(16))) unsigned char slices[];   // two table slices: S_j (jz = +j) in slot j & 1; then l_rng

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = p.h;
  // LDS rows of a table slice are SP float4 apart, SP = 4 mod 8 and >= S + 3 (tv_box_row): row offsets of 64 or 192 bytes
  // modulo the 256 bytes of the 64 banks.  ds_read_b128 serves a half wave as two groups of 16 lanes, {0-3, 12-15, 20-27}
  // and the rest (MI355X_MICROARCH.md); the lanes of a half wave are dealt to their 4 x 4 x 2 receivers so that each group
  // is the four rows of ONE receiver plane: four 64-byte segments on different banks, whatever the sender's offset.
  const int SP = p.sp;
  const int nsl = p.nsl;
  const i64 plane = (i64)p.nx * p.ny;
  const i64 nvox = plane * p.nz;
  int slot_has[2] = {-1, -1};                // which slice S_j each LDS slot holds (uniform)
  bool up = false;                           // direction of d for the next pass (flips after every pass)
  float4* const sl4 = reinterpret_cast<float4*>(slices);
  // per sender plane of a pass and wave: the stretch [i0, i1) of the plane's list the wave's rows can reach, i0 | i1 << 16
  unsigned* const l_rng = reinterpret_cast<unsigned*>(slices + 2 * (size_t)p.nsl * sizeof(float4));
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

    // (A wave owns the rows 4 w .. 4 w + 3 of both halves.  Round 3's mirrored row blocks -- the second half's rows dealt to the
    // waves in reverse order, against the skew of the waves' vote counts -- measured no difference in this kernel.)
    // The per-lane constants of a phase are RECOMPUTED from the lane number where the phase starts (the empty asm hides the
    // number's origin from the compiler): hoisted out of the step loop they stay live across the vote loops and are spilled.
    auto fresh_lane = [&]() -> unsigned {
      unsigned ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(ln));
      return ln;
    };

    float TT[NH][NSUB][6];

    // ---- TEST + VOTE: entries [i0, i1) of one list's share of the interval (first LDS slot `base`), 64 at a time.  Lane l
    // tests entry i0 + 64 c + l against the boxes of both sub-patches: with the sender at region position (ex, ey) and a
    // sub-patch's receivers at x in [bx, bx + 3], y in [by, by + 3], the nearest receiver is max(|ex - (bx + 1.5)| - 1.5, 0)
    // columns and as many rows (with by) away; it is reached if dx^2 + dy^2 <= rr = h^2 - (d-1)^2 (the nearer of the two
    // receiver planes is d - 1 planes from the sender plane).  All small integers: exact in float.
    auto test_vote = [&](auto ZN, int base, int i0, int i1, unsigned r16, float cy, float rr, unsigned null_e16) {
      constexpr bool ZNEG = decltype(ZN)::value;
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
        // the two 8-column halves of the tile, one after the other (the hit buffers hold one half's two sub-patches)
        auto half = [&](auto HH) {
          constexpr int hh = decltype(HH)::value;
          const float cx0 = (float)(h + 8 * hh) + 1.5f;
          int nh[NSUB];
#pragma unroll
          for (int s = 0; s < NSUB; s++) {
            const float dx = fmaxf(__builtin_fabsf(exf - (cx0 + 4.0f * (float)s)) - 1.5f, 0.0f);
            const bool hit = __builtin_fmaf(dx, dx, dy2) <= rr;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
            nh[s] = __builtin_popcountll(bal);
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            // (the table entries of sub-patch s of half hh are 8 hh + 4 s columns further: taken off the sender's offset here, so
            // that the vote loops have one table base)
            if (hit) lds_store_u2(hb + (unsigned)(2 * s * HCAP * 8) + (rank & 1u) * (unsigned)(HCAP * 8) + (rank >> 1) * 8u, ent,
                                  pw.y - (unsigned)(128 * hh + 64 * s));
            // the second stream's last step when the count is odd: the null sender (zero saliency, zero normal), placed on the
            // tile's first receiver of this row block so that every lane reads a table entry of the slice (finite; times 0)
            // (its offset for sub-patch (hh, s), less that sub-patch's 128 hh + 64 s, is the same for all four)
            if (ln == 0) lds_store_u2(hb + (unsigned)((2 * s + 1) * HCAP * 8) + (unsigned)(nh[s] >> 1) * 8u, null_ent, null_e16);
          }
#ifdef VH_TV_COUNT
          cnt_hits += (unsigned)(nh[0] + nh[1]);
          cnt_steps += (unsigned)(((nh[0] + 1) >> 1) + ((nh[1] + 1) >> 1));
#endif
          __builtin_amdgcn_wave_barrier();
          asm volatile("" ::: "memory");
          const unsigned hp = hb + (fresh_lane() >> 5) * (unsigned)(HCAP * 8);
          __builtin_amdgcn_s_setprio(1);   // a voting wave is on its workgroup's critical path; waves that fill are not (337 -> 333 ms)
          if (nh[0] > 0) vote_hits<MODE, ZNEG, FOLD, 0>(TT[hh][0], hp, (nh[0] + 1) >> 1, r16);              // (uniform)
          if (nh[1] > 0) vote_hits<MODE, ZNEG, FOLD, 2 * HCAP * 8>(TT[hh][1], hp, (nh[1] + 1) >> 1, r16);
          __builtin_amdgcn_s_setprio(0);
          asm volatile("" ::: "memory");
          __builtin_amdgcn_wave_barrier();
        };
#ifdef VH_TV_COUNT
        cnt_tested += (unsigned)min(64, i1 - c);
#endif
        half(std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 1>{});
      }
    };

    // rows of the sender lists this tile reaches: [y0 - h, y0 + TY + h) clipped to the image
    const int row_lo = max(y0 - h, 0), row_hi = min(y0 + TY - 1 + h, p.ny - 1);
    for (int rz = z_run0; rz < z_run1; rz += 2) {
      // sender planes that reach the LIVE receivers of this pass (a run may end inside a pass: nothing above the last
      // live receiver + h is needed -- or, in a slab run, complete -- then)
      const int sz_hi = min(min(rz + 1, z_run1 - 1) + h, p.nz - 1), sz_lo = max(rz - h, 0);
      // where this tile's rows start in each of those planes' lists and how many entries they have: the lists are in descending
      // row order, lst_rows[.. y ..] = index of the first entry of a row < y (see tvl_scan_kernel)
      __syncthreads();   // every wave is done with the previous pass's ranges (its trailing steps may have had no barrier)
      if (tid < sz_hi - sz_lo + 1) {
        const size_t r0 = ((size_t)(sz_lo + tid - p.zl0) * (size_t)(p.ny + 1)) * (size_t)p.tiles_x + (size_t)tile_x;
        const unsigned beg = lst_rows[r0 + (size_t)(row_hi + 1) * (size_t)p.tiles_x];
        const unsigned end = lst_rows[r0 + (size_t)row_lo * (size_t)p.tiles_x];
        plane_beg[sz_lo + tid - (rz - h)] = beg;
        plane_cnt[sz_lo + tid - (rz - h)] = (int)(end - beg);
      }
      // ... and, per plane and wave, which of those entries the wave's four rows can reach: the lists are in descending row
      // order with per-row offsets, so the stretch [i0, i1) follows from two of them (the waves used to count it from the
      // position words of every interval: ~100 vector instructions, 8 LDS reads and 16 ballots per wave and step)
      if (tid < (sz_hi - sz_lo + 1) * NW) {
        const int pl_i = tid / NW, w = tid - pl_i * NW;
        const int sz = sz_lo + pl_i;
        const int d = sz > rz ? sz - rz : rz + 1 - sz;          // plane A at z + d, plane B at z + 1 - d
        const int rho = rho_tab[d - 1];
        const int yhi = min(y0 + 4 * w + 3 + rho, row_hi), ylo = max(y0 + 4 * w - rho, row_lo);
        unsigned rg = 0u;
        if (yhi >= ylo) {
          const size_t r0 = ((size_t)(sz - p.zl0) * (size_t)(p.ny + 1)) * (size_t)p.tiles_x + (size_t)tile_x;
          const unsigned beg = lst_rows[r0 + (size_t)(row_hi + 1) * (size_t)p.tiles_x];
          const unsigned a0 = lst_rows[r0 + (size_t)(yhi + 1) * (size_t)p.tiles_x] - beg;
          const unsigned a1 = lst_rows[r0 + (size_t)ylo * (size_t)p.tiles_x] - beg;
          rg = a0 | (a1 << 16);
        }
        l_rng[(sz - (rz - h)) * NW + w] = rg;
      }

#pragma unroll
      for (int pp = 0; pp < NH; pp++)
#pragma unroll
        for (int s = 0; s < NSUB; s++)
#pragma unroll
          for (int k = 0; k < 6; k++) TT[pp][s][k] = 0.0f;
      __syncthreads();   // the list ranges of this pass are visible
      VH_STAMP(0);

      // d = 1 .. h+1.  Receiver planes z = rz and z + 1: sender planes A = z + d (above: jz = -d for the
      // lower receiver plane, 1-d for the upper one) and B = z + 1 - d (below: jz = d-1 and d).  All of them need the slices
      // S_(d-1) and S_d; the direction of d alternates from pass to pass, so that every step -- the first of a pass
      // included -- finds one of its two slices in LDS already.
      auto plane_slot = [&](int sz) -> int { return sz - (rz - h); };   // (only used for planes in [sz_lo, sz_hi])
      for (int step = 0; step <= h; step++) {
        const int d = up ? step + 1 : h + 1 - step;
        int lsz[NLIST], lcnt[NLIST];     // list 0: plane A = rz + d (above the pair), list 1: plane B = rz + 1 - d (below)
        lsz[0] = rz + d;
        lsz[1] = rz + 1 - d;
        lcnt[0] = lsz[0] <= sz_hi ? __builtin_amdgcn_readfirstlane(plane_cnt[plane_slot(lsz[0])]) : 0;
        lcnt[1] = lsz[1] >= sz_lo ? __builtin_amdgcn_readfirstlane(plane_cnt[plane_slot(lsz[1])]) : 0;
        const int cmax = max(lcnt[0], lcnt[1]);
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
        const float rr = (float)(h * h - (d - 1) * (d - 1));
        int pre[NLIST + 1];                                    // (uniform) first position of list k in the step's sequence
        pre[0] = 0;
#pragma unroll
        for (int k = 0; k < NLIST; k++) pre[k + 1] = pre[k] + lcnt[k];
        const int total = pre[NLIST];
        unsigned pl[NLIST];                                    // (uniform) first entry of list k in the global lists
#pragma unroll
        for (int k = 0; k < NLIST; k++) pl[k] = lcnt[k] > 0 ? __builtin_amdgcn_readfirstlane(plane_beg[plane_slot(lsz[k])]) : 0u;
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
          unsigned idx = pl[0] + (unsigned)g;
#pragma unroll
          for (int k = 1; k < NLIST; k++)
            if (k_me == k) idx = pl[k] + (unsigned)(g - pre[k]);
          // (unconditional loads from clamped indices: a load inside a branch is closed by a full s_waitcnt at the join, and
          //  the entries' round trip then ended before the slice loads below were even issued)
          const unsigned idxc = have ? idx : 0u;
          const float4 a = lst_ent[idxc];
          const unsigned m = lst_pos[idxc];
          // the new slice(s) of this step: every load is requested BEFORE the first LDS store waits for one (entries and slice
          // share one round trip; a copy loop that loads and stores element by element is three)
          const int ft = wave * 64 + (int)fresh_lane();     // (this thread's index again: addresses hoisted out of the step loop are spilled)
          // Both slots' loads are ALWAYS issued, from clamped indices: a slice the step does not need, and the slice of zeros
          // beyond the window (S_(h+1)), are read from the zero slice that follows the table's 2h+1 -- a float4 selected
          // between a load and a constant makes the compiler park the constant in scratch memory, and loads inside branches
          // are closed by full waits at their joins.
          const bool ld0 = done == 0 && need[0] >= 0 && need[0] <= h, ld1 = done == 0 && need[1] >= 0 && need[1] <= h;   // uniform
          const float4* const srcA = table + (i64)(ld0 ? need[0] + h : 2 * h + 1) * nsl;
          const float4* const srcB = table + (i64)(ld1 ? need[1] + h : 2 * h + 1) * nsl;
          const int f0 = min(ft, nsl - 1), f1 = min(ft + NT, nsl - 1);
          const float4 s00 = srcA[f0], s01 = srcA[f1], s10 = srcB[f0], s11 = srcB[f1];
          if (done == 0) {
            if (need[0] >= 0) {   // uniform
              float4* dst4 = sl4 + (need[0] & 1) * nsl;
              if (ft < nsl) dst4[ft] = s00;
              if (ft + NT < nsl) dst4[ft + NT] = s01;
              for (int i = ft + 2 * NT; i < nsl; i += NT) dst4[i] = srcA[i];   // (windows wider than h = 12)
            }
            if (need[1] >= 0) {
              float4* dst4 = sl4 + (need[1] & 1) * nsl;
              if (ft < nsl) dst4[ft] = s10;
              if (ft + NT < nsl) dst4[ft + NT] = s11;
              for (int i = ft + 2 * NT; i < nsl; i += NT) dst4[i] = srcB[i];
            }
          }
          if (have) {   // list entries carry {column within the tile column's window, image row}: the region row here
            l_ent[tid] = a;
            const unsigned ex = m & 0xffu, ey = (m >> 8) - (unsigned)(y0 - h);
            l_pos[tid] = make_uint2(ex | (ey << 8), 16u * (ey * (unsigned)SP + ex));
          }
          VH_STAMP(1);
          __syncthreads();   // lists (and slices) complete
          VH_STAMP(2);
          // entries are in descending row order: of list k, this wave needs those from the first one at or below region row
          // 4 wv + h + 3 + rho to the last one at or above row 4 wv + h - rho.  Every wave counts both kinds itself, from the
          // row bytes of the position words in LDS, 64 entries at a time.
          int i0[NLIST], i1[NLIST];
#pragma unroll
          for (int k = 0; k < NLIST; k++) {   // the wave's stretch of list k (l_rng, list positions) cut to this interval's share
            i0[k] = i1[k] = 0;
            if (len[k] > 0) {   // uniform
              const unsigned rg = __builtin_amdgcn_readfirstlane(l_rng[plane_slot(lsz[k]) * NW + wave]);
              const int off = c[k] + done - pre[k];     // list position of the share's first LDS slot
              i0[k] = min(max((int)(rg & 0xffffu) - off, 0), len[k]);
              i1[k] = min(max((int)(rg >> 16) - off, 0), len[k]);
            }
          }
          if (i1[0] > i0[0] || i1[1] > i0[1]) {   // uniform
            const float cy = (float)(4 * wave + h) + 1.5f;
            const unsigned ln = fresh_lane();
            const int fq = (int)((ln & 31u) >> 2);
            const int fcol = (int)(ln & 3u), frow = fq >> 1, fpl = (0x96 >> fq) & 1;
            // this lane's table entry of a sender at region position (0, 0), sub-patch 0 of the left half, in slice slot 0:
            // 4 guard entries, then row (jy + h + YPAD), column (jx + h) with jy = 4 w + frow + h - ey, jx = fcol + h - ex
            const unsigned r16_0 = lds_addr(slices) + 16u * (unsigned)(4 + (4 * wave + frow + 2 * h + YPAD) * SP + fcol + 2 * h);
            const unsigned null_e16 = 16u * (unsigned)((4 * wave + h) * SP + h);
            // plane A (above): the lower receiver plane sees it at jz = -d (slice S_d, rhat_z negated), the upper one at 1-d
            if (i1[0] > i0[0]) {
              const int js = fpl ? d - 1 : d;
              test_vote(std::true_type{}, c[0], i0[0], i1[0], r16_0 + (unsigned)(16 * nsl) * (unsigned)(js & 1), cy, rr, null_e16);
            }
            // plane B (below): jz = d-1 for the lower plane (S_(d-1)), d for the upper one (S_d)
            if (i1[1] > i0[1]) {
              const int js = fpl ? d : d - 1;
              test_vote(std::false_type{}, c[1], i0[1], i1[1], r16_0 + (unsigned)(16 * nsl) * (unsigned)(js & 1), cy, rr, null_e16);
            }
          }
          VH_STAMP(3);
          __syncthreads();   // everyone done reading before the lists or the slices are refilled
          VH_STAMP(4);
        }
      }
      up = !up;

      // ---- the pass's sums: a receiver's two streams are added; lanes 0-31 store sub-patch 0, lanes 32-63 sub-patch 1 ----
#pragma unroll
      for (int pp = 0; pp < NH; pp++) {
        const int rx = x0 + 8 * pp + 4 * strm + lcol, ry = y0 + 4 * wave + lrow, rzl = rz + lpl;
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

// =================================================================================================================
// The EXACT form of the same kernel structure (round 4): the reference's 32 multiplies and adds per vote in its order
// (feature.hpp:2312-2377, no FMA), every receiver taking its votes in the reference's order (sender planes z+h .. z-h, rows and
// columns descending), bit-identical to csrc/tv_tiled.hip -- which stays the general exact kernel (source masks, curve mode,
// odd exponents, weight sums, non-finite saliencies, WEIGHTED source masks) -- for the common case: surfaces, exponent 2 or
// 4, no source mask or one of zeros and ones.
// What carries over from the tolerance kernel: the launch-wide sender lists (already in vote order), the box-tested hit lists,
// the zero-padded slices (a zero-weight tap votes (s * 0) * dec = +-0 times finite numbers: adding it leaves a sum's bits,
// as the rim taps of tv_tiled's superset test already do), the branch-free pipelined vote loop.  What cannot: the order
// of accumulation is fixed, so
//   * a receiver has ONE stream: lanes 0-31 are the sub-patch x 0..3 and lanes 32-63 the sub-patch x 4..7 of a half, each
//     with its own hit list; a step serves the k-th hit of both (the shorter list is padded with null senders);
//   * sender planes come one at a time with jz ascending (2h+2 steps per receiver pair, no mirror pairing), each needing the
//     slices jz and jz+1 -- one new slice per step; TWO receiver pairs (z, z+1) and (z+2, z+3) share a step's two slices
//     (their sender planes differ by two), which gives a step two lists, as in the tolerance kernel.
constexpr int NPAIR = 2;
constexpr int HX = 72;    // hit entries per (wave, sub-patch): <= 64 hits of a chunk + the read-ahead of the vote loop

// The six sums of (pair Q, half PP) are PINNED to the vector registers v[56 + 12 Q + 6 PP ...] in every add (a physical-
// register constraint): the four inlined copies of the vote loop then agree on where the 24 sums live, and the register
// allocator has nothing to shuffle at the loop boundaries (it moved 20 of them through scratch memory around every step:
// 650 GB of spill traffic per launch at 1024^3).
#define VH_PIN_ADD(REG, var, val) asm("v_add_f32 " #REG ", " #REG ", %1" : "+{" #REG "}"(var) : "v"(val))
template <int BASE>
__device__ __forceinline__ void acc6_pinned(float (&T)[6], float p00, float p01, float p02, float p11, float p12, float p22) {
#define VH_PIN6(B, R0, R1, R2, R3, R4, R5)                                                                              \
  if constexpr (BASE == B) {                                                                                            \
    VH_PIN_ADD(R0, T[0], p00); VH_PIN_ADD(R3, T[3], p01); VH_PIN_ADD(R5, T[5], p02);                                    \
    VH_PIN_ADD(R1, T[1], p11); VH_PIN_ADD(R4, T[4], p12); VH_PIN_ADD(R2, T[2], p22);                                    \
  }
  VH_PIN6(56, v56, v57, v58, v59, v60, v61)
  VH_PIN6(62, v62, v63, v64, v65, v66, v67)
  VH_PIN6(68, v68, v69, v70, v71, v72, v73)
  VH_PIN6(74, v74, v75, v76, v77, v78, v79)
#undef VH_PIN6
}

template <int MODE, int BASE>
__device__ __forceinline__ void vote_exact(float (&T)[6], const f4v& snd /* sal, n */, const f4v& tw /* w, rhat */) {
  const float u = (tw.y * snd.y + tw.z * snd.z) + tw.w * snd.w;
  const float ux2 = u * 2.0f;
  const float u2 = u * u;
  const float c2 = 1.0f - u2;
  const float dec = (MODE == 0) ? c2 * c2 : c2;
  const float m0 = ux2 * tw.y - snd.y, m1 = ux2 * tw.z - snd.z, m2 = ux2 * tw.w - snd.w;
  const float bse = (snd.x * tw.x) * dec;
  const float b0 = bse * m0, b1 = bse * m1, b2 = bse * m2;
  const float p00 = b0 * m0, p01 = b0 * m1, p02 = b0 * m2, p11 = b1 * m1, p12 = b1 * m2, p22 = b2 * m2;
  acc6_pinned<BASE>(T, p00, p01, p02, p11, p12, p22);
}

// (the vote loop of vote_hits with the exact vote; one list per half wave, entries at hp)
template <int MODE, int BASE>
__device__ __forceinline__ void vote_hits_exact(float (&T)[6], unsigned hp, int nst, unsigned r16) {
  u4v h0 = lds_u4(hp), h1 = lds_u4(hp + 16u);
  f4v sa = lds_f4(h0.x), ta = lds_f4(r16 - h0.y);
  f4v sb, tb;
  int k = 0;
  for (;;) {   // uniform
    sb = lds_f4(h0.z);
    tb = lds_f4(r16 - h0.w);
    vote_exact<MODE, BASE>(T, sa, ta);
    if (++k >= nst) break;
    sa = lds_f4(h1.x);
    ta = lds_f4(r16 - h1.y);
    h0 = lds_u4(hp + 32u);
    vote_exact<MODE, BASE>(T, sb, tb);
    if (++k >= nst) break;
    sb = lds_f4(h1.z);
    tb = lds_f4(r16 - h1.w);
    vote_exact<MODE, BASE>(T, sa, ta);
    if (++k >= nst) break;
    sa = lds_f4(h0.x);
    ta = lds_f4(r16 - h0.y);
    h1 = lds_u4(hp + 48u);
    hp += 32u;
    vote_exact<MODE, BASE>(T, sb, tb);
    if (++k >= nst) break;
  }
}

template <int MODE>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(6, 6)))
tv_boxx_kernel(float* __restrict__ ten, const float* __restrict__ mask_dst,
               const float4* __restrict__ table /* [2h+1] slices of nsl entries: w, rhat at j, zero padding */,
               BoxParams p, unsigned* __restrict__ tile_counter, unsigned ntiles,
               const float4* __restrict__ lst_ent, const unsigned* __restrict__ lst_pos, const unsigned* __restrict__ lst_rows) {
  __shared__ __attribute__((aligned(16))) float4 l_ent[LSLOTS + 1];
  __shared__ __attribute__((aligned(16))) uint2 l_pos[LSLOTS];
  __shared__ __attribute__((aligned(16))) uint2 l_hit[NW][NSUB][HX];
  __shared__ unsigned claimed_tile;
  __shared__ unsigned plane_beg[88];         // per sender plane of a pass (slot = plane - (rz - h), < 2h + 4 <= 84)
  __shared__ int plane_cnt[88];
  __shared__ int rho_tab[44];
  extern __shared__ __attribute__((aligned(16))) unsigned char slices[];   // two table slices: jz in slot (jz + h + 1) & 1

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = p.h;
  const int SP = p.sp;
  const int nsl = p.nsl;
  const i64 plane = (i64)p.nx * p.ny;
  const i64 nvox = plane * p.nz;
  int slot_has[2] = {-1, -1};                // which slice (as jz + h + 1 = 0 .. 2h+2; 0 and 2h+2: zeros) each LDS slot holds
  float4* const sl4 = reinterpret_cast<float4*>(slices);
  unsigned* const l_rng = reinterpret_cast<unsigned*>(slices + 2 * (size_t)p.nsl * sizeof(float4));   // [plane][pair][wave]: i0 | i1 << 16
  const unsigned ent_base = lds_addr(l_ent);
  const unsigned null_ent = ent_base + 16u * (unsigned)LSLOTS;

#ifdef VH_TV_STAMPS
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
#ifdef VH_TV_COUNT
  unsigned cnt_tested = 0, cnt_hits = 0, cnt_steps = 0;
#endif
  // lane -> receiver: sub-patch = lane >> 5 (x 0..3 or 4..7 of a half); within it as in the tolerance kernel
  const int sub = lane >> 5;
  const int qd = (lane & 31) >> 2;
  const int lcol = lane & 3, lrow = qd >> 1, lpl = (0x96 >> qd) & 1;

  if (tid == 0) l_ent[LSLOTS] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (tid <= h) {
    const int r2 = h * h - tid * tid;
    int rho = (int)__builtin_sqrtf((float)r2);
    while (rho * rho > r2) rho--;
    while ((rho + 1) * (rho + 1) <= r2) rho++;
    rho_tab[tid] = rho;
  }
  for (int i = tid; i < NW * NSUB * HX; i += NT) (&l_hit[0][0][0])[i] = make_uint2(null_ent, 0u);

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

    auto fresh_lane = [&]() -> unsigned {
      unsigned ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(ln));
      return ln;
    };

    float TT[NPAIR][NH][6];

    // TEST + VOTE of entries [i0, i1) of one list (LDS slots from `base`) for the receiver pair whose sums are T2: 64 entries
    // at a time, each tested against the boxes of the two sub-patches of a half; the hits of sub-patch s go, in list order, to
    // l_hit[wave][s][0..]; lanes 32 s .. 32 s + 31 then vote them in that order.
    auto test_vote = [&](auto QQ, int base, int i0, int i1, unsigned r16, float cy, float rr, unsigned null_e16) {
      constexpr int qq = decltype(QQ)::value;
      float (&T2)[NH][6] = TT[qq];
      const unsigned hb = lds_addr(&l_hit[wave][0][0]);
      int c = i0;
      do {   // uniform; i1 > i0 at every call
        const int ln = (int)fresh_lane();
        const int e = c + ln;
        uint2 pw = make_uint2(0xffffffffu, 0u);
        if (e < i1) pw = l_pos[base + e];
        const float exf = (float)(pw.x & 0xffu), eyf = (float)((pw.x >> 8) & 0xffu);
        const float dy = fmaxf(__builtin_fabsf(eyf - cy) - 1.5f, 0.0f);
        const float dy2 = dy * dy;
        const unsigned ent = ent_base + 16u * (unsigned)(base + e);
        auto half = [&](auto HH) {
          constexpr int hh = decltype(HH)::value;
          const float cx0 = (float)(h + 8 * hh) + 1.5f;
          int nh[NSUB];
#pragma unroll
          for (int s = 0; s < NSUB; s++) {
            const float dx = fmaxf(__builtin_fabsf(exf - (cx0 + 4.0f * (float)s)) - 1.5f, 0.0f);
            const bool hit = __builtin_fmaf(dx, dx, dy2) <= rr;     // (small integers and halves: exact either way)
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
            nh[s] = __builtin_popcountll(bal);
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            // every slot a step may vote holds the null sender first (the two lists of a half are voted to the longer one's
            // length); the hits then overwrite their slots: LDS operations of a wave execute in order
            lds_store_u2(hb + (unsigned)(s * HX * 8) + (unsigned)ln * 8u, null_ent, null_e16);
            if (hit) lds_store_u2(hb + (unsigned)(s * HX * 8) + rank * 8u, ent, pw.y - (unsigned)(128 * hh));
          }
          __builtin_amdgcn_wave_barrier();
          asm volatile("" ::: "memory");
          const int nst = max(nh[0], nh[1]);
#ifdef VH_TV_COUNT
          cnt_hits += (unsigned)(nh[0] + nh[1]);
          cnt_steps += (unsigned)nst;
#endif
          if (nst > 0) {   // uniform
            const unsigned hp = hb + (fresh_lane() >> 5) * (unsigned)(HX * 8);
            __builtin_amdgcn_s_setprio(1);
            vote_hits_exact<MODE, 56 + 12 * qq + 6 * hh>(T2[hh], hp, nst, r16);
            __builtin_amdgcn_s_setprio(0);
          }
          asm volatile("" ::: "memory");
          __builtin_amdgcn_wave_barrier();
        };
#ifdef VH_TV_COUNT
        cnt_tested += (unsigned)min(64, i1 - c);
#endif
        half(std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 1>{});
        c += 64;
      } while (c < i1);
    };

    const int row_lo = max(y0 - h, 0), row_hi = min(y0 + TY - 1 + h, p.ny - 1);
    for (int rz = z_run0; rz < z_run1; rz += 2 * NPAIR) {
      // sender planes that reach the LIVE receivers of this pass
      const int sz_hi = min(min(rz + 2 * NPAIR - 1, z_run1 - 1) + h, p.nz - 1), sz_lo = max(rz - h, 0);
      __syncthreads();
      if (tid < sz_hi - sz_lo + 1) {
        const size_t r0 = ((size_t)(sz_lo + tid - p.zl0) * (size_t)(p.ny + 1)) * (size_t)p.tiles_x + (size_t)tile_x;
        const unsigned beg = lst_rows[r0 + (size_t)(row_hi + 1) * (size_t)p.tiles_x];
        const unsigned end = lst_rows[r0 + (size_t)row_lo * (size_t)p.tiles_x];
        plane_beg[sz_lo + tid - (rz - h)] = beg;
        plane_cnt[sz_lo + tid - (rz - h)] = (int)(end - beg);
      }
      // per plane, receiver pair and wave: the stretch of the plane's list the wave's rows can reach (see the tolerance kernel)
      if (tid < (sz_hi - sz_lo + 1) * NW * NPAIR) {
        const int pl_i = tid / (NW * NPAIR), rem = tid - pl_i * (NW * NPAIR);
        const int q = rem / NW, w = rem - q * NW;
        const int sz = sz_lo + pl_i;
        const int t = rz + 2 * q + 1 + h - sz;                  // the step at which pair q meets this plane
        unsigned rg = 0u;
        if (t >= 0 && t <= 2 * h + 1) {
          const int dmin = min(abs(t - h - 1), abs(t - h));
          const int rho = rho_tab[dmin];
          const int yhi = min(y0 + 4 * w + 3 + rho, row_hi), ylo = max(y0 + 4 * w - rho, row_lo);
          if (yhi >= ylo) {
            const size_t r0 = ((size_t)(sz - p.zl0) * (size_t)(p.ny + 1)) * (size_t)p.tiles_x + (size_t)tile_x;
            const unsigned beg = lst_rows[r0 + (size_t)(row_hi + 1) * (size_t)p.tiles_x];
            const unsigned a0 = lst_rows[r0 + (size_t)(yhi + 1) * (size_t)p.tiles_x] - beg;
            const unsigned a1 = lst_rows[r0 + (size_t)ylo * (size_t)p.tiles_x] - beg;
            rg = a0 | (a1 << 16);
          }
        }
        l_rng[((sz - (rz - h)) * NPAIR + q) * NW + w] = rg;
      }
#pragma unroll
      for (int q = 0; q < NPAIR; q++)
#pragma unroll
        for (int pp = 0; pp < NH; pp++)
#pragma unroll
          for (int k = 0; k < 6; k++) TT[q][pp][k] = 0.0f;
      __syncthreads();
      VH_STAMP(0);

      // step t = 0 .. 2h+1: pair q (receiver planes rz + 2q, rz + 2q + 1) meets sender plane rz + 2q + 1 + h - t at
      // jz = t - h - 1 (lower plane) and t - h (upper plane), ascending as in the reference.  Slice key = jz + h + 1.
      auto plane_slot = [&](int sz) -> int { return sz - (rz - h); };
      for (int t = 0; t <= 2 * h + 1; t++) {
        int lsz[NLIST], lcnt[NLIST];
#pragma unroll
        for (int q = 0; q < NPAIR; q++) {
          lsz[q] = rz + 2 * q + 1 + h - t;
          lcnt[q] = (lsz[q] <= sz_hi && lsz[q] >= sz_lo) ? __builtin_amdgcn_readfirstlane(plane_cnt[plane_slot(lsz[q])]) : 0;
        }
        // (a step without entries -- sender planes outside the volume -- still runs its interval: skipping it would give the
        //  24 sums a second path through the step, which the register allocator pays for with copies through scratch memory)
        int need[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const int j = t + k;             // key of the lower (k = 0) / upper (k = 1) receiver plane's slice
          need[k] = (slot_has[j & 1] != j) ? j : -1;
          if (need[k] >= 0) slot_has[j & 1] = j;
        }
        // the nearer of a pair's two receiver planes: |t - h - 1| or |t - h| planes away
        const int dmin = min(abs(t - h - 1), abs(t - h));
        const float rr = (float)(h * h - dmin * dmin);
        int pre[NLIST + 1];
        pre[0] = 0;
#pragma unroll
        for (int k = 0; k < NLIST; k++) pre[k + 1] = pre[k] + lcnt[k];
        const int total = pre[NLIST];
        unsigned pl[NLIST];
#pragma unroll
        for (int k = 0; k < NLIST; k++) pl[k] = lcnt[k] > 0 ? __builtin_amdgcn_readfirstlane(plane_beg[plane_slot(lsz[k])]) : 0u;
        int done = 0;
        do {   // uniform; at least one interval (a zero-trip path would make the 24 sums a merge of two register sets)
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
          unsigned idx = pl[0] + (unsigned)g;
#pragma unroll
          for (int k = 1; k < NLIST; k++)
            if (k_me == k) idx = pl[k] + (unsigned)(g - pre[k]);
          const unsigned idxc = have ? idx : 0u;   // (unconditional loads from a clamped index: see the tolerance kernel)
          const float4 a = lst_ent[idxc];
          const unsigned m = lst_pos[idxc];
          const int ft = wave * 64 + (int)fresh_lane();
          // A step normally needs ONE new slice (the upper plane's; the lower plane's was the previous step's upper one): that
          // one shares the entries' round trip.  The first step of a pass -- and a step behind skipped ones -- needs both: the
          // second is then copied by a plain loop with a round trip of its own (holding two slices' loads in registers beside
          // the 24 sums spilled those sums around every step).  Keys 0 and 2h+2 are the slices beyond the window: zeros.
          const int nk = done == 0 ? (need[1] >= 0 ? need[1] : need[0]) : -1;   // uniform
          const int nk2 = (done == 0 && need[1] >= 0) ? need[0] : -1;
          // (the slices of zeros beyond the window, and the load of a step that needs no slice, read the zero slice that follows
          //  the table's 2h+1: always a load from a clamped index, never a branch or a select against a constant)
          const bool ld = nk >= 1 && nk <= 2 * h + 1;
          const float4* const src4 = table + (i64)(ld ? nk - 1 : 2 * h + 1) * nsl;
          const float4 s0 = src4[min(ft, nsl - 1)], s1 = src4[min(ft + NT, nsl - 1)];
          if (nk >= 0) {
            float4* dst4 = sl4 + (nk & 1) * nsl;
            if (ft < nsl) dst4[ft] = s0;
            if (ft + NT < nsl) dst4[ft + NT] = s1;
            for (int i = ft + 2 * NT; i < nsl; i += NT) dst4[i] = src4[i];   // (windows wider than h = 12)
          }
          if (nk2 >= 0) {
            const bool ld2 = nk2 >= 1 && nk2 <= 2 * h + 1;
            float4* dst4 = sl4 + (nk2 & 1) * nsl;
            const float4* src2 = table + (i64)(ld2 ? nk2 - 1 : 2 * h + 1) * nsl;
            for (int i = ft; i < nsl; i += NT) dst4[i] = src2[i];
          }
          if (have) {
            l_ent[tid] = a;
            const unsigned ex = m & 0xffu, ey = (m >> 8) - (unsigned)(y0 - h);
            l_pos[tid] = make_uint2(ex | (ey << 8), 16u * (ey * (unsigned)SP + ex));
          }
          VH_STAMP(1);
          __syncthreads();   // lists (and slices) complete
          VH_STAMP(2);
          int i0[NLIST], i1[NLIST];
#pragma unroll
          for (int k = 0; k < NLIST; k++) {   // (list k = pair k) the wave's stretch cut to this interval's share of the list
            i0[k] = i1[k] = 0;
            if (len[k] > 0) {   // uniform
              const unsigned rg = __builtin_amdgcn_readfirstlane(l_rng[(plane_slot(lsz[k]) * NPAIR + k) * NW + wave]);
              const int off = c[k] + done - pre[k];
              i0[k] = min(max((int)(rg & 0xffffu) - off, 0), len[k]);
              i1[k] = min(max((int)(rg >> 16) - off, 0), len[k]);
            }
          }
          if (i1[0] > i0[0] || i1[1] > i0[1]) {   // uniform
            const float cy = (float)(4 * wave + h) + 1.5f;
            const unsigned ln = fresh_lane();
            const int fq = (int)((ln & 31u) >> 2);
            const int fcol = (int)(ln & 3u) + 4 * (int)(ln >> 5), frow = fq >> 1, fpl = (0x96 >> fq) & 1;
            // this lane's table entry of a sender at region position (0, 0), left half (its sub-patch's 4 columns are in fcol)
            const unsigned r16_0 = lds_addr(slices) + 16u * (unsigned)(4 + (4 * wave + frow + 2 * h + YPAD) * SP + fcol + 2 * h);
            const unsigned null_e16 = 16u * (unsigned)((4 * wave + h) * SP + h);
            const unsigned r16 = r16_0 + (unsigned)(16 * nsl) * (unsigned)((t + fpl) & 1);
            if (i1[0] > i0[0]) test_vote(std::integral_constant<int, 0>{}, c[0], i0[0], i1[0], r16, cy, rr, null_e16);
            if (i1[1] > i0[1]) test_vote(std::integral_constant<int, 1>{}, c[1], i0[1], i1[1], r16, cy, rr, null_e16);
          }
          VH_STAMP(3);
          __syncthreads();   // everyone done reading before the lists or the slices are refilled
          VH_STAMP(4);
          done += NT;
        } while (done < total);
      }

      // ---- the pass's sums ----
#pragma unroll
      for (int q = 0; q < NPAIR; q++)
#pragma unroll
        for (int pp = 0; pp < NH; pp++) {
          const int rx = x0 + 8 * pp + 4 * sub + lcol, ry = y0 + 4 * wave + lrow, rzl = rz + 2 * q + lpl;
          const bool in = rx < p.nx && ry < p.ny && rzl < z_run1;
          const i64 rc = (i64)rzl * plane + (i64)ry * p.nx + rx;
          const bool live = in && !(mask_dst && mask_dst[in ? rc : 0] == 0.0f);
          if (live) {
#pragma unroll
            for (int k = 0; k < 6; k++) __builtin_nontemporal_store(TT[q][pp][k], &ten[k * nvox + rc]);
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

// ---- THE SENDER LISTS, once per launch ---------------------------------------------------------------------------------
// For every listed plane z and every tile column tx (TX = 16 receiver columns), the salient, unmasked senders of the columns
// [TX tx - h, TX tx + TX + h) -- everything a tile of that column can reach in x -- as one list in DESCENDING (y, x) (the order
// the vote kernel's row-range culling needs), 20 bytes per entry: float4 {saliency * 1/4 or 1/2 (* mask value), normal} and one
// word {x - (TX tx - h), y << 8}.  A sender appears in the lists of the tile columns that reach it (2.5 on average at h = 12).
// rows[(zl (ny + 1) + y) ntx + tx]: index (into the global entry arrays) of the first entry of list (zl, tx) with a row
// below y -- so the entries of the rows [ylo, yhi] are [rows[.. yhi + 1 ..], rows[.. ylo ..]).
// Three kernels: count per (plane, row, tile column); suffix sums per (plane, tile column) with one atomic add per list for
// its place in the global arrays; write.  A WAVE takes one image row: its salient flags as a bit mask in LDS, every
// window's count / every sender's place in its windows by popcounts over at most four words.
struct ListGeo {
  int nx, ny, nz;
  int zl0, nzl;   // listed planes [zl0, zl0 + nzl)
  int ntx, h;
};
constexpr int LNT = 256;
constexpr int LWORDS_MAX = 512;    // nx <= 16384

__device__ __forceinline__ unsigned popc_range(const unsigned* w, int lo, int hi) {   // set bits of [lo, hi), hi - lo <= 96
  unsigned c = 0;
  for (int i = lo >> 5; i <= (hi - 1) >> 5 && lo < hi; i++) {
    unsigned m = w[i];
    if (i == (lo >> 5)) m &= ~0u << (lo & 31);
    if (i == ((hi - 1) >> 5) && (hi & 31)) m &= ~0u >> (32 - (hi & 31));
    c += (unsigned)__builtin_popcount(m);
  }
  return c;
}

template <bool WRITE, int MODE>
__global__ void __launch_bounds__(LNT)
tvl_row_kernel(const float* __restrict__ sal, const float* __restrict__ dir, const float* __restrict__ mask_src, ListGeo g,
               unsigned* __restrict__ rows, float4* __restrict__ ent, unsigned* __restrict__ pos,
               unsigned* __restrict__ neg_flag /* count pass: set if a listed saliency (times its mask value) is not positive */,
               int fold /* write pass: records {c, a n} instead of {s, n} (vote_fma) */) {
  // ONE WAVE PER IMAGE ROW (no workgroup barrier: a wave's bit mask is its own): its salient flags as a bit mask in LDS,
  // 64 voxels per ballot.  A wave is a chain of memory round trips, so every phase requests LB chunks' worth of loads before
  // it uses the first (the loops are otherwise one round trip per 64 voxels: 8 ms for the write pass at 1024^3)
  constexpr int LB = 8;
  __shared__ unsigned bits_all[LNT / 64][LWORDS_MAX + 4];
  __shared__ unsigned base_all[WRITE ? LNT / 64 : 1][WRITE ? LWORDS_MAX * 2 + 4 : 1];   // write pass: the row's first entry per list
  __shared__ unsigned cum_all[WRITE ? LNT / 64 : 1][WRITE ? LWORDS_MAX / 2 + 4 : 1];    // write pass: salient voxels before chunk c
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  const i64 r = (i64)blockIdx.x * (LNT / 64) + wave;      // row number among the listed rows
  if (r >= (i64)g.nzl * g.ny) return;                      // (uniform per wave)
  unsigned* const bits = bits_all[wave];
  const int y = (int)(r % g.ny), zl = (int)(r / g.ny);
  const i64 plane = (i64)g.nx * g.ny, nvox = plane * g.nz;
  const i64 row = (i64)(g.zl0 + zl) * plane + (i64)y * g.nx;
  const int nchunks = (g.nx + 63) >> 6;
  unsigned* const rrow = rows + ((size_t)zl * (size_t)(g.ny + 1) + (size_t)y) * (size_t)g.ntx;
  if (WRITE) {   // (requested first: used after the flags)
    unsigned* const base = base_all[WRITE ? wave : 0];
    for (int tx = lane; tx < g.ntx; tx += 64) base[tx] = rrow[g.ntx + tx];
  }
  unsigned any = 0, total = 0;
  unsigned* const cum = cum_all[WRITE ? wave : 0];
  for (int c0 = 0; c0 < nchunks; c0 += LB) {   // uniform
    float v[LB], m[LB];
#pragma unroll
    for (int k = 0; k < LB; k++) {
      const int x = 64 * (c0 + k) + lane;
      v[k] = x < g.nx ? sal[row + x] : 0.0f;
      m[k] = (mask_src && x < g.nx) ? mask_src[row + x] : 1.0f;
    }
#pragma unroll
    for (int k = 0; k < LB; k++) {
      if (c0 + k >= nchunks) break;   // uniform
      const bool f = v[k] != 0.0f && m[k] != 0.0f;
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(f);
      if (lane == 0) {
        bits[2 * (c0 + k)] = (unsigned)bal;
        bits[2 * (c0 + k) + 1] = (unsigned)(bal >> 32);
        if (WRITE) cum[c0 + k] = total;
      }
      any |= (unsigned)bal | (unsigned)(bal >> 32);
      if (WRITE) total += (unsigned)__builtin_popcountll(bal);
      if (!WRITE && f) {
        const float s = mask_src ? v[k] * m[k] : v[k];
        if (!(s > 0.0f)) atomicOr(neg_flag, 1u);   // (rare: negative peak heights, masks with negative values, NaN)
        if (!(__builtin_fabsf(s) <= 3.402823466e38f)) atomicOr(neg_flag, 2u);   // non-finite: the exact form declines
        if (mask_src && m[k] != 1.0f) atomicOr(neg_flag, 4u);   // a weighted source mask: its value is a factor of the exact vote
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (!WRITE) {
    for (int tx = lane; tx < g.ntx; tx += 64)
      rrow[tx] = any ? popc_range(bits, max(TX * tx - g.h, 0), min(TX * tx + TX + g.h, g.nx)) : 0u;
    return;
  }
  if (!any) return;   // (uniform)
  const unsigned* const base = base_all[WRITE ? wave : 0];
  // lane j takes the row's j-th salient voxel (all lanes busy: the fold's double-precision roots at 3 live lanes per chunk
  // were most of this pass)
  for (unsigned j0 = 0; j0 < total; j0 += 64) {   // uniform
    const unsigned j = j0 + lane;
    if (j >= total) break;
    int lo = 0, hi = nchunks;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (cum[mid] <= j) lo = mid; else hi = mid;
    }
    unsigned k = j - cum[lo];
    unsigned w = bits[2 * lo];
    int x = 64 * lo;
    {
      const unsigned t = (unsigned)__builtin_popcount(w);
      if (k >= t) { k -= t; x += 32; w = bits[2 * lo + 1]; }
    }
#pragma unroll
    for (int sft = 16; sft >= 1; sft >>= 1) {
      const unsigned t = (unsigned)__builtin_popcount(w & ((1u << sft) - 1u));
      if (k >= t) { k -= t; x += sft; w >>= sft; }
    }
    // (MODE 1: the exact form's lists carry the saliency itself)
    float4 q = make_float4(sal[row + x] * (MODE == 0 ? 0.25f : (MODE == 2 ? 0.5f : 1.0f)), dir[row + x], dir[nvox + row + x], dir[2 * nvox + row + x]);
    if (mask_src) q.x = q.x * mask_src[row + x];
    if (fold) {   // a = s^(1/6) (exponent 4) or s^(1/4) (exponent 2), rounded once from double
      const double r2 = sqrt((double)q.x);
      const float a = (float)(MODE == 0 ? cbrt(r2) : sqrt(r2));
      q = make_float4(2.0f * a * a, a * q.y, a * q.z, a * q.w);
    }
    // tile columns whose window holds x: TX tx - h <= x < TX tx + TX + h
    const int t0 = max((x - TX - g.h) / TX + ((x - TX - g.h) >= 0 ? 1 : 0), 0);
    const int t1 = min((x + g.h) / TX, g.ntx - 1);
    for (int tx = t0; tx <= t1; tx++) {
      const int lo_x = TX * tx - g.h, hi_x = min(TX * tx + TX + g.h, g.nx);
      if (x < lo_x || x >= hi_x) continue;
      // rows[.. y + 1 ..] = first entry of the rows below y + 1 = first entry of row y; within the row: descending x
      const unsigned idx = base[tx] + popc_range(bits, x + 1, hi_x);
      ent[idx] = q;
      pos[idx] = (unsigned)(x - lo_x) | ((unsigned)y << 8);
    }
  }
}

// one thread per list (zl, tx): counts -> offsets.  Before: rows[zl][y][tx] = entries of row y (y < ny).  After:
// rows[zl][y][tx] = base + (entries of the rows >= y): the index behind row y's last entry... see the header comment; the
// list's place `base` in the global arrays comes from one atomic add (the lists' order in memory does not matter).
__global__ void __launch_bounds__(LNT)
tvl_scan_kernel(ListGeo g, unsigned* __restrict__ rows, unsigned long long* __restrict__ total) {
  const int i = blockIdx.x * LNT + threadIdx.x;
  if (i >= g.nzl * g.ntx) return;
  const int zl = i / g.ntx, tx = i - zl * g.ntx;
  unsigned* const col = rows + (size_t)zl * (size_t)(g.ny + 1) * (size_t)g.ntx + tx;
  unsigned sum = 0;
  for (int y = 0; y < g.ny; y++) sum += col[(size_t)y * g.ntx];
  const unsigned base = (unsigned)atomicAdd(total, (unsigned long long)sum);
  // descending rows: the entries of row y sit behind those of every row above it
  unsigned run = base;
  unsigned prev = col[(size_t)(g.ny - 1) * g.ntx];
  col[(size_t)g.ny * g.ntx] = run;            // rows below ny: the list's first entry
  for (int y = g.ny - 1; y >= 0; y--) {
    const unsigned c = prev;
    if (y > 0) prev = col[(size_t)(y - 1) * g.ntx];
    run += c;
    col[(size_t)y * g.ntx] = run;              // first entry of a row below y = behind row y's entries
  }
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
// exact != 0: the exact form (tv_boxx_kernel) with dtab_box = the reference's {w, rhat} in the same slice layout; it declines
// weighted source masks (values other than 0 and 1) and non-finite saliencies (the caller falls back to tv_tiled.hip).
int dev_tv_box(visfd_hip_ctx* ctx, const float* sal, const float* dir, float* ten, const float* mask_src,
               const float* mask_dst, i64 nx, i64 ny, i64 nz, i64 z_out0, i64 z_out1, int h, const float4* dtab_box,
               int exponent, bool* handled, bool exact) {
  *handled = false;
  if (exponent != 2 && exponent != 4) return VISFD_HIP_OK;
  // (a source mask of zeros and ones only multiplies the kept senders' weights by 1.0 -- exactly nothing -- so the exact form
  //  takes it; the count pass of the listing reports any other mask value, see below)
  if (h < 1 || h > 40) return VISFD_HIP_OK;
  if (nx * ny >= (1LL << 29) || nx > 32 * LWORDS_MAX || ny >= (1 << 24)) return VISFD_HIP_OK;
  hipStream_t st = ctx->stream;
  BoxParams p;
  p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
  p.z_out0 = (int)z_out0; p.z_out1 = (int)z_out1;
  p.h = h;
  p.rw = TX + 2 * h;
  p.rh = TY + 2 * h;
  if (p.rw > 255 || p.rh > 255) return VISFD_HIP_OK;   // window columns and region rows travel as bytes
  p.sp = tv_box_row(h);
  p.nsl = tv_box_slice(h);
  const size_t slice_bytes = sizeof(float4) * (size_t)p.nsl;
  p.tiles_x = (int)((nx + TX - 1) / TX);
  p.tiles_y = (int)((ny + TY - 1) / TY);
  p.zrun = 8;    // (sweep at 1024^3, tools/tv_sweep.py: 2..8 planes 280-281 ms, 16: 283, 32: 290, 64: 296, 128: 306 -- short runs keep the
                 //  workgroups of the chip on neighbouring planes, whose lists and slices they then share in L2)
  if (ctx->opt.tv_zrun >= 1 && ctx->opt.tv_zrun <= 4096) p.zrun = ctx->opt.tv_zrun;
  if ((i64)p.zrun > z_out1 - z_out0) p.zrun = (int)(z_out1 - z_out0);
  if (p.zrun < 1) p.zrun = 1;
  const i64 nruns = (z_out1 - z_out0 + p.zrun - 1) / p.zrun;
  const i64 nblk = (i64)p.tiles_x * p.tiles_y * nruns;
  if (nblk > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  if (nblk <= 0) { *handled = true; return VISFD_HIP_OK; }
  // two slices + the per-pass table of row stretches ([2h + 4 planes][pairs][waves] words)
  const size_t lds = 2 * slice_bytes + sizeof(unsigned) * (size_t)(2 * h + 4) * NW * (exact ? NPAIR : 1);
  const size_t lds_static = sizeof(float4) * (LSLOTS + 1) + sizeof(uint2) * LSLOTS + sizeof(uint2) * NW * NSUB * 2 * HCAP + 1536;
  static_assert(2 * HCAP == HX, "the two kernels' hit lists take the same LDS");
  if (lds + lds_static > 150 * 1024) return VISFD_HIP_OK;   // window too wide: the caller falls back

  // ---- the sender lists of the planes the receiver planes [z_out0, z_out1) reach -------------------------------------------
  ListGeo g;
  g.nx = p.nx; g.ny = p.ny; g.nz = p.nz; g.h = h; g.ntx = p.tiles_x;
  g.zl0 = (int)std::max<i64>(z_out0 - h, 0);
  g.nzl = (int)(std::min<i64>(z_out1 + h, nz) - g.zl0);
  p.zl0 = g.zl0; p.nzl = g.nzl;
  const size_t nrows = (size_t)g.nzl * (size_t)(ny + 1) * (size_t)g.ntx;
  if ((i64)g.nzl * ny > 0x7fffffffLL) return VISFD_HIP_OK;
  unsigned* rows = nullptr;
  unsigned* counter = nullptr;
  VH_TRY(ws(ctx, WS_COUNTER, 16, &counter));
  if (ws(ctx, WS_TVLIST, nrows, &rows) != VISFD_HIP_OK) { set_error(""); (void)hipGetLastError(); return VISFD_HIP_OK; }
  unsigned long long* total_dev = reinterpret_cast<unsigned long long*>(counter + 4);
  VH_HIP(hipMemsetAsync(counter, 0, 16 * sizeof(unsigned), st));
  const unsigned row_blocks = (unsigned)(((i64)g.nzl * ny + LNT / 64 - 1) / (LNT / 64));
#define VH_TVL_ROWS(WR, ENT, POS)                                                                                         \
  do {                                                                                                                    \
    if (exact) tvl_row_kernel<WR, 1><<<dim3(row_blocks), dim3(LNT), 0, st>>>(sal, dir, mask_src, g, rows, ENT, POS, counter + 6, 0);             \
    else if (exponent == 4) tvl_row_kernel<WR, 0><<<dim3(row_blocks), dim3(LNT), 0, st>>>(sal, dir, mask_src, g, rows, ENT, POS, counter + 6, fold); \
    else tvl_row_kernel<WR, 2><<<dim3(row_blocks), dim3(LNT), 0, st>>>(sal, dir, mask_src, g, rows, ENT, POS, counter + 6, fold);             \
  } while (0)
  int fold = 0;
  VH_TVL_ROWS(false, nullptr, nullptr);
  tvl_scan_kernel<<<dim3((unsigned)(((size_t)g.nzl * g.ntx + LNT - 1) / LNT)), dim3(LNT), 0, st>>>(g, rows, total_dev);
  VH_HIP(hipGetLastError());
  // the lists' total length decides the size of the entry arrays: the one place this launch waits for the device
  unsigned long long tot2[2] = {0, 0};   // {total, (negative flag, -)}: counter words 4..7
  VH_HIP(hipMemcpyAsync(tot2, total_dev, sizeof(tot2), hipMemcpyDeviceToHost, st));
  VH_HIP(hipStreamSynchronize(st));
  const unsigned long long total = tot2[0];
  fold = ((unsigned)tot2[1] & 1u) ? 0 : 1;   // every listed saliency positive: the 18-instruction vote
  if (ctx->opt.tv_no_fold) fold = 0;         // (tests: the general form on positive saliencies too)
  if (exact && ((unsigned)tot2[1] & 2u)) return VISFD_HIP_OK;   // a non-finite saliency: the zero-padded slices would spread it
  if (exact && ((unsigned)tot2[1] & 4u)) return VISFD_HIP_OK;   // a weighted source mask: fv = w * mask value (feature.hpp:2262-2275)
  if (total >= (1ull << 32) - 2048) return VISFD_HIP_OK;   // 32-bit entry indices: the caller falls back
  unsigned char* lists = nullptr;
  if (ws(ctx, WS_TVSCRATCH, (size_t)(total + 16) * 20, &lists) != VISFD_HIP_OK) { set_error(""); (void)hipGetLastError(); return VISFD_HIP_OK; }
  float4* const lst_ent = reinterpret_cast<float4*>(lists);
  unsigned* const lst_pos = reinterpret_cast<unsigned*>(lists + (size_t)(total + 16) * 16);
  if (ctx->opt.tv_poison) VH_HIP(hipMemsetAsync(lists, 0xff, (size_t)(total + 16) * 20, st));
  VH_TVL_ROWS(true, lst_ent, lst_pos);
#undef VH_TVL_ROWS
  VH_HIP(hipGetLastError());

  size_t wg_per_cu = (160 * 1024) / (lds + lds_static);
  if (wg_per_cu > 3) wg_per_cu = 3;   // 6 waves per SIMD (80 VGPRs)
  if (wg_per_cu < 1) wg_per_cu = 1;
  i64 ngrid = (i64)ctx->num_cus * (i64)wg_per_cu;
  // slab runs: workgroup slots left free for the transport's kernels while a halo is in flight (slab.hip) -- counted
  // against THIS kernel's own chip-filling grid
  if (ctx->opt.tv_reserve_wg > 0) ngrid = std::max<i64>(ngrid - ctx->opt.tv_reserve_wg, 1);
  if (ctx->opt.tv_max_wg > 0 && ngrid > ctx->opt.tv_max_wg) ngrid = ctx->opt.tv_max_wg;
  if (ngrid > nblk) ngrid = nblk;
  if (ctx->opt.tv_poison) {   // tests: everything the kernel may read without having written it becomes NaN
    VH_HIP(hipMemsetAsync(ten, 0xff, sizeof(float) * 6 * (size_t)(nx * ny * nz), st));
    VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    lds_poison_kernel<<<dim3(1024), dim3(256), 160 * 1024, st>>>(counter + 8);
  }
#define VH_BOX_LAUNCH(MD, FD)                                                                                          \
  do {                                                                                                                 \
    VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tv_box_kernel<MD, FD>),                                  \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                 \
    tv_box_kernel<MD, FD><<<dim3((unsigned)ngrid), dim3(NT), lds, st>>>(ten, mask_dst, dtab_box, p, counter, (unsigned)nblk, \
                                                                       lst_ent, lst_pos, rows);                        \
  } while (0)
#define VH_BOXX_LAUNCH(MD)                                                                                             \
  do {                                                                                                                 \
    VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tv_boxx_kernel<MD>),                                     \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                 \
    tv_boxx_kernel<MD><<<dim3((unsigned)ngrid), dim3(NT), lds, st>>>(ten, mask_dst, dtab_box, p, counter, (unsigned)nblk, \
                                                                    lst_ent, lst_pos, rows);                           \
  } while (0)
  if (exact)              { if (exponent == 4) VH_BOXX_LAUNCH(0); else VH_BOXX_LAUNCH(2); }
  else if (exponent == 4) { if (fold) VH_BOX_LAUNCH(0, true); else VH_BOX_LAUNCH(0, false); }
  else                    { if (fold) VH_BOX_LAUNCH(2, true); else VH_BOX_LAUNCH(2, false); }
#undef VH_BOX_LAUNCH
#undef VH_BOXX_LAUNCH
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
