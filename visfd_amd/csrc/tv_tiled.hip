// tv_tiled.hip -- LDS-tiled dense stick tensor voting for gfx950
// (reference lib/visfd/feature.hpp:1914-2037 and :2217-2384).
//
// The reference walks, for every receiver voxel, the whole (2h+1)^3 window and skips senders whose saliency is zero
// (typically 95 % of them, feature.hpp:1704-1709).  Here the skipping is done once per workgroup instead of once per
// receiver:
//
//   * workgroups are PERSISTENT: as many as the chip holds, each claiming units of work -- an 8 x 32 tile of receivers
//     over a run of 32 consecutive receiver planes -- from a global counter until it is exhausted (a plain grid left wave
//     slots empty on volumes whose sender density varies, see the kernel).  Receiver planes are taken TWO AT A TIME: one
//     receiver per thread, a wave = 8 x 4 x 2 receivers (lanes 0-31 on plane z, lanes 32-63 on plane z+1), eight waves;
//   * LISTING: the salient, unmasked senders of the (8+2h) x (32+2h) region of a sender plane -- saliency, normal, position
//     bytes, table offset, mask value: 32 bytes -- are compacted IN VOTE ORDER (ordered, ballot-based prefix sums:
//     deterministic) into a per-workgroup ring of 2h+2 planes in global memory, once per unit and plane: consecutive
//     pairs of receiver planes share 2h of their 2h+2 sender planes, so all but the first pair of a run list two new planes;
//   * for the receiver planes z and z+1, sender planes are visited from z+1+h down to z-h (= jz ascending for both): the
//     plane's list is read back from the ring (L2) into LDS; the (2h+1)^2 slices of the vote table for jz and jz+1 sit in
//     two LDS slots chosen by the parity of jz, so a step of the sender plane copies one new slice;
//   * the SWEEP: every wave walks the list in order.  A sender is tested against the wave's 64 receivers --
//     jx^2+jy^2+jz^2 <= h^2 as ONE v_dot4_i32_i8 on signed bytes (see the kernel), an exact superset of the table's
//     spherical support -- and voted at once by the lanes it reaches, under their execution mask.  The sender's data
//     come from uniform-address (broadcast) LDS reads: no bank conflicts, no per-lane bookkeeping.  The region is as wide
//     as one wave's reach in x and the list is in row order, so a wave sweeps only the contiguous stretch of the list
//     whose rows it can reach on that plane (4 + 2 sqrt(h^2 - jz^2) of the 32 + 2h rows).  The upper receiver plane's
//     distance test is the lower one's plus a per-plane constant in the accumulator operand;
//   * weights and unit displacements come from the LDS copy of the table slice (w, rhat_x, rhat_y, rhat_z as one
//     float4, signs included).  The table index is linear in j = receiver - sender, so the byte address of a vote's
//     table entry is  R(lane) - E(sender):  one subtraction per vote.
//
// Order of accumulation per receiver: jz ascending (plane order), then jy, jx ascending, exactly the reference's; each
// vote is the same chain of float multiplies and adds (no FMA), so tensors are bit-identical to the CPU path for
// angular exponents 2 and 4.  A tap of zero weight on the rim of the support votes +-0, which leaves the sums unchanged.
//
// Listing and sweeping are separate phases that share no registers (round 2): the first sweep kernel compacted planes
// straight into the LDS list from inside the plane loop, and the state of that code -- seven region voxels per thread,
// five buffer descriptors -- stayed live across the sweeps: 140 VGPRs and 130 SGPRs spilled to scratch memory, whose
// loads and stores were most of the kernel's 2 TB of memory traffic per 1024^3 launch (profiles/r02_tv_design.txt).
#include <vector>

#include "common.hpp"

namespace vh {

namespace {

constexpr int NT = 512;
constexpr int NW = NT / 64;
constexpr int TX = 8, TY = 4 * NW;    // receivers of a workgroup: 8 x 32 on each of TWO consecutive planes; a wave owns four
                                      // rows of both planes (lanes 0-31: plane z, lanes 32-63: plane z+1)
static_assert(TY <= 64, "row coordinates relative to the tile centre are packed as signed bytes");
constexpr int CAP = 256;             // list entries held in LDS per sweep and list (one per thread of the list's share)
// RECEIVER PAIRS PER PASS (round 3).  The sender planes the pairs (z, z+1) and (z+2, z+3) meet at step t of their
// ascending jz are different planes, but jz -- and so the two table slices -- are the same: with NP = 2 a workgroup takes both
// pairs through the steps together (two lists, two sweeps per barrier interval, each wave 2 x 6 sums).  Every receiver still
// takes its votes in the reference's order; only the interleaving of DIFFERENT receivers' sums changes.
constexpr int NP = 2;
static_assert(NP * CAP <= NT, "the replay loads one entry per thread");
// PACKED LISTS (round 3): the lists of a step share the NT entry slots of LDS -- their entries are dealt to the
// threads as ONE sequence, each list's share of an interval lands contiguously (an even start, 8 never-hit entries of slack
// behind it), and an interval ends when NT entries are in, not when the longest list has had CAP.  Every list is still swept
// in its own order, interval after interval, so every receiver takes its votes in the reference's order.
constexpr int LSLOTS = NT + 12 * NP;   // l_pos / l_ent entry slots
constexpr int NCH_MAX = NT >= 512 ? 4 : 5;   // chunks of the region per wave the two-plane lister handles
constexpr int RING_BYTES = 32;       // bytes per entry of the scratch rings
constexpr unsigned OOB = 0x7ffffff0u;  // byte offset beyond any plane descriptor: reads give 0

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, 0));
}

typedef float f4v __attribute__((ext_vector_type(4)));

// LDS (address space 3) pointers as 32-bit integers and back
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(3))) T* lds_ptr(unsigned a) {
  return (const __attribute__((address_space(3))) T*)(uintptr_t)a;
}

struct TiledParams {
  int nx, ny, nz;
  int z_out0, z_out1;    // receiver planes [z_out0, z_out1)
  int h;                 // window halfwidth
  int rw, rh;            // region width = TX + 2h, height = TY + 2h
  int rw_magic;          // q / rw == (q * rw_magic) >> 20 for every region position q (checked by the launcher)
  int nchunk;            // 64-voxel chunks of the region per wave
  int tiles_x, tiles_y;
  int exponent, curves;
  int zrun;              // receiver planes per unit of work
  int relist;            // 1: list every sender plane for every receiver plane (option tv_no_replay; tests)
  int sp;                // row stride of a table slice in float4 entries (>= 2h+1)
};

__device__ __forceinline__ void acc(float& t, float x) {
  asm("v_add_f32 %0, %0, %1" : "+v"(t) : "v"(x));
}

// MODE 0: surfaces with angular exponent 4 (the CLI default, settings.cpp:154); MODE 2: surfaces with exponent 2;
// MODE 1: general (any exponent through pow, curve mode); MODE 3: no tensor at all -- the sum of the vote weights
// ("denominator" of feature.hpp:2376-2377) into channel 0.
template <int MODE>
__device__ __forceinline__ void vote_dir(float sal, float fv, float r0, float r1, float r2, float n0, float n1,
                                         float n2, int exponent, int curves, float& bse, float& m0, float& m1,
                                         float& m2) {
  const float u = (r0 * n0 + r1 * n1) + r2 * n2;
  const float ux2 = u * 2.0f;
  const float u2 = u * u;
  const float c2 = 1.0f - u2;
  float dec;
  if (MODE == 0 || MODE == 2) {
    dec = (MODE == 0) ? c2 * c2 : c2;
    m0 = ux2 * r0 - n0; m1 = ux2 * r1 - n1; m2 = ux2 * r2 - n2;
  } else {
    const float ang = curves ? u2 : c2;
    if (exponent == 4) dec = ang * ang;
    else if (exponent == 2) dec = ang;
    else dec = (float)pow((double)ang, 0.5 * (double)exponent);
    if (curves) { m0 = n0 - ux2 * r0; m1 = n1 - ux2 * r1; m2 = n2 - ux2 * r2; }
    else        { m0 = ux2 * r0 - n0; m1 = ux2 * r1 - n1; m2 = ux2 * r2 - n2; }
  }
  bse = (sal * fv) * dec;
}

__device__ __forceinline__ void vote_acc(float T[6], float bse, float m0, float m1, float m2) {
  const float b0 = bse * m0, b1 = bse * m1, b2 = bse * m2;
  // accumulate in place (tied operands keep the six sums in fixed registers across the sweep;
  // v_add_f32 is the same IEEE add the compiler emits for "+")
  acc(T[0], b0 * m0);
  acc(T[3], b0 * m1);
  acc(T[5], b0 * m2);
  acc(T[1], b1 * m1);
  acc(T[4], b1 * m2);
  acc(T[2], b2 * m2);
}

template <bool MASKED_SRC, int MODE>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(MODE != 1 ? 8 : 2, 8)))
tv_tiled_kernel(const float* __restrict__ sal, const float* __restrict__ dir, float* __restrict__ ten,
                const float* __restrict__ mask_src, const float* __restrict__ mask_dst,
                const float4* __restrict__ table /* [(2h+1)^3] : w, rhat_x, rhat_y, rhat_z at j */,
                TiledParams p, unsigned* __restrict__ tile_counter, unsigned ntiles,
                unsigned char* __restrict__ scratch /* per-workgroup rings of compacted sender planes */) {
  // Static LDS (compile-time addresses fold into the DS instructions' immediate offsets):
  //   l_ent[e]  float4 {sal, n0, n1, n2} of list entry e
  //   l_pos[e]  {distance-test operand, table offset E}: packed signed bytes (e'x, e'y, -(|e'|^2 >> 7), |e'|^2 & 127)
  //             with e' = sender position relative to the tile centre and the plane of the receivers; 8 entries of
  //             slack past the list hold a never-hit operand, so the sweep runs in whole batches of four and may
  //             prefetch one batch past the end
  //   l_mv[e]   source-mask value of the entry (masked kernels)
  __shared__ __attribute__((aligned(16))) float4 l_ent[LSLOTS];
  __shared__ __attribute__((aligned(16))) uint2 l_pos[LSLOTS];
  __shared__ float l_mv[MASKED_SRC ? LSLOTS : 1];
  __shared__ int wave_tot[2][2][NW];
  __shared__ unsigned claimed_tile;
  __shared__ int plane_cnt[88];              // entries per ring slot, [2h + 2 NP] (h <= 40)
  // dynamic LDS: two table slices (jz and jz + 1 of the current sender plane), [2][(2h+1)^2] float4
  extern __shared__ __attribute__((aligned(16))) unsigned char slices[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = p.h;
  const int S = 2 * h + 1;       // table row length = planes per ring
  // LDS rows of a slice are SP >= S float4 apart and the lanes of a half wave are dealt to its 8 x 4 receivers as two
  // 4-column blocks, one per 16-lane group of ds_read_b128: 2 instead of 4 LDS cycles per half-wave table read
  // (tools/lds_bank_model.py; round 3).  Which lane holds which receiver does not touch any receiver's sums.
  const int SP = p.sp;
  const int nsl = S * SP;        // float4 entries per slice (the table in global memory has the same padded rows)
  const int R = p.rw * p.rh;     // region positions per plane
  const i64 plane = (i64)p.nx * p.ny;
  const i64 nvox = plane * p.nz;
  const int plane_bytes = (int)(plane * 4);
  const size_t plane_stride = (size_t)R * RING_BYTES;
  const int P = S + 2 * NP - 1;  // sender planes the receiver planes of a pass reach = slots of the ring
  unsigned char* const ring = scratch + (size_t)blockIdx.x * plane_stride * P;
  int npar = 0;                  // parity of the wave-total buffers

  // ---- persistent workgroups: units are claimed from a global counter ---------------------------------------
  // The time a unit takes follows the local density of senders (membranes: tens of times the average), and the
  // hardware hands out workgroups of a plain grid in order, round-robin over the XCDs: on membrane-rich volumes
  // that left a fifth of the wave slots empty (rocprofv3 OccupancyPercent 40 of 50, VALUBusy 80 %; uniform
  // noise: 47.5 and 100 %).  Here the grid is just large enough to fill the chip and every workgroup keeps
  // claiming the next unit until the counter passes the last one -- an exit every wave reaches.
  for (;;) {
    if (tid == 0) claimed_tile = atomicAdd(tile_counter, 1u);
    __syncthreads();   // also: the previous unit's sweeps are complete
    unsigned b = claimed_tile;
    __syncthreads();   // everyone has read it before thread 0 claims again
    if (b >= ntiles) break;
    const int tile_x = b % p.tiles_x;
    b /= p.tiles_x;
    const int tile_y = b % p.tiles_y;
    // a unit of work: one 8 x 32 tile over a run of consecutive receiver planes [z_run0, z_run1)
    const int z_run0 = p.z_out0 + (int)(b / p.tiles_y) * p.zrun;
    const int z_run1 = min(z_run0 + p.zrun, p.z_out1);
    const int x0 = tile_x * TX, y0 = tile_y * TY;

    // receiver of this thread: wave w owns the rows [8w, 8w+8) of the tile.  Region-relative coordinates:
    // the sender region starts at (x0-h, y0-h, rz-h), so the receiver sits at (lx+h, ly+h, h).  The region is exactly
    // as wide as one wave's reach in x, and list order is row order, so the senders a wave can reach are one
    // contiguous stretch of the list (see the replay below).
    const int half = lane >> 5;                       // 0: receiver plane rz, 1: plane rz + 1
    const int l5 = lane & 31;
    const int lrow = (l5 < 8) ? 0 : (l5 < 16) ? 1 : (l5 < 24) ? 2 : 3;
    const int lx = (l5 & 3) + (((0xc33cu >> (l5 >> 1)) & 1u) ? 4 : 0), ly = wave * 4 + lrow;
    const int rx = x0 + lx, ry = y0 + ly;
    const bool r_in = rx < p.nx && ry < p.ny;
    // Distance test  |r - e|^2 <= h^2  as ONE dot product per (receiver, sender): with coordinates
    // relative to the tile centre (r' = (lx-4, ly-16, 0), e' = (ex-h-4, ey-h-16, ez-h)) and
    // |e'|^2 = 128 q + m,
    //   |r'-e'|^2 - h^2 - 1  =  (-2r'x, -2r'y, -128, 1) . (e'x, e'y, -q, m)  +  (|r'|^2 - h^2 - 1),
    // every factor a signed byte (|e'x| <= h+4, |e'y| <= h+16, |e'z| <= h <= 40, so q <= 52), the last term a per-lane
    // accumulator.
    const int rpx = lx - TX / 2, rpy = ly - TY / 2;
    const unsigned recv4 = (unsigned)((-2 * rpx) & 0xff) | ((unsigned)((-2 * rpy) & 0xff) << 8) | (0x80u << 16) | (1u << 24);
    const int recv_c = rpx * rpx + rpy * rpy - h * h - 1;
    constexpr unsigned NEVER_HIT = 0x009c0000u;   // operand whose dot product is positive for every receiver (-128 * -100)
    // (jy+h)*S + (jx+h) with jx = lx+h-ex, jy = ly+h-ey  =  [(ly+2h)*S + lx+2h] - [ey*S + ex]
    const unsigned r16_0 = lds_addr(slices) + (unsigned)(16 * ((ly + 2 * h) * SP + lx + 2 * h));   // in LDS slice slot 0
    unsigned r16s = r16_0;                                                                        // in this lane's slot
    const unsigned ent_base = lds_addr(l_ent);

    // ---- LISTING: sender plane sz of this tile's region into its ring slot, in vote order -------------------
    // Vote order inside a plane is DESCENDING region position (jy, jx ascending = sender y, x descending).  Wave w
    // owns the positions [w, w+1) * 64 nchunk, lane l of chunk j the position 64 (w nchunk + j) + l: two passes over
    // the saliencies (the second one hits in L1/L2) instead of state kept in registers.  Region reads are buffer loads
    // with hardware range checking: a per-plane descriptor (scalar) plus a 32-bit byte offset per voxel; positions
    // outside the image or the region use an out-of-range offset and read as 0.0f (= not salient) without branches.
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
      int running = 0, total = 0;   // entries at higher positions than this wave's; entries of the plane
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
            const int idx = running + (tb - below - 1);      // salient lanes above this one come first
            float4 a = make_float4(s, 0.0f, 0.0f, 0.0f);
            if (MODE != 3) { a.y = buf_load(rd0, off); a.z = buf_load(rd1, off); a.w = buf_load(rd2, off); }
            unsigned mv = 0u;
            if (MASKED_SRC) mv = __float_as_uint(buf_load(rm, off));
            const int epx = ex - h - TX / 2, epy = ey - h - TY / 2;
            unsigned char* dst_e = ring_plane + (size_t)idx * RING_BYTES;
            *reinterpret_cast<float4*>(dst_e) = a;
            *reinterpret_cast<uint4*>(dst_e + 16) =
                make_uint4((unsigned)(epx & 0xff) | ((unsigned)(epy & 0xff) << 8), (unsigned)(epx * epx + epy * epy),
                           (unsigned)(16 * (ey * SP + ex)), mv);
          }
          running += tb;
        }
      }
      if (tid == 0) plane_cnt[slot] = total;
    };

    // ---- LISTING, two planes at a time (round 3; window regions of <= 4 chunks per wave: h <= 12).  Every load
    // of a phase is in flight at once -- the saliencies of both planes (kept in registers across the barrier: one read per
    // voxel), then the normals of a plane's salient voxels -- and both planes share one barrier.  The entries and their order
    // are list_plane's.  A plane index < 0 means "no plane" (zero-length descriptors: nothing is salient).
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
        float n0[NCH], n1[NCH], n2[NCH];
        unsigned mvv[NCH];
#pragma unroll
        for (int j = 0; j < NCH; j++) {   // the normals of the salient voxels only, all chunks requested before the first use
          n0[j] = n1[j] = n2[j] = 0.0f;
          mvv[j] = 0u;
          if (sv[k][j] != 0.0f) {
            if (MODE != 3) { n0[j] = buf_load(rd0, off[j]); n1[j] = buf_load(rd1, off[j]); n2[j] = buf_load(rd2, off[j]); }
            if (MASKED_SRC) mvv[j] = __float_as_uint(buf_load(rm, off[j]));
          }
        }
#pragma unroll
        for (int j = NCH - 1; j >= 0; j--) {   // descending region position = vote order
          const bool f = sv[k][j] != 0.0f;
          const unsigned long long bal = __builtin_amdgcn_ballot_w64(f);
          const int tb = __builtin_popcountll(bal);
          const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
          if (f) {
            const int idx = running + (tb - below - 1);
            const int q = q0 + 64 * j;
            const int ey = (int)(((unsigned)q * (unsigned)p.rw_magic) >> 20);
            const int ex = q - ey * p.rw;
            const int epx = ex - h - TX / 2, epy = ey - h - TY / 2;
            unsigned char* dst_e = ring_plane + (size_t)idx * RING_BYTES;
            *reinterpret_cast<float4*>(dst_e) = make_float4(sv[k][j], n0[j], n1[j], n2[j]);
            *reinterpret_cast<uint4*>(dst_e + 16) =
                make_uint4((unsigned)(epx & 0xff) | ((unsigned)(epy & 0xff) << 8), (unsigned)(epx * epx + epy * epy),
                           (unsigned)(16 * (ey * SP + ex)), mvv[j]);
          }
          running += tb;
        }
      }
    };

    float TT[NP][6];

    // ---- the SWEEP over entries [i0, i1) of list li, in vote order, into the sums T --------------------------
    // recv_c_live: this lane's accumulator operand of the distance test (receivers that take no votes never hit: large, positive)
    auto sweep = [&](float (&T)[6], int li, int i0, int i1, int recv_c_live) {
      // the table entry is requested together with the sender's own data: its address needs only E, which came with
      // the batch; ent = LDS address of the batch's first entry (a vector register: DS addresses cannot be scalar)
      auto vote_one = [&](unsigned ent, int k, int s, unsigned e16) {
        const f4v tw = *lds_ptr<f4v>(r16s - e16);
        float fv = tw.x;
        if (MASKED_SRC) fv = fv * l_mv[li + s];    // fv = w * mask value first (feature.hpp:2262-2275), then sal * fv
        if (MODE == 3) {
          acc(T[0], fv);                      // "denominator += filter_val" (feature.hpp:2376-2377)
        } else {
          const f4v d = *lds_ptr<f4v>(ent + 16u * (unsigned)k);
          float bse, m0, m1, m2;
          vote_dir<MODE>(d.x, fv, tw.y, tw.z, tw.w, d.y, d.z, d.w, p.exponent, p.curves, bse, m0, m1, m2);
          vote_acc(T, bse, m0, m1, m2);
        }
      };
      auto batch = [&](const uint4& ca, const uint4& cb, unsigned ent, int s0) {
        int d0, d1, d2, d3;
        // four dots back to back: a dot result may be read by the VALU three instructions later at the earliest, and
        // the compiler does not see hazards of instructions inside an asm block
        asm("v_dot4_i32_i8 %0, %4, %6, %5\n\t"
            "v_dot4_i32_i8 %1, %4, %7, %5\n\t"
            "v_dot4_i32_i8 %2, %4, %8, %5\n\t"
            "v_dot4_i32_i8 %3, %4, %9, %5\n\t"
            "s_nop 2"
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
            : "v"(recv4), "v"(recv_c_live), "v"(ca.x), "v"(ca.z), "v"(cb.x), "v"(cb.z));
        // most tested senders vote (the list is already cut to the rows the wave can reach): votes on the fall-through path
        if (__builtin_expect(d0 < 0, 1)) vote_one(ent, 0, s0, ca.y);
        if (__builtin_expect(d1 < 0, 1)) vote_one(ent, 1, s0 + 1, ca.w);
        if (__builtin_expect(d2 < 0, 1)) vote_one(ent, 2, s0 + 2, cb.y);
        if (__builtin_expect(d3 < 0, 1)) vote_one(ent, 3, s0 + 3, cb.w);
      };
      // batches of four senders, two per trip: the next batch is in flight while this one is tested and voted, and the
      // two register sets swap roles without copies
      int s0 = i0 & ~1;                      // l_pos is read two entries at a time
      // (li: the list's first LDS slot -- packed lists -- or its number)
      const uint4* pq = reinterpret_cast<const uint4*>(l_pos + li) + (s0 >> 1);
      unsigned ent = ent_base + 16u * (unsigned)(li + s0);
      asm volatile("" : "+v"(ent));
      uint4 a0 = pq[0], a1 = pq[1];
      while (s0 < i1) {   // uniform
        const uint4 b0 = pq[2], b1 = pq[3];
        batch(a0, a1, ent, s0);
        if (s0 + 4 >= i1) break;
        a0 = pq[4];
        a1 = pq[5];
        batch(b0, b1, ent + 64u, s0 + 4);
        pq += 4;
        ent += 128u;
        asm volatile("" : "+v"(ent));
        s0 += 8;
      }
    };

    // ---- the unit's receiver planes, bottom up, TWO AT A TIME ----------------------------------------------
    // A sweep of sender plane sz serves the receivers of planes rz (jz = rz - sz) and rz + 1 (jz + 1) together: a wave's
    // 64 receivers are 8 x 4 x 2 instead of 8 x 8 x 1, which a sender's ball covers better (fewer, fuller vote steps), and
    // every list is brought into LDS once per two receiver planes.  The table slices of jz and jz + 1 sit in two LDS slots
    // chosen by the parity of jz, so a step of the sender plane needs ONE new slice.
    int cached_lo = 1, cached_hi = 0;     // sender planes whose lists are in the ring (slot of plane sz: sz mod P)
    int slot_a = 1 << 20, slot_b = 1 << 20;   // which table slice LDS slot 0 / 1 holds (uniform)
    float4* const sl4 = reinterpret_cast<float4*>(slices);
    for (int rz = z_run0; rz < z_run1; rz += 2 * NP) {
      // sender planes that reach the LIVE receivers of this pass: a run may end inside a pass, and nothing above the last live
      // receiver + h is needed then -- in a slab run that plane may not even be complete yet (visfd_amd/slab.py votes the
      // interior band while the halo planes above it are still in flight)
      const int sz_hi = min(min(rz + 2 * NP - 1, z_run1 - 1) + h, p.nz - 1), sz_lo = max(rz - h, 0);
      // the window holds at most P planes, so a plane that enters it takes the slot of one that has left
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

      i64 rc[NP];                                                // this lane's receiver of pair pp (plane rz + 2 pp + half)
      bool r_live[NP];
#pragma unroll
      for (int pp = 0; pp < NP; pp++) {
        const int rzl = rz + 2 * pp + half;
        const bool z_in = rzl < z_run1;                          // (a run may end with half a pair, or without the second pair)
        rc[pp] = (i64)rzl * plane + (i64)ry * p.nx + rx;
        r_live[pp] = r_in && z_in && !(mask_dst && mask_dst[(r_in && z_in) ? rc[pp] : 0] == 0.0f);
#pragma unroll
        for (int k = 0; k < 6; k++) TT[pp][k] = 0.0f;
      }
      __syncthreads();   // ring entries and counts of this pass are visible

      // step t = 0 .. 2h+1: pair pp meets sender plane (rz + 2 pp) + 1 + h - t, i.e. jz = t - h - 1 for its lower receiver plane
      // and jz + 1 for its upper one -- ascending, as the reference visits them, and the same for every pair
      for (int t = 0; t <= 2 * h + 1; t++) {
        const int jz0 = t - h - 1, jz1 = jz0 + 1;              // one of them may be outside the window (never hit then)
        const bool v0 = jz0 >= -h, v1 = jz1 <= h;
        int lsz[NP], lcnt[NP];
        int cmax = 0;
#pragma unroll
        for (int pp = 0; pp < NP; pp++) {
          const int z = rz + 2 * pp;
          lsz[pp] = z - jz0;
          // (planes above the last live receiver of the pair + h are not needed, and in a slab run not there yet)
          const bool have = z < z_run1 && lsz[pp] >= sz_lo && lsz[pp] <= min(min(z + 1, z_run1 - 1) + h, p.nz - 1);
          lcnt[pp] = have ? __builtin_amdgcn_readfirstlane(plane_cnt[lsz[pp] % P]) : 0;
          cmax = max(cmax, lcnt[pp]);
        }
        if (cmax == 0) continue;   // uniform
        // list and slices are free: every sweep ends with a barrier.  Slot of slice jz: (jz + h + 1) & 1.
        {
          const int s0 = (jz0 + h + 1) & 1, s1 = s0 ^ 1;
          if (v0 && (s0 ? slot_b : slot_a) != jz0) {
            const float4* src4 = table + (i64)(jz0 + h) * nsl;
            for (int i = tid; i < nsl; i += NT) sl4[s0 * nsl + i] = src4[i];
            (s0 ? slot_b : slot_a) = jz0;
          }
          if (v1 && (s1 ? slot_b : slot_a) != jz1) {
            const float4* src4 = table + (i64)(jz1 + h) * nsl;
            for (int i = tid; i < nsl; i += NT) sl4[s1 * nsl + i] = src4[i];
            (s1 ? slot_b : slot_a) = jz1;
          }
          r16s = r16_0 + (unsigned)(16 * nsl) * (unsigned)(half ? s1 : s0);
        }
        const int epz = -jz0;                                  // sender plane relative to the LOWER receiver plane
        const int epz2 = epz * epz;
        // rows a wave can reach on this plane: |r'y - e'y| <= rho = floor(sqrt(h^2 - jz^2)) for the nearer of its two
        // receiver planes, r'y in [4w-16, 4w-13]
        const int jn = min(v0 ? jz0 * jz0 : (1 << 20), v1 ? jz1 * jz1 : (1 << 20));
        int rho = (int)__builtin_sqrtf((float)(h * h - jn));
        while (rho * rho > h * h - jn) rho--;
        while ((rho + 1) * (rho + 1) <= h * h - jn) rho++;
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int t512 = wave * 64 + ln;                       // this thread's place in the interval's entry sequence
        int pre[NP + 1], pl[NP];                               // (uniform) first position of list k in the step's sequence; its ring slot
        pre[0] = 0;
#pragma unroll
        for (int k = 0; k < NP; k++) {
          pre[k + 1] = pre[k] + lcnt[k];
          pl[k] = __builtin_amdgcn_readfirstlane(((lsz[k] % P) + P) % P);
        }
        const int total = pre[NP];
        for (int done = 0; done < total; done += NT) {          // uniform
          int c[NP], len[NP], S[NP];                            // list k's share: sequence positions [c, c + len) of the NT, LDS slots from S
#pragma unroll
          for (int k = 0; k < NP; k++) {
            const int lo = min(max(pre[k], done), done + NT), hi = min(pre[k + 1], done + NT);
            c[k] = lo - done;
            len[k] = max(hi - lo, 0);
            S[k] = ((c[k] + 1) & ~1) + 10 * k;
          }
          const int g = done + t512;
          int k_me = 0;
#pragma unroll
          for (int k = 1; k < NP; k++) k_me += (g >= pre[k]) ? 1 : 0;
          int idx = g, slot = t512, pl_me = pl[0];
#pragma unroll
          for (int k = 0; k < NP; k++)
            if (k_me == k) { idx = g - pre[k]; slot = S[k] + (t512 - c[k]); pl_me = pl[k]; }
          if (g < total) {
            const unsigned char* src_e = ring + (size_t)pl_me * plane_stride + (size_t)idx * RING_BYTES;
            const float4 a = *reinterpret_cast<const float4*>(src_e);
            const uint4 m = *reinterpret_cast<const uint4*>(src_e + 16);
            l_ent[slot] = a;
            const int e2 = (int)m.y + epz2;
            l_pos[slot] = make_uint2(m.x | ((unsigned)((-(e2 >> 7)) & 0xff) << 16) | ((unsigned)(e2 & 127) << 24), m.z);
            if (MASKED_SRC) l_mv[slot] = __uint_as_float(m.w);
          }
          if (t512 < 8 * NP) {                                  // 8 never-hit entries behind every (non-empty) list's share
            int sk = S[0] + len[0], lk = len[0];
#pragma unroll
            for (int k = 1; k < NP; k++)
              if ((t512 >> 3) == k) { sk = S[k] + len[k]; lk = len[k]; }
            if (lk > 0) l_pos[sk + (t512 & 7)] = make_uint2(NEVER_HIT, 0u);
          }
          __syncthreads();   // lists (and slices) complete
          // entries are in descending row order: of list k, this wave needs those from the first one at or below row
          // 4w-13+rho to the last one at or above row 4w-16-rho; every wave counts both kinds itself, from the row bytes of the
          // position words in LDS, 64 entries at a time
          const int hi_row = 4 * wave - (TY / 2 - 3) + rho, lo_row = 4 * wave - TY / 2 - rho;
#pragma unroll
          for (int pp = 0; pp < NP; pp++) {
            int i0 = 0, i1 = 0;
            for (int j = 0; j < len[pp]; j += 64) {   // uniform
              int ey = -128;
              if (j + ln < len[pp]) ey = (int)(signed char)((l_pos[S[pp] + j + ln].x >> 8) & 0xff);
              i0 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ey > hi_row));
              i1 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ey >= lo_row));
            }
            // the upper plane's receivers see the sender one plane further down: |r - e|^2 grows by 1 - 2 epz
            const int rcl = r_live[pp] ? recv_c + (half ? 1 - 2 * epz : 0) : 0x100000;
            if (i1 > i0) sweep(TT[pp], S[pp], i0, i1, rcl);      // uniform
          }
          __syncthreads();   // everyone done reading before the lists or the slices are refilled
        }
      }

#pragma unroll
      for (int pp = 0; pp < NP; pp++)
        if (r_live[pp]) {
          if (MODE == 3) {
            ten[rc[pp]] = TT[pp][0];
          } else {
#pragma unroll
            for (int k = 0; k < 6; k++) __builtin_nontemporal_store(TT[pp][k], &ten[k * nvox + rc[pp]]);   // written once, not read here
          }
        }
    }   // next pass of the run
  }   // next unit
}

}  // namespace

// dtab: the vote table on the device with padded rows (tv.hip: tv_table_device).  weights_only: ten receives ONE plane, the sum of the
// weights of the votes each receiver takes (the normalisation denominator of feature.hpp:1784-1822) instead of tensors.
int dev_tv_tiled(visfd_hip_ctx* ctx, const float* sal, const float* dir, float* ten,
                 const float* mask_src, const float* mask_dst, i64 nx, i64 ny, i64 nz, i64 z_out0,
                 i64 z_out1, int h, const float4* dtab, int exponent, bool curves, bool weights_only, bool* handled) {
  *handled = false;
  if (h < 1 || h > 40) return VISFD_HIP_OK;  // table slice in LDS + byte-packed coordinates limits
  if (nx * ny >= (1LL << 29)) return VISFD_HIP_OK;  // plane descriptors are 32-bit
  const int n = 2 * h + 1;
  hipStream_t st = ctx->stream;

  TiledParams p;
  p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
  p.z_out0 = (int)z_out0; p.z_out1 = (int)z_out1;
  p.h = h;
  p.rw = TX + 2 * h;
  p.rh = TY + 2 * h;
  const int R = p.rw * p.rh;
  p.nchunk = (R + NT - 1) / NT;
  p.rw_magic = ((1 << 20) + p.rw - 1) / p.rw;
  for (int q = 0; q < p.nchunk * NT; q++)   // (<= 9216 positions)
    if ((int)(((unsigned)q * (unsigned)p.rw_magic) >> 20) != q / p.rw) return fail(VISFD_HIP_EINVAL, "tv_tiled: region index division");
  p.sp = tv_padded_row(h);
  const size_t slice_bytes = sizeof(float4) * (size_t)n * p.sp;
  p.tiles_x = (int)((nx + TX - 1) / TX);
  p.tiles_y = (int)((ny + TY - 1) / TY);
  p.exponent = exponent;
  p.curves = curves ? 1 : 0;
  p.relist = ctx->opt.tv_no_replay ? 1 : 0;
  // units of work: a tile over a run of receiver planes (sender-plane lists are shared within a run)
  p.zrun = 32;   // sweep at 1024^3: 16: 853 ms, 24-64: 820-833 ms, 128: 838 ms
  if (ctx->opt.tv_zrun >= 1 && ctx->opt.tv_zrun <= 4096) p.zrun = ctx->opt.tv_zrun;   // tuning aid
  if ((i64)p.zrun > z_out1 - z_out0) p.zrun = (int)(z_out1 - z_out0);
  if (p.zrun < 1) p.zrun = 1;
  const i64 nruns = (z_out1 - z_out0 + p.zrun - 1) / p.zrun;
  const i64 nblk = (i64)p.tiles_x * p.tiles_y * nruns;
  if (nblk > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  if (nblk <= 0) { *handled = true; return VISFD_HIP_OK; }
  const size_t lds = 2 * slice_bytes;   // dynamic part: the slices of jz and jz + 1
  const size_t lds_static = (sizeof(float4) + sizeof(uint2) + sizeof(float) * (mask_src ? 1 : 0)) * (size_t)(NT + 12 * NP + 16) + 2560;
  if (lds + lds_static > 150 * 1024) return VISFD_HIP_OK;   // window too wide for the LDS slice: baseline kernel
  const int mode = weights_only ? 3 : (curves ? 1 : (exponent == 4 ? 0 : (exponent == 2 ? 2 : 1)));
  // persistent workgroups (see the kernel): as many as the chip holds at once -- LDS allows 160 KB / (static +
  // dynamic) per CU, registers eight waves per SIMD = 4 workgroups of 8 waves -- each claiming units from a counter
  unsigned* counter = nullptr;
  VH_TRY(ws(ctx, WS_COUNTER, 16, &counter));
  VH_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned), st));
  size_t wg_per_cu = (160 * 1024) / (lds + lds_static);
  const size_t max_wg = mode == 1 ? 1 : 2048 / NT;   // eight waves per SIMD in all
  if (wg_per_cu > max_wg) wg_per_cu = max_wg;
  if (wg_per_cu < 1) wg_per_cu = 1;
  i64 ngrid = (i64)ctx->num_cus * (i64)wg_per_cu;
  // slab runs: workgroup slots left free for the transport's kernels while a halo is in flight (slab.hip) -- counted
  // against THIS kernel's own chip-filling grid
  if (ctx->opt.tv_reserve_wg > 0) ngrid = std::max<i64>(ngrid - ctx->opt.tv_reserve_wg, 1);
  if (ctx->opt.tv_max_wg > 0 && ngrid > ctx->opt.tv_max_wg) ngrid = ctx->opt.tv_max_wg;   // tests: many units per workgroup
  if (ngrid > nblk) ngrid = nblk;
  // scratch rings: (2h+1) planes x (8+2h)(32+2h) entries of 32 bytes per workgroup (2.9 GB for h = 12 on 256 CUs).  Very
  // wide windows are capped at 16 GB (fewer workgroups: their LDS slices allow only one or two per CU anyway); if the
  // allocation fails the grid is halved, and without any ring the caller's baseline kernel takes over.
  unsigned char* scratch = nullptr;
  const size_t per_wg = (size_t)(n + 2 * NP - 1) * R * RING_BYTES;
  if ((size_t)ngrid * per_wg > ((size_t)16 << 30)) ngrid = (i64)(((size_t)16 << 30) / per_wg);
  for (; ngrid >= 1; ngrid /= 2) {
    if (ws(ctx, WS_TVSCRATCH, per_wg * (size_t)ngrid, &scratch) == VISFD_HIP_OK) break;
    scratch = nullptr;
    set_error("");
    (void)hipGetLastError();
  }
  if (!scratch) return VISFD_HIP_OK;
#define VH_TV_LAUNCH(MSK, MD)                                                                        \
  do {                                                                                               \
    VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tv_tiled_kernel<MSK, MD>),             \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
    tv_tiled_kernel<MSK, MD><<<dim3((unsigned)ngrid), dim3(NT), lds, st>>>(sal, dir, ten, mask_src,  \
                                                                          mask_dst, dtab, p, counter, \
                                                                          (unsigned)nblk, scratch);  \
  } while (0)
  if (mask_src) {
    if (mode == 0) VH_TV_LAUNCH(true, 0); else if (mode == 2) VH_TV_LAUNCH(true, 2); else if (mode == 3) VH_TV_LAUNCH(true, 3);
    else VH_TV_LAUNCH(true, 1);
  } else {
    if (mode == 0) VH_TV_LAUNCH(false, 0); else if (mode == 2) VH_TV_LAUNCH(false, 2); else if (mode == 3) VH_TV_LAUNCH(false, 3);
    else VH_TV_LAUNCH(false, 1);
  }
#undef VH_TV_LAUNCH
  VH_HIP(hipGetLastError());
  *handled = true;
  return VISFD_HIP_OK;
}

}  // namespace vh
