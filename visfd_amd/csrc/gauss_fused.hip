// gauss_fused.hip -- single-sweep separable 3-D Gaussian for gfx950 (unmasked case; its Y+X half alone serves the
// masked filter after that filter's masking Z pass).
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
//   * Z pass: each thread owns NC columns of the (TX+2H) x (TY+2H) haloed tile and keeps, per column, the
//     2H+1 RUNNING SUMS of the output planes in flight in a register ring (the march is unrolled 2H+1
//     times so the ring is statically indexed).  The march goes DOWN in z: the reference adds the
//     terms of an output with j ascending, i.e. source planes in descending order, so each arriving
//     input plane can add its term to all 2H+1 sums in the reference's order (scatter form).  With
//     symmetric taps t[j]*f and t[-j]*f are the same IEEE product: H+1 multiplies per input, not 2H+1.
//     Right after the Z pass the next input plane is requested, so its HBM latency is covered by the
//     Y and X passes.  Loads are buffer loads with hardware range checking: columns outside the image
//     use an out-of-range offset and planes outside the image a zero-length descriptor, both of which
//     return 0.0f -- no branches, no exec masking;
//   * the Z-filtered haloed plane goes to LDS; Y pass: two adjacent x per lane (ds_read_b64 down a
//     column), result rows to LDS; X pass: four adjacent outputs per lane from ds_read_b128
//     windows, normalise, 16-byte buffer store (x-contiguous across lanes, dropped by the range
//     check outside the image; when nx is not a multiple of the group, the group cut by the end of a row is
//     written element by element).  The normaliser division uses a per-lane reciprocal with exact residual
//     corrections where that provably yields the IEEE quotient (see the X pass).
// Zero extension: a 0.0f sample adds an exact +0.0 to the accumulator, which equals skipping the
// term as the reference does (filter1d.hpp:98-99) for finite data.
//
// This file is compiled once per window half-width (-DVH_FUSED_H=h, see visfd_amd/build.py).
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

#ifndef VH_FUSED_H
#error "compile with -DVH_FUSED_H=<halfwidth>"
#endif

// Cache policy of the output stores: nt (aux = 2).  The result is not read again by this kernel, and stores that allocate
// in L2 push out the input rows neighbouring tiles share; measured on the tolerance kernel at 2048^3 (same box, alternating
// runs): 16.5 / 16.1 / 18.0 ms with default-policy stores, 15.3 / 15.2 / 14.9 ms with nt; 1-2 % at 1024^3, every variant.
constexpr int VH_FUSED_STORE_AUX = 2;

namespace vh {

namespace {

constexpr unsigned OOB = 0x7ffffff0u;  // byte offset beyond any descriptor: loads give 0, stores vanish

#ifdef VH_FUSED_STAMPS   // development build only (tools/build_variant.py): where a wave's cycles go, per phase of a plane step
__device__ unsigned long long g_fused_stamps[8];
#define VH_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define VH_STAMP(i) do {} while (0)
#endif

template <int H, int TX, int TY, int NT, int XV_, int YV_>
struct FusedCfg {
  static constexpr int W = 2 * H + 1;
  static constexpr int HX = TX + 2 * H;           // haloed tile width (even: TX even)
  static constexpr int HY = TY + 2 * H;
  static constexpr int SX = ((HX + 2 + 3) / 4) * 4; // LDS row stride in floats (16-byte rows)
  // Z-pass columns are dealt to threads over the PADDED tile (SX x HY): column id = tid + c * NT sits at LDS float
  // offset id, so the LDS address of a thread's c-th column is one base register plus a compile-time constant
  // (pad columns load 0.0f and land in the row padding)
  static constexpr int NCOL = SX * HY;
  static constexpr int NC = (NCOL + NT - 1) / NT;  // ring columns per thread
  // Y and X passes are WAVE-LOCAL: wave w owns output rows [w*RPW, (w+1)*RPW) of the tile, computes their
  // Y-filtered rows into sY and then reads only those rows back for the X pass (no workgroup barrier
  // between the two passes)
  static constexpr int NW = NT / 64;
  static constexpr int RPW = TY / NW;              // rows per wave
  static constexpr int YV = YV_;                   // adjacent x per lane in the Y pass (1 or 2)
  static constexpr int YTASKS = (HX / YV) * RPW;   // per wave: (x group, row)
  static constexpr int YROUNDS = (YTASKS + 63) / 64;
  // With column pairs (YV == 2) a last round that is at most half full is run on single columns instead
  // (two lanes per pair task): half the instructions for the same outputs
  static constexpr int YREM = YTASKS % 64;
  static constexpr bool YLAST_SINGLE = (YV == 2) && YREM > 0 && YREM <= 32;
  // ROW BLOCKING of the Y pass (YBLOCK): a lane filters BOTH rows of its wave for one unit of YV columns from ONE window
  // of 2H+2 source rows -- 2H+2 LDS reads for two output rows instead of 2(2H+1), and one LDS round trip instead of two:
  // the round trips in front of the Y and X arithmetic are what four waves per SIMD do not cover.  The
  // units beyond a multiple of 64 are filtered on single columns, one (column, row) per lane, as before.  Used when
  // that remainder fits one round.
  static constexpr int NUNIT = HX / YV;
  static constexpr int YB_FULL = NUNIT / 64;       // blocked rounds
  static constexpr int YB_REM = NUNIT % 64;        // units left for the single-column round
  static constexpr int YB_RCOLS = YB_REM * YV;     // their columns
  static constexpr bool YBLOCK = (RPW == 2) && YB_FULL >= 1 && YB_RCOLS * RPW <= 64;
  static constexpr int YB_ROUNDS = YB_FULL + (YB_REM > 0 ? 1 : 0);
  static constexpr int XV = XV_;                   // outputs per lane in the X pass (2 or 4)
  static constexpr int XTASKS = (TX / XV) * RPW;   // per wave
  static constexpr int XROUNDS = (XTASKS + 63) / 64;
  static constexpr int XWINV = (XV + 2 * H + XV - 1) / XV;  // XV-wide vector reads covering the X window
  static_assert(XV == 2 || XV == 4, "X pass vector width");
  static_assert(YV == 1 || YV == 2, "Y pass vector width");
  static_assert(HX % YV == 0 && TX % XV == 0, "tile width must be a multiple of the vector widths");
  static constexpr int SZ_FLOATS = NC * NT;        // >= HY * SX: slots past the tile are scratch
  static constexpr int SY_FLOATS = TY * SX;
};

template <int H>
struct TapsH {
  float t[2 * H + 1];
};

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1 (ring indices must be constants)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// Plain ds_read_b64: left to itself the compiler pairs two of these into ds_read2_b64, which
// the LDS serves at HALF the bytes per clock of the single form on gfx950 (MI355X_MICROARCH.md, LDS table: ds_read2_b64
// 8 cycles for 1 KB, ds_read_b64 2 cycles for 512 B) -- and the Y pass is bound by exactly these reads
// (profiles/r02_gauss_experiments.txt).  Volatile accesses are not merged.
#define VH_LDS __attribute__((address_space(3)))
__device__ __forceinline__ float lds_read_f1(const float* p) {
  return *(const volatile VH_LDS float*)(uintptr_t)(const VH_LDS void*)p;
}
__device__ __forceinline__ float2 lds_read_f2(const float* p) {
  typedef float v2f_ __attribute__((ext_vector_type(2)));
  const v2f_ v = *(const volatile VH_LDS v2f_*)(const VH_LDS void*)p;
  return make_float2(v.x, v.y);
}

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, 0));
}

// ZPASS == false: the Y and X passes only (the masked filter runs its Z pass, which applies the mask, as a separate
// kernel); DENOM: the result is a filtered mask denominator and the output is numer / result where result > 0, numer
// elsewhere (filter3d.hpp:986-996).
// FMA: the TOLERANCE MODE (context option gauss_fma): every tap is one fused multiply-add and the normaliser a
// multiplication by the per-lane reciprocal -- about half the vector instructions of the exact form.  Results are within
// a few ulp of the reference's (tests: 1e-5 of the field's scale) instead of bit-identical, so only callers whose output
// is a float field get it (ApplyGauss); the Gaussians of DoG/LoG feed index comparisons and never do.
template <int H, int TX, int TY, int NT, int XV_, int YV_, bool NORMALIZE, bool ISO, bool RAGGED, bool ZPASS = true,
          bool DENOM = false, bool FMA = false>
__global__ void __launch_bounds__(NT)
gauss_fused_kernel(const float* __restrict__ src, float* __restrict__ dst, TapsH<H> tz, TapsH<H> ty_,
                   TapsH<H> tx_, const float* __restrict__ Dx, const float* __restrict__ Dy,
                   const float* __restrict__ Dz, i64 dz_offset, int nx, int ny, int nz, int zchunk,
                   int tiles_x, int tiles_y, const float* __restrict__ minuend, float log_scale, float dz_int,
                   const float* __restrict__ numer) {
  typedef FusedCfg<H, TX, TY, NT, XV_, YV_> C;
  constexpr int W = C::W;
  static_assert(TY % (NT / 64) == 0, "tile rows must divide evenly among the waves");
  __shared__ __attribute__((aligned(16))) float sZ2[2][C::SZ_FLOATS];   // double-buffered: see the march loop
  __shared__ __attribute__((aligned(16))) float sY[C::SY_FLOATS];
  // Taps live in VGPRs: on gfx950 a VALU instruction with an SGPR operand issues at half the rate of the same
  // instruction on VGPRs (profiles/r02_microbench_valu.txt: v_mul_f32 v,s,v 37 T lane-ops/s against 70 T for v,v,v),
  // and every multiply of the three passes has a tap as operand.  Symmetric taps (checked on the host): tap |j| is
  // vt*[|j|].  The empty asm makes the copies opaque, so that the compiler cannot fold them back into the scalars.
  // Two instantiations have no registers to spare and keep scalar taps: unequal taps per axis / ragged rows with
  // the normaliser (all taps), and the same without it (Y and X taps).
  constexpr bool VZ = ZPASS && !(NORMALIZE && !ISO);   // Z taps in VGPRs
  constexpr bool VYX_OWN = !ZPASS;                      // the Y/X-only kernel: its own Y and X tap registers
  float vtz[H + 1], vty[H + 1], vtx[H + 1];
#pragma unroll
  for (int m = 0; m <= H; m++) {
    vtz[m] = tz.t[H + m];
    vty[m] = ty_.t[H + m];
    vtx[m] = tx_.t[H + m];
    if (VZ) asm volatile("" : "+v"(vtz[m]));
    if (VYX_OWN) {
      asm volatile("" : "+v"(vty[m]));
      asm volatile("" : "+v"(vtx[m]));
    }
  }
  auto tap_z = [&](int m) -> float { return VZ ? vtz[m] : tz.t[H + m]; };
  auto tap_y = [&](int jj) -> float {
    const int m = jj < H ? H - jj : jj - H;
    return VYX_OWN ? vty[m] : ((ISO && VZ) ? vtz[m] : (ISO ? tz.t[jj] : ty_.t[jj]));
  };
  auto tap_x = [&](int jj) -> float {
    const int m = jj < H ? H - jj : jj - H;
    return VYX_OWN ? vtx[m] : ((ISO && VZ) ? vtz[m] : (ISO ? tz.t[jj] : tx_.t[jj]));
  };

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
  const int plane_bytes = (int)(plane * 4);

  // ---- per-thread constants -------------------------------------------------------------------
  float ring[C::NC][W];     // running Z sums of the 2H+1 output planes in flight, per column
#pragma unroll
  for (int c = 0; c < C::NC; c++)
#pragma unroll
    for (int m = 0; m < W; m++) ring[c][m] = 0.0f;
  unsigned col_off[C::NC];  // byte offset of the column inside a plane (OOB outside the image and in the padding)
#pragma unroll
  for (int c = 0; c < C::NC; c++) {
    const int id = tid + c * NT;
    const int cy = id / C::SX, cx = id - cy * C::SX;
    const int gx = x0 - H + cx, gy = y0 - H + cy;
    const bool inside = cx < C::HX && cy < C::HY && gx >= 0 && gx < nx && gy >= 0 && gy < ny;
    col_off[c] = inside ? (unsigned)(gy * nx + gx) * 4u : OOB;
  }
  const int lds_base = 4 * tid;   // BYTE offset of column c in sZ: lds_base + 4 * NT * c
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NYR = C::YBLOCK ? C::YB_ROUNDS : C::YROUNDS;
  int y_off[NYR];    // LDS BYTE offset of (row y, column pair xp); -1: idle lane
  if constexpr (C::YBLOCK) {
#pragma unroll
    for (int r = 0; r < C::YB_FULL; r++) y_off[r] = 4 * ((wave * C::RPW) * C::SX + C::YV * (lane + 64 * r));   // both rows
    if (C::YB_REM > 0) {   // (column, row) per lane: lane = row * YB_RCOLS + column
      const int yy = lane / C::YB_RCOLS, xc = 64 * C::YB_FULL * C::YV + (lane - yy * C::YB_RCOLS);
      y_off[NYR - 1] = (lane < C::YB_RCOLS * C::RPW) ? 4 * ((wave * C::RPW + yy) * C::SX + xc) : -1;
    }
  } else {
#pragma unroll
    for (int r = 0; r < C::YROUNDS; r++) {
      const bool single = C::YLAST_SINGLE && r == C::YROUNDS - 1;
      const int task = single ? (r * 64 + (lane >> 1)) : (lane + r * 64);
      const int yy = task / (C::HX / C::YV), xp = task - yy * (C::HX / C::YV);
      y_off[r] = (task < C::YTASKS) ? 4 * ((wave * C::RPW + yy) * C::SX + C::YV * xp + (single ? (lane & 1) : 0)) : -1;
    }
  }
  int x_off[C::XROUNDS];
  unsigned o_off[C::XROUNDS];  // byte offset of the output quad inside a plane (OOB outside)
  unsigned e_off[RAGGED ? C::XROUNDS : 1];   // the same for a group cut by the end of its row (RAGGED only)
  int e_n[RAGGED ? C::XROUNDS : 1];          // and the number of its elements inside the row
  // per-lane normaliser constants, parked in LDS (the kernel is short of registers, not of LDS): group g = 2 * r (the
  // products Dx*Dy) and 2 * r + 1 (1 / ((Dx*Dy)*Dz) for the planes whose Dz is the interior value dz_int) of round r,
  // XV consecutive floats per thread
  __shared__ __attribute__((aligned(16))) float sK[NORMALIZE ? 2 * C::XROUNDS * C::XV * NT : 4];
  // wave_one[r]: on planes with the interior Dz every output of this wave's round r is divided by EXACTLY 1.0f -- the float sums
  // of a Gaussian's taps are 1 for about half of all sigmas (BlobDog's 24 filters at the bench's scales: 12), and then
  // (Dx*Dy)*Dz == 1 everywhere more than H voxels from a face.  x / 1.0f is x for every x (zeros, denormals, infinities and
  // NaN included), so such a wave skips the division altogether: a tenth of the kernel's vector instructions.
  bool wave_one[C::XROUNDS];
#pragma unroll
  for (int r = 0; r < C::XROUNDS; r++) {
    const int task = lane + r * 64;
    const int yy = task / (TX / C::XV), xq = task - yy * (TX / C::XV);
    const int y = wave * C::RPW + yy;
    const int gx = x0 + C::XV * xq, gy = y0 + y;
    const bool ok = (task < C::XTASKS) && gx < nx && gy < ny;
    x_off[r] = (task < C::XTASKS) ? 4 * (y * C::SX + C::XV * xq) : 0;   // bytes
    // nx a multiple of XV (RAGGED == false): an output group is inside the row or outside it as a whole.
    // Otherwise the group that straddles the end of a row is written (and its minuend read) element by element:
    // e_off/e_n describe it, and its vector offset is out of range like that of groups outside the image.
    const int nval = ok ? min(C::XV, nx - gx) : 0;
    o_off[r] = (nval == C::XV) ? (unsigned)(gy * nx + gx) * 4u : OOB;
    if (RAGGED) {
      e_n[r] = (nval < C::XV) ? nval : 0;
      e_off[r] = (nval > 0 && nval < C::XV) ? (unsigned)(gy * nx + gx) * 4u : OOB;
    }
    wave_one[r] = false;
    if (NORMALIZE) {
      const float dy = ok ? Dy[gy] : 1.0f;
      bool ones = true;
#pragma unroll
      for (int k = 0; k < C::XV; k++) {
        const float dxy = ((ok && gx + k < nx) ? Dx[gx + k] : 1.0f) * dy;  // (Dx*Dy) first, then *Dz (filter3d.hpp:1016-1018)
        sK[((2 * r) * NT + tid) * C::XV + k] = dxy;
        sK[((2 * r + 1) * NT + tid) * C::XV + k] = 1.0f / (dxy * dz_int);  // IEEE division: the correctly rounded reciprocal
        ones = ones && (dxy * dz_int == 1.0f);
      }
      wave_one[r] = __builtin_amdgcn_ballot_w64(!ones) == 0ull;   // (uniform; constant along the march)
    }
  }

  // first input plane of the march (the topmost one, ze-1+H); planes outside the image read as 0.0f
  const int ktop = ze - 1 + (ZPASS ? H : 0);
  const int nout = ze - zs;
  // Input planes in flight.  The exact kernel is VALU-bound and keeps ONE plane ahead (a second set was slower there,
  // profiles/r02_gauss_experiments.txt).  The tolerance kernel has half the arithmetic: at 2048^3 a workgroup has ~1.9 us per
  // plane, less than a loaded HBM round trip, so it keeps TWO planes ahead (PF = 2: plane k + 2 is requested into the
  // registers plane k has just left; the march is unrolled 2W times so that the set index stays a compile-time constant).
  constexpr int PF = (FMA && ZPASS) ? 2 : 1;
  float xin2[PF][C::NC];
  float (&xin)[C::NC] = xin2[0];
  auto request_plane = [&](int zn, bool wanted, int par = 0) {   // a zero-length descriptor fetches nothing and returns 0.0f
    const bool zin = wanted && zn >= 0 && zn < nz;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(src + (zin ? (i64)zn * plane : 0)), 0, zin ? plane_bytes : 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < C::NC; c++) xin2[par][c] = buf_load(rs, col_off[c]);
  };
  request_plane(ktop, true);
  if constexpr (PF == 2) request_plane(ktop - 1, true, 1);
#ifdef VH_FUSED_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif

  // ---- march DOWN along z, unrolled W times so that ring indices are compile-time -------------
  // Z pass in scatter form.  g[i] = sum_j t[j]*f[i-j] with j ascending means: for a fixed output plane i
  // the source planes contribute in DESCENDING order (i+H first, i-H last).  Marching down, the input
  // plane k = ktop - step adds its term t[j]*f[k] to the 2H+1 running sums of the outputs i = k+j in
  // exactly that order; the sum of output k+H is complete after this step.  The taps are symmetric
  // (checked on the host, bit for bit), so t[j]*f[k] and t[-j]*f[k] are the same IEEE product and only
  // H+1 multiplies are needed per input instead of 2H+1.  Slot of output i: (ktop + H - i) mod W
  // = (u + H - j) mod W at step number congruent to u.
  auto z_scatter_cols = [&](int u, int c_lo, int c_hi, int par = 0) {   // (columns [c_lo, c_hi); all arguments constants after inlining)
#pragma unroll
    for (int c = 0; c < C::NC; c++) {
      if (c < c_lo || c >= c_hi) continue;
      if constexpr (FMA) {
#pragma unroll
        for (int j = -H; j <= H; j++) {
          const int s = (u + H - j) % W;
          if (j == -H) ring[c][s] = tap_z(H) * xin2[par][c];
          else ring[c][s] = __builtin_fmaf(tap_z(j < 0 ? -j : j), xin2[par][c], ring[c][s]);
        }
        continue;
      }
      float pr[H + 1];
#pragma unroll
      for (int m = 0; m <= H; m++) pr[m] = tap_z(m) * xin[c];
#pragma unroll
      for (int j = -H; j <= H; j++) {
        const int s = (u + H - j) % W;
        if (j == -H) ring[c][s] = pr[H];                 // first term of the sum (see "signed zeros" below)
        else ring[c][s] = ring[c][s] + pr[j < 0 ? -j : j];
      }
    }
  };
  auto z_scatter = [&](int u, int par = 0) { z_scatter_cols(u, 0, C::NC, par); };
  // FILL: the Z pass of the NEXT plane is independent of the Y and X passes of this one, and every Y or X round starts
  // with an LDS round trip that four waves per SIMD do not cover.  At H <= 3 the ring update is cut into three
  // column groups that run right after the reads of the first Y round, the last Y round and the first X round have
  // been requested.
  constexpr int ZF0 = C::NC / 3, ZF1 = (2 * C::NC) / 3;
  // one term of a Y or X sum: acc + t * v (the reference's two roundings), or fused in the tolerance mode
  auto madd = [](float acc, float t, float v, bool first) -> float {
    if (first) return t * v;
    if constexpr (FMA) return __builtin_fmaf(t, v, acc);
    else return acc + t * v;
  };
  // Y and X passes of output plane z from the Z-filtered (or, without a Z pass, the source) tile sZ
  auto yx_passes = [&](int z, const float* sZ, auto&& zfill, int next_par = 0) {
        // Y pass: two adjacent x per lane; source rows y+2H (j=-H) down to y (j=+H)
#pragma unroll
        for (int r = 0; r < NYR; r++) {
          if (C::YBLOCK && r < C::YB_FULL) {
            // both rows of the wave for one column pair: window rows y+1+2H (v[0]) down to y (v[W]); the upper row's sum
            // runs over v[0..W-1], the lower row's over v[1..W], each with j ascending as the reference
            const float* base = reinterpret_cast<const float*>(reinterpret_cast<const char*>(sZ) + y_off[r]);
            if constexpr (C::YV == 2) {
              float2 v[W + 1];
#pragma unroll
              for (int jj = 0; jj <= W; jj++) v[jj] = lds_read_f2(base + (2 * H + 1 - jj) * C::SX);
              if (r == 0) {
                zfill(0);
                if (C::YB_REM == 0) zfill(1);
              }
              float a00 = 0.0f, a01 = 0.0f, a10 = 0.0f, a11 = 0.0f;
#pragma unroll
              for (int jj = 0; jj < W; jj++) {
                const float t = tap_y(jj);
                a10 = madd(a10, t, v[jj].x, jj == 0);
                a11 = madd(a11, t, v[jj].y, jj == 0);
                a00 = madd(a00, t, v[jj + 1].x, jj == 0);
                a01 = madd(a01, t, v[jj + 1].y, jj == 0);
              }
              *reinterpret_cast<float2*>(reinterpret_cast<char*>(sY) + y_off[r]) = make_float2(a00, a01);
              *reinterpret_cast<float2*>(reinterpret_cast<char*>(sY) + y_off[r] + 4 * C::SX) = make_float2(a10, a11);
            } else {
              float v[W + 1];
#pragma unroll
              for (int jj = 0; jj <= W; jj++) v[jj] = lds_read_f1(base + (2 * H + 1 - jj) * C::SX);
              if (r == 0) {
                zfill(0);
                if (C::YB_REM == 0) zfill(1);
              }
              float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
              for (int jj = 0; jj < W; jj++) {
                const float t = tap_y(jj);
                a1 = madd(a1, t, v[jj], jj == 0);
                a0 = madd(a0, t, v[jj + 1], jj == 0);
              }
              *reinterpret_cast<float*>(reinterpret_cast<char*>(sY) + y_off[r]) = a0;
              *reinterpret_cast<float*>(reinterpret_cast<char*>(sY) + y_off[r] + 4 * C::SX) = a1;
            }
          } else if (C::YBLOCK) {
            // the column pairs beyond the blocked rounds, one (column, row) per lane; every lane reads (idle lanes their
            // own first column) so that the fill work can stand between the reads and their use
            const float* base = reinterpret_cast<const float*>(reinterpret_cast<const char*>(sZ) + (y_off[r] >= 0 ? y_off[r] : 0));
            float a0 = 0.0f;
            float v[W];
#pragma unroll
            for (int jj = 0; jj < W; jj++) v[jj] = lds_read_f1(base + (2 * H - jj) * C::SX);
            zfill(1);
#pragma unroll
            for (int jj = 0; jj < W; jj++) a0 = madd(a0, tap_y(jj), v[jj], jj == 0);
            if (y_off[r] >= 0) *reinterpret_cast<float*>(reinterpret_cast<char*>(sY) + y_off[r]) = a0;
          } else if (y_off[r] >= 0) {
            const float* base = reinterpret_cast<const float*>(reinterpret_cast<const char*>(sZ) + y_off[r]);
            if (C::YV == 2 && !(C::YLAST_SINGLE && r == C::YROUNDS - 1)) {
              float a0 = 0.0f, a1 = 0.0f;
              float2 v[W];   // the whole window is requested before the first use: one LDS round trip per round
#pragma unroll
              for (int jj = 0; jj < W; jj++) v[jj] = lds_read_f2(base + (2 * H - jj) * C::SX);
#pragma unroll
              for (int jj = 0; jj < W; jj++) {
                const float t = tap_y(jj);
                a0 = madd(a0, t, v[jj].x, jj == 0);
                a1 = madd(a1, t, v[jj].y, jj == 0);
              }
              *reinterpret_cast<float2*>(reinterpret_cast<char*>(sY) + y_off[r]) = make_float2(a0, a1);
            } else {
              float a0 = 0.0f;
              float v[W];
#pragma unroll
              for (int jj = 0; jj < W; jj++) v[jj] = base[(2 * H - jj) * C::SX];   // (ds_read2_b32 pairs cost what two ds_read_b32 do)
#pragma unroll
              for (int jj = 0; jj < W; jj++) a0 = madd(a0, tap_y(jj), v[jj], jj == 0);
              *reinterpret_cast<float*>(reinterpret_cast<char*>(sY) + y_off[r]) = a0;
            }
          }
        }
        VH_STAMP(4);
        __builtin_amdgcn_wave_barrier();   // same wave wrote these sY rows: LDS ops of a wave execute in order
        // X pass: four adjacent outputs per lane; haloed source index x+2H (j=-H) down to x (j=+H)
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(dst + (i64)z * plane), 0, plane_bytes, 0x00020000);
        float dz = 1.0f;
        bool dz_is_int = false;
        if (NORMALIZE) {
          dz = Dz[z + dz_offset];
          dz_is_int = (dz == dz_int);   // uniform
        }
#pragma unroll
        for (int r = 0; r < C::XROUNDS; r++) {
          float v[C::XV * C::XWINV];
          typedef float v4f __attribute__((ext_vector_type(4)));
          typedef unsigned v4u __attribute__((ext_vector_type(4)));
          typedef float v2f __attribute__((ext_vector_type(2)));
          typedef unsigned v2u __attribute__((ext_vector_type(2)));
          // one output group of another volume (its minuend or numerator), with the same row-end handling
          auto load_group = [&](const float* vol, float m[C::XV]) {
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(vol + (i64)z * plane), 0, plane_bytes, 0x00020000);
            if (C::XV == 4) {
              const v4f mv = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rm, (int)o_off[r], 0, 0));
#pragma unroll
              for (int k = 0; k < 4; k++) m[k] = mv[k];
            } else {
              const v2f mv = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rm, (int)o_off[r], 0, 0));
              m[0] = mv[0]; m[1] = mv[1];
            }
            if (RAGGED) {
#pragma unroll
              for (int k = 0; k < C::XV - 1; k++) {
                const float e = buf_load(rm, k < e_n[r] ? e_off[r] + 4u * k : OOB);
                if (k < e_n[r]) m[k] = e;
              }
            }
          };
          // the minuend of a DoG/LoG second Gaussian is REQUESTED HERE, at the top of the round, and used at its end: asked for
          // where it is subtracted, every X round waited for a memory round trip (the minuend is the previous launch's
          // output, 4 GiB away in HBM) before it could store
          float mnd[C::XV];
          if (minuend) load_group(minuend, mnd);   // (wave-uniform)
          float dxy[C::XV], rcp_int[C::XV];   // requested together with the window (only this thread touches these LDS words)
          if (NORMALIZE) {   // one vector read each (XV consecutive floats per thread, 16-byte aligned for XV = 4)
            if constexpr (C::XV == 4) {
              const float4 q0 = *reinterpret_cast<const float4*>(&sK[((2 * r) * NT + tid) * 4]);
              const float4 q1 = *reinterpret_cast<const float4*>(&sK[((2 * r + 1) * NT + tid) * 4]);
              dxy[0] = q0.x; dxy[1] = q0.y; dxy[2] = q0.z; dxy[3] = q0.w;
              rcp_int[0] = q1.x; rcp_int[1] = q1.y; rcp_int[2] = q1.z; rcp_int[3] = q1.w;
            } else {
              const float2 q0 = *reinterpret_cast<const float2*>(&sK[((2 * r) * NT + tid) * 2]);
              const float2 q1 = *reinterpret_cast<const float2*>(&sK[((2 * r + 1) * NT + tid) * 2]);
              dxy[0] = q0.x; dxy[1] = q0.y;
              rcp_int[0] = q1.x; rcp_int[1] = q1.y;
            }
          }
          if (C::XV == 4) {
            const float4* base = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sY) + x_off[r]);
#pragma unroll
            for (int k = 0; k < C::XWINV; k++) {
              const float4 q = base[k];
              v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
            }
          } else {
            const float2* base = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(sY) + x_off[r]);
#pragma unroll
            for (int k = 0; k < C::XWINV; k++) {
              const float2 q = base[k];
              v[2 * k] = q.x; v[2 * k + 1] = q.y;
            }
          }
          if (r == 0) zfill(2);
          float a[C::XV];
#pragma unroll
          for (int k = 0; k < C::XV; k++) a[k] = 0.0f;
#pragma unroll
          for (int jj = 0; jj < W; jj++) {
            const float t = tap_x(jj);
#pragma unroll
            for (int k = 0; k < C::XV; k++) a[k] = madd(a[k], t, v[2 * H - jj + k], jj == 0);
          }
          // Signed zeros.  The reference starts every sum from +0.0 ("acc = 0; acc += term", filter1d.hpp:96-101); the
          // three passes above start from the first product instead, which saves one add per output and pass.  The two
          // differ only when every product of a sum is -0.0: the sum is then -0.0 here and +0.0 there.  A zero of either
          // sign contributes the same (zero) products to the next pass, so no non-zero value downstream can change, and a
          // final result can only be -0.0 where the reference has +0.0; adding +0.0 once, here, restores exactly that
          // (x + 0.0 == x for every other x).
          if (!FMA) {
#pragma unroll
            for (int k = 0; k < C::XV; k++) a[k] = a[k] + 0.0f;
          }
          if (NORMALIZE && dz_is_int && wave_one[r]) {
            // every divisor of this wave is exactly 1.0f: nothing to do (see wave_one)
          } else if (NORMALIZE && FMA) {
            // tolerance mode: interior planes multiply by the per-lane reciprocal of (Dx*Dy)*Dz, the others divide
            if (dz_is_int) {
#pragma unroll
              for (int k = 0; k < C::XV; k++) a[k] = a[k] * rcp_int[k];
            } else {
#pragma unroll
              for (int k = 0; k < C::XV; k++) a[k] = a[k] / (dxy[k] * dz);
            }
          } else if (NORMALIZE) {
            // dest /= (Dx*Dy)*Dz (filter3d.hpp:1016-1018), one correctly rounded division.  On planes with the
            // interior Dz the divisor is a per-lane constant whose correctly rounded reciprocal y is known, and
            // the quotient follows from two Newton corrections with exact FMA residuals (Markstein): q1 = q0 +
            // (a - d q0) y is a faithful quotient, q2 = q1 + (a - d q1) y is the correctly rounded one.  The
            // residuals are exact only while nothing under- or overflows, hence the range guard; everything
            // else (first/last H planes, zeros, extreme magnitudes) takes the full-range division.
            float d[C::XV];
#pragma unroll
            for (int k = 0; k < C::XV; k++) d[k] = dxy[k] * dz;
            bool fast = dz_is_int;
            if (fast) {
              float hi = __builtin_fmaxf(__builtin_fabsf(a[0]), __builtin_fabsf(a[1]));
              float lo = __builtin_fminf(__builtin_fabsf(a[0]), __builtin_fabsf(a[1]));
#pragma unroll
              for (int k = 2; k < C::XV; k++) {
                hi = __builtin_fmaxf(hi, __builtin_fabsf(a[k]));
                lo = __builtin_fminf(lo, __builtin_fabsf(a[k]));
              }
              fast = __builtin_amdgcn_ballot_w64(!((hi < 0x1p100f) && (lo >= 0x1p-100f))) == 0ull;
            }
            if (fast) {
#pragma unroll
              for (int k = 0; k < C::XV; k++) {
                const float y = rcp_int[k];
                const float q0 = a[k] * y;
                const float r0 = __builtin_fmaf(-d[k], q0, a[k]);
                const float q1 = __builtin_fmaf(r0, y, q0);
                const float r1 = __builtin_fmaf(-d[k], q1, a[k]);
                a[k] = __builtin_fmaf(r1, y, q1);
              }
            } else {
#pragma unroll
              for (int k = 0; k < C::XV; k++) a[k] = a[k] / d[k];
            }
          }
          if (DENOM) {
            // masked normalisation (filter3d.hpp:986-996): this kernel filtered the mask denominator; the
            // output is numer / denominator where the denominator is positive, numer elsewhere
            float m[C::XV];
            load_group(numer, m);
#pragma unroll
            for (int k = 0; k < C::XV; k++) a[k] = (a[k] > 0.0f) ? m[k] / a[k] : m[k];
          }
          if (minuend) {
            // DoG/LoG epilogue fused into the second Gaussian: out = (G_a - G_b) * scale with the two
            // roundings of filter3d.hpp:1387-1390 and :1495-1498 (wave-uniform branch)
#pragma unroll
            for (int k = 0; k < C::XV; k++) {
              const float dd = mnd[k] - a[k];
              a[k] = dd * log_scale;
            }
          }
          // vmcnt counts loads and stores in one queue: make the wait for the next input plane (requested
          // after the Z pass, long since arrived) happen BEFORE the first output store is issued, so that the
          // next Z pass does not have to drain that store to see its inputs
          VH_STAMP(5);
          if (r == 0) {
#pragma unroll
            for (int c = 0; c < C::NC; c++) asm volatile("" : "+v"(xin2[next_par][c]));   // (the set the next Z pass consumes)
          }
          if (C::XV == 4) {
            v4f out = {a[0], a[1], a[2], a[3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, out), ro, (int)o_off[r], 0, VH_FUSED_STORE_AUX);
          } else {
            v2f out = {a[0], a[1]};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, out), ro, (int)o_off[r], 0, VH_FUSED_STORE_AUX);
          }
          if (RAGGED) {
#pragma unroll
            for (int k = 0; k < C::XV - 1; k++)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a[k]), ro,
                                                    (int)(k < e_n[r] ? e_off[r] + 4u * k : OOB), 0, 0);
          }
        }
  };
  if constexpr (!ZPASS) {
    // no Z pass: plane z itself goes to the LDS tile (alternating buffers, one barrier per plane)
    for (int n = 0; n < nout; n++) {
      const int z = ze - 1 - n;
      float* sZ = sZ2[n & 1];
#pragma unroll
      for (int c = 0; c < C::NC; c++)
        *reinterpret_cast<float*>(reinterpret_cast<char*>(sZ) + lds_base + 4 * NT * c) = xin[c];
      request_plane(z - 1, n + 1 < nout);
      __syncthreads();
      yx_passes(z, sZ, [](int) {});
    }
    return;
  }
  // warm-up: the first 2H input planes complete no output of this chunk (each step waits for its plane;
  // a separate code region, so that the main loop below has a single wait state at its step boundaries)
  // (input plane number k = ktop - z, k = 0, 1, ...: consumed from set k % PF, which then takes plane k + PF;
  //  the last plane any output of this chunk needs is number nout - 1 + 2H)
  static_for<0, W - 1>([&](auto U) {
    constexpr int u = decltype(U)::value;
    z_scatter(u, u % PF);
    request_plane(ktop - u - PF, true, u % PF);
  });
  // Z pass of the step that completes output plane number n (z = ze-1-n; ring phase u = (n + W-1) % W, a constant at
  // every call site), its tile into LDS buffer n & 1, and the request for the next input plane, whose latency the
  // rest of the interval covers
  auto z_store = [&](auto U, int n, int par = 0) {   // the completed sums of plane n into its LDS buffer; request of the next input plane
    constexpr int u = decltype(U)::value;
    float* sZ = sZ2[n & 1];
#pragma unroll
    for (int c = 0; c < C::NC; c++)
      *reinterpret_cast<float*>(reinterpret_cast<char*>(sZ) + lds_base + 4 * NT * c) = ring[c][u];
    request_plane(ze - 1 - n - H - PF, n + PF < nout, par);   // plane number n + 2H + PF; the last one needed is nout - 1 + 2H
  };
  auto z_step = [&](auto U, int n, int par = 0) {   // consumes input plane number n + 2H (set par = (n + 2H) % PF = n % PF)
    constexpr int u = decltype(U)::value;
    z_scatter(u, par);
    z_store(U, n, par);
  };
  z_step(std::integral_constant<int, W - 1>{}, 0, 0);
  // Main loop, one workgroup barrier per plane.  Between two barriers every wave runs the Y and X passes of plane n
  // (reading buffer n & 1, complete since the barrier) AND the Z pass of plane n+1 (writing the other buffer, whose
  // readers finished before the barrier); the two pieces are independent.  (Letting the two halves of the workgroup take
  // them in OPPOSITE order, so that LDS-bound Y passes run beside VALU-bound Z passes, measured neutral at 1024^3 at 1.7x
  // the code: profiles/r02_gauss_experiments.txt.)  Unrolled W times (v) so that the ring indices are constants: plane
  // n+1 has phase v.
  for (int nb = 0; nb < nout; nb += PF * W) {
    static_for<0, PF * W>([&](auto V2) {
      constexpr int v2 = decltype(V2)::value;
      constexpr int v = v2 % W;               // ring phase of plane n + 1
      constexpr int npar = (v2 + 1) % PF;     // input set of plane n + 1 (nb is a multiple of PF)
      const std::integral_constant<int, v> V{};
      const int n = nb + v2;
      if (n < nout) {  // uniform across the workgroup
        VH_STAMP(0);
        __syncthreads();
        VH_STAMP(3);
        // (the Z pass, which updates the register ring, stands once in the code; the Y/X passes stand before and
        // after it and each wave runs one of the two copies)
        if (C::YBLOCK && H <= 3) {
          // the ring update of plane n+1 runs inside the Y/X passes of plane n, in three pieces behind their LDS requests.
          // Only for the small windows: the pieces need the next input plane EARLY in the interval, and at H >= 4 (one
          // 1024-thread workgroup per CU, 2.2 us per plane) that exposes the HBM latency the old order hides -- H = 2, 3:
          // -6...-9 %, H = 4: +-0, H = 5: +2.5 % (profiles/r02_gauss_experiments.txt)
          const bool more = n + 1 < nout;   // uniform
          yx_passes(ze - 1 - n, sZ2[n & 1], [&](int k) {
            if (more) z_scatter_cols(v, k == 0 ? 0 : (k == 1 ? ZF0 : ZF1), k == 0 ? ZF0 : (k == 1 ? ZF1 : C::NC), npar);
          }, npar);
          if (more) z_store(V, n + 1, npar);
          VH_STAMP(1);
        } else {
          yx_passes(ze - 1 - n, sZ2[n & 1], [](int) {}, npar);
          VH_STAMP(6);
          if (n + 1 < nout) z_step(V, n + 1, npar);
          VH_STAMP(1);
        }
      }
    });
  }
#ifdef VH_FUSED_STAMPS
  if (lane == 0 && blockIdx.x < 256) {
#pragma unroll
    for (int i = 0; i < 8; i++) atomicAdd(&g_fused_stamps[i], st_acc[i]);
  }
#endif
}

template <int H, int TX, int TY, int NT, int XV = 4, int YV = 2>
int launch_cfg(visfd_hip_ctx* ctx, const float* src, float* dst, i64 nx, i64 ny, i64 nz,
               const Taps& tx, const Taps& ty, const Taps& tz, const float* Dx, const float* Dy,
               const float* Dz, i64 dz_offset, bool normalize, const float* minuend, float log_scale,
               bool zpass = true, const float* numer = nullptr, bool fma = false) {
  TapsH<H> a, b, c;
  bool iso = true;
  for (int k = 0; k < 2 * H + 1; k++) {
    a.t[k] = tz.t[k]; b.t[k] = ty.t[k]; c.t[k] = tx.t[k];
    iso = iso && (std::memcmp(&tz.t[k], &ty.t[k], 4) == 0) && (std::memcmp(&tz.t[k], &tx.t[k], 4) == 0);
  }
  // Dz of a plane at least H away from both faces: the float sum of the Z taps in tap order (what
  // host_conv_ones produces there); planes with this Dz take the reciprocal-based division
  float dz_int = 0.0f;
  for (int k = 0; k < 2 * H + 1; k++) dz_int = dz_int + tz.t[k] * 1.0f;
  const int tiles_x = (int)((nx + TX - 1) / TX), tiles_y = (int)((ny + TY - 1) / TY);
  // z chunks: enough workgroups to fill the chip several times over, but marches long enough
  // to amortise the 2H-plane ring warm-up
  const i64 tiles = (i64)tiles_x * tiles_y;
  int per_cu = 2;   // sweep at 1024^3: 1-2 workgroups per CU are best (fewer ring warm-ups), 6 costs 2.5 %
  if (ctx->opt.gauss_wg_per_cu >= 1 && ctx->opt.gauss_wg_per_cu <= 64) per_cu = ctx->opt.gauss_wg_per_cu;   // tuning aid
  i64 want_chunks = ((i64)ctx->num_cus * per_cu + tiles - 1) / tiles;
  if (want_chunks < 1) want_chunks = 1;
  i64 zchunk = (nz + want_chunks - 1) / want_chunks;
  const i64 min_chunk = 12 * H;
  if (zchunk < min_chunk) zchunk = min_chunk;
  if (zchunk > nz) zchunk = nz;
  const i64 nchunks = (nz + zchunk - 1) / zchunk;
  const i64 nblk = tiles * nchunks;
  if (nblk > 0x7fffffffLL) return fail(VISFD_HIP_EINVAL, "volume too large for one launch");
  dim3 grid((unsigned)nblk), block(NT);
#define VH_GO(NORM, ISOV, RAG, ZP, DEN, ...)                                                     \
  gauss_fused_kernel<H, TX, TY, NT, XV, YV, NORM, ISOV, RAG, ZP, DEN, ##__VA_ARGS__><<<grid, block, 0, ctx->stream>>>(  \
      src, dst, a, b, c, Dx, Dy, Dz, dz_offset, (int)nx, (int)ny, (int)nz, (int)zchunk, tiles_x, tiles_y,  \
      minuend, log_scale, dz_int, numer)
  // anisotropic taps share the ragged-row instantiation (both are the uncommon cases)
  const bool ragged = (nx % XV) != 0;
  if (!zpass) { if (numer) VH_GO(false, false, true, false, true); else VH_GO(false, false, true, false, false); }
  else if (normalize && iso && !ragged && fma && !minuend) VH_GO(true, true, false, true, false, true);   // tolerance mode
  else if (normalize) { if (iso && !ragged) VH_GO(true, true, false, true, false); else VH_GO(true, false, true, true, false); }
  else                { if (iso && !ragged) VH_GO(false, true, false, true, false); else VH_GO(false, false, true, true, false); }
#undef VH_GO
  VH_HIP(hipGetLastError());
#ifdef VH_FUSED_STAMPS
  {
    unsigned long long st8[8], z8[8] = {};
    VH_HIP(hipStreamSynchronize(ctx->stream));
    VH_HIP(hipMemcpyFromSymbol(st8, HIP_SYMBOL(g_fused_stamps), sizeof(st8)));
    const double per = 1.0 / ((double)(nblk < 256 ? nblk : 256) * (NT / 64) * (double)(zchunk));
    fprintf(stderr, "[fused stamps H=%d] cycles per wave and plane: other %.0f | z-pass %.0f | lds write+request %.0f | barrier %.0f | y-pass %.0f | "
            "x-pass to store %.0f | stores+tail %.0f\n", H, st8[0] * per, st8[1] * per, st8[2] * per, st8[3] * per, st8[4] * per, st8[5] * per, st8[6] * per);
    VH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fused_stamps), z8, sizeof(z8)));
  }
#endif
  return VISFD_HIP_OK;
}

}  // namespace

#define VH_CAT2(a, b) a##b
#define VH_CAT(a, b) VH_CAT2(a, b)

// One entry point per compiled half-width: launch_gauss_fused_h<H>(..., cfg)
int VH_CAT(launch_gauss_fused_h, VH_FUSED_H)(visfd_hip_ctx* ctx, const float* src, float* dst, i64 nx,
                                            i64 ny, i64 nz, const Taps& tx, const Taps& ty,
                                            const Taps& tz, const float* Dx, const float* Dy,
                                            const float* Dz, i64 dz_offset, bool normalize, int cfg,
                                            const float* minuend, float log_scale, bool fma) {
  constexpr int H = VH_FUSED_H;
  (void)cfg;
  // tilings picked from sweeps on MI355X (1024^3; profiles/r01_gauss_tiling_sweep.txt): wide tiles cut the
  // halo recomputation of the Z and Y passes (the kernel is VALU-bound), until the register ring
  // (columns per thread x (2H+1)) no longer fits 128 VGPRs; with the 64-wide tile the X pass needs two
  // outputs per lane to fill its wave, and single-column Y tasks pack the rounds better
  if constexpr (H <= 3)
    return launch_cfg<H, 128, 16, 512, 4, 2>(ctx, src, dst, nx, ny, nz, tx, ty, tz, Dx, Dy, Dz, dz_offset, normalize, minuend, log_scale, true, nullptr, fma);
  else if constexpr (H <= 5)
    return launch_cfg<H, 128, 32, 1024, 4, 2>(ctx, src, dst, nx, ny, nz, tx, ty, tz, Dx, Dy, Dz, dz_offset, normalize, minuend, log_scale, true, nullptr, fma);
  else
    return launch_cfg<H, 64, 32, 1024, 2, 1>(ctx, src, dst, nx, ny, nz, tx, ty, tz, Dx, Dy, Dz, dz_offset, normalize, minuend, log_scale, true, nullptr, fma);
}

// The Y and X passes alone (no Z pass, no box normaliser), optionally with the masked-normalisation epilogue
// out = numer / result where result > 0 (numer elsewhere) and the DoG/LoG epilogue after it.
int VH_CAT(launch_gauss_fused_yx_h, VH_FUSED_H)(visfd_hip_ctx* ctx, const float* src, float* dst, i64 nx, i64 ny,
                                               i64 nz, const Taps& tx, const Taps& ty, const float* numer,
                                               const float* minuend, float log_scale) {
  constexpr int H = VH_FUSED_H;
  if constexpr (H <= 3)
    return launch_cfg<H, 128, 16, 512, 4, 2>(ctx, src, dst, nx, ny, nz, tx, ty, ty, nullptr, nullptr, nullptr, 0, false, minuend, log_scale, false, numer);
  else if constexpr (H <= 5)
    return launch_cfg<H, 128, 32, 1024, 4, 2>(ctx, src, dst, nx, ny, nz, tx, ty, ty, nullptr, nullptr, nullptr, 0, false, minuend, log_scale, false, numer);
  else
    return launch_cfg<H, 64, 32, 1024, 2, 1>(ctx, src, dst, nx, ny, nz, tx, ty, ty, nullptr, nullptr, nullptr, 0, false, minuend, log_scale, false, numer);
}

}  // namespace vh
