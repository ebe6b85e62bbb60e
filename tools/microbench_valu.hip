// microbench_valu.hip -- what the gfx950 SIMDs really issue per cycle (development aid, round 2).
//
// Every body is 256 straight-line VALU instructions per loop trip (scalar loop overhead < 2 %), run at
// exactly 1, 2, 4 or 8 waves per SIMD on every CU: one workgroup per CU (LDS-pinned) of 256/512/1024 threads,
// or two of 1024.  Each wave stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around the loop, so
// the report separates CYCLES PER INSTRUCTION from THE CLOCK THE CHIP HOLDS under that load:
//   cyc/inst/SIMD = (cycles of the slowest wave of a SIMD) / (instructions issued by all waves of that SIMD)
//   clock         = d(s_memtime) / d(s_memrealtime) * 100 MHz
//   T lane-ops/s  = lanes * instructions / wall time (HIP events), the number a kernel can be priced against.
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench_valu microbench_valu.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v2 __attribute__((ext_vector_type(2)));

// 16 instructions on 16 independent registers; OP has %0..%15 as destinations/sources and %16, %17 as constants
#define R16(OP) \
  OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7") \
  OP("%8") OP("%9") OP("%10") OP("%11") OP("%12") OP("%13") OP("%14") OP("%15")
#define X16 "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), \
            "+v"(x8), "+v"(x9), "+v"(x10), "+v"(x11), "+v"(x12), "+v"(x13), "+v"(x14), "+v"(x15)
#define P16 "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), \
            "+v"(p8), "+v"(p9), "+v"(p10), "+v"(p11), "+v"(p12), "+v"(p13), "+v"(p14), "+v"(p15)
#define D8  "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)

#define OP_MUL(r) "v_mul_f32 " r ", " r ", %16\n"
#define OP_ADD(r) "v_add_f32 " r ", " r ", %16\n"
#define OP_FMA(r) "v_fma_f32 " r ", " r ", %16, %17\n"
#define OP_MUL3(r) "v_mul_f32_e64 " r ", " r ", %16\n"
#define OP_PKMUL(r) "v_pk_mul_f32 " r ", " r ", %16\n"
#define OP_PKFMA(r) "v_pk_fma_f32 " r ", " r ", %16, %16\n"
#define OP_MULS(r) "v_mul_f32 " r ", %16, " r "\n"
#define OP_DOT4(r) "v_dot4_i32_i8 " r ", " r ", %16, %17\n"
#define OP_ALIGN(r) "v_alignbit_b32 " r ", " r ", %16, 31\n"
#define OP_AND(r) "v_and_b32 " r ", " r ", %16\n"
#define OP_MULINL(r) "v_mul_f32 " r ", 2.0, " r "\n"
#define OP_MULLIT(r) "v_mul_f32 " r ", 0x3f800347, " r "\n"
#define OP_ADDS(r) "v_add_f32 " r ", %16, " r "\n"
#define OP_MULS64(r) "v_mul_f32_e64 " r ", " r ", %16\n"
#define OP_FMAC(r) "v_fmac_f32 " r ", %16, %17\n"
#define OP_FFBL(r) "v_ffbl_b32 " r ", " r "\n"
#define OP_LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 5, %16\n"
#define OP_CMP(r) "v_cmp_eq_u32 vcc, 0, " r "\n"
#define OP_CNDMASK(r) "v_cndmask_b32 " r ", " r ", %16, vcc\n"
#define OP_SUBINL(r) "v_sub_f32 " r ", 1.0, " r "\n"
#define OP_DPP(r) "v_mov_b32_dpp " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_ADDDPP(r) "v_add_f32_dpp " r ", " r ", " r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_MAD24(r) "v_mad_u32_u24 " r ", " r ", %16, %17\n"
#define OP_ADDU(r) "v_add_u32 " r ", " r ", %16\n"
#define OP_BPERM(r) "ds_bpermute_b32 " r ", %16, " r "\n"
#define OP_SWIZ(r) "ds_swizzle_b32 " r ", " r " offset:0x041f\n"
#define OP_LDSB128(r) "ds_read_b128 %[q], %16\n"
#define OP_PERM(r) "v_perm_b32 " r ", " r ", %16, %17\n"
#define OP_BFE(r) "v_bfe_u32 " r ", " r ", 3, 5\n"
#define OP_MAX3(r) "v_max3_f32 " r ", " r ", %16, %17\n"
#define OP_READLANE(r) "v_readfirstlane_b32 s20, " r "\n"
#define OP_PKADD(r) "v_pk_add_f32 " r ", " r ", %16\n"
#define OP_SUBCO(r) "v_sub_co_u32 " r ", vcc, " r ", %16\n"
#define OP_CMPX(r) "v_cmp_gt_i32_e64 s[20:21], 0, " r "\n"
#define OP_CMPF(r) "v_cmp_lt_f32 vcc, " r ", %16\n"
#define OP_ASHR(r) "v_ashrrev_i32 " r ", 31, " r "\n"

enum Mode { MUL, ADD, FMA, MULADD, MUL_E64, PKMUL, PKFMA, MUL_SGPR, CHAIN1, CHAIN2, CHAIN4, DOT4, INT_AND,
            MUL_F64, FMA_F64, ADD_F64, VOTE_LDS,
            MUL_INLINE, MUL_LITERAL, ADD_SGPR, MUL_SGPR_E64, FMAC, DOT4_ONLY, ALIGNBIT_ONLY, FFBL, LSHL_ADD, CMP_VCC, CNDMASK,
            SUB_INLINE, MOV_DPP, ADD_DPP, MAD_U24, ADD_U32, BPERMUTE, SWIZZLE, PERM, BFE, MAX3, READFIRSTLANE, PKADD,
            SUB_CO, CMP_SGPRPAIR, CMP_F32, ASHR,
            BANK_SAME, BANK_DIFF, BANK_MIX, MUL_EXEC_LO32, MUL_EXEC_HI32, MUL_EXEC_LO16, MUL_EXEC_ALT, NMODES };
static const char* mode_name[NMODES] = {
    "v_mul_f32 (16 independent)", "v_add_f32 (16 independent)", "v_fma_f32 (16 independent)", "v_mul_f32/v_add_f32 alternating",
    "v_mul_f32_e64 (VOP3 encoding)", "v_pk_mul_f32 (2 lanes-ops each)", "v_pk_fma_f32 (2 fma each)", "v_mul_f32 SGPR operand",
    "v_mul_f32 1 dependent chain", "v_mul_f32 2 dependent chains", "v_mul_f32 4 dependent chains", "v_dot4_i32_i8/v_alignbit_b32",
    "v_and_b32 (16 independent)", "v_mul_f64 (8 independent)", "v_fma_f64 (8 independent)", "v_add_f64 (8 independent)",
    "vote mix: 35 f32 VALU + 2 ds_read_b128 + ds_read_b32",
    "v_mul_f32 inline constant 2.0", "v_mul_f32 32-bit literal", "v_add_f32 SGPR operand", "v_mul_f32_e64 SGPR operand", "v_fmac_f32 (VOP2 fma)",
    "v_dot4_i32_i8", "v_alignbit_b32", "v_ffbl_b32", "v_lshl_add_u32", "v_cmp_eq_u32 -> vcc", "v_cndmask_b32 (vcc)",
    "v_sub_f32 inline constant 1.0", "v_mov_b32_dpp row_shr:1", "v_add_f32_dpp quad_perm", "v_mad_u32_u24", "v_add_u32", "ds_bpermute_b32",
    "ds_swizzle_b32", "v_perm_b32", "v_bfe_u32", "v_max3_f32", "v_readfirstlane_b32", "v_pk_add_f32 (2 lane-ops each)",
    "v_sub_co_u32 -> vcc (carry out as a compare)", "v_cmp_gt_i32_e64 -> s[20:21]", "v_cmp_lt_f32 -> vcc", "v_ashrrev_i32 31",
    "v_add_f32 vD, vA, vB: A, B, D all = 0 mod 4", "v_add_f32 vD, vA, vB: A, B, D in three banks (mod 4)",
    "v_mul/v_add pairs, sources A = 0, B = 1 mod 4, 2 mod 4 dest",
    "v_mul_f32 with exec = lanes 0-31 only", "v_mul_f32 with exec = lanes 32-63 only", "v_mul_f32 with exec = lanes 0-15 only",
    "v_mul_f32 with exec = every other lane"};
// lane-operations per instruction (packed = 2) and instructions per loop trip
static const int mode_ops[NMODES] = {1, 1, 1, 1, 1, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,  1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
static const int mode_inst[NMODES] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 128, 128, 128, 280,
                                      256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256};

struct Stamp { unsigned long long c0, c1, r0, r1; unsigned hwid, pad; };

template <int MODE>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 8))) k(float* out, Stamp* stamps, int iters, float a, float b) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float x8 = x0 + 8, x9 = x0 + 9, x10 = x0 + 10, x11 = x0 + 11, x12 = x0 + 12, x13 = x0 + 13, x14 = x0 + 14, x15 = x0 + 15;
  v2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x8, x9}, p5 = {x10, x11}, p6 = {x12, x13}, p7 = {x14, x15};
  v2 p8 = p0 + 1.f, p9 = p1 + 1.f, p10 = p2 + 1.f, p11 = p3 + 1.f, p12 = p4 + 1.f, p13 = p5 + 1.f, p14 = p6 + 1.f, p15 = p7 + 1.f;
  double d0 = x0, d1 = x1, d2 = x2, d3 = x3, d4 = x4, d5 = x5, d6 = x6, d7 = x7;
  const v2 ab = {a, b};
  const double da = a;
  if (MODE == VOTE_LDS) {
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f + 1e-6f * i;
    __syncthreads();
  }
  const unsigned laddr = (threadIdx.x & 63) * 16;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if (MODE == MUL) {
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile(R16(OP_MUL) : X16 : "v"(a), "v"(b));
    } else if (MODE == ADD) {
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile(R16(OP_ADD) : X16 : "v"(a), "v"(b));
    } else if (MODE == FMA) {
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile(R16(OP_FMA) : X16 : "v"(a), "v"(b));
    } else if (MODE == MULADD) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        asm volatile(R16(OP_MUL) : X16 : "v"(a), "v"(b));
        asm volatile(R16(OP_ADD) : X16 : "v"(b), "v"(a));
      }
    } else if (MODE == MUL_E64) {
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile(R16(OP_MUL3) : X16 : "v"(a), "v"(b));
    } else if (MODE == PKMUL) {
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile(R16(OP_PKMUL) : P16 : "v"(ab), "v"(ab));
    } else if (MODE == PKFMA) {
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile(R16(OP_PKFMA) : P16 : "v"(ab), "v"(ab));
    } else if (MODE == MUL_SGPR) {
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile(R16(OP_MULS) : X16 : "s"(a), "s"(b));
    } else if (MODE == CHAIN1) {
#pragma unroll
      for (int u = 0; u < 16; u++)
        asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                     "v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                     "v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                     "v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                     : "+v"(x0) : "v"(a));
    } else if (MODE == CHAIN2) {
#pragma unroll
      for (int u = 0; u < 16; u++)
        asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n"
                     "v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n"
                     "v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n"
                     "v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n"
                     : "+v"(x0), "+v"(x1) : "v"(a));
    } else if (MODE == CHAIN4) {
#pragma unroll
      for (int u = 0; u < 16; u++)
        asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                     "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                     "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                     "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
    } else if (MODE == DOT4) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        asm volatile(R16(OP_DOT4) : X16 : "v"(a), "v"(b));
        asm volatile(R16(OP_ALIGN) : X16 : "v"(a), "v"(b));
      }
    } else if (MODE == INT_AND) {
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile(R16(OP_AND) : X16 : "v"(a), "v"(b));
    } else if (MODE == MUL_F64) {
#pragma unroll
      for (int u = 0; u < 16; u++)
        asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                     "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n" : D8 : "v"(da));
    } else if (MODE == FMA_F64) {
#pragma unroll
      for (int u = 0; u < 16; u++)
        asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                     "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n" : D8 : "v"(da));
    } else if (MODE == ADD_F64) {
#pragma unroll
      for (int u = 0; u < 16; u++)
        asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                     "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n" : D8 : "v"(da));
    } else if (MODE == VOTE_LDS) {
      // the LDS/VALU mix of one tensor vote (32 float operations, one 16-byte list entry + index word + one 16-byte
      // table entry); 8 votes per trip, reads one vote ahead
      typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int u = 0; u < 8; u++) {
        f4 e, t; float w;
        asm volatile("ds_read_b128 %0, %3\n ds_read_b32 %2, %3 offset:16\n ds_read_b128 %1, %3 offset:4096\n"
                     : "=v"(e), "=v"(t), "=v"(w) : "v"(laddr));
        asm volatile(R16(OP_MUL) : X16 : "v"(a), "v"(b));
        asm volatile(R16(OP_ADD) : X16 : "v"(b), "v"(a));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e), "+v"(t), "+v"(w));
        asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %3" : "+v"(x0) : "v"(e.x), "v"(t.x), "v"(w));
      }
    }
#define GEN(M, OP, A, B) else if (MODE == M) { _Pragma("unroll") for (int u = 0; u < 16; u++) asm volatile(R16(OP) : X16 : A, B : "vcc", "s20"); }
#define GENX(M, LO, HI) else if (MODE == M) { _Pragma("unroll") for (int u = 0; u < 16; u++) asm volatile( \
      "s_mov_b64 s[20:21], exec\n s_mov_b32 exec_lo, " LO "\n s_mov_b32 exec_hi, " HI "\n" R16(OP_MUL) "s_mov_b64 exec, s[20:21]\n" \
      : X16 : "v"(a), "v"(b) : "s20", "s21"); }
    GENX(MUL_EXEC_LO32, "-1", "0")
    GENX(MUL_EXEC_HI32, "0", "-1")
    GENX(MUL_EXEC_LO16, "0xffff", "0")
    GENX(MUL_EXEC_ALT, "0x55555555", "0x55555555")
    GEN(MUL_INLINE, OP_MULINL, "v"(a), "v"(b))
    GEN(MUL_LITERAL, OP_MULLIT, "v"(a), "v"(b))
    GEN(ADD_SGPR, OP_ADDS, "s"(a), "s"(b))
    GEN(MUL_SGPR_E64, OP_MULS64, "s"(a), "s"(b))
    GEN(FMAC, OP_FMAC, "v"(a), "v"(b))
    GEN(DOT4_ONLY, OP_DOT4, "v"(a), "v"(b))
    GEN(ALIGNBIT_ONLY, OP_ALIGN, "v"(a), "v"(b))
    GEN(FFBL, OP_FFBL, "v"(a), "v"(b))
    GEN(LSHL_ADD, OP_LSHLADD, "v"(a), "v"(b))
    GEN(CMP_VCC, OP_CMP, "v"(a), "v"(b))
    GEN(CNDMASK, OP_CNDMASK, "v"(a), "v"(b))
    GEN(SUB_INLINE, OP_SUBINL, "v"(a), "v"(b))
    GEN(MOV_DPP, OP_DPP, "v"(a), "v"(b))
    GEN(ADD_DPP, OP_ADDDPP, "v"(a), "v"(b))
    GEN(MAD_U24, OP_MAD24, "v"(a), "v"(b))
    GEN(ADD_U32, OP_ADDU, "v"(a), "v"(b))
    GEN(PERM, OP_PERM, "v"(a), "v"(b))
    GEN(BFE, OP_BFE, "v"(a), "v"(b))
    GEN(MAX3, OP_MAX3, "v"(a), "v"(b))
    GEN(READFIRSTLANE, OP_READLANE, "v"(a), "v"(b))
    else if (MODE == BANK_SAME) { _Pragma("unroll") for (int u = 0; u < 16; u++) asm volatile("v_add_f32 v16, v48, v68\n v_add_f32 v20, v52, v72\n v_add_f32 v24, v56, v76\n v_add_f32 v28, v60, v64\n v_add_f32 v32, v48, v68\n v_add_f32 v36, v52, v72\n v_add_f32 v40, v56, v76\n v_add_f32 v44, v60, v64\n v_add_f32 v16, v48, v68\n v_add_f32 v20, v52, v72\n v_add_f32 v24, v56, v76\n v_add_f32 v28, v60, v64\n v_add_f32 v32, v48, v68\n v_add_f32 v36, v52, v72\n v_add_f32 v40, v56, v76\n v_add_f32 v44, v60, v64" ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83"); }
    else if (MODE == BANK_DIFF) { _Pragma("unroll") for (int u = 0; u < 16; u++) asm volatile("v_add_f32 v16, v49, v70\n v_add_f32 v20, v53, v74\n v_add_f32 v24, v57, v78\n v_add_f32 v28, v61, v66\n v_add_f32 v32, v49, v70\n v_add_f32 v36, v53, v74\n v_add_f32 v40, v57, v78\n v_add_f32 v44, v61, v66\n v_add_f32 v16, v49, v70\n v_add_f32 v20, v53, v74\n v_add_f32 v24, v57, v78\n v_add_f32 v28, v61, v66\n v_add_f32 v32, v49, v70\n v_add_f32 v36, v53, v74\n v_add_f32 v40, v57, v78\n v_add_f32 v44, v61, v66" ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83"); }
    else if (MODE == BANK_MIX) { _Pragma("unroll") for (int u = 0; u < 16; u++) asm volatile("v_mul_f32 v18, v48, v69\n v_add_f32 v22, v52, v73\n v_mul_f32 v26, v56, v77\n v_add_f32 v30, v60, v65\n v_mul_f32 v34, v48, v69\n v_add_f32 v38, v52, v73\n v_mul_f32 v42, v56, v77\n v_add_f32 v46, v60, v65\n v_mul_f32 v18, v48, v69\n v_add_f32 v22, v52, v73\n v_mul_f32 v26, v56, v77\n v_add_f32 v30, v60, v65\n v_mul_f32 v34, v48, v69\n v_add_f32 v38, v52, v73\n v_mul_f32 v42, v56, v77\n v_add_f32 v46, v60, v65" ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83"); }
    else if (MODE == PKADD) { _Pragma("unroll") for (int u = 0; u < 16; u++) asm volatile(R16(OP_PKADD) : P16 : "v"(ab), "v"(ab)); }
    else if (MODE == SUB_CO) { _Pragma("unroll") for (int u = 0; u < 16; u++) asm volatile(R16(OP_SUBCO) : X16 : "v"(a), "v"(b) : "vcc"); }
    else if (MODE == CMP_SGPRPAIR) { _Pragma("unroll") for (int u = 0; u < 16; u++) asm volatile(R16(OP_CMPX) : X16 : "v"(a), "v"(b) : "s20", "s21"); }
    else if (MODE == CMP_F32) { _Pragma("unroll") for (int u = 0; u < 16; u++) asm volatile(R16(OP_CMPF) : X16 : "v"(a), "v"(b) : "vcc"); }
    else if (MODE == ASHR) { _Pragma("unroll") for (int u = 0; u < 16; u++) asm volatile(R16(OP_ASHR) : X16 : "v"(a), "v"(b)); }
    else if (MODE == BPERMUTE) { _Pragma("unroll") for (int u = 0; u < 16; u++) { asm volatile(R16(OP_BPERM) : X16 : "v"(laddr), "v"(b)); asm volatile("s_waitcnt lgkmcnt(0)" : X16); } }
    else if (MODE == SWIZZLE) { _Pragma("unroll") for (int u = 0; u < 16; u++) { asm volatile(R16(OP_SWIZ) : X16 : "v"(laddr), "v"(b)); asm volatile("s_waitcnt lgkmcnt(0)" : X16); } }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    Stamp s;
    s.c0 = c0; s.c1 = c1; s.r0 = r0; s.r1 = r1;
    s.hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    s.pad = 0;
    stamps[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s;
  }
  float r;
  if (MODE == PKMUL || MODE == PKFMA)
    r = p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + p8.x + p9.y + p10.x + p11.y + p12.x + p13.y + p14.x + p15.y;
  else if (MODE == MUL_F64 || MODE == FMA_F64 || MODE == ADD_F64)
    r = (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  else
    r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + x8 + x9 + x10 + x11 + x12 + x13 + x14 + x15;
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

typedef void (*kern_t)(float*, Stamp*, int, float, float);

static void run(int mode, kern_t fn, int waves_per_simd, int num_cus, float* out, Stamp* dstamps, FILE* f) {
  const int threads = waves_per_simd >= 4 ? 1024 : 256 * waves_per_simd;
  const int blocks_per_cu = waves_per_simd == 8 ? 2 : 1;
  const size_t lds = blocks_per_cu == 2 ? 72 * 1024 : 96 * 1024;   // pins exactly blocks_per_cu workgroups on a CU
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int blocks = num_cus * blocks_per_cu;
  const int iters = 2000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  // hold the load for >= 0.25 s first so that the clock the chip settles at is the one measured
  float total = 0.f, ms = 0.f;
  std::vector<float> times;
  int launches = 0;
  while (total < 250.f || times.size() < 5) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), lds, 0, out, dstamps, iters, 1.0001f, 0.9999f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    total += ms;
    launches++;
    if (total >= 250.f) times.push_back(ms);
  }
  std::sort(times.begin(), times.end());
  ms = times[times.size() / 2];
  const int waves = blocks * threads / 64;
  std::vector<Stamp> st(waves);
  CHECK(hipMemcpy(st.data(), dstamps, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
  std::vector<double> cyc(waves), clk(waves);
  for (int i = 0; i < waves; i++) {
    cyc[i] = (double)(st[i].c1 - st[i].c0);
    clk[i] = cyc[i] / (double)(st[i].r1 - st[i].r0) * 100e6;
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  const double inst_per_wave = (double)iters * mode_inst[mode];
  const double cyc_med = cyc[waves / 2];
  const double cpi_simd = cyc_med / (inst_per_wave * waves_per_simd);
  const double laneops = (double)waves * 64.0 * inst_per_wave * mode_ops[mode];
  char line[256];
  snprintf(line, sizeof line, "%-52s %d w/SIMD  %7.3f ms  cyc/inst/SIMD %5.2f  clock %5.3f GHz (min %5.3f)  %6.2f T lane-ops/s\n",
           mode_name[mode], waves_per_simd, ms, cpi_simd, clk[waves / 2] / 1e9, clk[0] / 1e9, laneops / ms / 1e9);
  fputs(line, stdout);
  if (f) fputs(line, f);
  fflush(stdout);
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int num_cus = prop.multiProcessorCount;
  FILE* f = argc > 1 ? fopen(argv[1], "w") : nullptr;
  char hdr[256];
  snprintf(hdr, sizeof hdr, "# %s, %d CUs, clockRate %d kHz; 256 unrolled instructions per loop trip, one LDS-pinned workgroup set per CU\n",
           prop.name, num_cus, prop.clockRate);
  fputs(hdr, stdout);
  if (f) fputs(hdr, f);
  float* out;
  Stamp* dstamps;
  CHECK(hipMalloc(&out, (size_t)num_cus * 2 * 1024 * sizeof(float)));
  CHECK(hipMalloc(&dstamps, (size_t)num_cus * 2 * 16 * sizeof(Stamp)));
  kern_t fns[NMODES] = {k<MUL>, k<ADD>, k<FMA>, k<MULADD>, k<MUL_E64>, k<PKMUL>, k<PKFMA>, k<MUL_SGPR>, k<CHAIN1>, k<CHAIN2>,
                        k<CHAIN4>, k<DOT4>, k<INT_AND>, k<MUL_F64>, k<FMA_F64>, k<ADD_F64>, k<VOTE_LDS>,
                        k<MUL_INLINE>, k<MUL_LITERAL>, k<ADD_SGPR>, k<MUL_SGPR_E64>, k<FMAC>, k<DOT4_ONLY>, k<ALIGNBIT_ONLY>, k<FFBL>,
                        k<LSHL_ADD>, k<CMP_VCC>, k<CNDMASK>, k<SUB_INLINE>, k<MOV_DPP>, k<ADD_DPP>, k<MAD_U24>, k<ADD_U32>, k<BPERMUTE>,
                        k<SWIZZLE>, k<PERM>, k<BFE>, k<MAX3>, k<READFIRSTLANE>, k<PKADD>, k<SUB_CO>, k<CMP_SGPRPAIR>, k<CMP_F32>, k<ASHR>, k<BANK_SAME>, k<BANK_DIFF>, k<BANK_MIX>, k<MUL_EXEC_LO32>, k<MUL_EXEC_HI32>, k<MUL_EXEC_LO16>, k<MUL_EXEC_ALT>};
  const int m0 = argc > 2 ? atoi(argv[2]) : 0;
  const int m1 = argc > 3 ? atoi(argv[3]) : NMODES;
  for (int m = m0; m < m1 && m < NMODES; m++)
    for (int w : {1, 2, 4, 8}) run(m, fns[m], w, num_cus, out, dstamps, f);
  if (f) fclose(f);
  return 0;
}
