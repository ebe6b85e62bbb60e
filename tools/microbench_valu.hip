// microbench_valu.hip -- issue rate of scalar vs packed fp32 VALU ops on gfx950 (development aid).
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench_valu microbench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  v2 ab = {a, b};
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // 8 independent v_mul_f32
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    } else if (MODE == 1) {  // 4 independent v_pk_mul_f32 (= 8 lane-multiplies)
      asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ab));
    } else if (MODE == 2) {  // 8 x (mul + add) scalar
      asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %0\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %2\n"
                   "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %4\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %6\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    } else if (MODE == 3) {  // packed mul + packed add: 4 instr = 8 lane-ops
      asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %0\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %2\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ab));
    } else if (MODE == 4) {  // 8 v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    } else if (MODE == 5) {  // mul with SGPR operand
      asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                   "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
int run(const char* name, int lane_ops_per_iter, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
  float* out;
  CHECK(hipMalloc(&out, blocks * 256 * sizeof(float)));
  const int iters = 100000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<MODE><<<blocks, 256>>>(out, 1000, 1.0001f, 0.9999f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.9999f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  double laneops = (double)blocks * 256 * iters * lane_ops_per_iter;
  printf("%-34s waves/SIMD=%d: %8.3f ms  %7.2f T lane-ops/s\n", name, waves_per_simd, ms, laneops / ms / 1e9);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_mul_f32 x8", 8, w);
    run<1>("v_pk_mul_f32 x4 (8 lane-mul)", 8, w);
    run<2>("v_mul+v_add x4 pairs", 8, w);
    run<3>("v_pk_mul+v_pk_add x2 pairs", 8, w);
    run<4>("v_fma_f32 x8 (8 fma)", 8, w);
    run<5>("v_mul_f32 sgpr x8", 8, w);
  }
  return 0;
}
