// select.hip -- exact global top-fraction threshold of the ridge saliency
// (reference bin/filter_mrc/handlers.cpp:1751-1797).
//
// The reference sorts all unmasked saliencies in descending order on one thread and reads entry
// floor(n*fraction).  Here the same order statistic is found exactly with a three-digit radix
// select (11 + 11 + 10 bits) over the order-preserving 32-bit key of the float: each round is one
// 4 B/voxel sweep that builds a 2048-bin histogram in LDS per workgroup and merges it with a few
// global atomics.  Rounds also shard across GPUs: per-rank histograms are summed by the caller
// (SURVEY.md §8e).
#include <cmath>
#include <vector>

#include "common.hpp"

namespace vh {

namespace {

constexpr int BLOCK = 256;
constexpr int NBINS = 2048;

__device__ __forceinline__ uint32_t order_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float key_to_float(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// digit layout of the 32-bit key: round 0 = bits 31..21, round 1 = bits 20..10, round 2 = bits 9..0.
// `prefix` holds the already-selected higher digits (right-aligned); only keys matching it count.
__global__ void __launch_bounds__(BLOCK)
histogram_kernel(const float* __restrict__ sal, const float* __restrict__ mask, i64 n, int round,
                 uint32_t prefix, unsigned long long* __restrict__ hist) {
  __shared__ unsigned int lh[NBINS];
  for (int i = threadIdx.x; i < NBINS; i += BLOCK) lh[i] = 0;
  __syncthreads();
  const int shift = (round == 0) ? 21 : (round == 1) ? 10 : 0;
  const uint32_t dmask = (round == 2) ? 0x3ffu : 0x7ffu;
  const int pshift = (round == 0) ? 32 : (round == 1) ? 21 : 10;
  auto count = [&](float v) {
    const uint32_t k = order_key(v);
    if (round > 0 && (k >> pshift) != prefix) return;
    atomicAdd(&lh[(k >> shift) & dmask], 1u);
  };
  const i64 step = (i64)gridDim.x * BLOCK;
  // four values per load, two loads in flight (one dword per thread and trip ran at 3.3 TB/s); the tail and unaligned or
  // masked volumes one by one
  const i64 n4 = (!mask && (reinterpret_cast<uintptr_t>(sal) & 15u) == 0) ? (n >> 2) : 0;
  const float4* sal4 = reinterpret_cast<const float4*>(sal);
  i64 j = (i64)blockIdx.x * BLOCK + threadIdx.x;
  for (; j + step < n4; j += 2 * step) {
    const float4 a = sal4[j], b = sal4[j + step];
    count(a.x); count(a.y); count(a.z); count(a.w);
    count(b.x); count(b.y); count(b.z); count(b.w);
  }
  if (j < n4) {
    const float4 a = sal4[j];
    count(a.x); count(a.y); count(a.z); count(a.w);
  }
  for (i64 i = 4 * n4 + (i64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += step) {
    if (mask && mask[i] == 0.0f) continue;
    count(sal[i]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NBINS; i += BLOCK) {
    const unsigned int c = lh[i];
    if (c) atomicAdd(&hist[i], (unsigned long long)c);
  }
}

__global__ void __launch_bounds__(BLOCK)
apply_threshold_kernel(float* __restrict__ sal, i64 n, float thr) {
  i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x;
  const i64 step = (i64)gridDim.x * BLOCK;
  for (; i < n; i += step) {
    const float s = sal[i];
    if (s < thr) sal[i] = 0.0f;
  }
}

}  // namespace

int dev_select_histogram(visfd_hip_ctx* ctx, const float* sal, const float* mask, i64 nvox, int pass,
                         uint32_t prefix, uint64_t* hist_host, uint64_t* n_unmasked_host) {
  // per-workgroup counters are 32-bit: each workgroup sees at most nvox/grid + BLOCK elements
  unsigned long long* hist = nullptr;
  VH_TRY(ws(ctx, WS_HIST, (size_t)NBINS, &hist));
  hipStream_t st = ctx->stream;
  VH_HIP(hipMemsetAsync(hist, 0, sizeof(unsigned long long) * NBINS, st));
  const unsigned g = grid_for(nvox, BLOCK, (i64)ctx->num_cus * 32);
  histogram_kernel<<<dim3(g), dim3(BLOCK), 0, st>>>(sal, mask, nvox, pass, prefix, hist);
  VH_HIP(hipGetLastError());
  VH_HIP(hipMemcpyAsync(hist_host, hist, sizeof(uint64_t) * NBINS, hipMemcpyDeviceToHost, st));
  VH_HIP(hipStreamSynchronize(st));
  if (n_unmasked_host) {
    uint64_t tot = 0;
    for (int b = 0; b < NBINS; b++) tot += hist_host[b];
    *n_unmasked_host = tot;
  }
  return VISFD_HIP_OK;
}

// The same histogram left on the device (2048 counters, zeroed here first), asynchronous: multi-GPU callers all-reduce it
// there and fetch the sum once.
int dev_select_histogram_todev(visfd_hip_ctx* ctx, const float* sal, const float* mask, i64 nvox, int pass, uint32_t prefix,
                               uint64_t* hist_dev) {
  hipStream_t st = ctx->stream;
  VH_HIP(hipMemsetAsync(hist_dev, 0, sizeof(unsigned long long) * NBINS, st));
  const unsigned g = grid_for(nvox, BLOCK, (i64)ctx->num_cus * 32);
  histogram_kernel<<<dim3(g), dim3(BLOCK), 0, st>>>(sal, mask, nvox, pass, prefix, reinterpret_cast<unsigned long long*>(hist_dev));
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

int dev_apply_threshold(visfd_hip_ctx* ctx, float* sal, i64 nvox, float thr) {
  const unsigned g = grid_for(nvox, BLOCK, (i64)ctx->num_cus * 32);
  apply_threshold_kernel<<<dim3(g), dim3(BLOCK), 0, ctx->stream>>>(sal, nvox, thr);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

// Walk a histogram from the largest key down; returns the bin holding the k-th largest
// (0-based) element and rewrites k to the rank inside that bin.
static int pick_bin_descending(const std::vector<uint64_t>& h, uint64_t* k) {
  uint64_t seen = 0;
  for (int b = NBINS - 1; b >= 0; b--) {
    if (seen + h[b] > *k) { *k -= seen; return b; }
    seen += h[b];
  }
  return -1;
}

// the host steps of one radix round on an (all-reduced) histogram, for the multi-rank select of slab.hip
int select_pick_digit(const uint64_t* hist, uint64_t* k) {
  return pick_bin_descending(std::vector<uint64_t>(hist, hist + NBINS), k);
}
float select_key_to_float(uint32_t key) { return key_to_float(key); }

int dev_threshold_fraction(visfd_hip_ctx* ctx, float* sal, const float* mask, i64 nvox, float fraction,
                           float* thr_out) {
  std::vector<uint64_t> h(NBINS);
  uint64_t n_unmasked = 0;
  VH_TRY(dev_select_histogram(ctx, sal, mask, nvox, 0, 0, h.data(), &n_unmasked));
  // i = floor(n_voxels * fraction): size_t -> float product (handlers.cpp:1781)
  const float prod = (float)n_unmasked * fraction;
  uint64_t k = (uint64_t)std::floor(prod);
  if (n_unmasked == 0 || k >= n_unmasked)
    return fail(VISFD_HIP_EINVAL, "threshold fraction selects no voxel (the reference would read past its array)");
  const int d0 = pick_bin_descending(h, &k);
  if (d0 < 0) return fail(VISFD_HIP_EDEVICE, "radix select: inconsistent histogram");
  VH_TRY(dev_select_histogram(ctx, sal, mask, nvox, 1, (uint32_t)d0, h.data(), nullptr));
  const int d1 = pick_bin_descending(h, &k);
  if (d1 < 0) return fail(VISFD_HIP_EDEVICE, "radix select: inconsistent histogram");
  VH_TRY(dev_select_histogram(ctx, sal, mask, nvox, 2, ((uint32_t)d0 << 11) | (uint32_t)d1, h.data(), nullptr));
  const int d2 = pick_bin_descending(h, &k);
  if (d2 < 0) return fail(VISFD_HIP_EDEVICE, "radix select: inconsistent histogram");
  const float thr = key_to_float(((uint32_t)d0 << 21) | ((uint32_t)d1 << 10) | (uint32_t)d2);
  if (thr_out) *thr_out = thr;
  return dev_apply_threshold(ctx, sal, nvox, thr);
}

}  // namespace vh
