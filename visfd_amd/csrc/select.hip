// select.hip -- exact global top-fraction threshold of the ridge saliency
// (reference bin/filter_mrc/handlers.cpp:1751-1797).
//
// The reference sorts all unmasked saliencies in descending order on one thread and reads entry
// floor(n*fraction).  Here the same order statistic is found exactly with a two-digit radix select
// over the order-preserving 32-bit key of the float (2 x 65536-bin histograms, 4 B/voxel/round),
// which also shards across GPUs: per-rank histograms are summed by the caller.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace vh {

namespace {

constexpr int BLOCK = 256;
constexpr int NBINS = 65536;

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

// pass 0: digit = key >> 16 over all unmasked voxels.  pass 1: digit = key & 0xffff over voxels
// whose key >> 16 == prefix.
__global__ void __launch_bounds__(BLOCK)
histogram_kernel(const float* __restrict__ sal, const float* __restrict__ mask, i64 n, int pass,
                 uint32_t prefix, unsigned long long* __restrict__ hist) {
  i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x;
  const i64 step = (i64)gridDim.x * BLOCK;
  for (; i < n; i += step) {
    bool use = !(mask && mask[i] == 0.0f);
    uint32_t digit = 0;
    if (use) {
      const uint32_t k = order_key(sal[i]);
      if (pass == 0) digit = k >> 16;
      else { use = (k >> 16) == prefix; digit = k & 0xffffu; }
    }
    // wave-level aggregation of the common "every lane hits the same bin" case
    const unsigned long long active = __ballot(use);
    if (active == 0) continue;
    const int leader = __ffsll((long long)active) - 1;
    const uint32_t lead_digit = __shfl(digit, leader);
    const unsigned long long same = __ballot(use && digit == lead_digit);
    if (same == active) {
      if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lead_digit], (unsigned long long)__popcll(active));
    } else if (use) {
      atomicAdd(&hist[digit], 1ULL);
    }
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

int dev_apply_threshold(visfd_hip_ctx* ctx, float* sal, i64 nvox, float thr) {
  const unsigned g = grid_for(nvox, BLOCK, (i64)ctx->num_cus * 32);
  apply_threshold_kernel<<<dim3(g), dim3(BLOCK), 0, ctx->stream>>>(sal, nvox, thr);
  VH_HIP(hipGetLastError());
  return VISFD_HIP_OK;
}

// Walk a 65536-bin histogram from the largest key down; returns the bin holding the k-th largest
// (0-based) element and rewrites k to the rank inside that bin.
static int pick_bin_descending(const std::vector<uint64_t>& h, uint64_t* k) {
  uint64_t seen = 0;
  for (int b = NBINS - 1; b >= 0; b--) {
    if (seen + h[b] > *k) { *k -= seen; return b; }
    seen += h[b];
  }
  return -1;
}

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
  const int hi = pick_bin_descending(h, &k);
  if (hi < 0) return fail(VISFD_HIP_EDEVICE, "radix select: inconsistent histogram");
  VH_TRY(dev_select_histogram(ctx, sal, mask, nvox, 1, (uint32_t)hi, h.data(), nullptr));
  const int lo = pick_bin_descending(h, &k);
  if (lo < 0) return fail(VISFD_HIP_EDEVICE, "radix select: inconsistent histogram");
  const float thr = key_to_float(((uint32_t)hi << 16) | (uint32_t)lo);
  if (thr_out) *thr_out = thr;
  return dev_apply_threshold(ctx, sal, nvox, thr);
}

}  // namespace vh
