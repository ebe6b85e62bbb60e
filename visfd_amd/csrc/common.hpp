// common.hpp -- context, workspace and error plumbing shared by the HIP translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/visfd_hip.h"

namespace vh {

typedef int64_t i64;

// ---- error reporting -------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define VH_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return vh::fail(_e == hipErrorOutOfMemory ? VISFD_HIP_ENOMEM : VISFD_HIP_EDEVICE,      \
                      std::string(#expr) + ": " + hipGetErrorString(_e));                    \
  } while (0)

#define VH_TRY(expr)           \
  do {                         \
    int _rc = (expr);          \
    if (_rc != VISFD_HIP_OK) return _rc; \
  } while (0)

#define VH_REQUIRE(cond, msg) \
  do {                        \
    if (!(cond)) return vh::fail(VISFD_HIP_EINVAL, msg); \
  } while (0)

// ---- workspace: named slots that only grow; freed by visfd_hip_trim/destroy -------------------
enum Slot {
  WS_A = 0, WS_B, WS_C, WS_D,      // full-volume float scratch
  WS_DEN_A, WS_DEN_B,              // masked-normalisation denominators
  WS_NORM,                         // Dx|Dy|Dz normaliser lines
  WS_LOG0, WS_LOG1, WS_LOG2,       // rolling LoG volumes of the blob detector
  WS_CAND, WS_COUNTER,             // candidate list + counters
  WS_SCANCNT,                      // counters of the pipelined blob scan: theirs alone (a scan's counts are read by the host after later stages
                                   // may have been queued, visfd_hip_blob_dog_begin_dev / _end)
  WS_HIST,                         // radix-select histograms
  WS_TVTAB,                        // tensor-voting lookup table
  WS_TVAUX,                        // tensor-voting auxiliaries
  WS_TVSCRATCH,                    // tensor voting: per-workgroup rings of compacted sender planes (exact kernel) / the launch's sender lists
  WS_TVLIST,                       // tolerance-mode tensor voting: row offsets of the sender lists
  WS_H2D_0, WS_H2D_1, WS_H2D_2, WS_H2D_3, WS_H2D_4,  // staging for the host-pointer face
  WS_NSLOTS
};

}  // namespace vh

// Tuning and test switches: read ONCE from the environment when the context is created (VISFD_HIP_<NAME>) and
// changed afterwards only through visfd_hip_set_option -- no getenv on any hot path.
struct visfd_hip_options {
  int gauss_3pass = 0;      // 1: the separable filter always takes its three single-axis passes
  int gauss_cfg = 0;        // development builds: alternative tilings of the single-sweep filter
  int gauss_wg_per_cu = 2;  // workgroups per CU the single-sweep filter cuts the volume into
  int tv_dense = 0;         // 1: tensor voting by the baseline kernel (csrc/tv.hip)
  int tv_fma = 0;           // 1: TOLERANCE MODE of tensor voting: fused multiply-adds, results within 1e-5 of the field's scale
                            //    instead of bit-identical (surfaces, exponent 2 or 4; everything else stays exact)
  int gauss_fma = 0;        // 1: TOLERANCE MODE of the single-sweep Gaussian (plain ApplyGauss only; DoG/LoG stay exact)
  int eig_f32 = 0;          // 1: TOLERANCE MODE of the device eigen solver: its one angle (atan2, sin, cos) in single precision --
                            //    eigenvalues and scores move by ~1 float ulp (eigen3.hpp); 0: the reference's double-precision angle
  int tv_zrun = 0;          // receiver planes per unit of work (0: default)
  int tv_no_replay = 0;     // 1: list every sender plane again for every receiver plane (nothing reused from the rings)
  int tv_exact_tiled = 0;   // 1: exact tensor voting always on tv_tiled.hip (the round-2 kernel), never on the exact form of tv_box.hip
  int tv_no_fold = 0;       // tests: tolerance-mode voting never folds the saliency into the listed normals (tv_box.hip: vote_fma)
  int tv_poison = 0;        // tests: NaN bit patterns in LDS, ring memory and the output before tensor voting runs (tv_box.hip)
  int tv_reserve_wg = 0;    // workgroup slots the persistent voting grid leaves free (slab runs: the halo transport's kernels)
  int tv_max_wg = 0;        // cap on the persistent grid (0: fill the chip); tests use it to make workgroups claim many units
  int64_t blob_test_cap = 0;   // pretend the pipelined blob scan's buffers hold this many entries (0: off)
  int debug = 0;
};

struct visfd_hip_ctx {
  visfd_hip_options opt;
  float* tv_table_dev = nullptr;   // cached vote table (tv_tiled.hip) and what it was built for
  float tv_table_key[2] = {0.0f, 0.0f};
  int tv_table_h = -1;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  void* slot_ptr[vh::WS_NSLOTS] = {};
  size_t slot_bytes[vh::WS_NSLOTS] = {};
  int num_cus = 256;
  hipStream_t aux_stream = nullptr;   // host copies that must not queue behind the main stream's kernels
};

namespace vh {

// returns a device buffer of at least `bytes` bytes in slot `s` (contents undefined)
int ws_get(visfd_hip_ctx* ctx, Slot s, size_t bytes, void** out);
template <typename T>
inline int ws(visfd_hip_ctx* ctx, Slot s, size_t count, T** out) {
  void* p = nullptr;
  int rc = ws_get(ctx, s, count * sizeof(T), &p);
  *out = static_cast<T*>(p);
  return rc;
}

inline int check_dims(i64 nx, i64 ny, i64 nz) {
  if (nx <= 0 || ny <= 0 || nz <= 0) return fail(VISFD_HIP_EINVAL, "image dimensions must be positive");
  return VISFD_HIP_OK;
}

// ---- taps carried in the kernel-argument segment (scalar loads, SGPR operands) ---------------
constexpr int MAX_HALFWIDTH = 64;
struct Taps {
  float t[2 * MAX_HALFWIDTH + 1];  // t[j + h], j = -h..h
  int h;
};

// row stride (in float4 entries) of the tiled kernel's vote table: 2h+1 rounded up to 4 modulo 8 (tv_tiled.hip: LDS banks)
inline int tv_padded_row(int h) {
  int sp = 2 * h + 1;
  while ((sp & 7) != 4) sp++;
  return sp;
}

// tolerance-mode vote table of tv_box.hip: a slice has 3 zero rows above and below its 2h+1 rows and rows of
// tv_box_row(h) entries -- at least 3 zero entries behind the 2h+1 of a row, 4 modulo 8 (LDS banks) -- behind 4 guard
// entries: entry (jy, jx) of slice jz at 4 + (jy + h + 3) * row + (jx + h); everything else is zero, so that the receivers
// of a 4 x 4 sub-patch a sender does not reach read a zero weight
inline int tv_box_row(int h) {
  int sp = 2 * h + 1 + 3;
  while ((sp & 7) != 4) sp++;
  return sp;
}
inline int tv_box_slice(int h) { return (2 * h + 1 + 6) * tv_box_row(h) + 8; }

// host-side arithmetic (taps.cpp)
void host_gauss_taps(float sigma, int h, float* t);
void host_conv_ones(i64 n, const float* t, int h, float* out);  // filter applied to a line of ones
int host_tv_halfwidth(float sigma, float cutoff);
void host_tv_tables(float sigma, int h, float* w, float* rhat);
float host_gengauss3d_peak(const float width[3], float m_exp, float ratio);

// ---- device stages (each in its own .hip; all asynchronous on ctx->stream) -------------------
struct SlabInfo {   // Z-slab placement for multi-GPU runs; whole volume: z_lo=0, nz_global=nz
  i64 z_lo;
  i64 nz_global;
};

int dev_separable3d(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask, i64 nx,
                    i64 ny, i64 nz, const float* tx, int hx, const float* ty, int hy,
                    const float* tz, int hz, bool normalize, SlabInfo slab, float* A_out,
                    // optional DoG/LoG epilogue: dst = (minuend - G(src)) * log_scale, fused into the
                    // single-sweep kernel when it applies (*epilogue_done tells whether it was)
                    const float* minuend = nullptr, float log_scale = 1.0f, bool* epilogue_done = nullptr,
                    // tolerance mode (option gauss_fma) for callers whose output is a float field; never with a minuend
                    bool fma = false);
// dst = (a - b) * scale  with two roundings (filter3d.hpp:1387-1390,1495-1498); scale==1: no multiply
int dev_sub_scale(visfd_hip_ctx* ctx, float* a_inout, const float* b, i64 n, float scale, bool do_scale);
// LocalFluctuations element-wise steps (filter3d.hpp:1776-1790 and :1819-1846)
int dev_sub_square(visfd_hip_ctx* ctx, const float* a, const float* b, float* out, i64 n);   // out = (a-b)*(a-b)
int dev_scale_clamp_sqrt(visfd_hip_ctx* ctx, float* a_inout, i64 n, float scale);           // a = sqrt(max(a*scale, 0))

int blob_scan_launch(visfd_hip_ctx* ctx, int set, hipEvent_t done, const float* lo, const float* mid, const float* hi,
                     const float* mask, i64 nx, i64 ny, i64 nz, float min_thr, float max_thr);
int blob_scan_collect(visfd_hip_ctx* ctx, int set, hipEvent_t done, hipStream_t aux, i64 nx, i64 ny, i64 nvox,
                      int scale_index,
                      float sigma, std::vector<visfd_hip_blob>* minima, std::vector<visfd_hip_blob>* maxima,
                      bool* overflow);
int dev_blob_scan(visfd_hip_ctx* ctx, const float* lo, const float* mid, const float* hi,
                  const float* mask, i64 nx, i64 ny, i64 nz, int scale_index, float sigma,
                  float min_thr, float max_thr, bool want_min, bool want_max,
                  std::vector<visfd_hip_blob>* minima, std::vector<visfd_hip_blob>* maxima);

int dev_hessian(visfd_hip_ctx* ctx, const float* smoothed, const float* mask, i64 nx, i64 ny, i64 nz,
                float sigma, float* grad_planar, float* hess_planar);
int dev_hessian_saliency(visfd_hip_ctx* ctx, const float* hess_planar, const float* mask, i64 nvox,
                         int order, float* saliency, float* dir_planar);
int dev_ridge_saliency_fused(visfd_hip_ctx* ctx, const float* smoothed, const float* mask, i64 nx,
                             i64 ny, i64 nz, float sigma, int order, float* saliency,
                             float* dir_planar);
// peak_img / peak_bg (nullable pair): score *= peak_img - peak_bg, the reference's optional peak-height factor
// (bin/filter_mrc/handlers.cpp:1698-1702 and :1883-1887, `-membrane-background`)
int dev_ridge_score(visfd_hip_ctx* ctx, const float* smoothed, const float* mask, i64 nx, i64 ny, i64 nz,
                    float sigma, int order, float* saliency, const float* peak_img = nullptr, const float* peak_bg = nullptr);
int dev_ridge_directions(visfd_hip_ctx* ctx, const float* smoothed, const float* saliency, i64 nx, i64 ny, i64 nz,
                         float sigma, int order, float* dir_planar);
int dev_diagonalize(visfd_hip_ctx* ctx, const float* m6_planar, float* out6_planar, i64 n, int order);
int dev_tensor_saliency(visfd_hip_ctx* ctx, const float* tensor_planar, const float* mask, i64 nvox,
                        int order, float* saliency, const float* peak_img = nullptr, const float* peak_bg = nullptr);

int dev_select_histogram(visfd_hip_ctx* ctx, const float* sal, const float* mask, i64 nvox, int pass,
                         uint32_t prefix, uint64_t* hist_host, uint64_t* n_unmasked_host);
int dev_select_histogram_todev(visfd_hip_ctx* ctx, const float* sal, const float* mask, i64 nvox, int pass, uint32_t prefix,
                               uint64_t* hist_dev);
int dev_apply_threshold(visfd_hip_ctx* ctx, float* sal, i64 nvox, float thr);
int dev_threshold_fraction(visfd_hip_ctx* ctx, float* sal, const float* mask, i64 nvox, float fraction,
                           float* thr_out);

int dev_tv_dense_stick(visfd_hip_ctx* ctx, const float* sal, const float* dir_planar,
                       float* tensor_planar, const float* mask_src, const float* mask_dst, i64 nx,
                       i64 ny, i64 nz, i64 z_out0, i64 z_out1, float sigma_tv, int exponent,
                       float cutoff, bool curves);

int dev_tv_weight_sum(visfd_hip_ctx* ctx, const float* sal, float* den, const float* mask_src, const float* mask_dst, i64 nx,
                      i64 ny, i64 nz, float sigma_tv, float cutoff);

// resample.hip (sizes are {nx, ny, nz}; offset nullable)
int dev_bin_array3d(visfd_hip_ctx* ctx, const float* src, const int64_t size_src[3], float* dst,
                    const int64_t size_dst[3], const int* offset);
int dev_unbin_array3d(visfd_hip_ctx* ctx, const float* src, const int64_t size_src[3], float* dst,
                      const int64_t size_dst[3], const int* offset);

// layout helpers for the host-pointer face
int dev_interleaved_to_planar(visfd_hip_ctx* ctx, const float* aos, float* planar, i64 n, int channels);
int dev_planar_to_interleaved(visfd_hip_ctx* ctx, const float* planar, float* aos, i64 n, int channels,
                              const float* mask /*nullable: only where mask!=0*/);

inline unsigned grid_for(i64 n, int block, i64 cap = (i64)1 << 30) {
  i64 g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace vh
