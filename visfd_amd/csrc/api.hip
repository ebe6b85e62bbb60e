// api.hip -- the extern "C" surface declared in include/visfd_hip.h.
// Device-pointer entry points orchestrate the stage functions; host-pointer entry points stage the
// caller's volumes through the context workspace (H2D, run, D2H) and are synchronous.
#include <chrono>
#include <cmath>
#include <cctype>
#include <cstdlib>
#include <limits>
#include <memory>
#include <vector>

#include "common.hpp"

namespace vh {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

int ws_get(visfd_hip_ctx* ctx, Slot s, size_t bytes, void** out) {
  if (bytes == 0) bytes = 16;
  if (ctx->slot_bytes[s] < bytes) {
    if (ctx->slot_ptr[s]) {
      // buffers may still be in use by queued kernels
      VH_HIP(hipStreamSynchronize(ctx->stream));
      VH_HIP(hipFree(ctx->slot_ptr[s]));
      ctx->slot_ptr[s] = nullptr;
      ctx->slot_bytes[s] = 0;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
      return fail(VISFD_HIP_ENOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed: " +
                                        hipGetErrorString(e));
    ctx->slot_ptr[s] = p;
    ctx->slot_bytes[s] = bytes;
  }
  *out = ctx->slot_ptr[s];
  return VISFD_HIP_OK;
}

namespace {

struct Staged {  // a host volume mirrored in a workspace slot
  float* d = nullptr;
};

int upload(visfd_hip_ctx* ctx, Slot s, const float* host, size_t count, float** dev) {
  if (!host) { *dev = nullptr; return VISFD_HIP_OK; }
  VH_TRY(ws(ctx, s, count, dev));
  VH_HIP(hipMemcpyAsync(*dev, host, sizeof(float) * count, hipMemcpyHostToDevice, ctx->stream));
  return VISFD_HIP_OK;
}
int download(visfd_hip_ctx* ctx, float* host, const float* dev, size_t count) {
  VH_HIP(hipMemcpyAsync(host, dev, sizeof(float) * count, hipMemcpyDeviceToHost, ctx->stream));
  VH_HIP(hipStreamSynchronize(ctx->stream));
  return VISFD_HIP_OK;
}

int halfwidths_from_ratio(const float sigma[3], float ratio, int hw[3]) {
  // filter3d.hpp:1240-1247: floor of the float product, at least 1
  for (int d = 0; d < 3; d++) {
    hw[d] = (int)std::floor(sigma[d] * ratio);
    if (hw[d] < 1) hw[d] = 1;
  }
  return VISFD_HIP_OK;
}

int gauss_dev(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask, i64 nx, i64 ny, i64 nz,
              const float sigma[3], const int hw[3], bool normalize, SlabInfo slab, float* A_out,
              const float* minuend = nullptr, float log_scale = 1.0f, bool* epilogue_done = nullptr, bool fma = false) {
  VH_REQUIRE(ctx && src && dst && sigma && hw, "null argument");
  std::vector<float> t[3];
  for (int d = 0; d < 3; d++) {
    VH_REQUIRE(sigma[d] >= 0.0f, "sigma must be non-negative");
    VH_REQUIRE(hw[d] >= 0 && hw[d] <= MAX_HALFWIDTH, "filter halfwidth must be in [0, 64]");
    t[d].resize(2 * hw[d] + 1);
    host_gauss_taps(sigma[d], hw[d], t[d].data());
  }
  return dev_separable3d(ctx, src, dst, mask, nx, ny, nz, t[0].data(), hw[0], t[1].data(), hw[1],
                         t[2].data(), hw[2], normalize, slab, A_out, minuend, log_scale, epilogue_done, fma);
}

// ApplyDog with a caller-provided temp volume (filter3d.hpp:1338-1402); with do_scale it is the body of
// ApplyLog (filter3d.hpp:1466-1498).  dst = G_a(src); then the second Gaussian writes
// (dst - G_b(src)) [* scale] straight into dst when the single-sweep kernel applies, else via tmp.
int dog_dev(visfd_hip_ctx* ctx, const float* src, float* dst, float* tmp, const float* mask, i64 nx, i64 ny,
            i64 nz, const float sa[3], const float sb[3], const int hw[3], float scale, bool do_scale,
            float* A, float* B) {
  const SlabInfo whole = {0, nz};
  VH_TRY(gauss_dev(ctx, src, dst, mask, nx, ny, nz, sa, hw, true, whole, A));
  {
    // the fused epilogue reads and writes the same element of dst in one thread: in-place is safe;
    // without scaling the multiplier is 1.0f, which is exact
    bool fused = false;
    VH_TRY(gauss_dev(ctx, src, dst, mask, nx, ny, nz, sb, hw, true, whole, B, dst, do_scale ? scale : 1.0f, &fused));
    if (fused) return VISFD_HIP_OK;
  }
  VH_TRY(gauss_dev(ctx, src, tmp, mask, nx, ny, nz, sb, hw, true, whole, B));
  return dev_sub_scale(ctx, dst, tmp, nx * ny * nz, scale, do_scale);
}

struct LogPlan {
  float sa[3], sb[3];
  int hw[3];
  float scale;
};
// ApplyLog parameter derivation (filter3d.hpp:1451-1464, :1493)
LogPlan plan_log(const float sigma[3], float delta, float ratio) {
  LogPlan p;
  for (int d = 0; d < 3; d++) {
    p.sa[d] = (float)(sigma[d] * (1.0 - 0.5 * delta));
    p.sb[d] = (float)(sigma[d] * (1.0 + 0.5 * delta));
    p.hw[d] = (int)std::floor(ratio * std::fmax(p.sa[d], p.sb[d]));
  }
  p.scale = (float)(1.0 / (delta * delta));
  return p;
}

int log_dev(visfd_hip_ctx* ctx, const float* src, float* dst, float* tmp, const float* mask, i64 nx, i64 ny,
            i64 nz, const float sigma[3], float delta, float ratio, float* A, float* B) {
  const LogPlan p = plan_log(sigma, delta, ratio);
  for (int d = 0; d < 3; d++)
    VH_REQUIRE(p.hw[d] >= 0 && p.hw[d] <= MAX_HALFWIDTH, "LoG filter halfwidth must be in [0, 64]");
  float a = 0, b = 0;
  VH_TRY(dog_dev(ctx, src, dst, tmp, mask, nx, ny, nz, p.sa, p.sb, p.hw, p.scale, true, &a, &b));
  if (A) *A = a * p.scale;   // filter3d.hpp:1502-1505
  if (B) *B = b * p.scale;
  return VISFD_HIP_OK;
}

// BlobDog in two halves (visfd_hip_blob_dog_begin_dev / _end): `begin` queues every filter and scan and collects the lists
// of all scales but the last few; `end` collects those, repeats overflowed scales, merges and hands the lists over.  A caller
// that has more device work for the same stream (the membrane stage of a pipeline) queues it between the two: the device
// then goes from the last scan straight into that work instead of idling through the host's list handling (6-9 ms at
// 1024^3 -- and an idle MI355X took up to 25 ms more to start the next kernel).
struct BlobJob {
  visfd_hip_ctx* ctx = nullptr;
  const float* src = nullptr;
  const float* mask = nullptr;
  i64 nx = 0, ny = 0, nz = 0;
  std::vector<float> sigma;
  float asp[3] = {1.0f, 1.0f, 1.0f};
  float delta = 0, ratio = 0, min_thr = 0, max_thr = 0, scan_min = 0, scan_max = 0;
  bool use_ratios = false, can_scan = false, merged = false;
  static constexpr int NSET = 3;
  hipEvent_t ev[NSET] = {nullptr, nullptr, nullptr};
  std::vector<std::vector<visfd_hip_blob>> smin, smax;   // lists per middle scale (output order is scale order, feature.hpp:236-358)
  std::vector<int> redo;                                  // scales whose buffers overflowed in the pipelined scan
  int pending_first = 0, pending_n = 0;                   // middle scales whose scans are queued but not collected yet (set: scale % NSET)
  std::vector<visfd_hip_blob> mins, maxs;                 // the merged lists (after `merged`)
  std::chrono::steady_clock::time_point t_start;
  double since() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); }
  ~BlobJob() {
    for (int k = 0; k < NSET; k++)
      if (ev[k]) (void)hipEventDestroy(ev[k]);
  }
  int collect(int scale) {
    bool overflow = false;
    VH_TRY(blob_scan_collect(ctx, scale % NSET, ev[scale % NSET], ctx->aux_stream, nx, ny, nx * ny * nz, scale, sigma[(size_t)scale],
                             &smin[(size_t)scale], &smax[(size_t)scale], &overflow));
    if (overflow) redo.push_back(scale);
    return VISFD_HIP_OK;
  }
};

int blob_dog_begin(visfd_hip_ctx* ctx, const float* src, const float* mask, i64 nx, i64 ny, i64 nz,
                   const float* blob_sigma, int n_sigma, const float* aspect, float delta, float ratio,
                   float min_thr, float max_thr, bool use_ratios, BlobJob** job_out) {
  VH_REQUIRE(ctx && src && (blob_sigma || n_sigma == 0) && job_out, "null argument");
  VH_REQUIRE(n_sigma >= 0, "negative scale count");
  VH_TRY(check_dims(nx, ny, nz));
  *job_out = nullptr;
  const i64 n = nx * ny * nz;
  const float inf = std::numeric_limits<float>::infinity();
  float* vol[3];
  VH_TRY(ws(ctx, WS_LOG0, (size_t)n, &vol[0]));
  VH_TRY(ws(ctx, WS_LOG1, (size_t)n, &vol[1]));
  VH_TRY(ws(ctx, WS_LOG2, (size_t)n, &vol[2]));
  float* tmp = nullptr;
  VH_TRY(ws(ctx, WS_C, (size_t)n, &tmp));
  std::unique_ptr<BlobJob> J(new BlobJob);
  J->ctx = ctx; J->src = src; J->mask = mask; J->nx = nx; J->ny = ny; J->nz = nz;
  J->sigma.assign(blob_sigma, blob_sigma + n_sigma);
  if (aspect) for (int d = 0; d < 3; d++) J->asp[d] = aspect[d];
  J->delta = delta; J->ratio = ratio; J->min_thr = min_thr; J->max_thr = max_thr; J->use_ratios = use_ratios;
  // running thresholds: absolute mode applies them in the scan (strict, feature.hpp:270-291);
  // ratio mode keeps every candidate and prunes at the end (feature.hpp:362-417), see header.
  J->scan_min = use_ratios ? inf : min_thr;
  J->scan_max = use_ratios ? -inf : max_thr;
  J->t_start = std::chrono::steady_clock::now();
  // The scan of scale k-1 is queued right behind the filters of scale k, and its list is fetched (auxiliary stream)
  // and sorted on the host while the GPU already filters scales k+1 and k+2 (three buffer sets: the host may fall two scales
  // -- ~20 ms of device work at 1024^3 -- behind before the device runs dry).
  J->can_scan = nx >= 3 && ny >= 3 && nz >= 3;
  if (J->can_scan && (nx >= (1LL << 31) || ny >= (1LL << 31) || nz >= (1LL << 31))) return fail(VISFD_HIP_EINVAL, "dimension too large");
  if (!ctx->aux_stream) VH_HIP(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
  for (int k = 0; k < BlobJob::NSET; k++) VH_HIP(hipEventCreateWithFlags(&J->ev[k], hipEventDisableTiming));
  J->smin.resize((size_t)std::max(n_sigma, 1));
  J->smax.resize((size_t)std::max(n_sigma, 1));
  for (int ir = 0; ir < n_sigma; ir++) {
    const float sg[3] = {blob_sigma[ir] * J->asp[0], blob_sigma[ir] * J->asp[1], blob_sigma[ir] * J->asp[2]};
    VH_TRY(log_dev(ctx, src, vol[ir % 3], tmp, mask, nx, ny, nz, sg, delta, ratio, nullptr, nullptr));
    if (ir < 2 || !J->can_scan) continue;
    VH_TRY(blob_scan_launch(ctx, (ir - 1) % BlobJob::NSET, J->ev[(ir - 1) % BlobJob::NSET], vol[(ir - 2) % 3], vol[(ir - 1) % 3],
                            vol[ir % 3], mask, nx, ny, nz, J->scan_min, J->scan_max));
    if (J->pending_n == 0) J->pending_first = ir - 1;
    J->pending_n++;
    if (J->pending_n == BlobJob::NSET) {   // every buffer set is in use: the oldest list now (its scan was queued two scales ago)
      VH_TRY(J->collect(J->pending_first));
      J->pending_first++;
      J->pending_n--;
    }
  }
  if (ctx->opt.debug) fprintf(stderr, "[blob_dog] everything queued at %.1f ms\n", J->since());
  *job_out = J.release();
  return VISFD_HIP_OK;
}

// Collects what `begin` left, merges, and copies out.  VISFD_HIP_ECAPACITY leaves the job alive (the counts are returned: call
// again with room for them); every other outcome frees it.
int blob_dog_end(BlobJob* job, visfd_hip_blob* minima, int64_t min_cap, int64_t* n_min, visfd_hip_blob* maxima, int64_t max_cap,
                 int64_t* n_max) {
  VH_REQUIRE(job && n_min && n_max, "null argument");
  std::unique_ptr<BlobJob> J(job);
  visfd_hip_ctx* ctx = J->ctx;
  const float inf = std::numeric_limits<float>::infinity();
  if (!J->merged) {
    while (J->pending_n > 0) {
      VH_TRY(J->collect(J->pending_first));
      J->pending_first++;
      J->pending_n--;
    }
    if (ctx->opt.debug) fprintf(stderr, "[blob_dog] last list collected at %.1f ms\n", J->since());
    // a candidate or survivor buffer overflowed (dense extrema): those scales again, one at a time, with buffers that grow
    // (the three LoG volumes of the scale are filtered again; the other scales keep their lists)
    if (!J->redo.empty()) {
      const i64 n = J->nx * J->ny * J->nz;
      float* vol[3];
      VH_TRY(ws(ctx, WS_LOG0, (size_t)n, &vol[0]));
      VH_TRY(ws(ctx, WS_LOG1, (size_t)n, &vol[1]));
      VH_TRY(ws(ctx, WS_LOG2, (size_t)n, &vol[2]));
      float* tmp = nullptr;
      VH_TRY(ws(ctx, WS_C, (size_t)n, &tmp));
      for (int sc : J->redo) {
        for (int k = 0; k < 3; k++) {
          const int ir = sc - 1 + k;
          const float sg[3] = {J->sigma[(size_t)ir] * J->asp[0], J->sigma[(size_t)ir] * J->asp[1], J->sigma[(size_t)ir] * J->asp[2]};
          VH_TRY(log_dev(ctx, J->src, vol[k], tmp, J->mask, J->nx, J->ny, J->nz, sg, J->delta, J->ratio, nullptr, nullptr));
        }
        J->smin[(size_t)sc].clear();
        J->smax[(size_t)sc].clear();
        VH_TRY(dev_blob_scan(ctx, vol[0], vol[1], vol[2], J->mask, J->nx, J->ny, J->nz, sc, J->sigma[(size_t)sc], J->scan_min,
                             J->scan_max, true, true, &J->smin[(size_t)sc], &J->smax[(size_t)sc]));
      }
      J->redo.clear();
    }
    std::vector<visfd_hip_blob>& mins = J->mins;
    std::vector<visfd_hip_blob>& maxs = J->maxs;
    for (auto& v : J->smin) mins.insert(mins.end(), v.begin(), v.end());
    for (auto& v : J->smax) maxs.insert(maxs.end(), v.begin(), v.end());
    J->smin.clear();
    J->smax.clear();
    // Ratio mode with max_thr = -inf: the reference's scan compares score > (-inf) * (its running best, initially -1) = +inf in
    // every thread, so it never records a maximum (feature.hpp:286-289) -- deterministically none.
    if (J->use_ratios && J->max_thr == -inf) maxs.clear();
    if ((J->min_thr != inf) || (J->max_thr != -inf)) {
      float tmin = J->min_thr, tmax = J->max_thr;
      if (J->use_ratios) {
        float gmin = 1.0f, gmax = -1.0f;  // feature.hpp:122-123
        for (auto& b : mins) if (b.score < gmin) gmin = b.score;
        for (auto& b : maxs) if (b.score > gmax) gmax = b.score;
        tmin = J->min_thr * gmin;   // feature.hpp:369-372, unconditionally: +inf * (negative best) = -inf keeps no minimum
        tmax = J->max_thr * gmax;
      }
      std::vector<visfd_hip_blob> a, b;
      for (auto& m : mins) if (m.score <= tmin) a.push_back(m);
      for (auto& m : maxs) if (m.score >= tmax) b.push_back(m);
      mins.swap(a);
      maxs.swap(b);
    }
    J->merged = true;
    if (ctx->opt.debug) fprintf(stderr, "[blob_dog] lists merged at %.1f ms\n", J->since());
  }
  *n_min = (int64_t)J->mins.size();
  *n_max = (int64_t)J->maxs.size();
  if ((int64_t)J->mins.size() > min_cap || (int64_t)J->maxs.size() > max_cap) {
    J.release();   // kept: the caller comes back with room for the counts just returned
    return fail(VISFD_HIP_ECAPACITY, "blob list capacity too small");
  }
  VH_REQUIRE((minima || J->mins.empty()) && (maxima || J->maxs.empty()), "null list array");
  for (size_t i = 0; i < J->mins.size(); i++) minima[i] = J->mins[i];
  for (size_t i = 0; i < J->maxs.size(); i++) maxima[i] = J->maxs[i];
  if (ctx->opt.debug) fprintf(stderr, "[blob_dog] lists copied out at %.1f ms\n", J->since());
  return VISFD_HIP_OK;
}

int blob_dog_dev(visfd_hip_ctx* ctx, const float* src, const float* mask, i64 nx, i64 ny, i64 nz,
                 const float* blob_sigma, int n_sigma, const float* aspect, float delta, float ratio,
                 float min_thr, float max_thr, bool use_ratios, visfd_hip_blob* minima, int64_t min_cap,
                 int64_t* n_min, visfd_hip_blob* maxima, int64_t max_cap, int64_t* n_max) {
  VH_REQUIRE(ctx && src && blob_sigma && n_min && n_max, "null argument");
  BlobJob* job = nullptr;
  VH_TRY(blob_dog_begin(ctx, src, mask, nx, ny, nz, blob_sigma, n_sigma, aspect, delta, ratio, min_thr, max_thr, use_ratios, &job));
  const int rc = blob_dog_end(job, minima, min_cap, n_min, maxima, max_cap, n_max);
  if (rc == VISFD_HIP_ECAPACITY) {   // (the one-call form has no second chance: as before, the counts come back with the error
    //  and, as before, the lists' first min_cap / max_cap records)
    for (int64_t i = 0; i < (int64_t)job->mins.size() && i < min_cap; i++) minima[i] = job->mins[i];
    for (int64_t i = 0; i < (int64_t)job->maxs.size() && i < max_cap; i++) maxima[i] = job->maxs[i];
    delete job;
  }
  return rc;
}

int calc_hessian_dev(visfd_hip_ctx* ctx, const float* src, float* grad, float* hess, const float* mask,
                     i64 nx, i64 ny, i64 nz, float sigma, float ratio) {
  VH_REQUIRE(ctx && src, "null argument");
  VH_TRY(check_dims(nx, ny, nz));
  const int hwv = (int)std::floor(sigma * ratio);  // feature.hpp:1223 (no lower bound of 1 here)
  VH_REQUIRE(hwv >= 0 && hwv <= MAX_HALFWIDTH, "filter halfwidth must be in [0, 64]");
  float* S = nullptr;
  VH_TRY(ws(ctx, WS_D, (size_t)(nx * ny * nz), &S));
  const float sg[3] = {sigma, sigma, sigma};
  const int hw[3] = {hwv, hwv, hwv};
  const SlabInfo whole = {0, nz};
  VH_TRY(gauss_dev(ctx, src, S, mask, nx, ny, nz, sg, hw, true, whole, nullptr));
  return dev_hessian(ctx, S, mask, nx, ny, nz, sigma, grad, hess);
}

}  // namespace
}  // namespace vh

using namespace vh;

extern "C" {

// ---- options: name -> field; VISFD_HIP_<NAME> in the environment gives the value a new context starts with ----------
namespace {
struct OptionDesc { const char* name; int visfd_hip_options::*i; int64_t visfd_hip_options::*l; };
const OptionDesc kOptions[] = {
    {"gauss_3pass", &visfd_hip_options::gauss_3pass, nullptr},   {"gauss_cfg", &visfd_hip_options::gauss_cfg, nullptr},
    {"gauss_wg_per_cu", &visfd_hip_options::gauss_wg_per_cu, nullptr}, {"tv_dense", &visfd_hip_options::tv_dense, nullptr},
    {"tv_zrun", &visfd_hip_options::tv_zrun, nullptr}, {"tv_fma", &visfd_hip_options::tv_fma, nullptr},
    {"gauss_fma", &visfd_hip_options::gauss_fma, nullptr}, {"eig_f32", &visfd_hip_options::eig_f32, nullptr},
    {"tv_no_replay", &visfd_hip_options::tv_no_replay, nullptr}, {"tv_max_wg", &visfd_hip_options::tv_max_wg, nullptr},
    {"tv_poison", &visfd_hip_options::tv_poison, nullptr}, {"tv_no_fold", &visfd_hip_options::tv_no_fold, nullptr}, {"tv_exact_tiled", &visfd_hip_options::tv_exact_tiled, nullptr}, {"tv_reserve_wg", &visfd_hip_options::tv_reserve_wg, nullptr},
    {"blob_test_cap", nullptr, &visfd_hip_options::blob_test_cap}, {"debug", &visfd_hip_options::debug, nullptr},
};
bool set_option(visfd_hip_options* o, const char* name, int64_t value) {
  for (const OptionDesc& d : kOptions) {
    if (std::strcmp(d.name, name) != 0) continue;
    if (d.i) o->*(d.i) = (int)value; else o->*(d.l) = value;
    return true;
  }
  return false;
}
bool get_option(const visfd_hip_options* o, const char* name, int64_t* value) {
  for (const OptionDesc& d : kOptions) {
    if (std::strcmp(d.name, name) != 0) continue;
    *value = d.i ? (int64_t)(o->*(d.i)) : o->*(d.l);
    return true;
  }
  return false;
}
void options_from_environment(visfd_hip_options* o) {
  for (const OptionDesc& d : kOptions) {
    std::string env = "VISFD_HIP_";
    for (const char* c = d.name; *c; c++) env += (char)std::toupper((unsigned char)*c);
    if (const char* e = std::getenv(env.c_str())) set_option(o, d.name, (int64_t)std::atoll(e));
  }
}
}  // namespace

int visfd_hip_abi_version(void) { return 9; }   // 9: + visfd_hip_blob_dog_begin_dev / _end / _abort (BlobDog in two halves); 8: + the peak-height factor (`-membrane-background`): visfd_hip_peak_background_dev, _ridge_scores_bg_dev, _tensor_saliency_bg_dev, _membrane_detect_bg[_dev], _membrane_detect_slab_bg[_dev]; slab Gaussian / blob entry points of the program; 7: + visfd_hip_membrane_detect_slab (host-memory face of the slab stage); 6: + visfd_hip_get_option, tolerance modes (tv_fma, gauss_fma), slab entry points; 5: + visfd_hip_set_option, CompactMultiChannelImage3D/TVDenseStick normalisation in the shim; 2: + blob post-processing, binning, LabelConnected and its host helpers; 3: + host DiagonalizeFlatSym3 / ConvertFlatSym2Evects3; 4: + LocalFluctuations, two-step ridge (scores / directions)
const char* visfd_hip_last_error(void) { return g_last_error.c_str(); }

int visfd_hip_create(int device, void* stream, visfd_hip_ctx** out) {
  VH_REQUIRE(out, "null output pointer");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(VISFD_HIP_EDEVICE, "no HIP device available (libvisfd_hip has no CPU fallback)");
  VH_REQUIRE(device >= 0 && device < count, "bad device ordinal");
  VH_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  VH_HIP(hipGetDeviceProperties(&prop, device));
  visfd_hip_ctx* ctx = new visfd_hip_ctx();
  ctx->device = device;
  ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  options_from_environment(&ctx->opt);
  if (stream) {
    ctx->stream = (hipStream_t)stream;
    ctx->own_stream = false;
  } else {
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete ctx;
      return fail(VISFD_HIP_EDEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    ctx->own_stream = true;
  }
  *out = ctx;
  return VISFD_HIP_OK;
}

int visfd_hip_set_option(visfd_hip_ctx* ctx, const char* name, int64_t value) {
  VH_REQUIRE(ctx && name, "null argument");
  if (!set_option(&ctx->opt, name, value)) return fail(VISFD_HIP_EINVAL, std::string("unknown option: ") + name);
  return VISFD_HIP_OK;
}

int visfd_hip_get_option(visfd_hip_ctx* ctx, const char* name, int64_t* value) {
  VH_REQUIRE(ctx && name && value, "null argument");
  if (!get_option(&ctx->opt, name, value)) return fail(VISFD_HIP_EINVAL, std::string("unknown option: ") + name);
  return VISFD_HIP_OK;
}

int visfd_hip_trim(visfd_hip_ctx* ctx) {
  VH_REQUIRE(ctx, "null context");
  VH_HIP(hipSetDevice(ctx->device));
  VH_HIP(hipStreamSynchronize(ctx->stream));
  ctx->tv_table_dev = nullptr;   // lives in a workspace slot
  ctx->tv_table_h = -1;
  for (int s = 0; s < WS_NSLOTS; s++) {
    if (ctx->slot_ptr[s]) VH_HIP(hipFree(ctx->slot_ptr[s]));
    ctx->slot_ptr[s] = nullptr;
    ctx->slot_bytes[s] = 0;
  }
  return VISFD_HIP_OK;
}

int visfd_hip_destroy(visfd_hip_ctx* ctx) {
  if (!ctx) return VISFD_HIP_OK;
  int rc = visfd_hip_trim(ctx);
  if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return rc;
}

void* visfd_hip_get_stream(visfd_hip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int visfd_hip_synchronize(visfd_hip_ctx* ctx) {
  VH_REQUIRE(ctx, "null context");
  VH_HIP(hipStreamSynchronize(ctx->stream));
  return VISFD_HIP_OK;
}

int64_t visfd_hip_workspace_bytes(visfd_hip_ctx* ctx) {
  if (!ctx) return 0;
  int64_t t = 0;
  for (int s = 0; s < WS_NSLOTS; s++) t += (int64_t)ctx->slot_bytes[s];
  return t;
}

// ---- a1 ------------------------------------------------------------------------------------
int visfd_hip_gauss_taps(float sigma, int halfwidth, float* taps_out) {
  VH_REQUIRE(taps_out && halfwidth >= 0 && sigma >= 0.0f, "bad tap request");
  host_gauss_taps(sigma, halfwidth, taps_out);
  return VISFD_HIP_OK;
}
float visfd_hip_ratio_from_threshold(float thr) { return std::sqrt(-2 * std::log(thr)); }
int visfd_hip_gauss_halfwidths(const float sigma[3], float ratio, int hw[3]) {
  VH_REQUIRE(sigma && hw, "null argument");
  return halfwidths_from_ratio(sigma, ratio, hw);
}

// ---- a4 ------------------------------------------------------------------------------------
int visfd_hip_separable3d_dev(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask,
                              int64_t nx, int64_t ny, int64_t nz, const float* tx, int hx,
                              const float* ty, int hy, const float* tz, int hz, int normalize,
                              float* A_out) {
  VH_REQUIRE(ctx && src && dst && tx && ty && tz, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  const SlabInfo whole = {0, nz};
  return dev_separable3d(ctx, src, dst, mask, nx, ny, nz, tx, hx, ty, hy, tz, hz, normalize != 0, whole,
                         A_out, nullptr, 1.0f, nullptr, ctx->opt.gauss_fma != 0);
}

int visfd_hip_separable3d(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask,
                          int64_t nx, int64_t ny, int64_t nz, const float* tx, int hx,
                          const float* ty, int hy, const float* tz, int hz, int normalize,
                          float* A_out) {
  VH_REQUIRE(ctx && src && dst, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const size_t n = (size_t)(nx * ny * nz);
  float *ds, *dm, *dd;
  VH_TRY(upload(ctx, WS_H2D_0, src, n, &ds));
  VH_TRY(upload(ctx, WS_H2D_1, mask, n, &dm));
  VH_TRY(ws(ctx, WS_H2D_2, n, &dd));
  VH_TRY(visfd_hip_separable3d_dev(ctx, ds, dd, dm, nx, ny, nz, tx, hx, ty, hy, tz, hz, normalize, A_out));
  return download(ctx, dst, dd, n);
}

// ---- a5 ------------------------------------------------------------------------------------
int visfd_hip_apply_gauss_dev(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask,
                              int64_t nx, int64_t ny, int64_t nz, const float sigma[3],
                              const int hw[3], int normalize, float* A_out) {
  VH_REQUIRE(ctx, "null context");
  VH_HIP(hipSetDevice(ctx->device));
  const SlabInfo whole = {0, nz};
  return gauss_dev(ctx, src, dst, mask, nx, ny, nz, sigma, hw, normalize != 0, whole, A_out, nullptr, 1.0f, nullptr,
                   ctx->opt.gauss_fma != 0);
}

int visfd_hip_apply_gauss_slab_dev(visfd_hip_ctx* ctx, const float* src, float* dst, int64_t nx,
                                   int64_t ny, int64_t nz_local, int64_t z_lo, int64_t nz_global,
                                   const float sigma[3], const int hw[3], int normalize, float* A_out) {
  VH_REQUIRE(ctx, "null context");
  VH_REQUIRE(z_lo >= 0 && z_lo + nz_local <= nz_global, "slab outside the volume");
  VH_HIP(hipSetDevice(ctx->device));
  const SlabInfo slab = {z_lo, nz_global};
  return gauss_dev(ctx, src, dst, nullptr, nx, ny, nz_local, sigma, hw, normalize != 0, slab, A_out, nullptr, 1.0f, nullptr,
                   ctx->opt.gauss_fma != 0);
}

int visfd_hip_apply_gauss(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask,
                          int64_t nx, int64_t ny, int64_t nz, const float sigma[3], const int hw[3],
                          int normalize, float* A_out) {
  VH_REQUIRE(ctx && src && dst, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const size_t n = (size_t)(nx * ny * nz);
  float *ds, *dm, *dd;
  VH_TRY(upload(ctx, WS_H2D_0, src, n, &ds));
  VH_TRY(upload(ctx, WS_H2D_1, mask, n, &dm));
  VH_TRY(ws(ctx, WS_H2D_2, n, &dd));
  VH_TRY(visfd_hip_apply_gauss_dev(ctx, ds, dd, dm, nx, ny, nz, sigma, hw, normalize, A_out));
  return download(ctx, dst, dd, n);
}

// ---- f4: LocalFluctuations (lib/visfd/filter3d.hpp:1698-1853), Gaussian weights only -------
int visfd_hip_local_fluctuations_dev(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask,
                                     int64_t nx, int64_t ny, int64_t nz, const float sigma[3], float exponent,
                                     float truncate_ratio, int normalize) {
  VH_REQUIRE(ctx && src && dst && sigma, "null argument");
  VH_REQUIRE(src != dst, "LocalFluctuations cannot run in place");
  VH_REQUIRE(exponent == 2.0f, "LocalFluctuations: only the Gaussian case (exponent 2) is provided");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const i64 n = nx * ny * nz;
  int hw[3];
  VH_TRY(halfwidths_from_ratio(sigma, truncate_ratio, hw));    // ApplyGauss(sigma[3], ratio), filter3d.hpp:1240-1247
  const float wpeak = host_gengauss3d_peak(sigma, exponent, truncate_ratio);
  const SlabInfo whole = {0, nz};
  float* p2 = nullptr;
  VH_TRY(ws(ctx, WS_C, (size_t)n, &p2));
  VH_TRY(gauss_dev(ctx, src, dst, mask, nx, ny, nz, sigma, hw, normalize != 0, whole, nullptr));   // local average
  VH_TRY(dev_sub_square(ctx, src, dst, p2, n));                                                      // (src - avg)^2
  VH_TRY(gauss_dev(ctx, p2, dst, mask, nx, ny, nz, sigma, hw, normalize != 0, whole, nullptr));    // its local average
  return dev_scale_clamp_sqrt(ctx, dst, n, wpeak);
}

int visfd_hip_local_fluctuations(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask,
                                 int64_t nx, int64_t ny, int64_t nz, const float sigma[3], float exponent,
                                 float truncate_ratio, int normalize) {
  VH_REQUIRE(ctx && src && dst, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const size_t n = (size_t)(nx * ny * nz);
  float *ds, *dm, *dd;
  VH_TRY(upload(ctx, WS_H2D_0, src, n, &ds));
  VH_TRY(upload(ctx, WS_H2D_1, mask, n, &dm));
  VH_TRY(ws(ctx, WS_H2D_2, n, &dd));
  VH_TRY(visfd_hip_local_fluctuations_dev(ctx, ds, dd, dm, nx, ny, nz, sigma, exponent, truncate_ratio, normalize));
  return download(ctx, dst, dd, n);
}

// sigma = radius / (9 pi / 2)^(1/6) (filter3d.hpp:1908-1914) and, for a negative ratio, the window from the decay
// threshold: ratio = (-log thr)^(1/exponent) (bin/filter_mrc/filter3d_variants.hpp:663-669); host arithmetic
int visfd_hip_fluctuation_sigmas(const float radius[3], float exponent, float truncate_ratio, float truncate_threshold,
                                 float sigma_out[3], float* ratio_out) {
  VH_REQUIRE(radius && sigma_out && ratio_out, "null argument");
  const float r_over_sigma = (float)std::pow((9.0 / 2) * M_PI, 1.0 / 6);
  for (int d = 0; d < 3; d++) sigma_out[d] = radius[d] / r_over_sigma;
  float ratio = truncate_ratio;
  if (ratio < 0.0f) ratio = (float)std::pow((double)(-std::log(truncate_threshold)), 1.0 / (double)exponent);
  *ratio_out = ratio;
  return VISFD_HIP_OK;
}

// ---- a6 ------------------------------------------------------------------------------------
int visfd_hip_apply_dog_dev(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask,
                            int64_t nx, int64_t ny, int64_t nz, const float sa[3], const float sb[3],
                            const int hw[3], float* A, float* B) {
  VH_REQUIRE(ctx && src && dst && sa && sb && hw, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  float* tmp = nullptr;
  VH_TRY(ws(ctx, WS_C, (size_t)(nx * ny * nz), &tmp));
  return dog_dev(ctx, src, dst, tmp, mask, nx, ny, nz, sa, sb, hw, 1.0f, false, A, B);
}

int visfd_hip_apply_dog(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask,
                        int64_t nx, int64_t ny, int64_t nz, const float sa[3], const float sb[3],
                        const int hw[3], float* A, float* B) {
  VH_REQUIRE(ctx && src && dst, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const size_t n = (size_t)(nx * ny * nz);
  float *ds, *dm, *dd;
  VH_TRY(upload(ctx, WS_H2D_0, src, n, &ds));
  VH_TRY(upload(ctx, WS_H2D_1, mask, n, &dm));
  VH_TRY(ws(ctx, WS_H2D_2, n, &dd));
  VH_TRY(visfd_hip_apply_dog_dev(ctx, ds, dd, dm, nx, ny, nz, sa, sb, hw, A, B));
  return download(ctx, dst, dd, n);
}

// ---- a7 ------------------------------------------------------------------------------------
int visfd_hip_apply_log_dev(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask,
                            int64_t nx, int64_t ny, int64_t nz, const float sigma[3], float delta,
                            float ratio, float* A, float* B) {
  VH_REQUIRE(ctx && src && dst && sigma, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  float* tmp = nullptr;
  VH_TRY(ws(ctx, WS_C, (size_t)(nx * ny * nz), &tmp));
  return log_dev(ctx, src, dst, tmp, mask, nx, ny, nz, sigma, delta, ratio, A, B);
}

int visfd_hip_apply_log(visfd_hip_ctx* ctx, const float* src, float* dst, const float* mask,
                        int64_t nx, int64_t ny, int64_t nz, const float sigma[3], float delta,
                        float ratio, float* A, float* B) {
  VH_REQUIRE(ctx && src && dst, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const size_t n = (size_t)(nx * ny * nz);
  float *ds, *dm, *dd;
  VH_TRY(upload(ctx, WS_H2D_0, src, n, &ds));
  VH_TRY(upload(ctx, WS_H2D_1, mask, n, &dm));
  VH_TRY(ws(ctx, WS_H2D_2, n, &dd));
  VH_TRY(visfd_hip_apply_log_dev(ctx, ds, dd, dm, nx, ny, nz, sigma, delta, ratio, A, B));
  return download(ctx, dst, dd, n);
}

// ---- a8 ------------------------------------------------------------------------------------
int visfd_hip_blob_dog_dev(visfd_hip_ctx* ctx, const float* src, const float* mask, int64_t nx,
                           int64_t ny, int64_t nz, const float* blob_sigma, int n_sigma,
                           const float* aspect, float delta, float ratio, float min_thr, float max_thr,
                           int use_ratios, visfd_hip_blob* minima, int64_t min_cap, int64_t* n_min,
                           visfd_hip_blob* maxima, int64_t max_cap, int64_t* n_max) {
  VH_REQUIRE(ctx, "null context");
  VH_HIP(hipSetDevice(ctx->device));
  return blob_dog_dev(ctx, src, mask, nx, ny, nz, blob_sigma, n_sigma, aspect, delta, ratio, min_thr,
                      max_thr, use_ratios != 0, minima, min_cap, n_min, maxima, max_cap, n_max);
}

int visfd_hip_blob_dog_begin_dev(visfd_hip_ctx* ctx, const float* src, const float* mask, int64_t nx, int64_t ny, int64_t nz,
                                 const float* blob_sigma, int n_sigma, const float* aspect, float delta, float ratio,
                                 float min_thr, float max_thr, int use_ratios, visfd_hip_blob_job** job) {
  VH_REQUIRE(ctx && job, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  BlobJob* j = nullptr;
  VH_TRY(blob_dog_begin(ctx, src, mask, nx, ny, nz, blob_sigma, n_sigma, aspect, delta, ratio, min_thr, max_thr, use_ratios != 0, &j));
  *job = reinterpret_cast<visfd_hip_blob_job*>(j);
  return VISFD_HIP_OK;
}
int visfd_hip_blob_dog_end(visfd_hip_blob_job* job, visfd_hip_blob* minima, int64_t min_cap, int64_t* n_min,
                           visfd_hip_blob* maxima, int64_t max_cap, int64_t* n_max) {
  VH_REQUIRE(job, "null job");
  BlobJob* j = reinterpret_cast<BlobJob*>(job);
  VH_HIP(hipSetDevice(j->ctx->device));
  return blob_dog_end(j, minima, min_cap, n_min, maxima, max_cap, n_max);
}
void visfd_hip_blob_dog_abort(visfd_hip_blob_job* job) {
  BlobJob* j = reinterpret_cast<BlobJob*>(job);
  if (!j) return;
  (void)hipSetDevice(j->ctx->device);
  if (j->ctx->aux_stream) (void)hipStreamSynchronize(j->ctx->aux_stream);
  (void)hipStreamSynchronize(j->ctx->stream);   // nothing of the job is in flight when its events go
  delete j;
}

int visfd_hip_blob_dog(visfd_hip_ctx* ctx, const float* src, const float* mask, int64_t nx, int64_t ny,
                       int64_t nz, const float* blob_sigma, int n_sigma, const float* aspect,
                       float delta, float ratio, float min_thr, float max_thr, int use_ratios,
                       visfd_hip_blob* minima, int64_t min_cap, int64_t* n_min, visfd_hip_blob* maxima,
                       int64_t max_cap, int64_t* n_max) {
  VH_REQUIRE(ctx && src, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const size_t n = (size_t)(nx * ny * nz);
  float *ds, *dm;
  VH_TRY(upload(ctx, WS_H2D_0, src, n, &ds));
  VH_TRY(upload(ctx, WS_H2D_1, mask, n, &dm));
  return blob_dog_dev(ctx, ds, dm, nx, ny, nz, blob_sigma, n_sigma, aspect, delta, ratio, min_thr, max_thr,
                      use_ratios != 0, minima, min_cap, n_min, maxima, max_cap, n_max);
}

int visfd_hip_blob_diameters_to_sigmas(const float* d, int n, float* s) {
  VH_REQUIRE(n >= 0 && (n == 0 || (d && s)), "bad argument");   // empty lists have no storage
  for (int i = 0; i < n; i++) s[i] = (float)(d[i] / (2.0 * std::sqrt(3.0)));  // feature.hpp:475
  return VISFD_HIP_OK;
}
int visfd_hip_blob_sigmas_to_diameters(const float* s, int n, float* d) {
  VH_REQUIRE(n >= 0 && (n == 0 || (d && s)), "bad argument");
  for (int i = 0; i < n; i++) d[i] = (float)(s[i] * 2.0 * std::sqrt(3.0));  // feature.hpp:504
  return VISFD_HIP_OK;
}

// ---- a9 ------------------------------------------------------------------------------------
int visfd_hip_calc_hessian_dev(visfd_hip_ctx* ctx, const float* src, float* grad, float* hess,
                               const float* mask, int64_t nx, int64_t ny, int64_t nz, float sigma,
                               float ratio) {
  VH_REQUIRE(ctx, "null context");
  VH_HIP(hipSetDevice(ctx->device));
  return calc_hessian_dev(ctx, src, grad, hess, mask, nx, ny, nz, sigma, ratio);
}

int visfd_hip_calc_hessian(visfd_hip_ctx* ctx, const float* src, float* grad, float* hess,
                           const float* mask, int64_t nx, int64_t ny, int64_t nz, float sigma,
                           float ratio) {
  VH_REQUIRE(ctx && src, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const size_t n = (size_t)(nx * ny * nz);
  float *ds, *dm, *dg = nullptr, *dh = nullptr, *aos = nullptr;
  VH_TRY(upload(ctx, WS_H2D_0, src, n, &ds));
  VH_TRY(upload(ctx, WS_H2D_1, mask, n, &dm));
  if (grad) VH_TRY(ws(ctx, WS_H2D_2, 3 * n, &dg));
  if (hess) VH_TRY(ws(ctx, WS_H2D_3, 6 * n, &dh));
  VH_TRY(ws(ctx, WS_H2D_4, 6 * n, &aos));
  VH_TRY(calc_hessian_dev(ctx, ds, dg, dh, dm, nx, ny, nz, sigma, ratio));
  // interleave on the device; voxels with mask==0 keep the caller's values
  if (grad) {
    VH_HIP(hipMemcpyAsync(aos, grad, sizeof(float) * 3 * n, hipMemcpyHostToDevice, ctx->stream));
    VH_TRY(dev_planar_to_interleaved(ctx, dg, aos, (i64)n, 3, dm));
    VH_TRY(download(ctx, grad, aos, 3 * n));
  }
  if (hess) {
    VH_HIP(hipMemcpyAsync(aos, hess, sizeof(float) * 6 * n, hipMemcpyHostToDevice, ctx->stream));
    VH_TRY(dev_planar_to_interleaved(ctx, dh, aos, (i64)n, 6, dm));
    VH_TRY(download(ctx, hess, aos, 6 * n));
  }
  return VISFD_HIP_OK;
}

// ---- a10 -----------------------------------------------------------------------------------
int visfd_hip_diagonalize_flat_sym3_dev(visfd_hip_ctx* ctx, const float* m6, float* out6, int64_t n,
                                        int order) {
  VH_REQUIRE(ctx && m6 && out6 && n >= 0, "bad argument");
  VH_REQUIRE(order == 0 || order == 1, "unsupported eigenvalue order");
  VH_HIP(hipSetDevice(ctx->device));
  if (n == 0) return VISFD_HIP_OK;
  return dev_diagonalize(ctx, m6, out6, n, order);
}

int visfd_hip_diagonalize_flat_sym3(visfd_hip_ctx* ctx, const float* m6, float* out6, int64_t n,
                                    int order) {
  VH_REQUIRE(ctx && m6 && out6 && n >= 0, "bad argument");
  VH_REQUIRE(order == 0 || order == 1, "unsupported eigenvalue order");
  VH_HIP(hipSetDevice(ctx->device));
  if (n == 0) return VISFD_HIP_OK;
  float *aos, *pin, *pout;
  VH_TRY(upload(ctx, WS_H2D_0, m6, 6 * (size_t)n, &aos));
  VH_TRY(ws(ctx, WS_H2D_1, 6 * (size_t)n, &pin));
  VH_TRY(ws(ctx, WS_H2D_2, 6 * (size_t)n, &pout));
  VH_TRY(dev_interleaved_to_planar(ctx, aos, pin, n, 6));
  VH_TRY(dev_diagonalize(ctx, pin, pout, n, order));
  VH_TRY(dev_planar_to_interleaved(ctx, pout, aos, n, 6, nullptr));
  return download(ctx, out6, aos, 6 * (size_t)n);
}

// ---- a12 (saliency) --------------------------------------------------------------------------
int visfd_hip_hessian_saliency_dev(visfd_hip_ctx* ctx, const float* hess, const float* mask, int64_t nvox,
                                   int order, float* sal, float* dir) {
  VH_REQUIRE(ctx && hess && sal && dir && nvox > 0, "bad argument");
  VH_REQUIRE(order == 0 || order == 1, "unsupported eigenvalue order");
  VH_HIP(hipSetDevice(ctx->device));
  return dev_hessian_saliency(ctx, hess, mask, nvox, order, sal, dir);
}

int visfd_hip_hessian_saliency(visfd_hip_ctx* ctx, const float* hess, const float* mask, int64_t nvox,
                               int order, float* sal, float* dir) {
  VH_REQUIRE(ctx && hess && sal && dir && nvox > 0, "bad argument");
  VH_REQUIRE(order == 0 || order == 1, "unsupported eigenvalue order");
  VH_HIP(hipSetDevice(ctx->device));
  const size_t n = (size_t)nvox;
  float *aos, *ph, *dm, *dsal, *pdir;
  VH_TRY(upload(ctx, WS_H2D_0, hess, 6 * n, &aos));
  VH_TRY(upload(ctx, WS_H2D_1, mask, n, &dm));
  VH_TRY(ws(ctx, WS_H2D_2, 6 * n, &ph));
  VH_TRY(ws(ctx, WS_H2D_3, n, &dsal));
  VH_TRY(ws(ctx, WS_H2D_4, 3 * n, &pdir));
  VH_TRY(dev_interleaved_to_planar(ctx, aos, ph, nvox, 6));
  VH_TRY(dev_hessian_saliency(ctx, ph, dm, nvox, order, dsal, pdir));
  VH_TRY(download(ctx, sal, dsal, n));
  // direction: only voxels with mask != 0 are written (handlers.cpp:1650-1651,1738-1740)
  VH_HIP(hipMemcpyAsync(aos, dir, sizeof(float) * 3 * n, hipMemcpyHostToDevice, ctx->stream));
  VH_TRY(dev_planar_to_interleaved(ctx, pdir, aos, nvox, 3, dm));
  return download(ctx, dir, aos, 3 * n);
}

int visfd_hip_ridge_saliency_dev(visfd_hip_ctx* ctx, const float* src, const float* mask, int64_t nx,
                                 int64_t ny, int64_t nz, float sigma, float ratio, int order, float* sal,
                                 float* dir) {
  VH_REQUIRE(ctx && src && sal && dir, "null argument");
  VH_REQUIRE(order == 0 || order == 1, "unsupported eigenvalue order");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const int hwv = (int)std::floor(sigma * ratio);
  VH_REQUIRE(hwv >= 0 && hwv <= MAX_HALFWIDTH, "filter halfwidth must be in [0, 64]");
  float* S = nullptr;
  VH_TRY(ws(ctx, WS_D, (size_t)(nx * ny * nz), &S));
  const float sg[3] = {sigma, sigma, sigma};
  const int hw[3] = {hwv, hwv, hwv};
  const SlabInfo whole = {0, nz};
  VH_TRY(gauss_dev(ctx, src, S, mask, nx, ny, nz, sg, hw, true, whole, nullptr));
  return dev_ridge_saliency_fused(ctx, S, mask, nx, ny, nz, sigma, order, sal, dir);
}

// The same in two steps for callers that threshold in between (HandleTV does: handlers.cpp:1751-1797): scores for
// every voxel (the smoothed volume is handed back), then directions of the voxels whose score survived.
int visfd_hip_ridge_scores_bg_dev(visfd_hip_ctx* ctx, const float* src, const float* mask, int64_t nx, int64_t ny,
                                  int64_t nz, float sigma, float ratio, int order, const float* background, float* sal,
                                  float* smoothed) {
  VH_REQUIRE(ctx && src && sal && smoothed, "null argument");
  VH_REQUIRE(smoothed != src && smoothed != sal, "the smoothed volume needs its own buffer");
  VH_REQUIRE(order == 0 || order == 1, "unsupported eigenvalue order");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const int hwv = (int)std::floor(sigma * ratio);
  VH_REQUIRE(hwv >= 0 && hwv <= MAX_HALFWIDTH, "filter halfwidth must be in [0, 64]");
  const float sg[3] = {sigma, sigma, sigma};
  const int hw[3] = {hwv, hwv, hwv};
  const SlabInfo whole = {0, nz};
  VH_TRY(gauss_dev(ctx, src, smoothed, mask, nx, ny, nz, sg, hw, true, whole, nullptr));
  return dev_ridge_score(ctx, smoothed, mask, nx, ny, nz, sigma, order, sal, background ? src : nullptr, background);
}
int visfd_hip_ridge_scores_dev(visfd_hip_ctx* ctx, const float* src, const float* mask, int64_t nx, int64_t ny,
                               int64_t nz, float sigma, float ratio, int order, float* sal, float* smoothed) {
  return visfd_hip_ridge_scores_bg_dev(ctx, src, mask, nx, ny, nz, sigma, ratio, order, nullptr, sal, smoothed);
}
// the background of the peak-height factor: ApplyGauss(image, sigma_background, floor(sigma_background * ratio), mask,
// normalize) -- bin/filter_mrc/handlers.cpp:1577-1592
int visfd_hip_peak_background_dev(visfd_hip_ctx* ctx, const float* src, const float* mask, int64_t nx, int64_t ny, int64_t nz,
                                  float sigma_background, float ratio, int normalize, float* background) {
  VH_REQUIRE(ctx && src && background && background != src, "bad argument");
  VH_REQUIRE(sigma_background > 0.0f, "the background width must be positive");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const int hwv = (int)std::floor(sigma_background * ratio);
  VH_REQUIRE(hwv >= 0 && hwv <= MAX_HALFWIDTH, "background filter halfwidth must be in [0, 64]");
  const float sg[3] = {sigma_background, sigma_background, sigma_background};
  const int hw[3] = {hwv, hwv, hwv};
  const SlabInfo whole = {0, nz};
  return gauss_dev(ctx, src, background, mask, nx, ny, nz, sg, hw, normalize != 0, whole, nullptr);
}

int visfd_hip_ridge_directions_dev(visfd_hip_ctx* ctx, const float* smoothed, int64_t nx, int64_t ny, int64_t nz,
                                   float sigma, int order, const float* sal, float* dir) {
  VH_REQUIRE(ctx && smoothed && sal && dir, "null argument");
  VH_REQUIRE(order == 0 || order == 1, "unsupported eigenvalue order");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  return dev_ridge_directions(ctx, smoothed, sal, nx, ny, nz, sigma, order, dir);
}

// ---- a12 (threshold) -------------------------------------------------------------------------
int visfd_hip_threshold_fraction_dev(visfd_hip_ctx* ctx, float* sal, const float* mask, int64_t nvox,
                                     float fraction, float* thr_out) {
  VH_REQUIRE(ctx && sal && nvox > 0, "bad argument");
  VH_REQUIRE(fraction >= 0.0f && fraction <= 1.0f, "fraction must be in [0,1]");
  VH_HIP(hipSetDevice(ctx->device));
  return dev_threshold_fraction(ctx, sal, mask, nvox, fraction, thr_out);
}

int visfd_hip_threshold_fraction(visfd_hip_ctx* ctx, float* sal, const float* mask, int64_t nvox,
                                 float fraction, float* thr_out) {
  VH_REQUIRE(ctx && sal && nvox > 0, "bad argument");
  VH_REQUIRE(fraction >= 0.0f && fraction <= 1.0f, "fraction must be in [0,1]");
  VH_HIP(hipSetDevice(ctx->device));
  float *ds, *dm;
  VH_TRY(upload(ctx, WS_H2D_0, sal, (size_t)nvox, &ds));
  VH_TRY(upload(ctx, WS_H2D_1, mask, (size_t)nvox, &dm));
  VH_TRY(dev_threshold_fraction(ctx, ds, dm, nvox, fraction, thr_out));
  return download(ctx, sal, ds, (size_t)nvox);
}

int visfd_hip_select_histogram_dev(visfd_hip_ctx* ctx, const float* sal, const float* mask, int64_t nvox,
                                   int pass, uint32_t prefix, uint64_t* hist_host, uint64_t* n_unmasked) {
  VH_REQUIRE(ctx && sal && hist_host && nvox > 0, "bad argument");
  VH_REQUIRE(pass >= 0 && pass <= 2, "round must be 0, 1 or 2");
  VH_HIP(hipSetDevice(ctx->device));
  return dev_select_histogram(ctx, sal, mask, nvox, pass, prefix, hist_host, n_unmasked);
}

int visfd_hip_select_histogram_todev(visfd_hip_ctx* ctx, const float* sal, const float* mask, int64_t nvox, int pass,
                                     uint32_t prefix, uint64_t* hist_dev) {
  VH_REQUIRE(ctx && sal && hist_dev && nvox > 0, "bad argument");
  VH_REQUIRE(pass >= 0 && pass <= 2, "round must be 0, 1 or 2");
  VH_HIP(hipSetDevice(ctx->device));
  return dev_select_histogram_todev(ctx, sal, mask, nvox, pass, prefix, hist_dev);
}

int visfd_hip_apply_threshold_dev(visfd_hip_ctx* ctx, float* sal, int64_t nvox, float thr) {
  VH_REQUIRE(ctx && sal && nvox > 0, "bad argument");
  VH_HIP(hipSetDevice(ctx->device));
  return dev_apply_threshold(ctx, sal, nvox, thr);
}

// ---- f4: binning ------------------------------------------------------------------------------
int visfd_hip_bin_array3d_dev(visfd_hip_ctx* ctx, const float* src, const int64_t size_src[3], float* dst,
                              const int64_t size_dst[3], const int* offset) {
  VH_REQUIRE(ctx && src && dst && size_src && size_dst, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  return dev_bin_array3d(ctx, src, size_src, dst, size_dst, offset);
}

int visfd_hip_unbin_array3d_dev(visfd_hip_ctx* ctx, const float* src, const int64_t size_src[3], float* dst,
                                const int64_t size_dst[3], const int* offset) {
  VH_REQUIRE(ctx && src && dst && size_src && size_dst, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  return dev_unbin_array3d(ctx, src, size_src, dst, size_dst, offset);
}

int visfd_hip_bin_array3d(visfd_hip_ctx* ctx, const float* src, const int64_t size_src[3], float* dst,
                          const int64_t size_dst[3], const int* offset) {
  VH_REQUIRE(ctx && src && dst && size_src && size_dst, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(size_src[0], size_src[1], size_src[2]));
  VH_TRY(check_dims(size_dst[0], size_dst[1], size_dst[2]));
  const size_t ns = (size_t)(size_src[0] * size_src[1] * size_src[2]), nd = (size_t)(size_dst[0] * size_dst[1] * size_dst[2]);
  float *ds, *dd;
  VH_TRY(upload(ctx, WS_H2D_0, src, ns, &ds));
  VH_TRY(ws(ctx, WS_H2D_2, nd, &dd));
  VH_TRY(dev_bin_array3d(ctx, ds, size_src, dd, size_dst, offset));
  return download(ctx, dst, dd, nd);
}

int visfd_hip_unbin_array3d(visfd_hip_ctx* ctx, const float* src, const int64_t size_src[3], float* dst,
                            const int64_t size_dst[3], const int* offset) {
  VH_REQUIRE(ctx && src && dst && size_src && size_dst, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(size_src[0], size_src[1], size_src[2]));
  VH_TRY(check_dims(size_dst[0], size_dst[1], size_dst[2]));
  const size_t ns = (size_t)(size_src[0] * size_src[1] * size_src[2]), nd = (size_t)(size_dst[0] * size_dst[1] * size_dst[2]);
  float *ds, *dd;
  VH_TRY(upload(ctx, WS_H2D_0, src, ns, &ds));
  VH_TRY(ws(ctx, WS_H2D_2, nd, &dd));
  VH_TRY(dev_unbin_array3d(ctx, ds, size_src, dd, size_dst, offset));
  return download(ctx, dst, dd, nd);
}

// ---- a13 + a14 -------------------------------------------------------------------------------
int visfd_hip_tv_tables(float sigma_tv, float cutoff, int* h_out, float* w, float* rhat) {
  const int h = host_tv_halfwidth(sigma_tv, cutoff);
  VH_REQUIRE(h >= 0, "negative window");
  if (h_out) *h_out = h;
  if (w) host_tv_tables(sigma_tv, h, w, rhat);
  return VISFD_HIP_OK;
}

int visfd_hip_tv_dense_stick_slab_dev(visfd_hip_ctx* ctx, const float* sal, const float* dir, float* ten,
                                      const float* mask_src, const float* mask_dst, int64_t nx,
                                      int64_t ny, int64_t nz_local, int64_t z_out0, int64_t z_out1,
                                      float sigma_tv, int exponent, float cutoff, int curves) {
  VH_REQUIRE(ctx && sal && dir && ten, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  return dev_tv_dense_stick(ctx, sal, dir, ten, mask_src, mask_dst, nx, ny, nz_local, z_out0, z_out1,
                            sigma_tv, exponent, cutoff, curves != 0);
}

int visfd_hip_tv_dense_stick_dev(visfd_hip_ctx* ctx, const float* sal, const float* dir, float* ten,
                                 const float* mask_src, const float* mask_dst, int64_t nx, int64_t ny,
                                 int64_t nz, float sigma_tv, int exponent, float cutoff, int curves) {
  return visfd_hip_tv_dense_stick_slab_dev(ctx, sal, dir, ten, mask_src, mask_dst, nx, ny, nz, 0, nz,
                                           sigma_tv, exponent, cutoff, curves);
}

int visfd_hip_tv_dense_stick(visfd_hip_ctx* ctx, const float* sal, const float* dir, float* ten,
                             const float* mask_src, const float* mask_dst, int64_t nx, int64_t ny,
                             int64_t nz, float sigma_tv, int exponent, float cutoff, int curves) {
  VH_REQUIRE(ctx && sal && dir && ten, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const size_t n = (size_t)(nx * ny * nz);
  float *dsal, *aos, *pdir, *pten, *dms = nullptr, *dmd = nullptr;
  VH_TRY(upload(ctx, WS_H2D_0, sal, n, &dsal));
  VH_TRY(upload(ctx, WS_H2D_1, dir, 3 * n, &aos));
  VH_TRY(ws(ctx, WS_H2D_2, 3 * n, &pdir));
  VH_TRY(dev_interleaved_to_planar(ctx, aos, pdir, (i64)n, 3));
  VH_TRY(ws(ctx, WS_H2D_3, 6 * n, &pten));
  VH_TRY(upload(ctx, WS_H2D_4, mask_src, n, &dms));
  if (mask_dst == mask_src) dmd = dms;
  else VH_TRY(upload(ctx, WS_A, mask_dst, n, &dmd));
  VH_TRY(dev_tv_dense_stick(ctx, dsal, pdir, pten, dms, dmd, nx, ny, nz, 0, nz, sigma_tv, exponent, cutoff,
                            curves != 0));
  // tensors of voxels with mask_dst == 0 keep the caller's values (no storage in the reference)
  float* aos6 = nullptr;
  VH_TRY(ws(ctx, WS_B, 6 * n, &aos6));
  VH_HIP(hipMemcpyAsync(aos6, ten, sizeof(float) * 6 * n, hipMemcpyHostToDevice, ctx->stream));
  VH_TRY(dev_planar_to_interleaved(ctx, pten, aos6, (i64)n, 6, dmd));
  return download(ctx, ten, aos6, 6 * n);
}

// TVDenseStick's normalisation denominators (feature.hpp:1761-1822): den[voxel] = sum of w(j) * mask_src(sender) over
// the votes the voxel receives; voxels with mask_dst == 0 keep the caller's value
int visfd_hip_tv_weight_sum(visfd_hip_ctx* ctx, const float* sal, float* den, const float* mask_src, const float* mask_dst,
                            int64_t nx, int64_t ny, int64_t nz, float sigma_tv, float cutoff) {
  VH_REQUIRE(ctx && sal && den, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const size_t n = (size_t)(nx * ny * nz);
  float *dsal, *dden, *dms = nullptr, *dmd = nullptr;
  VH_TRY(upload(ctx, WS_H2D_0, sal, n, &dsal));
  VH_TRY(upload(ctx, WS_H2D_1, den, n, &dden));
  VH_TRY(upload(ctx, WS_H2D_4, mask_src, n, &dms));
  if (mask_dst == mask_src) dmd = dms;
  else VH_TRY(upload(ctx, WS_A, mask_dst, n, &dmd));
  VH_TRY(dev_tv_weight_sum(ctx, dsal, dden, dms, dmd, nx, ny, nz, sigma_tv, cutoff));
  return download(ctx, den, dden, n);
}

// ---- HandleTV compute section ------------------------------------------------------------------
int visfd_hip_membrane_detect_bg_dev(visfd_hip_ctx* ctx, const float* src, const float* mask, int64_t nx,
                                     int64_t ny, int64_t nz, float sigma, float ratio, int order,
                                     float best_fraction, float threshold_abs, float sigma_tv, int exponent,
                                     float cutoff, float sigma_background, int normalize_background, float* sal, float* ten,
                                     float* dir, float* thr_out) {
  VH_REQUIRE(ctx && src && sal, "null argument");
  VH_REQUIRE(order == 0 || order == 1, "unsupported eigenvalue order");
  VH_REQUIRE(best_fraction <= 1.0f, "fraction must be <= 1");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const i64 n = nx * ny * nz;
  float* d = dir;
  if (!d) VH_TRY(ws(ctx, WS_TVAUX, (size_t)(3 * n), &d));
  float* smoothed = nullptr;
  VH_TRY(ws(ctx, WS_D, (size_t)n, &smoothed));
  // optional peak-height factor (`-membrane-background`): scores and post-vote scores are multiplied by image - background
  float* bg = nullptr;
  if (sigma_background > 0.0f) {
    VH_TRY(ws(ctx, WS_C, (size_t)n, &bg));
    VH_TRY(visfd_hip_peak_background_dev(ctx, src, mask, nx, ny, nz, sigma_background, ratio, normalize_background, bg));
  }
  VH_TRY(visfd_hip_ridge_scores_bg_dev(ctx, src, mask, nx, ny, nz, sigma, ratio, order, bg, sal, smoothed));
  float thr = threshold_abs;
  if (best_fraction >= 0.0f) VH_TRY(dev_threshold_fraction(ctx, sal, mask, n, best_fraction, &thr));
  else VH_TRY(dev_apply_threshold(ctx, sal, n, thr));
  if (thr_out) *thr_out = thr;
  // directions only where the score survived: nothing else is read downstream (a caller-supplied `dir`
  // receives zeros elsewhere)
  if (dir) VH_HIP(hipMemsetAsync(dir, 0, sizeof(float) * 3 * (size_t)n, ctx->stream));
  VH_TRY(dev_ridge_directions(ctx, smoothed, sal, nx, ny, nz, sigma, order, d));
  if (sigma_tv > 0.0f) {
    float* t = ten;
    if (!t) VH_TRY(ws(ctx, WS_B, (size_t)(6 * n), &t));
    VH_TRY(dev_tv_dense_stick(ctx, sal, d, t, mask, mask, nx, ny, nz, 0, nz, sigma_tv, exponent, cutoff, false));
    VH_TRY(dev_tensor_saliency(ctx, t, mask, n, order, sal, bg ? src : nullptr, bg));
  }
  return VISFD_HIP_OK;
}
int visfd_hip_membrane_detect_dev(visfd_hip_ctx* ctx, const float* src, const float* mask, int64_t nx,
                                  int64_t ny, int64_t nz, float sigma, float ratio, int order,
                                  float best_fraction, float threshold_abs, float sigma_tv, int exponent,
                                  float cutoff, float* sal, float* ten, float* dir, float* thr_out) {
  return visfd_hip_membrane_detect_bg_dev(ctx, src, mask, nx, ny, nz, sigma, ratio, order, best_fraction, threshold_abs, sigma_tv,
                                          exponent, cutoff, 0.0f, 1, sal, ten, dir, thr_out);
}

int visfd_hip_membrane_detect_bg(visfd_hip_ctx* ctx, const float* src, const float* mask, int64_t nx, int64_t ny,
                                 int64_t nz, float sigma, float ratio, int order, float best_fraction,
                                 float threshold_abs, float sigma_tv, int exponent, float cutoff, float sigma_background,
                                 int normalize_background, float* sal, float* ten, float* dir, float* thr_out) {
  VH_REQUIRE(ctx && src && sal, "null argument");
  VH_HIP(hipSetDevice(ctx->device));
  VH_TRY(check_dims(nx, ny, nz));
  const size_t n = (size_t)(nx * ny * nz);
  float *ds, *dm, *dsal, *pdir, *pten = nullptr, *aos;
  VH_TRY(upload(ctx, WS_H2D_0, src, n, &ds));
  VH_TRY(upload(ctx, WS_H2D_1, mask, n, &dm));
  VH_TRY(ws(ctx, WS_H2D_2, n, &dsal));
  VH_TRY(ws(ctx, WS_H2D_3, 3 * n, &pdir));
  if (ten && sigma_tv > 0.0f) {
    VH_TRY(ws(ctx, WS_H2D_4, 6 * n, &pten));
    VH_HIP(hipMemsetAsync(pten, 0, sizeof(float) * 6 * n, ctx->stream));
  }
  VH_TRY(visfd_hip_membrane_detect_bg_dev(ctx, ds, dm, nx, ny, nz, sigma, ratio, order, best_fraction, threshold_abs, sigma_tv,
                                          exponent, cutoff, sigma_background, normalize_background, dsal, pten, pdir, thr_out));
  VH_TRY(download(ctx, sal, dsal, n));
  if (ten && pten) {
    VH_TRY(ws(ctx, WS_A, 6 * n, &aos));
    VH_TRY(dev_planar_to_interleaved(ctx, pten, aos, (i64)n, 6, nullptr));
    VH_TRY(download(ctx, ten, aos, 6 * n));
  }
  if (dir) {
    VH_TRY(ws(ctx, WS_A, 6 * n, &aos));
    VH_TRY(dev_planar_to_interleaved(ctx, pdir, aos, (i64)n, 3, nullptr));
    VH_TRY(download(ctx, dir, aos, 3 * n));
  }
  return VISFD_HIP_OK;
}
int visfd_hip_membrane_detect(visfd_hip_ctx* ctx, const float* src, const float* mask, int64_t nx, int64_t ny,
                              int64_t nz, float sigma, float ratio, int order, float best_fraction,
                              float threshold_abs, float sigma_tv, int exponent, float cutoff, float* sal,
                              float* ten, float* dir, float* thr_out) {
  return visfd_hip_membrane_detect_bg(ctx, src, mask, nx, ny, nz, sigma, ratio, order, best_fraction, threshold_abs, sigma_tv,
                                      exponent, cutoff, 0.0f, 1, sal, ten, dir, thr_out);
}

// ---- a15 -------------------------------------------------------------------------------------
int visfd_hip_tensor_saliency_bg_dev(visfd_hip_ctx* ctx, const float* ten, const float* mask, int64_t nvox,
                                     int order, const float* image, const float* background, float* sal) {
  VH_REQUIRE(ctx && ten && sal && nvox > 0, "bad argument");
  VH_REQUIRE((image == nullptr) == (background == nullptr), "image and background come as a pair");
  VH_REQUIRE(order == 0 || order == 1, "unsupported eigenvalue order");
  VH_HIP(hipSetDevice(ctx->device));
  return dev_tensor_saliency(ctx, ten, mask, nvox, order, sal, image, background);
}
int visfd_hip_tensor_saliency_dev(visfd_hip_ctx* ctx, const float* ten, const float* mask, int64_t nvox,
                                  int order, float* sal) {
  return visfd_hip_tensor_saliency_bg_dev(ctx, ten, mask, nvox, order, nullptr, nullptr, sal);
}

int visfd_hip_tensor_saliency(visfd_hip_ctx* ctx, const float* ten, const float* mask, int64_t nvox,
                              int order, float* sal) {
  VH_REQUIRE(ctx && ten && sal && nvox > 0, "bad argument");
  VH_REQUIRE(order == 0 || order == 1, "unsupported eigenvalue order");
  VH_HIP(hipSetDevice(ctx->device));
  const size_t n = (size_t)nvox;
  float *aos, *pten, *dm, *dsal;
  VH_TRY(upload(ctx, WS_H2D_0, ten, 6 * n, &aos));
  VH_TRY(ws(ctx, WS_H2D_1, 6 * n, &pten));
  VH_TRY(dev_interleaved_to_planar(ctx, aos, pten, nvox, 6));
  VH_TRY(upload(ctx, WS_H2D_2, mask, n, &dm));
  VH_TRY(upload(ctx, WS_H2D_3, sal, n, &dsal));
  VH_TRY(dev_tensor_saliency(ctx, pten, dm, nvox, order, dsal));
  return download(ctx, sal, dsal, n);
}

}  // extern "C"
