// slab.hip -- the C-ABI face of the Z-slab decomposition (the reference is single-process, so there is
// no reference counterpart: these entry points are what a C++ host -- filter_mrc started once per GPU -- calls so that
// volumes larger than one GPU's HBM run across the GPUs of a node).
//
// One process per GPU.  A rank stores its owned planes [z0, z1) plus `ghost` planes on each INTERIOR face; every
// "outside the image" rule of the reference fires at the true faces of the volume only.  The transport is
//   * RCCL (loaded at run time from librccl.so: the library has no link-time dependency on it): grouped ncclSend/ncclRecv
//     with the two Z-neighbours -- one xGMI link each -- on a transfer stream of its own, and three all-reduces of a
//     2048-counter histogram for the exact global top-fraction threshold (handlers.cpp:1751-1797); or
//   * caller-provided callbacks (visfd_hip_transport): what the world-2 tests drive the same code with, and what an MPI
//     host would plug in.
// OVERLAP.  The (saliency, direction) halo of the voting stage travels on the transfer stream while the main stream votes
// the interior receiver planes (those whose windows touch owned sender planes only); the two bands next to the interior
// faces follow once the halo has landed.  The voting kernels are persistent grids sized to fill every wave slot of the
// chip, and a transfer kernel that arrives behind such a grid would wait for a workgroup to EXIT -- i.e. for the whole
// interior vote.  So while a halo is in flight the grid is capped (option tv_reserve_wg) to leave `reserve_wg` workgroup
// slots (default 64 = 16 CUs' worth) free for the transport's kernels.
#include <dlfcn.h>

#include <cmath>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace vh {
int select_pick_digit(const uint64_t* hist, uint64_t* k);   // select.hip
float select_key_to_float(uint32_t key);
}  // namespace vh

using namespace vh;

namespace {

// ---- RCCL through dlopen ---------------------------------------------------------------------------------------
typedef void* ncclComm_p;
struct NcclId { char internal[128]; };
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(ncclComm_p*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(ncclComm_p) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclUint64 = 5, kNcclFloat = 7, kNcclSum = 0;   // rccl.h: ncclDataType_t, ncclRedOp_t

int load_rccl(Rccl** out) {
  static Rccl r;
  static std::once_flag once;
  static std::string err;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) { err = "librccl.so not found (dlopen)"; return; }
#define VH_SYM(field, sym)                                                        \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym));               \
  if (!r.field) { err = std::string("librccl.so lacks ") + sym; return; }
    VH_SYM(GetUniqueId, "ncclGetUniqueId") VH_SYM(CommInitRank, "ncclCommInitRank") VH_SYM(CommDestroy, "ncclCommDestroy")
    VH_SYM(GroupStart, "ncclGroupStart") VH_SYM(GroupEnd, "ncclGroupEnd") VH_SYM(Send, "ncclSend") VH_SYM(Recv, "ncclRecv")
    VH_SYM(AllReduce, "ncclAllReduce") VH_SYM(GetErrorString, "ncclGetErrorString")
#undef VH_SYM
  });
  if (!err.empty()) return fail(VISFD_HIP_EDEVICE, "RCCL transport unavailable: " + err);
  *out = &r;
  return VISFD_HIP_OK;
}

}  // namespace

struct visfd_hip_slab {
  visfd_hip_ctx* ctx = nullptr;
  int rank = 0, world = 1, ghost = 0;
  i64 nz_global = 0, z0 = 0, z1 = 0, lo = 0, hi = 0;   // owned [z0, z1), stored [lo, hi) (global plane indices)
  i64 own0 = 0, own1 = 0, nz_local = 0;                // the owned planes inside the local array
  Rccl* rccl = nullptr;
  ncclComm_p comm = nullptr;
  visfd_hip_transport custom = {};
  bool use_custom = false;
  hipStream_t xfer = nullptr;                          // transfers run here, ordered against ctx->stream by events
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  uint64_t* hist_dev = nullptr;
  int reserve_wg = 64;
  bool in_flight = false;
};

namespace {

#define VH_NCCL(s, expr)                                                                         \
  do {                                                                                           \
    int _r = (expr);                                                                             \
    if (_r != 0) return fail(VISFD_HIP_EDEVICE, std::string(#expr) + ": " + (s)->rccl->GetErrorString(_r)); \
  } while (0)

int slab_common(visfd_hip_ctx* ctx, int rank, int world, i64 nz_global, int ghost, visfd_hip_slab** out) {
  VH_REQUIRE(ctx && out, "null argument");
  VH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank / world size");
  VH_REQUIRE(nz_global >= world && ghost >= 0, "bad slab geometry");
  visfd_hip_slab* s = new visfd_hip_slab();
  s->ctx = ctx; s->rank = rank; s->world = world; s->ghost = ghost; s->nz_global = nz_global;
  const i64 base = nz_global / world, rem = nz_global % world;
  s->z0 = rank * base + std::min<i64>(rank, rem);
  s->z1 = s->z0 + base + (rank < rem ? 1 : 0);
  s->lo = std::max<i64>(0, s->z0 - ghost);
  s->hi = std::min<i64>(nz_global, s->z1 + ghost);
  s->nz_local = s->hi - s->lo;
  s->own0 = s->z0 - s->lo;
  s->own1 = s->z1 - s->lo;
  if (world > 1 && s->z1 - s->z0 < ghost) {
    delete s;
    return fail(VISFD_HIP_EINVAL, "slabs thinner than the ghost depth are not supported");
  }
  VH_HIP(hipSetDevice(ctx->device));
  VH_HIP(hipStreamCreateWithFlags(&s->xfer, hipStreamNonBlocking));
  VH_HIP(hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming));
  VH_HIP(hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming));
  VH_HIP(hipMalloc(reinterpret_cast<void**>(&s->hist_dev), sizeof(uint64_t) * 2048));
  *out = s;
  return VISFD_HIP_OK;
}

// The ghost planes of `nvol` volumes within `depth` planes of the owned range, from the Z-neighbours' owned planes, as
// ONE group on the transfer stream: queued behind everything the main stream has done so far (ev_ready); the main stream
// is NOT made to wait here -- halo_wait() does that.
int halo_start(visfd_hip_slab* s, float* const* vols, int nvol, i64 nx, i64 ny, int depth) {
  if (s->world == 1 || depth == 0) return VISFD_HIP_OK;
  VH_REQUIRE(depth <= s->ghost, "halo deeper than the ghost zone");
  VH_REQUIRE(!s->in_flight, "a halo exchange is already in flight");
  const size_t plane = (size_t)nx * ny, cnt = plane * (size_t)depth;
  const int up = s->rank + 1, down = s->rank - 1;
  VH_HIP(hipEventRecord(s->ev_ready, s->ctx->stream));
  VH_HIP(hipStreamWaitEvent(s->xfer, s->ev_ready, 0));
  if (s->use_custom) {
    if (s->custom.group_start) VH_REQUIRE(s->custom.group_start(s->custom.user) == 0, "transport: group_start failed");
  } else {
    VH_NCCL(s, s->rccl->GroupStart());
  }
  for (int v = 0; v < nvol; v++) {
    float* t = vols[v];
    if (down >= 0) {
      const float* send = t + (size_t)s->own0 * plane;
      float* recv = t + (size_t)(s->own0 - depth) * plane;
      if (s->use_custom) {
        VH_REQUIRE(s->custom.sendrecv(s->custom.user, down, send, recv, cnt * sizeof(float), (void*)s->xfer) == 0, "transport: sendrecv failed");
      } else {
        VH_NCCL(s, s->rccl->Send(send, cnt, kNcclFloat, down, s->comm, s->xfer));
        VH_NCCL(s, s->rccl->Recv(recv, cnt, kNcclFloat, down, s->comm, s->xfer));
      }
    }
    if (up < s->world) {
      const float* send = t + (size_t)(s->own1 - depth) * plane;
      float* recv = t + (size_t)s->own1 * plane;
      if (s->use_custom) {
        VH_REQUIRE(s->custom.sendrecv(s->custom.user, up, send, recv, cnt * sizeof(float), (void*)s->xfer) == 0, "transport: sendrecv failed");
      } else {
        VH_NCCL(s, s->rccl->Send(send, cnt, kNcclFloat, up, s->comm, s->xfer));
        VH_NCCL(s, s->rccl->Recv(recv, cnt, kNcclFloat, up, s->comm, s->xfer));
      }
    }
  }
  if (s->use_custom) {
    if (s->custom.group_end) VH_REQUIRE(s->custom.group_end(s->custom.user) == 0, "transport: group_end failed");
  } else {
    VH_NCCL(s, s->rccl->GroupEnd());
  }
  VH_HIP(hipEventRecord(s->ev_done, s->xfer));
  s->in_flight = true;
  return VISFD_HIP_OK;
}

int halo_wait(visfd_hip_slab* s) {
  if (!s->in_flight) return VISFD_HIP_OK;
  VH_HIP(hipStreamWaitEvent(s->ctx->stream, s->ev_done, 0));
  s->in_flight = false;
  return VISFD_HIP_OK;
}

// An error between halo_start and halo_wait must not leave the handle "in flight" for ever (every later exchange would be
// refused): on such a path the transfer is drained and the flag cleared.
struct HaloGuard {
  visfd_hip_slab* s;
  ~HaloGuard() {
    if (s->in_flight) {
      (void)hipStreamSynchronize(s->xfer);
      s->in_flight = false;
    }
  }
};

// Exact global k-th largest saliency over all ranks' owned voxels (three radix rounds, each all-reducing 2048 counters
// on the device), then every owned voxel below it is zeroed (handlers.cpp:1751-1797).
int global_threshold(visfd_hip_slab* s, float* sal_owned, i64 nvox, float fraction, float* thr_out) {
  visfd_hip_ctx* ctx = s->ctx;
  std::vector<uint64_t> h(2048);
  uint32_t prefix = 0, key = 0;
  uint64_t k = 0;
  const int shifts[3] = {21, 10, 0};
  for (int rnd = 0; rnd < 3; rnd++) {
    VH_TRY(dev_select_histogram_todev(ctx, sal_owned, nullptr, nvox, rnd, prefix, s->hist_dev));
    if (s->world > 1) {
      if (s->use_custom) {
        VH_REQUIRE(s->custom.allreduce_sum_u64(s->custom.user, s->hist_dev, 2048, (void*)ctx->stream) == 0, "transport: allreduce failed");
      } else {
        VH_NCCL(s, s->rccl->AllReduce(s->hist_dev, s->hist_dev, 2048, kNcclUint64, kNcclSum, s->comm, ctx->stream));
      }
    }
    VH_HIP(hipMemcpyAsync(h.data(), s->hist_dev, sizeof(uint64_t) * 2048, hipMemcpyDeviceToHost, ctx->stream));
    VH_HIP(hipStreamSynchronize(ctx->stream));
    if (rnd == 0) {
      uint64_t n = 0;
      for (uint64_t c : h) n += c;
      const float prod = (float)n * fraction;   // size_t -> float product (handlers.cpp:1781)
      k = (uint64_t)std::floor(prod);
      if (n == 0 || k >= n) return fail(VISFD_HIP_EINVAL, "threshold fraction selects no voxel");
    }
    const int digit = select_pick_digit(h.data(), &k);
    if (digit < 0) return fail(VISFD_HIP_EDEVICE, "radix select: inconsistent histogram");
    key |= (uint32_t)digit << shifts[rnd];
    if (rnd < 2) prefix = (prefix << 11) | (uint32_t)digit;
  }
  const float thr = select_key_to_float(key);
  if (thr_out) *thr_out = thr;
  return dev_apply_threshold(ctx, sal_owned, nvox, thr);
}

}  // namespace

extern "C" {

int visfd_hip_slab_rccl_available(void) {
  Rccl* r = nullptr;
  const int rc = load_rccl(&r);
  if (rc != VISFD_HIP_OK) set_error("");
  return rc == VISFD_HIP_OK ? 1 : 0;
}

int visfd_hip_slab_unique_id(void* id_out) {
  VH_REQUIRE(id_out, "null argument");
  Rccl* r = nullptr;
  VH_TRY(load_rccl(&r));
  NcclId id;
  const int rc = r->GetUniqueId(&id);
  if (rc != 0) return fail(VISFD_HIP_EDEVICE, std::string("ncclGetUniqueId: ") + r->GetErrorString(rc));
  std::memcpy(id_out, &id, sizeof(id));
  return VISFD_HIP_OK;
}

int visfd_hip_slab_create_rccl(visfd_hip_ctx* ctx, const void* unique_id, int rank, int world, int64_t nz_global, int ghost,
                               visfd_hip_slab** out) {
  VH_REQUIRE(unique_id || world == 1, "null unique id");
  visfd_hip_slab* s = nullptr;
  VH_TRY(slab_common(ctx, rank, world, nz_global, ghost, &s));
  if (world > 1 || unique_id) {   // (world == 1 with an id: a one-rank communicator, for visfd_hip_slab_selftest)
    int rc = load_rccl(&s->rccl);
    if (rc != VISFD_HIP_OK) { visfd_hip_slab_destroy(s); return rc; }
    NcclId id;
    std::memcpy(&id, unique_id, sizeof(id));
    rc = s->rccl->CommInitRank(&s->comm, world, id, rank);
    if (rc != 0) {
      const std::string msg = std::string("ncclCommInitRank: ") + s->rccl->GetErrorString(rc);
      visfd_hip_slab_destroy(s);
      return fail(VISFD_HIP_EDEVICE, msg);
    }
  }
  *out = s;
  return VISFD_HIP_OK;
}

int visfd_hip_slab_create_custom(visfd_hip_ctx* ctx, const visfd_hip_transport* tr, int rank, int world, int64_t nz_global,
                                 int ghost, visfd_hip_slab** out) {
  VH_REQUIRE(tr && tr->sendrecv && tr->allreduce_sum_u64, "the transport needs sendrecv and allreduce_sum_u64");
  visfd_hip_slab* s = nullptr;
  VH_TRY(slab_common(ctx, rank, world, nz_global, ghost, &s));
  s->custom = *tr;
  s->use_custom = true;
  *out = s;
  return VISFD_HIP_OK;
}

int visfd_hip_slab_destroy(visfd_hip_slab* s) {
  if (!s) return VISFD_HIP_OK;
  (void)hipSetDevice(s->ctx->device);
  if (s->xfer) (void)hipStreamSynchronize(s->xfer);
  if (s->comm && s->rccl) (void)s->rccl->CommDestroy(s->comm);
  if (s->hist_dev) (void)hipFree(s->hist_dev);
  if (s->ev_ready) (void)hipEventDestroy(s->ev_ready);
  if (s->ev_done) (void)hipEventDestroy(s->ev_done);
  if (s->xfer) (void)hipStreamDestroy(s->xfer);
  delete s;
  return VISFD_HIP_OK;
}

int visfd_hip_slab_layout(visfd_hip_slab* s, int64_t out[7]) {
  VH_REQUIRE(s && out, "null argument");
  out[0] = s->z0; out[1] = s->z1; out[2] = s->lo; out[3] = s->hi; out[4] = s->own0; out[5] = s->own1; out[6] = s->nz_local;
  return VISFD_HIP_OK;
}

// The transport's own smoke test: a grouped send/receive of `count` floats from this rank to ITSELF on the transfer stream
// (RCCL allows self send/recv inside a group) and an all-reduce of 2048 counters on the context's stream, both verified.
// With one rank per GPU it checks the wiring of every rank; with a one-rank communicator (world == 1 created with an id) it
// is what a one-GPU box can run of the RCCL path: the run-time loading of librccl.so, the call signatures and enums, the
// stream/event ordering.
int visfd_hip_slab_selftest(visfd_hip_slab* s, int64_t count) {
  VH_REQUIRE(s && count > 0, "bad argument");
  VH_REQUIRE(s->use_custom || s->comm, "the slab has no communicator (world == 1 created without an id)");
  visfd_hip_ctx* ctx = s->ctx;
  VH_HIP(hipSetDevice(ctx->device));
  float *a = nullptr, *b = nullptr;
  VH_HIP(hipMalloc(reinterpret_cast<void**>(&a), sizeof(float) * (size_t)count));
  VH_HIP(hipMalloc(reinterpret_cast<void**>(&b), sizeof(float) * (size_t)count));
  std::vector<float> h((size_t)count), back((size_t)count, 0.0f);
  for (int64_t i = 0; i < count; i++) h[(size_t)i] = (float)(i % 977) + 0.25f * (float)s->rank;
  int rc = VISFD_HIP_OK;
  auto check = [&](hipError_t e) { if (e != hipSuccess && rc == VISFD_HIP_OK) rc = fail(VISFD_HIP_EDEVICE, hipGetErrorString(e)); };
  check(hipMemcpyAsync(a, h.data(), sizeof(float) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
  check(hipMemsetAsync(b, 0, sizeof(float) * (size_t)count, ctx->stream));
  check(hipEventRecord(s->ev_ready, ctx->stream));
  check(hipStreamWaitEvent(s->xfer, s->ev_ready, 0));
  if (rc == VISFD_HIP_OK) {
    if (s->use_custom) {
      if (s->custom.group_start && s->custom.group_start(s->custom.user) != 0) rc = fail(VISFD_HIP_EDEVICE, "transport: group_start failed");
      if (rc == VISFD_HIP_OK && s->custom.sendrecv(s->custom.user, s->rank, a, b, sizeof(float) * (size_t)count, (void*)s->xfer) != 0)
        rc = fail(VISFD_HIP_EDEVICE, "transport: sendrecv failed");
      if (rc == VISFD_HIP_OK && s->custom.group_end && s->custom.group_end(s->custom.user) != 0) rc = fail(VISFD_HIP_EDEVICE, "transport: group_end failed");
    } else {
      int r = s->rccl->GroupStart();
      if (r == 0) r = s->rccl->Send(a, (size_t)count, kNcclFloat, s->rank, s->comm, s->xfer);
      if (r == 0) r = s->rccl->Recv(b, (size_t)count, kNcclFloat, s->rank, s->comm, s->xfer);
      const int r2 = s->rccl->GroupEnd();
      if (r == 0) r = r2;
      if (r != 0) rc = fail(VISFD_HIP_EDEVICE, std::string("RCCL self send/recv: ") + s->rccl->GetErrorString(r));
    }
  }
  check(hipEventRecord(s->ev_done, s->xfer));
  check(hipStreamWaitEvent(ctx->stream, s->ev_done, 0));
  check(hipMemcpyAsync(back.data(), b, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
  // all-reduce: every rank contributes i + 1 in counter i
  std::vector<uint64_t> hh(2048), hs(2048);
  for (int i = 0; i < 2048; i++) hh[(size_t)i] = (uint64_t)i + 1;
  check(hipMemcpyAsync(s->hist_dev, hh.data(), sizeof(uint64_t) * 2048, hipMemcpyHostToDevice, ctx->stream));
  if (rc == VISFD_HIP_OK) {
    if (s->use_custom) {
      if (s->custom.allreduce_sum_u64(s->custom.user, s->hist_dev, 2048, (void*)ctx->stream) != 0) rc = fail(VISFD_HIP_EDEVICE, "transport: allreduce failed");
    } else {
      const int r = s->rccl->AllReduce(s->hist_dev, s->hist_dev, 2048, kNcclUint64, kNcclSum, s->comm, ctx->stream);
      if (r != 0) rc = fail(VISFD_HIP_EDEVICE, std::string("ncclAllReduce: ") + s->rccl->GetErrorString(r));
    }
  }
  check(hipMemcpyAsync(hs.data(), s->hist_dev, sizeof(uint64_t) * 2048, hipMemcpyDeviceToHost, ctx->stream));
  check(hipStreamSynchronize(ctx->stream));
  (void)hipFree(a);
  (void)hipFree(b);
  if (rc != VISFD_HIP_OK) return rc;
  for (int64_t i = 0; i < count; i++)
    if (back[(size_t)i] != h[(size_t)i]) return fail(VISFD_HIP_EDEVICE, "slab self-test: the self send/receive returned wrong data at element " + std::to_string(i));
  for (int i = 0; i < 2048; i++)
    if (hs[(size_t)i] != (uint64_t)s->world * ((uint64_t)i + 1))
      return fail(VISFD_HIP_EDEVICE, "slab self-test: the all-reduce returned a wrong sum in counter " + std::to_string(i));
  return VISFD_HIP_OK;
}

int visfd_hip_slab_set_reserve(visfd_hip_slab* s, int reserve_wg) {
  VH_REQUIRE(s && reserve_wg >= 0, "bad argument");
  s->reserve_wg = reserve_wg;
  return VISFD_HIP_OK;
}

int visfd_hip_slab_exchange_dev(visfd_hip_slab* s, float* const* volumes, int nvol, int64_t nx, int64_t ny, int depth) {
  VH_REQUIRE(s && volumes && nvol >= 1 && nx > 0 && ny > 0, "bad argument");
  VH_HIP(hipSetDevice(s->ctx->device));
  VH_TRY(halo_start(s, volumes, nvol, nx, ny, depth));
  return halo_wait(s);
}

// HandleTV (handlers.cpp:1501-1892) on one slab.  All volumes have the local shape [nz_local][ny][nx] (dirs: 3 planar
// channels, tensor: 6); src holds the owned planes (its ghost planes are filled here unless src_halo_ready); valid
// results are the OWNED planes of sal (the post-vote score) and tensor.
int visfd_hip_membrane_detect_slab_bg_dev(visfd_hip_slab* s, float* src, float* sal, float* dirs, float* tensor, float* scratch,
                                          float* background, int64_t nx, int64_t ny, float sigma, float ratio, int order,
                                          float best_fraction, float sigma_tv, int exponent, float cutoff, float sigma_background,
                                          int normalize_background, int src_halo_ready, float* thr_out) {
  VH_REQUIRE(s && src && sal && dirs && tensor && scratch, "null argument");
  VH_REQUIRE(sigma_background <= 0.0f || background, "the peak-height factor needs a volume for the background");
  VH_REQUIRE(order == 0 || order == 1, "unsupported eigenvalue order");
  visfd_hip_ctx* ctx = s->ctx;
  VH_HIP(hipSetDevice(ctx->device));
  const i64 nzl = s->nz_local, plane = nx * ny, nvl = plane * nzl;
  VH_TRY(check_dims(nx, ny, nzl));
  const int h_gauss = (int)std::floor(sigma * ratio);
  const int h_tv = host_tv_halfwidth(sigma_tv, cutoff);
  const int h_bg = sigma_background > 0.0f ? (int)std::floor(sigma_background * ratio) : 0;
  VH_REQUIRE(s->world == 1 || (h_gauss + 1 <= s->ghost && h_tv <= s->ghost && h_bg <= s->ghost), "ghost depth too small for this window");
  // 1. source halo deep enough for smoothing + the finite-difference stencil (and for the background filter)
  if (!src_halo_ready) {
    float* v[1] = {src};
    VH_TRY(halo_start(s, v, 1, nx, ny, std::min(s->ghost, std::max(h_gauss + 1, h_bg))));
    VH_TRY(halo_wait(s));
  }
  // optional peak-height factor: the background on every stored plane (owned planes exact, as the scores below)
  float* bg = nullptr;
  if (sigma_background > 0.0f) {
    bg = background;
    VH_TRY(visfd_hip_peak_background_dev(ctx, src, nullptr, nx, ny, nzl, sigma_background, ratio, normalize_background, bg));
  }
  // 2. scores on every stored plane (planes closer than h_gauss + 1 to an interior array end are garbage; owned planes exact)
  VH_TRY(visfd_hip_ridge_scores_bg_dev(ctx, src, nullptr, nx, ny, nzl, sigma, ratio, order, bg, sal, scratch));
  // 3. global top-fraction threshold over the owned voxels
  float thr = 0.0f;
  VH_TRY(global_threshold(s, sal + s->own0 * plane, (s->own1 - s->own0) * plane, best_fraction, &thr));
  if (thr_out) *thr_out = thr;
  VH_TRY(dev_ridge_directions(ctx, scratch, sal, nx, ny, nzl, sigma, order, dirs));
  // 4. stored planes beyond the voting halo must not vote; then the four channels of the halo as ONE group
  if (s->own0 - h_tv > 0) VH_HIP(hipMemsetAsync(sal, 0, sizeof(float) * (size_t)((s->own0 - h_tv) * plane), ctx->stream));
  if (s->own1 + h_tv < nzl)
    VH_HIP(hipMemsetAsync(sal + (s->own1 + h_tv) * plane, 0, sizeof(float) * (size_t)((nzl - s->own1 - h_tv) * plane), ctx->stream));
  float* chans[4] = {sal, dirs, dirs + nvl, dirs + 2 * nvl};
  VH_TRY(halo_start(s, chans, 4, nx, ny, std::min(s->ghost, h_tv)));
  HaloGuard guard{s};
  auto vote = [&](i64 za, i64 zb) -> int {
    return dev_tv_dense_stick(ctx, sal, dirs, tensor, nullptr, nullptr, nx, ny, nzl, za, zb, sigma_tv, exponent, cutoff, false);
  };
  const i64 lo_band = s->own0 + (s->rank > 0 ? h_tv : 0);             // first receiver plane that needs no ghost plane
  const i64 hi_band = s->own1 - (s->rank < s->world - 1 ? h_tv : 0);
  if (s->world > 1 && hi_band > lo_band) {
    // 5a. the interior, beside the transfer: leave workgroup slots free for the transport's kernels
    // (option tv_reserve_wg: each voting kernel subtracts it from its OWN chip-filling grid -- 4 workgroups per CU for the
    // exact kernel, 3 for the tolerance kernel)
    const int saved = ctx->opt.tv_reserve_wg;
    ctx->opt.tv_reserve_wg = std::max(saved, s->reserve_wg);
    const int rc = vote(lo_band, hi_band);
    ctx->opt.tv_reserve_wg = saved;
    VH_TRY(rc);
    VH_TRY(halo_wait(s));
    if (lo_band > s->own0) VH_TRY(vote(s->own0, lo_band));             // 5b. the bands that read ghost planes
    if (hi_band < s->own1) VH_TRY(vote(hi_band, s->own1));
  } else {
    VH_TRY(halo_wait(s));
    VH_TRY(vote(s->own0, s->own1));
  }
  // 6. post-vote score
  return dev_tensor_saliency(ctx, tensor, nullptr, nvl, order, sal, bg ? src : nullptr, bg);
}
int visfd_hip_membrane_detect_slab_dev(visfd_hip_slab* s, float* src, float* sal, float* dirs, float* tensor, float* scratch,
                                       int64_t nx, int64_t ny, float sigma, float ratio, int order, float best_fraction,
                                       float sigma_tv, int exponent, float cutoff, int src_halo_ready, float* thr_out) {
  return visfd_hip_membrane_detect_slab_bg_dev(s, src, sal, dirs, tensor, scratch, nullptr, nx, ny, sigma, ratio, order, best_fraction,
                                               sigma_tv, exponent, cutoff, 0.0f, 1, src_halo_ready, thr_out);
}

// The same stage for a host that keeps its volume in HOST memory (the filter_mrc program started once per GPU): the owned
// planes go up, the slab stage runs, the owned planes of the score (and, if asked for, of the vote tensors, six interleaved
// floats per voxel as visfd_hip_membrane_detect returns them) come back.  Device arrays live for the call only.
int visfd_hip_membrane_detect_slab_bg(visfd_hip_slab* s, const float* src_owned, int64_t nx, int64_t ny, float sigma, float ratio,
                                      int order, float best_fraction, float sigma_tv, int exponent, float cutoff,
                                      float sigma_background, int normalize_background, float* sal_owned, float* tensor_owned,
                                      float* thr_out) {
  VH_REQUIRE(s && src_owned && sal_owned, "null argument");
  visfd_hip_ctx* ctx = s->ctx;
  VH_HIP(hipSetDevice(ctx->device));
  const i64 nzl = s->nz_local, plane = nx * ny, nvl = plane * nzl, nown = (s->own1 - s->own0) * plane;
  VH_TRY(check_dims(nx, ny, nzl));
  struct Block {
    float* p = nullptr;
    ~Block() { if (p) (void)hipFree(p); }
  } blk;
  const bool with_bg = sigma_background > 0.0f;
  const size_t total = (size_t)nvl * (with_bg ? 13 : 12) + (tensor_owned ? (size_t)nown * 12 : 0);
  if (hipMalloc(&blk.p, total * sizeof(float)) != hipSuccess) {
    (void)hipGetLastError();
    return fail(VISFD_HIP_ENOMEM, "membrane_detect_slab: device allocation of the slab volumes failed");
  }
  float* src = blk.p;
  float* sal = src + nvl;
  float* dirs = sal + nvl;
  float* ten = dirs + 3 * nvl;
  float* scratch = ten + 6 * nvl;
  VH_HIP(hipMemsetAsync(src, 0, sizeof(float) * (size_t)nvl, ctx->stream));     // ghost planes: filled by the exchange
  VH_HIP(hipMemcpyAsync(src + s->own0 * plane, src_owned, sizeof(float) * (size_t)nown, hipMemcpyHostToDevice, ctx->stream));
  float* bgv = with_bg ? blk.p + (total - (size_t)nvl) : nullptr;   // (behind everything else)
  VH_TRY(visfd_hip_membrane_detect_slab_bg_dev(s, src, sal, dirs, ten, scratch, bgv, nx, ny, sigma, ratio, order, best_fraction, sigma_tv,
                                               exponent, cutoff, sigma_background, normalize_background, 0, thr_out));
  VH_HIP(hipMemcpyAsync(sal_owned, sal + s->own0 * plane, sizeof(float) * (size_t)nown, hipMemcpyDeviceToHost, ctx->stream));
  if (tensor_owned) {
    float* packed = scratch + nvl;          // [6][nown] planar, then [nown][6]
    float* aos = packed + 6 * nown;
    for (int c = 0; c < 6; c++)
      VH_HIP(hipMemcpyAsync(packed + (size_t)c * nown, ten + (size_t)c * nvl + s->own0 * plane, sizeof(float) * (size_t)nown,
                            hipMemcpyDeviceToDevice, ctx->stream));
    VH_TRY(dev_planar_to_interleaved(ctx, packed, aos, nown, 6, nullptr));
    VH_HIP(hipMemcpyAsync(tensor_owned, aos, sizeof(float) * 6 * (size_t)nown, hipMemcpyDeviceToHost, ctx->stream));
  }
  VH_HIP(hipStreamSynchronize(ctx->stream));
  return VISFD_HIP_OK;
}
int visfd_hip_membrane_detect_slab(visfd_hip_slab* s, const float* src_owned, int64_t nx, int64_t ny, float sigma, float ratio,
                                   int order, float best_fraction, float sigma_tv, int exponent, float cutoff,
                                   float* sal_owned, float* tensor_owned, float* thr_out) {
  return visfd_hip_membrane_detect_slab_bg(s, src_owned, nx, ny, sigma, ratio, order, best_fraction, sigma_tv, exponent, cutoff,
                                           0.0f, 1, sal_owned, tensor_owned, thr_out);
}

// BlobDog (feature.hpp:53-427) on one slab: the lists hold the blobs of OWNED planes only, iz as GLOBAL plane index;
// merging the ranks' lists (and ratio thresholds, which need the global best score) is the host's job.
// ---- host-memory faces of the Gaussian and of the blob detector for a host that starts one process per GPU
// (`filter_mrc ... -slab RANK WORLD IDFILE` with -gauss / -blob): owned planes in, owned planes / owned blobs out ----------
namespace {
struct DevBlock {
  float* p = nullptr;
  ~DevBlock() { if (p) (void)hipFree(p); }
};
// the rank's stored planes on the device: zeros in the ghost planes (the exchange fills them), the owned planes from the host
int upload_owned(visfd_hip_slab* s, const float* src_owned, i64 nx, i64 ny, float* src) {
  visfd_hip_ctx* ctx = s->ctx;
  const i64 plane = nx * ny;
  VH_HIP(hipMemsetAsync(src, 0, sizeof(float) * (size_t)(plane * s->nz_local), ctx->stream));
  VH_HIP(hipMemcpyAsync(src + s->own0 * plane, src_owned, sizeof(float) * (size_t)((s->own1 - s->own0) * plane), hipMemcpyHostToDevice,
                        ctx->stream));
  return VISFD_HIP_OK;
}
}  // namespace

int visfd_hip_apply_gauss_slab(visfd_hip_slab* s, const float* src_owned, int64_t nx, int64_t ny, const float sigma[3],
                               const int hw[3], int normalize, float* dst_owned, float* A_out) {
  VH_REQUIRE(s && src_owned && dst_owned && sigma && hw, "null argument");
  visfd_hip_ctx* ctx = s->ctx;
  VH_HIP(hipSetDevice(ctx->device));
  const i64 nzl = s->nz_local, plane = nx * ny, nvl = plane * nzl;
  VH_TRY(check_dims(nx, ny, nzl));
  VH_REQUIRE(s->world == 1 || hw[2] <= s->ghost, "ghost depth too small for this window");
  DevBlock blk;
  if (hipMalloc(&blk.p, sizeof(float) * 2 * (size_t)nvl) != hipSuccess) {
    (void)hipGetLastError();
    return fail(VISFD_HIP_ENOMEM, "apply_gauss_slab: device allocation of two slab volumes failed");
  }
  float* src = blk.p;
  float* dst = src + nvl;
  VH_TRY(upload_owned(s, src_owned, nx, ny, src));
  float* v[1] = {src};
  VH_TRY(halo_start(s, v, 1, nx, ny, std::min(s->ghost, hw[2])));
  VH_TRY(halo_wait(s));
  // the normaliser follows GLOBAL plane indices (filter3d.hpp:1004-1021): only the true faces of the volume are borders
  VH_TRY(visfd_hip_apply_gauss_slab_dev(ctx, src, dst, nx, ny, nzl, s->lo, s->nz_global, sigma, hw, normalize, A_out));
  VH_HIP(hipMemcpyAsync(dst_owned, dst + s->own0 * plane, sizeof(float) * (size_t)((s->own1 - s->own0) * plane), hipMemcpyDeviceToHost,
                        ctx->stream));
  VH_HIP(hipStreamSynchronize(ctx->stream));
  return VISFD_HIP_OK;
}

int visfd_hip_blob_dog_slab(visfd_hip_slab* s, const float* src_owned, int64_t nx, int64_t ny, const float* blob_sigma, int n_sigma,
                            float delta, float ratio, float min_thr, float max_thr, visfd_hip_blob* minima, int64_t min_cap,
                            int64_t* n_min, visfd_hip_blob* maxima, int64_t max_cap, int64_t* n_max) {
  VH_REQUIRE(s && src_owned, "null argument");
  visfd_hip_ctx* ctx = s->ctx;
  VH_HIP(hipSetDevice(ctx->device));
  const i64 nzl = s->nz_local, plane = nx * ny;
  VH_TRY(check_dims(nx, ny, nzl));
  DevBlock blk;
  if (hipMalloc(&blk.p, sizeof(float) * (size_t)(plane * nzl)) != hipSuccess) {
    (void)hipGetLastError();
    return fail(VISFD_HIP_ENOMEM, "blob_dog_slab: device allocation of the slab volume failed");
  }
  VH_TRY(upload_owned(s, src_owned, nx, ny, blk.p));
  return visfd_hip_blob_dog_slab_dev(s, blk.p, nx, ny, blob_sigma, n_sigma, delta, ratio, min_thr, max_thr, 0, minima, min_cap, n_min,
                                     maxima, max_cap, n_max);
}

int visfd_hip_blob_dog_slab_dev(visfd_hip_slab* s, float* src, int64_t nx, int64_t ny, const float* blob_sigma, int n_sigma,
                                float delta, float ratio, float min_thr, float max_thr, int src_halo_ready,
                                visfd_hip_blob* minima, int64_t min_cap, int64_t* n_min, visfd_hip_blob* maxima,
                                int64_t max_cap, int64_t* n_max) {
  VH_REQUIRE(s && src && blob_sigma && n_sigma >= 1 && minima && maxima && n_min && n_max, "bad argument");
  VH_HIP(hipSetDevice(s->ctx->device));
  float smax = 0.0f;
  for (int i = 0; i < n_sigma; i++) smax = std::max(smax, blob_sigma[i]);
  const int depth = (int)std::floor(ratio * (double)smax * (1.0 + 0.5 * delta)) + 1;
  VH_REQUIRE(s->world == 1 || depth <= s->ghost, "ghost depth too small for the widest LoG");
  if (!src_halo_ready) {
    float* v[1] = {src};
    VH_TRY(halo_start(s, v, 1, nx, ny, std::min(s->ghost, depth)));
    VH_TRY(halo_wait(s));
  }
  int64_t nmin = 0, nmax = 0;
  const int rc = visfd_hip_blob_dog_dev(s->ctx, src, nullptr, nx, ny, s->nz_local, blob_sigma, n_sigma, nullptr, delta, ratio, min_thr,
                                        max_thr, 0, minima, min_cap, &nmin, maxima, max_cap, &nmax);
  if (rc == VISFD_HIP_ECAPACITY) {   // the caller retries with the capacities it is told here -- locally: the ghost planes are in place
    *n_min = nmin;                   // (src_halo_ready = 1 on the retry), so no other rank takes part
    *n_max = nmax;
    return rc;
  }
  VH_TRY(rc);
  auto own = [&](visfd_hip_blob* b, int64_t n, int64_t cap) -> int64_t {
    int64_t m = 0;
    for (int64_t i = 0; i < std::min(n, cap); i++)
      if (b[i].iz >= s->own0 && b[i].iz < s->own1) {
        b[m] = b[i];
        b[m].iz += (int32_t)s->lo;
        m++;
      }
    return m;
  };
  *n_min = own(minima, nmin, min_cap);
  *n_max = own(maxima, nmax, max_cap);
  return VISFD_HIP_OK;
}

}  // extern "C"
