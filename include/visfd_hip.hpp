// visfd_hip.hpp -- the visfd:: template API of the hot path, re-created on top of the C ABI
// (include/visfd_hip.h, libvisfd_hip.so).  Header-only, C++11, no dependency on the reference.
//
// A program written against the reference's headers for this path keeps compiling when it
// includes this file instead of <visfd.hpp>: the functions below have the reference's names,
// argument order, argument meaning and error behaviour (they throw visfd::VisfdErr), and take
// the same `float***` arrays indexed [iz][iy][ix].  The arrays must be CONTIGUOUS, as produced by
// visfd::Alloc3D (lib/visfd/alloc3d.hpp:25-67) or MrcSimple: &a[0][0][0] is handed to the C ABI.
//
// Reference signatures mirrored (file:line under the reference root):
//   ApplyGauss   lib/visfd/filter3d.hpp:1086-1097, :1161-1173, :1226-1237, :1297-1308
//                bin/filter_mrc/filter3d_variants.hpp:500-528 (ratio OR threshold)
//   ApplyDog     lib/visfd/filter3d.hpp:1338-1351          ApplyLog  :1428-1441, :1531-1544
//   ApplySeparable lib/visfd/filter3d.hpp:686-695  (with GenFilterGauss1D, filter1d.hpp:409)
//   BlobDog      lib/visfd/feature.hpp:53-77               BlobDogD  :446-470
//   CalcHessian  lib/visfd/feature.hpp:1203-1219
//   TV3D         lib/visfd/feature.hpp:1645-1647, TVDenseStick :1711-1901 (with its normalize / diagonalize_dest steps)
//   BlobDogNM / _BlobDogNM  bin/filter_mrc/feature_variants.hpp:393-580
//   Alloc3D / Dealloc3D  lib/visfd/alloc3d.hpp:25, :75
//   CompactMultiChannelImage3D  lib/visfd/multichannel_image3d.hpp:41-204
// Only Scalar = float is provided (the hot path and the CLI use float throughout).
#ifndef VISFD_HIP_HPP
#define VISFD_HIP_HPP

#include <array>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <exception>
#include <limits>
#include <ostream>
#include <string>
#include <utility>
#include <vector>

#include "visfd_hip.h"

namespace visfd {

// lib/visfd/err_visfd.hpp:15-22
class VisfdErr : public std::exception {
  std::string msg;
 public:
  VisfdErr(const char* description) : msg(description) {}
  VisfdErr(std::string description) : msg(description) {}
  virtual const char* what() const throw() { return msg.c_str(); }
  virtual ~VisfdErr() throw() {}
};

namespace hip_detail {

// One process-wide context (device 0 unless VISFD_HIP_DEVICE is set), created on first use.
inline visfd_hip_ctx* context() {
  static visfd_hip_ctx* ctx = nullptr;
  if (!ctx) {
    int dev = 0;
    if (const char* e = std::getenv("VISFD_HIP_DEVICE")) dev = std::atoi(e);
    if (visfd_hip_create(dev, nullptr, &ctx) != VISFD_HIP_OK)
      throw VisfdErr(std::string("visfd_hip: ") + visfd_hip_last_error());
  }
  return ctx;
}
inline void check(int rc) {
  if (rc != VISFD_HIP_OK) throw VisfdErr(std::string("visfd_hip: ") + visfd_hip_last_error());
}
template <typename T>
inline T* flat(T* const* const* a) { return a ? &a[0][0][0] : nullptr; }
inline const float* flat(float const* const* const* a) { return a ? &a[0][0][0] : nullptr; }
inline void require_contiguous(float const* const* const* a, const int size[3]) {
  if (!a) return;
  const float* base = &a[0][0][0];
  const size_t nx = size[0], ny = size[1];
  if (&a[size[2] - 1][size[1] - 1][0] != base + ((size_t)(size[2] - 1) * ny + (size[1] - 1)) * nx)
    throw VisfdErr("visfd_hip: 3-D arrays must be contiguous (allocate them with Alloc3D)");
}

}  // namespace hip_detail

// ---- lib/visfd/alloc3d.hpp ------------------------------------------------------------------
template <typename Entry, typename Integer>
Entry*** Alloc3D(const Integer size[3]) {
  const size_t nx = size[0], ny = size[1], nz = size[2];
  Entry*** a = new Entry**[nz];
  Entry** rows = new Entry*[nz * ny];
  Entry* data = new Entry[nz * ny * nx];
  for (size_t iz = 0; iz < nz; iz++) {
    a[iz] = rows + iz * ny;
    for (size_t iy = 0; iy < ny; iy++) a[iz][iy] = data + (iz * ny + iy) * nx;
  }
  return a;
}
template <typename Entry>
void Dealloc3D(Entry*** a) {
  if (a) {
    delete[] a[0][0];
    delete[] a[0];
    delete[] a;
  }
}

// ---- lib/visfd/multichannel_image3d.hpp:41-204 ------------------------------------------------------
// One pointer per voxel (nullptr where mask == 0) into one compact array of n_good_voxels * channels numbers in scan
// order: the container HandleTV keeps its Hessians and vote tensors in (bin/filter_mrc/handlers.cpp:1564-1565).
template <typename Scalar>
class CompactMultiChannelImage3D {
  // Written for this shim (only the public face -- aaaafI, nchannels(), the constructors, Resize -- follows the reference):
  // the channel values of the unmasked voxels live in one std::vector in scan order, and `aaaafI` is an Alloc3D table of
  // pointers into it that a single pass binds with a running cursor.
  std::vector<Scalar> values_;
  int channels_;
  int dims_[3];

  static bool masked_out(Scalar const* const* const* mask, int iz, int iy, int ix) { return mask && mask[iz][iy][ix] == 0.0; }

  void release() {
    if (aaaafI) Dealloc3D(aaaafI);
    aaaafI = nullptr;
    values_.clear();
    values_.shrink_to_fit();
  }
  // (re)build the pointer table over values_: voxel by voxel in scan order, unmasked voxels take consecutive records
  template <typename Keep>
  void bind(Keep keep) {
    aaaafI = Alloc3D<Scalar*>(dims_);
    Scalar* cursor = values_.empty() ? nullptr : &values_[0];
    for (int iz = 0; iz < dims_[2]; iz++)
      for (int iy = 0; iy < dims_[1]; iy++)
        for (int ix = 0; ix < dims_[0]; ix++) {
          const bool k = keep(iz, iy, ix);
          aaaafI[iz][iy][ix] = k ? cursor : nullptr;
          if (k) cursor += channels_;
        }
  }

 public:
  Scalar**** aaaafI;   // a 3-D array of pointers into the compact array (nullptr where the mask is zero)

  int nchannels() { return channels_; }
  explicit CompactMultiChannelImage3D(int set_n_channels_per_voxel) : channels_(set_n_channels_per_voxel), aaaafI(nullptr) {
    dims_[0] = dims_[1] = dims_[2] = 0;
  }
  CompactMultiChannelImage3D(int set_n_channels_per_voxel, int const set_image_size[3],
                             Scalar const* const* const* aaafMask = nullptr, std::ostream* pReportProgress = nullptr)
      : channels_(set_n_channels_per_voxel), aaaafI(nullptr) {
    dims_[0] = dims_[1] = dims_[2] = 0;
    Resize(set_image_size, aaafMask, pReportProgress);
  }
  void Resize(int const set_image_size[3], Scalar const* const* const* aaafMask = nullptr,
              std::ostream* pReportProgress = nullptr) {
    release();
    for (int d = 0; d < 3; d++) dims_[d] = set_image_size[d];
    size_t kept = 0;
    for (int iz = 0; iz < dims_[2]; iz++)
      for (int iy = 0; iy < dims_[1]; iy++)
        for (int ix = 0; ix < dims_[0]; ix++) kept += masked_out(aaafMask, iz, iy, ix) ? 0 : 1;
    if (pReportProgress)
      *pReportProgress << " -- allocating " << channels_ << " channels for " << kept << " of "
                       << (size_t)dims_[0] * dims_[1] * dims_[2] << " voxels\n";
    values_.assign(kept * (size_t)channels_, Scalar());
    bind([&](int iz, int iy, int ix) { return !masked_out(aaafMask, iz, iy, ix); });
  }
  ~CompactMultiChannelImage3D() { release(); }
  CompactMultiChannelImage3D(const CompactMultiChannelImage3D<Scalar>& source)
      : values_(source.values_), channels_(source.channels_), aaaafI(nullptr) {
    for (int d = 0; d < 3; d++) dims_[d] = source.dims_[d];
    if (!source.aaaafI) return;
    Scalar**** const theirs = source.aaaafI;
    bind([&](int iz, int iy, int ix) { return theirs[iz][iy][ix] != nullptr; });   // same voxels, same order: same records
  }
  void swap(CompactMultiChannelImage3D<Scalar>& other) {
    values_.swap(other.values_);   // (a vector's buffer moves with it: both pointer tables stay valid)
    std::swap(channels_, other.channels_);
    for (int d = 0; d < 3; d++) std::swap(dims_[d], other.dims_[d]);
    std::swap(aaaafI, other.aaaafI);
  }
  CompactMultiChannelImage3D(CompactMultiChannelImage3D<Scalar>&& other) : channels_(other.channels_), aaaafI(nullptr) {
    dims_[0] = dims_[1] = dims_[2] = 0;
    this->swap(other);
  }
  CompactMultiChannelImage3D<Scalar>& operator=(CompactMultiChannelImage3D<Scalar> source) {
    this->swap(source);
    return *this;
  }
};

// ---- lib/visfd/filter1d.hpp:26-390 (the fields the hot path's callers touch) -------------------
template <typename Scalar, typename Integer>
class Filter1D {
 public:
  std::vector<Scalar> storage;
  Scalar* afH;        // indexable from -halfwidth .. +halfwidth
  Integer halfwidth;
  Integer array_size;
  Filter1D() : afH(nullptr), halfwidth(-1), array_size(-1) {}
  explicit Filter1D(Integer h) { Resize(h); }
  Filter1D(const Filter1D& o) : storage(o.storage), halfwidth(o.halfwidth), array_size(o.array_size) {
    afH = storage.empty() ? nullptr : storage.data() + halfwidth;
  }
  Filter1D& operator=(const Filter1D& o) {
    storage = o.storage; halfwidth = o.halfwidth; array_size = o.array_size;
    afH = storage.empty() ? nullptr : storage.data() + halfwidth;
    return *this;
  }
  void Resize(Integer h) {
    halfwidth = h;
    array_size = 2 * h + 1;
    storage.assign((size_t)array_size, (Scalar)-1.0e38);
    afH = storage.data() + h;
  }
};

// lib/visfd/filter1d.hpp:409-460
inline Filter1D<float, int> GenFilterGauss1D(float sigma, int halfwidth, std::ostream* = nullptr) {
  Filter1D<float, int> f(halfwidth);
  hip_detail::check(visfd_hip_gauss_taps(sigma, halfwidth, f.storage.data()));
  return f;
}

// ---- lib/visfd/filter3d.hpp:686-695 ---------------------------------------------------------------
inline float ApplySeparable(int const image_size[3], float const* const* const* aaafSource,
                            float*** aaafDest, float const* const* const* aaafMask,
                            Filter1D<float, int> aFilter[3], bool normalize = true,
                            std::ostream* pReportProgress = nullptr) {
  hip_detail::require_contiguous(aaafSource, image_size);
  hip_detail::require_contiguous(aaafDest, image_size);
  hip_detail::require_contiguous(aaafMask, image_size);
  if (pReportProgress) *pReportProgress << "  progress: Applying Z, Y, X filters on the GPU" << std::endl;
  float A = 0;
  hip_detail::check(visfd_hip_separable3d(
      hip_detail::context(), hip_detail::flat(aaafSource), hip_detail::flat(aaafDest),
      hip_detail::flat(aaafMask), image_size[0], image_size[1], image_size[2],
      aFilter[0].storage.data(), aFilter[0].halfwidth, aFilter[1].storage.data(), aFilter[1].halfwidth,
      aFilter[2].storage.data(), aFilter[2].halfwidth, normalize ? 1 : 0, &A));
  return A;
}

// ---- ApplyGauss: lib/visfd/filter3d.hpp:1086 (sigma[3], halfwidth[3]) --------------------------------
inline float ApplyGauss(const int image_size[3], float const* const* const* aaafSource, float*** aaafDest,
                        float const* const* const* aaafMask, float const sigma[3],
                        const int truncate_halfwidth[3], bool normalize = true,
                        std::ostream* pReportProgress = nullptr) {
  hip_detail::require_contiguous(aaafSource, image_size);
  hip_detail::require_contiguous(aaafDest, image_size);
  hip_detail::require_contiguous(aaafMask, image_size);
  if (pReportProgress) *pReportProgress << "  progress: Applying Z, Y, X filters on the GPU" << std::endl;
  float A = 0;
  hip_detail::check(visfd_hip_apply_gauss(hip_detail::context(), hip_detail::flat(aaafSource),
                                          hip_detail::flat(aaafDest), hip_detail::flat(aaafMask),
                                          image_size[0], image_size[1], image_size[2], sigma,
                                          truncate_halfwidth, normalize ? 1 : 0, &A));
  return A;
}
// :1161 (sigma, halfwidth)
inline float ApplyGauss(const int image_size[3], float const* const* const* src, float*** dest,
                        float const* const* const* mask, float sigma, int truncate_halfwidth,
                        bool normalize = true, std::ostream* pReportProgress = nullptr) {
  const float s[3] = {sigma, sigma, sigma};
  const int hw[3] = {truncate_halfwidth, truncate_halfwidth, truncate_halfwidth};
  return ApplyGauss(image_size, src, dest, mask, s, hw, normalize, pReportProgress);
}
// :1226 (sigma[3], truncate_ratio)
inline float ApplyGauss(const int image_size[3], float const* const* const* src, float*** dest,
                        float const* const* const* mask, const float sigma[3], float truncate_ratio = 2.5,
                        bool normalize = true, std::ostream* pReportProgress = nullptr) {
  int hw[3];
  hip_detail::check(visfd_hip_gauss_halfwidths(sigma, truncate_ratio, hw));
  return ApplyGauss(image_size, src, dest, mask, sigma, hw, normalize, pReportProgress);
}
// :1297 (sigma, truncate_ratio).  The reference falls off the end of this overload without a return
// statement (filter3d.hpp:1309-1319); here it returns the A coefficient like its siblings.
inline float ApplyGauss(const int image_size[3], float const* const* const* src, float*** dest,
                        float const* const* const* mask, float sigma, float truncate_ratio = 2.5,
                        bool normalize = true, std::ostream* pReportProgress = nullptr) {
  const float s[3] = {sigma, sigma, sigma};
  return ApplyGauss(image_size, src, dest, mask, s, truncate_ratio, normalize, pReportProgress);
}
// bin/filter_mrc/filter3d_variants.hpp:500-528 (ratio, or threshold when ratio <= 0)
inline float ApplyGauss(const int image_size[3], float const* const* const* src, float*** dest,
                        float const* const* const* mask, const float sigma[3], float filter_truncate_ratio,
                        float filter_truncate_threshold, bool normalize = true,
                        std::ostream* pReportProgress = nullptr) {
  if (filter_truncate_ratio <= 0) filter_truncate_ratio = visfd_hip_ratio_from_threshold(filter_truncate_threshold);
  return ApplyGauss(image_size, src, dest, mask, sigma, filter_truncate_ratio, normalize, pReportProgress);
}

// ---- ApplyDog: lib/visfd/filter3d.hpp:1338-1351 -------------------------------------------------------
inline void ApplyDog(const int image_size[3], float const* const* const* src, float*** dest,
                     float const* const* const* mask, float const sigma_a[3], float const sigma_b[3],
                     const int truncate_halfwidth[3], float* pA = nullptr, float* pB = nullptr,
                     std::ostream* = nullptr) {
  hip_detail::require_contiguous(src, image_size);
  hip_detail::require_contiguous(dest, image_size);
  hip_detail::require_contiguous(mask, image_size);
  hip_detail::check(visfd_hip_apply_dog(hip_detail::context(), hip_detail::flat(src), hip_detail::flat(dest),
                                        hip_detail::flat(mask), image_size[0], image_size[1], image_size[2],
                                        sigma_a, sigma_b, truncate_halfwidth, pA, pB));
}

// ---- ApplyLog: lib/visfd/filter3d.hpp:1428-1441 and :1531-1544 ----------------------------------------
inline void ApplyLog(const int image_size[3], float const* const* const* src, float*** dest,
                     float const* const* const* mask, const float sigma[3],
                     float delta_sigma_over_sigma = 0.02, float truncate_ratio = 2.5, float* pA = nullptr,
                     float* pB = nullptr, std::ostream* = nullptr) {
  hip_detail::require_contiguous(src, image_size);
  hip_detail::require_contiguous(dest, image_size);
  hip_detail::require_contiguous(mask, image_size);
  hip_detail::check(visfd_hip_apply_log(hip_detail::context(), hip_detail::flat(src), hip_detail::flat(dest),
                                        hip_detail::flat(mask), image_size[0], image_size[1], image_size[2], sigma,
                                        delta_sigma_over_sigma, truncate_ratio, pA, pB));
}
inline void ApplyLog(const int image_size[3], float const* const* const* src, float*** dest,
                     float const* const* const* mask, float sigma, float delta_sigma_over_sigma = 0.02,
                     float truncate_ratio = 2.5, float* pA = nullptr, float* pB = nullptr,
                     std::ostream* pReportProgress = nullptr) {
  const float s[3] = {sigma, sigma, sigma};
  ApplyLog(image_size, src, dest, mask, s, delta_sigma_over_sigma, truncate_ratio, pA, pB, pReportProgress);
}

// ---- LocalFluctuations: lib/visfd/filter3d.hpp:1698-1711 (Gaussian weights: exponent must be 2) -------
inline void LocalFluctuations(const int image_size[3], float const* const* const* src, float*** dest,
                              float const* const* const* mask, const float sigma[3],
                              float template_background_exponent = 2, float filter_truncate_ratio = 2.5,
                              bool normalize = true, std::ostream* = nullptr) {
  hip_detail::require_contiguous(src, image_size);
  hip_detail::require_contiguous(dest, image_size);
  hip_detail::require_contiguous(mask, image_size);
  hip_detail::check(visfd_hip_local_fluctuations(hip_detail::context(), hip_detail::flat(src), hip_detail::flat(dest),
                                                 hip_detail::flat(mask), image_size[0], image_size[1], image_size[2],
                                                 sigma, template_background_exponent, filter_truncate_ratio,
                                                 normalize ? 1 : 0));
}
// ---- LocalFluctuationsByRadius: lib/visfd/filter3d.hpp:1897-1926 and the variant with a decay threshold,
//      bin/filter_mrc/filter3d_variants.hpp:651-681 (a negative ratio selects the threshold) --------------
inline void LocalFluctuationsByRadius(const int image_size[3], float const* const* const* src, float*** dest,
                                      float const* const* const* mask, const float radius[3],
                                      float template_background_exponent, float filter_truncate_ratio,
                                      float filter_truncate_threshold, bool normalize = true,
                                      std::ostream* pReportProgress = nullptr) {
  float sigma[3], ratio;
  hip_detail::check(visfd_hip_fluctuation_sigmas(radius, template_background_exponent, filter_truncate_ratio,
                                                 filter_truncate_threshold, sigma, &ratio));
  LocalFluctuations(image_size, src, dest, mask, sigma, template_background_exponent, ratio, normalize,
                    pReportProgress);
}
inline void LocalFluctuationsByRadius(const int image_size[3], float const* const* const* src, float*** dest,
                                      float const* const* const* mask, const float radius[3],
                                      float template_background_exponent = 2, float filter_truncate_ratio = 2.5,
                                      bool normalize = true, std::ostream* pReportProgress = nullptr) {
  // a non-negative ratio is taken as is (the threshold argument is then unused)
  LocalFluctuationsByRadius(image_size, src, dest, mask, radius, template_background_exponent,
                            filter_truncate_ratio < 0 ? 0.0f : filter_truncate_ratio, 0.02f, normalize, pReportProgress);
}

// ---- BlobDog: lib/visfd/feature.hpp:53-77 --------------------------------------------------------------
// The optional preallocated-images argument of the reference (aaaafI) is accepted and ignored: the
// rolling LoG volumes live in HBM.
inline void BlobDog(int const image_size[3], float const* const* const* aaafSource,
                    float const* const* const* aaafMask, const std::vector<float>& blob_sigma,
                    std::vector<std::array<float, 3> >* pva_minima_crds = nullptr,
                    std::vector<std::array<float, 3> >* pva_maxima_crds = nullptr,
                    std::vector<float>* pv_minima_sigma = nullptr, std::vector<float>* pv_maxima_sigma = nullptr,
                    std::vector<float>* pv_minima_scores = nullptr, std::vector<float>* pv_maxima_scores = nullptr,
                    const float aspect_ratio[3] = nullptr, float delta_sigma_over_sigma = 0.02,
                    float truncate_ratio = 2.5,
                    float minima_threshold = std::numeric_limits<float>::infinity(),
                    float maxima_threshold = -std::numeric_limits<float>::infinity(),
                    bool use_threshold_ratios = true, std::ostream* pReportProgress = nullptr,
                    float**** /*aaaafI*/ = nullptr) {
  hip_detail::require_contiguous(aaafSource, image_size);
  hip_detail::require_contiguous(aaafMask, image_size);
  if (pReportProgress)
    *pReportProgress << "\n----- Blob detection initiated using " << blob_sigma.size()
                     << " trial Gaussians -----\n\n";
  int64_t cap = 1 << 16, nmin = 0, nmax = 0;
  std::vector<visfd_hip_blob> mins, maxs;
  for (int attempt = 0; attempt < 2; attempt++) {
    mins.resize((size_t)cap);
    maxs.resize((size_t)cap);
    int rc = visfd_hip_blob_dog(hip_detail::context(), hip_detail::flat(aaafSource), hip_detail::flat(aaafMask),
                                image_size[0], image_size[1], image_size[2], blob_sigma.data(),
                                (int)blob_sigma.size(), aspect_ratio, delta_sigma_over_sigma, truncate_ratio,
                                minima_threshold, maxima_threshold, use_threshold_ratios ? 1 : 0, mins.data(), cap,
                                &nmin, maxs.data(), cap, &nmax);
    if (rc == VISFD_HIP_ECAPACITY) { cap = (nmin > nmax ? nmin : nmax); continue; }
    hip_detail::check(rc);
    break;
  }
  for (int side = 0; side < 2; side++) {
    const std::vector<visfd_hip_blob>& L = side ? maxs : mins;
    const int64_t n = side ? nmax : nmin;
    std::vector<std::array<float, 3> >* crds = side ? pva_maxima_crds : pva_minima_crds;
    std::vector<float>* sig = side ? pv_maxima_sigma : pv_minima_sigma;
    std::vector<float>* sc = side ? pv_maxima_scores : pv_minima_scores;
    for (int64_t i = 0; i < n; i++) {
      if (crds) {
        std::array<float, 3> c;
        c[0] = (float)L[i].ix; c[1] = (float)L[i].iy; c[2] = (float)L[i].iz;
        crds->push_back(c);
      }
      if (sig) sig->push_back(L[i].sigma);
      if (sc) sc->push_back(L[i].score);
    }
  }
  if (pReportProgress)
    *pReportProgress << "--- (Found " << nmin << " and " << nmax
                     << " local minima and maxima, respectively) ---\n" << std::endl;
}

// ---- BlobDogD: lib/visfd/feature.hpp:446-470 ------------------------------------------------------------
inline void BlobDogD(int const image_size[3], float const* const* const* aaafSource,
                     float const* const* const* aaafMask, const std::vector<float>& blob_diameters,
                     std::vector<std::array<float, 3> >* pva_minima_crds = nullptr,
                     std::vector<std::array<float, 3> >* pva_maxima_crds = nullptr,
                     std::vector<float>* pv_minima_diameters = nullptr,
                     std::vector<float>* pv_maxima_diameters = nullptr,
                     std::vector<float>* pv_minima_scores = nullptr, std::vector<float>* pv_maxima_scores = nullptr,
                     const float aspect_ratio[3] = nullptr, float delta_sigma_over_sigma = 0.02,
                     float truncate_ratio = 2.5,
                     float minima_threshold = std::numeric_limits<float>::infinity(),
                     float maxima_threshold = -std::numeric_limits<float>::infinity(),
                     bool use_threshold_ratios = false, std::ostream* pReportProgress = nullptr,
                     float**** aaaafI = nullptr) {
  std::vector<float> blob_sigma(blob_diameters.size()), min_sig, max_sig;
  hip_detail::check(visfd_hip_blob_diameters_to_sigmas(blob_diameters.data(), (int)blob_diameters.size(),
                                                       blob_sigma.data()));
  BlobDog(image_size, aaafSource, aaafMask, blob_sigma, pva_minima_crds, pva_maxima_crds, &min_sig, &max_sig,
          pv_minima_scores, pv_maxima_scores, aspect_ratio, delta_sigma_over_sigma, truncate_ratio,
          minima_threshold, maxima_threshold, use_threshold_ratios, pReportProgress, aaaafI);
  if (pv_minima_diameters) {
    pv_minima_diameters->resize(min_sig.size());
    hip_detail::check(visfd_hip_blob_sigmas_to_diameters(min_sig.data(), (int)min_sig.size(),
                                                         pv_minima_diameters->data()));
  }
  if (pv_maxima_diameters) {
    pv_maxima_diameters->resize(max_sig.size());
    hip_detail::check(visfd_hip_blob_sigmas_to_diameters(max_sig.data(), (int)max_sig.size(),
                                                         pv_maxima_diameters->data()));
  }
}

// ---- BinArray3D / UnbinArray3D: lib/visfd/resample.hpp:53-58, :120-124 ---------------------------------
inline void BinArray3D(int const size_source[3], int const size_dest[3], float const* const* const* aaafSource,
                       float*** aaafDest, int const* offset = nullptr) {
  hip_detail::require_contiguous(aaafSource, size_source);
  hip_detail::require_contiguous(aaafDest, size_dest);
  const int64_t ss[3] = {size_source[0], size_source[1], size_source[2]}, ds[3] = {size_dest[0], size_dest[1], size_dest[2]};
  hip_detail::check(visfd_hip_bin_array3d(hip_detail::context(), hip_detail::flat(aaafSource), ss,
                                          hip_detail::flat(aaafDest), ds, offset));
}
inline void UnbinArray3D(int const size_source[3], int const size_dest[3], float const* const* const* aaafSource,
                         float*** aaafDest, int const* offset = nullptr) {
  hip_detail::require_contiguous(aaafSource, size_source);
  hip_detail::require_contiguous(aaafDest, size_dest);
  const int64_t ss[3] = {size_source[0], size_source[1], size_source[2]}, ds[3] = {size_dest[0], size_dest[1], size_dest[2]};
  hip_detail::check(visfd_hip_unbin_array3d(hip_detail::context(), hip_detail::flat(aaafSource), ss,
                                            hip_detail::flat(aaafDest), ds, offset));
}

// ---- blob list post-processing: lib/visfd/visfd_utils.hpp:49-55,95-118; feature.hpp:519-616,720-969 ------
typedef enum eSortCriteria {
  DO_NOT_SORT = VISFD_HIP_DO_NOT_SORT,
  SORT_DECREASING = VISFD_HIP_SORT_DECREASING,
  SORT_INCREASING = VISFD_HIP_SORT_INCREASING,
  SORT_DECREASING_MAGNITUDE = VISFD_HIP_SORT_DECREASING_MAGNITUDE,
  SORT_INCREASING_MAGNITUDE = VISFD_HIP_SORT_INCREASING_MAGNITUDE
} SortCriteria;

inline float CalcSphereOverlap(float rij, float Ri, float Rj) { return visfd_hip_sphere_overlap(rij, Ri, Rj); }

namespace hip_detail {
struct FlatBlobs {   // the three parallel vectors as the flat arrays of the C ABI, and back
  std::vector<float> c, d, s;
  FlatBlobs(const std::vector<std::array<float, 3> >& crds, const std::vector<float>& diam,
            const std::vector<float>& score) : c(3 * crds.size()), d(diam), s(score) {
    if (diam.size() != crds.size() || score.size() != crds.size())
      throw VisfdErr("Error: blob coordinate, diameter and score lists differ in length.\n");
    for (size_t i = 0; i < crds.size(); i++) { c[3 * i] = crds[i][0]; c[3 * i + 1] = crds[i][1]; c[3 * i + 2] = crds[i][2]; }
  }
  void store(size_t n, std::vector<std::array<float, 3> >& crds, std::vector<float>& diam, std::vector<float>& score) {
    crds.resize(n); diam.assign(d.begin(), d.begin() + n); score.assign(s.begin(), s.begin() + n);
    for (size_t i = 0; i < n; i++) { crds[i][0] = c[3 * i]; crds[i][1] = c[3 * i + 1]; crds[i][2] = c[3 * i + 2]; }
  }
};
}  // namespace hip_detail

inline void SortBlobs(std::vector<std::array<float, 3> >& blob_crds, std::vector<float>& blob_diameters,
                      std::vector<float>& blob_scores, SortCriteria sort_blob_criteria, bool ascending_order = true,
                      std::vector<size_t>* pPermutation = nullptr, std::ostream* pReportProgress = nullptr) {
  hip_detail::FlatBlobs f(blob_crds, blob_diameters, blob_scores);
  std::vector<uint64_t> perm(blob_crds.size());
  if (pReportProgress && !blob_crds.empty() && sort_blob_criteria != DO_NOT_SORT)
    *pReportProgress << "-- Sorting blobs according to their scores... ";
  hip_detail::check(visfd_hip_sort_blobs(f.c.data(), f.d.data(), f.s.data(), (int64_t)blob_crds.size(),
                                         (int)sort_blob_criteria, ascending_order ? 1 : 0, perm.data()));
  f.store(blob_crds.size(), blob_crds, blob_diameters, blob_scores);
  if (pPermutation && !blob_crds.empty() && sort_blob_criteria != DO_NOT_SORT) pPermutation->assign(perm.begin(), perm.end());
  if (pReportProgress && !blob_crds.empty() && sort_blob_criteria != DO_NOT_SORT) *pReportProgress << "done --" << std::endl;
}

// the (ascending_order, ignore_score_sign) form, feature.hpp:519-560
inline void SortBlobs(std::vector<std::array<float, 3> >& blob_crds, std::vector<float>& blob_diameters,
                      std::vector<float>& blob_scores, bool ascending_order = true, bool ignore_score_sign = true,
                      std::vector<size_t>* pPermutation = nullptr, std::ostream* pReportProgress = nullptr) {
  SortBlobs(blob_crds, blob_diameters, blob_scores, ignore_score_sign ? SORT_DECREASING_MAGNITUDE : SORT_DECREASING,
            ascending_order, pPermutation, pReportProgress);
}

// image_size is needed here (the reference indexes the mask unchecked): pass the mask's dimensions
inline void DiscardMaskedBlobs(std::vector<std::array<float, 3> >& blob_crds, std::vector<float>& blob_diameters,
                               std::vector<float>& blob_scores, float const* const* const* aaafMask,
                               int const image_size[3], std::ostream* pReportProgress = nullptr) {
  if (!aaafMask) return;
  hip_detail::FlatBlobs f(blob_crds, blob_diameters, blob_scores);
  int64_t n = (int64_t)blob_crds.size();
  hip_detail::check(visfd_hip_discard_masked_blobs(f.c.data(), f.d.data(), f.s.data(), &n, &aaafMask[0][0][0],
                                                   image_size[0], image_size[1], image_size[2]));
  if (pReportProgress)
    *pReportProgress << "  discarded " << (blob_crds.size() - (size_t)n) << "  blobs lying outside the mask." << std::endl;
  f.store((size_t)n, blob_crds, blob_diameters, blob_scores);
}

inline void DiscardOverlappingBlobs(std::vector<std::array<float, 3> >& blob_crds, std::vector<float>& blob_diameters,
                                    std::vector<float>& blob_scores, float min_radial_separation_ratio,
                                    float max_volume_overlap_large = std::numeric_limits<float>::infinity(),
                                    float max_volume_overlap_small = std::numeric_limits<float>::infinity(),
                                    SortCriteria sort_blob_criteria = SORT_DECREASING_MAGNITUDE,
                                    std::ostream* pReportProgress = nullptr, int scale = 6) {
  hip_detail::FlatBlobs f(blob_crds, blob_diameters, blob_scores);
  int64_t n = (int64_t)blob_crds.size();
  if (pReportProgress) *pReportProgress << "  detecting collisions between " << n << " blobs... ";
  hip_detail::check(visfd_hip_discard_overlapping_blobs(f.c.data(), f.d.data(), f.s.data(), &n,
                                                        min_radial_separation_ratio, max_volume_overlap_large,
                                                        max_volume_overlap_small, (int)sort_blob_criteria, scale));
  f.store((size_t)n, blob_crds, blob_diameters, blob_scores);
  if (pReportProgress) *pReportProgress << "done.\n";
}

// ---- BlobDogNM / _BlobDogNM: bin/filter_mrc/feature_variants.hpp:393-580 --------------------------------------
// BlobDogD followed by non-max suppression of overlapping blobs (minima best-first by increasing score, maxima by
// decreasing score); the `_` form takes the filter window either as a ratio or as a decay threshold.
inline void BlobDogNM(int const image_size[3], float const* const* const* aaafSource,
                      float const* const* const* aaafMask, const std::vector<float>& blob_diameters,
                      std::vector<std::array<float, 3> >* pva_minima_crds = nullptr,
                      std::vector<std::array<float, 3> >* pva_maxima_crds = nullptr,
                      std::vector<float>* pv_minima_diameters = nullptr,
                      std::vector<float>* pv_maxima_diameters = nullptr,
                      std::vector<float>* pv_minima_scores = nullptr, std::vector<float>* pv_maxima_scores = nullptr,
                      const float aspect_ratio[3] = nullptr, float delta_sigma_over_sigma = 0.02,
                      float truncate_ratio = 2.5, float minima_threshold = 0.5, float maxima_threshold = 0.5,
                      bool use_threshold_ratios = true, float sep_ratio_thresh = 1.0,
                      float nonmax_max_overlap_large = 1.0, float nonmax_max_overlap_small = 1.0,
                      std::ostream* pReportProgress = nullptr, float**** aaaafI = nullptr) {
  std::vector<std::array<float, 3> > minima_crds, maxima_crds;
  std::vector<float> minima_diameters, maxima_diameters, minima_scores, maxima_scores;
  if (!pva_minima_crds) pva_minima_crds = &minima_crds;
  if (!pva_maxima_crds) pva_maxima_crds = &maxima_crds;
  if (!pv_minima_diameters) pv_minima_diameters = &minima_diameters;
  if (!pv_maxima_diameters) pv_maxima_diameters = &maxima_diameters;
  if (!pv_minima_scores) pv_minima_scores = &minima_scores;
  if (!pv_maxima_scores) pv_maxima_scores = &maxima_scores;
  const float default_aspect_ratio[3] = {1.0f, 1.0f, 1.0f};
  BlobDogD(image_size, aaafSource, aaafMask, blob_diameters, pva_minima_crds, pva_maxima_crds, pv_minima_diameters,
           pv_maxima_diameters, pv_minima_scores, pv_maxima_scores, aspect_ratio ? aspect_ratio : default_aspect_ratio,
           delta_sigma_over_sigma, truncate_ratio, minima_threshold, maxima_threshold, use_threshold_ratios,
           pReportProgress, aaaafI);
  const bool discard_overlapping_blobs =
      (sep_ratio_thresh > 0.0f) || (nonmax_max_overlap_small < 1.0f) || (nonmax_max_overlap_large < 1.0f);
  if (!discard_overlapping_blobs) return;
  if (pReportProgress)
    *pReportProgress << "----------- Removing overlapping blobs -----------\n" << std::endl
                     << "--- Discarding overlapping minima blobs ---\n";
  DiscardOverlappingBlobs(*pva_minima_crds, *pv_minima_diameters, *pv_minima_scores, sep_ratio_thresh,
                          nonmax_max_overlap_large, nonmax_max_overlap_small, SORT_INCREASING, pReportProgress);
  if (pReportProgress) *pReportProgress << "done --\n" << "--- Discarding overlapping maxima blobs ---\n";
  DiscardOverlappingBlobs(*pva_maxima_crds, *pv_maxima_diameters, *pv_maxima_scores, sep_ratio_thresh,
                          nonmax_max_overlap_large, nonmax_max_overlap_small, SORT_DECREASING, pReportProgress);
}

inline void _BlobDogNM(int const image_size[3], float const* const* const* aaafSource,
                       float const* const* const* aaafMask, const std::vector<float>& blob_diameters,
                       std::vector<std::array<float, 3> >* pva_minima_crds = nullptr,
                       std::vector<std::array<float, 3> >* pva_maxima_crds = nullptr,
                       std::vector<float>* pv_minima_diameters = nullptr,
                       std::vector<float>* pv_maxima_diameters = nullptr,
                       std::vector<float>* pv_minima_scores = nullptr, std::vector<float>* pv_maxima_scores = nullptr,
                       const float aspect_ratio[3] = nullptr, float delta_sigma_over_sigma = 0.02,
                       float filter_truncate_ratio = 2.5, float filter_truncate_threshold = 0.02,
                       float minima_threshold = 0.0, float maxima_threshold = 0.0, bool use_threshold_ratios = true,
                       float sep_ratio_thresh = 1.0, float nonmax_max_overlap_large = 1.0,
                       float nonmax_max_overlap_small = 1.0, std::ostream* pReportProgress = nullptr,
                       float**** aaaafI = nullptr) {
  if (filter_truncate_ratio <= 0)   // exp(-ratio^2 / 2) = threshold, evaluated in double as the reference does (:543)
    filter_truncate_ratio = (float)std::sqrt(-2 * std::log((double)filter_truncate_threshold));
  BlobDogNM(image_size, aaafSource, aaafMask, blob_diameters, pva_minima_crds, pva_maxima_crds, pv_minima_diameters,
            pv_maxima_diameters, pv_minima_scores, pv_maxima_scores, aspect_ratio, delta_sigma_over_sigma,
            filter_truncate_ratio, minima_threshold, maxima_threshold, use_threshold_ratios, sep_ratio_thresh,
            nonmax_max_overlap_large, nonmax_max_overlap_small, pReportProgress, aaaafI);
}

// ---- LabelConnected: lib/visfd/connect.hpp:47-65, :168-197 ------------------------------------------------
// The form bin/filter_mrc/handlers.cpp:1985-2013 calls: Scalar = float, Label = ptrdiff_t, Coordinate = float,
// directions as array<float,3>*** (contiguous, Alloc3D), tensors as one float* per voxel (nullptr = no storage,
// e.g. CompactMultiChannelImage3D::aaaafI).
typedef enum eRegionSortCriteria { SORT_BY_VALUE, SORT_BY_SIZE } RegionSortCriteria;
typedef enum eDirectionPairType { SAME_DIRECTION, OPPOSITE_DIRECTION, AUTO } DirectionPairType;

inline size_t LabelConnected(
    const int image_size[3], float const* const* const* aaafSaliency, ptrdiff_t*** aaaiDest,
    float const* const* const* aaafMask,
    float threshold_saliency = -std::numeric_limits<float>::infinity(),
    std::array<float, 3> const* const* const* aaaafVector = nullptr,
    float threshold_vector_saliency = -std::numeric_limits<float>::infinity(),
    float threshold_vector_neighbor = -std::numeric_limits<float>::infinity(), bool consider_dot_product_sign = true,
    float* const* const* const* aaaafSymmetricTensor = nullptr,
    float threshold_tensor_saliency = -std::numeric_limits<float>::infinity(),
    float threshold_tensor_neighbor = -std::numeric_limits<float>::infinity(),
    bool tensor_is_positive_definite_near_target = true, int connectivity = 1, ptrdiff_t label_undefined = -1,
    std::vector<std::array<float, 3> >* pv_cluster_maxima = nullptr, std::vector<float>* pv_cluster_sizes = nullptr,
    std::vector<float>* pv_cluster_saliencies = nullptr, RegionSortCriteria sort_criteria = SORT_BY_SIZE,
    float const* const* const* aaafVoxelWeights = nullptr, std::array<float, 3>*** aaaafVectorStandardized = nullptr,
    const std::vector<std::vector<std::array<float, 3> > >* pMustLinkConstraints = nullptr,
    const std::vector<std::vector<DirectionPairType> >* pMustLinkDirections = nullptr,
    bool start_from_saliency_maxima = true, std::ostream* pReportProgress = nullptr) {
  static_assert(sizeof(ptrdiff_t) == sizeof(int64_t), "labels travel as 64-bit integers");
  hip_detail::require_contiguous(aaafSaliency, image_size);
  hip_detail::require_contiguous(aaafVoxelWeights, image_size);
  // must-link constraints as the flat arrays of the C ABI (connect.hpp:829-1045)
  std::vector<float> ml_crds;
  std::vector<int64_t> ml_sizes;
  std::vector<int> ml_dirs;
  if (pMustLinkConstraints) {
    for (size_t g = 0; g < pMustLinkConstraints->size(); g++) {
      ml_sizes.push_back((int64_t)(*pMustLinkConstraints)[g].size());
      for (size_t k = 0; k < (*pMustLinkConstraints)[g].size(); k++) {
        for (int d = 0; d < 3; d++) ml_crds.push_back((*pMustLinkConstraints)[g][k][d]);
        if (pMustLinkDirections) {
          const DirectionPairType how = (*pMustLinkDirections)[g][k];
          ml_dirs.push_back(how == SAME_DIRECTION ? 0 : (how == OPPOSITE_DIRECTION ? 1 : 2));
        }
      }
    }
  }
  const size_t n = (size_t)image_size[0] * image_size[1] * image_size[2];
  // directions: the standardized output array if one is given (it starts as a copy of the input), else a copy
  std::vector<float> dir_copy;
  float* dir = nullptr;
  if (aaaafVector) {
    const float* in = reinterpret_cast<const float*>(&aaaafVector[0][0][0]);
    if (aaaafVectorStandardized) {
      dir = reinterpret_cast<float*>(&aaaafVectorStandardized[0][0][0]);
      if (dir != in) for (size_t i = 0; i < 3 * n; i++) dir[i] = in[i];
    } else {
      dir_copy.assign(in, in + 3 * n);
      dir = dir_copy.data();
    }
  }
  std::vector<float> ten;
  if (aaaafSymmetricTensor) {
    ten.assign(6 * n, 0.0f);
    size_t v = 0;
    for (int iz = 0; iz < image_size[2]; iz++)
      for (int iy = 0; iy < image_size[1]; iy++)
        for (int ix = 0; ix < image_size[0]; ix++, v++)
          if (aaaafSymmetricTensor[iz][iy][ix])
            for (int c = 0; c < 6; c++) ten[6 * v + c] = aaaafSymmetricTensor[iz][iy][ix][c];
  }
  std::vector<float> cm(pv_cluster_maxima ? 3 * n : 0), cs(pv_cluster_sizes ? n : 0), csal(pv_cluster_saliencies ? n : 0);
  int64_t n_clusters = 0;
  hip_detail::check(visfd_hip_label_connected_ex(
      hip_detail::flat(aaafSaliency), reinterpret_cast<int64_t*>(&aaaiDest[0][0][0]), hip_detail::flat(aaafMask),
      image_size[0], image_size[1], image_size[2], threshold_saliency, dir, threshold_vector_saliency,
      threshold_vector_neighbor, consider_dot_product_sign ? 1 : 0, aaaafSymmetricTensor ? ten.data() : nullptr,
      threshold_tensor_saliency, threshold_tensor_neighbor, tensor_is_positive_definite_near_target ? 1 : 0, connectivity,
      (int64_t)label_undefined, sort_criteria == SORT_BY_SIZE ? 1 : 0, aaaafVectorStandardized ? 1 : 0,
      start_from_saliency_maxima ? 1 : 0, &n_clusters, cm.empty() ? nullptr : cm.data(), cs.empty() ? nullptr : cs.data(),
      csal.empty() ? nullptr : csal.data(), (int64_t)n, hip_detail::flat(aaafVoxelWeights),
      ml_crds.empty() ? nullptr : ml_crds.data(), ml_sizes.empty() ? nullptr : ml_sizes.data(), (int64_t)ml_sizes.size(),
      ml_dirs.empty() ? nullptr : ml_dirs.data()));
  if (pReportProgress) *pReportProgress << "Number of clusters found: " << n_clusters << "\n";
  if (pv_cluster_maxima) {
    pv_cluster_maxima->resize((size_t)n_clusters);
    for (int64_t k = 0; k < n_clusters; k++) (*pv_cluster_maxima)[(size_t)k] = {{cm[3 * k], cm[3 * k + 1], cm[3 * k + 2]}};
  }
  if (pv_cluster_sizes) pv_cluster_sizes->assign(cs.begin(), cs.begin() + n_clusters);
  if (pv_cluster_saliencies) pv_cluster_saliencies->assign(csal.begin(), csal.begin() + n_clusters);
  return (size_t)n_clusters;
}

// ---- eigenvalue order: lib/visfd/eigen3_simple.hpp:36-43 ------------------------------------------------
namespace selfadjoint_eigen3 {
typedef enum eEigenOrderType {
  INCREASING_EIVALS = VISFD_HIP_INCREASING_EIVALS,
  DECREASING_EIVALS = VISFD_HIP_DECREASING_EIVALS
} EigenOrderType;

// ---- DiagonalizeFlatSym3: lib/visfd/eigen3_simple.hpp:271-275 (one flat matrix, on the host) -------------
// source = {xx,yy,zz,xy,yz,xz}; dest = {lambda0,lambda1,lambda2, shoemake0..2}.  For whole volumes use the batch
// entry points of the C ABI (visfd_hip_diagonalize_flat_sym3 / visfd_hip_ridge_saliency) instead of a voxel loop.
inline void DiagonalizeFlatSym3(const float* source, float* dest, EigenOrderType eival_order = INCREASING_EIVALS) {
  hip_detail::check(visfd_hip_diagonalize_flat_sym3_host(source, dest, 1, (int)eival_order));
}

// ---- ConvertFlatSym2Evects3: lib/visfd/eigen3_simple.hpp:392-405 -----------------------------------------
inline void ConvertFlatSym2Evects3(const float m[6], float eivals[3], float eivects[3][3],
                                   EigenOrderType eival_order = INCREASING_EIVALS) {
  hip_detail::check(visfd_hip_convert_flat_sym2_evects3_host(m, (int)eival_order, eivals, &eivects[0][0]));
}
}  // namespace selfadjoint_eigen3

// ---- a11 scores of a diagonalised matrix: lib/visfd/feature.hpp:1526-1561, :1570-1581, :1591-1598, :1608-1612
template <typename TensorContainer, typename VectorContainer = const float*>
inline double ScoreHessianPlanar(TensorContainer diagonalizedHessian, VectorContainer = nullptr) {
  const double lambda1 = diagonalizedHessian[0], lambda2 = diagonalizedHessian[1];
  double N = lambda1 * lambda1 - lambda2 * lambda2;
  N *= N;
  return N;
}
template <typename TensorContainer, typename VectorContainer = const float*>
inline double ScoreHessianLinear(TensorContainer diagonalizedHessian, VectorContainer = nullptr) {
  const double lambda1 = diagonalizedHessian[0], lambda2 = diagonalizedHessian[1], lambda3 = diagonalizedHessian[2];
  return lambda1 * lambda2 - lambda3 * lambda3;
}
template <typename TensorContainer>
inline double ScoreTensorPlanar(const TensorContainer diagonalizedMatrix3) {
  const double lambda1 = diagonalizedMatrix3[0], lambda2 = diagonalizedMatrix3[1];
  return lambda1 - lambda2;
}
template <typename TensorContainer>
inline double ScoreTensorLinear(const TensorContainer diagonalizedMatrix3) {
  return ScoreHessianLinear(diagonalizedMatrix3, (const float*)nullptr);   // feature.hpp:1608-1612
}

// ---- CalcHessian: lib/visfd/feature.hpp:1203-1219 --------------------------------------------------------
// Containers as HandleTV instantiates them (bin/filter_mrc/handlers.cpp:1547-1565): gradient as
// array<float,3>***, Hessian as float**** with one pointer per voxel (nullptr where mask == 0).
inline void CalcHessian(int const image_size[3], float const* const* const* aaafSource,
                        std::array<float, 3>*** aaaafGradient, float**** aaaafHessian,
                        float const* const* const* aaafMask, float sigma, float truncate_ratio = 2.5,
                        std::ostream* pReportProgress = nullptr) {
  hip_detail::require_contiguous(aaafSource, image_size);
  hip_detail::require_contiguous(aaafMask, image_size);
  const size_t n = (size_t)image_size[0] * image_size[1] * image_size[2];
  std::vector<float> hess(aaaafHessian ? 6 * n : 0);
  float* grad = aaaafGradient ? &aaaafGradient[0][0][0][0] : nullptr;
  if (pReportProgress) *pReportProgress << "Calculating the Hessian associated with each voxel\n";
  hip_detail::check(visfd_hip_calc_hessian(hip_detail::context(), hip_detail::flat(aaafSource), grad,
                                           aaaafHessian ? hess.data() : nullptr, hip_detail::flat(aaafMask),
                                           image_size[0], image_size[1], image_size[2], sigma, truncate_ratio));
  if (aaaafHessian) {
    size_t v = 0;
    for (int iz = 0; iz < image_size[2]; iz++)
      for (int iy = 0; iy < image_size[1]; iy++)
        for (int ix = 0; ix < image_size[0]; ix++, v++)
          if (float* h = aaaafHessian[iz][iy][ix])
            for (int c = 0; c < 6; c++) h[c] = hess[6 * v + c];
  }
}

// ---- TV3D: lib/visfd/feature.hpp:1631-1724 (dense stick voting only) --------------------------------------
template <typename Scalar, typename Integer, typename VectorContainer, typename TensorContainer>
class TV3D {
  float sigma;
  Integer exponent;
  float cutoff;
 public:
  TV3D() : sigma(0), exponent(4), cutoff(2.5f) {}
  TV3D(Scalar set_sigma, Integer set_exponent, Scalar filter_cutoff_ratio = 2.5)
      : sigma(set_sigma), exponent(set_exponent), cutoff(filter_cutoff_ratio) {}
  void SetExponent(Scalar e) { exponent = (Integer)e; }
  void SetSigma(Scalar s, Scalar filter_cutoff_ratio = 2.5) { sigma = s; cutoff = filter_cutoff_ratio; }

  // aaaafV: array<float,3>*** ; aaaafDest: float**** (pointer per voxel, nullptr = no storage, e.g.
  // CompactMultiChannelImage3D::aaaafI).  The optional steps run on the host after the votes, AS THE REFERENCE
  // EXECUTES THEM (feature.hpp:1784-1901), oddities included:
  //   normalize (the reference's default; filter_mrc passes false, handlers.cpp:1832):
  //     nothing happens without a destination mask (the loops skip every voxel when aaafMaskDest is null, :1793, :1848);
  //     with a source mask every tensor entry is divided by the sum of the vote weights w * mask_src the voxel
  //     received, where that sum is positive; without one the divisor is (Dx*Dy)*Dz, D = GenFilterGauss1D(sigma, h)
  //     summed over the in-image taps (a different kernel than the votes), and because the loop runs over all nine
  //     (di, dj) the off-diagonal entries are divided TWICE (:1854-1858);
  //   diagonalize_dest: DiagonalizeHessianImage with DECREASING_EIVALS -- interior voxels 1..n-2 only (:1380-1386).
  // A null saliency array is refused: the reference then reads saliencies it never initialised (:1748-1756).
  void TVDenseStick(Integer const image_size[3], Scalar const* const* const* aaafSaliency,
                    VectorContainer const* const* const* aaaafV, TensorContainer*** aaaafDest,
                    Scalar const* const* const* aaafMaskSource = nullptr,
                    Scalar const* const* const* aaafMaskDest = nullptr, bool detect_curves_not_surfaces = false,
                    bool normalize = true, bool diagonalize_dest = false, std::ostream* pReportProgress = nullptr) {
    if (!aaafSaliency) throw VisfdErr("visfd_hip: TVDenseStick needs an explicit saliency array");
    int size[3] = {(int)image_size[0], (int)image_size[1], (int)image_size[2]};
    hip_detail::require_contiguous(aaafSaliency, size);
    hip_detail::require_contiguous(aaafMaskSource, size);
    hip_detail::require_contiguous(aaafMaskDest, size);
    const size_t n = (size_t)size[0] * size[1] * size[2];
    std::vector<float> ten(6 * n, 0.0f);
    const float* dir = reinterpret_cast<const float*>(&aaaafV[0][0][0]);
    if (pReportProgress) *pReportProgress << "---- Begin Tensor Voting (dense, stick) on the GPU ----" << std::endl;
    hip_detail::check(visfd_hip_tv_dense_stick(hip_detail::context(), hip_detail::flat(aaafSaliency), dir,
                                               ten.data(), hip_detail::flat(aaafMaskSource),
                                               hip_detail::flat(aaafMaskDest), size[0], size[1], size[2], sigma,
                                               (int)exponent, cutoff, detect_curves_not_surfaces ? 1 : 0));
    const float* mdst = hip_detail::flat(aaafMaskDest);
    if (normalize && mdst) {
      if (pReportProgress) *pReportProgress << "  Normalizing the result of tensor voting...";
      if (aaafMaskSource) {
        std::vector<float> den(n, 0.0f);
        hip_detail::check(visfd_hip_tv_weight_sum(hip_detail::context(), hip_detail::flat(aaafSaliency), den.data(),
                                                  hip_detail::flat(aaafMaskSource), mdst, size[0], size[1], size[2],
                                                  sigma, cutoff));
        for (size_t v = 0; v < n; v++)
          if (mdst[v] != 0.0f && den[v] > 0.0f)
            for (int c = 0; c < 6; c++) ten[6 * v + c] /= den[v];
      } else {
        int h = 0;
        hip_detail::check(visfd_hip_tv_tables(sigma, cutoff, &h, nullptr, nullptr));
        std::vector<float> taps(2 * (size_t)h + 1), D[3];
        hip_detail::check(visfd_hip_gauss_taps(sigma, h, taps.data()));
        for (int d = 0; d < 3; d++) {          // Filter1D::Apply on a line of ones (filter1d.hpp:47-104)
          D[d].resize((size_t)size[d]);
          for (int i = 0; i < size[d]; i++) {
            float a = 0.0f;
            for (int j = -h; j <= h; j++)
              if (i - j >= 0 && i - j < size[d]) a += taps[(size_t)(j + h)] * 1.0f;
            D[d][(size_t)i] = a;
          }
        }
        size_t v = 0;
        for (int iz = 0; iz < size[2]; iz++)
          for (int iy = 0; iy < size[1]; iy++)
            for (int ix = 0; ix < size[0]; ix++, v++) {
              if (mdst[v] == 0.0f) continue;
              const float denominator = (D[0][(size_t)ix] * D[1][(size_t)iy]) * D[2][(size_t)iz];
              float* T = &ten[6 * v];
              for (int c = 0; c < 3; c++) T[c] /= denominator;                            // (0,0), (1,1), (2,2)
              for (int c = 3; c < 6; c++) { T[c] /= denominator; T[c] /= denominator; }   // (di,dj) and (dj,di)
            }
      }
      if (pReportProgress) *pReportProgress << "done." << std::endl;
    }
    if (diagonalize_dest) {
      if (pReportProgress) *pReportProgress << "---- Diagonalizing Tensor Voting results ----" << std::endl;
      for (int iz = 1; iz < size[2] - 1; iz++)
        for (int iy = 1; iy < size[1] - 1; iy++)
          for (int ix = 1; ix < size[0] - 1; ix++) {
            const size_t v = ((size_t)iz * size[1] + iy) * size[0] + ix;
            if (mdst && mdst[v] == 0.0f) continue;
            float d6[6];
            hip_detail::check(visfd_hip_diagonalize_flat_sym3_host(&ten[6 * v], d6, 1, VISFD_HIP_DECREASING_EIVALS));
            for (int c = 0; c < 6; c++) ten[6 * v + c] = d6[c];
          }
    }
    size_t v = 0;
    for (int iz = 0; iz < size[2]; iz++)
      for (int iy = 0; iy < size[1]; iy++)
        for (int ix = 0; ix < size[0]; ix++, v++)
          if (aaaafDest[iz][iy][ix] && !(mdst && mdst[v] == 0.0f))
            for (int c = 0; c < 6; c++) aaaafDest[iz][iy][ix][c] = ten[6 * v + c];
  }
};

}  // namespace visfd

#endif  // VISFD_HIP_HPP
