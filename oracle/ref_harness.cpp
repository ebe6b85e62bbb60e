// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A thin extern "C" shim around the *real* VISFD templates, compiled from the
// headers where they lie under /root/reference/lib/visfd (see oracle/Makefile,
// target _ref/libvisfd_ref.so).  Nothing from the reference is copied here: this
// file only builds pointer tables over flat arrays and forwards to the library.
// It exists to (1) pin the CPU restatement in oracle/visfd_oracle.cpp,
// (2) generate the golden fixtures under tests/golden/ (tests/golden/make_golden.py),
// (3) optionally serve as the "reference" CPU baseline in bench.py.
//
// Build flags mirror the reference's setup_gcc.sh:7-10 (-O3 -DNDEBUG -fopenmp, no -march).
//
// All volumes are flat row-major [iz][iy][ix] float arrays (alloc3d.hpp:16-23 layout).

#include <cstdint>
#include <cstring>
#include <cmath>
#include <sstream>
#include <vector>
#include <array>
#include <limits>
#include <algorithm>
#include <iostream>
using namespace std;

#include <visfd.hpp>
using namespace visfd;

namespace {

// Pointer tables over a caller-owned contiguous block (the reference's own
// Alloc3D would allocate its own storage; we need views).
template <typename T>
struct View3 {
  std::vector<T**> zp;
  std::vector<T*> yp;
  T*** p = nullptr;
  View3(T* base, int nx, int ny, int nz) {
    if (!base) return;
    zp.resize(nz);
    yp.resize((size_t)nz * ny);
    for (int iz = 0; iz < nz; iz++) {
      zp[iz] = &yp[(size_t)iz * ny];
      for (int iy = 0; iy < ny; iy++)
        yp[(size_t)iz * ny + iy] = base + ((size_t)iz * ny + iy) * nx;
    }
    p = zp.data();
  }
};

typedef float const* const* const* cf3;

}  // namespace

extern "C" {

// filter1d.hpp:409-460
void vr_gauss_taps(float sigma, int halfwidth, float* taps_out) {
  Filter1D<float, int> f = GenFilterGauss1D(sigma, halfwidth);
  for (int i = -halfwidth; i <= halfwidth; i++) taps_out[i + halfwidth] = f.afH[i];
}

// filter3d_variants.hpp:513-518 : ratio = sqrt(-2 ln thr) evaluated in float
float vr_ratio_from_threshold(float thr) {
  float r = sqrt(-2 * log(thr));
  return r;
}

// filter3d.hpp:1086 (sigma[3], halfwidth[3])
float vr_apply_gauss_hw(const float* src, float* dst, const float* mask, int nx, int ny, int nz,
                        const float sigma[3], const int hw[3], int normalize) {
  int size[3] = {nx, ny, nz};
  View3<float> s(const_cast<float*>(src), nx, ny, nz), d(dst, nx, ny, nz),
      m(const_cast<float*>(mask), nx, ny, nz);
  return ApplyGauss<float>(size, (cf3)s.p, d.p, (cf3)m.p, sigma, hw, normalize != 0, nullptr);
}

// filter3d.hpp:1226 (sigma[3], truncate_ratio)
float vr_apply_gauss_ratio(const float* src, float* dst, const float* mask, int nx, int ny, int nz,
                           const float sigma[3], float ratio, int normalize) {
  int size[3] = {nx, ny, nz};
  View3<float> s(const_cast<float*>(src), nx, ny, nz), d(dst, nx, ny, nz),
      m(const_cast<float*>(mask), nx, ny, nz);
  return ApplyGauss<float>(size, (cf3)s.p, d.p, (cf3)m.p, sigma, ratio, normalize != 0, nullptr);
}

// filter3d.hpp:1338
void vr_apply_dog(const float* src, float* dst, const float* mask, int nx, int ny, int nz,
                  const float sigma_a[3], const float sigma_b[3], const int hw[3], float* pA,
                  float* pB) {
  int size[3] = {nx, ny, nz};
  View3<float> s(const_cast<float*>(src), nx, ny, nz), d(dst, nx, ny, nz),
      m(const_cast<float*>(mask), nx, ny, nz);
  ApplyDog<float>(size, (cf3)s.p, d.p, (cf3)m.p, sigma_a, sigma_b, hw, pA, pB, nullptr);
}

// filter3d.hpp:1428
void vr_apply_log(const float* src, float* dst, const float* mask, int nx, int ny, int nz,
                  const float sigma[3], float delta, float ratio, float* pA, float* pB) {
  int size[3] = {nx, ny, nz};
  View3<float> s(const_cast<float*>(src), nx, ny, nz), d(dst, nx, ny, nz),
      m(const_cast<float*>(mask), nx, ny, nz);
  ApplyLog<float>(size, (cf3)s.p, d.p, (cf3)m.p, sigma, delta, ratio, pA, pB, nullptr);
}

// feature.hpp:53 (BlobDog).  Rows of out_*: x,y,z,sigma,score.  Returns 0, or 1 if a
// capacity was too small (counts are still reported).
int vr_blob_dog(const float* src, const float* mask, int nx, int ny, int nz, const float* sigmas,
                int nsig, const float* aspect /*nullable*/, float delta, float ratio,
                float minima_threshold, float maxima_threshold, int use_ratios, float* out_min,
                int64_t cap_min, int64_t* n_min, float* out_max, int64_t cap_max,
                int64_t* n_max) {
  int size[3] = {nx, ny, nz};
  View3<float> s(const_cast<float*>(src), nx, ny, nz), m(const_cast<float*>(mask), nx, ny, nz);
  vector<float> sig(sigmas, sigmas + nsig);
  vector<array<float, 3>> cmin, cmax;
  vector<float> smin, smax, scmin, scmax;
  BlobDog<float>(size, (cf3)s.p, (cf3)m.p, sig, &cmin, &cmax, &smin, &smax, &scmin, &scmax,
                 aspect, delta, ratio, minima_threshold, maxima_threshold, use_ratios != 0,
                 nullptr, nullptr);
  *n_min = (int64_t)cmin.size();
  *n_max = (int64_t)cmax.size();
  int rc = 0;
  if ((int64_t)cmin.size() > cap_min || (int64_t)cmax.size() > cap_max) rc = 1;
  for (int64_t i = 0; i < (int64_t)cmin.size() && i < cap_min; i++) {
    out_min[5 * i + 0] = cmin[i][0]; out_min[5 * i + 1] = cmin[i][1]; out_min[5 * i + 2] = cmin[i][2];
    out_min[5 * i + 3] = smin[i];    out_min[5 * i + 4] = scmin[i];
  }
  for (int64_t i = 0; i < (int64_t)cmax.size() && i < cap_max; i++) {
    out_max[5 * i + 0] = cmax[i][0]; out_max[5 * i + 1] = cmax[i][1]; out_max[5 * i + 2] = cmax[i][2];
    out_max[5 * i + 3] = smax[i];    out_max[5 * i + 4] = scmax[i];
  }
  return rc;
}

// feature.hpp:446 diameter<->sigma conversion as the library evaluates it (:475, :504)
void vr_blob_diameters_to_sigmas(const float* diam, int n, float* sig) {
  for (int i = 0; i < n; i++) sig[i] = diam[i] / (2.0 * sqrt(3));
}
void vr_blob_sigmas_to_diameters(const float* sig, int n, float* diam) {
  for (int i = 0; i < n; i++) diam[i] = sig[i] * 2.0 * sqrt(3);
}

// feature.hpp:1203 (CalcHessian).  grad: 3 floats/voxel (nullable), hess: 6 floats/voxel
// in MapIndices order (lin3_utils.hpp:400-406).  Voxels with mask==0 are left untouched.
int vr_calc_hessian(const float* src, float* grad, float* hess, const float* mask, int nx, int ny,
                    int nz, float sigma, float ratio) {
  int size[3] = {nx, ny, nz};
  View3<float> s(const_cast<float*>(src), nx, ny, nz), m(const_cast<float*>(mask), nx, ny, nz);
  size_t n = (size_t)nx * ny * nz;
  std::vector<float*> hp(n);
  for (size_t i = 0; i < n; i++) hp[i] = hess + 6 * i;
  View3<float*> h(hp.data(), nx, ny, nz);
  View3<array<float, 3>> g(reinterpret_cast<array<float, 3>*>(grad), nx, ny, nz);
  try {
    CalcHessian<float, array<float, 3>, float*>(size, (cf3)s.p, g.p, h.p, (cf3)m.p, sigma, ratio,
                                               nullptr);
  } catch (VisfdErr& e) {
    return 1;
  }
  return 0;
}

// eigen3_simple.hpp:271 (DiagonalizeFlatSym3), batched over n voxels. order = EigenOrderType.
void vr_diagonalize_flat_sym3(const float* m6, float* out6, int64_t n, int order) {
  for (int64_t i = 0; i < n; i++)
    selfadjoint_eigen3::DiagonalizeFlatSym3(m6 + 6 * i, out6 + 6 * i,
                                            (selfadjoint_eigen3::EigenOrderType)order);
}

// eigen3_simple.hpp:392 (ConvertFlatSym2Evects3<float>), batched.
void vr_flat_sym_to_evects(const float* m6, float* eivals3, float* eivects9, int64_t n, int order) {
  for (int64_t i = 0; i < n; i++) {
    float ev[3];
    float evec[3][3];
    selfadjoint_eigen3::ConvertFlatSym2Evects3<float>(m6 + 6 * i, ev, evec,
                                                      (selfadjoint_eigen3::EigenOrderType)order);
    for (int d = 0; d < 3; d++) eivals3[3 * i + d] = ev[d];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) eivects9[9 * i + 3 * a + b] = evec[a][b];
  }
}

// The caller-side saliency/direction loop of HandleTV (handlers.cpp:1640-1746) built from the
// library calls it makes (ConvertFlatSym2Evects3 + ScoreHessianPlanar).  SURFACE_RIDGE only,
// no background subtraction (peak_height = 1).  saliency is zeroed first (handlers.cpp:1640-1643);
// dir (3 floats/voxel) is only written where mask != 0.
void vr_hessian_saliency(const float* hess, const float* mask, int64_t n, int order,
                         float* saliency, float* dir) {
  for (int64_t i = 0; i < n; i++) saliency[i] = 0.0;
  #pragma omp parallel for
  for (int64_t i = 0; i < n; i++) {
    if (mask && mask[i] == 0.0) continue;
    float eivals[3];
    float eivects[3][3];
    selfadjoint_eigen3::ConvertFlatSym2Evects3<float>(hess + 6 * i, eivals, eivects,
                                                      (selfadjoint_eigen3::EigenOrderType)order);
    float score;
    score = ScoreHessianPlanar(eivals, (float*)nullptr);
    float peak_height = 1.0;
    score *= peak_height;
    saliency[i] = score;
    dir[3 * i + 0] = eivects[0][0];
    dir[3 * i + 1] = eivects[0][1];
    dir[3 * i + 2] = eivects[0][2];
  }
}

// feature.hpp:1711 (TV3D::TVDenseStick) as HandleTV instantiates it
// (TV3D<float,int,array<float,3>,float*>, handlers.cpp:1821-1836).
// tensor: 6 floats/voxel; voxels whose mask_dst==0 have no storage in the reference's compact
// container (null pointer) and are left untouched here.
void vr_tv_dense_stick(const float* saliency, const float* dir, float* tensor,
                       const float* mask_src, const float* mask_dst, int nx, int ny, int nz,
                       float sigma_tv, int exponent, float cutoff_ratio, int curves,
                       int normalize, int diagonalize) {
  int size[3] = {nx, ny, nz};
  size_t n = (size_t)nx * ny * nz;
  View3<float> s(const_cast<float*>(saliency), nx, ny, nz),
      ms(const_cast<float*>(mask_src), nx, ny, nz), md(const_cast<float*>(mask_dst), nx, ny, nz);
  View3<array<float, 3>> v(reinterpret_cast<array<float, 3>*>(const_cast<float*>(dir)), nx, ny, nz);
  std::vector<float*> tp(n);
  for (size_t i = 0; i < n; i++)
    tp[i] = (mask_dst && mask_dst[i] == 0.0) ? nullptr : tensor + 6 * i;
  View3<float*> t(tp.data(), nx, ny, nz);
  TV3D<float, int, array<float, 3>, float*> tv(sigma_tv, exponent, cutoff_ratio);
  tv.TVDenseStick(size, (cf3)s.p, (array<float, 3> const* const* const*)v.p, t.p, (cf3)ms.p,
                  (cf3)md.p, curves != 0, normalize != 0, diagonalize != 0, nullptr);
}

// filter3d.hpp:546 radial table as TV3D::Resize builds it (feature.hpp:2419-2428), plus the
// displacement table (feature.hpp:2468-2482).  w: (2h+1)^3 floats, rhat: 3*(2h+1)^3 floats.
int vr_tv_tables(float sigma_tv, float cutoff_ratio, float* w, float* rhat, int cap_h) {
  int h = floor(sigma_tv * cutoff_ratio);
  if (h > cap_h) return -h;
  if (!w) return h;
  int hw[3] = {h, h, h};
  float sig[3] = {sigma_tv, sigma_tv, sigma_tv};
  Filter3D<float, int> f = GenFilterGenGauss3D(sig, static_cast<float>(2.0), hw);
  int n = 2 * h + 1;
  for (int iz = -h; iz <= h; iz++)
    for (int iy = -h; iy <= h; iy++)
      for (int ix = -h; ix <= h; ix++) {
        size_t k = ((size_t)(iz + h) * n + (iy + h)) * n + (ix + h);
        w[k] = f.aaafH[iz][iy][ix];
        float length = sqrt(ix * ix + iy * iy + iz * iz);
        if (length == 0) length = 1.0;
        rhat[3 * k + 0] = ix / length;
        rhat[3 * k + 1] = iy / length;
        rhat[3 * k + 2] = iz / length;
      }
  return h;
}

// Post-TV score loop of HandleTV (handlers.cpp:1870-1892), SURFACE (ScoreTensorPlanar).
void vr_tensor_saliency(const float* tensor, const float* mask, int64_t n, int order,
                        float* saliency_inout) {
  for (int64_t i = 0; i < n; i++) {
    if (mask && mask[i] == 0.0) continue;
    float diag[6];
    selfadjoint_eigen3::DiagonalizeFlatSym3(tensor + 6 * i, diag,
                                            (selfadjoint_eigen3::EigenOrderType)order);
    float score;
    score = ScoreTensorPlanar(diag);
    float peak_height = 1.0;
    score *= peak_height;
    saliency_inout[i] = score;
  }
}

// Global top-fraction threshold of HandleTV (handlers.cpp:1751-1797), same statements.
float vr_threshold_fraction(float* saliency, const float* mask, int64_t n, float fraction) {
  size_t n_voxels = 0;
  for (int64_t i = 0; i < n; i++) {
    if (mask && (mask[i] == 0)) continue;
    n_voxels++;
  }
  vector<float> saliencies(n_voxels);
  size_t k = 0;
  for (int64_t i = 0; i < n; i++) {
    if (mask && (mask[i] == 0)) continue;
    saliencies[k] = saliency[i];
    k++;
  }
  sort(saliencies.rbegin(), saliencies.rend());
  k = floor(n_voxels * fraction);
  float thr = saliencies[k];
  for (int64_t i = 0; i < n; i++)
    if (saliency[i] < thr) saliency[i] = 0.0;
  return thr;
}

// ---- blob list post-processing: the reference's own templates ---------------------
namespace {
void lists_from_flat(const float* crds, const float* diam, const float* score, int64_t n,
                     std::vector<std::array<float, 3> >& c, std::vector<float>& d, std::vector<float>& s) {
  c.resize(n); d.assign(diam, diam + n); s.assign(score, score + n);
  for (int64_t i = 0; i < n; i++) c[i] = {crds[3 * i], crds[3 * i + 1], crds[3 * i + 2]};
}
int64_t lists_to_flat(const std::vector<std::array<float, 3> >& c, const std::vector<float>& d,
                      const std::vector<float>& s, float* crds, float* diam, float* score) {
  for (size_t i = 0; i < c.size(); i++) {
    crds[3 * i] = c[i][0]; crds[3 * i + 1] = c[i][1]; crds[3 * i + 2] = c[i][2];
    diam[i] = d[i]; score[i] = s[i];
  }
  return (int64_t)c.size();
}
}  // namespace

// visfd_utils.hpp:95-118
float vr_sphere_overlap(float rij, float ri, float rj) { return CalcSphereOverlap(rij, ri, rj); }

// feature.hpp:573-616 (criteria overload); permutation (nullable) receives the new order
void vr_sort_blobs(float* crds, float* diam, float* score, int64_t n, int criteria, int ascending,
                   uint64_t* permutation) {
  std::vector<std::array<float, 3> > c; std::vector<float> d, s;
  lists_from_flat(crds, diam, score, n, c, d, s);
  std::vector<size_t> perm;
  SortBlobs(c, d, s, (SortCriteria)criteria, ascending != 0, &perm, (std::ostream*)nullptr);
  lists_to_flat(c, d, s, crds, diam, score);
  if (permutation) for (size_t i = 0; i < perm.size(); i++) permutation[i] = perm[i];
}

// feature.hpp:924-969
int64_t vr_discard_masked_blobs(float* crds, float* diam, float* score, int64_t n, const float* mask,
                                int nx, int ny, int nz) {
  std::vector<std::array<float, 3> > c; std::vector<float> d, s;
  lists_from_flat(crds, diam, score, n, c, d, s);
  View3<const float> vm(mask, nx, ny, nz);
  DiscardMaskedBlobs(c, d, s, (cf3)vm.p, (std::ostream*)nullptr);
  return lists_to_flat(c, d, s, crds, diam, score);
}

// feature.hpp:720-913
int64_t vr_discard_overlapping_blobs(float* crds, float* diam, float* score, int64_t n, float min_sep,
                                     float max_large, float max_small, int criteria, int scale) {
  std::vector<std::array<float, 3> > c; std::vector<float> d, s;
  lists_from_flat(crds, diam, score, n, c, d, s);
  DiscardOverlappingBlobs(c, d, s, min_sep, max_large, max_small, (SortCriteria)criteria,
                          (std::ostream*)nullptr, scale);
  return lists_to_flat(c, d, s, crds, diam, score);
}

// ---- binning: resample.hpp:53-166 --------------------------------------------------------------
int vr_bin_array3d(const float* src, const int* ssz, float* dst, const int* dsz, const int* offset) {
  View3<const float> vs(src, ssz[0], ssz[1], ssz[2]);
  View3<float> vd(dst, dsz[0], dsz[1], dsz[2]);
  try { BinArray3D(ssz, dsz, (cf3)vs.p, vd.p, offset); } catch (VisfdErr&) { return 1; }
  return 0;
}
int vr_unbin_array3d(const float* src, const int* ssz, float* dst, const int* dsz, const int* offset) {
  View3<const float> vs(src, ssz[0], ssz[1], ssz[2]);
  View3<float> vd(dst, dsz[0], dsz[1], dsz[2]);
  try { UnbinArray3D(ssz, dsz, (cf3)vs.p, vd.p, offset); } catch (VisfdErr&) { return 1; }
  return 0;
}

// ---- LabelConnected: connect.hpp:168-1427, called as bin/filter_mrc/handlers.cpp:1985-2013 does --------
// (Label = ptrdiff_t, Coordinate = float, per-voxel pointers into the caller's AoS arrays; the
// standardized-direction argument aliases the direction argument as in the handler.)
int64_t vr_label_connected_ex(const float* saliency, int64_t* labels, const float* mask, int nx, int ny, int nz,
                              float thr_saliency, float* direction, float thr_vs, float thr_vn, int consider_sign,
                              float* tensor, float thr_ts, float thr_tn, int tensor_posdef, int connectivity,
                              int64_t label_undefined, int sort_by_size, int standardize, int from_maxima,
                              float* cluster_maxima, float* cluster_sizes, float* cluster_saliencies, int64_t capacity,
                              const float* voxel_weights, const float* ml_crds, const int64_t* ml_group_sizes,
                              int64_t ml_ngroups, const int* ml_directions);

int64_t vr_label_connected(const float* saliency, int64_t* labels, const float* mask, int nx, int ny, int nz,
                           float thr_saliency, float* direction, float thr_vs, float thr_vn, int consider_sign,
                           float* tensor, float thr_ts, float thr_tn, int tensor_posdef, int connectivity,
                           int64_t label_undefined, int sort_by_size, int standardize, int from_maxima,
                           float* cluster_maxima, float* cluster_sizes, float* cluster_saliencies, int64_t capacity) {
  return vr_label_connected_ex(saliency, labels, mask, nx, ny, nz, thr_saliency, direction, thr_vs, thr_vn, consider_sign,
                               tensor, thr_ts, thr_tn, tensor_posdef, connectivity, label_undefined, sort_by_size, standardize,
                               from_maxima, cluster_maxima, cluster_sizes, cluster_saliencies, capacity, nullptr, nullptr,
                               nullptr, 0, nullptr);
}

int64_t vr_label_connected_ex(const float* saliency, int64_t* labels, const float* mask, int nx, int ny, int nz,
                              float thr_saliency, float* direction, float thr_vs, float thr_vn, int consider_sign,
                              float* tensor, float thr_ts, float thr_tn, int tensor_posdef, int connectivity,
                              int64_t label_undefined, int sort_by_size, int standardize, int from_maxima,
                              float* cluster_maxima, float* cluster_sizes, float* cluster_saliencies, int64_t capacity,
                              const float* voxel_weights, const float* ml_crds, const int64_t* ml_group_sizes,
                              int64_t ml_ngroups, const int* ml_directions) {
  const int size[3] = {nx, ny, nz};
  View3<const float> vw(voxel_weights, nx, ny, nz);
  std::vector<std::vector<std::array<float, 3> > > ml;
  std::vector<std::vector<DirectionPairType> > mld;
  {
    int64_t at = 0;
    for (int64_t gi = 0; gi < ml_ngroups; gi++) {
      ml.push_back(std::vector<std::array<float, 3> >());
      mld.push_back(std::vector<DirectionPairType>());
      for (int64_t k = 0; k < ml_group_sizes[gi]; k++, at++) {
        std::array<float, 3> c = {{ml_crds[3 * at], ml_crds[3 * at + 1], ml_crds[3 * at + 2]}};
        ml.back().push_back(c);
        const int how = ml_directions ? ml_directions[at] : 2;
        mld.back().push_back(how == 0 ? SAME_DIRECTION : (how == 1 ? OPPOSITE_DIRECTION : AUTO));
      }
    }
  }
  std::ostringstream progress;   // the must-link code dereferences its progress stream unconditionally (connect.hpp:865)
  View3<const float> vs(saliency, nx, ny, nz), vm(mask, nx, ny, nz);
  static_assert(sizeof(ptrdiff_t) == sizeof(int64_t), "labels are ptrdiff_t");
  View3<ptrdiff_t> vl(reinterpret_cast<ptrdiff_t*>(labels), nx, ny, nz);
  // per-voxel pointer tables float* [nz][ny][nx]
  const size_t n = (size_t)nx * ny * nz;
  std::vector<float*> pd(direction ? n : 0), pt(tensor ? n : 0);
  for (size_t i = 0; i < pd.size(); i++) pd[i] = direction + 3 * i;
  for (size_t i = 0; i < pt.size(); i++) pt[i] = tensor + 6 * i;
  View3<float*> vd(direction ? pd.data() : nullptr, nx, ny, nz), vt(tensor ? pt.data() : nullptr, nx, ny, nz);
  std::vector<std::array<float, 3> > centers;
  std::vector<float> sizes, sals;
  size_t nc = LabelConnected(size, (cf3)vs.p, vl.p, (cf3)vm.p, thr_saliency,
                             (float* const* const* const*)vd.p, thr_vs, thr_vn, consider_sign != 0,
                             (float* const* const* const*)vt.p, thr_ts, thr_tn, tensor_posdef != 0, connectivity,
                             (ptrdiff_t)label_undefined, &centers, &sizes, &sals,
                             sort_by_size ? RegionSortCriteria::SORT_BY_SIZE : RegionSortCriteria::SORT_BY_VALUE,
                             (cf3)vw.p, standardize ? vd.p : static_cast<float****>(nullptr),
                             ml_ngroups > 0 ? &ml : static_cast<const std::vector<std::vector<std::array<float, 3> > >*>(nullptr),
                             (ml_ngroups > 0 && ml_directions) ? &mld : static_cast<const std::vector<std::vector<DirectionPairType> >*>(nullptr),
                             from_maxima != 0, ml_ngroups > 0 ? (std::ostream*)&progress : (std::ostream*)nullptr);
  for (size_t k = 0; k < nc && (int64_t)k < capacity; k++) {
    if (cluster_maxima) { cluster_maxima[3 * k] = centers[k][0]; cluster_maxima[3 * k + 1] = centers[k][1]; cluster_maxima[3 * k + 2] = centers[k][2]; }
    if (cluster_sizes) cluster_sizes[k] = sizes[k];
    if (cluster_saliencies) cluster_saliencies[k] = sals[k];
  }
  return (int64_t)nc;
}

// lin3_utils.hpp:502-529 as compiled (see visfd_amd/csrc/connect.cpp)
float vr_trace_product_sym3(const float* a, const float* b) { return TraceProductSym3(a, b); }

// eigen3_simple.hpp:137-266 instantiated for float (the form bin/filter_mrc/handlers.cpp:2249-2252 uses)
void vr_diagonalize_sym3_f32(const float* m9, int order, float* eivals3, float* eivects9) {
  float M[3][3], E[3][3];
  for (int i = 0; i < 9; i++) M[i / 3][i % 3] = m9[i];
  selfadjoint_eigen3::DiagonalizeSym3(M, eivals3, E, (selfadjoint_eigen3::EigenOrderType)order);
  for (int i = 0; i < 9; i++) eivects9[i] = E[i / 3][i % 3];
}

// eigen3_simple.hpp:392-405 instantiated for float: eigenvalues + eigenvectors (rows) of a flat symmetric matrix
void vr_convert_flat_sym2_evects3(const float* m6, int order, float* eivals3, float* eivects9) {
  float E[3][3];
  selfadjoint_eigen3::ConvertFlatSym2Evects3(m6, eivals3, E, (selfadjoint_eigen3::EigenOrderType)order);
  for (int i = 0; i < 9; i++) eivects9[i] = E[i / 3][i % 3];
}

// filter3d.hpp:1698-1853
void vr_local_fluctuations(const float* src, float* dst, const float* mask, int nx, int ny, int nz,
                           const float sigma[3], float exponent, float ratio, int normalize) {
  int size[3] = {nx, ny, nz};
  View3<float> s(const_cast<float*>(src), nx, ny, nz), d(dst, nx, ny, nz),
      m(const_cast<float*>(mask), nx, ny, nz);
  float sg[3] = {sigma[0], sigma[1], sigma[2]};
  LocalFluctuations<float>(size, (cf3)s.p, d.p, (cf3)m.p, sg, exponent, ratio, normalize != 0, nullptr);
}

int vr_version() { return 7; }

}  // extern "C"
