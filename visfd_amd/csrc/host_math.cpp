// host_math.cpp -- the small host-side arithmetic of the hot path: filter taps and the
// tensor-voting lookup tables.  These are O(filter width) computations the reference also does
// on the CPU; their float/long-double evaluation order is part of the
// parity contract, so this file must be built without FMA contraction or fast-math.
#include <cmath>
#include <limits>
#include <vector>

#include "common.hpp"

namespace vh {

// Discrete-Gaussian taps: reference lib/visfd/filter1d.hpp:409-460.
//   sigma<=10 and |i|<=20 : exp(-s^2) * I_|i|(s^2)  (modified Bessel function, long double)
//   otherwise             : exp(-i^2/(2 s^2)) / sqrt(2 pi s^2)
//   sigma==0              : Kronecker delta
// Each tap is rounded to float before the long-double normalising sum is formed.
void host_gauss_taps(float sigma, int h, float* t) {
  long double norm = 0.0L;
  for (int i = -h; i <= h; i++) {
    float v;
    if (sigma == 0.0f) {
      v = (i == 0) ? 1.0f : 0.0f;
    } else {
      const long double s = sigma;
      const long double x = i;
      const long double ax = std::abs(x);
      if (s <= 10.0 && ax <= 20.0)
        v = (float)(std::exp(-s * s) * std::cyl_bessel_i(ax, s * s));
      else
        v = (float)(std::exp(-(x * x) / (2.0 * s * s)) / std::sqrt(2 * s * s * M_PI));
    }
    t[i + h] = v;
    norm += v;
  }
  for (int k = 0; k < 2 * h + 1; k++) t[k] = (float)(t[k] / norm);
}

// The separable filter's boundary normaliser: the axis filter applied to a line of n ones with
// zero extension (reference lib/visfd/filter3d.hpp:1004-1012 via filter1d.hpp:47-104).
void host_conv_ones(i64 n, const float* t, int h, float* out) {
  for (i64 i = 0; i < n; i++) {
    float acc = 0.0f;
    for (int j = -h; j <= h; j++) {
      i64 k = i - j;
      if (k < 0 || k >= n) continue;
      acc += t[j + h] * 1.0f;
    }
    out[i] = acc;
  }
}

// Tensor-voting window: reference lib/visfd/feature.hpp:1669-1675.
int host_tv_halfwidth(float sigma, float cutoff) { return (int)std::floor(sigma * cutoff); }

// Radial weight exp(-(r/sigma)^2) with spherical support, normalised to unit sum
// (reference lib/visfd/filter3d.hpp:546-601 with m_exp=2, as TV3D::Resize calls it,
// feature.hpp:2419-2428) and unit displacement vectors (feature.hpp:2468-2482).
void host_tv_tables(float sigma, int h, float* w, float* rhat) {
  float cut = 1.0f;
  if (sigma > 0) {
    float e = std::exp(-std::pow(h / sigma, 2.0f));
    if (e < cut) cut = e;
  }
  const int n = 2 * h + 1;
  float total = 0;
  for (int iz = -h; iz <= h; iz++)
    for (int iy = -h; iy <= h; iy++)
      for (int ix = -h; ix <= h; ix++) {
        const float x = (sigma == 0.0f && ix == 0) ? 0.0f : ix / sigma;
        const float y = (sigma == 0.0f && iy == 0) ? 0.0f : iy / sigma;
        const float z = (sigma == 0.0f && iz == 0) ? 0.0f : iz / sigma;
        const float r = std::sqrt(x * x + y * y + z * z);
        float v = (r > 0) ? std::exp(-std::pow(r, 2.0f)) : 1.0f;
        if (std::fabs(v) < cut) v = 0.0f;
        const size_t k = ((size_t)(iz + h) * n + (iy + h)) * n + (ix + h);
        w[k] = v;
        total += v;
        if (rhat) {
          float len = (float)std::sqrt((double)(ix * ix + iy * iy + iz * iz));
          if (len == 0) len = 1.0f;
          rhat[3 * k + 0] = ix / len;
          rhat[3 * k + 1] = iy / len;
          rhat[3 * k + 2] = iz / len;
        }
      }
  const size_t m = (size_t)n * n * n;
  for (size_t k = 0; k < m; k++) w[k] /= total;
}

// Central value A of the normalised generalised-Gaussian window GenFilterGenGauss3D(width, m, ratio)
// (reference lib/visfd/filter3d.hpp:546-640): window half-widths floor(width*ratio), entries exp(-r^m) with
// r = sqrt((x/wx)^2+(y/wy)^2+(z/wz)^2), entries below the smallest face value zeroed, divided by their float sum
// accumulated in z,y,x order.  LocalFluctuations multiplies its variance by this number (filter3d.hpp:1725,1836).
float host_gengauss3d_peak(const float width[3], float m_exp, float ratio) {
  int hw[3];
  for (int d = 0; d < 3; d++) hw[d] = (int)std::floor(width[d] * ratio);
  float cut = 1.0f;
  for (int d = 0; d < 3; d++) {
    const float e = (width[d] > 0) ? std::exp(-std::pow(hw[d] / width[d], m_exp)) : 1.0f;
    if (e < cut) cut = e;
  }
  float total = 0;
  for (int iz = -hw[2]; iz <= hw[2]; iz++)
    for (int iy = -hw[1]; iy <= hw[1]; iy++)
      for (int ix = -hw[0]; ix <= hw[0]; ix++) {
        const float x = (width[0] == 0.0f && ix == 0) ? 0.0f : ix / width[0];
        const float y = (width[1] == 0.0f && iy == 0) ? 0.0f : iy / width[1];
        const float z = (width[2] == 0.0f && iz == 0) ? 0.0f : iz / width[2];
        const float r = std::sqrt(x * x + y * y + z * z);
        float v = (r > 0) ? std::exp(-std::pow(r, m_exp)) : 1.0f;
        if (std::fabs(v) < cut) v = 0.0f;
        total += v;
      }
  return 1.0f / total;   // the centre entry is 1 before the division by the sum
}

}  // namespace vh
