// oracle/visfd_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A from-scratch CPU restatement of the arithmetic contract of VISFD's dense
// 3-D filtering hot path.  It is the *checker* for the HIP
// kernels: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library; the product (visfd_amd/csrc) never links or calls it.
//
// Parity status: PINNED.  Every function below is checked bit-for-bit (floats) against
// the real reference templates compiled from /root/reference (oracle/_ref/libvisfd_ref.so,
// built by oracle/Makefile from oracle/ref_harness.cpp) by tests/test_oracle_vs_ref.py, and
// against committed golden vectors generated from that build (tests/golden/, script
// tests/golden/make_golden.py) by tests/test_oracle_golden.py.
//
// Build: g++ -std=gnu++17 -O2 -ffp-contract=off -fopenmp (no -march, no -ffast-math): the
// reference is built without FMA or reassociation (setup_gcc.sh:7-10), and float
// evaluation order below is part of the contract.
//
// Layout: every volume is a flat row-major [iz][iy][ix] array, x fastest
// (reference: lib/visfd/alloc3d.hpp:16-23,57-61).  Indices are 64-bit here.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

typedef int64_t i64;

inline i64 vox(i64 ix, i64 iy, i64 iz, i64 nx, i64 ny) { return (iz * ny + iy) * nx + ix; }

// ---------------------------------------------------------------------------------------
// A.1  Gaussian taps.  Follows lib/visfd/filter1d.hpp:409-460.
//   |i|<=20 and sigma<=10: discrete Gaussian exp(-s^2) I_|i|(s^2) in long double;
//   otherwise the sampled continuous Gaussian; each tap is stored to float BEFORE the
//   long-double sum is formed; every float is then divided by that long-double sum.
// ---------------------------------------------------------------------------------------
void gauss_taps(float sigma, int h, float* t /* 2h+1, t[h] = centre */) {
  long double total = 0.0L;
  for (int i = -h; i <= h; i++) {
    float v;
    if (sigma == 0.0f) {
      v = (i == 0) ? 1.0f : 0.0f;
    } else {
      long double S = sigma;
      long double I = i;
      if ((S <= 10.0) && (std::abs(I) <= 20.0)) {
        long double b = std::exp(-S * S) * std::cyl_bessel_i(std::abs(I), S * S);
        v = (float)b;
      } else {
        long double g = std::exp(-(I * I) / (2.0 * S * S)) / std::sqrt(2 * S * S * M_PI);
        v = (float)g;
      }
    }
    t[i + h] = v;
    total += v;
  }
  for (int i = 0; i < 2 * h + 1; i++) t[i] = (float)(t[i] / total);
}

// ---------------------------------------------------------------------------------------
// A.2  One 1-D line.  Follows lib/visfd/filter1d.hpp:47-104 (plain) and :204-295 (masked).
//   g[i] = sum_{j=-h..h, 0<=i-j<n} t[j]*f[i-j], j ascending, float accumulate from 0.0f.
//   Output is forced to exactly 0.0 when no non-zero source (plain form) / mask (masked
//   form) sample lies in [i-h, i+h] (the reference's "sparse input" scan).
//   `stride` lets the caller walk a Z or Y line of the volume in place.
// ---------------------------------------------------------------------------------------
void conv_line_plain(i64 n, const float* f, i64 fs, float* g, i64 gs, const float* t, int h) {
  // since = number of samples visited since the last non-zero one, looking h ahead.
  i64 width = 2 * (i64)h + 1;
  i64 since = width;
  i64 look = std::min<i64>(h, n);
  for (i64 I = 0; I < look; I++) since = (f[I * fs] != 0.0f) ? 0 : since + 1;
  for (i64 i = 0; i < n; i++) {
    i64 I = i + h;
    since = ((I < n) && (f[I * fs] != 0.0f)) ? 0 : since + 1;
    if (since >= width) {
      g[i * gs] = 0.0f;
      continue;
    }
    float acc = 0.0f;
    for (int j = -h; j <= h; j++) {
      i64 k = i - j;
      if (k < 0 || k >= n) continue;
      acc += t[j + h] * f[k * fs];
    }
    g[i * gs] = acc;
  }
}

void conv_line_masked(i64 n, const float* f, i64 fs, float* g, i64 gs, const float* m /*nullable*/,
                      i64 ms, float* den /*nullable*/, i64 ds, const float* t, int h) {
  i64 width = 2 * (i64)h + 1;
  i64 since = width;
  i64 look = std::min<i64>(h, n);
  for (i64 I = 0; I < look; I++) since = (!m || m[I * ms] != 0.0f) ? 0 : since + 1;
  for (i64 i = 0; i < n; i++) {
    i64 I = i + h;
    since = ((I < n) && (!m || m[I * ms] != 0.0f)) ? 0 : since + 1;
    if (since >= width) {
      g[i * gs] = 0.0f;
      if (den) den[i * ds] = 0.0f;
      continue;
    }
    float acc = 0.0f, dacc = 0.0f;
    for (int j = -h; j <= h; j++) {
      i64 k = i - j;
      if (k < 0 || k >= n) continue;
      float w = t[j + h];
      if (m) w *= m[k * ms];
      float term = w * f[k * fs];
      acc += term;
      if (den) dacc += w;
    }
    if (den) den[i * ds] = dacc;
    g[i * gs] = acc;
  }
}

// ---------------------------------------------------------------------------------------
// A.3  Separable 3-D filter.  Follows lib/visfd/filter3d.hpp:686-1050.
//   Z lines (masked form, producing the denominator when normalize && mask), then Y lines,
//   then X lines (plain form on the image and on the denominator); epilogue:
//   masked: dst /= den where den > 0;  unmasked: dst /= (Dx[ix]*Dy[iy])*Dz[iz] with D* the
//   axis filter applied to a line of ones.  Returns tx[0]*ty[0]*tz[0] (centre taps).
// ---------------------------------------------------------------------------------------
float separable3d(const float* src, float* dst, const float* mask, i64 nx, i64 ny, i64 nz,
                  const float* tx, int hx, const float* ty, int hy, const float* tz, int hz,
                  bool normalize) {
  i64 n = nx * ny * nz;
  std::vector<float> a(src, src + n), b(n);
  std::vector<float> dena, denb;
  bool masked_norm = normalize && mask;
  if (masked_norm) {
    dena.assign(n, 1.0f);
    denb.assign(n, 1.0f);
  }
  // Z pass: a -> b
  #pragma omp parallel for collapse(2)
  for (i64 iy = 0; iy < ny; iy++)
    for (i64 ix = 0; ix < nx; ix++) {
      i64 o = vox(ix, iy, 0, nx, ny), s = nx * ny;
      conv_line_masked(nz, &a[o], s, &b[o], s, mask ? mask + o : nullptr, s,
                       masked_norm ? &denb[o] : nullptr, s, tz, hz);
    }
  a.swap(b);
  if (masked_norm) dena.swap(denb);
  // Y pass
  #pragma omp parallel for collapse(2)
  for (i64 iz = 0; iz < nz; iz++)
    for (i64 ix = 0; ix < nx; ix++) {
      i64 o = vox(ix, 0, iz, nx, ny), s = nx;
      conv_line_plain(ny, &a[o], s, &b[o], s, ty, hy);
      if (masked_norm) conv_line_plain(ny, &dena[o], s, &denb[o], s, ty, hy);
    }
  a.swap(b);
  if (masked_norm) dena.swap(denb);
  // X pass
  #pragma omp parallel for collapse(2)
  for (i64 iz = 0; iz < nz; iz++)
    for (i64 iy = 0; iy < ny; iy++) {
      i64 o = vox(0, iy, iz, nx, ny);
      conv_line_plain(nx, &a[o], 1, &b[o], 1, tx, hx);
      if (masked_norm) conv_line_plain(nx, &dena[o], 1, &denb[o], 1, tx, hx);
    }
  a.swap(b);
  if (masked_norm) dena.swap(denb);

  if (normalize) {
    if (mask) {
      #pragma omp parallel for
      for (i64 i = 0; i < n; i++)
        if (dena[i] > 0.0f) a[i] /= dena[i];
    } else {
      std::vector<float> ones, D[3];
      const float* taps[3] = {tx, ty, tz};
      int hh[3] = {hx, hy, hz};
      i64 len[3] = {nx, ny, nz};
      for (int d = 0; d < 3; d++) {
        ones.assign(len[d], 1.0f);
        D[d].resize(len[d]);
        conv_line_plain(len[d], ones.data(), 1, D[d].data(), 1, taps[d], hh[d]);
      }
      #pragma omp parallel for collapse(2)
      for (i64 iz = 0; iz < nz; iz++)
        for (i64 iy = 0; iy < ny; iy++)
          for (i64 ix = 0; ix < nx; ix++) {
            float den = (D[0][ix] * D[1][iy]) * D[2][iz];
            a[vox(ix, iy, iz, nx, ny)] /= den;
          }
    }
  }
  std::memcpy(dst, a.data(), sizeof(float) * n);
  return (tx[hx] * ty[hy]) * tz[hz];
}

float gauss_hw(const float* src, float* dst, const float* mask, i64 nx, i64 ny, i64 nz,
               const float sigma[3], const int hw[3], bool normalize) {
  // lib/visfd/filter3d.hpp:1086-1124
  std::vector<float> t[3];
  for (int d = 0; d < 3; d++) {
    t[d].resize(2 * hw[d] + 1);
    gauss_taps(sigma[d], hw[d], t[d].data());
  }
  return separable3d(src, dst, mask, nx, ny, nz, t[0].data(), hw[0], t[1].data(), hw[1],
                     t[2].data(), hw[2], normalize);
}

// ---------------------------------------------------------------------------------------
// A.7  3x3 symmetric eigen-decomposition in double (lib/visfd/eigen3_simple.hpp:47-342) and
// the float Shoemake pack/unpack of the eigenvector frame (lib/visfd/lin3_utils.hpp:230-394).
// Written from the mathematics (trigonometric solution of the characteristic cubic;
// eigenvectors as null-space vectors obtained from column cross products).
// ---------------------------------------------------------------------------------------
enum { ORDER_INCREASING = 0, ORDER_DECREASING = 1 };

struct V3 { double v[3]; };
inline V3 cross(const V3& a, const V3& b) {
  V3 c;
  c.v[2] = a.v[0] * b.v[1] - a.v[1] * b.v[0];
  c.v[0] = a.v[1] * b.v[2] - a.v[2] * b.v[1];
  c.v[1] = a.v[2] * b.v[0] - a.v[0] * b.v[2];
  return c;
}
inline double dot(const V3& a, const V3& b) {
  return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2];
}
inline void normalize_or_x(V3& a) {  // lin3_utils.hpp:142-155
  double L = std::sqrt(dot(a, a));
  if (L > 0.0) {
    double inv = 1.0 / L;
    for (int d = 0; d < 3; d++) a.v[d] *= inv;
  } else {
    a.v[0] = 1.0; a.v[1] = 0.0; a.v[2] = 0.0;
  }
}

// Null-space direction of the rank-2 symmetric matrix B (3x3): eigen3_simple.hpp:86-133.
// `rep` receives the column of B with the largest |diagonal| (a vector orthogonal to the result).
inline V3 null_vector(const double B[3][3], V3& rep) {
  int i0 = 0;
  double best = std::fabs(B[0][0]);
  for (int d = 1; d < 3; d++)
    if (std::fabs(B[d][d]) > best) { i0 = d; best = std::fabs(B[d][d]); }
  V3 col[3];
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) col[c].v[r] = B[r][c];
  rep = col[i0];
  V3 c0 = cross(rep, col[(i0 + 1) % 3]);
  V3 c1 = cross(rep, col[(i0 + 2) % 3]);
  double n0 = dot(c0, c0), n1 = dot(c1, c1);
  V3 out;
  if (n0 > n1) {
    double s = 1.0 / std::sqrt(n0);
    for (int d = 0; d < 3; d++) out.v[d] = c0.v[d] * s;
  } else {
    double s = 1.0 / std::sqrt(n1);
    for (int d = 0; d < 3; d++) out.v[d] = c1.v[d] * s;
  }
  return out;
}

// Eigenvalues lam[3] and eigenvectors as ROWS of E. eigen3_simple.hpp:137-266.
void eig_sym3(const double Ain[3][3], double lam[3], V3 E[3], int order) {
  const double eps = std::numeric_limits<double>::epsilon();
  double shift = (Ain[0][0] + Ain[1][1] + Ain[2][2]) / 3.0;
  double B[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) B[i][j] = Ain[i][j];
  for (int d = 0; d < 3; d++) B[d][d] -= shift;
  double scale = -1.0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      if (std::fabs(B[i][j]) > scale) scale = std::fabs(B[i][j]);
  if (scale > 0) {
    double inv = 1.0 / scale;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) B[i][j] *= inv;
  }
  {  // roots of x^3 - c2 x^2 + c1 x - c0 (eigen3_simple.hpp:47-82)
    const double inv3 = 1.0 / 3.0, sqrt3 = std::sqrt(3.0);
    double c0 = B[0][0] * B[1][1] * B[2][2] + 2.0 * B[1][0] * B[2][0] * B[2][1] -
                B[0][0] * B[2][1] * B[2][1] - B[1][1] * B[2][0] * B[2][0] -
                B[2][2] * B[1][0] * B[1][0];
    double c1 = B[0][0] * B[1][1] - B[1][0] * B[1][0] + B[0][0] * B[2][2] - B[2][0] * B[2][0] +
                B[1][1] * B[2][2] - B[2][1] * B[2][1];
    double c2 = B[0][0] + B[1][1] + B[2][2];
    double c2_3 = c2 * inv3;
    double a_3 = (c2 * c2_3 - c1) * inv3;
    a_3 = std::max(a_3, 0.0);
    double half_b = 0.5 * (c0 + c2_3 * (2.0 * c2_3 * c2_3 - c1));
    double q = a_3 * a_3 * a_3 - half_b * half_b;
    q = std::max(q, 0.0);
    double rho = std::sqrt(a_3);
    double theta = std::atan2(std::sqrt(q), half_b) * inv3;
    double ct = std::cos(theta), st = std::sin(theta);
    lam[0] = c2_3 - rho * (ct + sqrt3 * st);
    lam[1] = c2_3 - rho * (ct - sqrt3 * st);
    lam[2] = c2_3 + 2.0 * rho * ct;
  }
  if ((lam[2] - lam[0]) <= eps) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) E[i].v[j] = (i == j) ? 1.0 : 0.0;
  } else {
    double d0 = lam[2] - lam[1];
    double d1 = lam[1] - lam[0];
    int k = 0, l = 2;
    if (d0 > d1) { d0 = d1; std::swap(k, l); }
    double T[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) T[i][j] = B[i][j];
    for (int d = 0; d < 3; d++) T[d][d] -= lam[k];
    E[k] = null_vector(T, E[l]);
    if (d0 <= 2 * eps * d1) {
      // (eigen3_simple.hpp:214-224: the update uses E[l] on both sides)
      double kl = dot(E[k], E[l]);
      for (int d = 0; d < 3; d++) E[l].v[d] -= kl * E[l].v[d];
      normalize_or_x(E[l]);
    } else {
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[i][j] = B[i][j];
      for (int d = 0; d < 3; d++) T[d][d] -= lam[l];
      V3 dummy;
      E[l] = null_vector(T, dummy);
    }
    E[1] = cross(E[2], E[0]);
    normalize_or_x(E[1]);
  }
  for (int d = 0; d < 3; d++) { lam[d] *= scale; lam[d] += shift; }
  bool flip = (order == ORDER_INCREASING && lam[0] > lam[2]) ||
              (order == ORDER_DECREASING && lam[0] < lam[2]);
  if (flip) {
    std::swap(lam[0], lam[2]);
    std::swap(E[0], E[2]);
  }
}

// Rotation matrix (rows = eigenvectors) -> quaternion -> Shoemake triple, all in double.
// lin3_utils.hpp:230-269 and :343-375.
void frame_to_shoemake(const V3 M[3], double sm[3]) {
  double S, qw, qx, qy, qz;
  double tr = M[0].v[0] + M[1].v[1] + M[2].v[2];
  if (tr > 0) {
    S = std::sqrt(tr + 1.0) * 2;
    qw = 0.25 * S;
    qx = (M[2].v[1] - M[1].v[2]) / S;
    qy = (M[0].v[2] - M[2].v[0]) / S;
    qz = (M[1].v[0] - M[0].v[1]) / S;
  } else if ((M[0].v[0] > M[1].v[1]) && (M[0].v[0] > M[2].v[2])) {
    S = std::sqrt(1.0 + M[0].v[0] - M[1].v[1] - M[2].v[2]) * 2;
    qw = (M[2].v[1] - M[1].v[2]) / S;
    qx = 0.25 * S;
    qy = (M[0].v[1] + M[1].v[0]) / S;
    qz = (M[0].v[2] + M[2].v[0]) / S;
  } else if (M[1].v[1] > M[2].v[2]) {
    S = std::sqrt(1.0 + M[1].v[1] - M[0].v[0] - M[2].v[2]) * 2;
    qw = (M[0].v[2] - M[2].v[0]) / S;
    qx = (M[0].v[1] + M[1].v[0]) / S;
    qy = 0.25 * S;
    qz = (M[1].v[2] + M[2].v[1]) / S;
  } else {
    S = std::sqrt(1.0 + M[2].v[2] - M[0].v[0] - M[1].v[1]) * 2;
    qw = (M[1].v[0] - M[0].v[1]) / S;
    qx = (M[0].v[2] + M[2].v[0]) / S;
    qy = (M[1].v[2] + M[2].v[1]) / S;
    qz = 0.25 * S;
  }
  const double two_pi = 6.283185307179586;
  double r1 = std::sqrt(qw * qw + qx * qx);
  double r2 = std::sqrt(qy * qy + qz * qz);
  double th1 = 0.0, th2 = 0.0;
  if (r1 > 0) th1 = std::atan2(qw, qx);
  if (r2 > 0) th2 = std::atan2(qy, qz);
  sm[0] = r2 * r2;
  sm[1] = th1 / two_pi;
  sm[2] = th2 / two_pi;
}

// eigen3_simple.hpp:271-342: flat 6 floats -> [lam0,lam1,lam2, shoemake0..2] (floats).
void diagonalize_flat(const float* m6, float* out6, int order) {
  static const int MAP[3][3] = {{0, 3, 5}, {3, 1, 4}, {5, 4, 2}};  // lin3_utils.hpp:400-403
  double A[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = m6[MAP[i][j]];
  double lam[3];
  V3 E[3];
  eig_sym3(A, lam, E, order);
  V3 c01 = cross(E[0], E[1]);
  if (dot(E[2], c01) < 0.0)
    for (int d = 0; d < 3; d++) E[0].v[d] *= -1.0;
  // (the reference's in-place Transpose3 swaps every pair twice: rows stay eigenvectors)
  double sm[3];
  frame_to_shoemake(E, sm);
  out6[0] = (float)lam[0]; out6[1] = (float)lam[1]; out6[2] = (float)lam[2];
  out6[3] = (float)sm[0];  out6[4] = (float)sm[1];  out6[5] = (float)sm[2];
}

// lin3_utils.hpp:310-337 and :279-305 in float: Shoemake triple -> rows of the frame.
void shoemake_to_frame_f(const float sm[3], float M[3][3]) {
  const float two_pi = 6.283185307179586;
  float X0 = sm[0], X1 = sm[1], X2 = sm[2];
  float th1 = two_pi * X1, th2 = two_pi * X2;
  float r1 = std::sqrt(1.0 - X0);  // double subtraction, rounded on assignment
  float r2 = std::sqrt(X0);
  float s1 = std::sin(th1), c1 = std::cos(th1);
  float s2 = std::sin(th2), c2 = std::cos(th2);
  float q[4] = {s1 * r1, c1 * r1, s2 * r2, c2 * r2};
  M[0][0] = 1.0 - 2 * (q[2] * q[2]) - 2 * (q[3] * q[3]);
  M[1][1] = 1.0 - 2 * (q[1] * q[1]) - 2 * (q[3] * q[3]);
  M[2][2] = 1.0 - 2 * (q[1] * q[1]) - 2 * (q[2] * q[2]);
  M[0][1] = 2 * (q[1] * q[2] - q[3] * q[0]);
  M[1][0] = 2 * (q[1] * q[2] + q[3] * q[0]);
  M[1][2] = 2 * (q[2] * q[3] - q[1] * q[0]);
  M[2][1] = 2 * (q[2] * q[3] + q[1] * q[0]);
  M[0][2] = 2 * (q[1] * q[3] + q[2] * q[0]);
  M[2][0] = 2 * (q[1] * q[3] - q[2] * q[0]);
}

// ---------------------------------------------------------------------------------------
// A.9  Tensor-voting tables.  feature.hpp:1669-1675,2419-2482; filter3d.hpp:546-601.
// ---------------------------------------------------------------------------------------
int tv_halfwidth(float sigma, float cutoff) { return (int)std::floor(sigma * cutoff); }

void tv_tables(float sigma, int h, float* w, float* rhat) {
  float thr = 1.0f;
  {
    float e = (sigma > 0) ? std::exp(-std::pow(h / sigma, 2.0f)) : 1.0f;
    if (e < thr) thr = e;
  }
  int n = 2 * h + 1;
  float total = 0;
  for (int iz = -h; iz <= h; iz++)
    for (int iy = -h; iy <= h; iy++)
      for (int ix = -h; ix <= h; ix++) {
        float x = (!((sigma == 0.0f) && (ix == 0))) ? ix / sigma : 0.0f;
        float y = (!((sigma == 0.0f) && (iy == 0))) ? iy / sigma : 0.0f;
        float z = (!((sigma == 0.0f) && (iz == 0))) ? iz / sigma : 0.0f;
        float r = std::sqrt(x * x + y * y + z * z);
        float v = (r > 0) ? std::exp(-std::pow(r, 2.0f)) : 1.0f;
        if (std::abs(v) < thr) v = 0.0f;
        size_t k = ((size_t)(iz + h) * n + (iy + h)) * n + (ix + h);
        w[k] = v;
        total += v;
        float len = std::sqrt(ix * ix + iy * iy + iz * iz);  // double sqrt of an int, to float
        if (len == 0) len = 1.0f;
        rhat[3 * k + 0] = ix / len;
        rhat[3 * k + 1] = iy / len;
        rhat[3 * k + 2] = iz / len;
      }
  size_t m = (size_t)n * n * n;
  for (size_t k = 0; k < m; k++) w[k] /= total;
}

}  // namespace

// =========================================================================================
extern "C" {

void vo_gauss_taps(float sigma, int halfwidth, float* taps_out) {
  gauss_taps(sigma, halfwidth, taps_out);
}

// bin/filter_mrc/filter3d_variants.hpp:513-518 (all float)
float vo_ratio_from_threshold(float thr) { return std::sqrt(-2 * std::log(thr)); }

float vo_separable3d(const float* src, float* dst, const float* mask, int nx, int ny, int nz,
                     const float* tx, int hx, const float* ty, int hy, const float* tz, int hz,
                     int normalize) {
  return separable3d(src, dst, mask, nx, ny, nz, tx, hx, ty, hy, tz, hz, normalize != 0);
}

float vo_apply_gauss_hw(const float* src, float* dst, const float* mask, int nx, int ny, int nz,
                        const float sigma[3], const int hw[3], int normalize) {
  return gauss_hw(src, dst, mask, nx, ny, nz, sigma, hw, normalize != 0);
}

// lib/visfd/filter3d.hpp:1226-1258: hw = max(1, floor(sigma*ratio)) with a float product
float vo_apply_gauss_ratio(const float* src, float* dst, const float* mask, int nx, int ny, int nz,
                           const float sigma[3], float ratio, int normalize) {
  int hw[3];
  for (int d = 0; d < 3; d++) {
    hw[d] = (int)std::floor(sigma[d] * ratio);
    if (hw[d] < 1) hw[d] = 1;
  }
  return gauss_hw(src, dst, mask, nx, ny, nz, sigma, hw, normalize != 0);
}

// lib/visfd/filter3d.hpp:1698-1853 (LocalFluctuations, Gaussian weights) with the peak value of the generalised
// Gaussian window of :546-640 restated: dst = sqrt(max(A * G((src - G src)^2), 0))
void vo_local_fluctuations(const float* src, float* dst, const float* mask, int nx, int ny, int nz,
                           const float sigma[3], float exponent, float ratio, int normalize) {
  const i64 n = (i64)nx * ny * nz;
  int hw3[3];
  for (int d = 0; d < 3; d++) hw3[d] = (int)std::floor(sigma[d] * ratio);
  float cut = 1.0f;
  for (int d = 0; d < 3; d++) {
    const float e = (sigma[d] > 0) ? std::exp(-std::pow(hw3[d] / sigma[d], exponent)) : 1.0f;
    if (e < cut) cut = e;
  }
  float total = 0;
  for (int iz = -hw3[2]; iz <= hw3[2]; iz++)
    for (int iy = -hw3[1]; iy <= hw3[1]; iy++)
      for (int ix = -hw3[0]; ix <= hw3[0]; ix++) {
        const float x = (sigma[0] == 0.0f && ix == 0) ? 0.0f : ix / sigma[0];
        const float y = (sigma[1] == 0.0f && iy == 0) ? 0.0f : iy / sigma[1];
        const float z = (sigma[2] == 0.0f && iz == 0) ? 0.0f : iz / sigma[2];
        const float r = std::sqrt(x * x + y * y + z * z);
        float v = (r > 0) ? std::exp(-std::pow(r, exponent)) : 1.0f;
        if (std::fabs(v) < cut) v = 0.0f;
        total += v;
      }
  const float wpeak = 1.0f / total;
  std::vector<float> p(n);
  vo_apply_gauss_ratio(src, p.data(), mask, nx, ny, nz, sigma, ratio, normalize);
  for (i64 i = 0; i < n; i++) p[i] = src[i] - p[i];
  for (i64 i = 0; i < n; i++) p[i] *= p[i];
  vo_apply_gauss_ratio(p.data(), dst, mask, nx, ny, nz, sigma, ratio, normalize);
  for (i64 i = 0; i < n; i++) {
    float v = dst[i] * wpeak;
    if (v < 0.0f) v = 0.0f;
    dst[i] = std::sqrt(v);
  }
}

// lib/visfd/filter3d.hpp:1338-1402
void vo_apply_dog(const float* src, float* dst, const float* mask, int nx, int ny, int nz,
                  const float sigma_a[3], const float sigma_b[3], const int hw[3], float* pA,
                  float* pB) {
  i64 n = (i64)nx * ny * nz;
  std::vector<float> tmp(n);
  float A = gauss_hw(src, dst, mask, nx, ny, nz, sigma_a, hw, true);
  float B = gauss_hw(src, tmp.data(), mask, nx, ny, nz, sigma_b, hw, true);
  for (i64 i = 0; i < n; i++) dst[i] -= tmp[i];
  if (pA) *pA = A;
  if (pB) *pB = B;
}

// lib/visfd/filter3d.hpp:1428-1507
void vo_apply_log(const float* src, float* dst, const float* mask, int nx, int ny, int nz,
                  const float sigma[3], float delta, float ratio, float* pA, float* pB) {
  float sa[3], sb[3];
  int hw[3];
  for (int d = 0; d < 3; d++) {
    sa[d] = (float)(sigma[d] * (1.0 - 0.5 * delta));
    sb[d] = (float)(sigma[d] * (1.0 + 0.5 * delta));
    hw[d] = (int)std::floor(ratio * std::max(sa[d], sb[d]));
  }
  vo_apply_dog(src, dst, mask, nx, ny, nz, sa, sb, hw, pA, pB);
  // SQR(delta) is a float product (visfd_utils.hpp:32); 1.0/that is a double division
  // rounded to float on assignment (filter3d.hpp:1493).
  float scale = (float)(1.0 / (delta * delta));
  i64 n = (i64)nx * ny * nz;
  for (i64 i = 0; i < n; i++) dst[i] *= scale;
  if (pA) *pA *= scale;
  if (pB) *pB *= scale;
}

// A.5  lib/visfd/feature.hpp:53-427.  Rows: x,y,z,sigma,score.  The running per-thread
// threshold of the reference is an optimisation whose effect is removed by the final prune;
// this restatement applies the equivalent deterministic rule:
//   absolute thresholds: keep minima with score <  minima_threshold (strict, feature.hpp:272)
//   ratio thresholds:    keep minima with score <= minima_threshold*global_min (feature.hpp:386)
// (symmetric for maxima).  Output order is scan order per scale (the reference's order
// depends on thread scheduling; callers sort, handlers.cpp:876-909).
int vo_blob_dog(const float* src, const float* mask, int nx, int ny, int nz, const float* sigmas,
                int nsig, const float* aspect, float delta, float ratio, float minima_threshold,
                float maxima_threshold, int use_ratios, float* out_min, int64_t cap_min,
                int64_t* n_min, float* out_max, int64_t cap_max, int64_t* n_max) {
  const float inf = std::numeric_limits<float>::infinity();
  i64 n = (i64)nx * ny * nz;
  std::vector<float> vol[3];
  for (int k = 0; k < 3; k++) vol[k].resize(n);
  struct Blob { float x, y, z, s, score; };
  std::vector<Blob> mins, maxs;
  float asp[3] = {1.0f, 1.0f, 1.0f};
  if (aspect) for (int d = 0; d < 3; d++) asp[d] = aspect[d];
  float gmin = 1.0f, gmax = -1.0f;
  for (int ir = 0; ir < nsig; ir++) {
    float sg[3] = {sigmas[ir] * asp[0], sigmas[ir] * asp[1], sigmas[ir] * asp[2]};
    vo_apply_log(src, vol[ir % 3].data(), mask, nx, ny, nz, sg, delta, ratio, nullptr, nullptr);
    if (ir < 2) continue;
    const float* V[3] = {vol[(ir - 2) % 3].data(), vol[(ir - 1) % 3].data(), vol[ir % 3].data()};
    for (i64 iz = 0; iz < nz; iz++)
      for (i64 iy = 0; iy < ny; iy++)
        for (i64 ix = 0; ix < nx; ix++) {
          i64 c = vox(ix, iy, iz, nx, ny);
          if (mask && mask[c] == 0.0f) continue;
          float e = V[1][c];
          bool is_min = true, is_max = true;
          for (int jr = 0; jr < 3 && (is_min || is_max); jr++)
            for (int jz = -1; jz <= 1; jz++)
              for (int jy = -1; jy <= 1; jy++)
                for (int jx = -1; jx <= 1; jx++) {
                  if (jr == 1 && jz == 0 && jy == 0 && jx == 0) continue;
                  i64 X = ix + jx, Y = iy + jy, Z = iz + jz;
                  if (X < 0 || X >= nx || Y < 0 || Y >= ny || Z < 0 || Z >= nz) {
                    is_min = is_max = false;
                    continue;
                  }
                  i64 q = vox(X, Y, Z, nx, ny);
                  if (mask && mask[q] == 0.0f) { is_min = is_max = false; continue; }
                  float nb = V[jr][q];
                  if (nb <= e) is_min = false;
                  if (nb >= e) is_max = false;
                }
          if (is_min && e < 0.0f) {
            bool keep = use_ratios ? true : (e < minima_threshold);
            if (keep) {
              mins.push_back({(float)ix, (float)iy, (float)iz, sigmas[ir - 1], e});
              if (e < gmin) gmin = e;
            }
          }
          if (is_max && e > 0.0f) {
            bool keep = use_ratios ? true : (e > maxima_threshold);
            if (keep) {
              maxs.push_back({(float)ix, (float)iy, (float)iz, sigmas[ir - 1], e});
              if (e > gmax) gmax = e;
            }
          }
        }
  }
  // ratio mode, maxima_threshold = -inf: every thread of the reference compares score > (-inf) * (-1) = +inf
  // (feature.hpp:286-289; its running best starts at -1 and is never updated) and records no maximum at all
  if (use_ratios && maxima_threshold == -inf) { maxs.clear(); gmax = -1.0f; }
  if ((minima_threshold != inf) || (maxima_threshold != -inf)) {
    float tmin = minima_threshold, tmax = maxima_threshold;
    if (use_ratios) { tmin *= gmin; tmax *= gmax; }
    std::vector<Blob> a, b;
    for (auto& m : mins) if (m.score <= tmin) a.push_back(m);
    for (auto& m : maxs) if (m.score >= tmax) b.push_back(m);
    mins.swap(a);
    maxs.swap(b);
  }
  *n_min = (int64_t)mins.size();
  *n_max = (int64_t)maxs.size();
  int rc = ((int64_t)mins.size() > cap_min || (int64_t)maxs.size() > cap_max) ? 1 : 0;
  for (i64 i = 0; i < (i64)mins.size() && i < cap_min; i++) {
    const Blob& m = mins[i];
    float* o = out_min + 5 * i;
    o[0] = m.x; o[1] = m.y; o[2] = m.z; o[3] = m.s; o[4] = m.score;
  }
  for (i64 i = 0; i < (i64)maxs.size() && i < cap_max; i++) {
    const Blob& m = maxs[i];
    float* o = out_max + 5 * i;
    o[0] = m.x; o[1] = m.y; o[2] = m.z; o[3] = m.s; o[4] = m.score;
  }
  return rc;
}

// lib/visfd/feature.hpp:475 and :504 (float <-> double mixed arithmetic)
void vo_blob_diameters_to_sigmas(const float* diam, int n, float* sig) {
  for (int i = 0; i < n; i++) sig[i] = (float)(diam[i] / (2.0 * std::sqrt(3.0)));
}
void vo_blob_sigmas_to_diameters(const float* sig, int n, float* diam) {
  for (int i = 0; i < n; i++) diam[i] = (float)(sig[i] * 2.0 * std::sqrt(3.0));
}

// A.6  lib/visfd/feature.hpp:1203-1348 with lib/visfd/visfd_utils.hpp:528-669.
int vo_calc_hessian(const float* src, float* grad, float* hess, const float* mask, int nx, int ny,
                    int nz, float sigma, float ratio) {
  int hwv = (int)std::floor(sigma * ratio);
  i64 n = (i64)nx * ny * nz;
  std::vector<float> S(n);
  float sg[3] = {sigma, sigma, sigma};
  int hw[3] = {hwv, hwv, hwv};
  gauss_hw(src, S.data(), mask, nx, ny, nz, sg, hw, true);
  if (nx < 3 || ny < 3 || nz < 3) return 1;
  float s2 = sigma * sigma;
  #pragma omp parallel for collapse(2)
  for (i64 iz = 0; iz < nz; iz++)
    for (i64 iy = 0; iy < ny; iy++)
      for (i64 ix = 0; ix < nx; ix++) {
        i64 c = vox(ix, iy, iz, nx, ny);
        if (mask && mask[c] == 0.0f) continue;
        i64 x = std::min<i64>(std::max<i64>(ix, 1), nx - 2);
        i64 y = std::min<i64>(std::max<i64>(iy, 1), ny - 2);
        i64 z = std::min<i64>(std::max<i64>(iz, 1), nz - 2);
        auto F = [&](i64 dx, i64 dy, i64 dz) { return S[vox(x + dx, y + dy, z + dz, nx, ny)]; };
        if (grad) {
          float g0 = (float)(0.5 * (F(1, 0, 0) - F(-1, 0, 0)));
          float g1 = (float)(0.5 * (F(0, 1, 0) - F(0, -1, 0)));
          float g2 = (float)(0.5 * (F(0, 0, 1) - F(0, 0, -1)));
          grad[3 * c + 0] = g0 * sigma;
          grad[3 * c + 1] = g1 * sigma;
          grad[3 * c + 2] = g2 * sigma;
        }
        if (hess) {
          float f0 = F(0, 0, 0);
          float hxx = (F(1, 0, 0) + F(-1, 0, 0)) - 2 * f0;
          float hyy = (F(0, 1, 0) + F(0, -1, 0)) - 2 * f0;
          float hzz = (F(0, 0, 1) + F(0, 0, -1)) - 2 * f0;
          float hxy = (float)(0.25 * (((F(1, 1, 0) + F(-1, -1, 0)) - F(1, -1, 0)) - F(-1, 1, 0)));
          float hyz = (float)(0.25 * (((F(0, 1, 1) + F(0, -1, -1)) - F(0, 1, -1)) - F(0, -1, 1)));
          float hzx = (float)(0.25 * (((F(1, 0, 1) + F(-1, 0, -1)) - F(-1, 0, 1)) - F(1, 0, -1)));
          float* o = hess + 6 * c;
          o[0] = hxx * s2; o[1] = hyy * s2; o[2] = hzz * s2;
          o[3] = hxy * s2; o[4] = hyz * s2; o[5] = hzx * s2;
        }
      }
  return 0;
}

void vo_diagonalize_flat_sym3(const float* m6, float* out6, int64_t n, int order) {
  #pragma omp parallel for
  for (int64_t i = 0; i < n; i++) diagonalize_flat(m6 + 6 * i, out6 + 6 * i, order);
}

// lib/visfd/eigen3_simple.hpp:392-405 + lib/visfd/lin3_utils.hpp:566-584
void vo_flat_sym_to_evects(const float* m6, float* eivals3, float* eivects9, int64_t n, int order) {
  #pragma omp parallel for
  for (int64_t i = 0; i < n; i++) {
    float d[6], M[3][3];
    diagonalize_flat(m6 + 6 * i, d, order);
    shoemake_to_frame_f(d + 3, M);
    for (int k = 0; k < 3; k++) eivals3[3 * i + k] = d[k];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) eivects9[9 * i + 3 * a + b] = M[a][b];
  }
}

// A.8  bin/filter_mrc/handlers.cpp:1640-1746 (SURFACE_RIDGE, no background) with
// lib/visfd/feature.hpp:1526-1561.
void vo_hessian_saliency(const float* hess, const float* mask, int64_t n, int order,
                         float* saliency, float* dir) {
  #pragma omp parallel for
  for (int64_t i = 0; i < n; i++) {
    saliency[i] = 0.0f;
    if (mask && mask[i] == 0.0f) continue;
    float d[6], M[3][3];
    diagonalize_flat(hess + 6 * i, d, order);
    shoemake_to_frame_f(d + 3, M);
    double l1 = d[0], l2 = d[1];
    double N = l1 * l1 - l2 * l2;
    N *= N;
    float score = (float)N;
    score *= 1.0f;
    saliency[i] = score;
    dir[3 * i + 0] = M[0][0];
    dir[3 * i + 1] = M[0][1];
    dir[3 * i + 2] = M[0][2];
  }
}

// A.8  bin/filter_mrc/handlers.cpp:1751-1797
float vo_threshold_fraction(float* saliency, const float* mask, int64_t n, float fraction) {
  std::vector<float> s;
  s.reserve(n);
  for (int64_t i = 0; i < n; i++) {
    if (mask && mask[i] == 0.0f) continue;
    s.push_back(saliency[i]);
  }
  size_t nv = s.size();
  size_t k = (size_t)std::floor(nv * fraction);  // size_t -> float product, as in the reference
  std::nth_element(s.begin(), s.begin() + k, s.end(), std::greater<float>());
  float thr = s[k];
  for (int64_t i = 0; i < n; i++)
    if (saliency[i] < thr) saliency[i] = 0.0f;
  return thr;
}

int vo_tv_tables(float sigma_tv, float cutoff_ratio, float* w, float* rhat, int cap_h) {
  int h = tv_halfwidth(sigma_tv, cutoff_ratio);
  if (h > cap_h) return -h;
  if (!w) return h;
  tv_tables(sigma_tv, h, w, rhat);
  return h;
}

// A.10  lib/visfd/feature.hpp:1914-2037 and :2217-2384; normalize / diagonalize: the public wrapper, :1761-1901, as the
// reference EXECUTES it (off the CLI path, handlers.cpp:1832-1835 passes false, false):
//   normalize with a source mask: each tensor entry /= sum of the vote weights, where that sum is > 0 -- but only at
//     voxels with mask_dst != 0, and NOT AT ALL without a destination mask (the loop `continue`s on !aaafMaskDest, :1793);
//   normalize without a source mask: again only with a destination mask (:1848); the divisor is the product of three 1-D
//     Gaussian (GenFilterGauss1D(sigma_tv, h), NOT the vote weights) sums over the in-image taps, (Dx*Dy)*Dz, and the loop
//     runs over all nine (di, dj): diagonal entries are divided once, off-diagonal entries TWICE (:1854-1858);
//   diagonalize: DiagonalizeHessianImage (:1367-1471) with DECREASING_EIVALS, interior voxels 1..n-2 only, mask_dst != 0.
void vo_tv_dense_stick(const float* saliency, const float* dir, float* tensor,
                       const float* mask_src, const float* mask_dst, int nx, int ny, int nz,
                       float sigma_tv, int exponent, float cutoff_ratio, int curves,
                       int normalize, int diagonalize) {
  std::vector<float> den;
  if (normalize && mask_src) den.assign((size_t)nx * ny * nz, 0.0f);
  int h = tv_halfwidth(sigma_tv, cutoff_ratio);
  int nw = 2 * h + 1;
  std::vector<float> w((size_t)nw * nw * nw), rh((size_t)3 * nw * nw * nw);
  tv_tables(sigma_tv, h, w.data(), rh.data());
  #pragma omp parallel for collapse(2) schedule(dynamic, 8)
  for (i64 iz = 0; iz < nz; iz++)
    for (i64 iy = 0; iy < ny; iy++)
      for (i64 ix = 0; ix < nx; ix++) {
        i64 c = vox(ix, iy, iz, nx, ny);
        if (mask_dst && mask_dst[c] == 0.0f) continue;
        float T[6] = {0, 0, 0, 0, 0, 0};
        float denominator = 0.0f;
        for (int jz = -h; jz <= h; jz++) {
          i64 sz = iz - jz;
          if (sz < 0 || sz >= nz) continue;
          for (int jy = -h; jy <= h; jy++) {
            i64 sy = iy - jy;
            if (sy < 0 || sy >= ny) continue;
            for (int jx = -h; jx <= h; jx++) {
              i64 sx = ix - jx;
              if (sx < 0 || sx >= nx) continue;
              size_t k = ((size_t)(jz + h) * nw + (jy + h)) * nw + (jx + h);
              float fv = w[k];
              i64 s = vox(sx, sy, sz, nx, ny);
              if (mask_src) {
                float mv = mask_src[s];
                if (mv == 0.0f) continue;
                fv *= mv;
              }
              float sal = saliency[s];
              if (sal == 0.0f) continue;
              if (fv == 0.0f) continue;
              const float* r = &rh[3 * k];
              const float* nn = dir + 3 * s;
              float u = (r[0] * nn[0] + r[1] * nn[1]) + r[2] * nn[2];
              float ux2 = u * 2.0f;
              float u2 = u * u;
              float c2 = 1.0f - u2;
              float ang = curves ? u2 : c2;
              float dec;
              if (exponent == 2) dec = ang;
              else if (exponent == 4) dec = ang * ang;
              else dec = (float)std::pow((double)ang, 0.5 * exponent);
              float m[3];
              for (int d = 0; d < 3; d++) m[d] = curves ? (nn[d] - ux2 * r[d]) : (ux2 * r[d] - nn[d]);
              float base = (sal * fv) * dec;
              T[0] += (base * m[0]) * m[0];
              T[3] += (base * m[0]) * m[1];
              T[5] += (base * m[0]) * m[2];
              T[1] += (base * m[1]) * m[1];
              T[4] += (base * m[1]) * m[2];
              T[2] += (base * m[2]) * m[2];
              denominator += fv;      // feature.hpp:2376-2377
            }
          }
        }
        for (int d = 0; d < 6; d++) tensor[6 * c + d] = T[d];
        if (!den.empty()) den[c] = denominator;
      }
  if (normalize && mask_dst) {
    if (mask_src) {
      for (i64 c = 0; c < (i64)nx * ny * nz; c++)
        if (mask_dst[c] != 0.0f && den[c] > 0.0f)
          for (int d = 0; d < 6; d++) tensor[6 * c + d] /= den[c];
    } else {
      std::vector<float> t(nw);
      gauss_taps(sigma_tv, h, t.data());
      std::vector<float> D[3];
      const int dims[3] = {nx, ny, nz};
      for (int d = 0; d < 3; d++) {
        std::vector<float> ones((size_t)dims[d], 1.0f);
        D[d].resize(dims[d]);
        conv_line_plain(dims[d], ones.data(), 1, D[d].data(), 1, t.data(), h);   // Filter1D::Apply on a line of ones
      }
      for (i64 iz = 0; iz < nz; iz++)
        for (i64 iy = 0; iy < ny; iy++)
          for (i64 ix = 0; ix < nx; ix++) {
            i64 c = vox(ix, iy, iz, nx, ny);
            if (mask_dst[c] == 0.0f) continue;
            const float denominator = (D[0][ix] * D[1][iy]) * D[2][iz];
            float* T = tensor + 6 * c;
            for (int d = 0; d < 3; d++) T[d] /= denominator;                              // (0,0), (1,1), (2,2)
            for (int d = 3; d < 6; d++) { T[d] /= denominator; T[d] /= denominator; }   // (di,dj) and (dj,di)
          }
    }
  }
  if (diagonalize)
    for (i64 iz = 1; iz < nz - 1; iz++)
      for (i64 iy = 1; iy < ny - 1; iy++)
        for (i64 ix = 1; ix < nx - 1; ix++) {
          i64 c = vox(ix, iy, iz, nx, ny);
          if (mask_dst && mask_dst[c] == 0.0f) continue;
          float d6[6];
          diagonalize_flat(tensor + 6 * c, d6, 1 /* DECREASING_EIVALS */);
          for (int d = 0; d < 6; d++) tensor[6 * c + d] = d6[d];
        }
}

// A.11  bin/filter_mrc/handlers.cpp:1870-1892 with lib/visfd/feature.hpp:1591-1598
void vo_tensor_saliency(const float* tensor, const float* mask, int64_t n, int order,
                        float* saliency_inout) {
  #pragma omp parallel for
  for (int64_t i = 0; i < n; i++) {
    if (mask && mask[i] == 0.0f) continue;
    float d[6];
    diagonalize_flat(tensor + 6 * i, d, order);
    double l1 = d[0], l2 = d[1];
    float score = (float)(l1 - l2);
    score *= 1.0f;
    saliency_inout[i] = score;
  }
}

// ---- binning (resample.hpp:53-166) ----------------------------------------------
// sizes are {nx, ny, nz}; offset may be null.  Returns 0, or 1 for an offset outside [0, bin).
int vo_bin_array3d(const float* src, const int* ssz, float* dst, const int* dsz, const int* offset) {
  int b[3], o[3];
  for (int d = 0; d < 3; d++) {
    b[d] = ssz[d] / dsz[d];
    o[d] = offset ? offset[d] : 0;
    if (o[d] < 0 || o[d] >= b[d]) return 1;                                  // resample.hpp:63-69
  }
  for (int64_t Iz = 0; Iz < dsz[2]; Iz++)
    for (int64_t Iy = 0; Iy < dsz[1]; Iy++)
      for (int64_t Ix = 0; Ix < dsz[0]; Ix++) {
        float sum = 0.0f;                                                      // resample.hpp:74-91: z, y, x order
        for (int dz = 0; dz < b[2]; dz++)
          for (int dy = 0; dy < b[1]; dy++)
            for (int dx = 0; dx < b[0]; dx++) {
              const int64_t ix = Ix * b[0] + dx + o[0], iy = Iy * b[1] + dy + o[1], iz = Iz * b[2] + dz + o[2];
              sum = sum + src[(iz * ssz[1] + iy) * ssz[0] + ix];
            }
        dst[(Iz * dsz[1] + Iy) * dsz[0] + Ix] = sum / (float)(b[0] * b[1] * b[2]);   // :93 (Scalar / int)
      }
  return 0;
}

int vo_unbin_array3d(const float* src, const int* ssz, float* dst, const int* dsz, const int* offset) {
  int b[3], o[3];
  for (int d = 0; d < 3; d++) {
    b[d] = dsz[d] / ssz[d];
    o[d] = offset ? offset[d] : 0;
    if (o[d] < 0 || o[d] >= b[d]) return 1;                                  // resample.hpp:129-135
  }
  for (int64_t Iz = 0; Iz < dsz[2]; Iz++)
    for (int64_t Iy = 0; Iy < dsz[1]; Iy++)
      for (int64_t Ix = 0; Ix < dsz[0]; Ix++) {
        int64_t ix = (Ix - o[0]) / b[0], iy = (Iy - o[1]) / b[1], iz = (Iz - o[2]) / b[2];   // truncating division
        if (ix < 0) ix = 0;
        if (iy < 0) iy = 0;
        if (iz < 0) iz = 0;
        if (ix >= ssz[0]) ix = ssz[0] - 1;
        if (iy >= ssz[1]) iy = ssz[1] - 1;
        if (iz >= ssz[2]) iz = ssz[2] - 1;
        dst[(Iz * dsz[1] + Iy) * dsz[0] + Ix] = src[(iz * ssz[1] + iy) * ssz[0] + ix];
      }
  return 0;
}

int vo_version() { return 3; }

}  // extern "C"
