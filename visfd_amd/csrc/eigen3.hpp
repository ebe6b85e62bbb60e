// eigen3.hpp -- 3x3 symmetric eigen-decomposition and the Shoemake frame packing used by the ridge
// detector and the tensor-voting score (device), and by the host-side voxel clustering (connect.cpp).
//
// Behavioural contract (reference lib/visfd/eigen3_simple.hpp:47-342 and
// lib/visfd/lin3_utils.hpp:230-394).  Written from the mathematics:
//   * eigenvalues: trigonometric (Viete) solution of the characteristic cubic of the matrix after
//     shifting by trace/3 and scaling by max|entry|, in double; ascending;
//   * eigenvectors (rows): null vector of (B - lambda I) from the largest cross product of its
//     columns, for the most isolated eigenvalue first, then the one at the other end, the middle
//     one by cross product;
//   * requested order (increasing / decreasing) only swaps entries 0 and 2;
//   * frame -> quaternion -> Shoemake triple stored as float; unpacking in float.
// The device libm differs from glibc in the last ulp of atan2/sin/cos, so device results agree with
// the CPU path to ~1e-7 relative rather than bit-for-bit (tests use the 1e-5 relative bound of
// BASELINE.json).  Compiled for the host the same functions use glibc and are bit-identical to the
// reference's CPU results.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#define VH_HD __host__ __device__ __forceinline__

namespace vh {
namespace eig {

struct D3 { double x, y, z; };

VH_HD D3 cross3(const D3& a, const D3& b) {
  D3 c;
  c.z = a.x * b.y - a.y * b.x;
  c.x = a.y * b.z - a.z * b.y;
  c.y = a.z * b.x - a.x * b.z;
  return c;
}
VH_HD double dot3(const D3& a, const D3& b) {
  return a.x * b.x + a.y * b.y + a.z * b.z;
}
VH_HD void unit_or_x(D3& a) {
  const double L = sqrt(dot3(a, a));
  if (L > 0.0) {
    const double inv = 1.0 / L;
    a.x *= inv; a.y *= inv; a.z *= inv;
  } else {
    a.x = 1.0; a.y = 0.0; a.z = 0.0;
  }
}

// symmetric matrix, full storage by columns (col[k] = k-th column = k-th row)
struct Sym3 {
  double m00, m11, m22, m01, m12, m02;
};

// Null vector of the (numerically rank-2) matrix S - lam*I; rep = its column with the largest
// |diagonal| entry.
VH_HD D3 null_vector(const Sym3& S, double lam, D3& rep) {
  const double d0 = S.m00 - lam, d1 = S.m11 - lam, d2 = S.m22 - lam;
  const D3 c0 = {d0, S.m01, S.m02};
  const D3 c1 = {S.m01, d1, S.m12};
  const D3 c2 = {S.m02, S.m12, d2};
  int i0 = 0;
  double best = fabs(d0);
  if (fabs(d1) > best) { i0 = 1; best = fabs(d1); }
  if (fabs(d2) > best) { i0 = 2; }
  D3 a, b;
  if (i0 == 0) { rep = c0; a = c1; b = c2; }
  else if (i0 == 1) { rep = c1; a = c2; b = c0; }
  else { rep = c2; a = c0; b = c1; }
  const D3 x0 = cross3(rep, a);
  const D3 x1 = cross3(rep, b);
  const double n0 = dot3(x0, x0), n1 = dot3(x1, x1);
  D3 r;
  if (n0 > n1) {
    const double s = 1.0 / sqrt(n0);
    r.x = x0.x * s; r.y = x0.y * s; r.z = x0.z * s;
  } else {
    const double s = 1.0 / sqrt(n1);
    r.x = x1.x * s; r.y = x1.y * s; r.z = x1.z * s;
  }
  return r;
}

// m6 = (xx,yy,zz,xy,yz,xz).  lam[3] and rows E[3]; order 0 = increasing, 1 = decreasing.
// F32TRIG (device only; context option eig_f32, tolerance modes): the ONE angle of the root formula -- atan2, sin, cos -- in
// single precision.  The reference takes it in double (eigen3_simple.hpp:74-81); two double-precision libms (the device's,
// glibc's) agree to ~1e-16, which vanishes when the eigenvalue is stored as float -- so F32TRIG = false reproduces the
// reference's float eigenvalues in almost every voxel, while the single-precision angle moves almost every eigenvalue by
// about one float ulp (~1e-7 of the matrix's scale).  It is therefore NOT part of the default (exact) kernels.
template <bool F32TRIG = false>
VH_HD void eig_sym3(const float m6[6], int order, double lam[3], D3 E[3],
                                         bool want_vectors) {
  const double eps = 2.220446049250313e-16;
  const double a00 = m6[0], a11 = m6[1], a22 = m6[2], a01 = m6[3], a12 = m6[4], a02 = m6[5];
  const double shift = (a00 + a11 + a22) / 3.0;
  Sym3 B = {a00 - shift, a11 - shift, a22 - shift, a01, a12, a02};
  double scale = fmax(fmax(fabs(B.m00), fabs(B.m11)), fabs(B.m22));
  scale = fmax(scale, fmax(fmax(fabs(B.m01), fabs(B.m12)), fabs(B.m02)));
  if (scale > 0) {
    const double inv = 1.0 / scale;
    B.m00 *= inv; B.m11 *= inv; B.m22 *= inv; B.m01 *= inv; B.m12 *= inv; B.m02 *= inv;
  }
  {
    const double inv3 = 1.0 / 3.0;
    const double sqrt3 = 1.7320508075688772;
    // det(B), sum of principal 2x2 minors, trace
    const double c0 = B.m00 * B.m11 * B.m22 + 2.0 * B.m01 * B.m02 * B.m12 - B.m00 * B.m12 * B.m12 -
                      B.m11 * B.m02 * B.m02 - B.m22 * B.m01 * B.m01;
    const double c1 = B.m00 * B.m11 - B.m01 * B.m01 + B.m00 * B.m22 - B.m02 * B.m02 +
                      B.m11 * B.m22 - B.m12 * B.m12;
    const double c2 = B.m00 + B.m11 + B.m22;
    const double c2_3 = c2 * inv3;
    double a_3 = (c2 * c2_3 - c1) * inv3;
    a_3 = fmax(a_3, 0.0);
    const double half_b = 0.5 * (c0 + c2_3 * (2.0 * c2_3 * c2_3 - c1));
    double q = a_3 * a_3 * a_3 - half_b * half_b;
    q = fmax(q, 0.0);
    const double rho = sqrt(a_3);
    double st, ct;
#if defined(__HIP_DEVICE_COMPILE__)
    if (F32TRIG) {
      // (the matrix is scaled to max |entry| = 1, so both arguments of atan2f are of order one or smaller; everything
      // around the angle stays in double)
      const float thf = atan2f(sqrtf((float)q), (float)half_b) * (1.0f / 3.0f);
      float sf, cf;
      sincosf(thf, &sf, &cf);
      st = (double)sf; ct = (double)cf;
    } else {
      const double theta = atan2(sqrt(q), half_b) * inv3;
      sincos(theta, &st, &ct);
    }
#else
    const double theta = atan2(sqrt(q), half_b) * inv3;
    ct = std::cos(theta); st = std::sin(theta);
#endif
    lam[0] = c2_3 - rho * (ct + sqrt3 * st);
    lam[1] = c2_3 - rho * (ct - sqrt3 * st);
    lam[2] = c2_3 + 2.0 * rho * ct;
  }
  if (want_vectors) {
    if ((lam[2] - lam[0]) <= eps) {
      E[0] = {1.0, 0.0, 0.0};
      E[1] = {0.0, 1.0, 0.0};
      E[2] = {0.0, 0.0, 1.0};
    } else {
      double gap_hi = lam[2] - lam[1];
      const double gap_lo = lam[1] - lam[0];
      // the more isolated end first
      const bool top_first = !(gap_hi > gap_lo) ? false : true;
      // reference: k=0,l=2; if (d0 > d1) {d0 = d1; swap(k,l);}  with d0 = gap_hi, d1 = gap_lo
      double dmin = gap_hi;
      if (top_first) dmin = gap_lo;
      const double lam_k = top_first ? lam[2] : lam[0];
      const double lam_l = top_first ? lam[0] : lam[2];
      D3 vk, vl;
      vk = null_vector(B, lam_k, vl);
      if (dmin <= 2 * eps * gap_lo) {
        const double kl = dot3(vk, vl);
        vl.x -= kl * vl.x; vl.y -= kl * vl.y; vl.z -= kl * vl.z;
        unit_or_x(vl);
      } else {
        D3 dummy;
        vl = null_vector(B, lam_l, dummy);
      }
      if (top_first) { E[2] = vk; E[0] = vl; } else { E[0] = vk; E[2] = vl; }
      E[1] = cross3(E[2], E[0]);
      unit_or_x(E[1]);
    }
  }
  lam[0] = lam[0] * scale + shift;
  lam[1] = lam[1] * scale + shift;
  lam[2] = lam[2] * scale + shift;
  const bool swap = (order == 0) ? (lam[0] > lam[2]) : (lam[0] < lam[2]);
  if (swap) {
    const double t = lam[0]; lam[0] = lam[2]; lam[2] = t;
    if (want_vectors) { const D3 v = E[0]; E[0] = E[2]; E[2] = v; }
  }
}

// rows of a rotation -> quaternion (w,x,y,z) -> Shoemake triple (double in, float out)
VH_HD void frame_to_shoemake(const D3 M[3], float sm[3]) {
  const double m00 = M[0].x, m01 = M[0].y, m02 = M[0].z;
  const double m10 = M[1].x, m11 = M[1].y, m12 = M[1].z;
  const double m20 = M[2].x, m21 = M[2].y, m22 = M[2].z;
  double S, qw, qx, qy, qz;
  const double tr = m00 + m11 + m22;
  if (tr > 0) {
    S = sqrt(tr + 1.0) * 2;
    qw = 0.25 * S; qx = (m21 - m12) / S; qy = (m02 - m20) / S; qz = (m10 - m01) / S;
  } else if ((m00 > m11) && (m00 > m22)) {
    S = sqrt(1.0 + m00 - m11 - m22) * 2;
    qw = (m21 - m12) / S; qx = 0.25 * S; qy = (m01 + m10) / S; qz = (m02 + m20) / S;
  } else if (m11 > m22) {
    S = sqrt(1.0 + m11 - m00 - m22) * 2;
    qw = (m02 - m20) / S; qx = (m01 + m10) / S; qy = 0.25 * S; qz = (m12 + m21) / S;
  } else {
    S = sqrt(1.0 + m22 - m00 - m11) * 2;
    qw = (m10 - m01) / S; qx = (m02 + m20) / S; qy = (m12 + m21) / S; qz = 0.25 * S;
  }
  const double two_pi = 6.283185307179586;
  const double r1 = sqrt(qw * qw + qx * qx);
  const double r2 = sqrt(qy * qy + qz * qz);
  double th1 = 0.0, th2 = 0.0;
  if (r1 > 0) th1 = atan2(qw, qx);
  if (r2 > 0) th2 = atan2(qy, qz);
  sm[0] = (float)(r2 * r2);
  sm[1] = (float)(th1 / two_pi);
  sm[2] = (float)(th2 / two_pi);
}

// flat symmetric matrix -> [lam0, lam1, lam2, shoemake0..2] (eigen3_simple.hpp:271-342)
template <bool F32TRIG = false>
VH_HD void diagonalize_flat(const float m6[6], int order, float out6[6]) {
  double lam[3];
  D3 E[3];
  eig_sym3<F32TRIG>(m6, order, lam, E, true);
  const D3 c01 = cross3(E[0], E[1]);
  if (dot3(E[2], c01) < 0.0) { E[0].x = -E[0].x; E[0].y = -E[0].y; E[0].z = -E[0].z; }
  float sm[3];
  frame_to_shoemake(E, sm);
  out6[0] = (float)lam[0]; out6[1] = (float)lam[1]; out6[2] = (float)lam[2];
  out6[3] = sm[0]; out6[4] = sm[1]; out6[5] = sm[2];
}

// First row of the frame recovered from the float Shoemake triple (lin3_utils.hpp:310-337 and
// :279-305, float arithmetic): the principal direction handed to tensor voting.
VH_HD void shoemake_row0(const float sm[3], float row0[3]) {
  const float two_pi = 6.283185307179586f;
  const float X0 = sm[0];
  const float th1 = two_pi * sm[1], th2 = two_pi * sm[2];
  const float r1 = (float)sqrt(1.0 - (double)X0);
  const float r2 = sqrtf(X0);
  float s1, c1, s2, c2;
#if defined(__HIP_DEVICE_COMPILE__)
  sincosf(th1, &s1, &c1);
  sincosf(th2, &s2, &c2);
#else
  s1 = std::sin(th1); c1 = std::cos(th1);
  s2 = std::sin(th2); c2 = std::cos(th2);
#endif
  const float q0 = s1 * r1, q1 = c1 * r1, q2 = s2 * r2, q3 = c2 * r2;
  row0[0] = (float)(1.0 - (double)(2 * (q2 * q2)) - (double)(2 * (q3 * q3)));
  row0[1] = 2 * (q1 * q2 - q3 * q0);
  row0[2] = 2 * (q1 * q3 + q2 * q0);
}

// The whole frame (eigenvectors as rows) recovered from the float Shoemake triple: Shoemake2Quaternion and
// Quaternion2Matrix instantiated for float (lin3_utils.hpp:310-337, :279-305), where the double literal 1.0
// promotes the diagonal entries only.  Row 0 is shoemake_row0.
VH_HD void shoemake_frame(const float sm[3], float M[3][3]) {
  const float two_pi = 6.283185307179586f;
  const float X0 = sm[0];
  const float th1 = two_pi * sm[1], th2 = two_pi * sm[2];
  const float r1 = (float)sqrt(1.0 - (double)X0);
  const float r2 = sqrtf(X0);
  float s1, c1, s2, c2;
#if defined(__HIP_DEVICE_COMPILE__)
  sincosf(th1, &s1, &c1);
  sincosf(th2, &s2, &c2);
#else
  s1 = std::sin(th1); c1 = std::cos(th1);
  s2 = std::sin(th2); c2 = std::cos(th2);
#endif
  const float q0 = s1 * r1, q1 = c1 * r1, q2 = s2 * r2, q3 = c2 * r2;
  M[0][0] = (float)(1.0 - (double)(2 * (q2 * q2)) - (double)(2 * (q3 * q3)));
  M[1][1] = (float)(1.0 - (double)(2 * (q1 * q1)) - (double)(2 * (q3 * q3)));
  M[2][2] = (float)(1.0 - (double)(2 * (q1 * q1)) - (double)(2 * (q2 * q2)));
  M[0][1] = 2 * (q1 * q2 - q3 * q0);
  M[1][0] = 2 * (q1 * q2 + q3 * q0);
  M[1][2] = 2 * (q2 * q3 - q1 * q0);
  M[2][1] = 2 * (q2 * q3 + q1 * q0);
  M[0][2] = 2 * (q1 * q3 + q2 * q0);
  M[2][0] = 2 * (q1 * q3 - q2 * q0);
}

}  // namespace eig
}  // namespace vh
