// oracle_math.hpp -- tiny vector/quaternion algebra for the CPU oracle (TEST INFRASTRUCTURE ONLY).
//
// Part of oracle/: the CPU restatement used to check the HIP kernels. Nothing under
// maniskill_amd/ may include, link or call this. wxyz quaternions as in the reference's Pose
// (mani_skill/utils/structs/pose.py:37-38, 104-108).
#pragma once
#include <cmath>

namespace orc {

template <typename R>
struct V3 {
  R x, y, z;
  V3() : x(0), y(0), z(0) {}
  V3(R a, R b, R c) : x(a), y(b), z(c) {}
  R operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  R& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
  V3 operator+(const V3& o) const { return V3(x + o.x, y + o.y, z + o.z); }
  V3 operator-(const V3& o) const { return V3(x - o.x, y - o.y, z - o.z); }
  V3 operator-() const { return V3(-x, -y, -z); }
  V3 operator*(R s) const { return V3(x * s, y * s, z * s); }
  V3& operator+=(const V3& o) { x += o.x; y += o.y; z += o.z; return *this; }
  V3& operator-=(const V3& o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
};
template <typename R> inline R dot(const V3<R>& a, const V3<R>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename R> inline V3<R> cross(const V3<R>& a, const V3<R>& b) {
  return V3<R>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename R> inline R norm(const V3<R>& a) { return std::sqrt(dot(a, a)); }
// |w| <= wmax (free-body angular velocity limit, MSSIM_MAX_ANGULAR_VELOCITY in include/mssim.h)
template <typename R> inline V3<R> clamp_norm(const V3<R>& w, R wmax) {
  const R n = norm(w);
  return n > wmax ? w * (wmax / n) : w;
}
template <typename R> inline V3<R> normalized(const V3<R>& a) {
  R n = norm(a);
  return n > R(0) ? a * (R(1) / n) : V3<R>(1, 0, 0);
}

template <typename R>
struct Q4 {  // w, x, y, z
  R w, x, y, z;
  Q4() : w(1), x(0), y(0), z(0) {}
  Q4(R a, R b, R c, R d) : w(a), x(b), y(c), z(d) {}
};
template <typename R> inline Q4<R> qmul(const Q4<R>& a, const Q4<R>& b) {
  return Q4<R>(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
               a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w);
}
template <typename R> inline Q4<R> qnormalized(const Q4<R>& q) {
  R n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  R s = n > R(0) ? R(1) / n : R(1);
  return Q4<R>(q.w * s, q.x * s, q.y * s, q.z * s);
}
template <typename R> inline Q4<R> qconj(const Q4<R>& q) { return Q4<R>(q.w, -q.x, -q.y, -q.z); }
template <typename R> inline Q4<R> qaxis_angle(const V3<R>& axis, R angle) {
  R h = angle * R(0.5), s = std::sin(h);
  return Q4<R>(std::cos(h), axis.x * s, axis.y * s, axis.z * s);
}

template <typename R>
struct M3 {  // row major
  R m[3][3];
  V3<R> col(int j) const { return V3<R>(m[0][j], m[1][j], m[2][j]); }
  V3<R> row(int i) const { return V3<R>(m[i][0], m[i][1], m[i][2]); }
  V3<R> operator*(const V3<R>& v) const {
    return V3<R>(m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
                 m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z);
  }
  V3<R> tmul(const V3<R>& v) const {  // transpose * v
    return V3<R>(m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z, m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z,
                 m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z);
  }
};
template <typename R> inline M3<R> qmat(const Q4<R>& q) {
  M3<R> r;
  R w = q.w, x = q.x, y = q.y, z = q.z;
  r.m[0][0] = 1 - 2 * (y * y + z * z); r.m[0][1] = 2 * (x * y - w * z); r.m[0][2] = 2 * (x * z + w * y);
  r.m[1][0] = 2 * (x * y + w * z); r.m[1][1] = 1 - 2 * (x * x + z * z); r.m[1][2] = 2 * (y * z - w * x);
  r.m[2][0] = 2 * (x * z - w * y); r.m[2][1] = 2 * (y * z + w * x); r.m[2][2] = 1 - 2 * (x * x + y * y);
  return r;
}
template <typename R> inline V3<R> qrot(const Q4<R>& q, const V3<R>& v) { return qmat(q) * v; }
template <typename R> inline M3<R> mmul(const M3<R>& a, const M3<R>& b) {
  M3<R> r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
template <typename R> inline M3<R> mtranspose(const M3<R>& a) {
  M3<R> r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
  return r;
}
// symmetric 3x3 from (xx,yy,zz,xy,xz,yz)
template <typename R> inline M3<R> sym3(const R* v) {
  M3<R> r;
  r.m[0][0] = v[0]; r.m[1][1] = v[1]; r.m[2][2] = v[2];
  r.m[0][1] = r.m[1][0] = v[3]; r.m[0][2] = r.m[2][0] = v[4]; r.m[1][2] = r.m[2][1] = v[5];
  return r;
}
template <typename R> inline M3<R> minverse(const M3<R>& a) {
  M3<R> r;
  R c00 = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
  R c01 = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
  R c02 = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
  R det = a.m[0][0] * c00 + a.m[0][1] * c01 + a.m[0][2] * c02;
  R id = R(1) / det;
  r.m[0][0] = c00 * id; r.m[0][1] = (a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2]) * id;
  r.m[0][2] = (a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1]) * id;
  r.m[1][0] = c01 * id; r.m[1][1] = (a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0]) * id;
  r.m[1][2] = (a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2]) * id;
  r.m[2][0] = c02 * id; r.m[2][1] = (a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1]) * id;
  r.m[2][2] = (a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0]) * id;
  return r;
}

template <typename R>
struct Pose {
  V3<R> p;
  Q4<R> q;
};
template <typename R> inline Pose<R> pmul(const Pose<R>& a, const Pose<R>& b) {
  Pose<R> r;
  r.p = a.p + qrot(a.q, b.p);
  r.q = qnormalized(qmul(a.q, b.q));
  return r;
}

}  // namespace orc
