// mssim_dev.h -- device-side f32 vector / quaternion helpers for the gfx950 kernels.
// One env (or one env x pair) per lane; everything here is per-lane scalar code that the
// compiler keeps in VGPRs. wxyz quaternions (reference: mani_skill/utils/structs/pose.py:37-38).
#pragma once
#include <hip/hip_runtime.h>

#define MS_DEV __device__ __forceinline__

struct f3 {
  float x, y, z;
};
MS_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
MS_DEV f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MS_DEV f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MS_DEV f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
MS_DEV f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
MS_DEV f3& operator+=(f3& a, f3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
MS_DEV f3& operator-=(f3& a, f3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
MS_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MS_DEV f3 cross(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// single-instruction reciprocal / square root (v_rcp_f32, v_sqrt_f32, v_rsq_f32: 1 ulp). `1.f / x`, `a / b` and
// sqrtf() compile to 10-instruction IEEE / denormal-safe expansions even with
// -fno-hip-fp32-correctly-rounded-divide-sqrt; none of the quantities here is denormal or needs the last ulp
MS_DEV float rcp_f(float x) { return __builtin_amdgcn_rcpf(x); }  // callers keep |x| out of the denormal range (there the result is +-inf)
MS_DEV float rcp_safe(float x) { return __builtin_amdgcn_rcpf(fabsf(x) > 1e-30f ? x : copysignf(1e-30f, x)); }  // finite for any finite x
MS_DEV float sqrt_f(float x) { return __builtin_amdgcn_sqrtf(x); }
MS_DEV float rsq_f(float x) { return __builtin_amdgcn_rsqf(x); }
MS_DEV float norm(f3 a) { return sqrt_f(dot(a, a)); }
// |w| <= wmax (MSSIM_MAX_ANGULAR_VELOCITY)
MS_DEV f3 clamp_norm(f3 w, float wmax) {
  const float n2 = dot(w, w);
  return n2 > wmax * wmax ? w * (wmax * rsq_f(n2)) : w;
}
MS_DEV f3 normalized(f3 a) {
  const float nn = dot(a, a);
  return nn > 1e-30f ? a * rsq_f(nn) : f3{1.f, 0.f, 0.f};  // (v_rsq_f32 of a denormal is +inf)
}
MS_DEV float comp(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
// value select. `g ? a : b` on struct lvalues can be lowered to a select of ADDRESSES, which forces
// register arrays into scratch memory; selecting component by component keeps everything in VGPRs.
MS_DEV f3 sel3(bool g, f3 a, f3 b) { return f3{g ? a.x : b.x, g ? a.y : b.y, g ? a.z : b.z}; }

struct q4 {
  float w, x, y, z;
};
MS_DEV q4 qmul(q4 a, q4 b) {
  return q4{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
MS_DEV q4 qnormalized(q4 q) {
  const float nn = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  float s = nn > 1e-30f ? rsq_f(nn) : 1.f;
  return q4{q.w * s, q.x * s, q.y * s, q.z * s};
}
MS_DEV q4 qaxis_angle(f3 axis, float angle) {
  float s, c;
  sincosf(0.5f * angle, &s, &c);
  return q4{c, axis.x * s, axis.y * s, axis.z * s};
}

struct m3 {  // row major
  float m[3][3];
};
MS_DEV f3 mcol(const m3& a, int j) { return f3{a.m[0][j], a.m[1][j], a.m[2][j]}; }
MS_DEV f3 mmulv(const m3& a, f3 v) {
  return f3{a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
MS_DEV f3 mtmulv(const m3& a, f3 v) {
  return f3{a.m[0][0] * v.x + a.m[1][0] * v.y + a.m[2][0] * v.z, a.m[0][1] * v.x + a.m[1][1] * v.y + a.m[2][1] * v.z,
            a.m[0][2] * v.x + a.m[1][2] * v.y + a.m[2][2] * v.z};
}
MS_DEV m3 qmat(q4 q) {
  m3 r;
  float w = q.w, x = q.x, y = q.y, z = q.z;
  r.m[0][0] = 1.f - 2.f * (y * y + z * z); r.m[0][1] = 2.f * (x * y - w * z); r.m[0][2] = 2.f * (x * z + w * y);
  r.m[1][0] = 2.f * (x * y + w * z); r.m[1][1] = 1.f - 2.f * (x * x + z * z); r.m[1][2] = 2.f * (y * z - w * x);
  r.m[2][0] = 2.f * (x * z - w * y); r.m[2][1] = 2.f * (y * z + w * x); r.m[2][2] = 1.f - 2.f * (x * x + y * y);
  return r;
}
MS_DEV f3 qrot(q4 q, f3 v) { return mmulv(qmat(q), v); }

// symmetric 3x3 stored as xx yy zz xy xz yz
struct s3 {
  float xx, yy, zz, xy, xz, yz;
};
MS_DEV f3 smulv(const s3& a, f3 v) {
  return f3{a.xx * v.x + a.xy * v.y + a.xz * v.z, a.xy * v.x + a.yy * v.y + a.yz * v.z, a.xz * v.x + a.yz * v.y + a.zz * v.z};
}
MS_DEV s3 sadd(const s3& a, const s3& b) { return s3{a.xx + b.xx, a.yy + b.yy, a.zz + b.zz, a.xy + b.xy, a.xz + b.xz, a.yz + b.yz}; }
// R * diag-free symmetric * R^T
MS_DEV s3 srotate(const m3& R, const s3& a) {
  // T = R * A
  float t[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    t[i][0] = R.m[i][0] * a.xx + R.m[i][1] * a.xy + R.m[i][2] * a.xz;
    t[i][1] = R.m[i][0] * a.xy + R.m[i][1] * a.yy + R.m[i][2] * a.yz;
    t[i][2] = R.m[i][0] * a.xz + R.m[i][1] * a.yz + R.m[i][2] * a.zz;
  }
  s3 r;
  r.xx = t[0][0] * R.m[0][0] + t[0][1] * R.m[0][1] + t[0][2] * R.m[0][2];
  r.yy = t[1][0] * R.m[1][0] + t[1][1] * R.m[1][1] + t[1][2] * R.m[1][2];
  r.zz = t[2][0] * R.m[2][0] + t[2][1] * R.m[2][1] + t[2][2] * R.m[2][2];
  r.xy = t[0][0] * R.m[1][0] + t[0][1] * R.m[1][1] + t[0][2] * R.m[1][2];
  r.xz = t[0][0] * R.m[2][0] + t[0][1] * R.m[2][1] + t[0][2] * R.m[2][2];
  r.yz = t[1][0] * R.m[2][0] + t[1][1] * R.m[2][1] + t[1][2] * R.m[2][2];
  return r;
}
MS_DEV s3 sinverse(const s3& a) {
  float c00 = a.yy * a.zz - a.yz * a.yz;
  float c01 = a.yz * a.xz - a.xy * a.zz;
  float c02 = a.xy * a.yz - a.yy * a.xz;
  float det = a.xx * c00 + a.xy * c01 + a.xz * c02;
  float id = rcp_f(det);
  s3 r;
  r.xx = c00 * id;
  r.xy = c01 * id;
  r.xz = c02 * id;
  r.yy = (a.xx * a.zz - a.xz * a.xz) * id;
  r.yz = (a.xy * a.xz - a.xx * a.yz) * id;
  r.zz = (a.xx * a.yy - a.xy * a.xy) * id;
  return r;
}

struct pose_t {
  f3 p;
  q4 q;
};
MS_DEV pose_t pmul(pose_t a, pose_t b) { return pose_t{a.p + qrot(a.q, b.p), qnormalized(qmul(a.q, b.q))}; }
// pose from 7 consecutive floats of a constant table (wave-uniform address)
MS_DEV pose_t pose_from(const float* __restrict__ f) {
  return pose_t{f3{f[0], f[1], f[2]}, qnormalized(q4{f[3], f[4], f[5], f[6]})};
}
// pose from SoA state: item base `b`, stride N, env e
MS_DEV pose_t pose_soa(const float* __restrict__ s, int b, int N, int e) {
  const float* p = s + (size_t)b * N + e;
  return pose_t{f3{p[0], p[(size_t)N], p[2 * (size_t)N]}, qnormalized(q4{p[3 * (size_t)N], p[4 * (size_t)N], p[5 * (size_t)N], p[6 * (size_t)N]})};
}
MS_DEV void pose_store_soa(float* __restrict__ s, int b, int N, int e, pose_t P) {
  float* p = s + (size_t)b * N + e;
  p[0] = P.p.x; p[(size_t)N] = P.p.y; p[2 * (size_t)N] = P.p.z;
  p[3 * (size_t)N] = P.q.w; p[4 * (size_t)N] = P.q.x; p[5 * (size_t)N] = P.q.y; p[6 * (size_t)N] = P.q.z;
}
