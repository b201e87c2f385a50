// oracle_collide.hpp -- narrowphase of the CPU oracle (TEST INFRASTRUCTURE ONLY).
//
// Restates, from the published algorithms, the contact generation that SAPIEN/PhysX performs
// inside `px.step()` (reference call site mani_skill/envs/scene.py:374-375; shape types from
// mani_skill/utils/building/actor_builder.py:73-155; contact_offset/rest_offset from
// mani_skill/utils/structs/types.py:40-41). PhysX's own PCM/GJK/EPA source is not part of the
// reference => parity with PhysX is UNPINNED; this file pins the HIP kernels only.
//
//   plane  vs X      analytic (vertex / support depth)
//   box    vs box    SAT (15 axes) + reference-face clipping, <=4 points
//   convex vs convex Minkowski Portal Refinement on the pair, shape A inflated by the contact
//                    offset, 1 point (XenoCollide; G. Snethen, Game Programming Gems 7)
//
// Contact convention: normal n points from shape B towards shape A (pushing A along +n separates),
// sep is the signed gap (negative = penetration), x is the midpoint between the two surfaces.
#pragma once
#include "../include/mssim.h"
#include "oracle_math.hpp"

namespace orc {

enum { SH_PLANE = 0, SH_BOX = 1, SH_SPHERE = 2, SH_CAPSULE = 3, SH_CYLINDER = 4, SH_CONVEX = 5, SH_NONE = 6, SH_TRIMESH = 7 };

template <typename R>
struct Shape {
  int type;
  V3<R> c;        // world position of the shape frame
  M3<R> rot;      // world rotation of the shape frame (columns = axes)
  R param[4];
  const float* verts;  // hull vertices (shape frame), nverts of them
  int nverts;
};

template <typename R>
struct Manifold {
  int count;
  V3<R> n;
  V3<R> x[4];
  R sep[4];
};

// ---------------------------------------------------------------------------------------------
// support mapping in world space; `raw` excludes the spherical radius (sphere / capsule cores)
template <typename R>
inline V3<R> support(const Shape<R>& s, const V3<R>& d) {
  V3<R> dl = s.rot.tmul(d);
  V3<R> pl;
  switch (s.type) {
    case SH_BOX:
      pl = V3<R>(dl.x >= 0 ? s.param[0] : -s.param[0], dl.y >= 0 ? s.param[1] : -s.param[1],
                 dl.z >= 0 ? s.param[2] : -s.param[2]);
      break;
    case SH_SPHERE: {
      pl = normalized(dl) * s.param[0];
      break;
    }
    case SH_CAPSULE: {
      V3<R> u = normalized(dl) * s.param[0];
      pl = V3<R>((dl.x >= 0 ? s.param[1] : -s.param[1]) + u.x, u.y, u.z);
      break;
    }
    case SH_CYLINDER: {
      R rr = std::sqrt(dl.y * dl.y + dl.z * dl.z);
      R k = rr > R(1e-12) ? s.param[0] / rr : R(0);
      pl = V3<R>(dl.x >= 0 ? s.param[1] : -s.param[1], dl.y * k, dl.z * k);
      break;
    }
    case SH_CONVEX: {
      int best = 0;
      R bd = R(s.verts[0]) * dl.x + R(s.verts[1]) * dl.y + R(s.verts[2]) * dl.z;
      for (int i = 1; i < s.nverts; i++) {
        R v = R(s.verts[3 * i]) * dl.x + R(s.verts[3 * i + 1]) * dl.y + R(s.verts[3 * i + 2]) * dl.z;
        if (v > bd) { bd = v; best = i; }
      }
      pl = V3<R>(R(s.verts[3 * best]), R(s.verts[3 * best + 1]), R(s.verts[3 * best + 2]));
      break;
    }
    default:
      pl = V3<R>();
  }
  return s.c + s.rot * pl;
}

// keep the (up to) 4 candidates with the smallest separation, lowest index wins ties
template <typename R>
inline void keep4_deepest(int n, const V3<R>* pts, const R* seps, Manifold<R>& m) {
  bool used[64] = {false};
  m.count = 0;
  for (int k = 0; k < 4 && k < n; k++) {
    int best = -1;
    for (int i = 0; i < n; i++)
      if (!used[i] && (best < 0 || seps[i] < seps[best])) best = i;
    if (best < 0) break;
    used[best] = true;
    m.x[m.count] = pts[best];
    m.sep[m.count] = seps[best];
    m.count++;
  }
}

// ---------------------------------------------------------------------------------------------
// plane (A) vs anything (B).  Plane normal = +x axis of the plane frame.
template <typename R>
inline void collide_plane(const Shape<R>& pl, const Shape<R>& b, R offset, Manifold<R>& m) {
  m.count = 0;
  V3<R> np = pl.rot.col(0);
  m.n = -np;  // from B (the body) towards A (the plane)
  V3<R> pts[64];
  R seps[64];
  int n = 0;
  auto add = [&](const V3<R>& p, R radius) {
    R s = dot(np, p - pl.c) - radius;
    if (s < offset && n < 64) {
      pts[n] = p - np * (radius + R(0.5) * s);
      seps[n] = s;
      n++;
    }
  };
  if (b.type == SH_BOX) {
    for (int i = 0; i < 8; i++) {
      V3<R> l((i & 1) ? b.param[0] : -b.param[0], (i & 2) ? b.param[1] : -b.param[1], (i & 4) ? b.param[2] : -b.param[2]);
      add(b.c + b.rot * l, R(0));
    }
  } else if (b.type == SH_SPHERE) {
    add(b.c, b.param[0]);
  } else if (b.type == SH_CAPSULE) {
    V3<R> ax = b.rot.col(0) * b.param[1];
    add(b.c - ax, b.param[0]);
    add(b.c + ax, b.param[0]);
  } else if (b.type == SH_CONVEX) {
    for (int i = 0; i < b.nverts && i < 64; i++)
      add(b.c + b.rot * V3<R>(R(b.verts[3 * i]), R(b.verts[3 * i + 1]), R(b.verts[3 * i + 2])), R(0));
  } else {
    add(support(b, -np), R(0));
  }
  // A hull lying on a face has more than 4 vertices at (nearly) the lowest depth -- the rim of a cup, the foot of a post: which
  // four of them are "the deepest" is decided by rounding, they may all lie on one side of the rim, and the body rocks and
  // sinks. With more than 4 of the hull's vertices within MSSIM_PATCH_SLACK of its lowest one, the four are taken by EXTENT
  // among those -- the patch rule (include/mssim.h MSSIM_PATCH_*): the deepest, the farthest from it, the largest area on
  // either side of that edge, every scan taking the first candidate within the tie tolerance of the extremum. Otherwise (a
  // corner or an edge down, a box: 4 corners per face) the 4 deepest, as for every other shape.
  if (b.type == SH_CONVEX && n > 4) {
    R smin = seps[0];
    for (int i = 1; i < n; i++) smin = std::min(smin, seps[i]);
    int near_ = 0;
    for (int i = 0; i < n; i++) near_ += seps[i] <= smin + R(MSSIM_PATCH_SLACK) ? 1 : 0;
    if (near_ > 4) {
      auto cand = [&](int i) { return seps[i] <= smin + R(MSSIM_PATCH_SLACK); };
      int i0 = -1;
      for (int i = 0; i < n && i0 < 0; i++)
        if (seps[i] <= smin + R(MSSIM_PATCH_TIE_SEP)) i0 = i;
      const V3<R> p0 = pts[i0];
      auto first_near_max = [&](auto&& value, auto&& allowed, R floor_) {
        R mx = floor_;
        for (int i = 0; i < n; i++)
          if (cand(i) && allowed(i)) mx = std::max(mx, value(i));
        if (!(mx > floor_)) return -1;
        for (int i = 0; i < n; i++)
          if (cand(i) && allowed(i) && value(i) >= mx - R(MSSIM_PATCH_TIE_REL) * mx) return i;
        return -1;
      };
      const int i1 = first_near_max([&](int i) { const V3<R> d = pts[i] - p0; return dot(d, d); }, [&](int i) { return i != i0; }, R(-1));
      const V3<R> ed = pts[i1] - p0;
      auto area = [&](int i) { return dot(cross(ed, pts[i] - p0), m.n); };
      const int i2 = first_near_max([&](int i) { return std::fabs(area(i)); }, [&](int i) { return i != i0 && i != i1; }, R(-1));
      const R sgn2 = area(i2);
      const int i3 = first_near_max([&](int i) { return sgn2 >= 0 ? -area(i) : area(i); }, [&](int i) { return i != i0 && i != i1 && i != i2; }, R(MSSIM_PATCH_TIE_REL) * std::fabs(sgn2));
      const int pick[4] = {i0, i1, i2, i3};
      m.count = 0;
      for (int k = 0; k < 4; k++)
        if (pick[k] >= 0) { m.x[m.count] = pts[pick[k]]; m.sep[m.count] = seps[pick[k]]; m.count++; }
      return;
    }
  }
  keep4_deepest(n, pts, seps, m);
}

// ---------------------------------------------------------------------------------------------
// box vs box: SAT + clipping
template <typename R>
inline int clip_poly(int n, const V3<R>* in, V3<R>* out, const V3<R>& pn, R pd) {
  // keep the part with dot(pn, p) <= pd   (Sutherland-Hodgman, one plane)
  int m = 0;
  for (int i = 0; i < n; i++) {
    const V3<R>& a = in[i];
    const V3<R>& b = in[(i + 1) % n];
    R da = dot(pn, a) - pd, db = dot(pn, b) - pd;
    if (da <= 0) out[m++] = a;
    if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
      R t = da / (da - db);
      out[m++] = a + (b - a) * t;
    }
  }
  return m;
}

template <typename R>
inline void reduce4(int n, const V3<R>* pts, const R* seps, const V3<R>& nrm, Manifold<R>& m) {
  if (n <= 4) {
    m.count = n;
    for (int i = 0; i < n; i++) { m.x[i] = pts[i]; m.sep[i] = seps[i]; }
    return;
  }
  int i0 = 0;
  for (int i = 1; i < n; i++) if (seps[i] < seps[i0]) i0 = i;
  int i1 = -1; R best = R(-1);
  for (int i = 0; i < n; i++) {
    if (i == i0) continue;
    V3<R> d = pts[i] - pts[i0]; R v = dot(d, d);
    if (v > best) { best = v; i1 = i; }
  }
  V3<R> e = pts[i1] - pts[i0];
  int i2 = -1; best = R(-1); R sgn2 = R(0);
  for (int i = 0; i < n; i++) {
    if (i == i0 || i == i1) continue;
    R a = dot(cross(e, pts[i] - pts[i0]), nrm);
    if (std::fabs(a) > best) { best = std::fabs(a); i2 = i; sgn2 = a; }
  }
  int i3 = -1; best = R(0);
  for (int i = 0; i < n; i++) {
    if (i == i0 || i == i1 || i == i2) continue;
    R a = dot(cross(e, pts[i] - pts[i0]), nrm);
    R v = sgn2 >= 0 ? -a : a;  // opposite side of the line (i0,i1) from i2
    if (v > best) { best = v; i3 = i; }
  }
  int idx[4] = {i0, i1, i2, i3};
  m.count = 0;
  for (int k = 0; k < 4; k++)
    if (idx[k] >= 0) { m.x[m.count] = pts[idx[k]]; m.sep[m.count] = seps[idx[k]]; m.count++; }
}

template <typename R>
inline void collide_box_box(const Shape<R>& A, const Shape<R>& B, R offset, Manifold<R>& m) {
  m.count = 0;
  const R eps = R(1e-6);
  V3<R> a[3] = {A.rot.col(0), A.rot.col(1), A.rot.col(2)};
  V3<R> b[3] = {B.rot.col(0), B.rot.col(1), B.rot.col(2)};
  R hA[3] = {A.param[0], A.param[1], A.param[2]}, hB[3] = {B.param[0], B.param[1], B.param[2]};
  V3<R> tw = B.c - A.c;
  R T[3] = {dot(tw, a[0]), dot(tw, a[1]), dot(tw, a[2])};
  R Rm[3][3], Ra[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { Rm[i][j] = dot(a[i], b[j]); Ra[i][j] = std::fabs(Rm[i][j]) + eps; }
  // face axes of A
  R sA = R(-1e30); int iA = 0;
  for (int i = 0; i < 3; i++) {
    R s = std::fabs(T[i]) - (hA[i] + hB[0] * Ra[i][0] + hB[1] * Ra[i][1] + hB[2] * Ra[i][2]);
    if (s > sA) { sA = s; iA = i; }
  }
  R sB = R(-1e30); int iB = 0;
  for (int j = 0; j < 3; j++) {
    R tb = T[0] * Rm[0][j] + T[1] * Rm[1][j] + T[2] * Rm[2][j];
    R s = std::fabs(tb) - (hB[j] + hA[0] * Ra[0][j] + hA[1] * Ra[1][j] + hA[2] * Ra[2][j]);
    if (s > sB) { sB = s; iB = j; }
  }
  R sE = R(-1e30); int eI = -1, eJ = -1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      R l2 = R(1) - Rm[i][j] * Rm[i][j];
      if (l2 < R(1e-6)) continue;
      int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      R ra = hA[i1] * Ra[i2][j] + hA[i2] * Ra[i1][j];
      R rb = hB[j1] * Ra[i][j2] + hB[j2] * Ra[i][j1];
      R s = (std::fabs(T[i2] * Rm[i1][j] - T[i1] * Rm[i2][j]) - (ra + rb)) / std::sqrt(l2);
      if (s > sE) { sE = s; eI = i; eJ = j; }
    }
  R sF = sA > sB ? sA : sB;
  R smax = sF > sE ? sF : sE;
  if (smax > offset) return;

  if (eI >= 0 && sE > sF + R(1e-3)) {
    // ---- edge-edge ----
    V3<R> ax = normalized(cross(a[eI], b[eJ]));
    if (dot(ax, tw) < 0) ax = -ax;  // from A to B
    V3<R> pa = A.c, pb = B.c;
    for (int k = 0; k < 3; k++) {
      if (k != eI) pa += a[k] * (dot(ax, a[k]) >= 0 ? hA[k] : -hA[k]);
      if (k != eJ) pb -= b[k] * (dot(ax, b[k]) >= 0 ? hB[k] : -hB[k]);
    }
    // closest points of the two edge lines, clamped to the edge extents
    V3<R> u = a[eI], v = b[eJ], w0 = pa - pb;
    R uv = dot(u, v), uw = dot(u, w0), vw = dot(v, w0);
    R den = R(1) - uv * uv;
    R sa = den > R(1e-9) ? (uv * vw - uw) / den : R(0);
    R sb = den > R(1e-9) ? (vw - uv * uw) / den : R(0);
    sa = sa > hA[eI] ? hA[eI] : (sa < -hA[eI] ? -hA[eI] : sa);
    sb = sb > hB[eJ] ? hB[eJ] : (sb < -hB[eJ] ? -hB[eJ] : sb);
    V3<R> qa = pa + u * sa, qb = pb + v * sb;
    m.count = 1;
    m.n = -ax;
    m.sep[0] = dot(qb - qa, ax);
    m.x[0] = (qa + qb) * R(0.5);
    return;
  }
  // ---- face contact: reference box X, incident box Y ----
  bool refA = sA >= sB - R(1e-5);
  const Shape<R>& X = refA ? A : B;
  const Shape<R>& Y = refA ? B : A;
  const V3<R>* xa = refA ? a : b;
  const V3<R>* ya = refA ? b : a;
  const R* hX = refA ? hA : hB;
  const R* hY = refA ? hB : hA;
  int ir = refA ? iA : iB;
  V3<R> txy = Y.c - X.c;
  V3<R> nref = dot(xa[ir], txy) >= 0 ? xa[ir] : -xa[ir];  // from X towards Y
  // incident face: most anti-parallel face of Y
  int jinc = 0; R bestd = R(-1);
  for (int j = 0; j < 3; j++) {
    R d = std::fabs(dot(nref, ya[j]));
    if (d > bestd) { bestd = d; jinc = j; }
  }
  V3<R> ninc = dot(nref, ya[jinc]) > 0 ? -ya[jinc] : ya[jinc];
  int j1 = (jinc + 1) % 3, j2 = (jinc + 2) % 3;
  V3<R> fc = Y.c + ninc * hY[jinc];
  V3<R> poly[16], tmp[16];
  poly[0] = fc + ya[j1] * hY[j1] + ya[j2] * hY[j2];
  poly[1] = fc - ya[j1] * hY[j1] + ya[j2] * hY[j2];
  poly[2] = fc - ya[j1] * hY[j1] - ya[j2] * hY[j2];
  poly[3] = fc + ya[j1] * hY[j1] - ya[j2] * hY[j2];
  int np = 4;
  int r1 = (ir + 1) % 3, r2 = (ir + 2) % 3;
  np = clip_poly(np, poly, tmp, xa[r1], dot(xa[r1], X.c) + hX[r1]);
  np = clip_poly(np, tmp, poly, -xa[r1], -dot(xa[r1], X.c) + hX[r1]);
  np = clip_poly(np, poly, tmp, xa[r2], dot(xa[r2], X.c) + hX[r2]);
  np = clip_poly(np, tmp, poly, -xa[r2], -dot(xa[r2], X.c) + hX[r2]);
  V3<R> pts[16];
  R seps[16];
  int n = 0;
  for (int i = 0; i < np; i++) {
    R s = dot(poly[i] - X.c, nref) - hX[ir];
    if (s <= offset) { pts[n] = poly[i] - nref * (R(0.5) * s); seps[n] = s; n++; }
  }
  m.n = refA ? -nref : nref;
  reduce4(n, pts, seps, nref, m);
}

// ---------------------------------------------------------------------------------------------
// generic convex pair: MPR on the Minkowski difference A' (-) B with A' = A inflated by `margin`
template <typename R>
struct MVert {
  V3<R> v, a, b;  // v = a' - b ; a = raw support of A (without inflation), b = support of B
};
template <typename R>
inline MVert<R> msupport(const Shape<R>& A, const Shape<R>& B, const V3<R>& d, R margin) {
  MVert<R> r;
  V3<R> dn = normalized(d);
  r.a = support(A, dn);
  r.b = support(B, -dn);
  r.v = r.a + dn * margin - r.b;
  return r;
}

// closest point to the origin on triangle (p0,p1,p2): barycentric weights (Ericson, RTCD 5.1.5)
template <typename R>
inline void closest_on_triangle(const V3<R>& a, const V3<R>& b, const V3<R>& c, R w[3]) {
  V3<R> ab = b - a, ac = c - a, ap = -a;
  R d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0 && d2 <= 0) { w[0] = 1; w[1] = 0; w[2] = 0; return; }
  V3<R> bp = -b;
  R d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0 && d4 <= d3) { w[0] = 0; w[1] = 1; w[2] = 0; return; }
  R vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { R v = d1 / (d1 - d3); w[0] = 1 - v; w[1] = v; w[2] = 0; return; }
  V3<R> cp = -c;
  R d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0 && d5 <= d6) { w[0] = 0; w[1] = 0; w[2] = 1; return; }
  R vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { R v = d2 / (d2 - d6); w[0] = 1 - v; w[1] = 0; w[2] = v; return; }
  R va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    R v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    w[0] = 0; w[1] = 1 - v; w[2] = v; return;
  }
  R den = R(1) / (va + vb + vc);
  w[1] = vb * den; w[2] = vc * den; w[0] = 1 - w[1] - w[2];
}

template <typename R>
inline void collide_mpr(const Shape<R>& A, const Shape<R>& B, R offset, Manifold<R>& m, const V3<R>* b_inside = nullptr) {
  // `b_inside`: a point of B to take instead of its frame origin as B's part of the interior point A.c - B.c (a triangle
  // of a mesh: the point of the triangle nearest to A's centre, so that the origin ray runs along the contact normal)
  m.count = 0;
  const R margin = offset;
  const R tol = R(MSSIM_MPR_TOLERANCE);
  MVert<R> v0, v1, v2, v3, v4;
  v0.a = A.c; v0.b = b_inside ? *b_inside : B.c; v0.v = A.c - v0.b;
  if (dot(v0.v, v0.v) < R(1e-12)) v0.v = V3<R>(R(1e-5), 0, 0);
  V3<R> dir = -v0.v;
  v1 = msupport(A, B, dir, margin);
  if (dot(v1.v, dir) <= 0) return;
  dir = cross(v1.v, v0.v);
  if (dot(dir, dir) < R(1e-14)) {
    // origin lies on the ray v0->v1: v1 is the surface point in the centre direction
    V3<R> w = v1.v; R D = norm(w);
    m.count = 1;
    m.n = D > R(1e-9) ? w * (R(-1) / D) : normalized(v0.v);
    m.sep[0] = margin - D;
    m.x[0] = (v1.a + v1.b) * R(0.5);
    return;
  }
  v2 = msupport(A, B, dir, margin);
  if (dot(v2.v, dir) <= 0) return;
  dir = cross(v1.v - v0.v, v2.v - v0.v);
  if (dot(dir, v0.v) > 0) { MVert<R> t = v1; v1 = v2; v2 = t; dir = -dir; }
  // portal discovery
  bool found = false;
  for (int it = 0; it < 32; it++) {
    v3 = msupport(A, B, dir, margin);
    if (dot(v3.v, dir) <= 0) return;
    if (dot(cross(v1.v, v3.v), v0.v) < 0) { v2 = v3; dir = cross(v1.v - v0.v, v3.v - v0.v); continue; }
    if (dot(cross(v3.v, v2.v), v0.v) < 0) { v1 = v3; dir = cross(v3.v - v0.v, v2.v - v0.v); continue; }
    found = true;
    break;
  }
  if (!found) return;
  // portal refinement
  bool hit = false;
  for (int it = 0; it < 48; it++) {
    dir = cross(v2.v - v1.v, v3.v - v1.v);
    R dl = norm(dir);
    if (dl < R(1e-14)) break;  // degenerate portal: use what we have
    dir = dir * (R(1) / dl);
    if (dot(dir, v1.v) >= 0) hit = true;
    v4 = msupport(A, B, dir, margin);
    R reach = dot(v4.v, dir);
    if (reach < 0 && !hit) return;  // origin is beyond the support plane
    if (reach - dot(v3.v, dir) <= tol || it == 47) {
      if (!hit) return;
      break;
    }
    V3<R> cr = cross(v4.v, v0.v);
    if (dot(v1.v, cr) > 0) {
      if (dot(v2.v, cr) > 0) v1 = v4; else v3 = v4;
    } else {
      if (dot(v3.v, cr) > 0) v2 = v4; else v1 = v4;
    }
  }
  if (!hit) return;
  R w[3];
  closest_on_triangle(v1.v, v2.v, v3.v, w);
  V3<R> wp = v1.v * w[0] + v2.v * w[1] + v3.v * w[2];
  R D = norm(wp);
  V3<R> pn = normalized(cross(v2.v - v1.v, v3.v - v1.v));
  m.count = 1;
  m.n = D > R(1e-7) ? wp * (R(-1) / D) : -pn;
  m.sep[0] = margin - D;
  V3<R> pa = v1.a * w[0] + v2.a * w[1] + v3.a * w[2];
  V3<R> pb = v1.b * w[0] + v2.b * w[1] + v3.b * w[2];
  m.x[0] = (pa + pb) * R(0.5);
}

template <typename R>
inline void collide(const Shape<R>& A, const Shape<R>& B, R offset, Manifold<R>& m) {
  if (A.type == SH_PLANE) collide_plane(A, B, offset, m);
  else if (A.type == SH_BOX && B.type == SH_BOX) collide_box_box(A, B, offset, m);
  else collide_mpr(A, B, offset, m);
}

}  // namespace orc
