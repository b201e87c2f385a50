// mssim_collide.h -- per-lane narrowphase for gfx950 (one (env, shape pair) per lane).
//
// Replaces the contact generation hidden in `px.step()` (mani_skill/envs/scene.py:374-375; shape
// types mani_skill/utils/building/actor_builder.py:73-155; contact_offset / rest_offset
// mani_skill/utils/structs/types.py:40-41).
//
// Launch shape (see k_narrow): blockIdx.y = pair, so all 64 lanes of a wave work on the SAME
// shape pair -- shape types, sizes and hull vertex addresses are wave-uniform (SGPR / scalar
// loads), only poses differ per lane. Variable-length per-lane lists (clip polygons) live in
// LDS as [slot][lane] so a lane-varying slot index is conflict-free (bank = lane % 32).
//
//   plane  vs X      analytic, 4 deepest vertices
//   box    vs box    SAT over 15 axes + reference-face clipping, <= 4 points
//   convex vs convex Minkowski Portal Refinement, shape A inflated by the contact offset, 1 point
//
// Contact convention: n points from shape B to shape A, sep = signed gap, x = mid point.
#pragma once
#include "mssim_dev.h"

enum { SH_PLANE = 0, SH_BOX = 1, SH_SPHERE = 2, SH_CAPSULE = 3, SH_CYLINDER = 4, SH_CONVEX = 5, SH_NONE = 6, SH_TRIMESH = 7 };

struct shape_t {
  int type;  // wave-uniform
  f3 c;
  m3 rot;
  float p0, p1, p2;                 // params
  const float* __restrict__ verts;  // wave-uniform
  int nverts;                       // wave-uniform
};

struct manifold_t {
  int count;
  f3 n;
  f3 x[4];
  float sep[4];
};

MS_DEV void manifold_clear(manifold_t& m) {
  m.count = 0;
  m.n = f3{0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; k++) { m.x[k] = f3{0.f, 0.f, 0.f}; m.sep[k] = 0.f; }
}

MS_DEV f3 support(const shape_t& s, f3 d) {
  f3 dl = mtmulv(s.rot, d);
  f3 pl;
  switch (s.type) {
    case SH_BOX:
      pl = f3{dl.x >= 0.f ? s.p0 : -s.p0, dl.y >= 0.f ? s.p1 : -s.p1, dl.z >= 0.f ? s.p2 : -s.p2};
      break;
    case SH_SPHERE:
      pl = normalized(dl) * s.p0;
      break;
    case SH_CAPSULE: {
      f3 u = normalized(dl) * s.p0;
      pl = f3{(dl.x >= 0.f ? s.p1 : -s.p1) + u.x, u.y, u.z};
      break;
    }
    case SH_CYLINDER: {
      float rr = sqrt_f(dl.y * dl.y + dl.z * dl.z);
      float k = rr > 1e-12f ? s.p0 * rcp_f(rr) : 0.f;
      pl = f3{dl.x >= 0.f ? s.p1 : -s.p1, dl.y * k, dl.z * k};
      break;
    }
    case SH_CONVEX: {
      // batches of 8 vertices read as six aligned 16-byte loads (the host pads every hull to a
      // multiple of 8 vertices with copies of vertex 0 and aligns its start; a copy of vertex 0 can
      // never win the strict comparison, so the result equals the plain first-maximum scan)
      const float* __restrict__ v = s.verts;
      float bx = v[0], by = v[1], bz = v[2];
      float bd = bx * dl.x + by * dl.y + bz * dl.z;
      for (int i0 = 0; i0 < s.nverts; i0 += 8) {
        const float4* __restrict__ q = reinterpret_cast<const float4*>(v + 3 * i0);
        float f[24];
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const float4 t = q[k];
          f[4 * k] = t.x; f[4 * k + 1] = t.y; f[4 * k + 2] = t.z; f[4 * k + 3] = t.w;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
          float t = f[3 * k] * dl.x + f[3 * k + 1] * dl.y + f[3 * k + 2] * dl.z;
          bool g = t > bd;
          bd = g ? t : bd;
          bx = g ? f[3 * k] : bx; by = g ? f[3 * k + 1] : by; bz = g ? f[3 * k + 2] : bz;
        }
      }
      pl = f3{bx, by, bz};
      break;
    }
    default:
      pl = f3{0.f, 0.f, 0.f};
  }
  return s.c + mmulv(s.rot, pl);
}

// streaming "4 smallest separations" (stable: earlier index wins ties)
MS_DEV void keep4_insert(manifold_t& m, f3 p, float s) {
  // find insertion position: after all entries with sep <= s
  int pos = m.count;
#pragma unroll
  for (int k = 3; k >= 0; k--)
    if (k < m.count && s < m.sep[k]) pos = k;
  if (pos >= 4) return;
#pragma unroll
  for (int k = 3; k > 0; k--) {  // value selects: conditional struct stores would put `m` into scratch memory
    const bool g = k > pos && k <= m.count;
    m.x[k] = sel3(g, m.x[k - 1], m.x[k]); m.sep[k] = g ? m.sep[k - 1] : m.sep[k];
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const bool g = k == pos;
    m.x[k] = sel3(g, p, m.x[k]); m.sep[k] = g ? s : m.sep[k];
  }
  m.count = m.count < 4 ? m.count + 1 : 4;
}

MS_DEV void collide_plane(const shape_t& pl, const shape_t& b, float offset, manifold_t& m) {
  manifold_clear(m);
  f3 np = mcol(pl.rot, 0);
  m.n = -np;
  auto add = [&](f3 p, float radius) {
    float s = dot(np, p - pl.c) - radius;
    if (s < offset) keep4_insert(m, p - np * (radius + 0.5f * s), s);
  };
  if (b.type == SH_BOX) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      f3 l = f3{(i & 1) ? b.p0 : -b.p0, (i & 2) ? b.p1 : -b.p1, (i & 4) ? b.p2 : -b.p2};
      add(b.c + mmulv(b.rot, l), 0.f);
    }
  } else if (b.type == SH_SPHERE) {
    add(b.c, b.p0);
  } else if (b.type == SH_CAPSULE) {
    f3 ax = mcol(b.rot, 0) * b.p1;
    add(b.c - ax, b.p0);
    add(b.c + ax, b.p0);
  } else if (b.type == SH_CONVEX) {
    // (batches of 8 vertices as six aligned 16-byte loads, as in support(): a vertex at a time is three dependent trips to
    // L2 for the one lane that does this pair -- the Fetch's base hulls over the ground plane cost 10 % of its control step)
    const int nv = b.nverts < 64 ? b.nverts : 64;
    for (int i0 = 0; i0 < nv; i0 += 8) {
      const float4* __restrict__ q = reinterpret_cast<const float4*>(b.verts + 3 * i0);
      float f[24];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const float4 t = q[k];
        f[4 * k] = t.x; f[4 * k + 1] = t.y; f[4 * k + 2] = t.z; f[4 * k + 3] = t.w;
      }
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (i0 + k < nv) add(b.c + mmulv(b.rot, f3{f[3 * k], f[3 * k + 1], f[3 * k + 2]}), 0.f);
    }
  } else {
    add(support(b, -np), 0.f);
  }
}

// ------------------------------------------------------------------------------------------
// box-box.  LDS scratch: 2 polygon buffers of 8 points (x,y,z) + 8 separations -> 56 slots,
// [slot][64 lanes] (a quad clipped by 4 planes has at most 8 vertices)
#define CLIP_SLOTS 56
// STRIDE = distance (floats) between consecutive slots of one lane: 64 for [slot][64 lanes]
// (k_narrow), 16 when the scratch of a 16-lane group lives inside its env's LDS area (k_solve16 fused)
template <int STRIDE>
struct lds_poly {
  float* base;  // already offset by lane
  int buf;
  MS_DEV f3 get(int i) const { float* p = base + (size_t)(buf * 24 + 3 * i) * STRIDE; return f3{p[0], p[STRIDE], p[2 * STRIDE]}; }
  MS_DEV void put(int i, f3 v) { float* p = base + (size_t)(buf * 24 + 3 * i) * STRIDE; p[0] = v.x; p[STRIDE] = v.y; p[2 * STRIDE] = v.z; }
};

// One Sutherland-Hodgman pass. All 8 input slots are read with static indices first (one LDS
// latency, values in registers; slots >= n are stale and masked), only the appends use a dynamic
// slot. Same arithmetic and vertex order as the vertex-by-vertex loop.
template <int STRIDE>
MS_DEV int clip_poly(float* lds, int src, int n, f3 pn, float pd) {
  lds_poly<STRIDE> in{lds, src}, out{lds, src ^ 1};
  f3 P[8];
  float d[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { P[i] = in.get(i); d[i] = dot(pn, P[i]) - pd; }
  int m = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (i < n) {
      const bool wrap = !(i + 1 < n);
      const f3 a = P[i];
      const f3 b = sel3(i == 7 || wrap, P[0], P[(i + 1) & 7]);
      const float da = d[i], db = (i == 7 || wrap) ? d[0] : d[(i + 1) & 7];
      if (da <= 0.f && m < 8) out.put(m++, a);
      if (((da < 0.f && db > 0.f) || (da > 0.f && db < 0.f)) && m < 8) {
        float t = da * rcp_safe(da - db);
        out.put(m++, a + (b - a) * t);
      }
    }
  }
  return m;
}

// separating-axis search + the single-point edge-edge case + the choice of reference / incident face.
// Returns 0 = apart, 1 = edge-edge contact (manifold filled), 2 = face contact (F filled, clipping is
// the caller's: one lane per pair in collide_box_box, 16 lanes per pair in collide_box_box_coop)
struct bb_face_t {
  f3 nref, Xc, x1, x2, fc, y1, y2;
  float hXr, hX1, hX2, hY1, hY2;
  bool refA;
};
MS_DEV int bb_setup(const shape_t& A, const shape_t& B, float offset, manifold_t& m, bb_face_t& F) {
  manifold_clear(m);
  const float eps = 1e-6f;
  f3 a[3] = {mcol(A.rot, 0), mcol(A.rot, 1), mcol(A.rot, 2)};
  f3 b[3] = {mcol(B.rot, 0), mcol(B.rot, 1), mcol(B.rot, 2)};
  float hA[3] = {A.p0, A.p1, A.p2}, hB[3] = {B.p0, B.p1, B.p2};
  f3 tw = B.c - A.c;
  float T[3] = {dot(tw, a[0]), dot(tw, a[1]), dot(tw, a[2])};
  float Rm[3][3], Ra[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) { Rm[i][j] = dot(a[i], b[j]); Ra[i][j] = fabsf(Rm[i][j]) + eps; }
  float sA = -1e30f; int iA = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    float s = fabsf(T[i]) - (hA[i] + hB[0] * Ra[i][0] + hB[1] * Ra[i][1] + hB[2] * Ra[i][2]);
    if (s > sA) { sA = s; iA = i; }
  }
  float sB = -1e30f; int iB = 0;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    float tb = T[0] * Rm[0][j] + T[1] * Rm[1][j] + T[2] * Rm[2][j];
    float s = fabsf(tb) - (hB[j] + hA[0] * Ra[0][j] + hA[1] * Ra[1][j] + hA[2] * Ra[2][j]);
    if (s > sB) { sB = s; iB = j; }
  }
  float sE = -1e30f; int eI = -1, eJ = -1;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float l2 = 1.f - Rm[i][j] * Rm[i][j];
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      float ra = hA[i1] * Ra[i2][j] + hA[i2] * Ra[i1][j];
      float rb = hB[j1] * Ra[i][j2] + hB[j2] * Ra[i][j1];
      float s = (fabsf(T[i2] * Rm[i1][j] - T[i1] * Rm[i2][j]) - (ra + rb)) * rsq_f(fmaxf(l2, 1e-12f));
      if (l2 >= 1e-6f && s > sE) { sE = s; eI = i; eJ = j; }
    }
  float sF = sA > sB ? sA : sB;
  float smax = sF > sE ? sF : sE;
  if (smax > offset) return 0;

  if (eI >= 0 && sE > sF + 1e-3f) {
    f3 aE = sel3(eI == 0, a[0], sel3(eI == 1, a[1], a[2]));
    f3 bE = sel3(eJ == 0, b[0], sel3(eJ == 1, b[1], b[2]));
    f3 ax = normalized(cross(aE, bE));
    ax = sel3(dot(ax, tw) < 0.f, -ax, ax);
    f3 pa = A.c, pb = B.c;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (k != eI) pa += a[k] * (dot(ax, a[k]) >= 0.f ? hA[k] : -hA[k]);
      if (k != eJ) pb -= b[k] * (dot(ax, b[k]) >= 0.f ? hB[k] : -hB[k]);
    }
    float hAe = eI == 0 ? hA[0] : (eI == 1 ? hA[1] : hA[2]);
    float hBe = eJ == 0 ? hB[0] : (eJ == 1 ? hB[1] : hB[2]);
    f3 w0 = pa - pb;
    float uv = dot(aE, bE), uw = dot(aE, w0), vw = dot(bE, w0);
    float den = 1.f - uv * uv;
    float sa = den > 1e-9f ? (uv * vw - uw) * rcp_f(den) : 0.f;
    float sb = den > 1e-9f ? (vw - uv * uw) * rcp_f(den) : 0.f;
    sa = fminf(fmaxf(sa, -hAe), hAe);
    sb = fminf(fmaxf(sb, -hBe), hBe);
    f3 qa = pa + aE * sa, qb = pb + bE * sb;
    m.count = 1;
    m.n = -ax;
    m.sep[0] = dot(qb - qa, ax);
    m.x[0] = (qa + qb) * 0.5f;
    return 1;
  }
  bool refA = sA >= sB - 1e-5f;
  // select reference (X) / incident (Y) data without dynamic register indexing
  f3 xa[3], ya[3];
  float hX[3], hY[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    xa[k] = sel3(refA, a[k], b[k]);
    ya[k] = sel3(refA, b[k], a[k]);
    hX[k] = refA ? hA[k] : hB[k];
    hY[k] = refA ? hB[k] : hA[k];
  }
  f3 Xc = sel3(refA, A.c, B.c), Yc = sel3(refA, B.c, A.c);
  int ir = refA ? iA : iB;
  f3 xr = sel3(ir == 0, xa[0], sel3(ir == 1, xa[1], xa[2]));
  float hXr = ir == 0 ? hX[0] : (ir == 1 ? hX[1] : hX[2]);
  f3 nref = sel3(dot(xr, Yc - Xc) >= 0.f, xr, -xr);
  int jinc = 0;
  float bestd = -1.f;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    float d = fabsf(dot(nref, ya[j]));
    if (d > bestd) { bestd = d; jinc = j; }
  }
  f3 yi = sel3(jinc == 0, ya[0], sel3(jinc == 1, ya[1], ya[2]));
  f3 y1 = sel3(jinc == 0, ya[1], sel3(jinc == 1, ya[2], ya[0]));
  f3 y2 = sel3(jinc == 0, ya[2], sel3(jinc == 1, ya[0], ya[1]));
  float hYi = jinc == 0 ? hY[0] : (jinc == 1 ? hY[1] : hY[2]);
  float hY1 = jinc == 0 ? hY[1] : (jinc == 1 ? hY[2] : hY[0]);
  float hY2 = jinc == 0 ? hY[2] : (jinc == 1 ? hY[0] : hY[1]);
  f3 ninc = sel3(dot(nref, yi) > 0.f, -yi, yi);
  f3 fc = Yc + ninc * hYi;
  F.x1 = sel3(ir == 0, xa[1], sel3(ir == 1, xa[2], xa[0]));
  F.x2 = sel3(ir == 0, xa[2], sel3(ir == 1, xa[0], xa[1]));
  F.hX1 = ir == 0 ? hX[1] : (ir == 1 ? hX[2] : hX[0]);
  F.hX2 = ir == 0 ? hX[2] : (ir == 1 ? hX[0] : hX[1]);
  F.nref = nref; F.Xc = Xc; F.fc = fc; F.y1 = y1; F.y2 = y2;
  F.hXr = hXr; F.hY1 = hY1; F.hY2 = hY2; F.refA = refA;
  return 2;
}

template <int STRIDE = 64>
MS_DEV void collide_box_box(const shape_t& A, const shape_t& B, float offset, manifold_t& m, float* lds) {
  bb_face_t F;
  if (bb_setup(A, B, offset, m, F) != 2) return;
  const f3 nref = F.nref, Xc = F.Xc, fc = F.fc, y1 = F.y1, y2 = F.y2;
  const float hXr = F.hXr, hY1 = F.hY1, hY2 = F.hY2;
  const bool refA = F.refA;
  lds_poly<STRIDE> P{lds, 0};
  P.put(0, fc + y1 * hY1 + y2 * hY2);
  P.put(1, fc - y1 * hY1 + y2 * hY2);
  P.put(2, fc - y1 * hY1 - y2 * hY2);
  P.put(3, fc + y1 * hY1 - y2 * hY2);
  const f3 x1 = F.x1, x2 = F.x2;
  const float hX1 = F.hX1, hX2 = F.hX2;
  int np = 4;
  np = clip_poly<STRIDE>(lds, 0, np, x1, dot(x1, Xc) + hX1);
  np = clip_poly<STRIDE>(lds, 1, np, -x1, -dot(x1, Xc) + hX1);
  np = clip_poly<STRIDE>(lds, 0, np, x2, dot(x2, Xc) + hX2);
  np = clip_poly<STRIDE>(lds, 1, np, -x2, -dot(x2, Xc) + hX2);
  // result polygon is in buffer 0; compact the points within the offset into buffer 1 (x,y,z) and
  // their separations into the tail slots of buffer 1
  lds_poly<STRIDE> Q{lds, 0}, Rb{lds, 1};
  float* seps = lds + (size_t)48 * STRIDE;  // slots 48..55 of the scratch
  int n = 0;
  {
    f3 Qp[8];
#pragma unroll
    for (int i = 0; i < 8; i++) Qp[i] = Q.get(i);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (i < np) {
        f3 p = Qp[i];
        float s = dot(p - Xc, nref) - hXr;
        if (s <= offset) { Rb.put(n, p - nref * (0.5f * s)); seps[(size_t)n * STRIDE] = s; n++; }
      }
    }
  }
  m.n = refA ? -nref : nref;
  if (n <= 4) {
    m.count = n;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const f3 pk = Rb.get(k);
      const float sk = seps[(size_t)k * STRIDE];
      m.x[k] = sel3(k < n, pk, f3{0.f, 0.f, 0.f}); m.sep[k] = k < n ? sk : 0.f;
    }
    return;
  }
  // more than 4 candidates: deepest, farthest from it, then the two of largest area on either side
  // (candidates in registers; "first extremum wins" as in the sequential scans)
  f3 R[8];
  float sp[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { R[i] = Rb.get(i); sp[i] = seps[(size_t)i * STRIDE]; }
  int i0 = 0;
  float s0 = sp[0];
  f3 p0 = R[0];
#pragma unroll
  for (int i = 1; i < 8; i++) {
    const bool g = i < n && sp[i] < s0;
    s0 = g ? sp[i] : s0; i0 = g ? i : i0; p0 = sel3(g, R[i], p0);
  }
  int i1 = -1; float best = -1.f;
  f3 p1 = p0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    {
      f3 d = R[i] - p0; float v = dot(d, d);
      const bool g = i < n && i != i0 && v > best;
      best = g ? v : best; i1 = g ? i : i1; p1 = sel3(g, R[i], p1);
    }
  }
  f3 e = p1 - p0;
  int i2 = -1; best = -1.f; float sgn2 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (i < n && i != i0 && i != i1) {
      float ar = dot(cross(e, R[i] - p0), nref);
      if (fabsf(ar) > best) { best = fabsf(ar); i2 = i; sgn2 = ar; }
    }
  }
  int i3 = -1; best = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (i < n && i != i0 && i != i1 && i != i2) {
      float ar = dot(cross(e, R[i] - p0), nref);
      float v = sgn2 >= 0.f ? -ar : ar;
      if (v > best) { best = v; i3 = i; }
    }
  }
  int idx[4] = {i0, i1, i2, i3};
  m.count = 0;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (idx[k] >= 0) {
      f3 p = Rb.get(idx[k]);
      float s = seps[(size_t)idx[k] * STRIDE];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const bool g = t == m.count;
        m.x[t] = sel3(g, p, m.x[t]); m.sep[t] = g ? s : m.sep[t];
      }
      m.count++;
    }
}

// ------------------------------------------------------------------------------------------
// MPR on A' (-) B, A' = A inflated by `margin`
struct mvert {
  f3 v, a, b;
};
// support-function policy of the MPR: SupDirect evaluates `support()` per lane; the fused step kernel
// plugs in a 16-lane cooperative hull scan (mssim_solve16.h) with the same first-maximum result
struct SupDirect {
  MS_DEV f3 operator()(int /*which*/, const shape_t& s, f3 d) const { return support(s, d); }
};
template <class SUP>
MS_DEV mvert msupport(const SUP& sup, const shape_t& A, const shape_t& B, f3 d, float margin) {
  mvert r;
  f3 dn = normalized(d);
  r.a = sup(0, A, dn);
  r.b = sup(1, B, -dn);
  r.v = r.a + dn * margin - r.b;
  return r;
}

MS_DEV void closest_on_triangle(f3 a, f3 b, f3 c, float w[3]) {
  f3 ab = b - a, ac = c - a, ap = -a;
  float d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.f && d2 <= 0.f) { w[0] = 1.f; w[1] = 0.f; w[2] = 0.f; return; }
  f3 bp = -b;
  float d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0.f && d4 <= d3) { w[0] = 0.f; w[1] = 1.f; w[2] = 0.f; return; }
  float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { float v = d1 * rcp_f(fmaxf(d1 - d3, 1e-30f)); w[0] = 1.f - v; w[1] = v; w[2] = 0.f; return; }
  f3 cp = -c;
  float d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0.f && d5 <= d6) { w[0] = 0.f; w[1] = 0.f; w[2] = 1.f; return; }
  float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { float v = d2 * rcp_f(fmaxf(d2 - d6, 1e-30f)); w[0] = 1.f - v; w[1] = 0.f; w[2] = v; return; }
  float va = d3 * d6 - d5 * d4;
  if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
    float v = (d4 - d3) * rcp_f(fmaxf((d4 - d3) + (d5 - d6), 1e-30f));
    w[0] = 0.f; w[1] = 1.f - v; w[2] = v; return;
  }
  float den = rcp_f(fmaxf(va + vb + vc, 1e-30f));
  w[1] = vb * den; w[2] = vc * den; w[0] = 1.f - w[1] - w[2];
}

#ifdef MSSIM_PHASE_CLOCKS
__device__ unsigned g_mpr_hist[2][16];  // [0]: portal discovery iterations, [1]: refinement iterations (bins of 4)
#define MPR_COUNT(k, it) atomicAdd(&g_mpr_hist[k][(it) / 4 < 15 ? (it) / 4 : 15], 1u)
// cycles of thread 0 of a wave between the marks (other groups' divergent paths included): 0 entry ->
// portal found, 1 refinement, 2 contact point; 3 / 4 = refinement iterations / calls seen by thread 0
__device__ unsigned long long g_mpr_clk[8];
#define MPR_T0 unsigned long long mpr_t_ = clock64()
#define MPR_T(i) do { if (threadIdx.x == 0) { atomicAdd(&g_mpr_clk[i], clock64() - mpr_t_); } mpr_t_ = clock64(); } while (0)
#define MPR_ADD(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_mpr_clk[i], (unsigned long long)(v)); } while (0)
#else
#define MPR_COUNT(k, it)
#define MPR_T0
#define MPR_T(i)
#define MPR_ADD(i, v)
#endif
// `use_inside`: `inside` is a point of B to take instead of its frame origin as B's part of the interior point A.c - B.c
// (a triangle of a mesh: the point of the triangle nearest to A's centre, so that the origin ray runs along the contact normal)
template <class SUP>
MS_DEV void collide_mpr_t(const shape_t& A, const shape_t& B, float offset, manifold_t& m, const SUP& sup, bool use_inside = false, f3 inside = f3{0.f, 0.f, 0.f}) {
  manifold_clear(m);
  MPR_T0;
  MPR_ADD(4, 1);
  const float margin = offset;
  const float tol = MSSIM_MPR_TOLERANCE;
  mvert v0, v1, v2, v3, v4;
  v0.a = A.c; v0.b = use_inside ? inside : B.c; v0.v = A.c - v0.b;
  if (dot(v0.v, v0.v) < 1e-12f) v0.v = f3{1e-5f, 0.f, 0.f};
  f3 dir = -v0.v;
  v1 = msupport(sup, A, B, dir, margin);
  if (dot(v1.v, dir) <= 0.f) return;
  dir = cross(v1.v, v0.v);
  if (dot(dir, dir) < 1e-14f) {
    f3 w = v1.v; float D = norm(w);
    m.count = 1;
    m.n = D > 1e-9f ? w * (-rcp_f(D)) : normalized(v0.v);
    m.sep[0] = margin - D;
    m.x[0] = (v1.a + v1.b) * 0.5f;
    return;
  }
  v2 = msupport(sup, A, B, dir, margin);
  if (dot(v2.v, dir) <= 0.f) return;
  dir = cross(v1.v - v0.v, v2.v - v0.v);
  if (dot(dir, v0.v) > 0.f) { mvert t = v1; v1 = v2; v2 = t; dir = -dir; }
  bool found = false;
  for (int it = 0; it < 32; it++) {
    v3 = msupport(sup, A, B, dir, margin);
    if (dot(v3.v, dir) <= 0.f) return;
    if (dot(cross(v1.v, v3.v), v0.v) < 0.f) { v2 = v3; dir = cross(v1.v - v0.v, v3.v - v0.v); continue; }
    if (dot(cross(v3.v, v2.v), v0.v) < 0.f) { v1 = v3; dir = cross(v3.v - v0.v, v2.v - v0.v); continue; }
    found = true;
    MPR_COUNT(0, it);
    break;
  }
  if (!found) return;
  MPR_T(0);
  bool hit = false;
  for (int it = 0; it < 48; it++) {
    MPR_ADD(3, 1);
    dir = cross(v2.v - v1.v, v3.v - v1.v);
    float dl = norm(dir);
    if (dl < 1e-14f) break;
    dir = dir * rcp_f(dl);
    if (dot(dir, v1.v) >= 0.f) hit = true;
    v4 = msupport(sup, A, B, dir, margin);
    float reach = dot(v4.v, dir);
    if (reach < 0.f && !hit) return;
    if (reach - dot(v3.v, dir) <= tol || it == 47) {
      MPR_COUNT(1, it);
      if (!hit) return;
      break;
    }
    f3 cr = cross(v4.v, v0.v);
    if (dot(v1.v, cr) > 0.f) {
      if (dot(v2.v, cr) > 0.f) v1 = v4; else v3 = v4;
    } else {
      if (dot(v3.v, cr) > 0.f) v2 = v4; else v1 = v4;
    }
  }
  if (!hit) return;
  MPR_T(1);
  float w[3];
  closest_on_triangle(v1.v, v2.v, v3.v, w);
  f3 wp = v1.v * w[0] + v2.v * w[1] + v3.v * w[2];
  float D = norm(wp);
  f3 pn = normalized(cross(v2.v - v1.v, v3.v - v1.v));
  m.count = 1;
  m.n = D > 1e-7f ? wp * (-rcp_f(D)) : -pn;
  m.sep[0] = margin - D;
  f3 pa = v1.a * w[0] + v2.a * w[1] + v3.a * w[2];
  f3 pb = v1.b * w[0] + v2.b * w[1] + v3.b * w[2];
  m.x[0] = (pa + pb) * 0.5f;
}
MS_DEV void collide_mpr(const shape_t& A, const shape_t& B, float offset, manifold_t& m) { collide_mpr_t(A, B, offset, m, SupDirect{}); }
