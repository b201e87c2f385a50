// oracle_sim.cpp -- CPU restatement of the batched rigid-body step (TEST INFRASTRUCTURE ONLY).
//
// Exports the ABI of include/mssim.h under the prefix `mssim_ref_` with HOST pointers. It is the
// oracle the HIP kernels are checked against and the `cpu_baseline` of bench.py; nothing under
// maniskill_amd/ links or calls it.
//
// PARITY UNPINNED vs PhysX: the reference (ManiSkill) contains no physics source -- the step is
// `px.step()` into the closed SAPIEN/PhysX wheel (mani_skill/envs/scene.py:374-375), which is not
// installed here, and none of the reference's tests hold numeric physics vectors.
// This oracle therefore restates the stages of SURVEY.md 2.2 from the published algorithms with
// the reference's parameters (mani_skill/utils/structs/types.py:36-67,
// agents/robots/panda/panda.py:68-74, agents/controllers/pd_joint_pos.py:35-49) and is itself
// pinned by closed-form / independent-algorithm tests in tests/test_oracle_*.py.
//
// One substep (dt = timestep), per env:
//   1. FK                        world pose of every moving body           (Featherstone, RBDA ch.4)
//   2. narrowphase               oracle_collide.hpp on the candidate pairs
//   3. joint-space dynamics      CRBA mass matrix M, RNEA bias c           (RBDA ch.5, ch.6)
//      implicit PD drives + tendon:  A = M + dt*Kd + dt^2*Kp (+ tendon), force-limit active set
//      unconstrained velocity        qd* = A^-1 (M qd + dt (tau0 - c + qf))
//   4. rows: joint limits, contact normal + 2 friction (pyramid) per point
//   5. projected Gauss-Seidel: `position_iterations` with bias, then `velocity_iterations`
//      without penetration bias (types.py:42-43: 15 + 1)
//   6. semi-implicit Euler; FK at the new state for the link pose / velocity outputs
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define MSSIM_PREFIX mssim_ref_
#include "../include/mssim.h"
#include "oracle_collide.hpp"

#ifndef ORACLE_REAL
#define ORACLE_REAL double
#endif
typedef ORACLE_REAL Real;
using namespace orc;
typedef V3<Real> Vec;
typedef Q4<Real> Quat;
typedef M3<Real> Mat;

namespace {

const int MAXC = MSSIM_MAX_CONTACTS;  // contact points per env fed to the solver (after the patch reduction)
std::string g_create_error;

struct SpatialV { Vec w, v; };           // motion: angular, linear (of the point at the origin O)
struct SpatialF { Vec n, f; };           // force: moment about O, force
struct SpatialI { Real m; Vec h; Mat I; };  // inertia about O: mass, m*c, rotational

inline SpatialV crossm(const SpatialV& a, const SpatialV& b) { return {cross(a.w, b.w), cross(a.w, b.v) + cross(a.v, b.w)}; }
inline SpatialF crossf(const SpatialV& a, const SpatialF& b) { return {cross(a.w, b.n) + cross(a.v, b.f), cross(a.w, b.f)}; }
inline SpatialF imul(const SpatialI& I, const SpatialV& a) { return {I.I * a.w + cross(I.h, a.v), a.v * I.m - cross(I.h, a.w)}; }
inline Real sdot(const SpatialV& s, const SpatialF& f) { return dot(s.w, f.n) + dot(s.v, f.f); }

struct Model {
  int n_dof = 0, n_tendon = 0, n_link = 0, n_free = 0, n_kin = 0, n_shape = 0, n_pair = 0;
  std::vector<int32_t> dof_parent, dof_type, body_gravity, tendon_dof, link_body, free_gravity;
  std::vector<float> dof_frame, dof_axis, dof_limit, dof_drive, dof_armature, body_inertial, tendon_param, link_frame;
  std::vector<float> free_inertial, free_damping;
  std::vector<int32_t> shape_type, shape_kind, shape_index, shape_row, shape_hull, pair_shape;
  std::vector<float> shape_frame, shape_param, shape_material, shape_bound, hull_verts, tri_soup;
  // per-env overrides ([items][N])
  int N = 0, n_env_shape = 0, n_env_free = 0;
  std::vector<int32_t> shape_env_slot, free_env_slot;
  std::vector<float> env_shape_frame, env_shape_param, env_shape_bound, env_free_inertial;
  Real gravity[3], dt, contact_offset, rest_offset, erp, max_depen, sleep_threshold;
  bool cold = false;  // oracle-only (env var MSSIM_REF_COLD=1): no warm start of the contact multipliers, scripts/tgs_vs_pgs.py
  bool tgs = false;  // oracle-only experiment (env var MSSIM_REF_TGS=1): position sub-stepping instead of PGS, DESIGN.md section 2
  int pos_iters, vel_iters;
};

struct Contact {
  int pair, ka, ia, kb, ib;  // pair index, body kinds / indices
  Vec x, n;
  Real sep, mu;
  Real lam[3];
  int slot;  // index of the point in its shape pair's manifold (before the patch reduction): the warm-start key
  // > 0 on the LAST contact of a manifold whose shapes carry a torsional patch radius: a torsional friction row about
  // the normal follows it, bounded by tors_mu * (sum of the normal multipliers of the manifold's contacts, which start
  // at contact `tors_first`)
  Real tors_mu;
  int tors_first;
  Vec tors_n;  // the patch anchor's normal
};

// one persistent manifold (include/mssim.h, MSSIM_PCM_*)
struct PcmSlot {
  int pair = -1, npts = 0, stamp = 0;
  Vec relp;      // B's frame origin in A's frame at the last full query
  Mat relR;      // B's rotation in A's frame at the last full query
  Vec n_loc;     // normal in A's frame
  Vec pA[4], pB[4];  // the contact point in A's / B's frame
  Real sep0[4];      // gap it was generated with
  bool queried_empty = false;  // the last full query found no contact
  int grow = 0;                // full queries still owed to a young manifold with fewer than 3 points
};

struct EnvState {
  Pose<Real> root;
  std::vector<Real> q, qd, qt, qdt, qf, qacc;
  std::vector<Pose<Real>> free_pose;
  std::vector<Vec> free_v, free_w, free_force;
  std::vector<char> free_calm, free_disturbed;  // energy below the threshold at the start of the substep | touched by a disturber in it
  std::vector<Real> free_wake;  // seconds of low energy left before the body goes to sleep; <= 0: asleep (include/mssim.h)
  std::vector<Pose<Real>> kin_pose;
  // derived
  std::vector<Pose<Real>> body_pose;  // moving bodies
  std::vector<SpatialV> body_vel;     // about O = root position
  std::vector<Vec> pair_impulse;
  // warm start (include/mssim.h): multipliers (normal, t1, t2) of manifold point `slot` of shape pair p at [4 p + slot],
  // valid in the substep after the one that wrote them (stamp = its pcm_tick)
  struct Warm { Real lam[3]; int stamp; };
  std::vector<Warm> warm;
  std::vector<int> pair_count;
  std::vector<PcmSlot> pcm = std::vector<PcmSlot>(MSSIM_PCM_SLOTS);
  int pcm_tick = 0;
  int mpr_queries = 0;  // full convex queries of the last substep (test / diagnostics)
  int overflow = 0;
  int raw_points = 0;  // manifold points of the last substep before the patch reduction
};

template <typename T> std::vector<T> cp(const T* p, size_t n) { return p ? std::vector<T>(p, p + n) : std::vector<T>(n); }
inline Pose<Real> pose7(const float* f) {
  Pose<Real> p;
  p.p = Vec(f[0], f[1], f[2]);
  p.q = qnormalized(Quat(f[3], f[4], f[5], f[6]));
  return p;
}

}  // namespace

struct mssim_sim {
  Model M;
  int N = 0;
  std::vector<EnvState> env;
  mssim_buffers buf{};
  std::string err;
  std::vector<std::vector<int32_t>> pair_queries, body_queries;
  int overflow_total = 0;
};

namespace {

// ------------------------------------------------------------------ kinematics
void fk(const Model& M, EnvState& E, std::vector<Vec>* axis_w, std::vector<Vec>* anchor) {
  int n = M.n_dof;
  E.body_pose.resize(n);
  if (axis_w) axis_w->resize(n);
  if (anchor) anchor->resize(n);
  for (int j = 0; j < n; j++) {
    const Pose<Real>& P = M.dof_parent[j] < 0 ? E.root : E.body_pose[M.dof_parent[j]];
    Pose<Real> J = pmul(P, pose7(&M.dof_frame[7 * j]));
    Vec al(M.dof_axis[3 * j], M.dof_axis[3 * j + 1], M.dof_axis[3 * j + 2]);
    Vec aw = qrot(J.q, al);
    Pose<Real> B = J;
    if (M.dof_type[j] == MSSIM_JOINT_REVOLUTE) B.q = qnormalized(qmul(J.q, qaxis_angle(al, E.q[j])));
    else B.p = J.p + aw * E.q[j];
    E.body_pose[j] = B;
    if (axis_w) (*axis_w)[j] = aw;
    if (anchor) (*anchor)[j] = J.p;
  }
}

inline SpatialV joint_subspace(const Model& M, int j, const Vec& aw, const Vec& anchor, const Vec& O) {
  if (M.dof_type[j] == MSSIM_JOINT_REVOLUTE) return {aw, cross(anchor - O, aw)};
  return {Vec(), aw};
}

void body_velocities(const Model& M, EnvState& E, const std::vector<Vec>& axis_w, const std::vector<Vec>& anchor,
                     const std::vector<Real>& qd) {
  int n = M.n_dof;
  E.body_vel.resize(n);
  Vec O = E.root.p;
  for (int j = 0; j < n; j++) {
    SpatialV S = joint_subspace(M, j, axis_w[j], anchor[j], O);
    SpatialV V = M.dof_parent[j] < 0 ? SpatialV{Vec(), Vec()} : E.body_vel[M.dof_parent[j]];
    V.w += S.w * qd[j];
    V.v += S.v * qd[j];
    E.body_vel[j] = V;
  }
}

Pose<Real> body_world_pose(const Model& M, const EnvState& E, int kind, int index) {
  switch (kind) {
    case MSSIM_BODY_ART: return index < 0 ? E.root : E.body_pose[index];
    case MSSIM_BODY_FREE: return E.free_pose[index];
    case MSSIM_BODY_KIN: return E.kin_pose[index];
    default: return Pose<Real>();
  }
}

Shape<Real> make_shape(const Model& M, const EnvState& E, int s, int e) {
  Shape<Real> sh;
  const int slot = M.shape_env_slot.empty() ? -1 : M.shape_env_slot[s];
  float fr[7], pr[4];
  for (int k = 0; k < 7; k++) fr[k] = slot < 0 ? M.shape_frame[7 * s + k] : M.env_shape_frame[(size_t)(7 * slot + k) * M.N + e];
  for (int k = 0; k < 4; k++) pr[k] = slot < 0 ? M.shape_param[4 * s + k] : M.env_shape_param[(size_t)(4 * slot + k) * M.N + e];
  Pose<Real> W = pmul(body_world_pose(M, E, M.shape_kind[s], M.shape_index[s]), pose7(fr));
  sh.type = M.shape_type[s];
  if (slot >= 0 && pr[3] > 0) sh.type = (int)pr[3] - 1;  // this env's own shape type (include/mssim.h env_shape_param)
  sh.c = W.p;
  sh.rot = qmat(W.q);
  for (int k = 0; k < 4; k++) sh.param[k] = pr[k];
  sh.verts = M.hull_verts.data() + 3 * M.shape_hull[2 * s];
  sh.nverts = M.shape_hull[2 * s + 1];
  if (slot >= 0 && sh.type == SH_CONVEX) {  // this env's own hull (include/mssim.h env_shape_param)
    sh.verts = M.hull_verts.data() + 3 * (size_t)(int)pr[0];
    sh.nverts = (int)pr[1];
  }
  return sh;
}

// ------------------------------------------------------------------ narrowphase over the pair table
// bounding sphere of shape s in env e: centre in the BODY frame, radius
inline void shape_bound_body(const Model& M, int s, int e, Vec& c, Real& r) {
  const int slot = M.shape_env_slot.empty() ? -1 : M.shape_env_slot[s];
  if (slot >= 0) {
    const float* b = M.env_shape_bound.data();
    c = Vec(b[(size_t)(4 * slot) * M.N + e], b[(size_t)(4 * slot + 1) * M.N + e], b[(size_t)(4 * slot + 2) * M.N + e]);
    r = b[(size_t)(4 * slot + 3) * M.N + e];
  } else {
    Pose<Real> F = pose7(&M.shape_frame[7 * s]);
    c = F.p + qrot(F.q, Vec(M.shape_bound[4 * s], M.shape_bound[4 * s + 1], M.shape_bound[4 * s + 2]));
    r = M.shape_bound[4 * s + 3];
  }
}

// body of a shape as one id: all world-fixed shapes are one body (0), the articulation base 1, its moving bodies
// 2.., free bodies, kinematic bodies
inline int body_id(int kind, int index) {
  switch (kind) {
    case MSSIM_BODY_ART: return 2 + index;  // index -1 = base
    case MSSIM_BODY_FREE: return 2 + MSSIM_MAX_DOF + index;
    case MSSIM_BODY_KIN: return 2 + MSSIM_MAX_DOF + MSSIM_MAX_FREE + index;
    default: return 0;
  }
}

// Contact patches (include/mssim.h, MSSIM_PATCH_COS): `man` = the manifolds of this env in pair order (first contact,
// count, normal, body pair), `raw` their points. A manifold's anchor is the first manifold of the same body pair
// whose normal lies within the patch cone of its own (itself if none earlier does); the points of all manifolds
// with one anchor form a patch. A patch with more than 4 points keeps: the deepest point, the point farthest from
// it, and the points of largest area on either side of that edge (measured about the anchor's normal) -- first
// candidate wins ties, as in the box-box manifold. keep[i] = 1 for the surviving raw contacts.
struct RawManifold { int first, count, key; Vec n; };
void reduce_patches(const std::vector<RawManifold>& man, const std::vector<Contact>& raw, std::vector<char>& keep, std::vector<int>& anchor) {
  const int nm = (int)man.size();
  keep.assign(raw.size(), 1);
  anchor.assign(nm, 0);
  for (int i = 0; i < nm; i++) {
    anchor[i] = i;
    for (int k = 0; k < i; k++)
      if (man[k].key == man[i].key && dot(man[k].n, man[i].n) >= Real(MSSIM_PATCH_COS)) { anchor[i] = k; break; }
  }
  for (int a = 0; a < nm; a++) {
    std::vector<int> pts;  // raw contact indices of the patch, in (manifold, point) order
    for (int i = a; i < nm; i++)
      if (anchor[i] == a)
        for (int k = 0; k < man[i].count; k++) pts.push_back(man[i].first + k);
    const int n = (int)pts.size();
    if (n <= 4) continue;
    const Vec na = man[a].n;
    // every scan: the extremum, then the first candidate within the tie tolerance of it (include/mssim.h)
    Real best = raw[pts[0]].sep;
    for (int i = 1; i < n; i++) best = std::min(best, raw[pts[i]].sep);
    int i0 = -1;
    for (int i = 0; i < n && i0 < 0; i++)
      if (raw[pts[i]].sep <= best + Real(MSSIM_PATCH_TIE_SEP)) i0 = i;
    const Vec p0 = raw[pts[i0]].x;
    // a patch resting on three or more points: the ones well above them do not compete (include/mssim.h MSSIM_PATCH_SLACK)
    int near_ = 0;
    for (int i = 0; i < n; i++) near_ += raw[pts[i]].sep <= best + Real(MSSIM_PATCH_SLACK) ? 1 : 0;
    const bool slack = near_ >= 3;
    auto cand = [&](int i) { return !slack || raw[pts[i]].sep <= best + Real(MSSIM_PATCH_SLACK); };
    auto first_near_max = [&](auto&& value, auto&& allowed, Real floor_) {
      Real mx = floor_;
      for (int i = 0; i < n; i++)
        if (allowed(i) && cand(i)) mx = std::max(mx, value(i));
      if (!(mx > floor_)) return -1;
      for (int i = 0; i < n; i++)
        if (allowed(i) && cand(i) && value(i) >= mx - Real(MSSIM_PATCH_TIE_REL) * mx) return i;
      return -1;
    };
    const int i1 = first_near_max([&](int i) { const Vec d = raw[pts[i]].x - p0; return dot(d, d); }, [&](int i) { return i != i0; }, Real(-1));
    const Vec ed = raw[pts[i1]].x - p0;
    auto area = [&](int i) { return dot(cross(ed, raw[pts[i]].x - p0), na); };
    const int i2 = first_near_max([&](int i) { return std::fabs(area(i)); }, [&](int i) { return i != i0 && i != i1; }, Real(-1));
    const Real sgn2 = area(i2);
    // (the 4th point has to add area: one on the edge i0-i1 up to rounding -- candidates along one line, the edge of a box
    // over several triangles -- would be there or not by the sign of a rounding error)
    const int i3 = first_near_max([&](int i) { return sgn2 >= 0 ? -area(i) : area(i); }, [&](int i) { return i != i0 && i != i1 && i != i2; }, Real(MSSIM_PATCH_TIE_REL) * std::fabs(sgn2));
    for (int i = 0; i < n; i++) keep[pts[i]] = (i == i0 || i == i1 || i == i2 || i == i3) ? 1 : 0;
  }
}

// the 4 most significant of n <= 8 points (deepest, farthest from it, largest area on either side of that edge about
// `na`; every scan takes the first candidate within the tie tolerance of the extremum): keep[i] = 1 for the survivors
void select4(int n, const Vec* x, const Real* sep, const Vec& na, char* keep) {
  Real best = sep[0];
  for (int i = 1; i < n; i++) best = std::min(best, sep[i]);
  int i0 = -1;
  for (int i = 0; i < n && i0 < 0; i++)
    if (sep[i] <= best + Real(MSSIM_PATCH_TIE_SEP)) i0 = i;
  const Vec p0 = x[i0];
  auto first_near_max = [&](auto&& value, auto&& allowed, Real floor_) {
    Real mx = floor_;
    for (int i = 0; i < n; i++)
      if (allowed(i)) mx = std::max(mx, value(i));
    if (!(mx > floor_)) return -1;
    for (int i = 0; i < n; i++)
      if (allowed(i) && value(i) >= mx - Real(MSSIM_PATCH_TIE_REL) * mx) return i;
    return -1;
  };
  const int i1 = first_near_max([&](int i) { const Vec d = x[i] - p0; return dot(d, d); }, [&](int i) { return i != i0; }, Real(-1));
  const Vec ed = x[i1] - p0;
  auto area = [&](int i) { return dot(cross(ed, x[i] - p0), na); };
  const int i2 = first_near_max([&](int i) { return std::fabs(area(i)); }, [&](int i) { return i != i0 && i != i1; }, Real(-1));
  const Real sgn2 = area(i2);
  const int i3 = first_near_max([&](int i) { return sgn2 >= 0 ? -area(i) : area(i); }, [&](int i) { return i != i0 && i != i1 && i != i2; }, Real(MSSIM_PATCH_TIE_REL) * std::fabs(sgn2));
  for (int i = 0; i < n; i++) keep[i] = (i == i0 || i == i1 || i == i2 || i == i3) ? 1 : 0;
}

// Manifold of a convex shape A against ONE triangle T of a mesh (T: a 3-vertex hull in a frame at its centroid).
// The triangle is taken as a bounded piece of its plane, normal = its face normal on the side of A's centre:
//  (1) A's plane-contact points (the corners of a box, the end spheres of a capsule, the vertices of a hull, ... exactly what
//      plane-vs-shape uses) whose gap is inside the contact offset AND whose foot lies in the triangle (tolerance
//      MSSIM_PCM_MERGE) -- a point over the neighbouring triangle is that triangle's: no tripping over inner edges;
//  (2) for a box: the triangle's own corners that lie under the box (a line along the normal through the corner enters the
//      box within the contact offset) -- a face larger than the triangle, a box lying across a narrow strip or over a rim;
//  the 4 deepest of (1) and (2), exact gaps. Only when both are empty the generic query runs, with the triangle's point
//  nearest to A's centre as its interior point: contacts from the side (the rim of a mesh) keep its normal; an answer within
//  60 degrees of the face normal takes the face normal (for a box with the gap measured exactly at the point's foot).
void tri_manifold(const Shape<Real>& A, const Shape<Real>& T, Real offset, Manifold<Real>& m, int& queries) {
  m.count = 0;
  const Vec q[3] = {T.c + T.rot * Vec(T.verts[0], T.verts[1], T.verts[2]), T.c + T.rot * Vec(T.verts[3], T.verts[4], T.verts[5]),
                    T.c + T.rot * Vec(T.verts[6], T.verts[7], T.verts[8])};
  Vec nf = normalized(cross(q[1] - q[0], q[2] - q[0]));
  if (dot(nf, A.c - q[0]) < 0) nf = -nf;  // the side A's centre is on
  const Vec e0 = q[1] - q[0], e1 = q[2] - q[0];
  const Real d00 = dot(e0, e0), d01 = dot(e0, e1), d11 = dot(e1, e1);
  const Real den = d00 * d11 - d01 * d01, tol = Real(MSSIM_PCM_MERGE) / std::sqrt(std::min(d00, d11));
  Vec pts[72];
  Real seps[72];
  int n = 0;
  // (box A) where the line o + t nf enters the box, in the box frame against its three slabs: the gap between a point o of
  // the triangle's plane and the box above it; false if the line misses the box or the gap is beyond the contact offset
  auto box_above = [&](const Vec& o, Real& t_in) {
    const Vec dl = A.rot.tmul(nf), ol = A.rot.tmul(o - A.c);
    const Real hb[3] = {A.param[0], A.param[1], A.param[2]}, dv[3] = {dl.x, dl.y, dl.z}, ov[3] = {ol.x, ol.y, ol.z};
    Real t_out = Real(1e30);
    t_in = Real(-1e30);
    for (int a = 0; a < 3; a++) {
      if (std::fabs(dv[a]) < Real(1e-9)) {
        if (std::fabs(ov[a]) > hb[a]) return false;
        continue;
      }
      const Real t0 = (-hb[a] - ov[a]) / dv[a], t1 = (hb[a] - ov[a]) / dv[a];
      t_in = std::max(t_in, std::min(t0, t1));
      t_out = std::min(t_out, std::max(t0, t1));
    }
    return t_in <= t_out && t_in < offset;
  };
  // Points much higher above the plane than A's lowest one are left out (MSSIM_TRI_SLACK): the contact offset would admit
  // them as speculative contacts, but in the patch reduction -- which goes by extent, not by depth -- they would crowd out
  // the points that carry the load. A's lowest point itself is always a candidate: the first touch stays speculative.
  Real s_low = Real(1e30);
  auto lowest = [&](const Vec& p, Real radius) { s_low = std::min(s_low, dot(nf, p - q[0]) - radius); };
  auto add = [&](const Vec& p, Real radius) {
    const Real s = dot(nf, p - q[0]) - radius;
    if (!(s < offset) || !(s <= s_low + Real(MSSIM_TRI_SLACK)) || n >= 64) return;
    const Vec d = p - nf * (radius + s) - q[0];  // the foot of the surface point on the triangle's plane
    const Real d20 = dot(d, e0), d21 = dot(d, e1);
    const Real v = (d11 * d20 - d01 * d21) / den, w = (d00 * d21 - d01 * d20) / den;
    if (v < -tol || w < -tol || v + w > Real(1) + tol) return;
    pts[n] = p - nf * (radius + Real(0.5) * s);
    seps[n] = s;
    n++;
  };
  if (A.type == SH_BOX) {
    Vec corner[8];
    for (int i = 0; i < 8; i++) {
      corner[i] = A.c + A.rot * Vec((i & 1) ? A.param[0] : -A.param[0], (i & 2) ? A.param[1] : -A.param[1], (i & 4) ? A.param[2] : -A.param[2]);
      lowest(corner[i], Real(0));
    }
    for (int i = 0; i < 8; i++) add(corner[i], Real(0));
    // (2) the triangle's corners under the box
    for (int i = 0; i < 3; i++) {
      Real t_in;
      if (!box_above(q[i], t_in) || !(t_in <= s_low + Real(MSSIM_TRI_SLACK))) continue;
      pts[n] = q[i] + nf * (Real(0.5) * t_in);
      seps[n] = t_in;
      n++;
    }
    // (3) the triangle's edges under the box: where edge i enters and leaves the box's shadow along nf, and -- if deeper than
    // both -- the point in between where the box comes nearest (an edge of the box across the triangle). In the box frame a
    // point of the edge is P + s Q; the line through it along nf meets slab a for t in [al_a + be_a s, al_a + be_a s + wid_a].
    {
      const Vec dl = A.rot.tmul(nf);
      const Real hb[3] = {A.param[0], A.param[1], A.param[2]}, dv[3] = {dl.x, dl.y, dl.z};
      for (int i = 0; i < 3; i++) {
        const Vec qa = q[i], qb = q[(i + 1) % 3];
        const Vec Pv = A.rot.tmul(qa - A.c), Qv = A.rot.tmul(qb - qa);
        const Real P[3] = {Pv.x, Pv.y, Pv.z}, Q[3] = {Qv.x, Qv.y, Qv.z};
        Real s0 = 0, s1 = 1, al[3], be[3], wid[3];
        bool par[3], empty = false;
        auto clip = [&](Real a0, Real b0) {  // a0 + b0 s <= 0
          if (b0 > 0) s1 = std::min(s1, -a0 / b0);
          else if (b0 < 0) s0 = std::max(s0, -a0 / b0);
          else if (a0 > 0) empty = true;
        };
        for (int a = 0; a < 3; a++) {
          par[a] = std::fabs(dv[a]) < Real(1e-9);
          if (par[a]) {
            clip(P[a] - hb[a], Q[a]);
            clip(-P[a] - hb[a], -Q[a]);
            al[a] = be[a] = wid[a] = 0;
          } else {
            al[a] = (-(dv[a] > 0 ? hb[a] : -hb[a]) - P[a]) / dv[a];
            be[a] = -Q[a] / dv[a];
            wid[a] = 2 * hb[a] / std::fabs(dv[a]);
          }
        }
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++)
            if (a != b && !par[a] && !par[b]) clip(al[a] - al[b] - wid[b], be[a] - be[b]);
        if (empty || !(s0 <= s1)) continue;
        auto t_in_at = [&](Real sv) {
          Real t = Real(-1e30);
          for (int a = 0; a < 3; a++)
            if (!par[a]) t = std::max(t, al[a] + be[a] * sv);
          return t;
        };
        const Real f0 = t_in_at(s0), f1 = t_in_at(s1);
        Real sm = s0, fm = std::min(f0, f1);
        bool inner = false;
        for (int a = 0; a < 3; a++)
          for (int b = a + 1; b < 3; b++) {
            if (par[a] || par[b] || be[a] == be[b]) continue;
            const Real sx = (al[b] - al[a]) / (be[a] - be[b]);
            if (!(sx > s0 && sx < s1)) continue;
            const Real fx = t_in_at(sx);
            if (fx < fm - Real(1e-6)) { fm = fx; sm = sx; inner = true; }
          }
        const Real sv[3] = {s0, s1, sm}, fv[3] = {f0, f1, fm};
        const bool use[3] = {s0 > 0, s1 < 1 && s1 > s0, inner};
        for (int k = 0; k < 3; k++) {
          if (!use[k] || !(fv[k] < offset) || !(fv[k] <= s_low + Real(MSSIM_TRI_SLACK))) continue;
          pts[n] = qa + (qb - qa) * sv[k] + nf * (Real(0.5) * fv[k]);
          seps[n] = fv[k];
          n++;
        }
      }
    }
  } else if (A.type == SH_SPHERE) {
    lowest(A.c, A.param[0]);
    add(A.c, A.param[0]);
  } else if (A.type == SH_CAPSULE) {
    const Vec ax = A.rot.col(0) * A.param[1];
    lowest(A.c - ax, A.param[0]);
    lowest(A.c + ax, A.param[0]);
    add(A.c - ax, A.param[0]);
    add(A.c + ax, A.param[0]);
  } else if (A.type == SH_CONVEX) {
    const int nv = A.nverts < 64 ? A.nverts : 64;
    for (int i = 0; i < nv; i++) lowest(A.c + A.rot * Vec(Real(A.verts[3 * i]), Real(A.verts[3 * i + 1]), Real(A.verts[3 * i + 2])), Real(0));
    for (int i = 0; i < nv; i++) add(A.c + A.rot * Vec(Real(A.verts[3 * i]), Real(A.verts[3 * i + 1]), Real(A.verts[3 * i + 2])), Real(0));
  } else {
    lowest(support(A, -nf), Real(0));
    add(support(A, -nf), Real(0));
  }
  m.n = nf;
  // the 4 deepest; gaps within MSSIM_TRI_TIE of the deepest count as equal and the first in the order above is taken (a face
  // lying on a triangle: all its candidates tie up to rounding)
  {
    bool used[72] = {false};
    for (int k = 0; k < 4 && k < n; k++) {
      Real lowest_gap = Real(1e30);
      for (int i = 0; i < n; i++)
        if (!used[i]) lowest_gap = std::min(lowest_gap, seps[i]);
      int best = -1;
      for (int i = 0; i < n && best < 0; i++)
        if (!used[i] && seps[i] <= lowest_gap + Real(MSSIM_TRI_TIE)) best = i;
      used[best] = true;
      m.x[m.count] = pts[best];
      m.sep[m.count] = seps[best];
      m.count++;
    }
  }
  if (m.count > 0) return;
  if (!(s_low < offset)) return;  // no part of A is nearer to the triangle than its lowest point is to the triangle's plane
  Vec inside;
  {
    Real w[3];
    closest_on_triangle(q[0] - A.c, q[1] - A.c, q[2] - A.c, w);
    inside = q[0] * w[0] + q[1] * w[1] + q[2] * w[2];
  }
  Manifold<Real> g;
  queries++;
  collide_mpr(A, T, offset, g, &inside);
  if (g.count == 0) return;
  // (A shifted along nf by -s_low clears the triangle's plane, so no true gap is below s_low: an answer far below it comes
  // from an origin ray that left the difference body through a side -- A's centre far off to the side of the triangle)
  if (g.sep[0] < s_low - Real(1e-3)) return;
  const bool face = dot(nf, g.n) > Real(0.5);
  m.n = face ? nf : g.n;
  m.x[0] = g.x[0];
  m.sep[0] = g.sep[0];
  if (face && A.type == SH_BOX) {
    // a box face across an edge of the mesh (no corner of either over / under the other): the query's point is only
    // approximate -- next to exact corner points of the neighbouring triangles it would displace them in the patch
    // reduction with a gap that is off by millimetres. Its foot on the triangle's plane, if in the triangle, is measured
    // against the box exactly instead.
    const Vec foot = g.x[0] - nf * dot(g.x[0] - q[0], nf), d = foot - q[0];
    const Real d20 = dot(d, e0), d21 = dot(d, e1);
    const Real v = (d11 * d20 - d01 * d21) / den, w = (d00 * d21 - d01 * d20) / den;
    Real t_in;
    if (v < -tol || w < -tol || v + w > Real(1) + tol || !box_above(foot, t_in)) return;
    m.x[0] = foot + nf * (Real(0.5) * t_in);
    m.sep[0] = t_in;
  }
  m.count = 1;
}

// generic convex pair through the persistent manifold cache (include/mssim.h, MSSIM_PCM_*)
void pcm_collide(EnvState& E, int p, const Shape<Real>& A, const Shape<Real>& B, Real offset, Manifold<Real>& m) {
  m.count = 0;
  const int tick = E.pcm_tick;
  PcmSlot* sl = nullptr;
  for (auto& s : E.pcm)
    if (s.pair == p) { sl = &s; break; }
  bool fresh = false;
  if (!sl) {
    for (auto& s : E.pcm)
      if (s.pair < 0) { sl = &s; break; }
    if (!sl) {  // the least recently used slot, unless every slot was used in this substep
      for (auto& s : E.pcm)
        if (s.stamp < tick && (!sl || s.stamp < sl->stamp)) sl = &s;
    }
    if (!sl) {  // no slot: the plain one-point query
      E.mpr_queries++;
      collide_mpr(A, B, offset, m);
      return;
    }
    fresh = true;
    sl->pair = p; sl->npts = 0;
  }
  sl->stamp = tick;
  const Vec relp = A.rot.tmul(B.c - A.c);
  const Mat relR = mmul(mtranspose(A.rot), B.rot);
  // refresh
  auto world = [&](int j, Vec& wA, Vec& wB) { wA = A.c + A.rot * sl->pA[j]; wB = B.c + B.rot * sl->pB[j]; };
  {
    const Vec nw = A.rot * sl->n_loc;
    int k = 0;
    for (int j = 0; j < sl->npts; j++) {
      Vec wA, wB;
      world(j, wA, wB);
      const Vec d = wA - wB;
      const Real dn = dot(d, nw);
      const Vec t = d - nw * dn;
      if (dot(t, t) > Real(MSSIM_PCM_DRIFT) * Real(MSSIM_PCM_DRIFT) || sl->sep0[j] + dn > offset) continue;
      sl->pA[k] = sl->pA[j]; sl->pB[k] = sl->pB[j]; sl->sep0[k] = sl->sep0[j];
      k++;
    }
    sl->npts = k;
  }
  // the full query runs for a new pair, for a pair that has moved since the last one, and for a manifold that lost
  // all its points (unless the last query of the unmoved pair already found nothing)
  bool moved = false;
  if (!fresh) {
    const Vec dp = relp - sl->relp;
    Real tr = 0;
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) tr += sl->relR.m[a][b] * relR.m[a][b];
    moved = dot(dp, dp) > Real(MSSIM_PCM_MOVE) * Real(MSSIM_PCM_MOVE) || tr < Real(MSSIM_PCM_ROT_TRACE);
  }
  // a point found by a query on shape A posed as `Aq` joins the manifold: it replaces a cached point within
  // MSSIM_PCM_MERGE of it, else it is appended; a fifth point makes the patch selection rule pick four
  auto merge = [&](const Shape<Real>& Aq, const Manifold<Real>& g) {
    const Vec pA = Aq.rot.tmul(g.x[0] - Aq.c), pB = B.rot.tmul(g.x[0] - B.c);
    int at = -1;
    for (int j = 0; j < sl->npts && at < 0; j++) {
      const Vec d = pA - sl->pA[j];
      if (dot(d, d) < Real(MSSIM_PCM_MERGE) * Real(MSSIM_PCM_MERGE)) at = j;
    }
    if (at < 0) at = sl->npts < 4 ? sl->npts++ : 4;
    if (at < 4) {
      sl->pA[at] = pA; sl->pB[at] = pB; sl->sep0[at] = g.sep[0];
      return;
    }
    // 5 candidates -> 4: world points and gaps under the true pose and the manifold's normal, the new point last
    Vec x[5], qa[5], qb[5]; Real sp[5], s0[5]; char keep[5];
    const Vec nw = A.rot * sl->n_loc;
    for (int j = 0; j < 4; j++) { qa[j] = sl->pA[j]; qb[j] = sl->pB[j]; s0[j] = sl->sep0[j]; }
    qa[4] = pA; qb[4] = pB; s0[4] = g.sep[0];
    for (int j = 0; j < 5; j++) {
      const Vec wA = A.c + A.rot * qa[j], wB = B.c + B.rot * qb[j];
      x[j] = (wA + wB) * Real(0.5);
      sp[j] = s0[j] + dot(wA - wB, nw);
    }
    select4(5, x, sp, nw, keep);
    int k = 0;
    for (int j = 0; j < 5; j++)
      if (keep[j] && k < 4) { sl->pA[k] = qa[j]; sl->pB[k] = qb[j]; sl->sep0[k] = s0[j]; k++; }
    sl->npts = k;
  };
  // A growth query looks for a corner the manifold does not have yet: shape A is tilted by MSSIM_PCM_TILT about the
  // manifold (about a tangent through its single point; about the line through its first two points; `flip` picks the
  // side), which pushes the far side of the contact face into B. The point found is stored like any other -- as a
  // material point of both shapes -- so its gap under the true pose comes out of the refresh.
  auto growth_query = [&](bool flip) {
    const Vec nw = A.rot * sl->n_loc;
    Vec w0, wb0;
    world(0, w0, wb0);
    const Vec pivot = (w0 + wb0) * Real(0.5);
    Vec axis;
    if (sl->npts >= 2) {
      Vec w1, wb1;
      world(1, w1, wb1);
      axis = normalized((w1 + wb1) * Real(0.5) - pivot);
    } else {
      axis = std::fabs(nw.x) < Real(0.57735) ? normalized(cross(nw, Vec(1, 0, 0))) : normalized(cross(nw, Vec(0, 1, 0)));
    }
    if (flip) axis = -axis;
    const Mat Rt = qmat(qaxis_angle(axis, Real(MSSIM_PCM_TILT)));
    Shape<Real> Aq = A;
    Aq.rot = mmul(Rt, A.rot);
    Aq.c = pivot + Rt * (A.c - pivot);
    Manifold<Real> g;
    E.mpr_queries++;
    collide_mpr(Aq, B, offset, g);
    if (g.count > 0) merge(Aq, g);
  };
  const bool growing = !moved && sl->npts > 0 && sl->npts < 3 && sl->grow > 0;  // a young manifold short of a face contact
  if (fresh || moved || (sl->npts == 0 && !sl->queried_empty)) {
    E.mpr_queries++;
    Manifold<Real> g;
    collide_mpr(A, B, offset, g);
    sl->relp = relp; sl->relR = relR;
    sl->queried_empty = g.count == 0;
    if (g.count == 0) {
      sl->npts = 0;
    } else {
      const Vec n_new = A.rot.tmul(g.n);
      if (sl->npts > 0 && dot(n_new, sl->n_loc) < Real(MSSIM_PATCH_COS)) sl->npts = 0;  // the contact turned: start over
      if (sl->npts == 0) sl->grow = MSSIM_PCM_GROW;  // a manifold starts: growth queries are owed to it
      sl->n_loc = n_new;
      merge(A, g);
    }
  } else if (growing) {
    sl->grow--;
    growth_query((sl->grow & 1) != 0);
  }
  // the manifold as it stands
  const Vec nw = A.rot * sl->n_loc;
  m.count = sl->npts;
  m.n = nw;
  for (int j = 0; j < sl->npts; j++) {
    Vec wA, wB;
    world(j, wA, wB);
    m.x[j] = (wA + wB) * Real(0.5);
    m.sep[j] = sl->sep0[j] + dot(wA - wB, nw);
  }
}

// oriented box around shape s (axes = the shape frame, centred at its bounding-sphere centre): half extents
inline Vec shape_obb_half(const Model& M, int s, int e) {
  const int slot = M.shape_env_slot.empty() ? -1 : M.shape_env_slot[s];
  float pr[4];
  for (int k = 0; k < 4; k++) pr[k] = slot < 0 ? M.shape_param[4 * s + k] : M.env_shape_param[(size_t)(4 * slot + k) * M.N + e];
  Vec h;
  const int type = (slot >= 0 && pr[3] > 0) ? (int)pr[3] - 1 : M.shape_type[s];
  switch (type) {
    case MSSIM_SHAPE_BOX: h = Vec(pr[0], pr[1], pr[2]); break;
    case MSSIM_SHAPE_SPHERE: h = Vec(pr[0], pr[0], pr[0]); break;
    case MSSIM_SHAPE_CAPSULE: h = Vec(pr[1] + pr[0], pr[0], pr[0]); break;
    case MSSIM_SHAPE_CYLINDER: h = Vec(pr[1], pr[0], pr[0]); break;
    case MSSIM_SHAPE_CONVEX: {
      float b[3] = {M.shape_bound[4 * s], M.shape_bound[4 * s + 1], M.shape_bound[4 * s + 2]};
      int first = M.shape_hull[2 * s], count = M.shape_hull[2 * s + 1];
      if (slot >= 0) {
        // this env's hull; its bound centre is stored in the BODY frame: back into the shape frame
        first = (int)pr[0]; count = (int)pr[1];
        float fr[7];
        for (int k = 0; k < 7; k++) fr[k] = M.env_shape_frame[(size_t)(7 * slot + k) * M.N + e];
        const Pose<Real> F = pose7(fr);
        const Vec cb(M.env_shape_bound[(size_t)(4 * slot) * M.N + e], M.env_shape_bound[(size_t)(4 * slot + 1) * M.N + e], M.env_shape_bound[(size_t)(4 * slot + 2) * M.N + e]);
        const Vec cs = qmat(F.q).tmul(cb - F.p);
        b[0] = (float)cs.x; b[1] = (float)cs.y; b[2] = (float)cs.z;
      }
      for (int i = 0; i < count; i++) {
        const float* v = &M.hull_verts[3 * (size_t)(first + i)];
        h.x = std::max(h.x, (Real)std::fabs(v[0] - b[0])); h.y = std::max(h.y, (Real)std::fabs(v[1] - b[1])); h.z = std::max(h.z, (Real)std::fabs(v[2] - b[2]));
      }
      break;
    }
    case MSSIM_SHAPE_TRIMESH: {
      float b[3] = {M.shape_bound[4 * s], M.shape_bound[4 * s + 1], M.shape_bound[4 * s + 2]};
      int first = (int)M.shape_param[4 * s], count = (int)M.shape_param[4 * s + 1];
      if (slot >= 0) {
        // this env's mesh (its triangle range; bound centre stored in the BODY frame: back into the shape frame)
        first = (int)pr[0]; count = (int)pr[1];
        float fr[7];
        for (int k = 0; k < 7; k++) fr[k] = M.env_shape_frame[(size_t)(7 * slot + k) * M.N + e];
        const Pose<Real> F = pose7(fr);
        const Vec cb(M.env_shape_bound[(size_t)(4 * slot) * M.N + e], M.env_shape_bound[(size_t)(4 * slot + 1) * M.N + e], M.env_shape_bound[(size_t)(4 * slot + 2) * M.N + e]);
        const Vec cs = qmat(F.q).tmul(cb - F.p);
        b[0] = (float)cs.x; b[1] = (float)cs.y; b[2] = (float)cs.z;
      }
      for (int t = first; t < first + count; t++) {
        const float* q = &M.tri_soup[12 * (size_t)t];
        for (int k = 0; k < 3; k++) {
          const float x = q[0] + q[3 + 3 * k], y = q[1] + q[4 + 3 * k], z = q[2] + q[5 + 3 * k];
          h.x = std::max(h.x, (Real)std::fabs(x - b[0])); h.y = std::max(h.y, (Real)std::fabs(y - b[1])); h.z = std::max(h.z, (Real)std::fabs(z - b[2]));
        }
      }
      break;
    }
    default: h = Vec(3e30, 3e30, 3e30);
  }
  if (slot < 0 && type != MSSIM_SHAPE_CONVEX && type != MSSIM_SHAPE_PLANE && type != MSSIM_SHAPE_TRIMESH) {
    // primitives are centred on their frame; the box stays valid if the bound centre is offset
    h.x += std::fabs(M.shape_bound[4 * s]); h.y += std::fabs(M.shape_bound[4 * s + 1]); h.z += std::fabs(M.shape_bound[4 * s + 2]);
  }
  return h;
}

// separating-axis test of two oriented boxes, radii enlarged by `margin`; true = certainly apart (Gottschalk et al.)
inline bool obb_separated(const Mat& RA, const Vec& ha, const Mat& RB, const Vec& hb, const Vec& d, Real margin) {
  Real R[3][3], AR[3][3];
  const Real t[3] = {dot(RA.col(0), d), dot(RA.col(1), d), dot(RA.col(2), d)};
  const Real a[3] = {ha.x, ha.y, ha.z}, b[3] = {hb.x, hb.y, hb.z};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { R[i][j] = dot(RA.col(i), RB.col(j)); AR[i][j] = std::fabs(R[i][j]) + Real(1e-6); }
  for (int i = 0; i < 3; i++)
    if (std::fabs(t[i]) > a[i] + b[0] * AR[i][0] + b[1] * AR[i][1] + b[2] * AR[i][2] + margin) return true;
  for (int j = 0; j < 3; j++)
    if (std::fabs(t[0] * R[0][j] + t[1] * R[1][j] + t[2] * R[2][j]) > b[j] + a[0] * AR[0][j] + a[1] * AR[1][j] + a[2] * AR[2][j] + margin) return true;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const Real ra = a[i1] * AR[i2][j] + a[i2] * AR[i1][j];
      const Real rb = b[j1] * AR[i][j2] + b[j2] * AR[i][j1];
      if (std::fabs(t[i2] * R[i1][j] - t[i1] * R[i2][j]) > ra + rb + margin) return true;
    }
  return false;
}

void narrowphase(const Model& M, EnvState& E, int e, std::vector<Contact>& out) {
  out.clear();
  E.pair_count.assign(M.n_pair, 0);
  std::vector<Shape<Real>> sh(M.n_shape);
  for (int s = 0; s < M.n_shape; s++) sh[s] = make_shape(M, E, s, e);
  std::vector<Contact> raw;
  std::vector<RawManifold> man;
  // (the manifolds of mesh triangles follow those of all other pairs, as in the kernels, which append them to the hit list)
  std::vector<Contact> raw_tri;
  std::vector<RawManifold> man_tri;
  E.pcm_tick++;
  E.mpr_queries = 0;
  int hits = 0;  // pairs that survive the cull (the kernels' hit list holds MSSIM_MAX_HITS of them)
  // the cull of pair p; for a survivor also the world centres of the two bounding spheres and A's radius
  auto survives = [&](int p, Vec& ca, Vec& cb, Real& ra) {
    int sa = M.pair_shape[2 * p], sb = M.pair_shape[2 * p + 1];
    const Shape<Real>&A = sh[sa], &B = sh[sb];
    // a pair of bodies that cannot move in this substep -- fixed in the env frame or asleep at its start -- needs no
    // manifold (a sleeping body woken later in the substep gets its contacts with the static scene back in the next one)
    auto inactive = [&](int s) {
      return M.shape_kind[s] == MSSIM_BODY_WORLD || (M.shape_kind[s] == MSSIM_BODY_FREE && E.free_wake[M.shape_index[s]] <= 0);
    };
    if (inactive(sa) && inactive(sb)) return false;
    if (A.type == MSSIM_SHAPE_NONE || B.type == MSSIM_SHAPE_NONE) return false;  // no shape in this env's slot
    // bounding-sphere cull
    Real rb;
    Vec cla, clb;
    shape_bound_body(M, sa, e, cla, ra);
    shape_bound_body(M, sb, e, clb, rb);
    Pose<Real> PA = body_world_pose(M, E, M.shape_kind[sa], M.shape_index[sa]), PB = body_world_pose(M, E, M.shape_kind[sb], M.shape_index[sb]);
    ca = PA.p + qrot(PA.q, cla);
    cb = PB.p + qrot(PB.q, clb);
    if (A.type == SH_PLANE) return !(dot(A.rot.col(0), cb - A.c) > rb + M.contact_offset);
    Vec d = cb - ca;
    Real rr = ra + rb + M.contact_offset;
    if (dot(d, d) > rr * rr) return false;
    // second stage, as the kernels: oriented boxes of the two shapes (conservative; it decides which pairs reach
    // the persistent manifold cache, so both implementations must apply it alike)
    return !obb_separated(A.rot, shape_obb_half(M, sa, e), B.rot, shape_obb_half(M, sb, e), d, M.contact_offset);
  };
  // Triangle mesh (include/mssim.h MSSIM_SHAPE_TRIMESH): the triangles of pair p's mesh in `range` of its convex shape, in
  // index order -- the triangle's box against the shape's bounding sphere and against the bounds of its oriented box (mesh
  // frame), the triangle's corners along the oriented box's own axes, then the oriented box against the triangle's plane
  auto triangles_in_range = [&](int p, const Vec& ca, Real ra, Real range, std::vector<int>& found) {
    found.clear();
    int sa = M.pair_shape[2 * p], sb = M.pair_shape[2 * p + 1];
    const Shape<Real>&A = sh[sa], &B = sh[sb];
    const Vec cq = B.rot.tmul(ca - B.c);
    const Real rq = ra + range;
    const Vec hA = shape_obb_half(M, sa, e);
    Vec un[3], ax[3];
    for (int i = 0; i < 3; i++) { un[i] = B.rot.tmul(A.rot.col(i)); ax[i] = un[i] * (i == 0 ? hA.x : i == 1 ? hA.y : hA.z); }
    const Vec ext(std::fabs(ax[0].x) + std::fabs(ax[1].x) + std::fabs(ax[2].x), std::fabs(ax[0].y) + std::fabs(ax[1].y) + std::fabs(ax[2].y),
                  std::fabs(ax[0].z) + std::fabs(ax[1].z) + std::fabs(ax[2].z));
    int first = (int)M.shape_param[4 * sb], count = (int)M.shape_param[4 * sb + 1];
    {  // (a slot may hold a different mesh in every env: its triangle range is in the env's row)
      const int slot = M.shape_env_slot.empty() ? -1 : M.shape_env_slot[sb];
      if (slot >= 0) {
        first = (int)M.env_shape_param[(size_t)(4 * slot) * M.N + e];
        count = (int)M.env_shape_param[(size_t)(4 * slot + 1) * M.N + e];
      }
    }
    for (int t = first; t < first + count; t++) {
      const float* q = &M.tri_soup[12 * (size_t)t];
      float lo[3], hi[3];
      for (int a = 0; a < 3; a++) {
        const float x0 = q[a] + q[3 + a], x1 = q[a] + q[6 + a], x2 = q[a] + q[9 + a];
        lo[a] = std::min(x0, std::min(x1, x2)); hi[a] = std::max(x0, std::max(x1, x2));
      }
      const Real dx = std::max(std::max((Real)lo[0] - cq.x, cq.x - (Real)hi[0]), Real(0)), dy = std::max(std::max((Real)lo[1] - cq.y, cq.y - (Real)hi[1]), Real(0)),
                 dz = std::max(std::max((Real)lo[2] - cq.z, cq.z - (Real)hi[2]), Real(0));
      if (dx * dx + dy * dy + dz * dz > rq * rq) continue;
      if ((Real)lo[0] - (cq.x + ext.x) > range || (cq.x - ext.x) - (Real)hi[0] > range || (Real)lo[1] - (cq.y + ext.y) > range || (cq.y - ext.y) - (Real)hi[1] > range ||
          (Real)lo[2] - (cq.z + ext.z) > range || (cq.z - ext.z) - (Real)hi[2] > range) continue;
      {  // the triangle's corners along the oriented box's own axes
        const Real hv[3] = {hA.x, hA.y, hA.z};
        bool apart = false;
        for (int i = 0; i < 3; i++) {
          Real dmin = Real(1e30), dmax = Real(-1e30);
          for (int k = 0; k < 3; k++) {
            const Real d = dot(un[i], Vec((Real)q[0] + (Real)q[3 + 3 * k], (Real)q[1] + (Real)q[4 + 3 * k], (Real)q[2] + (Real)q[5 + 3 * k]) - cq);
            dmin = std::min(dmin, d); dmax = std::max(dmax, d);
          }
          apart = apart || dmin > hv[i] + range || dmax < -(hv[i] + range);
        }
        if (apart) continue;
      }
      const Vec e1((Real)q[6] - (Real)q[3], (Real)q[7] - (Real)q[4], (Real)q[8] - (Real)q[5]), e2((Real)q[9] - (Real)q[3], (Real)q[10] - (Real)q[4], (Real)q[11] - (Real)q[5]);
      const Vec nn = cross(e1, e2);
      const Real len = std::sqrt(dot(nn, nn));
      if (len > Real(0)) {
        const Real dist = dot(nn, cq - Vec((Real)q[0] + (Real)q[3], (Real)q[1] + (Real)q[4], (Real)q[2] + (Real)q[5]));
        const Real rad = std::fabs(dot(nn, ax[0])) + std::fabs(dot(nn, ax[1])) + std::fabs(dot(nn, ax[2]));
        if (std::fabs(dist) - rad > range * len) continue;
      }
      found.push_back(t);
    }
  };
  // the search range of this env's mesh pairs (MSSIM_TRI_RANGE_STEPS): the largest of offset, offset/2, offset/4, 0 whose
  // triangles fit -- per pair, in the task list, and in what the surviving pairs leave of the hit list
  std::vector<std::vector<int>> tri_found(!M.tri_soup.empty() ? M.n_pair : 0);
  if (!M.tri_soup.empty()) {
    int nh0 = 0;
    std::vector<int> mesh_pairs;
    std::vector<Vec> mesh_ca;
    std::vector<Real> mesh_ra;
    for (int p = 0; p < M.n_pair; p++) {
      Vec ca, cb;
      Real ra;
      if (!survives(p, ca, cb, ra)) continue;
      // (a mesh pair takes no entry of the kernels' hit list itself: its triangles in range do)
      if (sh[M.pair_shape[2 * p + 1]].type == SH_TRIMESH) { mesh_pairs.push_back(p); mesh_ca.push_back(ca); mesh_ra.push_back(ra); }
      else nh0++;
    }
    nh0 = std::min(nh0, (int)MSSIM_MAX_HITS);
    Real range = M.contact_offset;
    for (int step = 0; step < MSSIM_TRI_RANGE_STEPS; step++) {
      int total = 0;
      bool fits = true;
      for (size_t i = 0; i < mesh_pairs.size(); i++) {
        std::vector<int>& f = tri_found[mesh_pairs[i]];
        triangles_in_range(mesh_pairs[i], mesh_ca[i], mesh_ra[i], range, f);
        fits = fits && (int)f.size() <= MSSIM_MAX_TRI_HITS;
        total += (int)f.size();
      }
      fits = fits && total <= std::min((int)MSSIM_MAX_HITS - nh0, (int)MSSIM_MAX_TRI_TASKS);
      if (fits) break;
      if (step == MSSIM_TRI_RANGE_STEPS - 1) {  // cut off in index order, pair after pair
        E.overflow |= MSSIM_OVERFLOW_TRI;
        int room = std::min((int)MSSIM_MAX_HITS - nh0, (int)MSSIM_MAX_TRI_TASKS);
        for (int p : mesh_pairs) {
          std::vector<int>& f = tri_found[p];
          if ((int)f.size() > MSSIM_MAX_TRI_HITS) f.resize(MSSIM_MAX_TRI_HITS);
          if ((int)f.size() > room) f.resize(room);
          room -= (int)f.size();
        }
      }
      range = step == MSSIM_TRI_RANGE_STEPS - 2 ? Real(0) : Real(0.5) * range;
    }
  }
  for (int p = 0; p < M.n_pair; p++) {
    int sa = M.pair_shape[2 * p], sb = M.pair_shape[2 * p + 1];
    const Shape<Real>&A = sh[sa], &B = sh[sb];
    Vec ca, cb;
    Real ra;
    if (!survives(p, ca, cb, ra)) continue;
    if (B.type == SH_TRIMESH) {
      // each triangle in range is a 3-vertex hull in a frame at its centroid and gives a manifold of its own, which the
      // patch pass merges with those of its coplanar neighbours
      for (int t : tri_found[p]) {
        const float* q = &M.tri_soup[12 * (size_t)t];
        Shape<Real> T;
        T.type = SH_CONVEX;
        T.rot = B.rot;
        T.c = B.c + B.rot * Vec(q[0], q[1], q[2]);
        T.verts = q + 3;
        T.nverts = 3;
        for (int k = 0; k < 4; k++) T.param[k] = 0;
        Manifold<Real> m;
        tri_manifold(A, T, M.contact_offset, m, E.mpr_queries);
        if (m.count <= 0) continue;
        if ((int)(raw.size() + raw_tri.size()) + m.count > MSSIM_MAX_RAW_POINTS) { E.overflow |= MSSIM_OVERFLOW_RAW; break; }
        if ((int)(man.size() + man_tri.size()) >= MSSIM_MAX_HITS) { E.overflow |= MSSIM_OVERFLOW_HITS; break; }
        const Real mu = Real(0.5) * (M.shape_material[4 * sa + 1] + M.shape_material[4 * sb + 1]);
        man_tri.push_back({(int)raw_tri.size(), m.count, body_id(M.shape_kind[sa], M.shape_index[sa]) * 64 + body_id(M.shape_kind[sb], M.shape_index[sb]), m.n});
        if (getenv("MSSIM_REF_DEBUG_TRI")) for (int k = 0; k < m.count; k++) fprintf(stderr, "tri %d pt %d x %.4f %.4f %.4f n %.3f %.3f %.3f sep %.5f\n", t, k, (double)m.x[k].x, (double)m.x[k].y, (double)m.x[k].z, (double)m.n.x, (double)m.n.y, (double)m.n.z, (double)m.sep[k]);
        for (int k = 0; k < m.count; k++) {
          Contact c;
          c.pair = p;
          c.ka = M.shape_kind[sa]; c.ia = M.shape_index[sa];
          c.kb = M.shape_kind[sb]; c.ib = M.shape_index[sb];
          c.x = m.x[k]; c.n = m.n; c.sep = m.sep[k] - M.rest_offset; c.mu = mu;
          c.lam[0] = c.lam[1] = c.lam[2] = 0;
          c.slot = -1;  // (several manifolds of one pair: no warm-start key)
          c.tors_mu = 0; c.tors_first = -1; c.tors_n = Vec();
          raw_tri.push_back(c);
        }
      }
      continue;
    }
    Manifold<Real> m;
    if (A.type == SH_PLANE || (A.type == SH_BOX && B.type == SH_BOX)) collide(A, B, M.contact_offset, m);
    else pcm_collide(E, p, A, B, M.contact_offset, m);
    if (m.count <= 0) continue;
    hits++;
    if ((int)raw.size() + m.count > MSSIM_MAX_RAW_POINTS) { E.overflow |= MSSIM_OVERFLOW_RAW; break; }
    Real mu = Real(0.5) * (M.shape_material[4 * sa + 1] + M.shape_material[4 * sb + 1]);
    man.push_back({(int)raw.size(), m.count, body_id(M.shape_kind[sa], M.shape_index[sa]) * 64 + body_id(M.shape_kind[sb], M.shape_index[sb]), m.n});
    for (int k = 0; k < m.count; k++) {
      Contact c;
      c.pair = p;
      c.ka = M.shape_kind[sa]; c.ia = M.shape_index[sa];
      c.kb = M.shape_kind[sb]; c.ib = M.shape_index[sb];
      c.x = m.x[k]; c.n = m.n; c.sep = m.sep[k] - M.rest_offset; c.mu = mu;
      c.lam[0] = c.lam[1] = c.lam[2] = 0;
      c.slot = k;
      c.tors_mu = 0; c.tors_first = -1; c.tors_n = Vec();
      raw.push_back(c);
    }
  }
  (void)hits;
  for (const RawManifold& mf : man_tri) {
    man.push_back({(int)raw.size(), mf.count, mf.key, mf.n});
    for (int k = 0; k < mf.count; k++) raw.push_back(raw_tri[mf.first + k]);
  }
  // Sleeping free bodies (include/mssim.h sleep_threshold). A "disturber" is an articulation link or a free body that is
  // awake and not calm (energy above the threshold at the start of the substep). A sleeping body touched by a disturber
  // (a manifold with points) wakes; the manifolds of a body that stays asleep are dropped (its partners are fixed,
  // asleep or calm); `free_disturbed` feeds the sleep counters at the end of the substep.
  {
    const int nf = M.n_free;
    auto disturber = [&](int kind, int index) {
      if (kind == MSSIM_BODY_ART) return index >= 0;
      if (kind == MSSIM_BODY_FREE) return E.free_wake[index] > 0 && !E.free_calm[index];
      return false;
    };
    E.free_disturbed.assign(nf, 0);
    for (const RawManifold& mf : man) {
      const Contact& c = raw[mf.first];
      if (c.ka == MSSIM_BODY_FREE && disturber(c.kb, c.ib)) E.free_disturbed[c.ia] = 1;
      if (c.kb == MSSIM_BODY_FREE && disturber(c.ka, c.ia)) E.free_disturbed[c.ib] = 1;
    }
    for (int b = 0; b < nf; b++)
      if (E.free_wake[b] <= 0 && E.free_disturbed[b]) E.free_wake[b] = Real(MSSIM_WAKE_TIME);
    std::vector<Contact> raw2;
    std::vector<RawManifold> man2;
    for (const RawManifold& mf : man) {
      const Contact& c = raw[mf.first];
      const bool drop = (c.ka == MSSIM_BODY_FREE && E.free_wake[c.ia] <= 0) || (c.kb == MSSIM_BODY_FREE && E.free_wake[c.ib] <= 0);
      if (drop) continue;
      man2.push_back({(int)raw2.size(), mf.count, mf.key, mf.n});
      for (int k = 0; k < mf.count; k++) raw2.push_back(raw[mf.first + k]);
    }
    raw.swap(raw2);
    man.swap(man2);
  }
  E.raw_points = (int)raw.size();
  std::vector<char> keep;
  std::vector<int> anchor;
  reduce_patches(man, raw, keep, anchor);
  // Solver order: patch by patch (patches in the order of their anchors), inside a patch manifold by manifold -- a
  // manifold belongs to one patch, so the points of a shape pair stay together.
  // Torsional friction (mani_skill/agents/robots/panda/panda.py:24-31: patch_radius = min_patch_radius = 0.1 on the finger
  // links; shape_material[3] = the larger of the two): a patch whose shapes carry a radius r > 0 resists spinning about
  // its (anchor's) normal with a torque of up to mu * r * (its normal force) -- one extra solver row per patch, which
  // takes one of the MAXC contact slots.
  int slots = 0;
  bool full = false;
  for (int a = 0; a < (int)man.size() && !full; a++) {
    if (anchor[a] != a) continue;
    const int first = (int)out.size();
    Real rt = 0;
    for (int i = a; i < (int)man.size() && !full; i++) {
      if (anchor[i] != a) continue;
      const int p = raw[man[i].first].pair;
      rt = std::max(rt, (Real)std::max(M.shape_material[4 * M.pair_shape[2 * p] + 3], M.shape_material[4 * M.pair_shape[2 * p + 1] + 3]));
      for (int k = 0; k < man[i].count && !full; k++) {
        const int ri = man[i].first + k;
        if (!keep[ri]) continue;
        if (slots >= MAXC) { E.overflow |= MSSIM_OVERFLOW_CONTACTS; full = true; break; }
        out.push_back(raw[ri]);
        out.back().n = raw[ri].n;
        slots++;
        E.pair_count[p]++;
      }
    }
    if (!full && rt > 0 && (int)out.size() > first) {
      if (slots >= MAXC) { E.overflow |= MSSIM_OVERFLOW_CONTACTS; full = true; break; }
      slots++;
      out.back().tors_mu = raw[man[a].first].mu * rt;
      out.back().tors_first = first;
      out.back().tors_n = man[a].n;
    }
  }
}

// ------------------------------------------------------------------ dense SPD solve (LDL^T)
bool ldl_factor(int n, std::vector<Real>& A) {  // in place, lower: L (unit) below diag, D on diag
  for (int j = 0; j < n; j++) {
    Real d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k] * A[k * n + k];
    if (!(d > Real(0))) return false;
    A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      Real v = A[i * n + j];
      for (int k = 0; k < j; k++) v -= A[i * n + k] * A[j * n + k] * A[k * n + k];
      A[i * n + j] = v / d;
    }
  }
  return true;
}
void ldl_solve(int n, const std::vector<Real>& L, Real* b) {
  for (int i = 0; i < n; i++)
    for (int k = 0; k < i; k++) b[i] -= L[i * n + k] * b[k];
  for (int i = 0; i < n; i++) b[i] /= L[i * n + i];
  for (int i = n - 1; i >= 0; i--)
    for (int k = i + 1; k < n; k++) b[i] -= L[k * n + i] * b[k];
}

struct Row {
  // sparse-dense row: articulation part (n_dof) + up to two free bodies
  std::vector<Real> Ja, Wa;
  int f[2];
  Vec Jl[2], Jw[2], Wl[2], Ww[2];
  Real diag, bias_pos, bias_vel, lo, hi, lam;
  int friction_of;  // index of the normal row bounding this friction row, -1 otherwise
  std::vector<int> tors_of;  // torsional row: the normal rows of its manifold (bound = mu * sum of their multipliers)
  Real mu;
  int contact, dirk;  // contact index / direction (0 n, 1 t1, 2 t2), -1 for limits
};

// ------------------------------------------------------------------ one substep
void substep(mssim_sim* S, EnvState& E, int e) {
  const Model& M = S->M;
  const int n = M.n_dof, nf = M.n_free;
  const Real dt = M.dt;
  Vec g(M.gravity[0], M.gravity[1], M.gravity[2]);
  Vec O = E.root.p;

  // 1. FK
  std::vector<Vec> axis_w, anchor;
  fk(M, E, &axis_w, &anchor);
  // free-body inertials; "calm" = mass-normalised kinetic energy below the sleep threshold at the start of the substep
  std::vector<float> finert(10 * (nf > 0 ? nf : 1));
  for (int b = 0; b < nf; b++) {
    const int fslot = M.free_env_slot.empty() ? -1 : M.free_env_slot[b];
    for (int k = 0; k < 10; k++)
      finert[10 * b + k] = fslot < 0 ? M.free_inertial[10 * b + k] : M.env_free_inertial[(size_t)(10 * fslot + k) * M.N + e];
  }
  auto normalised_energy = [&](int b, const Vec& v, const Vec& w) {
    const float* in = &finert[10 * b];
    Mat Rm = qmat(E.free_pose[b].q);
    Real Iv[6] = {in[4], in[5], in[6], in[7], in[8], in[9]};
    Mat Iw = mmul(mmul(Rm, sym3(Iv)), mtranspose(Rm));
    return Real(0.5) * (dot(v, v) + dot(w, Iw * w) / in[0]);
  };
  E.free_calm.assign(nf, 0);
  for (int b = 0; b < nf; b++) {
    if (!(finert[10 * b] > 0)) {  // mass 0: the body does not exist in this env (include/mssim.h env_shape_param) -- never awake
      E.free_wake[b] = 0; E.free_v[b] = Vec(); E.free_w[b] = Vec();
      E.free_calm[b] = 1;
      continue;
    }
    E.free_calm[b] = normalised_energy(b, E.free_v[b], E.free_w[b]) < M.sleep_threshold;
  }
  // 2. narrowphase
  std::vector<Contact> contacts;
  narrowphase(M, E, e, contacts);

  // 3. joint-space dynamics
  std::vector<SpatialV> Sj(n);
  std::vector<SpatialI> I(n), Ic(n);
  std::vector<SpatialV> V(n), Ab(n);
  std::vector<SpatialF> F(n);
  for (int j = 0; j < n; j++) {
    Sj[j] = joint_subspace(M, j, axis_w[j], anchor[j], O);
    const float* in = &M.body_inertial[10 * j];
    Mat Rm = qmat(E.body_pose[j].q);
    Real Iv[6] = {in[4], in[5], in[6], in[7], in[8], in[9]};
    Mat Iw = mmul(mmul(Rm, sym3(Iv)), mtranspose(Rm));
    Vec c = E.body_pose[j].p + Rm * Vec(in[1], in[2], in[3]) - O;
    Real m = in[0];
    Real cc = dot(c, c);
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) Iw.m[a][b] += m * ((a == b ? cc : Real(0)) - c[a] * c[b]);
    I[j] = {m, c * m, Iw};
  }
  for (int j = 0; j < n; j++) {
    int p = M.dof_parent[j];
    SpatialV Vp = p < 0 ? SpatialV{Vec(), Vec()} : V[p];
    SpatialV Ap = p < 0 ? SpatialV{Vec(), Vec()} : Ab[p];
    V[j] = {Vp.w + Sj[j].w * E.qd[j], Vp.v + Sj[j].v * E.qd[j]};
    SpatialV cr = crossm(V[j], Sj[j]);
    Ab[j] = {Ap.w + cr.w * E.qd[j], Ap.v + cr.v * E.qd[j]};
    SpatialF f1 = imul(I[j], Ab[j]);
    SpatialF f2 = crossf(V[j], imul(I[j], V[j]));
    F[j] = {f1.n + f2.n, f1.f + f2.f};
    if (M.body_gravity[j]) {  // external force m*g at the COM
      F[j].f -= g * I[j].m;
      F[j].n -= cross(I[j].h, g);
    }
  }
  std::vector<Real> bias(n), Mq(n * n, Real(0));
  for (int j = 0; j < n; j++) Ic[j] = I[j];
  for (int j = n - 1; j >= 0; j--) {
    bias[j] = sdot(Sj[j], F[j]);
    int p = M.dof_parent[j];
    if (p >= 0) {
      F[p].n += F[j].n; F[p].f += F[j].f;
      Ic[p].m += Ic[j].m; Ic[p].h += Ic[j].h;
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Ic[p].I.m[a][b] += Ic[j].I.m[a][b];
    }
  }
  for (int j = 0; j < n; j++) {
    SpatialF Fc = imul(Ic[j], Sj[j]);
    Mq[j * n + j] = sdot(Sj[j], Fc) + M.dof_armature[j];
    for (int i = M.dof_parent[j]; i >= 0; i = M.dof_parent[i]) Mq[j * n + i] = Mq[i * n + j] = sdot(Sj[i], Fc);
  }
  // drives, tendons (implicit)
  std::vector<Real> A(Mq), rhs(n), rhs0(n), tau0(n), Dj(n), kp(n), kd(n);
  for (int j = 0; j < n; j++) {
    kp[j] = M.dof_drive[4 * j]; kd[j] = M.dof_drive[4 * j + 1];
    if ((int)M.dof_drive[4 * j + 3] == MSSIM_DRIVE_ACCELERATION) { kp[j] *= Mq[j * n + j]; kd[j] *= Mq[j * n + j]; }
    tau0[j] = kp[j] * (E.qt[j] - E.q[j]) + kd[j] * E.qdt[j];
    Dj[j] = dt * kd[j] + dt * dt * kp[j];
    A[j * n + j] += Dj[j];
  }
  std::vector<Real> tau_t(n, Real(0));
  for (int t = 0; t < M.n_tendon; t++) {
    int a = M.tendon_dof[2 * t], b = M.tendon_dof[2 * t + 1];
    const float* tp = &M.tendon_param[5 * t];
    Real ca = tp[0], cb = tp[1], c = ca * E.q[a] + cb * E.q[b] - tp[2];
    Real k = tp[3], d = tp[4], w = dt * dt * k + dt * d;
    tau_t[a] -= k * c * ca; tau_t[b] -= k * c * cb;
    A[a * n + a] += w * ca * ca; A[b * n + b] += w * cb * cb;
    A[a * n + b] += w * ca * cb; A[b * n + a] += w * ca * cb;
  }
  for (int j = 0; j < n; j++) {
    Real mv = 0;
    for (int k = 0; k < n; k++) mv += Mq[j * n + k] * E.qd[k];
    rhs0[j] = mv + dt * (tau_t[j] - bias[j] + E.qf[j]);  // without the drive torque: a saturated joint swaps it, no cancellation
    rhs[j] = rhs0[j] + dt * tau0[j];
  }
  std::vector<Real> L(A), qds(rhs);
  std::vector<Real> Ainv(n * n, Real(0));
  if (n > 0) {
    ldl_factor(n, L);
    ldl_solve(n, L, qds.data());
    // force-limit active set (one pass)
    bool any = false;
    for (int j = 0; j < n; j++) {
      Real fmax = M.dof_drive[4 * j + 2];
      if (!(fmax < Real(1e30))) continue;
      Real td = kp[j] * (E.qt[j] - E.q[j] - dt * qds[j]) + kd[j] * (E.qdt[j] - qds[j]);
      if (std::fabs(td) > fmax) {
        Real sat = td > 0 ? fmax : -fmax;
        A[j * n + j] -= Dj[j];
        rhs[j] = rhs0[j] + dt * sat;
        any = true;
      }
    }
    if (any) {
      L = A; qds = rhs;
      ldl_factor(n, L);
      ldl_solve(n, L, qds.data());
    }
    for (int c = 0; c < n; c++) {
      std::vector<Real> e(n, Real(0));
      e[c] = 1;
      ldl_solve(n, L, e.data());
      for (int r = 0; r < n; r++) Ainv[r * n + c] = e[r];
    }
  }
  // free bodies: unconstrained velocity
  std::vector<Vec> fv(nf), fw(nf), fcom(nf);
  std::vector<Mat> fIinv(nf);
  std::vector<Real> fminv(nf);
  for (int b = 0; b < nf; b++) {
    const float* in = &finert[10 * b];
    Mat Rm = qmat(E.free_pose[b].q);
    Real Iv[6] = {in[4], in[5], in[6], in[7], in[8], in[9]};
    Mat Iw = mmul(mmul(Rm, sym3(Iv)), mtranspose(Rm));
    fIinv[b] = minverse(Iw);
    fminv[b] = Real(1) / in[0];
    fcom[b] = E.free_pose[b].p + Rm * Vec(in[1], in[2], in[3]);
    Vec acc = E.free_force[b] * fminv[b];
    if (M.free_gravity[b]) acc += g;
    Vec v = E.free_v[b] + acc * dt;
    const Vec w0 = clamp_norm(E.free_w[b], Real(MSSIM_MAX_ANGULAR_VELOCITY));  // include/mssim.h
    Vec w = w0 - fIinv[b] * cross(w0, Iw * w0) * dt;
    Real ld = Real(1) - dt * M.free_damping[2 * b], ad = Real(1) - dt * M.free_damping[2 * b + 1];
    fv[b] = v * (ld > 0 ? ld : Real(0));
    fw[b] = w * (ad > 0 ? ad : Real(0));
    if (E.free_wake[b] <= 0) {  // asleep: at rest, out of the solver (none of its manifolds was kept)
      fv[b] = Vec(); fw[b] = Vec();
      fminv[b] = 0; fIinv[b] = Mat();
    }
  }

  // 4. rows
  std::vector<Row> rows;
  auto art_point_jac = [&](int body, const Vec& x, const Vec& d, Real sign, std::vector<Real>& Ja) {
    for (int i = body; i >= 0; i = M.dof_parent[i]) {
      Vec col = M.dof_type[i] == MSSIM_JOINT_REVOLUTE ? cross(axis_w[i], x - anchor[i]) : axis_w[i];
      Ja[i] += sign * dot(d, col);
    }
  };
  for (int j = 0; j < n; j++) {
    Real lo = M.dof_limit[2 * j], hi = M.dof_limit[2 * j + 1];
    if (!(lo > Real(-1e30)) && !(hi < Real(1e30))) continue;
    Real dlo = E.q[j] - lo, dhi = hi - E.q[j];
    Real side = dlo <= dhi ? Real(1) : Real(-1);
    Real C = dlo <= dhi ? dlo : dhi;
    Row r;
    r.Ja.assign(n, 0); r.Wa.assign(n, 0);
    r.Ja[j] = side;
    for (int k = 0; k < n; k++) r.Wa[k] = side * Ainv[k * n + j];
    r.f[0] = r.f[1] = -1;
    r.diag = Ainv[j * n + j];
    r.bias_pos = C >= 0 ? C / dt : std::max(M.erp * C / dt, -M.max_depen);
    r.bias_vel = C >= 0 ? C / dt : Real(0);
    r.lo = 0; r.hi = Real(1e30); r.lam = 0; r.friction_of = -1; r.mu = 0; r.contact = -1; r.dirk = -1;
    rows.push_back(r);
  }
  const size_t n_limit_rows = rows.size();
  std::vector<int> contact_row(contacts.size(), -1);  // normal row of every contact
  for (size_t ci = 0; ci < contacts.size(); ci++) {
    Contact& c = contacts[ci];
    contact_row[ci] = (int)rows.size();
    Vec nrm = c.n;
    Vec t1 = std::fabs(nrm.x) < Real(0.57735) ? normalized(cross(nrm, Vec(1, 0, 0))) : normalized(cross(nrm, Vec(0, 1, 0)));
    Vec t2 = cross(nrm, t1);
    Vec dirs[3] = {nrm, t1, t2};
    int nrow = (int)rows.size();
    for (int k = 0; k < 3; k++) {
      Row r;
      r.Ja.assign(n, 0); r.Wa.assign(n, 0);
      r.f[0] = r.f[1] = -1;
      int nfree = 0;
      bool hasart = false;
      const int kinds[2] = {c.ka, c.kb}, idx[2] = {c.ia, c.ib};
      for (int s = 0; s < 2; s++) {
        Real sign = s == 0 ? Real(1) : Real(-1);
        if (kinds[s] == MSSIM_BODY_ART && idx[s] >= 0) { art_point_jac(idx[s], c.x, dirs[k], sign, r.Ja); hasart = true; }
        else if (kinds[s] == MSSIM_BODY_FREE) {
          int b = idx[s];
          r.f[nfree] = b;
          r.Jl[nfree] = dirs[k] * sign;
          r.Jw[nfree] = cross(c.x - fcom[b], dirs[k]) * sign;
          r.Wl[nfree] = r.Jl[nfree] * fminv[b];
          r.Ww[nfree] = fIinv[b] * r.Jw[nfree];
          nfree++;
        }
      }
      Real diag = 0;
      if (hasart)
        for (int i = 0; i < n; i++) {
          Real w = 0;
          for (int j = 0; j < n; j++) w += Ainv[i * n + j] * r.Ja[j];
          r.Wa[i] = w;
          diag += r.Ja[i] * w;
        }
      for (int s = 0; s < nfree; s++) diag += dot(r.Jl[s], r.Wl[s]) + dot(r.Jw[s], r.Ww[s]);
      r.diag = diag;
      r.lam = 0; r.contact = (int)ci; r.dirk = k; r.mu = c.mu;
      if (k == 0) {
        r.bias_pos = c.sep >= 0 ? c.sep / dt : std::max(M.erp * c.sep / dt, -M.max_depen);
        r.bias_vel = c.sep >= 0 ? c.sep / dt : Real(0);
        r.lo = 0; r.hi = Real(1e30); r.friction_of = -1;
      } else {
        r.bias_pos = r.bias_vel = 0; r.lo = r.hi = 0; r.friction_of = nrow;
      }
      rows.push_back(r);
    }
    if (c.tors_mu > 0) {
      // torsional friction row of the manifold that ends with this contact: relative angular velocity about the normal
      Row r;
      r.Ja.assign(n, 0); r.Wa.assign(n, 0);
      r.f[0] = r.f[1] = -1;
      int nfree = 0;
      bool hasart = false;
      const int kinds[2] = {c.ka, c.kb}, idx[2] = {c.ia, c.ib};
      for (int s = 0; s < 2; s++) {
        Real sign = s == 0 ? Real(1) : Real(-1);
        if (kinds[s] == MSSIM_BODY_ART && idx[s] >= 0) {
          for (int i = idx[s]; i >= 0; i = M.dof_parent[i])
            if (M.dof_type[i] == MSSIM_JOINT_REVOLUTE) r.Ja[i] += sign * dot(c.tors_n, axis_w[i]);
          hasart = true;
        } else if (kinds[s] == MSSIM_BODY_FREE) {
          int b = idx[s];
          r.f[nfree] = b;
          r.Jl[nfree] = Vec();
          r.Jw[nfree] = c.tors_n * sign;
          r.Wl[nfree] = Vec();
          r.Ww[nfree] = fIinv[b] * r.Jw[nfree];
          nfree++;
        }
      }
      Real diag = 0;
      if (hasart)
        for (int i = 0; i < n; i++) {
          Real w = 0;
          for (int j = 0; j < n; j++) w += Ainv[i * n + j] * r.Ja[j];
          r.Wa[i] = w;
          diag += r.Ja[i] * w;
        }
      for (int s = 0; s < nfree; s++) diag += dot(r.Jw[s], r.Ww[s]);
      r.diag = diag;
      r.lam = 0; r.contact = -2; r.dirk = -1; r.mu = c.tors_mu;
      r.bias_pos = r.bias_vel = 0; r.lo = r.hi = 0; r.friction_of = -1;
      for (size_t cj = (size_t)c.tors_first; cj <= ci; cj++) r.tors_of.push_back(contact_row[cj]);
      rows.push_back(r);
    }
  }

  // 5. PGS
  std::vector<Real> v(qds);
  // Warm start: a contact whose (shape pair, manifold slot) carried multipliers in the PREVIOUS substep starts from
  // them -- the rows' initial impulses are applied to the velocities before the first sweep. Torsional rows and limit
  // rows start from zero.
  if ((int)E.warm.size() != 4 * M.n_pair) E.warm.assign((size_t)4 * M.n_pair, EnvState::Warm{{0, 0, 0}, -1});
  if (!M.cold)
    for (size_t ci = 0; ci < contacts.size(); ci++) {
      if (contacts[ci].slot < 0) continue;
      const EnvState::Warm& w = E.warm[4 * contacts[ci].pair + contacts[ci].slot];
      if (w.stamp != E.pcm_tick - 1) continue;
      for (int d = 0; d < 3; d++) {
        Row& r = rows[contact_row[ci] + d];
        const Real dl = w.lam[d];
        r.lam = dl;
        for (int i = 0; i < n; i++) v[i] += r.Wa[i] * dl;
        for (int s2 = 0; s2 < 2; s2++)
          if (r.f[s2] >= 0) { fv[r.f[s2]] += r.Wl[s2] * dl; fw[r.f[s2]] += r.Ww[s2] * dl; }
      }
    }
  // one Gauss-Seidel sweep: the contact rows in pair order, then the joint-limit rows (an articulation's internal
  // constraints are solved after its contacts, as in PhysX: a jammed arm gives way at the contact, not at the limit)
  auto sweep = [&](bool use_bias) {
    for (size_t k = 0; k < rows.size(); k++) {
      const size_t ri = k < rows.size() - n_limit_rows ? k + n_limit_rows : k - (rows.size() - n_limit_rows);
      Row& r = rows[ri];
      if (!(r.diag > Real(1e-12))) continue;
      Real jv = 0;
      for (int i = 0; i < n; i++) jv += r.Ja[i] * v[i];
      for (int s = 0; s < 2; s++)
        if (r.f[s] >= 0) jv += dot(r.Jl[s], fv[r.f[s]]) + dot(r.Jw[s], fw[r.f[s]]);
      Real lo = r.lo, hi = r.hi;
      if (r.friction_of >= 0) { hi = r.mu * rows[r.friction_of].lam; lo = -hi; }
      if (!r.tors_of.empty()) {
        Real nsum = 0;
        for (int k2 : r.tors_of) nsum += rows[k2].lam;
        hi = r.mu * nsum; lo = -hi;
      }
      Real b = use_bias ? r.bias_pos : r.bias_vel;
      Real nl = r.lam - (jv + b) / r.diag;
      nl = nl < lo ? lo : (nl > hi ? hi : nl);
      Real dl = nl - r.lam;
      r.lam = nl;
      if (dl != Real(0)) {
        for (int i = 0; i < n; i++) v[i] += r.Wa[i] * dl;
        for (int s = 0; s < 2; s++)
          if (r.f[s] >= 0) { fv[r.f[s]] += r.Wl[s] * dl; fw[r.f[s]] += r.Ww[s] * dl; }
      }
    }
  };
  std::vector<Real> v_pos;
  std::vector<Vec> fv_pos, fw_pos;
  if (!M.tgs) {
    for (int it = 0; it < M.pos_iters; it++) {
      const std::vector<Real> v0 = v;
      const std::vector<Vec> fv0 = fv, fw0 = fw;
      sweep(true);
      // early exit (include/mssim.h MSSIM_PGS_EXIT_TOLERANCE): no velocity component moved by more than the tolerance
      Real mx = 0;
      for (int i = 0; i < n; i++) mx = std::max(mx, std::fabs(v[i] - v0[i]));
      for (int b = 0; b < nf; b++) {
        const Vec d = fv[b] - fv0[b], r = fw[b] - fw0[b];
        mx = std::max(mx, std::max(std::max(std::fabs(d.x), std::fabs(d.y)), std::fabs(d.z)));
        mx = std::max(mx, std::max(std::max(std::fabs(r.x), std::fabs(r.y)), std::fabs(r.z)));
      }
      if (!(mx > Real(MSSIM_PGS_EXIT_TOLERANCE))) break;
    }
    v_pos = v; fv_pos = fv; fw_pos = fw;
  } else {
    // TGS-style position sub-stepping (the experiment behind DESIGN.md's PGS-vs-TGS comparison; never used by the
    // parity tests): the substep is cut into pos_iters slices of h = dt / pos_iters. Every slice makes ONE Gauss-Seidel
    // sweep whose position bias is the row's CURRENT gap (the initial gap plus the relative displacement of the earlier
    // slices) over h, then advances the rows' gaps by h (J v). Positions integrate with the mean of the slice velocities.
    const int n_it = M.pos_iters > 0 ? M.pos_iters : 1;
    const Real h = dt / n_it;
    std::vector<Real> gap(rows.size());
    for (size_t k = 0; k < rows.size(); k++) {
      const Row& r = rows[k];
      // recover the signed gap from the PGS biases: bias_pos = gap / dt (gap >= 0) or erp * gap / dt (capped) otherwise
      gap[k] = r.bias_vel > 0 || r.bias_pos >= 0 ? r.bias_pos * dt : r.bias_pos * dt / M.erp;
    }
    v_pos.assign(n, 0); fv_pos.assign(nf, Vec()); fw_pos.assign(nf, Vec());
    for (int it = 0; it < n_it; it++) {
      for (size_t k = 0; k < rows.size(); k++) {
        Row& r = rows[k];
        if (r.friction_of >= 0 || !r.tors_of.empty()) continue;
        r.bias_pos = gap[k] >= 0 ? gap[k] / h : std::max(M.erp * gap[k] / h, -M.max_depen);
      }
      sweep(true);
      for (size_t k = 0; k < rows.size(); k++) {
        const Row& r = rows[k];
        if (r.friction_of >= 0 || !r.tors_of.empty() || !(r.diag > Real(1e-12))) continue;
        Real jv = 0;
        for (int i = 0; i < n; i++) jv += r.Ja[i] * v[i];
        for (int s2 = 0; s2 < 2; s2++)
          if (r.f[s2] >= 0) jv += dot(r.Jl[s2], fv[r.f[s2]]) + dot(r.Jw[s2], fw[r.f[s2]]);
        gap[k] += h * jv;
      }
      for (int i = 0; i < n; i++) v_pos[i] += v[i] / n_it;
      for (int b = 0; b < nf; b++) { fv_pos[b] += fv[b] * (Real(1) / n_it); fw_pos[b] += fw[b] * (Real(1) / n_it); }
    }
  }
  for (int it = 0; it < M.vel_iters; it++) sweep(false);

  for (size_t ci = 0; ci < contacts.size(); ci++) {
    if (contacts[ci].slot < 0) continue;
    EnvState::Warm& w = E.warm[4 * contacts[ci].pair + contacts[ci].slot];
    for (int d = 0; d < 3; d++) w.lam[d] = rows[contact_row[ci] + d].lam;
    w.stamp = E.pcm_tick;
  }
  // contact impulses per pair (world frame, on shape A's body)
  E.pair_impulse.assign(M.n_pair, Vec());
  for (const Row& r : rows) {
    if (r.contact < 0) continue;
    const Contact& c = contacts[r.contact];
    Vec nrm = c.n;
    Vec t1 = std::fabs(nrm.x) < Real(0.57735) ? normalized(cross(nrm, Vec(1, 0, 0))) : normalized(cross(nrm, Vec(0, 1, 0)));
    Vec t2 = cross(nrm, t1);
    Vec d = r.dirk == 0 ? nrm : (r.dirk == 1 ? t1 : t2);
    E.pair_impulse[c.pair] += d * r.lam;
  }

  // 6. integrate
  for (int j = 0; j < n; j++) {
    const Real vmax = Real(MSSIM_MAX_JOINT_VELOCITY);  // include/mssim.h
    const Real vj = std::min(std::max(v[j], -vmax), vmax);
    E.qacc[j] = (vj - E.qd[j]) / dt;
    E.q[j] += dt * std::min(std::max(v_pos[j], -vmax), vmax);
    E.qd[j] = vj;
  }
  for (int b = 0; b < nf; b++) {
    const float* in = &finert[10 * b];
    if (E.free_wake[b] <= 0) { E.free_force[b] = Vec(); continue; }  // asleep: pose and (zero) velocity stay bit for bit
    Vec com = fcom[b] + fv_pos[b] * dt;
    Quat q = E.free_pose[b].q;
    Vec w = clamp_norm(fw_pos[b], Real(MSSIM_MAX_ANGULAR_VELOCITY));
    Quat dq = qmul(Quat(0, w.x, w.y, w.z), q);
    q = qnormalized(Quat(q.w + Real(0.5) * dt * dq.w, q.x + Real(0.5) * dt * dq.x, q.y + Real(0.5) * dt * dq.y,
                         q.z + Real(0.5) * dt * dq.z));
    E.free_pose[b].q = q;
    E.free_pose[b].p = com - qrot(q, Vec(in[1], in[2], in[3]));
    E.free_v[b] = fv[b];
    E.free_w[b] = fw[b];
    E.free_force[b] = Vec();
    // sleep counter: runs down while the body is calm and no disturber touches it, restarts otherwise
    if (E.free_wake[b] > 0) {
      const bool calm = M.sleep_threshold > 0 && normalised_energy(b, E.free_v[b], E.free_w[b]) < M.sleep_threshold;
      E.free_wake[b] = (calm && !E.free_disturbed[b]) ? E.free_wake[b] - dt : Real(MSSIM_WAKE_TIME);
      if (E.free_wake[b] <= 0) { E.free_wake[b] = 0; E.free_v[b] = Vec(); E.free_w[b] = Vec(); }
    }
  }
  fk(M, E, &axis_w, &anchor);
  body_velocities(M, E, axis_w, anchor, E.qd);
}

void refresh_kinematics(mssim_sim* S, EnvState& E) {
  std::vector<Vec> axis_w, anchor;
  fk(S->M, E, &axis_w, &anchor);
  body_velocities(S->M, E, axis_w, anchor, E.qd);
}

}  // namespace

// ===================================================================== C ABI
extern "C" {

int mssim_ref_abi_version(void) { return MSSIM_ABI_VERSION; }

const char* mssim_ref_last_error(mssim_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mssim_ref_create(const mssim_model_desc* d, int32_t num_envs, int32_t device, mssim_handle* out) {
  (void)device;
  if (!d || !out || num_envs <= 0) { g_create_error = "bad arguments"; return 1; }
  if (d->abi_version != MSSIM_ABI_VERSION) { g_create_error = "ABI version mismatch"; return 2; }
  if (d->n_dof > MSSIM_MAX_DOF || d->n_free > MSSIM_MAX_FREE) { g_create_error = "model exceeds MSSIM_MAX_DOF / MSSIM_MAX_FREE"; return 3; }
  mssim_sim* S = new mssim_sim();
  Model& M = S->M;
  M.n_dof = d->n_dof; M.n_tendon = d->n_tendon; M.n_link = d->n_link; M.n_free = d->n_free; M.n_kin = d->n_kin;
  M.n_shape = d->n_shape; M.n_pair = d->n_pair;
  int n = d->n_dof;
  M.dof_parent = cp(d->dof_parent, n); M.dof_type = cp(d->dof_type, n); M.body_gravity = cp(d->body_gravity, n);
  M.dof_frame = cp(d->dof_frame, 7 * n); M.dof_axis = cp(d->dof_axis, 3 * n); M.dof_limit = cp(d->dof_limit, 2 * n);
  M.dof_drive = cp(d->dof_drive, 4 * n); M.dof_armature = cp(d->dof_armature, n); M.body_inertial = cp(d->body_inertial, 10 * n);
  M.tendon_dof = cp(d->tendon_dof, 2 * d->n_tendon); M.tendon_param = cp(d->tendon_param, 5 * d->n_tendon);
  M.link_body = cp(d->link_body, d->n_link); M.link_frame = cp(d->link_frame, 7 * d->n_link);
  M.free_inertial = cp(d->free_inertial, 10 * d->n_free); M.free_damping = cp(d->free_damping, 2 * d->n_free);
  M.free_gravity = cp(d->free_gravity, d->n_free);
  int ns = d->n_shape;
  M.shape_type = cp(d->shape_type, ns); M.shape_kind = cp(d->shape_body_kind, ns); M.shape_index = cp(d->shape_body_index, ns);
  M.shape_row = cp(d->shape_row, ns); M.shape_frame = cp(d->shape_frame, 7 * ns); M.shape_param = cp(d->shape_param, 4 * ns);
  M.shape_material = cp(d->shape_material, 4 * ns); M.shape_hull = cp(d->shape_hull, 2 * ns); M.shape_bound = cp(d->shape_bound, 4 * ns);
  M.hull_verts = cp(d->hull_verts, 3 * d->n_hull_verts); M.pair_shape = cp(d->pair_shape, 2 * d->n_pair);
  M.tri_soup = cp(d->tri_soup, (size_t)12 * d->n_tri);  // (the BVH is an acceleration structure of the kernels: not used here)
  M.N = num_envs;
  M.n_env_shape = d->n_env_shape; M.n_env_free = d->n_env_free;
  if ((d->n_env_shape > 0 || d->n_env_free > 0) && d->num_envs != num_envs) { g_create_error = "per-env arrays were built for a different num_envs"; delete S; return 5; }
  M.shape_env_slot = cp(d->shape_env_slot, ns); M.free_env_slot = cp(d->free_env_slot, d->n_free);
  if (d->n_env_shape == 0) std::fill(M.shape_env_slot.begin(), M.shape_env_slot.end(), -1);
  if (d->n_env_free == 0) std::fill(M.free_env_slot.begin(), M.free_env_slot.end(), -1);
  M.env_shape_frame = cp(d->env_shape_frame, (size_t)7 * d->n_env_shape * num_envs);
  M.env_shape_param = cp(d->env_shape_param, (size_t)4 * d->n_env_shape * num_envs);
  M.env_shape_bound = cp(d->env_shape_bound, (size_t)4 * d->n_env_shape * num_envs);
  M.env_free_inertial = cp(d->env_free_inertial, (size_t)10 * d->n_env_free * num_envs);
  for (int k = 0; k < 3; k++) M.gravity[k] = d->gravity[k];
  M.dt = d->timestep; M.contact_offset = d->contact_offset; M.rest_offset = d->rest_offset; M.erp = d->erp;
  M.max_depen = d->max_depenetration_velocity; M.pos_iters = d->position_iterations; M.vel_iters = d->velocity_iterations;
  M.sleep_threshold = d->sleep_threshold;
  { const char* t = getenv("MSSIM_REF_TGS"); M.tgs = t && t[0] == '1'; }
  { const char* t = getenv("MSSIM_REF_COLD"); M.cold = t && t[0] == '1'; }
  for (int j = 0; j < n; j++)
    if (M.dof_parent[j] >= j) { g_create_error = "dof_parent must be topologically sorted"; delete S; return 4; }
  S->N = num_envs;
  S->env.resize(num_envs);
  for (auto& E : S->env) {
    E.q.assign(n, 0); E.qd.assign(n, 0); E.qt.assign(n, 0); E.qdt.assign(n, 0); E.qf.assign(n, 0); E.qacc.assign(n, 0);
    E.free_pose.resize(M.n_free); E.free_v.resize(M.n_free); E.free_w.resize(M.n_free); E.free_force.resize(M.n_free); E.free_wake.assign(M.n_free, Real(MSSIM_WAKE_TIME)); E.free_calm.assign(M.n_free, 0); E.free_disturbed.assign(M.n_free, 0);
    E.kin_pose.resize(M.n_kin);
    E.pair_impulse.assign(M.n_pair, Vec()); E.pair_count.assign(M.n_pair, 0);
    refresh_kinematics(S, E);
  }
  *out = S;
  return 0;
}

void mssim_ref_destroy(mssim_handle h) { delete h; }

int mssim_ref_bind_buffers(mssim_handle h, const mssim_buffers* b) {
  if (!h || !b) return 1;
  h->buf = *b;
  return 0;
}

int mssim_ref_set_timestep(mssim_handle h, float dt) {
  if (!(dt > 0)) { h->err = "timestep must be positive"; return 1; }
  h->M.dt = dt;
  return 0;
}
float mssim_ref_get_timestep(mssim_handle h) { return (float)h->M.dt; }

int mssim_ref_set_drive_properties(mssim_handle h, const float* drive) {
  h->M.dof_drive.assign(drive, drive + 4 * h->M.n_dof);
  return 0;
}

int mssim_ref_apply(mssim_handle h, uint32_t what, void*) {
  const Model& M = h->M;
  const int N = h->N, n = M.n_dof;
  const mssim_buffers& B = h->buf;
  for (int e = 0; e < N; e++) {
    EnvState& E = h->env[e];
    if ((what & MSSIM_RIGID_DATA) && B.rigid_body_data) {
      for (int b = 0; b < M.n_free; b++) {
        const float* r = B.rigid_body_data + 13 * ((size_t)(M.n_link + b) * N + e);
        // a row that differs from what the last fetch wrote (the f32 image of the state) wakes the body
        const Pose<Real>& P0 = E.free_pose[b];
        const float was[13] = {(float)P0.p.x, (float)P0.p.y, (float)P0.p.z, (float)P0.q.w, (float)P0.q.x, (float)P0.q.y, (float)P0.q.z,
                               (float)E.free_v[b].x, (float)E.free_v[b].y, (float)E.free_v[b].z, (float)E.free_w[b].x, (float)E.free_w[b].y, (float)E.free_w[b].z};
        bool changed = false;
        for (int k = 0; k < 13; k++) changed = changed || r[k] != was[k];
        if (!changed) continue;
        E.free_wake[b] = Real(MSSIM_WAKE_TIME);
        E.free_pose[b] = pose7(r);
        E.free_v[b] = Vec(r[7], r[8], r[9]);
        E.free_w[b] = Vec(r[10], r[11], r[12]);
      }
      // a kinematic body given a different pose (than the f32 image the last fetch wrote) wakes the env's free bodies
      bool kin_moved = false;
      for (int k = 0; k < M.n_kin; k++) {
        const float* r = B.rigid_body_data + 13 * ((size_t)(M.n_link + M.n_free + k) * N + e);
        const Pose<Real>& P0 = E.kin_pose[k];
        const float was[7] = {(float)P0.p.x, (float)P0.p.y, (float)P0.p.z, (float)P0.q.w, (float)P0.q.x, (float)P0.q.y, (float)P0.q.z};
        bool changed = false;
        for (int c = 0; c < 7; c++) changed = changed || r[c] != was[c];
        if (!changed) continue;
        kin_moved = true;
        E.kin_pose[k] = pose7(r);
      }
      if (kin_moved) std::fill(E.free_wake.begin(), E.free_wake.end(), Real(MSSIM_WAKE_TIME));
    }
    if ((what & MSSIM_ART_ROOT_POSE) && B.rigid_body_data && M.n_link > 0) E.root = pose7(B.rigid_body_data + 13 * (size_t)e);
    if ((what & MSSIM_ART_QPOS) && B.art_qpos) for (int j = 0; j < n; j++) E.q[j] = B.art_qpos[(size_t)e * n + j];
    if ((what & MSSIM_ART_QVEL) && B.art_qvel) for (int j = 0; j < n; j++) E.qd[j] = B.art_qvel[(size_t)e * n + j];
    if ((what & MSSIM_ART_QF) && B.art_qf) for (int j = 0; j < n; j++) E.qf[j] = B.art_qf[(size_t)e * n + j];
    if ((what & MSSIM_ART_TARGET_POS) && B.art_target_qpos) for (int j = 0; j < n; j++) E.qt[j] = B.art_target_qpos[(size_t)e * n + j];
    if ((what & MSSIM_ART_TARGET_VEL) && B.art_target_qvel) for (int j = 0; j < n; j++) E.qdt[j] = B.art_target_qvel[(size_t)e * n + j];
    if ((what & MSSIM_RIGID_FORCE) && B.rigid_body_force)
      for (int b = 0; b < M.n_free; b++) {
        const float* f = B.rigid_body_force + 4 * ((size_t)(M.n_link + b) * N + e);
        E.free_force[b] = Vec(f[0], f[1], f[2]);
        if (f[0] != 0 || f[1] != 0 || f[2] != 0) E.free_wake[b] = Real(MSSIM_WAKE_TIME);
      }
  }
  return 0;
}

int mssim_ref_fetch(mssim_handle h, uint32_t what, void*) {
  const Model& M = h->M;
  const int N = h->N, n = M.n_dof;
  const mssim_buffers& B = h->buf;
  for (int e = 0; e < N; e++) {
    EnvState& E = h->env[e];
    if (B.rigid_body_data) {
      if (what & MSSIM_RIGID_DATA) {
        for (int b = 0; b < M.n_free; b++) {
          float* r = B.rigid_body_data + 13 * ((size_t)(M.n_link + b) * N + e);
          const Pose<Real>& P = E.free_pose[b];
          r[0] = P.p.x; r[1] = P.p.y; r[2] = P.p.z; r[3] = P.q.w; r[4] = P.q.x; r[5] = P.q.y; r[6] = P.q.z;
          r[7] = E.free_v[b].x; r[8] = E.free_v[b].y; r[9] = E.free_v[b].z;
          r[10] = E.free_w[b].x; r[11] = E.free_w[b].y; r[12] = E.free_w[b].z;
        }
        for (int k = 0; k < M.n_kin; k++) {
          float* r = B.rigid_body_data + 13 * ((size_t)(M.n_link + M.n_free + k) * N + e);
          const Pose<Real>& P = E.kin_pose[k];
          r[0] = P.p.x; r[1] = P.p.y; r[2] = P.p.z; r[3] = P.q.w; r[4] = P.q.x; r[5] = P.q.y; r[6] = P.q.z;
          for (int c = 7; c < 13; c++) r[c] = 0;
        }
      }
      if (what & (MSSIM_LINK_POSE | MSSIM_LINK_VEL)) {
        Vec O = E.root.p;
        for (int l = 0; l < M.n_link; l++) {
          float* r = B.rigid_body_data + 13 * ((size_t)l * N + e);
          int b = M.link_body[l];
          Pose<Real> P = pmul(b < 0 ? E.root : E.body_pose[b], pose7(&M.link_frame[7 * l]));
          if (what & MSSIM_LINK_POSE) { r[0] = P.p.x; r[1] = P.p.y; r[2] = P.p.z; r[3] = P.q.w; r[4] = P.q.x; r[5] = P.q.y; r[6] = P.q.z; }
          if (what & MSSIM_LINK_VEL) {
            Vec w, v;
            if (b >= 0) { w = E.body_vel[b].w; v = E.body_vel[b].v + cross(w, P.p - O); }
            r[7] = v.x; r[8] = v.y; r[9] = v.z; r[10] = w.x; r[11] = w.y; r[12] = w.z;
          }
        }
      }
    }
    if ((what & MSSIM_ART_QPOS) && B.art_qpos) for (int j = 0; j < n; j++) B.art_qpos[(size_t)e * n + j] = (float)E.q[j];
    if ((what & MSSIM_ART_QVEL) && B.art_qvel) for (int j = 0; j < n; j++) B.art_qvel[(size_t)e * n + j] = (float)E.qd[j];
    if ((what & MSSIM_ART_QACC) && B.art_qacc) for (int j = 0; j < n; j++) B.art_qacc[(size_t)e * n + j] = (float)E.qacc[j];
    if ((what & MSSIM_ART_TARGET_POS) && B.art_target_qpos) for (int j = 0; j < n; j++) B.art_target_qpos[(size_t)e * n + j] = (float)E.qt[j];
    if ((what & MSSIM_ART_TARGET_VEL) && B.art_target_qvel) for (int j = 0; j < n; j++) B.art_target_qvel[(size_t)e * n + j] = (float)E.qdt[j];
  }
  return 0;
}

int mssim_ref_step(mssim_handle h, int32_t n_substeps, void*) {
  // envs are independent (per-env state only; the model is read-only): parallel over envs when
  // built with -fopenmp (used for the multi-core cpu_baseline of bench.py)
  const int N = h->N;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < N; e++)
    for (int s = 0; s < n_substeps; s++) substep(h, h->env[e], e);
  return 0;
}

int mssim_ref_wake_all(mssim_handle h, void*) {
  for (auto& E : h->env) {
    std::fill(E.free_wake.begin(), E.free_wake.end(), Real(MSSIM_WAKE_TIME));
    for (auto& s : E.pcm) s = PcmSlot();
    E.warm.clear();
  }
  return 0;
}

int mssim_ref_wake_envs(mssim_handle h, const int64_t* env_idx, int32_t n_idx, void*) {
  for (int i = 0; i < n_idx; i++) {
    if (env_idx[i] < 0 || env_idx[i] >= h->N) continue;
    EnvState& E = h->env[(size_t)env_idx[i]];
    std::fill(E.free_wake.begin(), E.free_wake.end(), Real(MSSIM_WAKE_TIME));
    for (auto& s : E.pcm) s = PcmSlot();
    E.warm.clear();
  }
  return 0;
}

int mssim_ref_update_kinematics(mssim_handle h, void*) {
  for (auto& E : h->env) refresh_kinematics(h, E);
  return 0;
}

int mssim_ref_create_pair_query(mssim_handle h, const int32_t* body_pairs, int32_t n_pairs, int32_t* qid) {
  h->pair_queries.emplace_back(body_pairs, body_pairs + 2 * n_pairs);
  *qid = (int)h->pair_queries.size() - 1;
  return 0;
}

int mssim_ref_query_pair_impulses(mssim_handle h, int32_t qid, float* out, void*) {
  if (qid < 0 || qid >= (int)h->pair_queries.size()) { h->err = "bad query id"; return 1; }
  const Model& M = h->M;
  const auto& Q = h->pair_queries[qid];
  int nq = (int)Q.size() / 2;
  for (int k = 0; k < nq; k++)
    for (int e = 0; e < h->N; e++) {
      Vec s;
      for (int p = 0; p < M.n_pair; p++) {
        int ra = M.shape_row[M.pair_shape[2 * p]], rb = M.shape_row[M.pair_shape[2 * p + 1]];
        if (ra == Q[2 * k] && rb == Q[2 * k + 1]) s += h->env[e].pair_impulse[p];
        else if (ra == Q[2 * k + 1] && rb == Q[2 * k]) s -= h->env[e].pair_impulse[p];
      }
      float* o = out + 3 * ((size_t)k * h->N + e);
      o[0] = (float)s.x; o[1] = (float)s.y; o[2] = (float)s.z;
    }
  return 0;
}

int mssim_ref_create_body_query(mssim_handle h, const int32_t* rows, int32_t n, int32_t* qid) {
  h->body_queries.emplace_back(rows, rows + n);
  *qid = (int)h->body_queries.size() - 1;
  return 0;
}

int mssim_ref_query_body_impulses(mssim_handle h, int32_t qid, float* out, void*) {
  if (qid < 0 || qid >= (int)h->body_queries.size()) { h->err = "bad query id"; return 1; }
  const Model& M = h->M;
  const auto& Q = h->body_queries[qid];
  for (size_t k = 0; k < Q.size(); k++)
    for (int e = 0; e < h->N; e++) {
      Vec s;
      for (int p = 0; p < M.n_pair; p++) {
        int ra = M.shape_row[M.pair_shape[2 * p]], rb = M.shape_row[M.pair_shape[2 * p + 1]];
        if (ra == Q[k]) s += h->env[e].pair_impulse[p];
        if (rb == Q[k]) s -= h->env[e].pair_impulse[p];
      }
      float* o = out + 3 * (k * h->N + e);
      o[0] = (float)s.x; o[1] = (float)s.y; o[2] = (float)s.z;
    }
  return 0;
}

int mssim_ref_read_internal(mssim_handle h, const char* name, float* out, int32_t max_items, void*) {
  const Model& M = h->M;
  const int N = h->N;
  std::string s(name);
  int items = 0;
  auto put = [&](int item, int e, Real v) { if (item < max_items) out[(size_t)item * N + e] = (float)v; };
  if (s == "q" || s == "qd") {
    items = M.n_dof;
    for (int e = 0; e < N; e++) for (int j = 0; j < items; j++) put(j, e, s == "q" ? h->env[e].q[j] : h->env[e].qd[j]);
  } else if (s == "free") {
    items = 13 * M.n_free;
    for (int e = 0; e < N; e++)
      for (int b = 0; b < M.n_free; b++) {
        const EnvState& E = h->env[e];
        Real v[13] = {E.free_pose[b].p.x, E.free_pose[b].p.y, E.free_pose[b].p.z, E.free_pose[b].q.w, E.free_pose[b].q.x,
                      E.free_pose[b].q.y, E.free_pose[b].q.z, E.free_v[b].x, E.free_v[b].y, E.free_v[b].z,
                      E.free_w[b].x, E.free_w[b].y, E.free_w[b].z};
        for (int c = 0; c < 13; c++) put(13 * b + c, e, v[c]);
      }
  } else if (s == "bodypose") {
    items = 7 * M.n_dof;
    for (int e = 0; e < N; e++)
      for (int b = 0; b < M.n_dof; b++) {
        const Pose<Real>& P = h->env[e].body_pose[b];
        Real v[7] = {P.p.x, P.p.y, P.p.z, P.q.w, P.q.x, P.q.y, P.q.z};
        for (int c = 0; c < 7; c++) put(7 * b + c, e, v[c]);
      }
  } else if (s == "contact_count") {
    items = M.n_pair;
    for (int e = 0; e < N; e++) for (int p = 0; p < items; p++) put(p, e, h->env[e].pair_count[p]);
  } else if (s == "pair_impulse") {
    items = 3 * M.n_pair;
    for (int e = 0; e < N; e++)
      for (int p = 0; p < M.n_pair; p++) {
        const Vec& v = h->env[e].pair_impulse[p];
        put(3 * p, e, v.x); put(3 * p + 1, e, v.y); put(3 * p + 2, e, v.z);
      }
  } else if (s == "mpr_queries") {  // (oracle only: full convex queries of the last substep)
    items = 1;
    for (int e = 0; e < N; e++) put(0, e, h->env[e].mpr_queries);
  } else if (s == "free_wake") {
    items = M.n_free;
    for (int e = 0; e < N; e++) for (int b = 0; b < items; b++) put(b, e, h->env[e].free_wake[b]);
  } else if (s == "raw_contact_count") {  // (oracle only: test construction aid)
    items = 1;
    for (int e = 0; e < N; e++) put(0, e, h->env[e].raw_points);
  } else if (s == "overflow") {
    items = 1;
    for (int e = 0; e < N; e++) put(0, e, h->env[e].overflow);
  } else {
    h->err = "unknown internal array: " + s;
    return -1;
  }
  return items;
}

// fused callers are an optimisation of the HIP product; the oracle keeps the plain (torch) path
int mssim_ref_set_action_map(mssim_handle h, const int32_t*, const float*, const float*, const int32_t*) { h->err = "not available in the oracle"; return 1; }
// geometric Jacobian of a link in the root frame, from the oracle's own FK (include/mssim.h)
int mssim_ref_link_jacobian(mssim_handle h, int32_t link, float* out, void*) {
  const Model& M = h->M;
  if (link < 0 || link >= M.n_link || !out) { h->err = "link_jacobian: bad link index / output"; return 1; }
  const int n = M.n_dof;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < h->N; e++) {
    EnvState& E = h->env[e];
    std::vector<Vec> axis_w, anchor;
    fk(M, E, &axis_w, &anchor);
    const int b = M.link_body[link];
    const Pose<Real> Pb = b < 0 ? E.root : E.body_pose[b];
    const Vec pe = pmul(Pb, pose7(&M.link_frame[7 * link])).p;
    std::vector<char> on_path(n, 0);
    for (int i = b; i >= 0; i = M.dof_parent[i]) on_path[i] = 1;
    float* o = out + (size_t)e * 6 * n;
    for (int j = 0; j < n; j++) {
      Vec jv, jw;
      if (on_path[j]) {
        if (M.dof_type[j] == MSSIM_JOINT_REVOLUTE) { jv = cross(axis_w[j], pe - anchor[j]); jw = axis_w[j]; }
        else jv = axis_w[j];
        jv = qrot(qconj(E.root.q), jv);
        jw = qrot(qconj(E.root.q), jw);
      }
      o[0 * n + j] = (float)jv.x; o[1 * n + j] = (float)jv.y; o[2 * n + j] = (float)jv.z;
      o[3 * n + j] = (float)jw.x; o[4 * n + j] = (float)jw.y; o[5 * n + j] = (float)jw.z;
    }
  }
  return 0;
}
int mssim_ref_apply_action(mssim_handle h, const float*, int32_t, void*) { h->err = "not available in the oracle"; return 1; }
int mssim_ref_set_ee_action_map(mssim_handle h, int32_t, int32_t, int32_t, float, float, float, int32_t) { h->err = "not available in the oracle"; return 1; }
int mssim_ref_defer_fetch(mssim_handle h, uint32_t) { h->err = "not available in the oracle"; return 1; }
int mssim_ref_defer_step_action(mssim_handle h, const float*, int32_t, int32_t, void*) { h->err = "not available in the oracle"; return 1; }
int mssim_ref_step_action(mssim_handle h, const float*, int32_t, int32_t, void*) { h->err = "not available in the oracle"; return 1; }
int mssim_ref_task_pick_outputs(mssim_handle h, const mssim_pick_task*, float*, float*, uint8_t*, void*) { h->err = "not available in the oracle"; return 1; }
int mssim_ref_task_peg_outputs(mssim_handle h, const mssim_peg_task*, float*, float*, uint8_t*, float*, void*) { h->err = "not available in the oracle"; return 1; }
int mssim_ref_task_push_outputs(mssim_handle h, const mssim_push_task*, float*, float*, uint8_t*, void*) { h->err = "not available in the oracle"; return 1; }
int mssim_ref_profile_enable(mssim_handle, int32_t) { return 0; }
int mssim_ref_profile_read(mssim_handle, float* ms, int32_t* cnt) { ms[0] = ms[1] = 0; cnt[0] = cnt[1] = 0; return 0; }

int mssim_ref_overflow_count(mssim_handle h, void*) {
  int c = 0;
  for (auto& E : h->env) { c += E.overflow != 0; E.overflow = 0; }
  return c;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// test hook (not part of the mssim ABI): run the narrowphase on one explicit shape pair.
// pose = p(3) q(4); out = count, n(3), then 4 x (x(3), sep)
extern "C" int mssim_ref_test_collide(int type_a, const float* pose_a, const float* param_a, const float* verts_a, int nverts_a,
                                      int type_b, const float* pose_b, const float* param_b, const float* verts_b, int nverts_b,
                                      float offset, int force_mpr, float* out) {
  Shape<Real> A, B;
  auto fill = [](Shape<Real>& s, int type, const float* pose, const float* param, const float* verts, int nv) {
    Pose<Real> P = pose7(pose);
    s.type = type; s.c = P.p; s.rot = qmat(P.q);
    for (int k = 0; k < 4; k++) s.param[k] = param[k];
    s.verts = verts; s.nverts = nv;
  };
  fill(A, type_a, pose_a, param_a, verts_a, nverts_a);
  fill(B, type_b, pose_b, param_b, verts_b, nverts_b);
  Manifold<Real> m;
  m.count = 0;
  if (force_mpr) collide_mpr(A, B, (Real)offset, m);
  else collide(A, B, (Real)offset, m);
  out[0] = (float)m.count; out[1] = (float)m.n.x; out[2] = (float)m.n.y; out[3] = (float)m.n.z;
  for (int k = 0; k < m.count; k++) {
    out[4 + 4 * k] = (float)m.x[k].x; out[5 + 4 * k] = (float)m.x[k].y; out[6 + 4 * k] = (float)m.x[k].z; out[7 + 4 * k] = (float)m.sep[k];
  }
  return m.count;
}
