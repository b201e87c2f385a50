// mssim_kernels.hip -- gfx950 kernels + C ABI (include/mssim.h) of the batched rigid-body step.
//
// Replaces `px.step()` and the px.gpu_apply_* / px.gpu_fetch_* / contact-query calls of
// ManiSkill's GPU path (mani_skill/envs/scene.py:374-375, 736-796, 941-977).
//
// Data layout in HBM: every per-env quantity is struct-of-arrays `[item][N]` with the env index
// fastest, so lane e of a wave touches address base + e*4: one 256-B coalesced request per
// wave-instruction. Constant model tables are env-shared and wave-uniform (scalar loads, L2
// resident).
//
// One kernel advances the simulation: k_solve16 (mssim_solve16.h), 16 lanes per env, a whole control step per
// launch (narrowphase, dynamics, solver, integration, FK for every substep; optionally the action map at its head and
// the copy-out + task epilogue at its tail). All launches go to the caller's stream, no host sync.
// `Topo` (FK-only kernel) selects compile-time (Panda: 9-DoF tree unrolled into VGPRs) or run-time topology.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/mssim.h"
#include "mssim_collide.h"

#define MAXC 52  // solver blocks per env: contact points + torsional blocks (overflow is reported, never silent); 4 envs x the LDS tables = 40.0 KB per block, 4 blocks per CU

struct DevModel {
  int n_dof, n_tendon, n_link, n_free, n_kin, n_shape, n_pair;
  const int *dof_parent, *dof_type, *body_gravity, *tendon_dof, *link_body, *free_gravity;
  const unsigned* dof_anc;  // bitmask of strict ancestors of each dof
  const float *dof_frame, *dof_axis, *dof_limit, *dof_drive, *dof_armature, *body_inertial, *tendon_param, *link_frame;
  const float *free_inertial, *free_damping;
  const int *shape_type, *shape_kind, *shape_index, *shape_row, *shape_hull, *pair_shape;
  const int* pair_packed;  // [n_pair rounded up to 128, + 128] shape a | shape b << 8 (-1 behind the last pair): what the control-step kernel's cull reads, one word per pair
  const float *shape_frame, *shape_param, *shape_material, *shape_bound, *hull_verts;
  const float* shape_center;  // [n_shape][3] bounding-sphere centre in the BODY frame (shape_frame applied)
  const float* dof_pack;      // [n_dof][32]  all per-joint constants of the cooperative kernel in one 128-byte record
  const float* shape_pack;    // [n_shape][24] all per-shape constants of its narrowphase in one 96-byte record
  const float* shape_half;    // [n_shape][3] half extents of a box in the SHAPE frame, centred at the bound centre, that contains the shape
  const float* tri_soup;      // [n_tri][12] triangle meshes: centroid, three corners relative to it (include/mssim.h)
  const float* tri_bvh;       // [n_tri_node][112] their 16-wide BVH
  const int* pair_mesh_slot;  // [n_pair] ordinal of the pair among those whose second shape is a triangle mesh, else -1 (rows of DevState::tri_clear)
  // per-env overrides ([items][N], env fastest); slot < 0 = shared value
  const int *shape_env_slot, *free_env_slot;
  const float *env_shape_frame, *env_shape_param, *env_shape_bound, *env_free_inertial;
  float gx, gy, gz, dt, contact_offset, rest_offset, erp, max_depen, sleep_threshold;
  int pos_iters, vel_iters;
};

struct DevState {
  int N;
  float *root, *q, *qd, *qt, *qdt, *qf, *qacc;  // [7][N], [n_dof][N] ...
  float *free_s, *free_force, *kin;             // [n_free*13][N], [n_free*3][N], [n_kin*7][N]
  float* free_wake;                             // [n_free][N] seconds of low energy left before the body sleeps; <= 0: asleep
  float* pcm;                                   // [N][MSSIM_PCM_SLOTS][48] persistent contact manifolds (mssim_solve16.h S16_PCM_LEN)
  int* pcm_tick;                                // [N] substep counter of the cache
  float* warm;                                  // [4 n_pair][N][4] contact multipliers (n, t1, t2) + substep stamp per (pair, manifold slot)
  float* tri_clear;                             // [4 n_mesh_pair][N] per (convex shape, mesh) pair: the shape's bounding-sphere centre in the mesh frame at its last
                                                // full BVH traversal and how far it may move from there before a triangle can be in range (<= 0: traverse)
  float *bodypose, *bodyvel;                    // [n_dof*7][N], [n_dof*6][N] (velocity about O = root position)
  float* bodyaux;                               // [n_dof*6][N] world joint axis (3) + joint anchor (3)
  int* pair_cnt;                                // [n_pair][N] contact points of the pair in the last substep (after the patch reduction)
  float* pair_imp;                              // [n_pair*3][N]
  int* hit_list;                                // [1 + MAXC][N]: count, then the pairs in contact after the last fused step
  float* rows;                                  // [N][S16_ROWS_GLB][32] J | W rows of the contacts beyond the register / LDS resident ones
  int* overflow;                                // [N]
  // action of this control step, mapped to drive targets at the head of the fused launch (mssim_step_action);
  // null = targets were set before the launch
  const float* act;                             // [N][act_dim]
  int act_dim;
  const int* act_col;                           // [n_dof] action column, < 0: joint untouched
  const float *act_lo, *act_hi;                 // [n_dof]
  const int* act_flags;                         // [n_dof] 1: delta on the current position, 2: clip + affine map
  const float* act_qpos;                        // user-visible qpos buffer (what the controller reads) or null
  float* act_target;                            // user-visible target_qpos buffer or null
  float* act_target_vel;                        // user-visible target_qvel buffer or null
  // copy-out + task epilogue at the tail of the fused launch (whole control step = one launch); 0 = none
  unsigned tail_fetch;                          // mssim_fetch mask
  mssim_buffers tail_buf;
  union { mssim_pick_task pick; mssim_push_task push; mssim_peg_task peg; } tail_task;  // kind = template TASK
  const int* tail_pairs; int tail_npairs;       // finger <-> object candidate pairs
  float *tail_obs, *tail_reward, *tail_head;
  uint8_t* tail_flags;
};

// ------------------------------------------------------------------------------------------------
// topology policies
struct TopoDyn {
  static constexpr int MAXD = MSSIM_MAX_DOF;
  static constexpr bool STATIC = false;
  static constexpr int UNROLL = 1;
  int nd;
  const int* par;
  const int* typ;
  const unsigned* ancm;
  MS_DEV TopoDyn(const DevModel& M) : nd(M.n_dof), par(M.dof_parent), typ(M.dof_type), ancm(M.dof_anc) {}
  MS_DEV int n() const { return nd; }
  MS_DEV int parent(int j) const { return par[j]; }
  MS_DEV bool revolute(int j) const { return typ[j] == MSSIM_JOINT_REVOLUTE; }
  MS_DEV unsigned anc(int j) const { return ancm[j]; }
};
struct TopoPanda {  // panda_v2/v3: 7 revolute chain + 2 prismatic fingers on body 6
  static constexpr int MAXD = 9;
  static constexpr bool STATIC = true;
  static constexpr int UNROLL = 9;
  MS_DEV TopoPanda(const DevModel&) {}
  MS_DEV constexpr int n() const { return 9; }
  MS_DEV constexpr int parent(int j) const { return j == 0 ? -1 : (j <= 7 ? j - 1 : 6); }
  MS_DEV constexpr bool revolute(int j) const { return j < 7; }
  MS_DEV constexpr unsigned anc(int j) const { return j <= 7 ? ((1u << j) - 1u) : 0x7Fu; }
};
static const int kPandaParent[9] = {-1, 0, 1, 2, 3, 4, 5, 6, 6};
static const int kPandaType[9] = {0, 0, 0, 0, 0, 0, 0, 1, 1};

// spatial helpers (world frame, about the origin O = articulation root position)
struct sv6 { f3 w, v; };                    // motion
struct sf6 { f3 n, f; };                    // force
struct si10 { float m; f3 h; s3 I; };       // inertia
MS_DEV sv6 crossm(sv6 a, sv6 b) { return sv6{cross(a.w, b.w), cross(a.w, b.v) + cross(a.v, b.w)}; }
MS_DEV sf6 crossf(sv6 a, sf6 b) { return sf6{cross(a.w, b.n) + cross(a.v, b.f), cross(a.w, b.f)}; }
MS_DEV sf6 imul(const si10& I, sv6 a) { return sf6{smulv(I.I, a.w) + cross(I.h, a.v), a.v * I.m - cross(I.h, a.w)}; }
MS_DEV float sdot(sv6 s, sf6 f) { return dot(s.w, f.n) + dot(s.v, f.f); }

#define SOA(ptr, item) ((ptr)[(size_t)(item) * N + e])

// XCD-aware block -> env-chunk mapping. Workgroups are dealt round-robin to the 8 XCDs (each with its
// own L2), so with the identity mapping every XCD touches a 1/8 slice of every cache line of the
// [item][N] arrays (false sharing across the 8 L2s, 8x fetch amplification). Grids are padded to a
// multiple of 8 blocks and block b works on chunk (b % 8) * (grid / 8) + b / 8: XCD x owns one
// contiguous range of envs in every kernel, so the state a kernel writes is re-read from the same L2.
MS_DEV int xcd_chunk(int b, int grid) { return (b & 7) * (grid >> 3) + (b >> 3); }

template <class T>
MS_DEV void fk_bodies(const T& topo, const DevModel& M, pose_t root, const float* q, pose_t* bp, f3* aw, f3* anchor) {
#pragma unroll T::UNROLL
  for (int j = 0; j < T::MAXD; j++) {
    if (j >= topo.n()) break;
    int p = topo.parent(j);
    pose_t P = root;
    if (T::STATIC) {
#pragma unroll
      for (int k = 0; k < T::MAXD; k++)
        if (k == p) P = bp[k];
    } else if (p >= 0) {
      P = bp[p];
    }
    pose_t J = pmul(P, pose_from(M.dof_frame + 7 * j));
    f3 al = f3{M.dof_axis[3 * j], M.dof_axis[3 * j + 1], M.dof_axis[3 * j + 2]};
    f3 a = qrot(J.q, al);
    pose_t B = J;
    if (topo.revolute(j)) B.q = qnormalized(qmul(J.q, qaxis_angle(al, q[j])));
    else B.p = J.p + a * q[j];
    bp[j] = B;
    aw[j] = a;
    anchor[j] = J.p;
  }
}

template <class T>
MS_DEV sv6 subspace(const T& topo, int j, f3 a, f3 r) {  // r = anchor - O
  if (topo.revolute(j)) return sv6{a, cross(r, a)};
  return sv6{f3{0.f, 0.f, 0.f}, a};
}

// body spatial velocities about O from qd; writes bodypose / bodyvel SoA
template <class T>
MS_DEV void write_kinematics(const T& topo, const DevModel& M, const DevState& S, int e, pose_t root, const float* qd,
                             const pose_t* bp, const f3* aw, const f3* anchor) {
  const int N = S.N;
  sv6 V[T::MAXD];
#pragma unroll T::UNROLL
  for (int j = 0; j < T::MAXD; j++) {
    if (j >= topo.n()) break;
    int p = topo.parent(j);
    sv6 Vp = sv6{f3{0, 0, 0}, f3{0, 0, 0}};
    if (T::STATIC) {
#pragma unroll
      for (int k = 0; k < T::MAXD; k++)
        if (k == p) Vp = V[k];
    } else if (p >= 0) {
      Vp = V[p];
    }
    sv6 Sj = subspace(topo, j, aw[j], anchor[j] - root.p);
    V[j] = sv6{Vp.w + Sj.w * qd[j], Vp.v + Sj.v * qd[j]};
    pose_store_soa(S.bodypose, 7 * j, N, e, bp[j]);
    float* o = S.bodyvel + (size_t)(6 * j) * N + e;
    o[0] = V[j].w.x; o[(size_t)N] = V[j].w.y; o[2 * (size_t)N] = V[j].w.z;
    o[3 * (size_t)N] = V[j].v.x; o[4 * (size_t)N] = V[j].v.y; o[5 * (size_t)N] = V[j].v.z;
    float* a = S.bodyaux + (size_t)(6 * j) * N + e;
    a[0] = aw[j].x; a[(size_t)N] = aw[j].y; a[2 * (size_t)N] = aw[j].z;
    a[3 * (size_t)N] = anchor[j].x; a[4 * (size_t)N] = anchor[j].y; a[5 * (size_t)N] = anchor[j].z;
  }
}

// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(64) void k_fk(DevModel M, DevState S) {
  const int N = S.N;
  int e = xcd_chunk(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (e >= N) return;
  T topo(M);
  pose_t root = pose_soa(S.root, 0, N, e);
  float q[T::MAXD], qd[T::MAXD];
#pragma unroll T::UNROLL
  for (int j = 0; j < T::MAXD; j++) {
    if (j >= topo.n()) break;
    q[j] = SOA(S.q, j);
    qd[j] = SOA(S.qd, j);
  }
  pose_t bp[T::MAXD];
  f3 aw[T::MAXD], anchor[T::MAXD];
  fk_bodies(topo, M, root, q, bp, aw, anchor);
  write_kinematics(topo, M, S, e, root, qd, bp, aw, anchor);
}

// ------------------------------------------------------------------------------------------------
MS_DEV pose_t body_pose_of(const DevModel& M, const DevState& S, int kind, int index, int e) {
  const int N = S.N;
  if (kind == MSSIM_BODY_ART) return index < 0 ? pose_soa(S.root, 0, N, e) : pose_soa(S.bodypose, 7 * index, N, e);
  if (kind == MSSIM_BODY_FREE) return pose_soa(S.free_s, 13 * index, N, e);
  if (kind == MSSIM_BODY_KIN) return pose_soa(S.kin, 7 * index, N, e);
  return pose_t{f3{0, 0, 0}, q4{1, 0, 0, 0}};
}

// the 10 inertial parameters of free body b in env e (per-env override or shared table)
MS_DEV void free_inertial_of(const DevModel& M, int N, int b, int e, float* out) {
  const int slot = M.free_env_slot[b];
#pragma unroll
  for (int k = 0; k < 10; k++) out[k] = slot < 0 ? M.free_inertial[10 * b + k] : M.env_free_inertial[(size_t)(10 * slot + k) * N + e];
}

// ------------------------------------------------------------------------------------------------
// apply / fetch: transposes between the user-visible AoS buffers and the SoA state
__global__ void k_apply(DevModel M, DevState S, mssim_buffers B, unsigned what) {
  const int N = S.N;
  int e = xcd_chunk(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const int n = M.n_dof;
  if ((what & MSSIM_RIGID_DATA) && B.rigid_body_data) {
    for (int b = 0; b < M.n_free; b++) {
      const float* r = B.rigid_body_data + 13 * ((size_t)(M.n_link + b) * N + e);
      // a row that differs from what the last fetch wrote wakes the body (include/mssim.h sleep_threshold)
      bool changed = false;
      for (int c = 0; c < 13; c++) changed = changed || r[c] != SOA(S.free_s, 13 * b + c);
      if (!changed) continue;
      for (int c = 0; c < 13; c++) SOA(S.free_s, 13 * b + c) = r[c];
      SOA(S.free_wake, b) = MSSIM_WAKE_TIME;
    }
    // a kinematic body given a different pose wakes the env's free bodies (include/mssim.h sleep_threshold): it may have been
    // moved into a sleeping body, or away from under one
    bool kin_moved = false;
    for (int k = 0; k < M.n_kin; k++) {
      const float* r = B.rigid_body_data + 13 * ((size_t)(M.n_link + M.n_free + k) * N + e);
      for (int c = 0; c < 7; c++) {
        kin_moved = kin_moved || r[c] != SOA(S.kin, 7 * k + c);
        SOA(S.kin, 7 * k + c) = r[c];
      }
    }
    if (kin_moved)
      for (int b = 0; b < M.n_free; b++) SOA(S.free_wake, b) = MSSIM_WAKE_TIME;
  }
  if ((what & MSSIM_ART_ROOT_POSE) && B.rigid_body_data && M.n_link > 0) {
    const float* r = B.rigid_body_data + 13 * (size_t)e;
    for (int c = 0; c < 7; c++) SOA(S.root, c) = r[c];
  }
  if ((what & MSSIM_ART_QPOS) && B.art_qpos) for (int j = 0; j < n; j++) SOA(S.q, j) = B.art_qpos[(size_t)e * n + j];
  if ((what & MSSIM_ART_QVEL) && B.art_qvel) for (int j = 0; j < n; j++) SOA(S.qd, j) = B.art_qvel[(size_t)e * n + j];
  if ((what & MSSIM_ART_QF) && B.art_qf) for (int j = 0; j < n; j++) SOA(S.qf, j) = B.art_qf[(size_t)e * n + j];
  if ((what & MSSIM_ART_TARGET_POS) && B.art_target_qpos) for (int j = 0; j < n; j++) SOA(S.qt, j) = B.art_target_qpos[(size_t)e * n + j];
  if ((what & MSSIM_ART_TARGET_VEL) && B.art_target_qvel) for (int j = 0; j < n; j++) SOA(S.qdt, j) = B.art_target_qvel[(size_t)e * n + j];
  if ((what & MSSIM_RIGID_FORCE) && B.rigid_body_force)
    for (int b = 0; b < M.n_free; b++) {
      const float* f = B.rigid_body_force + 4 * ((size_t)(M.n_link + b) * N + e);
      for (int c = 0; c < 3; c++) SOA(S.free_force, 3 * b + c) = f[c];
      if (f[0] != 0.f || f[1] != 0.f || f[2] != 0.f) SOA(S.free_wake, b) = MSSIM_WAKE_TIME;
    }
}

// fetch: grid (N/64, n_rows + 1). blockIdx.y < n_rows: one (body row, env) per lane -> consecutive
// lanes write consecutive 52-byte records (coalesced); blockIdx.y == n_rows: the articulation arrays
// (device functions: k_fetch runs them one (row, env) per lane, the fetch + task-epilogue launches run the
// rows of an env over 4 lanes of one block)
MS_DEV void fetch_row(const DevModel& M, const DevState& S, const mssim_buffers& B, unsigned what, int e, int row) {
  const int N = S.N;
  if (!B.rigid_body_data) return;
  float* r = B.rigid_body_data + 13 * ((size_t)row * N + e);
  if (row < M.n_link) {
    if (!(what & (MSSIM_LINK_POSE | MSSIM_LINK_VEL))) return;
    const pose_t root = pose_soa(S.root, 0, N, e);
    const int b = M.link_body[row];
    const pose_t P = pmul(b < 0 ? root : pose_soa(S.bodypose, 7 * b, N, e), pose_from(M.link_frame + 7 * row));
    if (what & MSSIM_LINK_POSE) { r[0] = P.p.x; r[1] = P.p.y; r[2] = P.p.z; r[3] = P.q.w; r[4] = P.q.x; r[5] = P.q.y; r[6] = P.q.z; }
    if (what & MSSIM_LINK_VEL) {
      f3 w = f3{0, 0, 0}, vv = f3{0, 0, 0};
      if (b >= 0) {
        w = f3{SOA(S.bodyvel, 6 * b), SOA(S.bodyvel, 6 * b + 1), SOA(S.bodyvel, 6 * b + 2)};
        vv = f3{SOA(S.bodyvel, 6 * b + 3), SOA(S.bodyvel, 6 * b + 4), SOA(S.bodyvel, 6 * b + 5)} + cross(w, P.p - root.p);
      }
      r[7] = vv.x; r[8] = vv.y; r[9] = vv.z; r[10] = w.x; r[11] = w.y; r[12] = w.z;
    }
  } else if (what & MSSIM_RIGID_DATA) {
    if (row < M.n_link + M.n_free) {
      const int b = row - M.n_link;
      for (int c = 0; c < 13; c++) r[c] = SOA(S.free_s, 13 * b + c);
    } else {
      const int k = row - M.n_link - M.n_free;
      const pose_t P = pose_soa(S.kin, 7 * k, N, e);
      r[0] = P.p.x; r[1] = P.p.y; r[2] = P.p.z; r[3] = P.q.w; r[4] = P.q.x; r[5] = P.q.y; r[6] = P.q.z;
      for (int c = 7; c < 13; c++) r[c] = 0.f;
    }
  }
}
// one joint per lane (the control-step kernel's tail)
MS_DEV void fetch_art_joint(const DevModel& M, const DevState& S, const mssim_buffers& B, unsigned what, int e, int j) {
  const int N = S.N;
  const int n = M.n_dof;
  if ((what & MSSIM_ART_QPOS) && B.art_qpos) B.art_qpos[(size_t)e * n + j] = SOA(S.q, j);
  if ((what & MSSIM_ART_QVEL) && B.art_qvel) B.art_qvel[(size_t)e * n + j] = SOA(S.qd, j);
  if ((what & MSSIM_ART_QACC) && B.art_qacc) B.art_qacc[(size_t)e * n + j] = SOA(S.qacc, j);
  if ((what & MSSIM_ART_TARGET_POS) && B.art_target_qpos) B.art_target_qpos[(size_t)e * n + j] = SOA(S.qt, j);
  if ((what & MSSIM_ART_TARGET_VEL) && B.art_target_qvel) B.art_target_qvel[(size_t)e * n + j] = SOA(S.qdt, j);
}
MS_DEV void fetch_art(const DevModel& M, const DevState& S, const mssim_buffers& B, unsigned what, int e) {
  const int N = S.N;
  const int n = M.n_dof;
  if ((what & MSSIM_ART_QPOS) && B.art_qpos) for (int j = 0; j < n; j++) B.art_qpos[(size_t)e * n + j] = SOA(S.q, j);
  if ((what & MSSIM_ART_QVEL) && B.art_qvel) for (int j = 0; j < n; j++) B.art_qvel[(size_t)e * n + j] = SOA(S.qd, j);
  if ((what & MSSIM_ART_QACC) && B.art_qacc) for (int j = 0; j < n; j++) B.art_qacc[(size_t)e * n + j] = SOA(S.qacc, j);
  if ((what & MSSIM_ART_TARGET_POS) && B.art_target_qpos) for (int j = 0; j < n; j++) B.art_target_qpos[(size_t)e * n + j] = SOA(S.qt, j);
  if ((what & MSSIM_ART_TARGET_VEL) && B.art_target_qvel) for (int j = 0; j < n; j++) B.art_target_qvel[(size_t)e * n + j] = SOA(S.qdt, j);
}
__global__ __launch_bounds__(64) void k_fetch(DevModel M, DevState S, mssim_buffers B, unsigned what) {
  const int e = xcd_chunk(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (e >= S.N) return;
  const int R = M.n_link + M.n_free + M.n_kin;
  if ((int)blockIdx.y < R) fetch_row(M, S, B, what, e, blockIdx.y);
  else fetch_art(M, S, B, what, e);
}
// fetch inside a task-epilogue launch (256 threads = 64 envs x 4 lanes): the 4 lanes of an env share its
// rows, then one of them runs the epilogue on what the block just wrote (same CU, same L1)
MS_DEV int fetch_in_block(const DevModel& M, const DevState& S, const mssim_buffers& B, unsigned what) {
  const int e = xcd_chunk(blockIdx.x, gridDim.x) * 64 + (threadIdx.x & 63);
  const int ty = threadIdx.x >> 6;
  if (e < S.N) {
    const int R = M.n_link + M.n_free + M.n_kin;
    for (int row = ty; row < R; row += 4) fetch_row(M, S, B, what, e, row);
    if (ty == 3) fetch_art(M, S, B, what, e);
  }
  __threadfence_block();
  __syncthreads();
  return (ty == 0 && e < S.N) ? e : -1;
}

// contact impulse queries: q = [nq][2] body rows (pair query) or [nq] rows (body query)
__global__ void k_query(DevModel M, DevState S, const int* q, int nq, int body_query, float* out) {
  const int N = S.N;
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  if (e >= N || k >= nq) return;
  f3 s = f3{0, 0, 0};
  int qa = body_query ? q[k] : q[2 * k], qb = body_query ? -2 : q[2 * k + 1];
  for (int p = 0; p < M.n_pair; p++) {
    int ra = M.shape_row[M.pair_shape[2 * p]], rb = M.shape_row[M.pair_shape[2 * p + 1]];
    float sign = 0.f;
    if (body_query) sign = (ra == qa ? 1.f : 0.f) - (rb == qa ? 1.f : 0.f);
    else if (ra == qa && rb == qb) sign = 1.f;
    else if (ra == qb && rb == qa) sign = -1.f;
    if (sign != 0.f && S.pair_cnt[(size_t)p * N + e] > 0)
      s += f3{SOA(S.pair_imp, 3 * p), SOA(S.pair_imp, 3 * p + 1), SOA(S.pair_imp, 3 * p + 2)} * sign;
  }
  float* o = out + 3 * ((size_t)k * N + e);
  o[0] = s.x; o[1] = s.y; o[2] = s.z;
}

// End-effector block of the action map (pd_ee_delta_pos): 3 action columns = translation of link `link` in the
// root frame; the joints flagged 4 in the joint map get  target = qpos + J^T (J J^T + 1e-9 I)^-1 a  with J the
// translational Jacobian of the link over all joints on its path (agents/controllers/utils/kinematics.py:156-171)
struct EeMap {
  int link;    // < 0: no end-effector block
  int col0;    // first action column
  int rows;    // 3: translation (pd_ee_delta_pos), 6: translation + rotation vector (pd_ee_delta_pose)
  float lo, hi;
  float rot_scale;  // rows == 6: the rotation columns are clipped by their norm to 1 and scaled by this (pd_ee_pose.py:197-210)
  int flags;   // 2: clip to [-1, 1] and map to [lo, hi] (translation) / clip by norm and scale (rotation)
};
// affine action -> drive targets (user-visible buffer + simulation state)
__global__ void k_apply_action(DevModel M, DevState S, mssim_buffers B, const float* __restrict__ action, int adim,
                               const int* __restrict__ col, const float* __restrict__ lo, const float* __restrict__ hi,
                               const int* __restrict__ flags, EeMap ee) {
  const int N = S.N;
  int e = xcd_chunk(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const int n = M.n_dof;
  for (int j = 0; j < n; j++) {
    const int cj = col[j];
    if (cj < 0) continue;
    float a = action[(size_t)e * adim + cj];
    if (flags[j] & 2) {
      a = fminf(fmaxf(a, -1.f), 1.f);
      a = 0.5f * (hi[j] + lo[j]) + 0.5f * (hi[j] - lo[j]) * a;
    }
    if (flags[j] & 48) {  // forward velocity of a planar base in its own frame (include/mssim.h set_action_map)
      const int jy = (flags[j] >> 8) & 31;
      const float yaw = B.art_qpos ? B.art_qpos[(size_t)e * n + jy] : SOA(S.q, jy);
      a *= (flags[j] & 16) ? cosf(yaw) : sinf(yaw);
    }
    if (flags[j] & 8) {  // velocity drive target (pd_joint_vel, agents/controllers/pd_joint_vel.py:31-33)
      SOA(S.qdt, j) = a;
      if (B.art_target_qvel) B.art_target_qvel[(size_t)e * n + j] = a;
      continue;
    }
    const float qj = B.art_qpos ? B.art_qpos[(size_t)e * n + j] : SOA(S.q, j);  // what `controller.qpos` reads
    const float t = ((flags[j] & 1) ? qj : 0.f) + a;
    SOA(S.qt, j) = t;
    if (B.art_target_qpos) B.art_target_qpos[(size_t)e * n + j] = t;
  }
  if (ee.link >= 0) {
    const pose_t root = pose_soa(S.root, 0, N, e);
    const m3 Rr = qmat(root.q);
    const int b = M.link_body[ee.link];
    const pose_t Pb = b < 0 ? root : pose_soa(S.bodypose, 7 * b, N, e);
    const f3 pe = pmul(Pb, pose_from(M.link_frame + 7 * ee.link)).p;
    const unsigned path = b < 0 ? 0u : (M.dof_anc[b] | (1u << b));
    float av[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < 3; r++) {
      float a = action[(size_t)e * adim + ee.col0 + r];
      if (ee.flags & 2) {
        a = fminf(fmaxf(a, -1.f), 1.f);
        a = 0.5f * (ee.hi + ee.lo) + 0.5f * (ee.hi - ee.lo) * a;
      }
      av[r] = a;
    }
    if (ee.rows == 6) {
      f3 rot = f3{action[(size_t)e * adim + ee.col0 + 3], action[(size_t)e * adim + ee.col0 + 4], action[(size_t)e * adim + ee.col0 + 5]};
      if (ee.flags & 2) {
        const float nr = sqrtf(dot(rot, rot));
        if (nr > 1.f) rot = rot * (1.f / fmaxf(nr, 1e-12f));
        rot = rot * ee.rot_scale;
      }
      av[3] = rot.x; av[4] = rot.y; av[5] = rot.z;
    }
    // column j of the link's Jacobian in the root frame: linear part, angular part
    auto jcol = [&](int j, f3& jw) {
      const f3 a = f3{SOA(S.bodyaux, 6 * j), SOA(S.bodyaux, 6 * j + 1), SOA(S.bodyaux, 6 * j + 2)};
      const f3 an = f3{SOA(S.bodyaux, 6 * j + 3), SOA(S.bodyaux, 6 * j + 4), SOA(S.bodyaux, 6 * j + 5)};
      const bool rev = M.dof_type[j] == MSSIM_JOINT_REVOLUTE;
      jw = rev ? mtmulv(Rr, a) : f3{0.f, 0.f, 0.f};
      return mtmulv(Rr, rev ? cross(a, pe - an) : a);
    };
    float y[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ee.rows == 3) {
      s3 G = s3{1e-9f, 1e-9f, 1e-9f, 0.f, 0.f, 0.f};  // xx yy zz xy xz yz
      for (int j = 0; j < n; j++)
        if ((path >> j) & 1u) {
          f3 w;
          const f3 v = jcol(j, w);
          G.xx += v.x * v.x; G.yy += v.y * v.y; G.zz += v.z * v.z; G.xy += v.x * v.y; G.xz += v.x * v.z; G.yz += v.y * v.z;
        }
      // (G >= 1e-9 I in exact arithmetic; a determinant lost to cancellation moves nothing rather than everything)
      const float detG = G.xx * (G.yy * G.zz - G.yz * G.yz) - G.xy * (G.xy * G.zz - G.yz * G.xz) + G.xz * (G.xy * G.yz - G.yy * G.xz);
      const f3 y3 = detG > 1e-30f ? smulv(sinverse(G), f3{av[0], av[1], av[2]}) : f3{0.f, 0.f, 0.f};
      y[0] = y3.x; y[1] = y3.y; y[2] = y3.z;
    } else {
      // 6 x 6: J J^T + 1e-9 I is symmetric positive definite -> Cholesky without pivoting, two triangular solves
      float G[6][6];
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int q = 0; q < 6; q++) G[r][q] = r == q ? 1e-9f : 0.f;
      for (int j = 0; j < n; j++)
        if ((path >> j) & 1u) {
          f3 w;
          const f3 v = jcol(j, w);
          const float cj[6] = {v.x, v.y, v.z, w.x, w.y, w.z};
#pragma unroll
          for (int r = 0; r < 6; r++)
#pragma unroll
            for (int q = 0; q <= r; q++) G[r][q] += cj[r] * cj[q];
        }
      float Lc[6][6];
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int q = 0; q <= r; q++) {
          float sum = G[r][q];
#pragma unroll
          for (int m = 0; m < q; m++) sum -= Lc[r][m] * Lc[q][m];
          Lc[r][q] = r == q ? sqrtf(fmaxf(sum, 1e-20f)) : sum / Lc[q][q];
        }
      float z[6];
#pragma unroll
      for (int r = 0; r < 6; r++) {
        float sum = av[r];
#pragma unroll
        for (int m = 0; m < r; m++) sum -= Lc[r][m] * z[m];
        z[r] = sum / Lc[r][r];
      }
#pragma unroll
      for (int r = 5; r >= 0; r--) {
        float sum = z[r];
#pragma unroll
        for (int m = r + 1; m < 6; m++) sum -= Lc[m][r] * y[m];
        y[r] = sum / Lc[r][r];
      }
    }
    for (int j = 0; j < n; j++)
      if (((path >> j) & 1u) && (flags[j] & 4)) {
        f3 w;
        const f3 v = jcol(j, w);
        const float qj = B.art_qpos ? B.art_qpos[(size_t)e * n + j] : SOA(S.q, j);
        const float t = qj + v.x * y[0] + v.y * y[1] + v.z * y[2] + w.x * y[3] + w.y * y[4] + w.z * y[5];
        SOA(S.qt, j) = t;
        if (B.art_target_qpos) B.art_target_qpos[(size_t)e * n + j] = t;
      }
  }
}

// PickCube-style evaluate / obs / reward
// FETCH: the launch first performs mssim_fetch(what) for its envs (fetch_in_block; 256 threads per block)
MS_DEV void task_pick_env(const DevModel& M, const DevState& S, const mssim_buffers& B, const mssim_pick_task& T, const int* __restrict__ pairs, int npairs,
                                                    float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ flags, int e) {
  const int N = S.N;
  const int n = M.n_dof;
  const int D = 2 * n + 24;
  float* o = obs + (size_t)e * D;
  auto rowp = [&](int row) { return B.rigid_body_data + 13 * ((size_t)row * N + e); };
  float qv_max = 0.f, qv_sq = 0.f;
  for (int j = 0; j < n; j++) {
    const float q = B.art_qpos[(size_t)e * n + j], v = B.art_qvel[(size_t)e * n + j];
    o[j] = q;
    o[n + j] = v;
    if (j < T.n_static_dofs) { qv_max = fmaxf(qv_max, fabsf(v)); qv_sq += v * v; }
  }
  const float* tcp = rowp(T.tcp_row);
  const float* ob = rowp(T.obj_row);
  const float* gl = rowp(T.goal_row);
  const f3 ptcp = f3{tcp[0], tcp[1], tcp[2]}, pobj = f3{ob[0], ob[1], ob[2]}, pgoal = f3{gl[0], gl[1], gl[2]};
  // pairwise contact impulses finger <-> object during the last substep (scene.py:736-796); the few
  // candidate pairs are listed by the host: entry = pair | finger (bit 30: 0 left, 1 right) | object is shape A (bit 31)
  f3 lf = f3{0, 0, 0}, rf = f3{0, 0, 0};
  for (int k = 0; k < npairs; k++) {
    const unsigned ent = (unsigned)pairs[k];
    const int p = (int)(ent & 0x3FFFFFFFu);
    if (S.pair_cnt[(size_t)p * N + e] <= 0) continue;
    // impulse on the finger from the object: +imp if the finger is shape A, -imp otherwise
    f3 imp = f3{SOA(S.pair_imp, 3 * p), SOA(S.pair_imp, 3 * p + 1), SOA(S.pair_imp, 3 * p + 2)} * ((ent >> 31) ? -1.f : 1.f);
    if ((ent >> 30) & 1u) rf += imp; else lf += imp;
  }
  const float inv_dt = 1.f / M.dt;
  lf = lf * inv_dt; rf = rf * inv_dt;
  auto yaxis = [&](const float* r) { return mcol(qmat(qnormalized(q4{r[3], r[4], r[5], r[6]})), 1); };
  auto angle_deg = [&](f3 a, f3 b) {
    const float na = norm(a), nb = norm(b);
    a = a * (1.f / (na < 1e-6f ? 1.f : na));
    b = b * (1.f / (nb < 1e-6f ? 1.f : nb));
    return acosf(fminf(fmaxf(dot(a, b), -1.f), 1.f)) * 57.29577951308232f;
  };
  const f3 ldir = yaxis(rowp(T.finger1_row)), rdir = -yaxis(rowp(T.finger2_row));
  const bool lflag = norm(lf) >= T.min_force && angle_deg(ldir, lf) <= T.max_angle_deg;
  const bool rflag = norm(rf) >= T.min_force && angle_deg(rdir, rf) <= T.max_angle_deg;
  const bool grasped = lflag && rflag;
  const float d_goal = norm(pgoal - pobj);
  const bool placed = d_goal <= T.goal_thresh;
  const bool is_static = qv_max <= T.static_thresh;
  const bool success = placed && is_static;
  // observation
  int k = 2 * n;
  o[k++] = grasped ? 1.f : 0.f;
  for (int i = 0; i < 7; i++) o[k++] = tcp[i];
  o[k++] = pgoal.x; o[k++] = pgoal.y; o[k++] = pgoal.z;
  for (int i = 0; i < 7; i++) o[k++] = ob[i];
  o[k++] = pobj.x - ptcp.x; o[k++] = pobj.y - ptcp.y; o[k++] = pobj.z - ptcp.z;
  o[k++] = pgoal.x - pobj.x; o[k++] = pgoal.y - pobj.y; o[k++] = pgoal.z - pobj.z;
  // dense reward (pick_cube.py:128-158)
  float r = 1.f - tanhf(5.f * norm(pobj - ptcp));
  if (grasped) r += 1.f + (1.f - tanhf(5.f * d_goal));
  if (placed) r += 1.f - tanhf(5.f * sqrtf(qv_sq));
  if (success) r = 5.f;
  reward[e] = r * T.reward_scale;
  uint8_t* f = flags + 4 * (size_t)e;
  f[0] = success; f[1] = placed; f[2] = is_static; f[3] = grasped;
  if (T.terminated_out) T.terminated_out[e] = success;
  if (T.elapsed_steps) { const int v = T.elapsed_steps[e] + 1; T.elapsed_steps[e] = v; if (T.elapsed_out) T.elapsed_out[e] = v; if (T.truncated_out) T.truncated_out[e] = v >= T.time_limit ? 1 : 0; }
}
// FETCH: the launch first performs mssim_fetch(what) for its envs (fetch_in_block; 256 threads per block)
template <bool FETCH>
__global__ __launch_bounds__(256) void k_task_pick(DevModel M, DevState S, mssim_buffers B, unsigned what, mssim_pick_task T, const int* __restrict__ pairs, int npairs,
                                                    float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ flags) {
  int e;
  if (FETCH) { e = fetch_in_block(M, S, B, what); if (e < 0) return; }
  else { e = xcd_chunk(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; if (e >= S.N) return; }
  task_pick_env(M, S, B, T, pairs, npairs, obs, reward, flags, e);
}

MS_DEV void task_push_env(const DevModel& M, const DevState& S, const mssim_buffers& B, const mssim_push_task& T, float* __restrict__ obs, float* __restrict__ reward,
                                                    uint8_t* __restrict__ flags, int e) {
  const int N = S.N;
  const int n = M.n_dof;
  float* o = obs + (size_t)e * (2 * n + 17);
  auto rowp = [&](int row) { return B.rigid_body_data + 13 * ((size_t)row * N + e); };
  for (int j = 0; j < n; j++) {
    o[j] = B.art_qpos[(size_t)e * n + j];
    o[n + j] = B.art_qvel[(size_t)e * n + j];
  }
  const float* tcp = rowp(T.tcp_row);
  const float* ob = rowp(T.obj_row);
  const float* gl = rowp(T.goal_row);
  int k = 2 * n;
  for (int i = 0; i < 7; i++) o[k++] = tcp[i];
  for (int i = 0; i < 3; i++) o[k++] = gl[i];
  for (int i = 0; i < 7; i++) o[k++] = ob[i];
  // evaluate (push_cube.py:165-176) and dense reward (:209-232)
  const float dx = ob[0] - gl[0], dy = ob[1] - gl[1];
  const float obj_to_goal = sqrtf(dx * dx + dy * dy);
  const bool success = obj_to_goal < T.goal_radius && ob[2] < T.cube_half_size + 5e-3f;
  const f3 push_p = f3{ob[0] - T.cube_half_size - 0.005f, ob[1], ob[2]};
  const float dist = norm(push_p - f3{tcp[0], tcp[1], tcp[2]});
  float r = 1.f - tanhf(5.f * dist);
  if (dist < 0.01f) r += 1.f - tanhf(5.f * obj_to_goal);
  if (success) r = 3.f;
  reward[e] = r * T.reward_scale;
  flags[e] = success;
  if (T.terminated_out) T.terminated_out[e] = success;
  if (T.elapsed_steps) { const int v = T.elapsed_steps[e] + 1; T.elapsed_steps[e] = v; if (T.elapsed_out) T.elapsed_out[e] = v; if (T.truncated_out) T.truncated_out[e] = v >= T.time_limit ? 1 : 0; }
}
// FETCH: the launch first performs mssim_fetch(what) for its envs (fetch_in_block; 256 threads per block)
template <bool FETCH>
__global__ __launch_bounds__(256) void k_task_push(DevModel M, DevState S, mssim_buffers B, unsigned what, mssim_push_task T, float* __restrict__ obs, float* __restrict__ reward,
                                                    uint8_t* __restrict__ flags) {
  int e;
  if (FETCH) { e = fetch_in_block(M, S, B, what); if (e < 0) return; }
  else { e = xcd_chunk(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; if (e >= S.N) return; }
  task_push_env(M, S, B, T, obs, reward, flags, e);
}

// Pose algebra exactly as the Python `Pose` class does it (utils/structs/pose.py, rotation_conversions.py:
// no re-normalisation, rotation as p + w t + v x t with t = 2 v x p, product standardised to w >= 0), so
// that the fused task outputs equal the torch path also for the slightly non-unit quaternions users set
MS_DEV f3 tq_apply(q4 q, f3 p) {
  const f3 v = f3{q.x, q.y, q.z};
  const f3 t = cross(v, p) * 2.f;
  return p + t * q.w + cross(v, t);
}
MS_DEV pose_t tq_mul(pose_t a, pose_t b) {
  q4 q = qmul(a.q, b.q);
  if (q.w < 0.f) q = q4{-q.w, -q.x, -q.y, -q.z};
  return pose_t{a.p + tq_apply(a.q, b.p), q};
}
MS_DEV pose_t tq_inv(pose_t a) {
  const q4 qc = q4{a.q.w, -a.q.x, -a.q.y, -a.q.z};
  return pose_t{tq_apply(qc, -a.p), qc};
}
// PegInsertionSide evaluate / obs / reward (peg_insertion_side.py:247-355)
MS_DEV void task_peg_env(const DevModel& M, const DevState& S, const mssim_buffers& B, const mssim_peg_task& T, const int* __restrict__ pairs, int npairs,
                                                   float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ flags, float* __restrict__ head_out, int e) {
  const int N = S.N;
  const int n = M.n_dof;
  float* o = obs + (size_t)e * (2 * n + 25);
  auto rowp = [&](int row) { return B.rigid_body_data + 13 * ((size_t)row * N + e); };
  auto rowpose = [&](int row) { const float* r = rowp(row); return pose_t{f3{r[0], r[1], r[2]}, q4{r[3], r[4], r[5], r[6]}}; };
  for (int j = 0; j < n; j++) {
    o[j] = B.art_qpos[(size_t)e * n + j];
    o[n + j] = B.art_qvel[(size_t)e * n + j];
  }
  const float* tcp = rowp(T.tcp_row);
  const float* pg = rowp(T.peg_row);
  const pose_t Ptcp = rowpose(T.tcp_row), Ppeg = rowpose(T.peg_row), Pbox = rowpose(T.box_row);
  const f3 hs = f3{T.peg_half_sizes[3 * e], T.peg_half_sizes[3 * e + 1], T.peg_half_sizes[3 * e + 2]};
  const f3 hoff = f3{T.box_hole_offsets[3 * e], T.box_hole_offsets[3 * e + 1], T.box_hole_offsets[3 * e + 2]};
  const float rad = T.box_hole_radii[e];
  const q4 qi = q4{1, 0, 0, 0};
  const pose_t head = tq_mul(Ppeg, pose_t{f3{hs.x, 0, 0}, qi});      // peg head (orange end)
  const pose_t hole = tq_mul(Pbox, pose_t{hoff, qi});                 // hole centre frame
  const f3 hah = tq_mul(tq_inv(hole), head).p;                          // head in the hole frame
  const bool success = hah.x >= -0.015f && fabsf(hah.y) <= rad && fabsf(hah.z) <= rad;
  // finger <-> peg contact forces of the last substep (Panda.is_grasping, panda.py:236-264)
  f3 lf = f3{0, 0, 0}, rf = f3{0, 0, 0};
  for (int k = 0; k < npairs; k++) {
    const unsigned ent = (unsigned)pairs[k];
    const int p = (int)(ent & 0x3FFFFFFFu);
    if (S.pair_cnt[(size_t)p * N + e] <= 0) continue;
    f3 imp = f3{SOA(S.pair_imp, 3 * p), SOA(S.pair_imp, 3 * p + 1), SOA(S.pair_imp, 3 * p + 2)} * ((ent >> 31) ? -1.f : 1.f);
    if ((ent >> 30) & 1u) rf += imp; else lf += imp;
  }
  const float inv_dt = 1.f / M.dt;
  lf = lf * inv_dt; rf = rf * inv_dt;
  auto yaxis = [&](int row) { return mcol(qmat(qnormalized(rowpose(row).q)), 1); };
  auto angle_deg = [&](f3 a, f3 b) {
    const float na = norm(a), nb = norm(b);
    a = a * (1.f / (na < 1e-6f ? 1.f : na));
    b = b * (1.f / (nb < 1e-6f ? 1.f : nb));
    return acosf(fminf(fmaxf(dot(a, b), -1.f), 1.f)) * 57.29577951308232f;
  };
  const f3 ldir = yaxis(T.finger1_row), rdir = -yaxis(T.finger2_row);
  const bool grasped = norm(lf) >= T.min_force && angle_deg(ldir, lf) <= T.max_angle_deg && norm(rf) >= T.min_force && angle_deg(rdir, rf) <= T.max_angle_deg;
  // observation
  int k = 2 * n;
  for (int i = 0; i < 7; i++) o[k++] = tcp[i];
  for (int i = 0; i < 7; i++) o[k++] = pg[i];
  o[k++] = hs.x; o[k++] = hs.y; o[k++] = hs.z;
  o[k++] = hole.p.x; o[k++] = hole.p.y; o[k++] = hole.p.z; o[k++] = hole.q.w; o[k++] = hole.q.x; o[k++] = hole.q.y; o[k++] = hole.q.z;
  o[k++] = rad;
  // dense reward
  const f3 grasp_target = tq_mul(Ppeg, pose_t{f3{-0.06f, 0, 0}, qi}).p;
  float r = 1.f - tanhf(4.f * norm(Ptcp.p - grasp_target));
  if (grasped) r += 1.f;
  const pose_t goal_inv = tq_inv(tq_mul(hole, pose_t{f3{-hs.x, 0, 0}, qi}));   // goal = box * hole_offset * head_offset^-1
  const f3 hg = tq_mul(goal_inv, head).p, bg = tq_mul(goal_inv, Ppeg).p;
  const float head_yz = sqrtf(hg.y * hg.y + hg.z * hg.z), body_yz = sqrtf(bg.y * bg.y + bg.z * bg.z);
  if (grasped) r += 3.f * (1.f - tanhf(0.5f * (head_yz + body_yz) + 4.5f * fmaxf(head_yz, body_yz)));
  if (grasped && head_yz < 0.01f && body_yz < 0.01f) r += 5.f * (1.f - tanhf(5.f * norm(hah)));
  if (success) r = 10.f;
  reward[e] = r * T.reward_scale;
  flags[e] = success;
  head_out[3 * (size_t)e] = hah.x; head_out[3 * (size_t)e + 1] = hah.y; head_out[3 * (size_t)e + 2] = hah.z;
  if (T.terminated_out) T.terminated_out[e] = success;
  if (T.elapsed_steps) { const int v = T.elapsed_steps[e] + 1; T.elapsed_steps[e] = v; if (T.elapsed_out) T.elapsed_out[e] = v; if (T.truncated_out) T.truncated_out[e] = v >= T.time_limit ? 1 : 0; }
}
// FETCH: the launch first performs mssim_fetch(what) for its envs (fetch_in_block; 256 threads per block)
template <bool FETCH>
__global__ __launch_bounds__(256) void k_task_peg(DevModel M, DevState S, mssim_buffers B, unsigned what, mssim_peg_task T, const int* __restrict__ pairs, int npairs,
                                                   float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ flags, float* __restrict__ head_out) {
  int e;
  if (FETCH) { e = fetch_in_block(M, S, B, what); if (e < 0) return; }
  else { e = xcd_chunk(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; if (e >= S.N) return; }
  task_peg_env(M, S, B, T, pairs, npairs, obs, reward, flags, head_out, e);
}

// geometric Jacobian of link `link` in the root frame: out [N][6][n_dof] (see include/mssim.h)
__global__ void k_link_jacobian(DevModel M, DevState S, int link, float* __restrict__ out) {
  const int N = S.N;
  int e = xcd_chunk(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const int n = M.n_dof;
  const pose_t root = pose_soa(S.root, 0, N, e);
  const m3 Rr = qmat(root.q);
  const int b = M.link_body[link];
  const pose_t Pb = b < 0 ? root : pose_soa(S.bodypose, 7 * b, N, e);
  const f3 pe = pmul(Pb, pose_from(M.link_frame + 7 * link)).p;
  const unsigned path = b < 0 ? 0u : (M.dof_anc[b] | (1u << b));
  float* o = out + (size_t)e * 6 * n;
  for (int j = 0; j < n; j++) {
    f3 jv = f3{0, 0, 0}, jw = f3{0, 0, 0};
    if ((path >> j) & 1u) {
      const f3 a = f3{SOA(S.bodyaux, 6 * j), SOA(S.bodyaux, 6 * j + 1), SOA(S.bodyaux, 6 * j + 2)};
      const f3 an = f3{SOA(S.bodyaux, 6 * j + 3), SOA(S.bodyaux, 6 * j + 4), SOA(S.bodyaux, 6 * j + 5)};
      if (M.dof_type[j] == MSSIM_JOINT_REVOLUTE) { jv = cross(a, pe - an); jw = a; }
      else jv = a;
      jv = mtmulv(Rr, jv);
      jw = mtmulv(Rr, jw);
    }
    o[0 * n + j] = jv.x; o[1 * n + j] = jv.y; o[2 * n + j] = jv.z;
    o[3 * n + j] = jw.x; o[4 * n + j] = jw.y; o[5 * n + j] = jw.z;
  }
}

__global__ void k_fill(float* dst, float v, size_t count) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < count) dst[i] = v;
}
__global__ void k_i2f(const int* src, float* dst, size_t count) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < count) dst[i] = (float)src[i];
}

// =================================================================================================
// host side
#include "mssim_solve16.h"

struct mssim_sim {
  int device = 0;
  int N = 0;
  DevModel M{};
  DevState S{};
  mssim_buffers buf{};
  std::vector<void*> allocs;
  std::vector<float> drive_host;
  float* d_drive = nullptr;
  bool panda = false;
  bool has_tri = false;  // the model has triangle-mesh shapes: control steps run the kernel variant with the mesh stage
  int rows_per_env = 1;  // 16-lane rows an env takes in the control-step kernel (its template parameter NR): 2 when the model has more than 16 velocity components
  bool dirty = true;
  int n_cu = 256;  // compute units of the device
  unsigned deferred_fetch = 0u;  // mssim_defer_fetch: copy-out owed to the next call on the handle
  // mssim_defer_step_action: a step_action owed to the next call; a task epilogue runs it, the copy-out and
  // itself as ONE launch of the control-step kernel
  const float* deferred_action = nullptr; int deferred_adim = 0, deferred_nsub = 0; hipStream_t deferred_stream = nullptr;
  std::vector<float> h_dof_pack; float* d_dof_pack = nullptr;  // (drive gains are patched by set_drive_properties)
  std::vector<int32_t> h_shape_row, h_pair_shape;  // host copies (contact-pair lists of the task epilogues)
  int* d_pick_pairs = nullptr; int n_pick_pairs = 0; int pick_rows[3] = {-1, -1, -1};
  std::vector<int*> queries;
  std::vector<int> query_n;
  std::vector<int> query_kind;
  std::string err;
  // profiling (bench roofline block): event pairs recorded on the launch stream
  int* d_act_col = nullptr; float* d_act_lo = nullptr; float* d_act_hi = nullptr; int* d_act_flags = nullptr;
  EeMap ee{-1, 0, 3, 0.f, 0.f, 0.f, 0};
  int act_max_col = -1;  // highest action column the joint map reads
  bool profiling = false;
  std::vector<hipEvent_t> ev[2];  // [kernel] start/stop interleaved
  size_t ev_used[2] = {0, 0};
};

static std::string g_create_error;

#define HIPCHK(h, call)                                                                   \
  do {                                                                                    \
    hipError_t _e = (call);                                                               \
    if (_e != hipSuccess) {                                                               \
      std::string _m = std::string(#call) + ": " + hipGetErrorString(_e);                 \
      if (h) (h)->err = _m; else g_create_error = _m;                                     \
      return 100 + (int)_e;                                                               \
    }                                                                                     \
  } while (0)

template <typename Tt>
static int upload(mssim_sim* S, const Tt* src, size_t count, const Tt** dst) {
  void* d = nullptr;
  size_t bytes = (count > 0 ? count : 1) * sizeof(Tt);
  HIPCHK(S, hipMalloc(&d, bytes));
  S->allocs.push_back(d);
  if (count > 0 && src) HIPCHK(S, hipMemcpy(d, src, count * sizeof(Tt), hipMemcpyHostToDevice));
  else HIPCHK(S, hipMemset(d, 0, bytes));
  *dst = (const Tt*)d;
  return 0;
}
template <typename Tt>
static int dalloc(mssim_sim* S, size_t count, Tt** dst) {
  void* d = nullptr;
  size_t bytes = (count > 0 ? count : 1) * sizeof(Tt);
  HIPCHK(S, hipMalloc(&d, bytes));
  HIPCHK(S, hipMemset(d, 0, bytes));
  S->allocs.push_back(d);
  *dst = (Tt*)d;
  return 0;
}

extern "C" {

int mssim_abi_version(void) { return MSSIM_ABI_VERSION; }
const char* mssim_last_error(mssim_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void mssim_destroy(mssim_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  for (void* p : h->allocs) (void)hipFree(p);
  for (int k = 0; k < 2; k++)
    for (auto& e : h->ev[k]) (void)hipEventDestroy(e);
  delete h;
}

int mssim_create(const mssim_model_desc* d, int32_t num_envs, int32_t device, mssim_handle* out) {
  if (!d || !out || num_envs <= 0) { g_create_error = "bad arguments"; return 1; }
  if (d->abi_version != MSSIM_ABI_VERSION) { g_create_error = "ABI version mismatch"; return 2; }
  if (d->n_dof > MSSIM_MAX_DOF || d->n_free > MSSIM_MAX_FREE) { g_create_error = "model exceeds MSSIM_MAX_DOF / MSSIM_MAX_FREE"; return 3; }
  // the control-step kernel keeps an env on one or two 16-lane rows (one velocity component per lane: the joints in row 0, the free
  // bodies behind them or, when that is more than 16 components, in a row of their own) and its scene in fixed LDS tables
  // (rows per env: 1 while joints and free bodies fit 16 lanes together; else the joints in row 0 and two free bodies per further row)
  const int rows_per_env = d->n_dof + 6 * d->n_free <= S16_LANES ? 1 : (d->n_free <= 2 ? 2 : 4);
  if (d->n_dof > S16_LANES || d->n_free > S16_MAX_FREE_(4) || d->n_kin > S16_MAX_KIN || d->n_shape > S16_MAX_SHAPE_(rows_per_env) || d->n_pair > 56 * 16) {
    char msg[320];
    snprintf(msg, sizeof msg, "model exceeds the control-step kernel's tables: %d joints (max %d), %d free bodies (max %d), %d kinematic bodies (max %d), "
             "%d shapes (max %d with %d velocity components), %d candidate pairs (max %d)", d->n_dof, S16_LANES, d->n_free, S16_MAX_FREE_(4), d->n_kin, S16_MAX_KIN, d->n_shape,
             S16_MAX_SHAPE_(rows_per_env), d->n_dof + 6 * d->n_free, d->n_pair, 56 * 16);
    g_create_error = msg;
    return 9;
  }
  for (int j = 0; j < d->n_dof; j++)
    if (d->dof_parent[j] >= j) { g_create_error = "dof_parent must be topologically sorted"; return 4; }
  for (int p = 0; p < d->n_pair; p++)
    if (d->pair_shape[2 * p] < 0 || d->pair_shape[2 * p] >= d->n_shape || d->pair_shape[2 * p + 1] < 0 || d->pair_shape[2 * p + 1] >= d->n_shape) {
      g_create_error = "pair_shape names a shape that does not exist";
      return 4;
    }
  for (int s = 0; s < d->n_shape; s++)
    if (d->shape_type[s] == MSSIM_SHAPE_CONVEX && (d->shape_hull[2 * s + 1] < 4 || d->shape_hull[2 * s + 1] > MSSIM_MAX_HULL_VERTS)) {
      g_create_error = "convex hull vertex count out of range";
      return 5;
    }
  // tables the kernels index without checks (a malformed model gets an error string here, not an out-of-bounds device read)
  if (d->n_tri_node >= (1 << 17) || d->n_tri >= (1 << 24)) { g_create_error = "triangle meshes: more than 131071 BVH nodes or 16777215 triangles (the shape word packs the root node into 17 bits)"; return 5; }
  for (int nd = 0; nd < d->n_tri_node; nd++)
    for (int c = 0; c < 16; c++) {
      const float* nb = d->tri_bvh + (size_t)nd * 112;
      if (!(nb[6 * c] <= nb[6 * c + 3])) continue;  // (min > max: no child)
      int32_t ref;
      std::memcpy(&ref, nb + 96 + c, 4);
      if (ref >= 0 ? ref >= d->n_tri_node : ~ref >= d->n_tri) { g_create_error = "tri_bvh: a child reference points outside the node / triangle tables"; return 5; }
    }
  if (d->n_env_shape > 0)
    for (int s = 0; s < d->n_shape; s++)
      if (d->shape_env_slot[s] >= d->n_env_shape) { g_create_error = "shape_env_slot names a slot beyond n_env_shape"; return 5; }
  if (d->n_env_free > 0)
    for (int b = 0; b < d->n_free; b++)
      if (d->free_env_slot[b] >= d->n_env_free) { g_create_error = "free_env_slot names a slot beyond n_env_free"; return 5; }
  for (int s = 0; s < d->n_shape; s++)
    if (d->shape_type[s] == MSSIM_SHAPE_CONVEX && (d->shape_hull[2 * s] < 0 || d->shape_hull[2 * s] + d->shape_hull[2 * s + 1] > d->n_hull_verts)) {
      g_create_error = "convex hull: vertex range outside hull_verts";
      return 5;
    }
  mssim_sim* S = new mssim_sim();
  S->device = device;
  S->N = num_envs;
  S->rows_per_env = rows_per_env;
  hipError_t e0 = hipSetDevice(device);
  if (e0 != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e0); delete S; return 6; }
  { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) S->n_cu = ncu; }
  DevModel& M = S->M;
  M.n_dof = d->n_dof; M.n_tendon = d->n_tendon; M.n_link = d->n_link; M.n_free = d->n_free; M.n_kin = d->n_kin;
  M.n_shape = d->n_shape; M.n_pair = d->n_pair;
  const int n = d->n_dof, ns = d->n_shape;
  std::vector<unsigned> anc(n > 0 ? n : 1, 0u);
  for (int j = 0; j < n; j++)
    for (int i = d->dof_parent[j]; i >= 0; i = d->dof_parent[i]) anc[j] |= 1u << i;
  int rc = 0;
#define UP(field, count) if ((rc = upload(S, d->field, (size_t)(count), &M.field))) { g_create_error = S->err; mssim_destroy(S); return rc; }
  UP(dof_parent, n) UP(dof_type, n) UP(body_gravity, n) UP(tendon_dof, 2 * d->n_tendon) UP(link_body, d->n_link) UP(free_gravity, d->n_free)
  UP(dof_frame, 7 * n) UP(dof_axis, 3 * n) UP(dof_limit, 2 * n) UP(dof_drive, 4 * n) UP(dof_armature, n) UP(body_inertial, 10 * n)
  UP(tendon_param, 5 * d->n_tendon) UP(link_frame, 7 * d->n_link) UP(free_inertial, 10 * d->n_free) UP(free_damping, 2 * d->n_free)
  UP(shape_type, ns) UP(shape_row, ns) UP(pair_shape, 2 * d->n_pair)
  S->h_shape_row.assign(d->shape_row, d->shape_row + ns);
  S->h_pair_shape.assign(d->pair_shape, d->pair_shape + 2 * (size_t)d->n_pair);
  UP(shape_frame, 7 * ns) UP(shape_param, 4 * ns) UP(shape_material, 4 * ns) UP(shape_bound, 4 * ns)
#undef UP
  // device copy of env_shape_param (row format below)
  std::vector<float> env_param_dev;
  {
    // hull vertices are repacked so that every hull starts on a multiple of 8 vertices and is padded
    // to a multiple of 8 with copies of its vertex 0: `support()` reads whole 8-vertex batches as six
    // aligned 16-byte loads, and a copy of vertex 0 can never win its strict first-maximum scan
    std::vector<float> hv;
    std::vector<int32_t> sh(2 * (ns > 0 ? ns : 1), 0);
    std::vector<std::pair<std::pair<int, int>, int>> seen;  // (start, count) -> new start
    auto repacked = [&](int st, int cnt) {
      for (auto& kv : seen) if (kv.first.first == st && kv.first.second == cnt) return kv.second;
      const int found = (int)(hv.size() / 3);
      for (int i = 0; i < (cnt + 7) / 8 * 8; i++) {
        const float* v = d->hull_verts + 3 * (size_t)(st + (i < cnt ? i : 0));
        hv.push_back(v[0]); hv.push_back(v[1]); hv.push_back(v[2]);
      }
      seen.push_back({{st, cnt}, found});
      return found;
    };
    for (int s2 = 0; s2 < ns; s2++) {
      const int st = d->shape_hull[2 * s2], cnt = d->shape_hull[2 * s2 + 1];
      if (d->shape_type[s2] == MSSIM_SHAPE_TRIMESH) {  // (root node of its BVH, no hull vertices)
        if (st < 0 || st >= d->n_tri_node || d->shape_body_kind[s2] == MSSIM_BODY_FREE || d->shape_body_kind[s2] == MSSIM_BODY_ART) {
          g_create_error = "triangle mesh: BVH root out of range, or the mesh belongs to a moving body (fixed / kinematic bodies only)";
          mssim_destroy(S);
          return 8;
        }
        sh[2 * s2] = st;
        sh[2 * s2 + 1] = 0;
        S->has_tri = true;
        continue;
      }
      sh[2 * s2 + 1] = cnt;
      if (cnt <= 0) continue;
      sh[2 * s2] = repacked(st, cnt);
    }
    if (d->n_env_shape > 0 && d->num_envs == num_envs) {
      // device rows of every per-env shape: word 0 = type | vertex count << 3 | first vertex << 10 (bit pattern),
      // rows 1..3 = the three parameters of a primitive, or the half extents of a hull's box for the cull
      const size_t NE = (size_t)num_envs;
      env_param_dev.assign(d->env_shape_param, d->env_shape_param + (size_t)4 * d->n_env_shape * NE);
      for (int s2 = 0; s2 < ns; s2++) {
        const int slot = d->shape_env_slot[s2];
        if (slot < 0) continue;
        for (size_t e = 0; e < NE; e++) {
          const float* src = d->env_shape_param;
          const int tcode = (int)src[(size_t)(4 * slot + 3) * NE + e];
          const int type = tcode > 0 ? tcode - 1 : d->shape_type[s2];
          // (a triangle mesh only in a slot whose own type is TRIMESH: the mesh variant of the kernel is chosen by the shared types)
          const bool env_mesh = type == MSSIM_SHAPE_TRIMESH && d->shape_type[s2] == MSSIM_SHAPE_TRIMESH;
          if ((type < MSSIM_SHAPE_BOX || type > MSSIM_SHAPE_NONE) && !env_mesh) { g_create_error = "per-env shape type out of range (planes cannot be per-env shapes; a triangle mesh only in a slot that is a triangle mesh)"; mssim_destroy(S); return 8; }
          int32_t word = type;
          float rows[3] = {src[(size_t)(4 * slot) * NE + e], src[(size_t)(4 * slot + 1) * NE + e], src[(size_t)(4 * slot + 2) * NE + e]};
          if (type == MSSIM_SHAPE_NONE) rows[0] = rows[1] = rows[2] = 0.f;
          if (env_mesh) {
            // this env's mesh: rows = first triangle, triangle count, root node of its BVH. The device row gets the root in
            // the word's upper bits and the mesh's half extents about its bound centre (shape frame) in the rows.
            const int first = (int)rows[0], count = (int)rows[1], root = (int)rows[2];
            if (first < 0 || count < 0 || first + count > d->n_tri || root < 0 || root >= d->n_tri_node) { g_create_error = "per-env triangle mesh: triangle range / root node out of range"; mssim_destroy(S); return 8; }
            float fr[7], cb[3];
            for (int k = 0; k < 7; k++) fr[k] = d->env_shape_frame[(size_t)(7 * slot + k) * NE + e];
            for (int k = 0; k < 3; k++) cb[k] = d->env_shape_bound[(size_t)(4 * slot + k) * NE + e] - fr[k];
            const float nq = std::sqrt(fr[3] * fr[3] + fr[4] * fr[4] + fr[5] * fr[5] + fr[6] * fr[6]);
            const float qw = fr[3] / nq, qx = fr[4] / nq, qy = fr[5] / nq, qz = fr[6] / nq;
            const float R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)},
                                   {2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)},
                                   {2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)}};
            float cs[3];
            for (int k = 0; k < 3; k++) cs[k] = R[0][k] * cb[0] + R[1][k] * cb[1] + R[2][k] * cb[2];
            rows[0] = rows[1] = rows[2] = 0.f;
            for (int t = first; t < first + count; t++) {
              const float* q = d->tri_soup + 12 * (size_t)t;
              for (int c3 = 0; c3 < 3; c3++)
                for (int k = 0; k < 3; k++) rows[k] = std::max(rows[k], std::fabs(q[k] + q[3 + 3 * c3 + k] - cs[k]));
            }
            word |= root << 10;
          }
          if (type == MSSIM_SHAPE_CONVEX) {
            const int st = (int)rows[0], cnt = (int)rows[1];
            if (cnt < 1 || cnt > MSSIM_MAX_HULL_VERTS || st < 0 || st + cnt > d->n_hull_verts) { g_create_error = "per-env hull reference out of range"; mssim_destroy(S); return 8; }
            const int found = repacked(st, cnt);
            // bound centre (body frame) back into the shape frame, then the extents of the hull about it
            float fr[7], cb[3];
            for (int k = 0; k < 7; k++) fr[k] = d->env_shape_frame[(size_t)(7 * slot + k) * NE + e];
            for (int k = 0; k < 3; k++) cb[k] = d->env_shape_bound[(size_t)(4 * slot + k) * NE + e] - fr[k];
            const float nq = std::sqrt(fr[3] * fr[3] + fr[4] * fr[4] + fr[5] * fr[5] + fr[6] * fr[6]);
            const float qw = fr[3] / nq, qx = fr[4] / nq, qy = fr[5] / nq, qz = fr[6] / nq;
            const float R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)},
                                   {2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)},
                                   {2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)}};
            float cs[3];
            for (int k = 0; k < 3; k++) cs[k] = R[0][k] * cb[0] + R[1][k] * cb[1] + R[2][k] * cb[2];
            rows[0] = rows[1] = rows[2] = 0.f;
            for (int i = 0; i < cnt; i++) {
              const float* v = d->hull_verts + 3 * (size_t)(st + i);
              for (int k = 0; k < 3; k++) rows[k] = std::max(rows[k], std::fabs(v[k] - cs[k]));
            }
            word |= (cnt << 3) | (found << 10);
          }
          float wf;
          std::memcpy(&wf, &word, 4);
          env_param_dev[(size_t)(4 * slot) * NE + e] = wf;
          for (int k = 0; k < 3; k++) env_param_dev[(size_t)(4 * slot + 1 + k) * NE + e] = rows[k];
        }
      }
    }
    if (hv.size() / 3 >= (1u << 17)) { g_create_error = "too many hull vertices"; mssim_destroy(S); return 8; }
    if ((rc = upload(S, sh.data(), sh.size(), &M.shape_hull)) || (rc = upload(S, hv.data(), hv.size(), &M.hull_verts))) { g_create_error = S->err; mssim_destroy(S); return rc; }
  }
  if ((rc = upload(S, d->shape_body_kind, (size_t)ns, &M.shape_kind))) { g_create_error = S->err; mssim_destroy(S); return rc; }
  if ((rc = upload(S, d->shape_body_index, (size_t)ns, &M.shape_index))) { g_create_error = S->err; mssim_destroy(S); return rc; }
  if ((rc = upload(S, anc.data(), (size_t)n, &M.dof_anc))) { g_create_error = S->err; mssim_destroy(S); return rc; }
  {
    std::vector<float> ctr(3 * (ns > 0 ? ns : 1), 0.f);
    for (int s2 = 0; s2 < ns; s2++) {
      const float* f = d->shape_frame + 7 * s2;
      const float* b = d->shape_bound + 4 * s2;
      const float w = f[3], x = f[4], y = f[5], z = f[6];
      const float nq = std::sqrt(w * w + x * x + y * y + z * z);
      const float qw = w / nq, qx = x / nq, qy = y / nq, qz = z / nq;
      const float R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)},
                             {2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)},
                             {2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)}};
      for (int i = 0; i < 3; i++) ctr[3 * s2 + i] = f[i] + R[i][0] * b[0] + R[i][1] * b[1] + R[i][2] * b[2];
    }
    if ((rc = upload(S, ctr.data(), (size_t)3 * ns, &M.shape_center))) { g_create_error = S->err; mssim_destroy(S); return rc; }
    // oriented bounding boxes for the cull (shape frame axes, centred at the bound centre)
    std::vector<float> half(3 * (ns > 0 ? ns : 1), 0.f);
    for (int s2 = 0; s2 < ns; s2++) {
      const float* pp = d->shape_param + 4 * s2;
      const float* b = d->shape_bound + 4 * s2;
      float* h = &half[3 * s2];
      switch (d->shape_type[s2]) {
        case MSSIM_SHAPE_BOX: h[0] = pp[0]; h[1] = pp[1]; h[2] = pp[2]; break;
        case MSSIM_SHAPE_SPHERE: h[0] = h[1] = h[2] = pp[0]; break;
        case MSSIM_SHAPE_CAPSULE: h[0] = pp[1] + pp[0]; h[1] = h[2] = pp[0]; break;
        case MSSIM_SHAPE_CYLINDER: h[0] = pp[1]; h[1] = h[2] = pp[0]; break;
        case MSSIM_SHAPE_CONVEX:
          for (int i = 0; i < d->shape_hull[2 * s2 + 1]; i++) {
            const float* v = d->hull_verts + 3 * (size_t)(d->shape_hull[2 * s2] + i);
            for (int k = 0; k < 3; k++) h[k] = std::max(h[k], std::fabs(v[k] - b[k]));
          }
          break;
        case MSSIM_SHAPE_TRIMESH: {
          const int first = (int)pp[0], count = (int)pp[1];
          if (first < 0 || count < 0 || first + count > d->n_tri) { g_create_error = "triangle mesh: triangle range out of tri_soup"; mssim_destroy(S); return 8; }
          for (int t = first; t < first + count; t++) {
            const float* q = d->tri_soup + 12 * (size_t)t;
            for (int c3 = 0; c3 < 3; c3++)
              for (int k = 0; k < 3; k++) h[k] = std::max(h[k], std::fabs(q[k] + q[3 + 3 * c3 + k] - b[k]));
          }
          break;
        }
        default: h[0] = h[1] = h[2] = 3e30f;  // plane: never used
      }
      // primitive shapes are centred on their frame; keep the box valid if the bound centre is offset
      if (d->shape_type[s2] != MSSIM_SHAPE_CONVEX && d->shape_type[s2] != MSSIM_SHAPE_PLANE && d->shape_type[s2] != MSSIM_SHAPE_TRIMESH)
        for (int k = 0; k < 3; k++) h[k] += std::fabs(b[k]);
    }
    if ((rc = upload(S, half.data(), (size_t)3 * ns, &M.shape_half))) { g_create_error = S->err; mssim_destroy(S); return rc; }
    if ((rc = upload(S, d->tri_soup, (size_t)12 * d->n_tri, &M.tri_soup)) || (rc = upload(S, d->tri_bvh, (size_t)112 * d->n_tri_node, &M.tri_bvh))) { g_create_error = S->err; mssim_destroy(S); return rc; }
    // packed constant records (one or two cache lines per joint / shape instead of ~10 arrays)
    auto fbits = [](int32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
    std::vector<float> sp(24 * (size_t)(ns > 0 ? ns : 1), 0.f);
    for (int s2 = 0; s2 < ns; s2++) {
      float* r = &sp[24 * (size_t)s2];
      for (int k = 0; k < 7; k++) r[k] = d->shape_frame[7 * s2 + k];
      for (int k = 0; k < 3; k++) { r[7 + k] = d->shape_param[4 * s2 + k]; r[10 + k] = ctr[3 * s2 + k]; r[14 + k] = half[3 * s2 + k]; }
      r[13] = d->shape_bound[4 * s2 + 3];
      r[17] = d->shape_material[4 * s2 + 1];
      r[18] = fbits(d->shape_type[s2]);
      r[19] = fbits(d->shape_body_kind[s2]);
      r[20] = fbits(d->shape_body_index[s2]);
      r[21] = fbits((d->n_env_shape > 0) ? d->shape_env_slot[s2] : -1);
      r[22] = d->shape_material[4 * s2 + 3];  // torsional patch radius
    }
    if ((rc = upload(S, sp.data(), sp.size(), &M.shape_pack))) { g_create_error = S->err; mssim_destroy(S); return rc; }
    S->h_dof_pack.assign(32 * (size_t)(n > 0 ? n : 1), 0.f);
    for (int j = 0; j < n; j++) {
      float* r = &S->h_dof_pack[32 * (size_t)j];
      for (int k = 0; k < 7; k++) r[k] = d->dof_frame[7 * j + k];
      for (int k = 0; k < 3; k++) r[7 + k] = d->dof_axis[3 * j + k];
      r[10] = fbits(d->dof_parent[j]); r[11] = fbits(d->dof_type[j]); r[12] = fbits((int32_t)anc[j]);
      for (int k = 0; k < 4; k++) r[13 + k] = d->dof_drive[4 * j + k];
      r[17] = d->dof_armature[j]; r[18] = d->dof_limit[2 * j]; r[19] = d->dof_limit[2 * j + 1];
      for (int k = 0; k < 10; k++) r[20 + k] = d->body_inertial[10 * j + k];
      r[30] = fbits(d->body_gravity[j]);
    }
    if ((rc = upload(S, S->h_dof_pack.data(), S->h_dof_pack.size(), &M.dof_pack))) { g_create_error = S->err; mssim_destroy(S); return rc; }
    S->d_dof_pack = const_cast<float*>(M.dof_pack);
  }
  {
    const bool has_es = d->n_env_shape > 0, has_ef = d->n_env_free > 0;
    if ((has_es || has_ef) && d->num_envs != num_envs) { g_create_error = "per-env arrays were built for a different num_envs"; mssim_destroy(S); return 7; }
    std::vector<int32_t> sslot(ns > 0 ? ns : 1, -1), fslot(d->n_free > 0 ? d->n_free : 1, -1);
    if (has_es) for (int i = 0; i < ns; i++) sslot[i] = d->shape_env_slot[i];
    if (has_ef) for (int i = 0; i < d->n_free; i++) fslot[i] = d->free_env_slot[i];
    if ((rc = upload(S, sslot.data(), sslot.size(), &M.shape_env_slot)) || (rc = upload(S, fslot.data(), fslot.size(), &M.free_env_slot)) ||
        (rc = upload(S, d->env_shape_frame, (size_t)7 * d->n_env_shape * num_envs, &M.env_shape_frame)) ||
        (rc = upload(S, has_es ? env_param_dev.data() : d->env_shape_param, (size_t)4 * d->n_env_shape * num_envs, &M.env_shape_param)) ||
        (rc = upload(S, d->env_shape_bound, (size_t)4 * d->n_env_shape * num_envs, &M.env_shape_bound)) ||
        (rc = upload(S, d->env_free_inertial, (size_t)10 * d->n_env_free * num_envs, &M.env_free_inertial))) {
      mssim_destroy(S);
      return rc;
    }
  }
  S->d_drive = const_cast<float*>(M.dof_drive);
  M.gx = d->gravity[0]; M.gy = d->gravity[1]; M.gz = d->gravity[2];
  M.dt = d->timestep; M.contact_offset = d->contact_offset; M.rest_offset = d->rest_offset; M.erp = d->erp;
  M.max_depen = d->max_depenetration_velocity; M.pos_iters = d->position_iterations; M.vel_iters = d->velocity_iterations;
  M.sleep_threshold = d->sleep_threshold;
  S->panda = (n == 9);
  for (int j = 0; j < n && S->panda; j++)
    if (d->dof_parent[j] != kPandaParent[j] || d->dof_type[j] != kPandaType[j]) S->panda = false;
  DevState& D = S->S;
  D.N = num_envs;
  const size_t N = (size_t)num_envs;
#define AL(field, count) if ((rc = dalloc(S, (size_t)(count) * N, &D.field))) { g_create_error = S->err; mssim_destroy(S); return rc; }
  AL(root, 7) AL(q, n) AL(qd, n) AL(qt, n) AL(qdt, n) AL(qf, n) AL(qacc, n)
  AL(free_s, 13 * d->n_free) AL(free_force, 3 * d->n_free) AL(kin, 7 * d->n_kin) AL(free_wake, d->n_free)
  AL(bodypose, 7 * n) AL(bodyvel, 6 * n) AL(bodyaux, 6 * n)
  AL(pair_cnt, d->n_pair) AL(pair_imp, 3 * d->n_pair)
  if ((rc = dalloc(S, (size_t)num_envs * S16_ROWS_GLB * S16_ROWLEN_(S->rows_per_env), &D.rows))) { g_create_error = S->err; mssim_destroy(S); return rc; }
  AL(overflow, 1)
  AL(hit_list, 1 + MAXC)
  AL(pcm_tick, 1)
  if ((rc = dalloc(S, (size_t)num_envs * MSSIM_PCM_SLOTS * S16_PCM_LEN, &D.pcm))) { g_create_error = S->err; mssim_destroy(S); return rc; }
#define HIPCHK_NEW(call) do { hipError_t _e = (call); if (_e != hipSuccess) { g_create_error = std::string(#call) + ": " + hipGetErrorString(_e); mssim_destroy(S); return 100 + (int)_e; } } while (0)
  HIPCHK_NEW(hipMemset(D.pcm, 0xFF, (size_t)num_envs * MSSIM_PCM_SLOTS * S16_PCM_LEN * sizeof(float)));  // pair = -1: every slot empty
  {
    const size_t nw = (size_t)4 * (d->n_pair > 0 ? d->n_pair : 1) * N * 4;
    if ((rc = dalloc(S, nw, &D.warm))) { g_create_error = S->err; mssim_destroy(S); return rc; }
    HIPCHK_NEW(hipMemset(D.warm, 0xFF, nw * sizeof(float)));  // stamp -1: nothing to start from
  }
  {
    // clearance of every (convex shape, mesh) pair (mssim_solve16.h stage T0); NaN bit pattern = no clearance known
    std::vector<int32_t> slot((size_t)(d->n_pair > 0 ? d->n_pair : 1), -1);
    int n_mesh_pair = 0;
    for (int p = 0; p < d->n_pair; p++)
      if (d->shape_type[d->pair_shape[2 * p + 1]] == MSSIM_SHAPE_TRIMESH) slot[p] = n_mesh_pair++;
    if ((rc = upload(S, slot.data(), slot.size(), &M.pair_mesh_slot))) { g_create_error = S->err; mssim_destroy(S); return rc; }
    std::vector<int32_t> packed((size_t)((d->n_pair + 127) / 128 * 128 + 128), -1);  // (read in chunks of 8 rounds of 16)
    for (int p = 0; p < d->n_pair; p++) packed[p] = d->pair_shape[2 * p] | (d->pair_shape[2 * p + 1] << 8);
    if ((rc = upload(S, packed.data(), packed.size(), &M.pair_packed))) { g_create_error = S->err; mssim_destroy(S); return rc; }
    const size_t nc = (size_t)4 * (n_mesh_pair > 0 ? n_mesh_pair : 1) * N;
    if ((rc = dalloc(S, nc, &D.tri_clear))) { g_create_error = S->err; mssim_destroy(S); return rc; }
    HIPCHK_NEW(hipMemset(D.tri_clear, 0xFF, nc * sizeof(float)));
#undef HIPCHK_NEW
  }
#undef AL
  // identity quaternions
  std::vector<float> ones(N, 1.0f);
  hipMemcpy(D.root + 3 * N, ones.data(), N * sizeof(float), hipMemcpyHostToDevice);
  for (int b = 0; b < d->n_free; b++) hipMemcpy(D.free_s + (13 * b + 3) * N, ones.data(), N * sizeof(float), hipMemcpyHostToDevice);
  {
    std::vector<float> awake(N * (size_t)(d->n_free > 0 ? d->n_free : 1), MSSIM_WAKE_TIME);
    if (d->n_free > 0) hipMemcpy(D.free_wake, awake.data(), awake.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  for (int k = 0; k < d->n_kin; k++) hipMemcpy(D.kin + (7 * k + 3) * N, ones.data(), N * sizeof(float), hipMemcpyHostToDevice);
  *out = S;
  return 0;
}

static void flush_deferred(mssim_handle h, hipStream_t st);
int mssim_bind_buffers(mssim_handle h, const mssim_buffers* b) {
  if (h) flush_deferred(h, (hipStream_t)0);
  if (!h || !b) return 1;
  h->buf = *b;
  return 0;
}

int mssim_set_timestep(mssim_handle h, float dt) {
  flush_deferred(h, h->deferred_stream);
  if (!(dt > 0.f)) { h->err = "timestep must be positive"; return 1; }
  h->M.dt = dt;
  return 0;
}
float mssim_get_timestep(mssim_handle h) { return h->M.dt; }

int mssim_set_drive_properties(mssim_handle h, const float* drive) {
  flush_deferred(h, h->deferred_stream);
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy(h->d_drive, drive, sizeof(float) * 4 * h->M.n_dof, hipMemcpyHostToDevice));
  for (int j = 0; j < h->M.n_dof; j++)
    for (int k = 0; k < 4; k++) h->h_dof_pack[32 * (size_t)j + 13 + k] = drive[4 * j + k];
  HIPCHK(h, hipMemcpy(h->d_dof_pack, h->h_dof_pack.data(), sizeof(float) * h->h_dof_pack.size(), hipMemcpyHostToDevice));
  return 0;
}

static inline int pad8(int n) { return (n + 7) / 8 * 8; }
static inline dim3 env_grid(int N, int block) { return dim3(pad8((N + block - 1) / block)); }  // kernels map blocks with xcd_chunk

int mssim_apply(mssim_handle h, uint32_t what, void* stream) {
  flush_deferred(h, (hipStream_t)stream);
  hipLaunchKernelGGL(k_apply, env_grid(h->N, 256), dim3(256), 0, (hipStream_t)stream, h->M, h->S, h->buf, what);
  h->dirty = true;
  HIPCHK(h, hipGetLastError());
  return 0;
}

// Deferred fetch (mssim_defer_fetch): the copy-out is owed until the next call on the handle. A task epilogue
// performs it inside its own launch (k_task_*<true>); every other entry point that touches the state or
// the buffers performs it first, so the deferral is only an ordering of launches, never a change of results.
static unsigned take_deferred_fetch(mssim_handle h) {
  const unsigned w = h->deferred_fetch;
  h->deferred_fetch = 0u;
  return w;
}
static void flush_deferred_fetch(mssim_handle h, hipStream_t st) {
  if (const unsigned w = take_deferred_fetch(h))
    hipLaunchKernelGGL(k_fetch, dim3(pad8((h->N + 63) / 64), h->M.n_link + h->M.n_free + h->M.n_kin + 1), dim3(64), 0, st, h->M, h->S, h->buf, w);
}
int mssim_defer_fetch(mssim_handle h, uint32_t what) {
  h->deferred_fetch |= what;
  return 0;
}

int mssim_fetch(mssim_handle h, uint32_t what, void* stream) {
  what |= take_deferred_fetch(h);
  flush_deferred(h, (hipStream_t)stream);
  hipLaunchKernelGGL(k_fetch, dim3(pad8((h->N + 63) / 64), h->M.n_link + h->M.n_free + h->M.n_kin + 1), dim3(64), 0, (hipStream_t)stream, h->M, h->S, h->buf, what);
  HIPCHK(h, hipGetLastError());
  return 0;
}

static void launch_fk(mssim_handle h, hipStream_t st) {
  if (h->panda) hipLaunchKernelGGL(k_fk<TopoPanda>, env_grid(h->N, 64), dim3(64), 0, st, h->M, h->S);
  else hipLaunchKernelGGL(k_fk<TopoDyn>, env_grid(h->N, 64), dim3(64), 0, st, h->M, h->S);
}

int mssim_wake_all(mssim_handle h, void* stream) {
  flush_deferred(h, (hipStream_t)stream);
  const size_t cnt = (size_t)h->M.n_free * h->N;
  if (cnt > 0) hipLaunchKernelGGL(k_fill, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->S.free_wake, MSSIM_WAKE_TIME, cnt);
  HIPCHK(h, hipMemsetAsync(h->S.pcm, 0xFF, (size_t)h->N * MSSIM_PCM_SLOTS * S16_PCM_LEN * sizeof(float), (hipStream_t)stream));  // every slot empty
  HIPCHK(h, hipMemsetAsync(h->S.warm, 0xFF, (size_t)16 * (h->M.n_pair > 0 ? h->M.n_pair : 1) * h->N * sizeof(float), (hipStream_t)stream));
  HIPCHK(h, hipGetLastError());
  return 0;
}

// hidden state of the listed envs back to "fresh": every free body awake, manifold cache empty, no multipliers to start from
__global__ void k_wake_envs(DevModel M, DevState S, const long long* __restrict__ idx, int n_idx) {
  const int N = S.N;
  if ((int)blockIdx.x >= n_idx) return;
  const long long e64 = idx[blockIdx.x];
  if (e64 < 0 || e64 >= N) return;
  const int e = (int)e64;
  for (int b = threadIdx.x; b < M.n_free; b += blockDim.x) SOA(S.free_wake, b) = MSSIM_WAKE_TIME;
  for (int s = threadIdx.x; s < MSSIM_PCM_SLOTS; s += blockDim.x) reinterpret_cast<int*>(S.pcm)[((size_t)e * MSSIM_PCM_SLOTS + s) * S16_PCM_LEN] = -1;  // pair = -1: slot empty
  for (int k = threadIdx.x; k < 4 * M.n_pair; k += blockDim.x) reinterpret_cast<int*>(S.warm)[((size_t)k * N + e) * 4 + 3] = -1;  // stamp -1
}
int mssim_wake_envs(mssim_handle h, const int64_t* env_idx, int32_t n_idx, void* stream) {
  flush_deferred(h, (hipStream_t)stream);
  if (n_idx <= 0) return 0;
  if (!env_idx) { h->err = "wake_envs: no index array"; return 1; }
  hipLaunchKernelGGL(k_wake_envs, dim3((unsigned)n_idx), dim3(256), 0, (hipStream_t)stream, h->M, h->S, reinterpret_cast<const long long*>(env_idx), (int)n_idx);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int mssim_update_kinematics(mssim_handle h, void* stream) {
  flush_deferred(h, (hipStream_t)stream);
  launch_fk(h, (hipStream_t)stream);
  h->dirty = false;
  HIPCHK(h, hipGetLastError());
  return 0;
}

static inline void prof_mark(mssim_handle h, int k, hipStream_t st) {
  if (!h->profiling || h->ev_used[k] >= h->ev[k].size()) return;
  (void)hipEventRecord(h->ev[k][h->ev_used[k]++], st);
}

extern "C++" {
template <int TASK>
static void launch_control_step(mssim_handle h, const DevState& S, int n_substeps, hipStream_t st);
}
int mssim_step(mssim_handle h, int32_t n_substeps, void* stream) {
  flush_deferred(h, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  if (h->dirty) { launch_fk(h, st); h->dirty = false; }
  if (n_substeps > 0) launch_control_step<0>(h, h->S, (int)n_substeps, st);
  HIPCHK(h, hipGetLastError());
  return 0;
}

// TASK: 0 = plain control step, 1 / 2 / 3 = copy-out + PickCube / PushCube / PegInsertionSide epilogue at its tail
extern "C++" {
template <int TASK>
static void launch_control_step(mssim_handle h, const DevState& S, int n_substeps, hipStream_t st) {
  prof_mark(h, 0, st);
  const dim3 block(64 * S16_WAVES);
#ifdef MSSIM_ONLY_PANDA
  // (timing experiments, scripts/ab_variants.sh: only the benchmark's instances are compiled -- a fifth of the build time)
  hipLaunchKernelGGL((k_solve16<9, TASK>), env_grid(h->N, S16_WAVES * S16_ENVS_PER_BLOCK), block, 0, st, h->M, S, n_substeps);
#else
  if (h->rows_per_env == 4) {  // three to six free bodies: four 16-lane rows (a whole wave) per env, 4 envs per block; the generic-topology instances
    const dim3 grid4 = env_grid(h->N, S16_WAVES * S16_ENVS_PER_BLOCK / 4);
    if (h->has_tri) hipLaunchKernelGGL((k_solve16<0, 0, true, 4>), grid4, block, 0, st, h->M, S, n_substeps);
    else hipLaunchKernelGGL((k_solve16<0, 0, false, 4>), grid4, block, 0, st, h->M, S, n_substeps);
    prof_mark(h, 0, st);
    return;
  }
  if (h->rows_per_env == 2) {  // more than 16 velocity components: two 16-lane rows per env, 8 envs per block
    const dim3 grid2 = env_grid(h->N, S16_WAVES * S16_ENVS_PER_BLOCK / 2);
    if (h->has_tri) {
      if (h->M.n_dof == 9) hipLaunchKernelGGL((k_solve16<9, 0, true, 2>), grid2, block, 0, st, h->M, S, n_substeps);
      else if (h->M.n_dof == 15) hipLaunchKernelGGL((k_solve16<15, 0, true, 2>), grid2, block, 0, st, h->M, S, n_substeps);
      else hipLaunchKernelGGL((k_solve16<0, 0, true, 2>), grid2, block, 0, st, h->M, S, n_substeps);
    } else if (h->M.n_dof == 9) hipLaunchKernelGGL((k_solve16<9, 0, false, 2>), grid2, block, 0, st, h->M, S, n_substeps);
    else if (h->M.n_dof == 15) hipLaunchKernelGGL((k_solve16<15, 0, false, 2>), grid2, block, 0, st, h->M, S, n_substeps);
    else hipLaunchKernelGGL((k_solve16<0, 0, false, 2>), grid2, block, 0, st, h->M, S, n_substeps);
    prof_mark(h, 0, st);
    return;
  }
  const dim3 grid = env_grid(h->N, S16_WAVES * S16_ENVS_PER_BLOCK);
  if (h->has_tri) {  // models with triangle meshes: the variant that carries the mesh stage (never with a task tail)
    if (h->M.n_dof == 9) hipLaunchKernelGGL((k_solve16<9, 0, true>), grid, block, 0, st, h->M, S, n_substeps);
    else if (h->M.n_dof == 15) hipLaunchKernelGGL((k_solve16<15, 0, true>), grid, block, 0, st, h->M, S, n_substeps);
    else hipLaunchKernelGGL((k_solve16<0, 0, true>), grid, block, 0, st, h->M, S, n_substeps);
  } else if (h->M.n_dof == 9) hipLaunchKernelGGL((k_solve16<9, TASK>), grid, block, 0, st, h->M, S, n_substeps);
  else if (h->M.n_dof == 15) hipLaunchKernelGGL((k_solve16<15, 0>), grid, block, 0, st, h->M, S, n_substeps);  // (the Fetch)
  else hipLaunchKernelGGL((k_solve16<0, 0>), grid, block, 0, st, h->M, S, n_substeps);
#endif
  prof_mark(h, 0, st);
}
}  // extern "C++"
static DevState state_with_action(mssim_handle h, const float* action, int action_dim) {
  DevState S = h->S;
  S.act = action; S.act_dim = action_dim;
  S.act_col = h->d_act_col; S.act_lo = h->d_act_lo; S.act_hi = h->d_act_hi; S.act_flags = h->d_act_flags;
  S.act_qpos = h->buf.art_qpos; S.act_target = h->buf.art_target_qpos; S.act_target_vel = h->buf.art_target_qvel;
  return S;
}
// every column the maps read must exist: the kernels index action[env * action_dim + column] unchecked
static int check_action_dim(mssim_handle h, int32_t action_dim) {
  if (!h->d_act_col) { h->err = "set_action_map has not been called"; return 1; }
  const int need = std::max(h->act_max_col + 1, h->ee.link >= 0 ? h->ee.col0 + h->ee.rows : 0);
  if (action_dim < need) {
    h->err = "action has " + std::to_string(action_dim) + " columns, the action map reads " + std::to_string(need);
    return 2;
  }
  return 0;
}
static int step_action_now(mssim_handle h, const float* action, int32_t action_dim, int32_t n_substeps, hipStream_t st) {
  if (n_substeps <= 0 || h->ee.link >= 0) {  // end-effector block: apply_action, then step
    hipLaunchKernelGGL(k_apply_action, env_grid(h->N, 256), dim3(256), 0, st, h->M, h->S, h->buf, action, action_dim,
                       h->d_act_col, h->d_act_lo, h->d_act_hi, h->d_act_flags, h->ee);
    return mssim_step(h, n_substeps, st);
  }
  if (h->dirty) { launch_fk(h, st); h->dirty = false; }
  launch_control_step<0>(h, state_with_action(h, action, action_dim), n_substeps, st);
  HIPCHK(h, hipGetLastError());
  return 0;
}
// everything owed to the handle (see mssim_defer_step_action / mssim_defer_fetch), in order
static void flush_deferred(mssim_handle h, hipStream_t st) {
  if (h->deferred_action) {
    const float* a = h->deferred_action;
    h->deferred_action = nullptr;
    (void)step_action_now(h, a, h->deferred_adim, h->deferred_nsub, h->deferred_stream);
  }
  flush_deferred_fetch(h, st);
}

int mssim_step_action(mssim_handle h, const float* action, int32_t action_dim, int32_t n_substeps, void* stream) {
  if (int rc = check_action_dim(h, action_dim)) return rc;
  flush_deferred(h, (hipStream_t)stream);
  return step_action_now(h, action, action_dim, n_substeps, (hipStream_t)stream);
}

int mssim_defer_step_action(mssim_handle h, const float* action, int32_t action_dim, int32_t n_substeps, void* stream) {
  if (int rc = check_action_dim(h, action_dim)) return rc;
  flush_deferred(h, (hipStream_t)stream);
  h->deferred_action = action; h->deferred_adim = action_dim; h->deferred_nsub = n_substeps; h->deferred_stream = (hipStream_t)stream;
  return 0;
}

int mssim_link_jacobian(mssim_handle h, int32_t link_index, float* out, void* stream) {
  flush_deferred(h, (hipStream_t)stream);
  if (link_index < 0 || link_index >= h->M.n_link || !out) { h->err = "link_jacobian: bad link index / output"; return 1; }
  hipStream_t st = (hipStream_t)stream;
  if (h->dirty) { launch_fk(h, st); h->dirty = false; }
  hipLaunchKernelGGL(k_link_jacobian, env_grid(h->N, 256), dim3(256), 0, st, h->M, h->S, (int)link_index, out);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int mssim_profile_enable(mssim_handle h, int32_t on) {
  HIPCHK(h, hipSetDevice(h->device));
  if (on && h->ev[0].empty()) {
    for (int k = 0; k < 2; k++) {
      h->ev[k].resize(2 * 4096);
      for (auto& e : h->ev[k]) HIPCHK(h, hipEventCreate(&e));
    }
  }
  h->profiling = on != 0;
  h->ev_used[0] = h->ev_used[1] = 0;
  return 0;
}

int mssim_profile_read(mssim_handle h, float* out_ms, int32_t* out_counts) {
  flush_deferred(h, h->deferred_stream);
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipDeviceSynchronize());
  for (int k = 0; k < 2; k++) {
    float total = 0.f;
    size_t pairs = h->ev_used[k] / 2;
    for (size_t i = 0; i < pairs; i++) {
      float ms = 0.f;
      HIPCHK(h, hipEventElapsedTime(&ms, h->ev[k][2 * i], h->ev[k][2 * i + 1]));
      total += ms;
    }
    out_ms[k] = total;
    out_counts[k] = (int32_t)pairs;
    h->ev_used[k] = 0;
  }
  return 0;
}

int mssim_set_action_map(mssim_handle h, const int32_t* column, const float* low, const float* high, const int32_t* flags) {
  flush_deferred(h, h->deferred_stream);
  HIPCHK(h, hipSetDevice(h->device));
  const int n = h->M.n_dof > 0 ? h->M.n_dof : 1;
  for (int j = 0; j < h->M.n_dof; j++)
    if ((flags[j] & 48) && (((flags[j] >> 8) & 31) >= h->M.n_dof || (flags[j] & 48) == 48)) { h->err = "set_action_map: base-frame flags need one of cos / sin and a yaw joint of the articulation"; return 1; }
  if (!h->d_act_col) {
    int rc;
    if ((rc = dalloc(h, n, &h->d_act_col)) || (rc = dalloc(h, n, &h->d_act_lo)) || (rc = dalloc(h, n, &h->d_act_hi)) || (rc = dalloc(h, n, &h->d_act_flags))) return rc;
  }
  HIPCHK(h, hipMemcpy(h->d_act_col, column, sizeof(int) * h->M.n_dof, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_act_lo, low, sizeof(float) * h->M.n_dof, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_act_hi, high, sizeof(float) * h->M.n_dof, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_act_flags, flags, sizeof(int) * h->M.n_dof, hipMemcpyHostToDevice));
  h->act_max_col = -1;
  for (int j = 0; j < h->M.n_dof; j++) h->act_max_col = column[j] > h->act_max_col ? column[j] : h->act_max_col;
  return 0;
}

int mssim_set_ee_action_map(mssim_handle h, int32_t link_index, int32_t column0, int32_t rows, float low, float high, float rot_scale, int32_t flags) {
  flush_deferred(h, h->deferred_stream);
  if (link_index >= h->M.n_link || (link_index >= 0 && rows != 3 && rows != 6)) { h->err = "set_ee_action_map: bad link index / rows"; return 1; }
  h->ee = EeMap{link_index < 0 ? -1 : (int)link_index, (int)column0, (int)rows, low, high, rot_scale, (int)flags};
  return 0;
}

int mssim_apply_action(mssim_handle h, const float* action, int32_t action_dim, void* stream) {
  flush_deferred(h, (hipStream_t)stream);
  if (int rc = check_action_dim(h, action_dim)) return rc;
  hipLaunchKernelGGL(k_apply_action, env_grid(h->N, 256), dim3(256), 0, (hipStream_t)stream, h->M, h->S, h->buf, action, action_dim,
                     h->d_act_col, h->d_act_lo, h->d_act_hi, h->d_act_flags, h->ee);
  HIPCHK(h, hipGetLastError());
  return 0;
}

// candidate finger <-> object pairs (a handful of the ~100 pairs of the scene) for the task epilogues:
// entry = pair | finger (bit 30: 0 left, 1 right) | object is shape A (bit 31); cached per (object, fingers)
static int finger_pair_list(mssim_handle h, int obj_row, int f1_row, int f2_row) {
  if (h->pick_rows[0] == obj_row && h->pick_rows[1] == f1_row && h->pick_rows[2] == f2_row && h->d_pick_pairs) return 0;
  std::vector<int32_t> lst;
  for (int p = 0; p < h->M.n_pair; p++) {
    const int ra = h->h_shape_row[h->h_pair_shape[2 * p]], rb = h->h_shape_row[h->h_pair_shape[2 * p + 1]];
    const bool a_obj = ra == obj_row, b_obj = rb == obj_row;
    if (!(a_obj || b_obj)) continue;
    const int other = a_obj ? rb : ra;
    if (other != f1_row && other != f2_row) continue;
    lst.push_back((int32_t)((unsigned)p | (other == f2_row ? 1u << 30 : 0u) | (a_obj ? 1u << 31 : 0u)));
  }
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->d_pick_pairs) { HIPCHK(h, hipMalloc((void**)&h->d_pick_pairs, sizeof(int32_t) * (h->M.n_pair > 0 ? h->M.n_pair : 1))); h->allocs.push_back(h->d_pick_pairs); }
  if (!lst.empty()) HIPCHK(h, hipMemcpy(h->d_pick_pairs, lst.data(), sizeof(int32_t) * lst.size(), hipMemcpyHostToDevice));
  h->n_pick_pairs = (int)lst.size();
  h->pick_rows[0] = obj_row; h->pick_rows[1] = f1_row; h->pick_rows[2] = f2_row;
  return 0;
}

// A deferred step_action + deferred fetch + this epilogue = one launch of the control-step kernel (Panda
// models: the task tail is compiled into k_solve16<9, TASK>). Returns false if that does not apply.
extern "C++" {
template <int TASK>
static bool control_step_with_task(mssim_handle h, DevState& S, hipStream_t st) {
  if (!(h->deferred_action && h->deferred_fetch && h->M.n_dof == 9 && h->deferred_nsub > 0 && st == h->deferred_stream && h->ee.link < 0) || h->has_tri || h->rows_per_env != 1) return false;
  // The tail runs at the kernel's one wave per SIMD: worth it while all blocks are resident at once (4 per CU) and
  // the launch is latency-bound anyway; with more blocks the separate, fully occupied copy-out + epilogue launch
  // is cheaper than a tail per block.
  if ((h->N + S16_ENVS_PER_BLOCK - 1) / S16_ENVS_PER_BLOCK > 4 * h->n_cu) return false;
  const DevState A = state_with_action(h, h->deferred_action, h->deferred_adim);
  S.act = A.act; S.act_dim = A.act_dim; S.act_col = A.act_col; S.act_lo = A.act_lo; S.act_hi = A.act_hi; S.act_flags = A.act_flags;
  S.act_qpos = A.act_qpos; S.act_target = A.act_target; S.act_target_vel = A.act_target_vel;
  S.tail_fetch = take_deferred_fetch(h);
  S.tail_buf = h->buf;
  h->deferred_action = nullptr;
  if (h->dirty) { launch_fk(h, st); h->dirty = false; }
  launch_control_step<TASK>(h, S, h->deferred_nsub, st);
  return true;
}
}  // extern "C++"

int mssim_task_peg_outputs(mssim_handle h, const mssim_peg_task* task, float* obs, float* reward, uint8_t* flags, float* head_at_hole, void* stream) {
  const int R = h->M.n_link + h->M.n_free + h->M.n_kin;
  const int rows[5] = {task->tcp_row, task->peg_row, task->box_row, task->finger1_row, task->finger2_row};
  for (int r : rows)
    if (r < 0 || r >= R) { h->err = "task_peg_outputs: body row out of range"; return 1; }
  if (!h->buf.rigid_body_data || !h->buf.art_qpos || !h->buf.art_qvel) { h->err = "buffers not bound"; return 2; }
  if (!task->peg_half_sizes || !task->box_hole_offsets || !task->box_hole_radii || !head_at_hole) { h->err = "task_peg_outputs: missing per-env geometry / output"; return 3; }
  { int rc = finger_pair_list(h, task->peg_row, task->finger1_row, task->finger2_row); if (rc) return rc; }
  {
    DevState S = h->S;
    S.tail_task.peg = *task; S.tail_pairs = h->d_pick_pairs; S.tail_npairs = h->n_pick_pairs;
    S.tail_obs = obs; S.tail_reward = reward; S.tail_flags = flags; S.tail_head = head_at_hole;
    if (control_step_with_task<3>(h, S, (hipStream_t)stream)) { HIPCHK(h, hipGetLastError()); return 0; }
  }
  if (h->deferred_action) { const unsigned w = take_deferred_fetch(h); flush_deferred(h, (hipStream_t)stream); h->deferred_fetch = w; }
  if (const unsigned what = take_deferred_fetch(h))
    hipLaunchKernelGGL(k_task_peg<true>, env_grid(h->N, 64), dim3(256), 0, (hipStream_t)stream, h->M, h->S, h->buf, what, *task, h->d_pick_pairs, h->n_pick_pairs, obs, reward, flags, head_at_hole);
  else
    hipLaunchKernelGGL(k_task_peg<false>, env_grid(h->N, 64), dim3(64), 0, (hipStream_t)stream, h->M, h->S, h->buf, 0u, *task, h->d_pick_pairs, h->n_pick_pairs, obs, reward, flags, head_at_hole);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int mssim_task_pick_outputs(mssim_handle h, const mssim_pick_task* task, float* obs, float* reward, uint8_t* flags, void* stream) {
  const int R = h->M.n_link + h->M.n_free + h->M.n_kin;
  const int rows[5] = {task->tcp_row, task->obj_row, task->goal_row, task->finger1_row, task->finger2_row};
  for (int r : rows)
    if (r < 0 || r >= R) { h->err = "task_pick_outputs: body row out of range"; return 1; }
  if (!h->buf.rigid_body_data || !h->buf.art_qpos || !h->buf.art_qvel) { h->err = "buffers not bound"; return 2; }
  { int rc = finger_pair_list(h, task->obj_row, task->finger1_row, task->finger2_row); if (rc) return rc; }
  {
    DevState S = h->S;
    S.tail_task.pick = *task; S.tail_pairs = h->d_pick_pairs; S.tail_npairs = h->n_pick_pairs;
    S.tail_obs = obs; S.tail_reward = reward; S.tail_flags = flags; S.tail_head = nullptr;
    if (control_step_with_task<1>(h, S, (hipStream_t)stream)) { HIPCHK(h, hipGetLastError()); return 0; }
  }
  if (h->deferred_action) { const unsigned w = take_deferred_fetch(h); flush_deferred(h, (hipStream_t)stream); h->deferred_fetch = w; }
  if (const unsigned what = take_deferred_fetch(h))
    hipLaunchKernelGGL(k_task_pick<true>, env_grid(h->N, 64), dim3(256), 0, (hipStream_t)stream, h->M, h->S, h->buf, what, *task, h->d_pick_pairs, h->n_pick_pairs, obs, reward, flags);
  else
    hipLaunchKernelGGL(k_task_pick<false>, env_grid(h->N, 64), dim3(64), 0, (hipStream_t)stream, h->M, h->S, h->buf, 0u, *task, h->d_pick_pairs, h->n_pick_pairs, obs, reward, flags);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int mssim_task_push_outputs(mssim_handle h, const mssim_push_task* task, float* obs, float* reward, uint8_t* flags, void* stream) {
  const int R = h->M.n_link + h->M.n_free + h->M.n_kin;
  const int rows[3] = {task->tcp_row, task->obj_row, task->goal_row};
  for (int r : rows)
    if (r < 0 || r >= R) { h->err = "task_push_outputs: body row out of range"; return 1; }
  if (!h->buf.rigid_body_data || !h->buf.art_qpos || !h->buf.art_qvel) { h->err = "buffers not bound"; return 2; }
  {
    DevState S = h->S;
    S.tail_task.push = *task; S.tail_pairs = nullptr; S.tail_npairs = 0;
    S.tail_obs = obs; S.tail_reward = reward; S.tail_flags = flags; S.tail_head = nullptr;
    if (control_step_with_task<2>(h, S, (hipStream_t)stream)) { HIPCHK(h, hipGetLastError()); return 0; }
  }
  if (h->deferred_action) { const unsigned w = take_deferred_fetch(h); flush_deferred(h, (hipStream_t)stream); h->deferred_fetch = w; }
  if (const unsigned what = take_deferred_fetch(h))
    hipLaunchKernelGGL(k_task_push<true>, env_grid(h->N, 64), dim3(256), 0, (hipStream_t)stream, h->M, h->S, h->buf, what, *task, obs, reward, flags);
  else
    hipLaunchKernelGGL(k_task_push<false>, env_grid(h->N, 256), dim3(256), 0, (hipStream_t)stream, h->M, h->S, h->buf, 0u, *task, obs, reward, flags);
  HIPCHK(h, hipGetLastError());
  return 0;
}

static int make_query(mssim_handle h, const int32_t* data, int count_ints, int nq, int kind, int32_t* qid) {
  HIPCHK(h, hipSetDevice(h->device));
  int* d = nullptr;
  HIPCHK(h, hipMalloc((void**)&d, sizeof(int) * (count_ints > 0 ? count_ints : 1)));
  h->allocs.push_back(d);
  if (count_ints > 0) HIPCHK(h, hipMemcpy(d, data, sizeof(int) * count_ints, hipMemcpyHostToDevice));
  h->queries.push_back(d);
  h->query_n.push_back(nq);
  h->query_kind.push_back(kind);
  *qid = (int)h->queries.size() - 1;
  return 0;
}

int mssim_create_pair_query(mssim_handle h, const int32_t* body_pairs, int32_t n_pairs, int32_t* qid) {
  return make_query(h, body_pairs, 2 * n_pairs, n_pairs, 0, qid);
}
int mssim_create_body_query(mssim_handle h, const int32_t* rows, int32_t n, int32_t* qid) { return make_query(h, rows, n, n, 1, qid); }

static int run_query(mssim_handle h, int32_t qid, int kind, float* out, void* stream) {
  flush_deferred(h, (hipStream_t)stream);
  if (qid < 0 || qid >= (int)h->queries.size() || h->query_kind[qid] != kind) { h->err = "bad query id"; return 1; }
  int nq = h->query_n[qid];
  if (nq == 0) return 0;
  hipLaunchKernelGGL(k_query, dim3((h->N + 255) / 256, nq), dim3(256), 0, (hipStream_t)stream, h->M, h->S, h->queries[qid], nq, kind, out);
  HIPCHK(h, hipGetLastError());
  return 0;
}
int mssim_query_pair_impulses(mssim_handle h, int32_t qid, float* out, void* stream) { return run_query(h, qid, 0, out, stream); }
int mssim_query_body_impulses(mssim_handle h, int32_t qid, float* out, void* stream) { return run_query(h, qid, 1, out, stream); }

int mssim_read_internal(mssim_handle h, const char* name, float* out, int32_t max_items, void* stream) {
  flush_deferred(h, (hipStream_t)stream);
  std::string s(name);
  const size_t N = (size_t)h->N;
  const float* src = nullptr;
  int items = 0;
  hipStream_t st = (hipStream_t)stream;
  if (s == "q") { src = h->S.q; items = h->M.n_dof; }
  else if (s == "qd") { src = h->S.qd; items = h->M.n_dof; }
  else if (s == "free") { src = h->S.free_s; items = 13 * h->M.n_free; }
  else if (s == "kin") { src = h->S.kin; items = 7 * h->M.n_kin; }
  else if (s == "free_wake") { src = h->S.free_wake; items = h->M.n_free; }
  else if (s == "root") { src = h->S.root; items = 7; }
  else if (s == "bodypose") { src = h->S.bodypose; items = 7 * h->M.n_dof; }
  else if (s == "pair_impulse") { src = h->S.pair_imp; items = 3 * h->M.n_pair; }
  else if (s == "contact_count" || s == "overflow") {
    const int* isrc = s == "overflow" ? h->S.overflow : h->S.pair_cnt;
    items = s == "overflow" ? 1 : h->M.n_pair;
    int take = items < max_items ? items : max_items;
    size_t cnt = (size_t)take * N;
    if (cnt > 0) hipLaunchKernelGGL(k_i2f, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, isrc, out, cnt);
    HIPCHK(h, hipGetLastError());
    return items;
  } else {
    h->err = "unknown internal array: " + s;
    return -1;
  }
  int take = items < max_items ? items : max_items;
  if (take > 0) HIPCHK(h, hipMemcpyAsync(out, src, sizeof(float) * (size_t)take * N, hipMemcpyDeviceToDevice, st));
  return items;
}

int mssim_overflow_count(mssim_handle h, void* stream) {
  flush_deferred(h, (hipStream_t)stream);
  std::vector<int> host(h->N);
  hipStream_t st = (hipStream_t)stream;
  if (hipStreamSynchronize(st) != hipSuccess) return -1;
  if (hipMemcpy(host.data(), h->S.overflow, sizeof(int) * h->N, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  (void)hipMemset(h->S.overflow, 0, sizeof(int) * h->N);
  int c = 0;
  for (int v : host) c += v != 0;
  return c;
}

}  // extern "C"

#ifdef MSSIM_PHASE_CLOCKS
// debug builds only (not part of include/mssim.h): cycles per phase of k_solve16 summed over blocks
extern "C" int mssim_debug_phase_clocks(unsigned long long* out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_phase_clk), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset < 0) return 0;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_clk), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" int mssim_debug_mpr_hist(unsigned* out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_mpr_hist), 32 * sizeof(unsigned)) == hipSuccess ? 0 : -1;
}
extern "C" int mssim_debug_mpr_clocks(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_mpr_clk), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_mpr_clk), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" int mssim_debug_phase_blocks(unsigned* out, int nblocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_blk), (size_t)nblocks * 32 * sizeof(unsigned)) == hipSuccess ? 0 : -1;
}
#elif defined(MSSIM_BLOCK_TIMES)
extern "C" int mssim_debug_phase_blocks(unsigned* out, int nblocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_blk), (size_t)nblocks * 32 * sizeof(unsigned)) == hipSuccess ? 0 : -1;
}
#endif
