/*
 * mssim.h -- C ABI of the MI355X-native batched rigid-body simulation core.
 *
 * This header is the drop-in boundary of the hot path: it is what a
 * ManiSkill maintainer would bind behind `ManiSkillScene.px` in place of
 * `sapien.physx.PhysxGpuSystem`.  Every entry point names the reference call site it replaces
 * (paths relative to the reference checkout, mani_skill/...).
 *
 * Plain C: pointers and sizes only, no torch / C++ types.  All device pointers are raw HIP device
 * pointers (e.g. `torch.Tensor.data_ptr()`); `stream` is a `hipStream_t` passed as `void*`
 * (`torch.cuda.current_stream().cuda_stream`), NULL = the HIP null stream.
 *
 * Two libraries export this same ABI shape:
 *   - libmssim.so          symbols `mssim_*`      HIP / gfx950 kernels (the product).
 *   - oracle/_build/...    symbols `mssim_ref_*`  plain C++ CPU restatement (test oracle only;
 *                                                 pointers are host pointers, stream ignored).
 *
 * Data conventions
 *   - quaternions are (w, x, y, z); poses are 7 floats p(3), q(4); Z is up.
 *   - user-visible ("cuda_*") buffers follow the reference contract (structs/base.py:103-114,
 *     structs/articulation.py:567-659):
 *        rigid_body_data  [R*N][13]   pos3 quat4 linvel3 angvel3   (row = body_row*N + env)
 *        articulation_*   [N][n_dof]
 *     Row order inside `cuda_rigid_body_data` is unspecified by SAPIEN
 *     (docs/source/user_guide/concepts/gpu_simulation.md); this core uses body-major rows so a
 *     body's rows for all envs are one contiguous slice.
 *     Deviation (documented): velocity columns are lin 7:10 / ang 10:13 for ALL
 *     rows, including articulation links.
 *   - internal simulation state is struct-of-arrays with the env index fastest (coalesced);
 *     apply/fetch move data between the two, exactly as px.gpu_apply_* / px.gpu_fetch_* do.
 */
#ifndef MSSIM_H
#define MSSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSSIM_ABI_VERSION 6
/* Free bodies: the angular velocity a substep starts from, and the one its pose is integrated with, are clamped
 * to this magnitude (rad/s) -- PhysX's default PxRigidDynamic maxAngularVelocity. Without it a thin body knocked
 * into a fast spin (a peg squeezed out of the gripper) feeds the explicitly integrated gyroscopic term until the
 * velocity overflows. */
#define MSSIM_MAX_ANGULAR_VELOCITY 100.0f
/* Articulation joints: the joint velocity a substep ends with (what is stored and carried) and the one positions
 * are integrated with are clamped to this magnitude (rad/s or m/s) -- the articulation counterpart of the limit
 * above (PhysX: PxArticulationJointReducedCoordinate::setMaxJointVelocity). Without it drive targets far from the
 * current pose (a pseudo-inverse IK step next to a singular arm configuration asks for tens of radians) accelerate
 * the arm until the explicitly integrated velocity-product terms diverge. */
#define MSSIM_MAX_JOINT_VELOCITY 100.0f
/* Persistent contact manifolds (types.py:44 enable_pcm) for the generic convex pairs (everything that is not plane-X or
 * box-box, i.e. whatever the one-point MPR query serves): every env keeps MSSIM_PCM_SLOTS manifolds of up to 4 points, each
 * point stored in the frames of both shapes with the gap it was generated with. Per substep a pair's points are REFRESHED
 * (moved with their shapes; gap += relative displacement along the normal; dropped when they drift sideways by more than
 * MSSIM_PCM_DRIFT or open beyond the contact offset) and the full query runs only when the manifold is empty or the
 * relative pose of the two shapes has changed by more than MSSIM_PCM_MOVE / MSSIM_PCM_ROT_TRACE since the last query; its
 * point is merged in (replacing a point within MSSIM_PCM_MERGE, else appended; 5 -> 4 by the patch selection rule). A
 * young manifold with fewer than 3 points whose pair is NOT moving gets up to MSSIM_PCM_GROW GROWTH queries, one per
 * substep: shape A tilted by MSSIM_PCM_TILT about the manifold, so that the query finds a corner it does not have yet
 * (about a tangent through the single point, about the edge of the first two points, alternating sides) -- a body that
 * comes to rest on a face has its 3-4 points after as many substeps; a pair in relative motion costs one query per
 * substep, as without the cache. A pair that is in range but not in contact keeps an empty slot, so it is not queried
 * again until it moves. */
/* The portal refinement of a generic-convex query (MPR) stops when the support plane along the portal normal lies within
 * this distance of the portal: the accuracy of the reported gap. (1e-4, Bullet's btMprPenetration value, saves a third of
 * the refinement iterations on tessellated round hulls and 1-3 % of the control step; it is not used because the f32 kernel
 * and the f64 oracle then stop at different portals and part by up to that much at every first contact.) */
#define MSSIM_MPR_TOLERANCE 1e-5f
#define MSSIM_PCM_SLOTS 16
#define MSSIM_PCM_DRIFT 5e-3f       /* m, sideways drift that breaks a cached point (0.25 x contact offset)          */
#define MSSIM_PCM_MOVE 1e-3f        /* m, relative translation since the last full query that forces a new one       */
#define MSSIM_PCM_ROT_TRACE 2.9996f /* trace(R_last^T R_now) below this (relative rotation > ~1.15 deg) forces one   */
#define MSSIM_PCM_TILT 0.05f        /* rad, tilt of shape A in a growth query                                                */
#define MSSIM_PCM_GROW 6            /* a new manifold with fewer than 3 points is queried again in up to this many substeps  */
#define MSSIM_PCM_MERGE 2e-3f       /* m, a new point this close (in shape A's frame) to a cached one replaces it    */
/* Warm start of the contact multipliers (PhysX caches the applied impulses with its persistent manifolds). A contact
 * point is identified by its shape pair and its slot (0..3) in that pair's manifold, before the patch reduction. The
 * multipliers (normal, tangent 1, tangent 2) a point ends a substep with are the initial guess of the point with the same
 * key in the NEXT substep (also across control steps); their impulses are applied to the velocities before the first
 * sweep. Points without a predecessor, torsional rows and joint-limit rows start from zero. With 15 + 1 Gauss-Seidel
 * sweeps a cold start does not converge on stiff loads -- the gripper's saturated 100 N squeeze leaves each pad 3.3 mm
 * inside the cube, the upper of two stacked cubes creeps 1 mm in 2 s -- the warm start does (0.02 mm, 6 um; scripts/tgs_vs_pgs.py).
 * The cache is hidden state like the sleep counters and the manifolds: mssim_wake_all clears it. */
/* The position sweeps of a substep end early once a whole sweep moves no velocity component (joint velocities, free-body
 * linear and angular velocities) by more than this (m/s or rad/s): the sweeps that would follow contract further, so
 * what is left out is below the solver's own residual by orders of magnitude. solver_position_iterations is the maximum.
 * An env without contacts and without an active joint limit takes one sweep, a warm-started resting contact two or three. */
#ifndef MSSIM_PGS_EXIT_TOLERANCE
#define MSSIM_PGS_EXIT_TOLERANCE 1e-6f
#endif
/* PxSceneDesc::wakeCounterResetValue (PhysX default 20 * 0.02 s): how long the energy of a free body has to stay below
 * sleep_threshold before it is put to sleep */
#define MSSIM_WAKE_TIME 0.4f
#define MSSIM_MAX_DOF 16        /* max articulation degrees of freedom per env            */
#define MSSIM_MAX_FREE 8        /* max free (dynamic, non-articulated) bodies per env     */
#define MSSIM_MAX_POINTS 4      /* contact points kept per shape pair (PCM-style cap)     */
#define MSSIM_MAX_HULL_VERTS 64 /* per convex hull (PhysX GPU-compatible hull limit)      */
#define MSSIM_TRI_SLACK 4e-3f    /* m: a convex shape's points this much higher above a triangle's plane than its lowest one give no contact */
#define MSSIM_TRI_TIE 1e-5f      /* m: candidates of a triangle whose gaps differ by less are equally deep (the first in order is kept) */
#define MSSIM_MAX_TRI_HITS 32   /* triangles of one mesh in range of one convex shape at a time (more: reported overflow) */
#define MSSIM_MAX_TRI_TASKS 56  /* triangles in range over all (convex shape, mesh) pairs of an env (they also share MSSIM_MAX_HITS with the
                                   surviving pairs of other kinds; a (convex shape, mesh) pair itself takes no entry there) */
/* The range in which triangles are looked for is the contact offset. When that finds more triangles than there is room
 * for (either capacity above, or the hit list), the env's search is repeated with the range halved, quartered and finally
 * zero (triangles the convex shape's oriented box touches): the speculative contacts furthest out are given up first and
 * the triangles that carry load stay. Only a search that does not fit at range zero either is cut off in index order and
 * reported (MSSIM_OVERFLOW_TRI). The manifold of a triangle that is kept does not depend on the range it was found with. */
#define MSSIM_TRI_RANGE_STEPS 4
/* Contact patches. A compound body (the Panda finger: 4 boxes, panda_v3.urdf:244-283; the box with a hole: 4 boxes,
 * envs/tasks/tabletop/peg_insertion_side.py:150-181) touching another compound body yields one <= 4-point manifold per
 * SHAPE pair -- 16 shape pairs x 4 points for one finger inside the hole. Before the solver the manifolds of one BODY
 * pair whose normals lie within a cone of each other are merged into a patch and the patch is cut to its 4 most
 * significant points (deepest, farthest from it, largest area on either side -- the rule the box-box manifold uses):
 * the per-body-pair persistent-manifold cap of PCM-style narrowphases (types.py:44 enable_pcm). A manifold joins the
 * patch of the FIRST manifold (pair order) of its body pair whose normal is within MSSIM_PATCH_COS of its own. */
#define MSSIM_PATCH_COS 0.985f     /* cos of the patch cone half angle (~10 degrees)          */
/* The four selection scans take the FIRST candidate (pair order, then point order) within a tolerance of the
 * extremum: flat contacts put many points at equal depth / distance / area, and a bare "first extremum" would be decided
 * by rounding noise (a different point set in f32 and f64, and from one substep to the next). The 4th point (largest
 * area on the other side of the edge) is only taken if its area exceeds MSSIM_PATCH_TIE_REL of the 3rd point's: a point
 * on the edge itself up to rounding adds nothing and would be there or not by the sign of a rounding error. */
#define MSSIM_PATCH_TIE_SEP 1e-5f  /* separations within 10 micrometres of the deepest count as equal */
/* The selection goes by extent, not by depth: in a patch that rests on three or more points, the points the contact offset
 * admits well above them (a compound body lying on two of its parts while a third hangs 8 mm over the ground) would take
 * the places of the ones that carry the load, and the body rocks and sinks on what is left. When at least 3 points of a
 * patch with more than 4 lie within MSSIM_PATCH_SLACK of its deepest one, only those are candidates; with fewer (a corner
 * or an edge touching down) all points compete as before. (A mesh triangle applies the same bound to its own points: MSSIM_TRI_SLACK.) */
#define MSSIM_PATCH_SLACK 4e-3f
#define MSSIM_PATCH_TIE_REL 1e-3f  /* squared distances / areas within 0.1 % of the largest count as equal */
#define MSSIM_MAX_CONTACTS 52      /* contact points per env fed to the solver, after the patch reduction */
#define MSSIM_MAX_HITS 64          /* shape pairs per env that survive the cull               */
#define MSSIM_MAX_RAW_POINTS 128   /* manifold points per env before the patch reduction      */

enum {
  MSSIM_OVERFLOW_HITS = 1,     /* more than MSSIM_MAX_HITS shape pairs survived the cull            */
  MSSIM_OVERFLOW_CONVEX = 2,   /* more generic convex (MPR) pairs than the kernel's per-env list      */
  MSSIM_OVERFLOW_RAW = 4,      /* more than MSSIM_MAX_RAW_POINTS manifold points before the reduction */
  MSSIM_OVERFLOW_CONTACTS = 8, /* more than MSSIM_MAX_CONTACTS contact points after the reduction     */
  MSSIM_OVERFLOW_TRI = 16      /* more than MSSIM_MAX_TRI_HITS triangles of a mesh near one convex shape */
};
/* joint types of the moving articulation bodies (fixed joints are folded at compile time) */
enum { MSSIM_JOINT_REVOLUTE = 0, MSSIM_JOINT_PRISMATIC = 1 };
/* drive modes, articulation_joint.py:184-201 (`set_drive_properties(..., mode)`) */
enum { MSSIM_DRIVE_FORCE = 0, MSSIM_DRIVE_ACCELERATION = 1 };
/* collision shape types, utils/building/actor_builder.py:73-155 */
enum {
  MSSIM_SHAPE_PLANE = 0,   /* half space; normal = +x of the shape frame (PhysX convention)   */
  MSSIM_SHAPE_BOX = 1,     /* param = half extents                                          */
  MSSIM_SHAPE_SPHERE = 2,  /* param[0] = radius                                             */
  MSSIM_SHAPE_CAPSULE = 3, /* param[0] = radius, param[1] = half length, axis = +x of frame  */
  MSSIM_SHAPE_CYLINDER = 4,/* param[0] = radius, param[1] = half length, axis = +x of frame  */
  MSSIM_SHAPE_CONVEX = 5,  /* hull vertices in `hull_verts[hull_offset .. +hull_count)`      */
  MSSIM_SHAPE_NONE = 6,    /* per-env types only: the env has no shape in this slot         */
  MSSIM_SHAPE_TRIMESH = 7  /* triangle mesh of a fixed or kinematic body (the reference's nonconvex collision,
                              actor_builder.py:136-150): shape_hull = (root node in tri_bvh, triangle count). Every
                              triangle is a 3-vertex hull; a convex shape near the mesh is tested against the triangles
                              in range (found through the 16-wide BVH: the triangle's box within the shape's bounding
                              sphere and within the bounds of its oriented box, the oriented box within range of the
                              triangle's plane; range = contact offset, see MSSIM_TRI_RANGE_STEPS), in index order. A triangle is a bounded piece of its plane (face normal on the side of the
                              convex shape's centre): the shape's plane-contact points (box corners, capsule ends, hull
                              vertices, ...) whose gap is within the contact offset and within MSSIM_TRI_SLACK of the
                              shape's lowest point and whose foot lies in the triangle, plus -- for a box -- the
                              triangle's corners under the box and the points where its edges enter and leave the box's
                              shadow along the normal (or come nearest to the box in between): together the outline of
                              (box footprint) x (triangle), as in face clipping; the 4 deepest (gaps within MSSIM_TRI_TIE
                              are equal: first in that order), exact gaps, no tripping over inner edges
                              (a point over the neighbouring triangle is that triangle's). Only when there is no such
                              point the generic query runs (interior point: the triangle's point nearest to the shape's
                              centre): from the side it keeps its normal, within 60 degrees of the face normal it takes
                              the face normal. The contact patches merge the manifolds of coplanar triangles. At most
                              MSSIM_MAX_TRI_HITS triangles per shape pair.                                          */
};
/* what a shape / body row is attached to */
enum {
  MSSIM_BODY_WORLD = 0, /* static, fixed in the env frame                                   */
  MSSIM_BODY_ART = 1,   /* articulation: index = moving body (== dof) or -1 for the base    */
  MSSIM_BODY_FREE = 2,  /* free dynamic rigid body                                          */
  MSSIM_BODY_KIN = 3    /* kinematic body (pose set by the user, infinite mass)             */
};

/* Compiled, env-shared model (constant tables; uploaded once at create). All pointers are HOST
 * pointers, only read during mssim_create.  Built by maniskill_amd/model/compile.py from the
 * URDF/SRDF + actor builders (reference build path: utils/building/actor_builder.py:57-260,
 * articulation_builder.py:65-212, urdf_loader.py:28-47). */
typedef struct mssim_model_desc {
  int32_t abi_version;

  /* ---- articulation (0 or 1 per env; fixed base) ---- */
  int32_t n_dof;                 /* moving bodies == active joints                               */
  const int32_t* dof_parent;     /* [n_dof] parent moving body, -1 = base                        */
  const int32_t* dof_type;       /* [n_dof] MSSIM_JOINT_*                                        */
  const float* dof_frame;        /* [n_dof][7] joint frame in the parent body frame (p, q)       */
  const float* dof_axis;         /* [n_dof][3] unit axis in the joint frame                      */
  const float* dof_limit;        /* [n_dof][2] lower, upper                                      */
  const float* dof_drive;        /* [n_dof][4] stiffness, damping, force_limit, mode             */
  const float* dof_armature;     /* [n_dof] extra joint-space inertia                            */
  const float* body_inertial;    /* [n_dof][10] mass, com(3), inertia about com xx yy zz xy xz yz
                                    (fixed-joint children already folded in), body frame        */
  const int32_t* body_gravity;   /* [n_dof] 1 = gravity acts on this body (base_agent.py:272-282
                                    disables it on every robot link)                            */
  int32_t n_tendon;              /* mimic joints, articulation_builder.py:160-199                */
  const int32_t* tendon_dof;     /* [n_tendon][2] dof a, dof b                                   */
  const float* tendon_param;     /* [n_tendon][5] coef_a, coef_b, rest_length, stiffness, damping */

  /* ---- rigid body rows of `rigid_body_data` ---- */
  int32_t n_link;                /* articulation links (all URDF links incl. fixed ones), rows 0..n_link-1 */
  const int32_t* link_body;      /* [n_link] moving body the link is rigidly attached to, -1 = base */
  const float* link_frame;       /* [n_link][7] link frame in that body's frame                   */
  int32_t n_free;                /* free bodies, rows n_link .. n_link+n_free-1                   */
  const float* free_inertial;    /* [n_free][10] as body_inertial                                 */
  const float* free_damping;     /* [n_free][2] linear, angular damping                           */
  const int32_t* free_gravity;   /* [n_free]                                                      */
  int32_t n_kin;                 /* kinematic bodies, rows n_link+n_free ..                       */

  /* ---- collision shapes ---- */
  int32_t n_shape;
  const int32_t* shape_type;     /* [n_shape] MSSIM_SHAPE_*                                       */
  const int32_t* shape_body_kind;/* [n_shape] MSSIM_BODY_*                                        */
  const int32_t* shape_body_index;/* [n_shape] index within its kind                              */
  const int32_t* shape_row;      /* [n_shape] rigid_body_data row of the owning body, -1 = none   */
  const float* shape_frame;      /* [n_shape][7] shape frame in the body frame                    */
  const float* shape_param;      /* [n_shape][4] see MSSIM_SHAPE_*                                */
  const float* shape_material;   /* [n_shape][4] static friction, dynamic friction, restitution, torsional patch radius
                                    (max of patch_radius / min_patch_radius, agents/robots/panda/panda.py:24-31): a contact
                                    patch (MSSIM_PATCH_COS) between shapes with radius r > 0 gets a torsional friction row
                                    about its normal, bounded by mu * r * (sum of the normal multipliers of the patch) */
  const int32_t* shape_hull;     /* [n_shape][2] hull_offset, hull_count (CONVEX only)            */
  const float* shape_bound;      /* [n_shape][4] bounding sphere: centre in shape frame, radius   */
  int32_t n_hull_verts;
  const float* hull_verts;       /* [n_hull_verts][3] in the shape frame                          */

  /* ---- candidate collision pairs (static broadphase result: group bits, SRDF
   *      disable_collisions, same-body and static-static pairs already removed) ---- */
  int32_t n_pair;
  const int32_t* pair_shape;     /* [n_pair][2] shape a, shape b                                  */

  /* ---- scene configuration, utils/structs/types.py:36-67 ---- */
  float gravity[3];
  float timestep;                /* 1 / sim_freq, envs/sapien_env.py:1111                         */
  float contact_offset;          /* 0.02                                                          */
  float rest_offset;             /* 0                                                             */
  float bounce_threshold;        /* 2.0                                                           */
  int32_t position_iterations;   /* 15                                                            */
  int32_t velocity_iterations;   /* 1                                                             */
  float erp;                     /* fraction of penetration removed per substep by the bias       */
  float max_depenetration_velocity;
  float sleep_threshold;         /* 0.005 (types.py:39): a free body whose mass-normalised kinetic energy 0.5 (v^2 + w.Iw/m)
                                    stays below it for MSSIM_WAKE_TIME seconds, without a moving partner in range, goes to
                                    sleep: zero velocity, out of the solver, its contacts with fixed bodies dropped. It wakes
                                    when an awake moving body comes into range of one of its shape pairs (cull level), when
                                    the user writes a different pose / velocity or a force for it, or a different pose for ANY
                                    kinematic body of its env (apply: a platform moved into a sleeping body, or away from under
                                    it, must not leave it asleep -- PhysX wakes what a kinematic target touches). 0 = never sleeps */

  /* ---- per-env geometry overrides (ABI v2): same shape types in every env, different sizes /
   *      local poses / inertias -- the reference builds such actors per sub-scene and merges them
   *      (Actor.merge, utils/structs/actor.py:99-126; PegInsertionSide peg_insertion_side.py:114-181).
   *      Arrays are [items][num_envs], env fastest; num_envs must equal mssim_create's. ---- */
  int32_t num_envs;              /* N the env arrays were built for (0 = no env arrays)           */
  int32_t n_env_shape;
  const int32_t* shape_env_slot; /* [n_shape] slot into the env_shape_* arrays, -1 = shared       */
  const float* env_shape_frame;  /* [n_env_shape*7][N]                                            */
  const float* env_shape_param;  /* [n_env_shape*4][N]: rows 0..2 = the parameters of the env's shape (see MSSIM_SHAPE_*; a
                                    convex shape: first vertex in hull_verts, vertex count <= MSSIM_MAX_HULL_VERTS, -, as
                                    float-valued integers: a different hull per env), row 3 = the env's shape type + 1 as a
                                    float-valued integer, 0 = shape_type[s]. Envs may carry different shape types in a slot
                                    (never a plane; a triangle mesh only in a slot whose own type is a triangle mesh: rows 0..2 =
                                    first triangle in tri_soup, triangle count, root node in tri_bvh of THIS env's mesh --
                                    scenery that differs from sub-scene to sub-scene, or is absent: MSSIM_SHAPE_NONE) or none
                                    at all -- the reference's per-env object sets:
                                    one object model per sub-scene, merged into one actor (utils/structs/actor.py:99-126). A
                                    free body whose mass (env_free_inertial row 0) is 0 in an env does not exist there: it
                                    is never awake, keeps the pose it is given and takes part in nothing              */
  const float* env_shape_bound;  /* [n_env_shape*4][N] bounding-sphere centre in the BODY frame, radius */
  int32_t n_env_free;
  const int32_t* free_env_slot;  /* [n_free] slot into env_free_inertial, -1 = shared             */
  const float* env_free_inertial;/* [n_env_free*10][N]                                            */
  /* ---- triangle meshes (ABI v5) ---- */
  int32_t n_tri;
  const float* tri_soup;         /* [n_tri][12] centroid (3), then the three corners relative to it (shape frame)  */
  int32_t n_tri_node;
  const float* tri_bvh;          /* [n_tri_node][112] 16-wide BVH: child c's box in words 6c..6c+5 (min xyz, max xyz; min > max:
                                    no child), its reference in word 96 + c as an int32 bit pattern (>= 0: node, < 0: ~triangle) */
} mssim_model_desc;

/* User-visible buffers (device pointers owned by the caller, e.g. torch tensors); mirrors
 * `px.cuda_*` (structs/base.py:112-114, structs/articulation.py:567-659, actor.py:305-316). */
typedef struct mssim_buffers {
  float* rigid_body_data;   /* [R*N][13], R = n_link + n_free + n_kin                            */
  float* rigid_body_force;  /* [R*N][4] force xyz (+pad) applied for the next step only, may be NULL */
  float* art_qpos;          /* [N][n_dof]                                                         */
  float* art_qvel;
  float* art_qacc;
  float* art_qf;
  float* art_target_qpos;
  float* art_target_qvel;
} mssim_buffers;

/* apply / fetch selector bits, one per px.gpu_apply_* / px.gpu_fetch_* call
 * (envs/scene.py:941-977) */
enum {
  MSSIM_RIGID_DATA = 1u << 0,   /* gpu_apply_rigid_dynamic_data / gpu_fetch_rigid_dynamic_data   */
  MSSIM_ART_QPOS = 1u << 1,
  MSSIM_ART_QVEL = 1u << 2,
  MSSIM_ART_QF = 1u << 3,
  MSSIM_ART_ROOT_POSE = 1u << 4,
  MSSIM_ART_ROOT_VEL = 1u << 5,
  MSSIM_ART_TARGET_POS = 1u << 6,
  MSSIM_ART_TARGET_VEL = 1u << 7,
  MSSIM_RIGID_FORCE = 1u << 8,  /* gpu_apply_rigid_dynamic_force                                  */
  MSSIM_LINK_POSE = 1u << 9,    /* gpu_fetch_articulation_link_pose                               */
  MSSIM_LINK_VEL = 1u << 10,    /* gpu_fetch_articulation_link_velocity                           */
  MSSIM_ART_QACC = 1u << 11,
  MSSIM_ALL = 0xFFFu
};

typedef struct mssim_sim* mssim_handle;

#ifndef MSSIM_PREFIX
#define MSSIM_PREFIX mssim_
#endif
#define MSSIM_CAT2(a, b) a##b
#define MSSIM_CAT(a, b) MSSIM_CAT2(a, b)
#define MSSIM_FN(name) MSSIM_CAT(MSSIM_PREFIX, name)

/* replaces sapien.physx.PhysxGpuSystem(device) + px.gpu_init()   (sapien_env.py:1077, scene.py:905).
 * device >= 0: HIP device ordinal.  Returns 0 on success. */
int MSSIM_FN(create)(const mssim_model_desc* model, int32_t num_envs, int32_t device, mssim_handle* out);
void MSSIM_FN(destroy)(mssim_handle h);
/* binds the px.cuda_* tensors (structs/base.py:112-114) */
int MSSIM_FN(bind_buffers)(mssim_handle h, const mssim_buffers* buffers);
/* px.timestep setter/getter (sapien_env.py:1111) */
int MSSIM_FN(set_timestep)(mssim_handle h, float dt);
float MSSIM_FN(get_timestep)(mssim_handle h);
/* px.gpu_apply_*  (scene.py:941-957): user buffers -> simulation state, whole buffers */
int MSSIM_FN(apply)(mssim_handle h, uint32_t what, void* stream);
/* px.gpu_fetch_*  (scene.py:959-977): simulation state -> user buffers */
int MSSIM_FN(fetch)(mssim_handle h, uint32_t what, void* stream);
/* The same copy-out, owed until the next call on the handle: a task epilogue (task_*_outputs) performs it
 * inside its own launch, any other call that touches the state or the buffers performs it first. For the
 * caller this is fetch() with one launch less per control step -- provided it does not read the bound buffers
 * itself before that next call (BaseEnv.step: _gpu_fetch_all is directly followed by evaluate / get_obs,
 * envs/sapien_env.py:1023-1049). */
int MSSIM_FN(defer_fetch)(mssim_handle h, uint32_t what);
/* px.step() x n_substeps (scene.py:374-375; loop at sapien_env.py:1016-1021).  No host sync. */
int MSSIM_FN(step)(mssim_handle h, int32_t n_substeps, void* stream);
/* Wake every sleeping free body (PxRigidDynamic::wakeUp for all of them; see sleep_threshold), forget the persistent
 * contact manifolds and the warm-start multipliers: the sleep counters and the two caches are simulation state that
 * `rigid_body_data` does not carry, so a caller that restores a state and wants the run that follows to depend on that
 * state alone (BaseEnv.set_state_dict, envs/sapien_env.py:1167-1179) calls this after apply. */
int MSSIM_FN(wake_all)(mssim_handle h, void* stream);
/* The same for the listed envs only (ABI v6): env_idx = device array of n_idx env indices (int64, what torch index tensors
 * are; host array for the oracle). BaseEnv.reset calls it for the envs it resets (envs/sapien_env.py:776-879: a partial
 * reset re-initialises those envs and nothing else), so that an episode never starts from the previous episode's sleep
 * counters, manifolds and multipliers -- with PhysX the reset writes go through setGlobalPose / setLinearVelocity, which
 * wake the body and drop its cached contacts. No host synchronisation. */
int MSSIM_FN(wake_envs)(mssim_handle h, const int64_t* env_idx, int32_t n_idx, void* stream);
/* px.gpu_update_articulation_kinematics() (sapien_env.py:861-865) */
int MSSIM_FN(update_kinematics)(mssim_handle h, void* stream);
/* px.gpu_create_contact_pair_impulse_query (scene.py:769-772): body_pairs = [n_pairs][2]
 * rigid_body_data body rows (row / N), -1 = static world. */
int MSSIM_FN(create_pair_query)(mssim_handle h, const int32_t* body_pairs, int32_t n_pairs, int32_t* query_id);
/* px.gpu_query_contact_pair_impulses (scene.py:773-776): out = [n_pairs*N][3] device floats,
 * row = pair*N + env, impulse on body A from body B during the LAST substep. */
int MSSIM_FN(query_pair_impulses)(mssim_handle h, int32_t query_id, float* out, void* stream);
/* px.gpu_create_contact_body_impulse_query / gpu_query_contact_body_impulses
 * (structs/base.py:116-136): net contact impulse on each listed body row. */
int MSSIM_FN(create_body_query)(mssim_handle h, const int32_t* body_rows, int32_t n_bodies, int32_t* query_id);
int MSSIM_FN(query_body_impulses)(mssim_handle h, int32_t query_id, float* out, void* stream);
/* Update drive gains after create (ArticulationJoint.set_drive_properties,
 * articulation_joint.py:184-201; allowed post-init in the reference): drive = [n_dof][4] host. */
int MSSIM_FN(set_drive_properties)(mssim_handle h, const float* dof_drive);
/* Debug/parity access to internal SoA state: copies the named array ([items][N] floats) into
 * `out` (device pointer for the HIP build, host for the oracle).  Names: "q", "qd", "free"
 * (13 per body), "kin" (7 per body), "root" (7), "bodypose" (7 per moving body),
 * "contact_count" (per pair, as float), "overflow" (1).  Returns number of items or <0. */
int MSSIM_FN(read_internal)(mssim_handle h, const char* name, float* out, int32_t max_items, void* stream);
/* capacity overflow must be a reported condition (SURVEY 8b error conventions): number of envs in which a capacity
 * was exceeded since the last call (synchronises the stream). read_internal("overflow") gives the reason per env as a
 * bit set of MSSIM_OVERFLOW_* (cleared by overflow_count). */
int MSSIM_FN(overflow_count)(mssim_handle h, void* stream);
/* ---- fused callers of the step: what the reference does with ~10 + ~100 tiny
 * torch kernels per control step around px.step(), as one launch each. Results are identical to
 * the torch restatements in maniskill_amd/agents/controllers and envs/tasks (parity-tested). ----
 *
 * Affine action -> joint drive targets, for every PD joint-position style controller
 * (agents/controllers/pd_joint_pos.py:73-90, base_controller.py:120-133, utils/gym_utils.py:102-105):
 *   a = action[env][column[j]];  if (flags[j] & 2) a = low[j] + 0.5*(clip(a,-1,1)+1)*(high[j]-low[j]);
 *   target[j] = (flags[j] & 1 ? qpos[j] : 0) + a        (column[j] < 0: joint left untouched)
 *   flags[j] & 8: a is the joint's velocity drive target instead (pd_joint_vel.py:31-33); flags[j] & 4: the joint
 *   is driven by the end-effector block (set_ee_action_map)
 *   flags[j] & 16 / & 32: a is multiplied by cos / sin of qpos[(flags[j] >> 8) & 31] -- the forward velocity of a planar
 *   base whose x, y and yaw are joints of the articulation, given in the base's own frame
 *   (agents/controllers/pd_base_vel.py:43-70: x joint: 16, y joint: 32, both on the forward column, yaw index in bits 8..12)
 * writes both the user-visible target_qpos buffer and the simulation state. All arrays [n_dof], host.
 * apply_action / step_action / defer_step_action fail (non-zero, last_error) when `action_dim` does not cover every
 * mapped column (the reference asserts action.shape == (num_envs, action_dim), base_controller.py:120-133). */
int MSSIM_FN(set_action_map)(mssim_handle h, const int32_t* column, const float* low, const float* high, const int32_t* flags);
/* End-effector block of the action map, for `pd_ee_delta_pos` (rows = 3) and `pd_ee_delta_pose` (rows = 6)
 * (agents/controllers/pd_ee_pose.py:79-96, 197-210 with Kinematics.compute_ik's delta solver,
 * controllers/utils/kinematics.py:156-171): action columns column0..+rows-1 are a translation (and a rotation
 * vector) of link `link_index` in the root frame. flags & 2: translation clipped to [-1,1] and mapped to
 * [low, high]; rotation clipped by its norm to 1 and multiplied by rot_scale. The joints flagged 4 in
 * set_action_map get  target = qpos + J^T (J J^T + 1e-9 I)^-1 a,  J = Jacobian of the link over the joints on
 * its path, from the state of the last FK. link_index < 0 removes the block. */
int MSSIM_FN(set_ee_action_map)(mssim_handle h, int32_t link_index, int32_t column0, int32_t rows, float low, float high, float rot_scale, int32_t flags);
int MSSIM_FN(apply_action)(mssim_handle h, const float* action /* device [N][action_dim] */, int32_t action_dim, void* stream);
/* apply_action followed by step(n_substeps) -- BaseEnv._step_action's set_action + substep loop
 * (envs/sapien_env.py:1009-1021) -- as ONE launch when the control-step kernel is in use (the action map runs
 * at its head); two launches otherwise. Same results as the two calls. */
int MSSIM_FN(step_action)(mssim_handle h, const float* action /* device [N][action_dim] */, int32_t action_dim, int32_t n_substeps, void* stream);
/* The same, owed until the next call on the handle (the action array must stay valid until then). If that call
 * is a task epilogue (task_*_outputs) on the same stream with a copy-out owed as well (defer_fetch), the WHOLE
 * control step -- action map, substeps, copy-out, evaluate / obs / reward -- is one launch of the control-step
 * kernel; any other call first performs what is owed, in order. Results are those of the separate calls. */
int MSSIM_FN(defer_step_action)(mssim_handle h, const float* action, int32_t action_dim, int32_t n_substeps, void* stream);

/* PickCube-style evaluate + state observation + dense reward in one launch
 * (envs/tasks/tabletop/pick_cube.py:99-158, agents/robots/panda/panda.py:236-298). Reads the
 * user-visible buffers (after fetch) and the last substep's contact impulses. */
typedef struct mssim_pick_task {
  int32_t tcp_row, obj_row, goal_row, finger1_row, finger2_row; /* rigid_body_data body rows */
  int32_t n_static_dofs;      /* qvel[:n] used by is_static and the static reward (7 for the Panda) */
  float goal_thresh;          /* 0.025 */
  float static_thresh;        /* 0.2   */
  float min_force;            /* 0.5 N */
  float max_angle_deg;        /* 85    */
  float reward_scale;         /* 1 (dense) or 1/5 (normalized_dense) */
  int32_t* elapsed_steps;     /* optional device [N]: incremented in place (BaseEnv.step, sapien_env.py:951) ... */
  int32_t* elapsed_out;       /* ... and the new value copied here (info["elapsed_steps"], sapien_env.py:739)      */
  uint8_t* truncated_out;     /* optional device [N]: new elapsed_steps >= time_limit (TimeLimitWrapper, utils/registration.py:160-168) */
  int32_t time_limit;
  uint8_t* terminated_out;    /* optional device [N]: `terminated` of BaseEnv.step = a COPY of success (envs/sapien_env.py:954-964: callers
                                 such as ManiSkillVectorEnv(ignore_terminations=True) overwrite it in place) */
} mssim_pick_task;
/* obs [N][2*n_dof+24] f32 (qpos, qvel, is_grasped, tcp_pose7, goal_pos3, obj_pose7, tcp_to_obj3,
 * obj_to_goal3), reward [N] f32, flags [N][4] u8 = success, is_obj_placed, is_robot_static, is_grasped */
int MSSIM_FN(task_pick_outputs)(mssim_handle h, const mssim_pick_task* task, float* obs, float* reward, uint8_t* flags, void* stream);

/* PushCube-style evaluate + state observation + dense reward in one launch
 * (envs/tasks/tabletop/push_cube.py:165-232). obs [N][2*n_dof+17] f32 (qpos, qvel, tcp_pose7, goal_pos3,
 * obj_pose7), reward [N] f32, flags [N][1] u8 = success */
typedef struct mssim_push_task {
  int32_t tcp_row, obj_row, goal_row; /* rigid_body_data body rows */
  float goal_radius;                  /* 0.1  */
  float cube_half_size;               /* 0.02 */
  float reward_scale;                 /* 1 (dense) or 1/3 (normalized_dense) */
  int32_t* elapsed_steps;             /* optional, as in mssim_pick_task */
  int32_t* elapsed_out;
  uint8_t* truncated_out;     /* optional device [N]: new elapsed_steps >= time_limit (TimeLimitWrapper, utils/registration.py:160-168) */
  int32_t time_limit;
  uint8_t* terminated_out;    /* optional device [N]: `terminated` of BaseEnv.step = a COPY of success (envs/sapien_env.py:954-964: callers
                                 such as ManiSkillVectorEnv(ignore_terminations=True) overwrite it in place) */
} mssim_push_task;
int MSSIM_FN(task_push_outputs)(mssim_handle h, const mssim_push_task* task, float* obs, float* reward, uint8_t* flags, void* stream);

/* PegInsertionSide-style evaluate + state observation + dense reward in one launch
 * (envs/tasks/tabletop/peg_insertion_side.py:247-355). Per-env geometry comes as device arrays.
 * obs [N][2*n_dof+25] f32 (qpos, qvel, tcp_pose7, peg_pose7, peg_half_size3, box_hole_pose7,
 * box_hole_radius), reward [N] f32, flags [N][1] u8 = success, head_at_hole [N][3] f32 = peg head in the
 * hole frame (info["peg_head_pos_at_hole"]) */
typedef struct mssim_peg_task {
  int32_t tcp_row, peg_row, box_row, finger1_row, finger2_row; /* rigid_body_data body rows */
  float min_force;            /* 0.5 N */
  float max_angle_deg;        /* 20    */
  float reward_scale;         /* 1 (dense) or 1/10 (normalized_dense) */
  const float* peg_half_sizes;   /* device [N][3] */
  const float* box_hole_offsets; /* device [N][3]: hole centre in the box frame */
  const float* box_hole_radii;   /* device [N]    */
  int32_t* elapsed_steps;     /* optional, as in mssim_pick_task */
  int32_t* elapsed_out;
  uint8_t* truncated_out;     /* optional device [N]: new elapsed_steps >= time_limit (TimeLimitWrapper, utils/registration.py:160-168) */
  int32_t time_limit;
  uint8_t* terminated_out;    /* optional device [N]: `terminated` of BaseEnv.step = a COPY of success (envs/sapien_env.py:954-964: callers
                                 such as ManiSkillVectorEnv(ignore_terminations=True) overwrite it in place) */
} mssim_peg_task;
int MSSIM_FN(task_peg_outputs)(mssim_handle h, const mssim_peg_task* task, float* obs, float* reward, uint8_t* flags, float* head_at_hole, void* stream);

/* Geometric Jacobian of an articulation link at the CURRENT simulation state, for the end-effector
 * controllers (agents/controllers/pd_ee_pose.py:96-121, controllers/utils/kinematics.py:156-171, where
 * the reference calls pytorch_kinematics' `chain.jacobian`): out [N][6][n_dof] f32, rows 0-2 linear
 * and 3-5 angular velocity of the link frame origin per unit joint velocity, expressed in the
 * articulation ROOT frame; columns of joints that do not move the link are zero. It falls out of the
 * step's own FK state (world joint axes / anchors), no separate kinematics pass. */
int MSSIM_FN(link_jacobian)(mssim_handle h, int32_t link_index, float* out, void* stream);

/* Measurement aid (bench.py roofline block): when enabled, every launch of the control-step kernel (k_solve16: a whole
 * control step, or the substeps of mssim_step) is bracketed by HIP events on the SAME stream it is launched on. profile_read
 * synchronises, returns the accumulated milliseconds and launch counts since the last read (index 0 = the control-step
 * kernel; index 1 is reserved and reads zero: there is no separate narrowphase kernel) and clears them. The oracle returns zeros. */
int MSSIM_FN(profile_enable)(mssim_handle h, int32_t on);
int MSSIM_FN(profile_read)(mssim_handle h, float* out_ms2, int32_t* out_counts2);
/* last error message of this handle (or of create when h == NULL) */
const char* MSSIM_FN(last_error)(mssim_handle h);
int MSSIM_FN(abi_version)(void);

#ifdef __cplusplus
}
#endif
#endif /* MSSIM_H */
