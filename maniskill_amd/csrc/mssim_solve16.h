// mssim_solve16.h -- the control-step kernel: an env on one or two 16-lane DPP rows, a whole control step per launch.
//
// Why: with one env per lane N = 4096 envs are 64 wavefronts on a chip with 1024 SIMDs and the
// kernel is instruction-issue bound at one wave per SIMD. Here every env is
// spread over 16-lane DPP rows: a lane owns one velocity component (a joint, or one linear / angular
// component of a free body), so N = 4096 is 1024 waves and the long serial chains become
//   * tree recursions  -> sums over ancestor / descendant bit sets of per-body quantities staged
//                         in LDS; FK by pointer jumping,
//   * dense 9x9 algebra-> Gauss-Jordan with one matrix row per lane, pivot row broadcast by DPP row_newbcast,
//   * Gauss-Seidel row -> one multiply, a DPP row-rotate all-reduce, a clamp, one FMA.
//
// k_solve16<NDOF, TASK, TRI, NR>: one launch runs a whole control step. A wave keeps its envs' state in registers /
//   LDS across the substeps and does the narrowphase itself between them (per-env world shape table in LDS, 16 pairs
//   per env culled per round, generic-convex / box-box / mesh-triangle pairs of the whole block taken round-robin by
//   its sixteen 16-lane groups, persistent manifolds, contact patches, contact records in LDS). State crosses HBM once
//   per control step; there is no global contact buffer. Optional head: the action map (mssim_step_action); optional
//   tail (TASK > 0): copy-out + a task's evaluate / obs / reward.
//   NR = 16-lane rows per env. NR = 1: 4 envs per wave, the articulation's joints and up to two free bodies share the
//   row (n_dof + 6 n_free <= 16: the Panda with one object, the Fetch alone). NR = 2: 2 envs per wave; row 0 holds
//   the articulation (<= 16 joints) and does the env's own narrowphase bookkeeping, row 1 holds two free bodies (6
//   lanes each). A^-1 is block diagonal over (articulation | free body | free body), so W = A^-1 J^T never crosses a
//   row; only the J.v reductions of the solver and the Delassus terms of the row build add one cross-row step
//   (v_permlane16_swap). The shared narrowphase stages see sixteen groups per block either way.
//
// Same math and the same row order as the oracle (contacts patch by patch in pair order, normal + 2 friction rows
// each, then the joint limits), so the parity tests cover every variant unchanged.
// Requires n_dof <= 16, n_free <= 2, n_kin <= 6, n_shape <= 28 (mssim_create picks NR and says what a model exceeds).
#pragma once

#define S16_LANES 16
// A block is S16_WAVES waves of 4 envs each. Every wave works on its own envs (its own slice of the LDS, wave-level
// ordering only: WSYNC) except in narrowphase stage B, where the generic-convex pairs of ALL the block's envs form one
// task list taken round-robin by all its 16-lane groups (BSYNC = block barrier around it): the launch lasts as long as
// its slowest env, and an arm folded onto itself has 15-20 such pairs per substep -- 4-5 rounds for the 4 groups of one
// wave, 1-2 rounds for the 16 groups of four.
#ifndef S16_WAVES
#define S16_WAVES 4
#endif
// ordering of LDS traffic among the lanes of ONE wave (its LDS instructions execute in program order): a compiler fence
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define BSYNC() __syncthreads()
#ifndef S16_ENVS_PER_BLOCK
#define S16_ENVS_PER_BLOCK 4
#endif
// per-env LDS layout (floats)
#define S16_COM_(nr) ((nr) == 4 ? 300 : 0)  // [S16_MAX_FREE][3] free-body centres of mass (NR = 4: behind the longer pose table)
#define S16_COM S16_COM_(NR)
#define S16_ANC 8      // [8..23] ancestor-or-self masks of the dofs
#define S16_VEC 32     // 4 x [16] scratch vectors
#define S16_PT 96      // pose table [23 + S16_MAX_FREE][7]: root | 16 links | the free bodies | 6 kinematic
#define S16_PT_LINK 1
#define S16_PT_FREE 17
#define S16_MAX_FREE_(nr) ((nr) == 4 ? 6 : 2)  // free bodies per env: two per 16-lane row that holds free bodies
#define S16_MAX_FREE S16_MAX_FREE_(NR)
#define S16_PT_KIN_(nr) (S16_PT_FREE + S16_MAX_FREE_(nr))
#define S16_PT_KIN S16_PT_KIN_(NR)
#define S16_MAX_KIN 6
#define S16_MAX_SHAPE_(nr) ((nr) == 4 ? 64 : ((nr) > 1 ? 48 : 28))  // shapes per model: a lane builds the world-table entries of two (NR > 1: all lanes of the env)
#define S16_MAX_HIT 64
#define S16_BP (S16_PT + 7 * S16_PT_LINK)  // link poses
#define S16_U_(nr) ((nr) == 4 ? 320 : 272)  // union: dynamics staging | solver rows | narrowphase scratch
#define S16_U S16_U_(NR)
#define S16_S (S16_U)          // [16][6]
#define S16_V (S16_U + 96)     // [16][6]
#define S16_T (S16_U + 192)    // [16][6]
#define S16_F (S16_U + 288)    // [16][6]
#define S16_IC (S16_U + 384)   // [16][10]
#define S16_MAT (S16_U + 544)  // [16][16]
#define S16_LIMW (S16_U)       // [16][16] W = A^-1 J^T of the joint-limit rows
#define S16_CS (S16_U + 256)   // [MAXC][16] block scalars of every contact: 1/d0 bias+ bias- mu | 1/d1 k10 1/d2 k20 | k21 lam0 lam1 lam2 | pair
#define S16_REGC 12            // first contacts of an env: this lane's J / W entries and the multipliers stay in registers
#define S16_LDSC 7             // next contacts: J | W rows in LDS; the rest stream from the per-env global scratch
// (sizes that depend on NR, the number of 16-lane rows per env: `_(nr)` forms for the host; inside the kernel the plain
// names read its template parameter NR)
#define S16_JWLEN_(nr) (96 * (nr))  // 3 x (J[16 nr] W[16 nr]) of one contact
#define S16_JW (S16_CS + 16 * MAXC)                     // [S16_LDSC][S16_JWLEN]
#define S16_REC_(nr) (S16_U_(nr) + 256 + 16 * MAXC + S16_JWLEN_(nr) * S16_LDSC)  // contact records [MAXC][S16_REC_LEN]
#define S16_REC_LEN_(nr) ((nr) == 4 ? 14 : ((nr) > 1 ? 12 : 10))  // n(3) x(3) sep pair lane-masks of row 0 (A | B << 16) mu [lane-masks of rows 1.., -]
// per env: NR = 1: 2552 floats (== 24 mod 32 banks); NR = 2: 3344 floats (== 16 mod 32: the wave's two envs on different banks)
// (NR = 4: behind the records, box-box scratch of the three rows that are not the env's own: 3 x (24 polygon words + 20 result words))
#define S16_XSCR_(nr) (S16_REC_(nr) + S16_REC_LEN_(nr) * MAXC)
#define S16_XSCR S16_XSCR_(NR)
#define S16_ENV_FLOATS_(nr) (S16_REC_(nr) + S16_REC_LEN_(nr) * MAXC + ((nr) == 2 ? 16 : ((nr) == 4 ? 136 : 0)))
#define S16_ROWLEN_(nr) (32 * (nr))
#define S16_ROWS_GLB (3 * (MAXC - S16_REGC - S16_LDSC))  // global scratch rows per env (x S16_ROWLEN floats)
#define S16_JWLEN S16_JWLEN_(NR)
#define S16_REC S16_REC_(NR)
#define S16_REC_LEN S16_REC_LEN_(NR)
#define S16_ENV_FLOATS S16_ENV_FLOATS_(NR)
#define S16_ROWLEN S16_ROWLEN_(NR)
// narrowphase scratch, overlays the union below the contact records
#define S16_SHP 20                // floats per entry: pose7 param3 centre3 radius packed mu half3 torsional-radius
#define S16_NP_SHP (S16_U)        // [S16_MAX_SHAPE][20] world shape table
#define S16_NP_B_(nr) (S16_U_(nr) + S16_SHP * S16_MAX_SHAPE_(nr))  // what follows the shape table
#define S16_NP_B S16_NP_B_(NR)
#define S16_MAX_SHAPE S16_MAX_SHAPE_(NR)
#define S16_NP_HIT (S16_NP_B)        // [64] surviving pairs: pair | sa << 16 | sb << 24
#define S16_NP_CNT (S16_NP_B + 64)   // [64] manifold sizes (raw, before the patch reduction)
#define S16_NP_OFF (S16_NP_B + 128)  // [64] first point of each manifold in the point pool
// [896]: pair table during the cull (<= 896 pairs) | [56][16] box-box clip scratch of the one-lane-per-pair path |
// afterwards the staged manifolds: normals, point pool, patch bookkeeping
#define S16_NP_SCR (S16_NP_B + 192)
#define S16_NP_HN (S16_NP_SCR)            // [64][3] manifold normals
#define S16_NP_POOL (S16_NP_SCR + 192)    // [MSSIM_MAX_RAW_POINTS][4] x y z sep, in allocation order (16-byte aligned)
#define S16_NP_BSCR (S16_NP_SCR + 704)    // [24] polygon scatter / gather words of the group (cooperative box-box)
#define S16_NP_KEEP (S16_NP_SCR + 728)    // [64] patch anchor << 4 | mask of the manifold's points that survive
#define S16_NP_KEY (S16_NP_SCR + 792)     // [64] body pair of the manifold
#define S16_NP_ALLOC (S16_NP_SCR + 856)   // [1] points handed out from the pool
#define S16_NP_BOUT (S16_NP_SCR + 860)    // [20] manifold of the group's current cooperative box-box pair
#define S16_NP_ML (S16_NP_B + 1088)  // [64 bytes] hit indices of this env's MPR (generic convex) pairs
#define S16_NP_BL (S16_NP_B + 1104)  // [64 bytes] hit indices of this env's box-box pairs
#define S16_NP_SLOT (S16_NP_B + 1120) // [64 bytes] persistent-manifold slot of each hit (255 none | slot | 0x80 newly assigned)
#define S16_NP_PL (S16_NP_B + 1136)   // [64 bytes] hit indices of this env's (plane, hull) pairs
#define S16_PCM_LEN 48            // floats per cache slot: pair npts stamp flags | relp(3) - | relR(9) n_loc(3) | 4 x (pA(3) pB(3) gap)
#define S16_MAX_BBC 16            // box-box pairs per wave up to which they are worked on by 16-lane groups
#define S16_MAX_MPR 64            // (= every hit: an arm folded onto itself and jammed into the table has 20+ hull pairs in range)
static_assert(860 + 20 <= 896, "narrowphase staging exceeds the scratch area");
static_assert(S16_NP_B_(1) + 1136 + 16 <= S16_REC_(1) && S16_NP_B_(2) + 1136 + 16 <= S16_REC_(2) && S16_NP_B_(4) + 1136 + 16 <= S16_REC_(4), "narrowphase lists run into the contact records");
static_assert((S16_NP_B_(1) + 192 + 192) % 4 == 0 && (S16_NP_B_(2) + 192 + 192) % 4 == 0 && (S16_NP_B_(4) + 192 + 192) % 4 == 0, "point pool: 16-byte aligned");
static_assert(S16_PT + 7 * (S16_PT_KIN_(1) + S16_MAX_KIN) <= S16_U_(1) && S16_PT + 7 * (S16_PT_KIN_(4) + S16_MAX_KIN) <= S16_COM_(4) && S16_COM_(4) + 18 <= S16_U_(4), "pose table runs into the union");
static_assert(MSSIM_MAX_HITS == S16_MAX_HIT && MSSIM_MAX_CONTACTS == MAXC, "capacity constants out of sync with include/mssim.h");
static_assert(MSSIM_MAX_TRI_TASKS + MSSIM_MAX_TRI_HITS <= 792 - 704, "triangle task list + candidates exceed the scratch they borrow (BSCR + KEEP)");

// Timing by repetition (scripts/ab_variants.sh with -DEXP_DUP_<PHASE>): an idempotent phase is run twice, the launch gets
// longer by what the phase costs inside the product kernel -- same physics, no clock reads, no extra waits. (A knock-out that
// leaves a phase out changes what runs downstream; the phase clocks wait for all outstanding memory operations at every
// read: DESIGN.md section 3 has what each of them got wrong.)
#define EXP_DUP(flag) for (int dup_ = 0; dup_ < ((flag) ? 2 : 1); dup_++)
#ifdef EXP_DUP_SHAPE
#define DUP_SHAPE 1
#else
#define DUP_SHAPE 0
#endif
#ifdef EXP_DUP_CULL
#define DUP_CULL 1
#else
#define DUP_CULL 0
#endif
#ifdef EXP_DUP_CULL1
#define DUP_CULL1 1
#else
#define DUP_CULL1 0
#endif
#ifdef EXP_DUP_CULL2
#define DUP_CULL2 1
#else
#define DUP_CULL2 0
#endif
#ifdef EXP_DUP_GJ
#define DUP_GJ 1
#else
#define DUP_GJ 0
#endif
#ifdef EXP_DUP_FK
#define DUP_FK 1
#else
#define DUP_FK 0
#endif
#ifdef EXP_DUP_IROT
#define DUP_IROT 1
#else
#define DUP_IROT 0
#endif
// Phase timing aid (scripts/phase_clocks.py builds a separate library with -DMSSIM_PHASE_CLOCKS):
// per-phase cycle deltas are kept in registers and flushed once at the end.
#ifdef MSSIM_PHASE_CLOCKS
__device__ unsigned long long g_phase_clk[32];
__device__ unsigned g_phase_blk[2048 * 32];  // per-block deltas of the most recent launch
#define PH_INIT                 \
  unsigned ph_d[32];            \
  _Pragma("unroll") for (int i_ = 0; i_ < 32; i_++) ph_d[i_] = 0u; \
  unsigned ph_t = (unsigned)clock64();
#define PH(i)                                   \
  do {                                          \
    __builtin_amdgcn_s_waitcnt(0);              \
    ph_d[i] += (unsigned)clock64() - ph_t;      \
    ph_t = (unsigned)clock64();                 \
  } while (0)
#define PH_ADD(i, v) ph_d[i] += (unsigned)(v)
#define BT_ADD(i, v)
#define BT_T0
#define BT_T0b
#define BT_T(i)
#define PH_FLUSH                                                                  \
  if ((threadIdx.x & 63) == 0) {  /* one record per wave */                        \
    const int w_ = blockIdx.x * S16_WAVES + (threadIdx.x >> 6);                    \
    _Pragma("unroll") for (int i_ = 0; i_ < 32; i_++)                             \
      if (ph_d[i_]) atomicAdd(&g_phase_clk[i_], (unsigned long long)ph_d[i_]);    \
    if (w_ < 2048) {                                                              \
      _Pragma("unroll") for (int i_ = 0; i_ < 32; i_++) g_phase_blk[w_ * 32 + i_] = ph_d[i_]; \
    }                                                                             \
  }
#elif defined(MSSIM_BLOCK_TIMES)
// scripts/block_times.py: start / duration (100 MHz wall clock), core-clock cycles and placement (HW_ID,
// XCC_ID) of every block of the most recent launch, with the work counters of the phase-clock build --
// two clock reads per block, so the timing is that of the product kernel
__device__ unsigned g_phase_blk[2048 * 32];
#define PH_INIT                                             \
  unsigned ph_d[32];                                        \
  _Pragma("unroll") for (int i_ = 0; i_ < 32; i_++) ph_d[i_] = 0u; \
  const unsigned long long bt_w0 = wall_clock64(), bt_c0 = clock64();
#define PH(i)
#define PH_ADD(i, v) ph_d[i] += (unsigned)(v)
#define BT_ADD(i, v) ph_d[i] += (unsigned)(v)  // (slots that are cycle counts in the phase-clock build)
#define BT_T0 unsigned bt_t_ = (unsigned)clock64()
#define BT_T0b bt_t_ = (unsigned)clock64()
#define BT_T(i) do { __builtin_amdgcn_s_waitcnt(0); ph_d[i] += (unsigned)clock64() - bt_t_; bt_t_ = (unsigned)clock64(); } while (0)
#define PH_FLUSH                                                                   \
  if ((threadIdx.x & 63) == 0 && blockIdx.x * S16_WAVES + (threadIdx.x >> 6) < 2048) {  /* one record per wave */ \
    unsigned* o_ = g_phase_blk + (blockIdx.x * S16_WAVES + (threadIdx.x >> 6)) * 32;  \
    _Pragma("unroll") for (int i_ = 8; i_ < 32; i_++) o_[i_] = ph_d[i_];           \
    o_[0] = (unsigned)bt_w0; o_[1] = (unsigned)(bt_w0 >> 32);                      \
    o_[2] = (unsigned)(wall_clock64() - bt_w0);                                    \
    o_[3] = (unsigned)(clock64() - bt_c0);                                         \
    o_[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                             \
    o_[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                            \
  }
#else
#define PH_INIT
#define PH(i)
#define PH_ADD(i, v)
#define BT_ADD(i, v)
#define BT_T0
#define BT_T0b
#define BT_T(i)
#define PH_FLUSH
#endif

template <int CTRL>
MS_DEV float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
// all-reduce (sum) inside each 16-lane DPP row: rotations by 8, 4, 2, 1
MS_DEV float gsum16(float x) {
  x += dpp_f<0x128>(x);
  x += dpp_f<0x124>(x);
  x += dpp_f<0x122>(x);
  x += dpp_f<0x121>(x);
  return x;
}
// three independent 16-lane all-reduces, interleaved, each step ONE v_add_f32_dpp (x += row_ror(x)).
// Written as inline asm because the compiler pairs the independent adds into v_pk_add_f32, which cannot
// take a DPP operand, and then needs a separate v_mov_b32_dpp (+ a mov for its `old` operand) per
// step. Hazard: a VGPR written by a VALU op may be read by a DPP op only 2 wait states later -- inside the
// block two other instructions always sit between a write and the next read of the same register; the
// leading s_nop covers the producers of the inputs.
MS_DEV void gsum16x3(float& a, float& b, float& c) {
  asm volatile(
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 row_ror:2 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      : "+v"(a), "+v"(b), "+v"(c));
}
// NR rows per env: the sums over an env's lanes. One more step after the row all-reduce: v_permlane16_swap exchanges the odd
// rows of its first operand with the even rows of its second, so with both operands = x the two results are (row 0's
// value in rows 0 and 1 | row 2's in rows 2 and 3) and (row 1's | row 3's): their sum is the total of each row pair.
template <int NR>
MS_DEV float xrow_sum(float x) {
  if constexpr (NR == 1) {
    return x;
  } else {
    static_assert(NR == 2 || NR == 4, "rows per env");
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    if constexpr (NR == 4) {  // (and the two halves of the wave: v_permlane32_swap exchanges the upper half of one operand with the lower half of the other)
      const auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
      s = __uint_as_float(h[0]) + __uint_as_float(h[1]);
    }
    return s;
  }
}
template <int NR>
MS_DEV void gsumx3(float& a, float& b, float& c) {
  gsum16x3(a, b, c);
  if constexpr (NR > 1) { a = xrow_sum<NR>(a); b = xrow_sum<NR>(b); c = xrow_sum<NR>(c); }
}
// W = A^-1 row (pre-rotated, see the row build) * J of the lane a row rotation by k reads from, for the
// three rows of a contact at once: one v_fmac_f32 with a DPP operand per term, no LDS round trip.
// Inline asm: the compiler keeps v_mov_b32_dpp + v_fma pairs otherwise. The leading s_nop covers the
// VALU-write -> DPP-read hazard for the producers of J; nothing inside the block writes a DPP source.
MS_DEV void rot_fma3(const float (&I)[16], const float (&J3)[3], float (&W3)[3]) {
  float w0 = I[0] * J3[0], w1 = I[0] * J3[1], w2 = I[0] * J3[2];
  asm("s_nop 1\n"
      "v_fmac_f32_dpp %0, %3, %7 row_ror:1 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %7 row_ror:1 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %7 row_ror:1 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %8 row_ror:2 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %8 row_ror:2 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %8 row_ror:2 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %9 row_ror:3 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %9 row_ror:3 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %9 row_ror:3 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %10 row_ror:4 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %10 row_ror:4 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %10 row_ror:4 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %11 row_ror:5 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %11 row_ror:5 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %11 row_ror:5 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %12 row_ror:6 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %12 row_ror:6 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %12 row_ror:6 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %13 row_ror:7 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %13 row_ror:7 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %13 row_ror:7 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %14 row_ror:8 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %14 row_ror:8 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %14 row_ror:8 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %15 row_ror:9 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %15 row_ror:9 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %15 row_ror:9 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %16 row_ror:10 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %16 row_ror:10 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %16 row_ror:10 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %17 row_ror:11 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %17 row_ror:11 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %17 row_ror:11 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %18 row_ror:12 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %18 row_ror:12 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %18 row_ror:12 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %19 row_ror:13 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %19 row_ror:13 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %19 row_ror:13 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %20 row_ror:14 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %20 row_ror:14 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %20 row_ror:14 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %3, %21 row_ror:15 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %1, %4, %21 row_ror:15 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %2, %5, %21 row_ror:15 row_mask:0xf bank_mask:0xf\n"
      : "+v"(w0), "+v"(w1), "+v"(w2)
      : "v"(J3[0]), "v"(J3[1]), "v"(J3[2]), "v"(I[0]), "v"(I[1]), "v"(I[2]), "v"(I[3]), "v"(I[4]), "v"(I[5]), "v"(I[6]), "v"(I[7]),
        "v"(I[8]), "v"(I[9]), "v"(I[10]), "v"(I[11]), "v"(I[12]), "v"(I[13]), "v"(I[14]), "v"(I[15]));
  W3[0] = w0; W3[1] = w1; W3[2] = w2;
}
template <int K>
MS_DEV void rot_gather(const float* row, int c, float (&Irot)[16]) {
  if constexpr (K == 0) {
    Irot[0] = row[c];
    rot_gather<1>(row, c, Irot);
  } else if constexpr (K < 16) {
    Irot[K] = row[__builtin_amdgcn_update_dpp(0, c, 0x120 + K, 0xF, 0xF, false)];
    rot_gather<K + 1>(row, c, Irot);
  }
}
// One pivot of the Gauss-Jordan elimination of k_solve16 (matrix row per lane, env per 16-lane DPP row), see the call site.
template <int K, int NA>
MS_DEV void gj_eliminate(float (&A)[NA], float (&I)[16], float& diag, int c, int n) {
  if constexpr (K < NA) {
    if (K >= n) return;
    const float inv = rcp_f(dpp_f<0x150 + K>(A[K]));  // row_newbcast:K
    const float nfac = c == K ? 0.f : -A[K] * inv;
    diag = c == K ? inv : diag;
#pragma unroll
    for (int j = K + 1; j < NA; j++) A[j] = fmaf(dpp_f<0x150 + K>(A[j]), nfac, A[j]);
#pragma unroll
    for (int j = 0; j < K; j++) I[j] = fmaf(dpp_f<0x150 + K>(I[j]), nfac, I[j]);
    I[K] += nfac;
    gj_eliminate<K + 1, NA>(A, I, diag, c, n);
  }
}

// The joint-limit rows of one sweep of k_solve16, joints in order (see the call site): rows without a limit have lim_inv = 0 and
// lim_lam = 0, so their d(lambda) is 0.
template <int J, int NL>
MS_DEV void limit_rows_pass(const float (&wj)[NL], float& v_c, float& lim_lam, float lim_side, float lim_inv, float bl, int c, int n) {
  if constexpr (J < NL) {
    if (J >= n) return;
    const float nl = fmaxf(lim_lam - (lim_side * v_c + bl) * lim_inv, 0.f);
    const float dl = dpp_f<0x150 + J>(nl - lim_lam);  // row_newbcast:J
    lim_lam = c == J ? nl : lim_lam;
    v_c = fmaf(wj[J], dl, v_c);
    limit_rows_pass<J + 1, NL>(wj, v_c, lim_lam, lim_side, lim_inv, bl, c, n);
  }
}
MS_DEV float gbc(float x, int j) { return __shfl(x, j, 16); }
MS_DEV int gbci(int x, int j) { return __shfl(x, j, 16); }

MS_DEV void ld16(const float* p, float* out) {  // 16 consecutive floats (16-B aligned)
  const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float4 t = q[i];
    out[4 * i] = t.x; out[4 * i + 1] = t.y; out[4 * i + 2] = t.z; out[4 * i + 3] = t.w;
  }
}
MS_DEV pose_t lds_pose(const float* b) { return pose_t{f3{b[0], b[1], b[2]}, q4{b[3], b[4], b[5], b[6]}}; }
MS_DEV void lds_pose_store(float* b, pose_t P) {
  b[0] = P.p.x; b[1] = P.p.y; b[2] = P.p.z; b[3] = P.q.w; b[4] = P.q.x; b[5] = P.q.y; b[6] = P.q.z;
}
MS_DEV float gmax16(float x) {
  x = fmaxf(x, dpp_f<0x128>(x));
  x = fmaxf(x, dpp_f<0x124>(x));
  x = fmaxf(x, dpp_f<0x122>(x));
  x = fmaxf(x, dpp_f<0x121>(x));
  return x;
}
MS_DEV int gmin16i(int x) {
  x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, false));
  x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x124, 0xF, 0xF, false));
  x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x122, 0xF, 0xF, false));
  x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x121, 0xF, 0xF, false));
  return x;
}
// Box-box manifold by the 16 lanes of a group (same result as collide_box_box, which spends ~3400
// instructions of one lane on it): the separating-axis search is evaluated redundantly, then lane
// i < 8 owns polygon vertex i. A Sutherland-Hodgman pass is one step for all vertices: distance of
// the own and the next vertex, output slots from two ballots (vertex order preserved), scatter /
// gather through 8 x 3 LDS words. The 4-point reduction ("first extremum wins" scans) becomes
// max-reductions + ballots. Result -> out[20]: count | n | 4 x (x y z sep); written by the owning lanes.
// `scr`: 24 floats of the group. The group's lanes must be converged; no barrier is used (one wave:
// LDS operations complete in order), so other groups of the wave may be inactive.
MS_DEV void collide_box_box_coop(const shape_t& A, const shape_t& B, float offset, float* scr, float* out, int c, int g) {
  manifold_t m;
  bb_face_t F;
  const int mode = bb_setup(A, B, offset, m, F);
  if (mode != 2) {
    if (c == 0) {
      out[0] = __int_as_float(m.count);
      out[1] = m.n.x; out[2] = m.n.y; out[3] = m.n.z;
      out[4] = m.x[0].x; out[5] = m.x[0].y; out[6] = m.x[0].z; out[7] = m.sep[0];
    }
    return;
  }
  auto ballot16 = [&](bool b) __attribute__((always_inline)) { return (unsigned)(__ballot(b) >> (16 * g)) & 0xFFFFu; };
  const unsigned below = (1u << c) - 1u;
  // incident face: vertex c of (+,+) (-,+) (-,-) (+,-)
  f3 P = F.fc + F.y1 * ((c == 0 || c == 3) ? F.hY1 : -F.hY1) + F.y2 * (c < 2 ? F.hY2 : -F.hY2);
  int np = 4;
  auto clip = [&](f3 pn, float pd) __attribute__((always_inline)) {
    const bool valid = c < np;
    const float da = dot(pn, P) - pd;
    const int nxt = (c + 1 < np) ? c + 1 : 0;
    const f3 b = f3{gbc(P.x, nxt), gbc(P.y, nxt), gbc(P.z, nxt)};
    const float db = gbc(da, nxt);
    const bool ea = valid && da <= 0.f;
    const bool et = valid && ((da < 0.f && db > 0.f) || (da > 0.f && db < 0.f));
    const unsigned ma = ballot16(ea), mt = ballot16(et);
    const int pa = __popc(ma & below) + __popc(mt & below);
    const int pt = pa + (ea ? 1 : 0);
    if (ea && pa < 8) { scr[3 * pa] = P.x; scr[3 * pa + 1] = P.y; scr[3 * pa + 2] = P.z; }
    if (et && pt < 8) {
      const float t = da * rcp_safe(da - db);
      const f3 x = P + (b - P) * t;
      scr[3 * pt] = x.x; scr[3 * pt + 1] = x.y; scr[3 * pt + 2] = x.z;
    }
    const int tot = __popc(ma) + __popc(mt);
    np = tot < 8 ? tot : 8;
    const int rd = c & 7;
    P = f3{scr[3 * rd], scr[3 * rd + 1], scr[3 * rd + 2]};
  };
  clip(F.x1, dot(F.x1, F.Xc) + F.hX1);
  clip(-F.x1, -dot(F.x1, F.Xc) + F.hX1);
  clip(F.x2, dot(F.x2, F.Xc) + F.hX2);
  clip(-F.x2, -dot(F.x2, F.Xc) + F.hX2);
  // points within the offset, projected half way onto the reference face
  const float sp = dot(P - F.Xc, F.nref) - F.hXr;
  const bool keep = c < np && sp <= offset;
  const unsigned mk = ballot16(keep);
  const int n = __popc(mk);
  const f3 q = P - F.nref * (0.5f * sp);
  int slot = -1, count = n;
  if (n <= 4) {
    slot = keep ? __popc(mk & below) : -1;
  } else {
    // deepest, farthest from it, then the two of largest area on either side; earliest candidate wins ties
    const float mn = -gmax16(keep ? -sp : -3e38f);
    const int l0 = __ffs(ballot16(keep && sp == mn)) - 1;
    const f3 p0 = f3{gbc(q.x, l0), gbc(q.y, l0), gbc(q.z, l0)};
    const f3 d0 = q - p0;
    const float v1 = dot(d0, d0);
    const bool c1 = keep && c != l0;
    const float m1 = gmax16(c1 ? v1 : -1.f);
    const int l1 = __ffs(ballot16(c1 && v1 == m1)) - 1;
    const f3 e = f3{gbc(q.x, l1), gbc(q.y, l1), gbc(q.z, l1)} - p0;
    const float ar = dot(cross(e, q - p0), F.nref);
    const bool c2 = c1 && c != l1;
    const float m2 = gmax16(c2 ? fabsf(ar) : -1.f);
    const int l2 = __ffs(ballot16(c2 && fabsf(ar) == m2)) - 1;
    const float sgn2 = gbc(ar, l2);
    const bool c3 = c2 && c != l2;
    const float v3 = sgn2 >= 0.f ? -ar : ar;
    const float m3 = gmax16(c3 ? v3 : 0.f);
    const int l3 = m3 > 0.f ? __ffs(ballot16(c3 && v3 == m3)) - 1 : -1;
    slot = c == l0 ? 0 : (c == l1 ? 1 : (c == l2 ? 2 : (c == l3 ? 3 : -1)));
    count = l3 >= 0 ? 4 : 3;
  }
  if (c == 0) {
    const f3 nn = F.refA ? -F.nref : F.nref;
    out[0] = __int_as_float(count);
    out[1] = nn.x; out[2] = nn.y; out[3] = nn.z;
  }
  if (slot >= 0) { out[4 + 4 * slot] = q.x; out[5 + 4 * slot] = q.y; out[6 + 4 * slot] = q.z; out[7 + 4 * slot] = sp; }
}
// A convex hull against a plane by the 16 lanes of a group (lane c holds vertices c, c + 16, c + 32, c + 48): the hull's
// vertices inside the contact offset; when more than 4 of them lie within MSSIM_PATCH_SLACK of the lowest -- a hull lying on
// a face: the rim of a cup, the foot of a post -- four are taken by EXTENT among those (the patch rule: the deepest, the
// farthest from it, the largest area on either side of that edge, first candidate within the tie tolerance of every
// extremum), else the 4 deepest (lowest index among equals). Same points in the same order as the oracle's collide_plane.
// Result -> out[20] like collide_box_box_coop: count | n | 4 x (x y z sep). The group's lanes must be converged.
MS_DEV void collide_plane_hull_coop(const shape_t& pl, const shape_t& b, float offset, float* out, int c, int g) {
  auto ballot16 = [&](bool v) __attribute__((always_inline)) { return (unsigned)(__ballot(v) >> (16 * g)) & 0xFFFFu; };
  const f3 np = mcol(pl.rot, 0);
  const int nv = b.nverts < 64 ? b.nverts : 64;
  f3 X[4];
  float Sp[4];
  bool in[4];
  int n = 0;
  float smin = 3e38f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = c + 16 * k;
    const bool exists = i < nv;
    const float* v = b.verts + 3 * (exists ? i : 0);
    const f3 p = b.c + mmulv(b.rot, f3{v[0], v[1], v[2]});
    const float s = dot(np, p - pl.c);
    X[k] = p - np * (0.5f * s);
    Sp[k] = s;
    in[k] = exists && s < offset;
    n += __popc(ballot16(in[k]));
    smin = fminf(smin, in[k] ? s : 3e38f);
  }
  smin = -gmax16(-smin);
  // the point at vertex index `id` (owner lane id & 15, slot id >> 4), from its owner
  auto point_at = [&](int id) __attribute__((always_inline)) {
    const int owner = id & 15, slot = id >> 4;
    const f3 mine = slot == 0 ? X[0] : (slot == 1 ? X[1] : (slot == 2 ? X[2] : X[3]));
    const float ms = slot == 0 ? Sp[0] : (slot == 1 ? Sp[1] : (slot == 2 ? Sp[2] : Sp[3]));
    return float4{gbc(mine.x, owner), gbc(mine.y, owner), gbc(mine.z, owner), gbc(ms, owner)};
  };
  auto first_pos = [&](const bool (&ok)[4]) __attribute__((always_inline)) {
    int p = 1 << 20;
#pragma unroll
    for (int k = 3; k >= 0; k--) p = ok[k] ? c + 16 * k : p;
    p = gmin16i(p);
    return p == (1 << 20) ? -1 : p;
  };
  int pick[4] = {-1, -1, -1, -1};
  int near_ = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) near_ += __popc(ballot16(in[k] && Sp[k] <= smin + MSSIM_PATCH_SLACK));
  if (n > 4 && near_ > 4) {
    bool has[4], ok[4];
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { has[k] = in[k] && Sp[k] <= smin + MSSIM_PATCH_SLACK; ok[k] = has[k] && Sp[k] <= smin + MSSIM_PATCH_TIE_SEP; }
    const int q0 = first_pos(ok);
    const float4 P0 = point_at(q0);
    const f3 p0 = f3{P0.x, P0.y, P0.z};
    float best = -1.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const f3 d = X[k] - p0;
      v[k] = dot(d, d);
      has[k] = has[k] && c + 16 * k != q0;
      best = fmaxf(best, has[k] ? v[k] : -1.f);
    }
    best = gmax16(best);
#pragma unroll
    for (int k = 0; k < 4; k++) ok[k] = has[k] && v[k] >= best - MSSIM_PATCH_TIE_REL * best;
    const int q1 = first_pos(ok);
    const float4 P1 = point_at(q1);
    const f3 ed = f3{P1.x, P1.y, P1.z} - p0;
    const f3 na = -np;
    best = -1.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      v[k] = dot(cross(ed, X[k] - p0), na);
      has[k] = has[k] && c + 16 * k != q1;
      best = fmaxf(best, has[k] ? fabsf(v[k]) : -1.f);
    }
    best = gmax16(best);
#pragma unroll
    for (int k = 0; k < 4; k++) ok[k] = has[k] && fabsf(v[k]) >= best - MSSIM_PATCH_TIE_REL * best;
    const int q2 = first_pos(ok);
    const float4 P2 = point_at(q2);
    const float sgn2 = dot(cross(ed, f3{P2.x, P2.y, P2.z} - p0), na);
    best = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      v[k] = sgn2 >= 0.f ? -v[k] : v[k];
      has[k] = has[k] && c + 16 * k != q2;
      best = fmaxf(best, has[k] ? v[k] : 0.f);
    }
    best = gmax16(best);
    int q3 = -1;
    if (best > MSSIM_PATCH_TIE_REL * fabsf(sgn2)) {
#pragma unroll
      for (int k = 0; k < 4; k++) ok[k] = has[k] && v[k] >= best - MSSIM_PATCH_TIE_REL * best;
      q3 = first_pos(ok);
    }
    pick[0] = q0; pick[1] = q1; pick[2] = q2; pick[3] = q3;
  } else {
    // the 4 deepest, the lowest index among equals
    bool rem[4];
#pragma unroll
    for (int k = 0; k < 4; k++) rem[k] = in[k];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float best = 3e38f;
#pragma unroll
      for (int k = 0; k < 4; k++) best = fminf(best, rem[k] ? Sp[k] : 3e38f);
      best = -gmax16(-best);
      int id = 1 << 20;
#pragma unroll
      for (int k = 3; k >= 0; k--) id = (rem[k] && Sp[k] == best) ? c + 16 * k : id;
      id = gmin16i(id);
      pick[r] = id == (1 << 20) ? -1 : id;
#pragma unroll
      for (int k = 0; k < 4; k++) rem[k] = rem[k] && c + 16 * k != id;
    }
  }
  int count = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (pick[r] < 0) continue;  // (group-uniform)
    const float4 P = point_at(pick[r]);
    if (c == 0) { out[4 + 4 * count] = P.x; out[5 + 4 * count] = P.y; out[6 + 4 * count] = P.z; out[7 + 4 * count] = P.w; }
    count++;
  }
  if (c == 0) {
    out[0] = __int_as_float(count);
    out[1] = -np.x; out[2] = -np.y; out[3] = -np.z;
  }
}
// pose-table slot of a body: -1 = fixed in the env frame
template <int NR>
MS_DEV int pose_slot(int kind, int index) {
  if (kind == MSSIM_BODY_ART) return index < 0 ? 0 : S16_PT_LINK + index;
  if (kind == MSSIM_BODY_FREE) return S16_PT_FREE + index;
  if (kind == MSSIM_BODY_KIN) return S16_PT_KIN + index;
  return -1;
}
// lanes (velocity components) that move with the body in pose-table slot `sl`: the dofs on the path
// to a link, the 6 components of a free body, nothing for fixed / kinematic bodies
// (`row`: which 16-lane row of the env the mask is for. NR = 1: joints, then the free bodies, all in row 0; NR > 1: the joints
// in row 0, free bodies 2k and 2k + 1 from lane 0 of row 1 + k)
template <int NR>
MS_DEV unsigned slot_lane_mask(const float* L, int sl, int n, int row) {
  if (sl >= S16_PT_LINK && sl < S16_PT_FREE) return row == 0 ? reinterpret_cast<const unsigned*>(L)[S16_ANC + sl - S16_PT_LINK] : 0u;
  if (sl >= S16_PT_FREE && sl < S16_PT_KIN) {
    const int b = sl - S16_PT_FREE;
    if constexpr (NR == 1) return 0x3Fu << (n + 6 * b);
    else return row == 1 + (b >> 1) ? 0x3Fu << (6 * (b & 1)) : 0u;
  }
  return 0u;
}
// separating-axis test of two oriented boxes (rotations RA / RB, half extents ha / hb, d = centre B -
// centre A), radii enlarged by `margin`; true = certainly apart (Gottschalk et al., OBBTree)
MS_DEV bool obb_separated(const m3& RA, f3 ha, const m3& RB, f3 hb, f3 d, float margin) {
  float R[3][3], AR[3][3];
  const float t[3] = {dot(mcol(RA, 0), d), dot(mcol(RA, 1), d), dot(mcol(RA, 2), d)};
  const float a[3] = {ha.x, ha.y, ha.z}, b[3] = {hb.x, hb.y, hb.z};
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) { R[i][j] = dot(mcol(RA, i), mcol(RB, j)); AR[i][j] = fabsf(R[i][j]) + 1e-6f; }
  bool sep = false;
#pragma unroll
  for (int i = 0; i < 3; i++) sep = sep || fabsf(t[i]) > a[i] + b[0] * AR[i][0] + b[1] * AR[i][1] + b[2] * AR[i][2] + margin;
#pragma unroll
  for (int j = 0; j < 3; j++)
    sep = sep || fabsf(t[0] * R[0][j] + t[1] * R[1][j] + t[2] * R[2][j]) > b[j] + a[0] * AR[0][j] + a[1] * AR[1][j] + a[2] * AR[2][j] + margin;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const float ra = a[i1] * AR[i2][j] + a[i2] * AR[i1][j];
      const float rb = b[j1] * AR[i][j2] + b[j2] * AR[i][j1];
      sep = sep || fabsf(t[i2] * R[i1][j] - t[i1] * R[i2][j]) > ra + rb + margin;
    }
  return sep;
}
// world shape from the LDS shape table of one env
MS_DEV shape_t shape_from_table(const DevModel& M, const float* t) {
  shape_t sh;
  sh.c = f3{t[0], t[1], t[2]};
  sh.rot = qmat(q4{t[3], t[4], t[5], t[6]});
  sh.p0 = t[7]; sh.p1 = t[8]; sh.p2 = t[9];
  const unsigned pk = __float_as_uint(t[14]);  // type:3 | nverts:7 | (pose slot + 1):5 | first hull vertex:17
  sh.type = (int)(pk & 7u);
  sh.nverts = (int)((pk >> 3) & 127u);
  sh.verts = M.hull_verts + 3 * (size_t)(pk >> 15);
  return sh;
}

// MPR support policy of a 16-lane group working on ONE pair: lane c keeps hull vertices c, c + 16,
// c + 32, c + 48 of each convex shape in registers for the whole task; a support query is 4 dot
// products per lane and a DPP row all-reduce of (value, index, point) with "larger value, then lower
// index" -- exactly the first maximum of the sequential scan, identical in every lane of the group.
struct SupCoop16 {
  f3 va[4], vb[4];
  int c;
  static MS_DEV void load_one(f3 (&dst)[4], const shape_t& s, int c) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      dst[k] = f3{0, 0, 0};
      if (s.type == SH_CONVEX) {
        const int i = c + 16 * k < s.nverts ? c + 16 * k : 0;
        dst[k] = f3{s.verts[3 * i], s.verts[3 * i + 1], s.verts[3 * i + 2]};
      }
    }
  }
  static MS_DEV f3 scan(const f3 (&v)[4], const shape_t& s, f3 d, int c) {
    const f3 dl = mtmulv(s.rot, d);
    // this lane's first maximum over its vertices c, c + 16, c + 32, c + 48
    float t = dot(v[0], dl);
    int i = c;
#pragma unroll
    for (int k = 1; k < 4; k++) {
      const float tk = dot(v[k], dl);
      const bool g = tk > t && c + 16 * k < s.nverts;
      t = g ? tk : t; i = g ? c + 16 * k : i;
    }
    if (c >= s.nverts) { t = -3e38f; i = 1 << 20; }
    // group maximum of the value, then the lowest index among the lanes that reach it (= the first
    // maximum of the sequential scan), then the point from its owner lane
    float tm = t;
    tm = fmaxf(tm, dpp_f<0x128>(tm)); tm = fmaxf(tm, dpp_f<0x124>(tm)); tm = fmaxf(tm, dpp_f<0x122>(tm)); tm = fmaxf(tm, dpp_f<0x121>(tm));
    int im = t == tm ? i : (1 << 20);
    im = min(im, __builtin_amdgcn_update_dpp(0, im, 0x128, 0xF, 0xF, false));
    im = min(im, __builtin_amdgcn_update_dpp(0, im, 0x124, 0xF, 0xF, false));
    im = min(im, __builtin_amdgcn_update_dpp(0, im, 0x122, 0xF, 0xF, false));
    im = min(im, __builtin_amdgcn_update_dpp(0, im, 0x121, 0xF, 0xF, false));
    const int slot = im >> 4, owner = im & 15;
    const f3 mine = sel3(slot == 0, v[0], sel3(slot == 1, v[1], sel3(slot == 2, v[2], v[3])));
    const f3 p = f3{__shfl(mine.x, owner, 16), __shfl(mine.y, owner, 16), __shfl(mine.z, owner, 16)};
    return s.c + mmulv(s.rot, p);
  }
  // (no pointer select between the two arrays: that would force them out of registers)
  MS_DEV f3 operator()(int which, const shape_t& s, f3 d) const {
    if (s.type != SH_CONVEX) return support(s, d);
    if (which == 0) return scan(va, s, d, c);
    return scan(vb, s, d, c);
  }
};

// NDOF > 0: the number of joints is a compile-time constant (9 for the Panda), so the loops over the
// joints are unrolled and their masks folded; NDOF = 0 reads it from the model.
// TASK > 0 (with FUSED): the launch ends with the copy-out (mssim_fetch) of its envs and the evaluate / obs /
// reward epilogue of a task -- 1 PickCube, 2 PushCube, 3 PegInsertionSide -- so that a whole control step
// (action map, substeps, copy-out, epilogue) is one launch.
// TRI: the model has triangle-mesh shapes (MSSIM_SHAPE_TRIMESH): the narrowphase carries the mesh stage (BVH traversal, one
// multi-point manifold per triangle in range); instantiated without a task tail only.
template <int NDOF = 0, int TASK = 0, bool TRI = false, int NR = 1>
__global__ __launch_bounds__(64 * S16_WAVES) void k_solve16(DevModel M, DevState S, int n_sub) {
  constexpr bool FUSED = true;  // (the per-substep variant fed by a separate narrowphase kernel is gone)
  static_assert(NR == 1 || ((NR == 2 || NR == 4) && TASK == 0), "rows per env: 1, 2 or 4; the task tails are compiled for one row only");
  constexpr int EPW = 4 / NR;                 // envs per wave
  constexpr int BLK_ENVS = S16_WAVES * EPW;   // envs per block
  constexpr int BLK_GRPS = S16_WAVES * 4;     // 16-lane groups per block: who takes the tasks of the shared narrowphase stages
  constexpr int GW = 16 * NR;                 // lanes per env
  __shared__ __attribute__((aligned(16))) float sm[BLK_ENVS * S16_ENV_FLOATS];
  __shared__ int blk_nml[BLK_ENVS];  // generic-convex pairs of every env of the block (stage B task list)
  __shared__ int blk_nbl[BLK_ENVS];  // its box-box pairs that go to 16-lane groups (stage C task list)
  __shared__ int blk_ntl[BLK_ENVS];  // its mesh-triangle tasks (stage T task list; TRI variants only)
  __shared__ int blk_npl[BLK_ENVS];  // its (plane, hull) pairs (stage C, after the box-box pairs)
  const int N = S.N;
  const int wv = threadIdx.x >> 6, lane64 = threadIdx.x & 63;
  const int g = lane64 >> 4, c = threadIdx.x & 15;  // 16-lane group within the wave (ballot slices), lane within the group
  const int r = NR > 1 ? (g & (NR - 1)) : 0;        // row of this lane within its env
  const int ew = g / NR;                            // env within the wave
  const bool lead = r == 0;                         // row 0: the articulation's lanes; does the env's own narrowphase bookkeeping
  const int cl = 16 * r + c;                        // lane within the env
  const int gb = wv * EPW + ew;                     // env slot within the block
  const int grp = wv * 4 + g;                       // group within the block
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  if (chunk * BLK_ENVS >= N) return;  // grid padding
  const int e_raw = chunk * BLK_ENVS + gb;
  const bool live = e_raw < N;
  const int e = live ? e_raw : N - 1;  // dead groups shadow the last env and never store
  float* L = sm + gb * S16_ENV_FLOATS;
  float* const smw = sm + wv * EPW * S16_ENV_FLOATS;  // the envs of this wave
  const int n = NDOF > 0 ? NDOF : M.n_dof, nf = M.n_free;
  const float dt = M.dt;
  const float inv_dt = rcp_f(dt);
  const f3 g3 = f3{M.gx, M.gy, M.gz};
  const bool art = lead && c < n;
  // lane role among the free-body components (NR = 1: behind the joints in the same row; NR > 1: rows 1.. from lane 0, two bodies
  // per row). fb0 = first free body of this lane's row, nfr = how many it holds
  const int fc = NR == 1 ? c - n : (r >= 1 ? c : -1);
  const bool frow = NR == 1 || r >= 1;  // a row that holds free bodies' lanes
  const int fb0 = (NR == 4 && r >= 1) ? 2 * (r - 1) : 0;
  const int nfr = frow ? min(max(nf - fb0, 0), 2) : 0;
  const bool freel = fc >= 0 && fc < 6 * nfr;
  const int fb_id = fb0 + (freel ? fc / 6 : 0);
  const int fk = freel ? fc % 6 : 0;  // 0..2 linear xyz, 3..5 angular xyz
  const int fcol0 = NR == 1 ? n : 0;  // this row's first free-body column
  const int fbase = fcol0 + 6 * (fb_id - fb0);
  // ballot over the lanes of this env / broadcast from its lane `j` (j < 16: a lane of row 0)
  auto benv = [&](bool v) __attribute__((always_inline)) -> unsigned {  // (nonzero iff any lane of the env says so)
    if constexpr (NR == 1) return (unsigned)(__ballot(v) >> (16 * g)) & 0xFFFFu;
    else if constexpr (NR == 2) return (unsigned)(__ballot(v) >> (32 * ew));
    else return __ballot(v) != 0ull ? 1u : 0u;
  };
  auto env_bci = [&](int x, int j) __attribute__((always_inline)) -> int {
    return __shfl(x, j, 16 * NR);
  };
  (void)cl; (void)grp; (void)benv; (void)env_bci; (void)frow;

  PH_INIT
  // ---------------------------------------------------------------- carried state (loaded once)
  // Only what changes over the step stays in registers across the substeps (q, qd, body pose, joint
  // axis / anchor, free-body velocity, applied force); per-env constants are (re)loaded inside the
  // loop after the narrowphase so that they are not live -- and spilled -- while it runs.
  const pose_t root0 = pose_soa(S.root, 0, N, e);  // (kept in slot 0 of the LDS pose table, not in registers)
  const f3 O = root0.p;
  float q_c = 0.f, qd_c = 0.f;
  // (the body pose lives in the LDS pose table between its uses; the world joint axis / anchor are recomputed from
  // the parent's pose after every narrowphase: neither is carried in registers across it)
  if (art) {
    q_c = SOA(S.q, c); qd_c = SOA(S.qd, c);
  }
  // this control step's action -> drive target of joint c (mssim_step_action; same arithmetic as
  // k_apply_action). The target is read back by this lane in every substep.
  if (FUSED && S.act && art) {
    const int cj = S.act_col[c];
    if (cj >= 0) {
      float a = S.act[(size_t)e * S.act_dim + cj];
      const int fl = S.act_flags[c];
      if (fl & 2) {
        a = fminf(fmaxf(a, -1.f), 1.f);
        a = 0.5f * (S.act_hi[c] + S.act_lo[c]) + 0.5f * (S.act_hi[c] - S.act_lo[c]) * a;
      }
      if (fl & 48) {  // forward velocity of a planar base, given in its own frame: x / y joint get its cos / sin share
        const int jy = (fl >> 8) & 31;
        const float yaw = S.act_qpos ? S.act_qpos[(size_t)e * n + jy] : SOA(S.q, jy);
        a *= (fl & 16) ? cosf(yaw) : sinf(yaw);
      }
      const float qj = S.act_qpos ? S.act_qpos[(size_t)e * n + c] : q_c;
      const float t = ((fl & 1) ? qj : 0.f) + a;
      if (live) {
        if (fl & 8) {  // velocity drive target
          SOA(S.qdt, c) = a;
          if (S.act_target_vel) S.act_target_vel[(size_t)e * n + c] = a;
        } else {
          SOA(S.qt, c) = t;
          if (S.act_target) S.act_target[(size_t)e * n + c] = t;
        }
      }
    }
  }
  const unsigned self_c = art ? (1u << c) : 0u;
  float qacc_c = 0.f;
  // free-body velocity component of this lane; pose and the external force live in the pose table / registers
  float vfree_c = freel ? SOA(S.free_s, 13 * fb_id + 7 + fk) : 0.f;
  float fforce_c = (freel && fk < 3) ? SOA(S.free_force, 3 * fb_id + fk) : 0.f;
  // sleeping (include/mssim.h sleep_threshold): per free body the seconds left before it goes to sleep (<= 0: asleep)
  // and whether its mass-normalised kinetic energy is below the threshold ("calm"); both group-uniform
  int pcm_tick = S.pcm_tick[e];  // substep counter of the env's persistent-manifold cache (least-recently-used stamps)
  float fwake[S16_MAX_FREE];
  bool fcalm[S16_MAX_FREE];
  // 0.5 (v^2 + w . I w / m) of free body b at pose quaternion q (body-frame inertia about the centre of mass)
  auto norm_energy = [&](const float* in, f3 v, f3 w, q4 q) __attribute__((always_inline)) {
    const f3 wl = mtmulv(qmat(q), w);  // w . (R I R^T) w = (R^T w) . I (R^T w)
    return 0.5f * (dot(v, v) + dot(wl, smulv(s3{in[4], in[5], in[6], in[7], in[8], in[9]}, wl)) * rcp_f(in[0]));
  };
#pragma unroll
  for (int b = 0; b < S16_MAX_FREE; b++) {
    fwake[b] = 1.f;
    fcalm[b] = false;
    if (b < nf) {
      fwake[b] = SOA(S.free_wake, b);
      const pose_t P = pose_soa(S.free_s, 13 * b, N, e);
      const f3 v = f3{SOA(S.free_s, 13 * b + 7), SOA(S.free_s, 13 * b + 8), SOA(S.free_s, 13 * b + 9)};
      const f3 w = f3{SOA(S.free_s, 13 * b + 10), SOA(S.free_s, 13 * b + 11), SOA(S.free_s, 13 * b + 12)};
      float in0[10];
      free_inertial_of(M, N, b, e, in0);
      fcalm[b] = norm_energy(in0, v, w, P.q) < M.sleep_threshold;
      if (!(in0[0] > 0.f)) { fwake[b] = 0.f; fcalm[b] = true; }  // mass 0: the body does not exist in this env -- never awake
    }
  }
  // (body index known at run time: a select chain over the registers)
  auto fw_at = [&](int b) __attribute__((always_inline)) {
    float w = fwake[0];
#pragma unroll
    for (int k = 1; k < S16_MAX_FREE; k++) w = b == k ? fwake[k] : w;
    return w;
  };
  auto fc_at = [&](int b) __attribute__((always_inline)) {
    bool w = fcalm[0];
#pragma unroll
    for (int k = 1; k < S16_MAX_FREE; k++) w = b == k ? fcalm[k] : w;
    return w;
  };
  // pose table: root, links, free bodies, kinematic bodies
  if (lead) {
    reinterpret_cast<unsigned*>(L)[S16_ANC + c] = (art ? M.dof_anc[c] : 0u) | self_c;
    if (c == 0) lds_pose_store(L + S16_PT, root0);
    if (art) lds_pose_store(L + S16_BP + 7 * c, pose_soa(S.bodypose, 7 * c, N, e));
    if (c < nf) lds_pose_store(L + S16_PT + 7 * (S16_PT_FREE + c), pose_soa(S.free_s, 13 * c, N, e));
    if (FUSED && c < M.n_kin) lds_pose_store(L + S16_PT + 7 * (S16_PT_KIN + c), pose_soa(S.kin, 7 * c, N, e));
  }
  // previous step's hit list (pairs whose dense pair_cnt entry must be cleared if they are no longer in contact after
  // this step): lane c holds entries c, c + 16, c + 32. Read after the narrowphase of the LAST substep: early enough
  // for the two dependent global loads to be done when the impulse phase needs them, late enough not to be live
  // across the narrowphase.
  int nold = 0, oldp[3] = {-1, -1, -1};
  WSYNC();
  PH(0);

  float v_c = 0.f;
  int nrow_con = 0;  // contacts of this env in the current substep (group-uniform)
  float lim_lam = 0.f;

  for (int sub = 0; sub < n_sub; sub++) {
    const bool last = sub == n_sub - 1;
    // ================================================================ contacts -> LDS records
    pcm_tick++;
    BT_T0;
    int nc = 0;
    unsigned fdist = 0u;  // free bodies touched by a disturber in this substep (sleep counters)
#ifdef EXP_NO_NP
    if (false) {
#else
    if (FUSED) {
#endif
      // shape-local data of shapes c and c + 16 (two rows per env: cl and cl + 32) and the cull pairs of this lane (model constants,
      // fetched per substep rather than held in registers over the whole step)
      pose_t shF[2];
      float shP[2][3], shBr[2], shMu[2], shTr[2];
      f3 shBc[2], shH[2];
      unsigned shPk[2];
      int shSlot[2];
      {
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const int s = cl + GW * k;
          shF[k] = pose_t{f3{0, 0, 0}, q4{1, 0, 0, 0}};
          shP[k][0] = shP[k][1] = shP[k][2] = 0.f; shBr[k] = 0.f; shMu[k] = 0.f; shTr[k] = 0.f; shBc[k] = f3{0, 0, 0}; shH[k] = f3{0, 0, 0}; shPk[k] = 0u; shSlot[k] = -1;
          if (s < M.n_shape) {
            float r[24];  // the shape's 96-byte constant record
            const float4* rp = reinterpret_cast<const float4*>(M.shape_pack + 24 * s);
#pragma unroll
            for (int k2 = 0; k2 < 6; k2++) { const float4 t = rp[k2]; r[4 * k2] = t.x; r[4 * k2 + 1] = t.y; r[4 * k2 + 2] = t.z; r[4 * k2 + 3] = t.w; }
            int ty = __float_as_int(r[18]);
            const int slot = __float_as_int(r[21]);
            int hull_first = M.shape_hull[2 * s], hull_count = M.shape_hull[2 * s + 1];
            if (slot < 0) {
              shF[k] = pose_t{f3{r[0], r[1], r[2]}, qnormalized(q4{r[3], r[4], r[5], r[6]})};
              shP[k][0] = r[7]; shP[k][1] = r[8]; shP[k][2] = r[9];
              shBc[k] = f3{r[10], r[11], r[12]};
              shBr[k] = r[13];
              shH[k] = f3{r[14], r[15], r[16]};
            } else {
              shF[k] = pose_soa(M.env_shape_frame, 7 * slot, N, e);
              // device rows of a per-env shape (mssim_create): word 0 = type | vertex count << 3 | first vertex << 10,
              // rows 1..3 = the parameters of a primitive / the half extents of a hull's box
              const float* pp = M.env_shape_param + (size_t)(4 * slot) * N + e;
              const int w0 = __float_as_int(pp[0]);
              ty = w0 & 7;
              shP[k][0] = pp[(size_t)N]; shP[k][1] = pp[2 * (size_t)N]; shP[k][2] = pp[3 * (size_t)N];
              const float* bb = M.env_shape_bound + (size_t)(4 * slot) * N + e;
              shBc[k] = f3{bb[0], bb[(size_t)N], bb[2 * (size_t)N]};
              shBr[k] = bb[3 * (size_t)N];
              // per-env primitives are centred on their frame: box of the type's extents; a per-env hull / mesh brings its own
              shH[k] = (ty == SH_BOX || ty == SH_CONVEX || ty == SH_TRIMESH) ? f3{shP[k][0], shP[k][1], shP[k][2]}
                     : ty == SH_SPHERE ? f3{shP[k][0], shP[k][0], shP[k][0]}
                     : ty == SH_CAPSULE ? f3{shP[k][1] + shP[k][0], shP[k][0], shP[k][0]}
                     : f3{shP[k][1], shP[k][0], shP[k][0]};
              hull_count = (w0 >> 3) & 127; hull_first = w0 >> 10;
            }
            shMu[k] = r[17];
            shTr[k] = r[22];
            shSlot[k] = pose_slot<NR>(__float_as_int(r[19]), __float_as_int(r[20]));
            shPk[k] = (unsigned)ty | ((unsigned)hull_count << 3) | ((unsigned)(shSlot[k] + 1) << 10) | ((unsigned)hull_first << 15);
          }
        }
      }
      // ---- world shape table of this env
      EXP_DUP(DUP_SHAPE)
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int s = cl + GW * k;
        if (s < M.n_shape) {
          pose_t P = pose_t{f3{0, 0, 0}, q4{1, 0, 0, 0}};
          if (shSlot[k] >= 0) P = lds_pose(L + S16_PT + 7 * shSlot[k]);
          const pose_t W = pmul(P, shF[k]);
          const f3 bcw = P.p + qrot(P.q, shBc[k]);
          float* o = L + S16_NP_SHP + S16_SHP * s;
          lds_pose_store(o, W);
          o[7] = shP[k][0]; o[8] = shP[k][1]; o[9] = shP[k][2];
          o[10] = bcw.x; o[11] = bcw.y; o[12] = bcw.z; o[13] = shBr[k];
          o[14] = __uint_as_float(shPk[k]);
          o[15] = shMu[k];
          o[16] = shH[k].x; o[17] = shH[k].y; o[18] = shH[k].z; o[19] = shTr[k];
        }
      }
      WSYNC();
      PH(22);
      // ---- cull: 16 pairs of this env per round, survivors appended in pair order
      int nh = 0;
      int nmesh = 0;  // (TRI) surviving (convex shape, mesh) pairs of this env: a list of its own in the record area, idle until the narrowphase ends
      bool hit_over = false;
      // stage 1: bounding spheres / plane distance, all pairs (16 per round); survivors appended in pair order
      auto cull_round = [&](int p, int sab) __attribute__((always_inline)) {
        bool surv = false;
        const int sa = sab & 0xFF, sb = (sab >> 8) & 0xFF;
        if (sab >= 0) {
          const float* ta_ = L + S16_NP_SHP + S16_SHP * sa;
          const float* tb_ = L + S16_NP_SHP + S16_SHP * sb;
          const int ta = (int)(__float_as_uint(ta_[14]) & 7u);
          const f3 cb = f3{tb_[10], tb_[11], tb_[12]};
          const float ra = ta_[13], rb = tb_[13];
          bool cull;
          if (ta == SH_PLANE) {
            const m3 R = qmat(q4{ta_[3], ta_[4], ta_[5], ta_[6]});
            cull = dot(mcol(R, 0), cb - f3{ta_[0], ta_[1], ta_[2]}) > rb + M.contact_offset;
          } else {
            const f3 d = cb - f3{ta_[10], ta_[11], ta_[12]};
            const float rr = ra + rb + M.contact_offset;
            cull = dot(d, d) > rr * rr;
          }
          // a pair of bodies that cannot move in this substep (fixed in the env frame, or asleep at its start) needs no manifold
          const int s1a = (int)((__float_as_uint(ta_[14]) >> 10) & 31u), s1b = (int)((__float_as_uint(tb_[14]) >> 10) & 31u);
          auto inactive = [&](int s1) {
            const int b = s1 - 1 - S16_PT_FREE;
            return s1 == 0 || (b >= 0 && b < S16_MAX_FREE && fw_at(b) <= 0.f);
          };
          const int tb = (int)(__float_as_uint(tb_[14]) & 7u);
          surv = !cull && !(inactive(s1a) && inactive(s1b)) && ta != SH_NONE && tb != SH_NONE;  // (SH_NONE: no shape in this env's slot)
        }
        // survivors in pair order into the clip-scratch area (idle until the box-box manifolds)
        const unsigned long long bal = __ballot(surv);
        const unsigned m16 = (unsigned)(bal >> (16 * g)) & 0xFFFFu;
        if (surv) reinterpret_cast<int*>(L)[S16_NP_SCR + nh + __popc(m16 & ((1u << c) - 1u))] = p | (sa << 16) | (sb << 24);
        nh += __popc(m16);
      };
#ifdef EXP_CULL_ROUNDS  // (timing experiments only: wrong contacts)
      const int n_pair_cull = M.n_pair < 16 * EXP_CULL_ROUNDS ? M.n_pair : 16 * EXP_CULL_ROUNDS;
#else
      const int n_pair_cull = M.n_pair;
#endif
      EXP_DUP(DUP_CULL) {
      if (dup_) { WSYNC(); nh = 0; nmesh = 0; }  // (timing by repetition)
      // (the pair table -- sa | sb << 8 per pair, -1 behind the last -- comes straight from the model, one word per lane and
      // round: staging it in LDS first cost six dependent L2 round trips per substep)
      EXP_DUP(DUP_CULL1) {
        if (dup_) { WSYNC(); nh = 0; }
        int sab_next = lead ? M.pair_packed[c] : -1;  // (the next round's word is loaded before this round's tests)
#pragma unroll 1
        for (int base = 0; base < n_pair_cull; base += 16) {
          const int sab = sab_next;
          sab_next = lead ? M.pair_packed[base + 16 + c] : -1;
          cull_round(base + c, sab);  // (the lanes of an env's other rows idle through its own bookkeeping)
        }
      }
      WSYNC();
      // stage 2, on the survivors only (usually one or two rounds): 15-axis separating-axis test of the two
      // shapes' oriented boxes, contact offset added to the radii. It discards the pairs whose bounding
      // spheres overlap but whose shapes are apart -- most of the hull pairs that would otherwise run a
      // full MPR only to find no contact. Survivors go to the hit list, order preserved.
      const int nh_s1 = nh;  // survivors of stage 1
      EXP_DUP(DUP_CULL2) {
        int nh2 = 0;
        if (dup_) { WSYNC(); nmesh = 0; }
#pragma unroll 1
        for (int base = 0; base < nh_s1; base += 16) {
          const int idx = base + c;
          bool keep = false, mesh = false;
          int pk = 0;
          if (idx < nh_s1) {
            pk = reinterpret_cast<const int*>(L)[S16_NP_SCR + idx];
            const float* ta_ = L + S16_NP_SHP + S16_SHP * ((pk >> 16) & 0xFF);
            const float* tb_ = L + S16_NP_SHP + S16_SHP * ((pk >> 24) & 0xFF);
            keep = true;
            if ((int)(__float_as_uint(ta_[14]) & 7u) != SH_PLANE) {
              const m3 RA = qmat(q4{ta_[3], ta_[4], ta_[5], ta_[6]}), RB = qmat(q4{tb_[3], tb_[4], tb_[5], tb_[6]});
              const f3 d = f3{tb_[10], tb_[11], tb_[12]} - f3{ta_[10], ta_[11], ta_[12]};
              keep = !obb_separated(RA, f3{ta_[16], ta_[17], ta_[18]}, RB, f3{tb_[16], tb_[17], tb_[18]}, d, M.contact_offset);
            }
            // (a surviving (convex shape, mesh) pair goes to the mesh list of stage T: only its triangles in range take hit entries)
            mesh = TRI && keep && (int)(__float_as_uint(tb_[14]) & 7u) == SH_TRIMESH;
            keep = keep && !mesh;
          }
          if (TRI) {
            const unsigned mm = (unsigned)(__ballot(mesh) >> (16 * g)) & 0xFFFFu;
            const int rankm = nmesh + __popc(mm & ((1u << c) - 1u));
            if (mesh) {
              if (rankm < S16_MAX_HIT) reinterpret_cast<int*>(L)[S16_REC + rankm] = pk;
              else hit_over = true;
            }
            nmesh = min(nmesh + __popc(mm), S16_MAX_HIT);
          }
          const unsigned m16 = (unsigned)(__ballot(keep) >> (16 * g)) & 0xFFFFu;
          const int rank = nh2 + __popc(m16 & ((1u << c) - 1u));
          if (keep) {
            if (rank < S16_MAX_HIT) reinterpret_cast<int*>(L)[S16_NP_HIT + rank] = pk;
            else hit_over = true;
          }
          nh2 += __popc(m16);
        }
        nh = nh2 < S16_MAX_HIT ? nh2 : S16_MAX_HIT;
        if (!live) { nh = 0; nmesh = 0; }  // (a shadow group of the last env produces nothing: its pairs would touch that env's manifold cache twice)
      }
      }  // EXP_DUP
      WSYNC();
      if (__any(hit_over) && hit_over && live) atomicOr(&S.overflow[e], MSSIM_OVERFLOW_HITS);
      PH(23);
      // ---- classification: generic convex pairs (MPR) and box-box pairs of this env (byte lists of hit indices)
      int nml = 0;  // MPR pairs of this env
      int nbl = 0;  // box-box pairs of this env
      int npl = 0;  // (plane, hull) pairs of this env: by 16-lane groups as well (stage C)
      {
        unsigned char* const ml = reinterpret_cast<unsigned char*>(L + S16_NP_ML);
        unsigned char* const bl = reinterpret_cast<unsigned char*>(L + S16_NP_BL);
        unsigned char* const pll = reinterpret_cast<unsigned char*>(L + S16_NP_PL);
        bool over = false;
        for (int base = 0; base < nh; base += 16) {
          const int idx = base + c;
          bool is_mpr = false, is_bb = false, is_ph = false;
          if (idx < nh) {
            const int pk = reinterpret_cast<const int*>(L)[S16_NP_HIT + idx];
            const int ta = (int)(__float_as_uint(L[S16_NP_SHP + S16_SHP * ((pk >> 16) & 0xFF) + 14]) & 7u);
            const int tb = (int)(__float_as_uint(L[S16_NP_SHP + S16_SHP * ((pk >> 24) & 0xFF) + 14]) & 7u);
            is_bb = ta == SH_BOX && tb == SH_BOX;
            is_ph = ta == SH_PLANE && tb == SH_CONVEX;
            is_mpr = !(ta == SH_PLANE || is_bb || (TRI && tb == SH_TRIMESH));  // (mesh pairs: stage T)
            reinterpret_cast<int*>(L)[S16_NP_CNT + idx] = 0;
          }
          const unsigned m16 = (unsigned)(__ballot(is_mpr) >> (16 * g)) & 0xFFFFu;
          const int rank = nml + __popc(m16 & ((1u << c) - 1u));
          if (is_mpr) {
            if (rank < S16_MAX_MPR) ml[rank] = (unsigned char)idx;
            else over = true;
          }
          nml += __popc(m16);
          const unsigned b16 = (unsigned)(__ballot(is_bb) >> (16 * g)) & 0xFFFFu;
          if (is_bb) bl[nbl + __popc(b16 & ((1u << c) - 1u))] = (unsigned char)idx;
          nbl += __popc(b16);
          const unsigned p16 = (unsigned)(__ballot(is_ph) >> (16 * g)) & 0xFFFFu;
          if (is_ph) pll[npl + __popc(p16 & ((1u << c) - 1u))] = (unsigned char)idx;
          npl += __popc(p16);
        }
        if (__any(over) && over && live) atomicOr(&S.overflow[e], MSSIM_OVERFLOW_CONVEX);
        nml = nml < S16_MAX_MPR ? nml : S16_MAX_MPR;
      }
      // wave totals: all hits, MPR tasks, box-box tasks
      int cum[EPW + 1], mcum[EPW + 1], bcum[EPW + 1];
      cum[0] = mcum[0] = bcum[0] = 0;
#pragma unroll
      for (int j = 0; j < EPW; j++) {
        cum[j + 1] = cum[j] + __shfl(nh, GW * j);
        mcum[j + 1] = mcum[j] + __shfl(nml, GW * j);
        bcum[j + 1] = bcum[j] + __shfl(nbl, GW * j);
      }
      const int T = cum[EPW], TM = mcum[EPW], TB = bcum[EPW];
      // Box-box pairs: with few of them in the wave (the usual case: cube on table, peg on table) one lane per pair
      // leaves the wave almost empty for ~3400 instructions, so the 4 groups take them round-robin, 16 lanes per
      // pair (stage C). Many of them (fingers on the table: 8 per env): one lane per pair (stage A), whose clip
      // scratch is this area -- only possible while all tasks of the wave fit one round.
      const bool bb_lane = NR == 1 && TB > S16_MAX_BBC && T <= 64;  // wave-uniform (the clip scratch holds one column per lane of a 16-lane env)
      bool pool_over = false;
      WSYNC();
      PH_ADD(14, (TM + EPW - 1) / EPW);
      (void)mcum;
      PH_ADD(26, T);
      // a manifold goes to the staging tables of its env: normal, size, `cnt` points from the pool
      auto pool_alloc = [&](float* Lg, int cnt) __attribute__((always_inline)) -> int {
        int off = atomicAdd(reinterpret_cast<int*>(Lg + S16_NP_ALLOC), cnt);
        if (off + cnt > MSSIM_MAX_RAW_POINTS) { pool_over = true; off = -1; }
        return off;
      };
      // ---- stage A: plane pairs (and the box-box pairs when bb_lane), all such (env, pair) tasks of the wave
      // spread over the 64 lanes
      for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane64;
        const bool has = t < T;
        int ge = 0;
#pragma unroll
        for (int j = 1; j < EPW; j++) ge += (has && t >= cum[j]) ? 1 : 0;
        const int idx = has ? t - cum[ge] : 0;
        float* Lg = smw + ge * S16_ENV_FLOATS;
        manifold_t m;
        manifold_clear(m);
        bool mine = false;
        if (has) {
          const int pk = reinterpret_cast<const int*>(Lg)[S16_NP_HIT + idx];
          const shape_t A = shape_from_table(M, Lg + S16_NP_SHP + S16_SHP * ((pk >> 16) & 0xFF));
          const shape_t B = shape_from_table(M, Lg + S16_NP_SHP + S16_SHP * ((pk >> 24) & 0xFF));
          PH_ADD(27, __popcll(__ballot(A.type == SH_PLANE)));
          PH_ADD(28, __popcll(__ballot(A.type == SH_BOX && B.type == SH_BOX)));
          PH(10);
          const bool is_plane = A.type == SH_PLANE && B.type != SH_CONVEX, is_bb = bb_lane && A.type == SH_BOX && B.type == SH_BOX;  // ((plane, hull): stage C)
          if (is_plane) collide_plane(A, B, M.contact_offset, m);
          PH(11);
          if (is_bb) collide_box_box<16>(A, B, M.contact_offset, m, L + S16_NP_SCR + c);
          PH(12);
          mine = is_plane || is_bb;
        }
        WSYNC();  // the clip scratch is dead: the staging tables take its place
        if (t0 == 0) {
          if (c == 0 && lead) reinterpret_cast<int*>(L)[S16_NP_ALLOC] = 0;
          WSYNC();
        }
        if (mine && m.count > 0) {
          const int off = pool_alloc(Lg, m.count);
          if (off >= 0) {
            reinterpret_cast<int*>(Lg)[S16_NP_CNT + idx] = m.count;
            reinterpret_cast<int*>(Lg)[S16_NP_OFF + idx] = off;
            float* hn = Lg + S16_NP_HN + 3 * idx;
            hn[0] = m.n.x; hn[1] = m.n.y; hn[2] = m.n.z;
#pragma unroll
            for (int k = 0; k < 4; k++)
              if (k < m.count) *reinterpret_cast<float4*>(Lg + S16_NP_POOL + 4 * (off + k)) = float4{m.x[k].x, m.x[k].y, m.x[k].z, m.sep[k]};
          }
        }
      }
      if (T == 0) {  // (wave-uniform) no round ran: the pool counter is still to be cleared
        if (c == 0 && lead) reinterpret_cast<int*>(L)[S16_NP_ALLOC] = 0;
      }
      if (c == 0 && lead) { blk_nml[gb] = nml; blk_nbl[gb] = bb_lane ? 0 : nbl; blk_npl[gb] = npl; }
      BSYNC();  // every wave is done with its clip scratch: stage B may write manifolds into any env's staging tables
      PH(13);
      // ---- stage B: generic convex pairs through the persistent manifold cache (include/mssim.h MSSIM_PCM_*).
      // B0: every env's own group assigns cache slots to its pairs, in pair order (same pair -> its slot; else the
      // first empty slot; else the least recently used one not touched in this substep; else none: plain query).
      // Lane k holds the header word of slot k.
      BT_T0b;
      {
        int4 hd = int4{-1, 0, 0, 0};
        if (live && nml > 0) hd = *reinterpret_cast<const int4*>(S.pcm + ((size_t)e * MSSIM_PCM_SLOTS + c) * S16_PCM_LEN);  // (an env without such pairs leaves its cache alone)
        const int4 hd0 = hd;
        unsigned char* const slot_of = reinterpret_cast<unsigned char*>(L + S16_NP_SLOT);
        const unsigned char* const ml = reinterpret_cast<const unsigned char*>(L + S16_NP_ML);
        auto b16 = [&](bool v) __attribute__((always_inline)) { return (unsigned)(__ballot(v) >> (16 * g)) & 0xFFFFu; };
        for (int k = 0; k < nml; k++) {
          const int idx = ml[k];
          const int p = reinterpret_cast<const int*>(L)[S16_NP_HIT + idx] & 0xFFFF;
          unsigned m16 = b16(hd.x == p);
          bool fresh = false;
          if (m16 == 0u) {
            fresh = true;
            m16 = b16(hd.x < 0);
            if (m16 == 0u) {
              int st = hd.z < pcm_tick ? hd.z : 0x7FFFFFFF;
              int mn = st;
              mn = min(mn, __builtin_amdgcn_update_dpp(0, mn, 0x128, 0xF, 0xF, false));
              mn = min(mn, __builtin_amdgcn_update_dpp(0, mn, 0x124, 0xF, 0xF, false));
              mn = min(mn, __builtin_amdgcn_update_dpp(0, mn, 0x122, 0xF, 0xF, false));
              mn = min(mn, __builtin_amdgcn_update_dpp(0, mn, 0x121, 0xF, 0xF, false));
              m16 = mn != 0x7FFFFFFF ? b16(st == mn) : 0u;
            }
          }
          const int si = m16 ? (__ffs(m16) - 1) : -1;
          if (c == si) {
            if (fresh) { hd.x = p; hd.y = 0; hd.w = 0; }
            hd.z = pcm_tick;
          }
          if (c == 0) slot_of[idx] = si < 0 ? 255 : (unsigned char)(si | (fresh ? 0x80 : 0));
        }
        if (live && (hd.x != hd0.x || hd.y != hd0.y || hd.z != hd0.z || hd.w != hd0.w))
          *reinterpret_cast<int4*>(S.pcm + ((size_t)e * MSSIM_PCM_SLOTS + c) * S16_PCM_LEN) = hd;
      }
      __threadfence_block();  // (the headers are read back below by other groups of this block, through the CU's L1)
      BSYNC();
      // B: a pair is worked on by one 16-lane group (hull scans shared by its lanes); the pairs of all the block's envs
      // form one task list that all its groups take round-robin. Lanes 0..3 of the group hold the manifold's points.
      BT_T(17);
      int TMb = 0;
#pragma unroll
      for (int j = 0; j < BLK_ENVS; j++) TMb += blk_nml[j];
      for (int t0 = 0; t0 < TMb; t0 += BLK_GRPS) {
        const bool has = t0 + grp < TMb;
        const int t = has ? t0 + grp : t0;  // (idle groups shadow the round's first pair: valid shapes, no output)
        int ge = 0, k = t;
#pragma unroll
        for (int j = 0; j < BLK_ENVS - 1; j++) {
          const int nj = blk_nml[j];
          const bool past = ge == j && k >= nj;
          k -= past ? nj : 0;
          ge += past ? 1 : 0;
        }
        float* Lg = sm + ge * S16_ENV_FLOATS;
        const int idx = reinterpret_cast<const unsigned char*>(Lg + S16_NP_ML)[k];
        const int pk = reinterpret_cast<const int*>(Lg)[S16_NP_HIT + idx];
        const shape_t A = shape_from_table(M, Lg + S16_NP_SHP + S16_SHP * ((pk >> 16) & 0xFF));
        const shape_t B = shape_from_table(M, Lg + S16_NP_SHP + S16_SHP * ((pk >> 24) & 0xFF));
        SupCoop16 sup;
        sup.c = c;
        SupCoop16::load_one(sup.va, A, c);
        SupCoop16::load_one(sup.vb, B, c);
        const int sbyte = reinterpret_cast<const unsigned char*>(Lg + S16_NP_SLOT)[idx];
        const bool cached = has && sbyte != 255;
        const bool fresh = cached && (sbyte & 0x80);
        float* const slot = S.pcm + ((size_t)(chunk * BLK_ENVS + ge) * MSSIM_PCM_SLOTS + (cached ? (sbyte & 15) : 0)) * S16_PCM_LEN;
        auto b16 = [&](bool v) __attribute__((always_inline)) { return (unsigned)(__ballot(v) >> (16 * g)) & 0xFFFFu; };
        // ---- slot -> registers: header in every lane, point c in lane c < 4
        int npts = 0, grow = 0;
        bool queried_empty = false;
        f3 relp0 = f3{0, 0, 0}, nloc = f3{1, 0, 0};
        float relR0[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        f3 pA = f3{0, 0, 0}, pB = f3{0, 0, 0};
        float s0 = 0.f;
        if (cached && !fresh) {
          const int4 h4 = *reinterpret_cast<const int4*>(slot);
          npts = h4.y; grow = h4.w >> 8; queried_empty = h4.w & 1;
          const float4 r1 = *reinterpret_cast<const float4*>(slot + 4), r2 = *reinterpret_cast<const float4*>(slot + 8), r3 = *reinterpret_cast<const float4*>(slot + 12),
                       r4 = *reinterpret_cast<const float4*>(slot + 16);
          relp0 = f3{r1.x, r1.y, r1.z};
          relR0[0] = r2.x; relR0[1] = r2.y; relR0[2] = r2.z; relR0[3] = r2.w; relR0[4] = r3.x; relR0[5] = r3.y; relR0[6] = r3.z; relR0[7] = r3.w; relR0[8] = r4.x;
          nloc = f3{r4.y, r4.z, r4.w};
          if (c < 4) {
            const float* pp = slot + 20 + 7 * c;
            pA = f3{pp[0], pp[1], pp[2]}; pB = f3{pp[3], pp[4], pp[5]}; s0 = pp[6];
          }
        }
        BT_T(18);
        const float offset = M.contact_offset;
        // ---- refresh: the cached points move with their shapes; sideways drift or an open gap drops a point
        {
          const f3 nw = mmulv(A.rot, nloc);
          const f3 wA = A.c + mmulv(A.rot, pA), wB = B.c + mmulv(B.rot, pB);
          const f3 d = wA - wB;
          const float dn = dot(d, nw);
          const f3 tt = d - nw * dn;
          const bool ok = c < npts && !(dot(tt, tt) > MSSIM_PCM_DRIFT * MSSIM_PCM_DRIFT || s0 + dn > offset);
          const unsigned mk = b16(ok) & 15u;
          // compaction: lane k takes the k-th surviving point
          int src = 0;
          {
            unsigned r = mk;
            for (int q = 0; q < 4; q++) {
              const int bit = r ? (__ffs(r) - 1) : 0;
              if (q == c) src = bit;
              r &= r - 1u;
            }
          }
          pA = f3{gbc(pA.x, src), gbc(pA.y, src), gbc(pA.z, src)};
          pB = f3{gbc(pB.x, src), gbc(pB.y, src), gbc(pB.z, src)};
          s0 = gbc(s0, src);
          npts = __popc(mk);
        }
        bool moved = false;
        if (cached && !fresh) {
          const f3 dp = mtmulv(A.rot, B.c - A.c) - relp0;
          float tr = 0.f;
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) tr += relR0[3 * i + j] * dot(mcol(A.rot, i), mcol(B.rot, j));
          moved = dot(dp, dp) > MSSIM_PCM_MOVE * MSSIM_PCM_MOVE || tr < MSSIM_PCM_ROT_TRACE;
        }
        const float relR0_8 = relR0[8];
        // a point found by a query with shape A posed as (Rq, cq) joins the manifold
        auto merge = [&](const m3& Rq, f3 cq, f3 gx, float gsep) __attribute__((always_inline)) {
          const f3 nA = mtmulv(Rq, gx - cq), nB = mtmulv(B.rot, gx - B.c);
          const f3 dd = nA - pA;
          const unsigned dup = b16(c < npts && dot(dd, dd) < MSSIM_PCM_MERGE * MSSIM_PCM_MERGE) & 15u;
          int at = dup ? (__ffs(dup) - 1) : (npts < 4 ? npts : 4);
          if (!dup && npts < 4) npts++;
          if (c == at) { pA = nA; pB = nB; s0 = gsep; }  // (at == 4: lane 4 holds the fifth candidate)
          if (at < 4) return;
          // 5 candidates -> 4 by the patch selection rule, on lanes 0..4
          const f3 nw = mmulv(A.rot, nloc);
          const f3 wA = A.c + mmulv(A.rot, pA), wB = B.c + mmulv(B.rot, pB);
          const f3 x = (wA + wB) * 0.5f;
          const float sp = s0 + dot(wA - wB, nw);
          const bool in5 = c < 5;
          float best = in5 ? sp : 3e38f;
          best = -gmax16(-best);
          const int i0 = __ffs(b16(in5 && sp <= best + MSSIM_PATCH_TIE_SEP)) - 1;
          const f3 p0 = f3{gbc(x.x, i0), gbc(x.y, i0), gbc(x.z, i0)};
          auto first_near_max = [&](float v, bool allowed, float floor_) __attribute__((always_inline)) {
            const float mx = gmax16(allowed ? v : floor_);
            if (!(mx > floor_)) return -1;
            return __ffs(b16(allowed && v >= mx - MSSIM_PATCH_TIE_REL * mx)) - 1;
          };
          const f3 d0 = x - p0;
          const int i1 = first_near_max(dot(d0, d0), in5 && c != i0, -1.f);
          const f3 ed = f3{gbc(x.x, i1), gbc(x.y, i1), gbc(x.z, i1)} - p0;
          const float ar = dot(cross(ed, x - p0), nw);
          const int i2 = first_near_max(fabsf(ar), in5 && c != i0 && c != i1, -1.f);
          const float sgn2 = gbc(ar, i2);
          const int i3 = first_near_max(sgn2 >= 0.f ? -ar : ar, in5 && c != i0 && c != i1 && c != i2, MSSIM_PATCH_TIE_REL * fabsf(sgn2));  // (the 4th point has to add area)
          const unsigned keep = (1u << i0) | (1u << i1) | (1u << i2) | (i3 >= 0 ? 1u << i3 : 0u);
          int src = 0;
          {
            unsigned r = keep;
            for (int q = 0; q < 4; q++) {
              const int bit = r ? (__ffs(r) - 1) : 0;
              if (q == c) src = bit;
              r &= r - 1u;
            }
          }
          pA = f3{gbc(pA.x, src), gbc(pA.y, src), gbc(pA.z, src)};
          pB = f3{gbc(pB.x, src), gbc(pB.y, src), gbc(pB.z, src)};
          s0 = gbc(s0, src);
          npts = __popc(keep);
        };
        // ---- the query of this substep, if any: the full one (new pair, moved pair, or a manifold that lost all its
        // points), or a growth query for a young manifold short of a face contact whose pair is not moving
        const bool main_q = has && (!cached || fresh || moved || (npts == 0 && !queried_empty));
        const bool growing = cached && !main_q && npts > 0 && npts < 3 && grow > 0;
        if (growing) grow--;
        // (work counters of the debug builds: why this round's tasks query or not)
        BT_ADD(10, __popcll(__ballot(c == 0 && has && !cached)));
        BT_ADD(11, __popcll(__ballot(c == 0 && has && fresh)));
        BT_ADD(12, __popcll(__ballot(c == 0 && has && cached && !fresh && moved)));
        BT_ADD(13, __popcll(__ballot(c == 0 && has && cached && !fresh && !moved && main_q)));
        BT_ADD(15, __popcll(__ballot(c == 0 && has && growing)));
        BT_ADD(16, __popcll(__ballot(c == 0 && has && !main_q && !growing)));
        int mu_count = 0;  // the uncached pair's one-point result
        f3 mu_n = f3{0, 0, 0}, mu_x = f3{0, 0, 0};
        float mu_sep = 0.f;
        {
          shape_t Aq = A;
          if (growing) {
            // shape A tilted about the manifold: about a tangent through its single point, about the edge of its
            // first two points otherwise; the side alternates
            const f3 nw = mmulv(A.rot, nloc);
            const f3 wA = A.c + mmulv(A.rot, pA), wB = B.c + mmulv(B.rot, pB);
            const f3 xm = (wA + wB) * 0.5f;
            const f3 pivot = f3{gbc(xm.x, 0), gbc(xm.y, 0), gbc(xm.z, 0)};
            f3 axis;
            if (npts >= 2) axis = normalized(f3{gbc(xm.x, 1), gbc(xm.y, 1), gbc(xm.z, 1)} - pivot);
            else axis = fabsf(nw.x) < 0.57735f ? normalized(cross(nw, f3{1, 0, 0})) : normalized(cross(nw, f3{0, 1, 0}));
            if (grow & 1) axis = -axis;
            const m3 Rt = qmat(qaxis_angle(axis, MSSIM_PCM_TILT));
#pragma unroll
            for (int j = 0; j < 3; j++) {
              const f3 col = mmulv(Rt, mcol(A.rot, j));
              Aq.rot.m[0][j] = col.x; Aq.rot.m[1][j] = col.y; Aq.rot.m[2][j] = col.z;
            }
            Aq.c = pivot + mmulv(Rt, A.c - pivot);
          }
          manifold_t gq;
          manifold_clear(gq);
          BT_T(19);
          if (main_q || growing) collide_mpr_t(Aq, B, offset, gq, sup);
          BT_T(20);
          if (main_q) {
            if (!cached) {
              mu_count = gq.count; mu_n = gq.n; mu_x = gq.x[0]; mu_sep = gq.sep[0];
            } else {
              queried_empty = gq.count == 0;
              if (gq.count == 0) {
                npts = 0;
              } else {
                const f3 nnew = mtmulv(A.rot, gq.n);
                if (npts > 0 && dot(nnew, nloc) < MSSIM_PATCH_COS) npts = 0;  // the contact turned: start over
                if (npts == 0) grow = MSSIM_PCM_GROW;  // a manifold starts: growth queries are owed to it
                nloc = nnew;
                merge(A.rot, A.c, gq.x[0], gq.sep[0]);
              }
            }
          } else if (growing && gq.count > 0) {
            merge(Aq.rot, Aq.c, gq.x[0], gq.sep[0]);
          }
        }
        // ---- the manifold as it stands -> staging tables of its env; the slot goes back to the cache
        BT_T(21);
        if (has) {
          const f3 nw = cached ? mmulv(A.rot, nloc) : mu_n;
          const f3 wA = A.c + mmulv(A.rot, pA), wB = B.c + mmulv(B.rot, pB);
          const int cnt = cached ? npts : mu_count;
          const float4 P = cached ? float4{0.5f * (wA.x + wB.x), 0.5f * (wA.y + wB.y), 0.5f * (wA.z + wB.z), s0 + dot(wA - wB, nw)}
                                  : float4{mu_x.x, mu_x.y, mu_x.z, mu_sep};
          if (cnt > 0) {
            int off = 0;
            if (c == 0) off = pool_alloc(Lg, cnt);
            off = gbci(off, 0);
            if (off >= 0) {
              if (c == 0) {
                reinterpret_cast<int*>(Lg)[S16_NP_CNT + idx] = cnt;
                reinterpret_cast<int*>(Lg)[S16_NP_OFF + idx] = off;
                float* hn = Lg + S16_NP_HN + 3 * idx;
                hn[0] = nw.x; hn[1] = nw.y; hn[2] = nw.z;
              }
              if (c < cnt) *reinterpret_cast<float4*>(Lg + S16_NP_POOL + 4 * (off + c)) = P;
            }
          }
          if (cached) {
            if (c == 0) {
              int4 h4 = *reinterpret_cast<const int4*>(slot);
              h4.y = npts; h4.w = (grow << 8) | (queried_empty ? 1 : 0);
              *reinterpret_cast<int4*>(slot) = h4;
              float r8 = relR0_8;
              if (main_q) {  // (the pose the last full query saw)
                const f3 relp = mtmulv(A.rot, B.c - A.c);
                float relR[9];
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                  for (int j = 0; j < 3; j++) relR[3 * i + j] = dot(mcol(A.rot, i), mcol(B.rot, j));
                *reinterpret_cast<float4*>(slot + 4) = float4{relp.x, relp.y, relp.z, 0.f};
                *reinterpret_cast<float4*>(slot + 8) = float4{relR[0], relR[1], relR[2], relR[3]};
                *reinterpret_cast<float4*>(slot + 12) = float4{relR[4], relR[5], relR[6], relR[7]};
                r8 = relR[8];
              }
              *reinterpret_cast<float4*>(slot + 16) = float4{r8, nloc.x, nloc.y, nloc.z};
            }
            if (c < npts) {
              float* pp = slot + 20 + 7 * c;
              pp[0] = pA.x; pp[1] = pA.y; pp[2] = pA.z; pp[3] = pB.x; pp[4] = pB.y; pp[5] = pB.z; pp[6] = s0;
            }
          }
        }
      }
      // ---- stage T (models with triangle meshes): every (convex shape, mesh) pair of an env becomes one task per triangle
      // in range; a task gives a manifold of its own -- a new entry of the env's hit list -- which the patch pass merges with
      // those of the coplanar neighbours (include/mssim.h MSSIM_SHAPE_TRIMESH).
      if (TRI) {
        PH(25);
        // T0, the env's own group: BVH traversal. A node is 16 child boxes, one per lane; children in range of the convex
        // shape's bounding sphere (+ contact offset, mesh frame) are pushed (nodes) or collected (triangles). The triangles
        // are then ranked by index (the order of the oracle's plain loop): rank r becomes hit nh + r and task ntask + r.
        int* const tl = reinterpret_cast<int*>(L + S16_NP_BSCR);         // [MSSIM_MAX_TRI_TASKS] tasks: triangle | hit index << 24 (BSCR + KEEP, both idle here)
        int* const cand = reinterpret_cast<int*>(L + S16_NP_BSCR) + MSSIM_MAX_TRI_TASKS;  // [32] triangles of the pair being traversed
        int* const stack = reinterpret_cast<int*>(L + S16_NP_BOUT);      // [20] nodes to visit
        int* const hitw = reinterpret_cast<int*>(L) + S16_NP_HIT;
        int* const cntw = reinterpret_cast<int*>(L) + S16_NP_CNT;
        auto b16 = [&](bool v) __attribute__((always_inline)) { return (unsigned)(__ballot(v) >> (16 * g)) & 0xFFFFu; };
        int ntask = 0;
        bool tri_over = false;
        const int nh0 = nh;
        // Clearance of a (convex shape, mesh) pair: its last full traversal left the centre of the shape's bounding sphere
        // (mesh frame) and the distance that centre may move before any triangle's box can come into range. While it has not,
        // the pair is not traversed at all -- an arm in the middle of a room has 40 such pairs and is near nothing. The
        // result is that of a traversal (every triangle lies in a box the traversal rejected by the same sphere test).
        const int* const meshl = reinterpret_cast<const int*>(L) + S16_REC;  // the env's mesh pairs (from the cull)
        unsigned long long clear_mask = 0ull;  // bit i: mesh pair i is known to be out of range
        for (int base = 0; base < nmesh; base += 16) {
          const int idx = base + c;
          bool clear_ = false;
          if (idx < nmesh) {
            const int pk = meshl[idx];
            const float* tb_ = L + S16_NP_SHP + S16_SHP * ((pk >> 24) & 0xFF);
            const int slot = M.pair_mesh_slot[pk & 0xFFFF];
            if (slot >= 0) {
              const float* ta_ = L + S16_NP_SHP + S16_SHP * ((pk >> 16) & 0xFF);
              const f3 cq = mtmulv(qmat(q4{tb_[3], tb_[4], tb_[5], tb_[6]}), f3{ta_[10], ta_[11], ta_[12]} - f3{tb_[0], tb_[1], tb_[2]});
              const float* cl = S.tri_clear + (size_t)(4 * slot) * N + e;
              const f3 dd = cq - f3{cl[0], cl[(size_t)N], cl[2 * (size_t)N]};
              const float slack = cl[3 * (size_t)N];
              clear_ = slack > 0.f && dot(dd, dd) < slack * slack;  // (an unset entry is a NaN: false)
            }
          }
          clear_mask |= (unsigned long long)b16(clear_) << base;
        }
        int nc_node[2] = {-1, -1}, nc_ref[2] = {0, 0}, nc_victim = 0;  // (node cache of the traversal, see below)
        float nc_box[2][6];
#pragma unroll
        for (int w = 0; w < 2; w++)
#pragma unroll
          for (int k = 0; k < 6; k++) nc_box[w][k] = 0.f;
        // (the search range: the contact offset, narrowed while the triangles found do not fit -- MSSIM_TRI_RANGE_STEPS)
        float range = M.contact_offset;
#pragma unroll 1
        for (int step = 0; step < MSSIM_TRI_RANGE_STEPS; step++) {
        nh = nh0; ntask = 0; tri_over = false;
        for (int idx = 0; idx < nmesh; idx++) {  // (group-uniform)
          const int pk = meshl[idx];
          const float* tb_ = L + S16_NP_SHP + S16_SHP * ((pk >> 24) & 0xFF);
          if ((clear_mask >> idx) & 1ull) continue;
          const float* ta_ = L + S16_NP_SHP + S16_SHP * ((pk >> 16) & 0xFF);
          const m3 RB = qmat(q4{tb_[3], tb_[4], tb_[5], tb_[6]});
          const f3 cq = mtmulv(RB, f3{ta_[10], ta_[11], ta_[12]} - f3{tb_[0], tb_[1], tb_[2]});
          const float rq = ta_[13] + range;
          // (the convex shape's oriented box in the mesh frame: axes scaled by the half extents, and its bounds)
          const m3 RA = qmat(q4{ta_[3], ta_[4], ta_[5], ta_[6]});
          const f3 un0 = mtmulv(RB, mcol(RA, 0)), un1 = mtmulv(RB, mcol(RA, 1)), un2 = mtmulv(RB, mcol(RA, 2));
          const f3 ax0 = un0 * ta_[16], ax1 = un1 * ta_[17], ax2 = un2 * ta_[18];
          const f3 ext = f3{fabsf(ax0.x) + fabsf(ax1.x) + fabsf(ax2.x), fabsf(ax0.y) + fabsf(ax1.y) + fabsf(ax2.y), fabsf(ax0.z) + fabsf(ax1.z) + fabsf(ax2.z)};
          const float off_ = range;
          int sp = 1, ncand = 0;
          float near2 = 3e38f;  // squared distance from cq to the nearest box that was examined and not descended into (this lane's share)
          if (c == 0) stack[0] = (int)(__float_as_uint(tb_[14]) >> 15);
          WSYNC();
          while (sp > 0) {
            const int node = stack[sp - 1];
            sp--;
            WSYNC();
            // (this lane's child box and reference; the two nodes visited last are kept in registers: the pairs of an env mostly
            // walk the same few nodes -- the roots of its one or two meshes --, and a node is an L2 round trip away)
            float lx, ly, lz, hx, hy, hz;
            int ref;
            if (node == nc_node[0] || node == nc_node[1]) {  // (group-uniform)
              const int w = node == nc_node[0] ? 0 : 1;
              lx = nc_box[w][0]; ly = nc_box[w][1]; lz = nc_box[w][2]; hx = nc_box[w][3]; hy = nc_box[w][4]; hz = nc_box[w][5];
              ref = nc_ref[w];
              nc_victim = 1 - w;
            } else {
              const float* nd = M.tri_bvh + (size_t)node * 112;
              lx = nd[6 * c]; ly = nd[6 * c + 1]; lz = nd[6 * c + 2]; hx = nd[6 * c + 3]; hy = nd[6 * c + 4]; hz = nd[6 * c + 5];
              ref = __float_as_int(nd[96 + c]);
#pragma unroll
              for (int w = 0; w < 2; w++)
                if (w == nc_victim) { nc_node[w] = node; nc_box[w][0] = lx; nc_box[w][1] = ly; nc_box[w][2] = lz; nc_box[w][3] = hx; nc_box[w][4] = hy; nc_box[w][5] = hz; nc_ref[w] = ref; }
              nc_victim = 1 - nc_victim;
            }
            const float dx = fmaxf(fmaxf(lx - cq.x, cq.x - hx), 0.f), dy = fmaxf(fmaxf(ly - cq.y, cq.y - hy), 0.f), dz = fmaxf(fmaxf(lz - cq.z, cq.z - hz), 0.f);
            bool in = lx <= hx && dx * dx + dy * dy + dz * dz <= rq * rq;
            in = in && !(lx - (cq.x + ext.x) > off_ || (cq.x - ext.x) - hx > off_ || ly - (cq.y + ext.y) > off_ || (cq.y - ext.y) - hy > off_ ||
                         lz - (cq.z + ext.z) > off_ || (cq.z - ext.z) - hz > off_);
            if (in && ref >= 0) {  // a node: the child's box along the oriented box's own axes
              const f3 bc = f3{0.5f * (lx + hx), 0.5f * (ly + hy), 0.5f * (lz + hz)} - cq, bh = f3{0.5f * (hx - lx), 0.5f * (hy - ly), 0.5f * (hz - lz)};
              auto apart = [&](f3 u, float h) __attribute__((always_inline)) {
                return fabsf(dot(u, bc)) - (fabsf(u.x) * bh.x + fabsf(u.y) * bh.y + fabsf(u.z) * bh.z) > h + off_;
              };
              in = !(apart(un0, ta_[16]) || apart(un1, ta_[17]) || apart(un2, ta_[18]));
            }
            if (in && ref < 0) {  // a triangle: its corners along the oriented box's own axes, then the box against its plane
              const float* tq = M.tri_soup + (size_t)(~ref) * 12;
              const f3 cen = f3{tq[0], tq[1], tq[2]} - cq;
              const f3 k0 = cen + f3{tq[3], tq[4], tq[5]}, k1 = cen + f3{tq[6], tq[7], tq[8]}, k2 = cen + f3{tq[9], tq[10], tq[11]};
              auto apart = [&](f3 u, float h) __attribute__((always_inline)) {
                const float d0 = dot(u, k0), d1 = dot(u, k1), d2 = dot(u, k2);
                return fminf(d0, fminf(d1, d2)) > h + off_ || fmaxf(d0, fmaxf(d1, d2)) < -(h + off_);
              };
              in = !(apart(un0, ta_[16]) || apart(un1, ta_[17]) || apart(un2, ta_[18]));
            }
            if (in && ref < 0) {
              const float* tq = M.tri_soup + (size_t)(~ref) * 12;
              const f3 v0 = f3{tq[3], tq[4], tq[5]}, nn = cross(f3{tq[6], tq[7], tq[8]} - v0, f3{tq[9], tq[10], tq[11]} - v0);
              const float len = sqrtf(dot(nn, nn));
              if (len > 0.f) {
                const float dist = dot(nn, cq - (f3{tq[0], tq[1], tq[2]} + v0));
                const float rad = fabsf(dot(nn, ax0)) + fabsf(dot(nn, ax1)) + fabsf(dot(nn, ax2));
                in = !(fabsf(dist) - rad > off_ * len);
              }
            }
            const unsigned mn = b16(in && ref >= 0), ml2 = b16(in && ref < 0), lt = (1u << c) - 1u;
            bool descend = false;
            if (in && ref >= 0) {
              const int at = sp + __popc(mn & lt);
              if (at < 20) { stack[at] = ref; descend = true; } else tri_over = true;
            }
            if (lx <= hx && !descend) near2 = fminf(near2, dx * dx + dy * dy + dz * dz);  // (its triangles are all inside this box)
            if (in && ref < 0) {
              const int at = ncand + __popc(ml2 & lt);
              if (at < MSSIM_MAX_TRI_HITS) cand[at] = ~ref; else tri_over = true;
            }
            sp = min(sp + __popc(mn), 20);
            ncand = min(ncand + __popc(ml2), MSSIM_MAX_TRI_HITS);
            WSYNC();
          }
          if (step == 0) {  // (full range: what this traversal says about the pair's clearance)
            const int slot = M.pair_mesh_slot[pk & 0xFFFF];
            const float slack = fminf(sqrt_f(-gmax16(-near2)) - rq, 1e6f);
            if (c == 0 && live && slot >= 0) {
              float* cl = S.tri_clear + (size_t)(4 * slot) * N + e;
              cl[0] = cq.x; cl[(size_t)N] = cq.y; cl[2 * (size_t)N] = cq.z; cl[3 * (size_t)N] = slack;
            }
          }
          // rank by triangle index; room in the hit list and the task list permitting
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const int i = c + 16 * k;
            if (i < ncand) {
              const int t = cand[i];
              int rank = 0;
              for (int j = 0; j < ncand; j++) rank += cand[j] < t ? 1 : 0;
              if (nh + rank < S16_MAX_HIT && ntask + rank < MSSIM_MAX_TRI_TASKS) {
                tl[ntask + rank] = t | ((nh + rank) << 24);
                hitw[nh + rank] = pk;
                cntw[nh + rank] = 0;
              } else {
                tri_over = true;
              }
            }
          }
          const int room = min(S16_MAX_HIT - nh, MSSIM_MAX_TRI_TASKS - ntask);
          const int take = min(ncand, room);
          nh += take; ntask += take;
          WSYNC();
        }
        tri_over = b16(tri_over) != 0u;  // (group-uniform from here)
        if (!tri_over) break;
        range = step == MSSIM_TRI_RANGE_STEPS - 2 ? 0.f : 0.5f * range;
        }
        if (tri_over && live && c == 0) atomicOr(&S.overflow[e], MSSIM_OVERFLOW_TRI);
        // T1: the triangle tasks of all the block's envs form one list that its 16 groups take round-robin, like the
        // generic-convex pairs of stage B (an env with the whole arm on a mesh has 40-56 of them, its neighbours none)
        if (c == 0 && lead) blk_ntl[gb] = ntask;
        BSYNC();  // every env's task list is complete
        int TT = 0;
#pragma unroll
        for (int j = 0; j < BLK_ENVS; j++) TT += blk_ntl[j];
        for (int t0 = 0; t0 < TT; t0 += BLK_GRPS) {
          const bool has = t0 + grp < TT;
          const int t = has ? t0 + grp : t0;
          int ge = 0, kt = t;
#pragma unroll
          for (int j = 0; j < BLK_ENVS - 1; j++) {
            const int nj = blk_ntl[j];
            const bool past = ge == j && kt >= nj;
            kt -= past ? nj : 0;
            ge += past ? 1 : 0;
          }
          float* Lg = sm + ge * S16_ENV_FLOATS;
          const int tw = reinterpret_cast<const int*>(Lg + S16_NP_BSCR)[kt];
          const int tri = tw & 0xFFFFFF, idx = tw >> 24;
          const int pk = reinterpret_cast<const int*>(Lg)[S16_NP_HIT + idx];
          const shape_t A = shape_from_table(M, Lg + S16_NP_SHP + S16_SHP * ((pk >> 16) & 0xFF));
          const float* tb_ = Lg + S16_NP_SHP + S16_SHP * ((pk >> 24) & 0xFF);
          const float* tq = M.tri_soup + (size_t)tri * 12;
          shape_t Tr;
          Tr.type = SH_CONVEX;
          Tr.rot = qmat(q4{tb_[3], tb_[4], tb_[5], tb_[6]});
          Tr.c = f3{tb_[0], tb_[1], tb_[2]} + mmulv(Tr.rot, f3{tq[0], tq[1], tq[2]});
          Tr.p0 = Tr.p1 = Tr.p2 = 0.f;
          Tr.verts = tq + 3;
          Tr.nverts = 3;
          SupCoop16 sup;
          sup.c = c;
          SupCoop16::load_one(sup.va, A, c);
          SupCoop16::load_one(sup.vb, Tr, c);
          const float offset = M.contact_offset;
          const f3 q0 = Tr.c + mmulv(Tr.rot, f3{tq[3], tq[4], tq[5]}), q1 = Tr.c + mmulv(Tr.rot, f3{tq[6], tq[7], tq[8]}), q2 = Tr.c + mmulv(Tr.rot, f3{tq[9], tq[10], tq[11]});
          f3 nf = normalized(cross(q1 - q0, q2 - q0));
          if (dot(nf, A.c - q0) < 0.f) nf = -nf;  // the side A's centre is on
          const f3 e0 = q1 - q0, e1 = q2 - q0;
          const float d00 = dot(e0, e0), d01 = dot(e0, e1), d11 = dot(e1, e1);
          const float den = d00 * d11 - d01 * d01, tol = MSSIM_PCM_MERGE * rsq_f(fminf(d00, d11));
          // (box A) where the line o + t nf enters the box, in the box frame against its three slabs: the gap between a point
          // o of the triangle's plane and the box above it; false: the line misses the box or the gap is beyond the offset
          auto box_above = [&](f3 o, float& t_in) __attribute__((always_inline)) {
            const f3 dl = mtmulv(A.rot, nf), ol = mtmulv(A.rot, o - A.c);
            float t_out = 1e30f;
            bool miss = false;
            t_in = -1e30f;
#pragma unroll
            for (int a = 0; a < 3; a++) {
              const float hb = a == 0 ? A.p0 : (a == 1 ? A.p1 : A.p2), ov = comp(ol, a), dv = comp(dl, a);
              const bool par = fabsf(dv) < 1e-9f;
              miss = miss || (par && fabsf(ov) > hb);
              const float t0 = (-hb - ov) / dv, t1 = (hb - ov) / dv;
              t_in = par ? t_in : fmaxf(t_in, fminf(t0, t1));
              t_out = par ? t_out : fminf(t_out, fmaxf(t0, t1));
            }
            return !miss && t_in <= t_out && t_in < offset;
          };
          // candidates of this lane, at most 4 (hull vertices c, c + 16, ..); id = their place in the oracle's order:
          // (1) A's plane-contact points over the triangle, 0..63; (2) the triangle's corners under a box, 64..66; (3) the
          // triangle's edges under a box, 67..75
          f3 X[4];
          float Sp[4];
          bool ok[4];
          float s_low = 3e38f;  // gap of A's lowest plane-contact point (this lane's share; reduced over the group below)
#pragma unroll
          for (int k = 0; k < 4; k++) { X[k] = f3{0, 0, 0}; Sp[k] = 3e38f; ok[k] = false; }
          auto cand1 = [&](int k, f3 p, float radius, bool exists) __attribute__((always_inline)) {
            const float sgap = dot(nf, p - q0) - radius;
            const f3 dd = p - nf * (radius + sgap) - q0;
            const float d20 = dot(dd, e0), d21 = dot(dd, e1);
            const float v = (d11 * d20 - d01 * d21) / den, w = (d00 * d21 - d01 * d20) / den;
            ok[k] = exists && sgap < offset && !(v < -tol || w < -tol || v + w > 1.f + tol);
            X[k] = p - nf * (radius + 0.5f * sgap);
            Sp[k] = sgap;
            s_low = fminf(s_low, exists ? sgap : 3e38f);
          };
          if (A.type == SH_BOX) {
            cand1(0, A.c + mmulv(A.rot, f3{(c & 1) ? A.p0 : -A.p0, (c & 2) ? A.p1 : -A.p1, (c & 4) ? A.p2 : -A.p2}), 0.f, c < 8);
            // the triangle's corner (c - 8) under the box
            const f3 qc = c == 8 ? q0 : (c == 9 ? q1 : q2);
            float t_in;
            const bool above = box_above(qc, t_in);
            if (c >= 8 && c < 11) {
              ok[1] = above;
              X[1] = qc + nf * (0.5f * t_in);
              Sp[1] = t_in;
            }
            // (3) lanes 11..13, the triangle's edge (c - 11) under the box: entry and exit of the box's shadow along nf and the
            // nearest point in between (see the oracle's tri_manifold): slots 1..3
            {
              const int ei = c == 12 ? 1 : (c == 13 ? 2 : 0);
              const f3 qa = ei == 0 ? q0 : (ei == 1 ? q1 : q2), qb = ei == 0 ? q1 : (ei == 1 ? q2 : q0);
              const f3 dl = mtmulv(A.rot, nf), Pv = mtmulv(A.rot, qa - A.c), Qv = mtmulv(A.rot, qb - qa);
              float s0 = 0.f, s1 = 1.f, al[3], be[3], wid[3];
              bool par[3], empty = false;
              auto clip = [&](float a0, float b0) __attribute__((always_inline)) {  // a0 + b0 s <= 0
                if (b0 > 0.f) s1 = fminf(s1, -a0 / b0);
                else if (b0 < 0.f) s0 = fmaxf(s0, -a0 / b0);
                else if (a0 > 0.f) empty = true;
              };
#pragma unroll
              for (int a = 0; a < 3; a++) {
                const float hb = a == 0 ? A.p0 : (a == 1 ? A.p1 : A.p2), dv = comp(dl, a), P = comp(Pv, a), Q = comp(Qv, a);
                par[a] = fabsf(dv) < 1e-9f;
                if (par[a]) {
                  clip(P - hb, Q);
                  clip(-P - hb, -Q);
                  al[a] = be[a] = wid[a] = 0.f;
                } else {
                  al[a] = (-(dv > 0.f ? hb : -hb) - P) / dv;
                  be[a] = -Q / dv;
                  wid[a] = 2.f * hb / fabsf(dv);
                }
              }
#pragma unroll
              for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++)
                  if (a != b && !par[a] && !par[b]) clip(al[a] - al[b] - wid[b], be[a] - be[b]);
              auto t_in_at = [&](float sv) __attribute__((always_inline)) {
                float t = -1e30f;
#pragma unroll
                for (int a = 0; a < 3; a++) t = par[a] ? t : fmaxf(t, al[a] + be[a] * sv);
                return t;
              };
              const float f0 = t_in_at(s0), f1 = t_in_at(s1);
              float sm = s0, fm = fminf(f0, f1);
              bool inner = false;
#pragma unroll
              for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = a + 1; b < 3; b++) {
                  if (par[a] || par[b] || be[a] == be[b]) continue;
                  const float sx = (al[b] - al[a]) / (be[a] - be[b]);
                  if (!(sx > s0 && sx < s1)) continue;
                  const float fx = t_in_at(sx);
                  if (fx < fm - 1e-6f) { fm = fx; sm = sx; inner = true; }
                }
              if (c >= 11 && c < 14 && !empty && s0 <= s1) {
                ok[1] = s0 > 0.f && f0 < offset;
                X[1] = qa + (qb - qa) * s0 + nf * (0.5f * f0); Sp[1] = f0;
                ok[2] = s1 < 1.f && s1 > s0 && f1 < offset;
                X[2] = qa + (qb - qa) * s1 + nf * (0.5f * f1); Sp[2] = f1;
                ok[3] = inner && fm < offset;
                X[3] = qa + (qb - qa) * sm + nf * (0.5f * fm); Sp[3] = fm;
              }
            }
          } else if (A.type == SH_SPHERE) {
            cand1(0, A.c, A.p0, c == 0);
          } else if (A.type == SH_CAPSULE) {
            const f3 ax = mcol(A.rot, 0) * A.p1;
            cand1(0, c == 0 ? A.c - ax : A.c + ax, A.p0, c < 2);
          } else if (A.type == SH_CONVEX) {
#pragma unroll
            for (int k = 0; k < 4; k++) cand1(k, A.c + mmulv(A.rot, sup.va[k]), 0.f, c + 16 * k < A.nverts);
          } else {
            cand1(0, support(A, -nf), 0.f, c == 0);
          }
          // points much higher above the plane than A's lowest one give no contact (MSSIM_TRI_SLACK, see the oracle); the
          // triangle's corners under a box (slot 1 of lanes 8..10) are held to the same bound
          s_low = -gmax16(-s_low);
#pragma unroll
          for (int k = 0; k < 4; k++) ok[k] = ok[k] && Sp[k] <= s_low + MSSIM_TRI_SLACK;
          // candidate ids: (1) box corner / capsule end / hull vertex index; (2) 64 + corner
          auto cid = [&](int k) __attribute__((always_inline)) {
            if (A.type == SH_BOX && k >= 1) return c < 11 ? 64 + (c - 8) : 67 + 3 * (c - 11) + (k - 1);
            return c + 16 * k;
          };
          // the 4 deepest, the lowest id among equals first (the oracle's rule in tri_manifold)
          f3 mx[4];
          float msep[4];
          int cnt = 0;
#pragma unroll 1
          for (int r = 0; r < 4; r++) {
            float best = 3e38f;
#pragma unroll
            for (int k = 0; k < 4; k++) best = fminf(best, ok[k] ? Sp[k] : 3e38f);
            best = -gmax16(-best);
            if (!(best < 3e38f)) break;  // (group-uniform)
            int id = 1 << 20;
#pragma unroll
            for (int k = 3; k >= 0; k--) id = (ok[k] && Sp[k] <= best + MSSIM_TRI_TIE) ? min(id, cid(k)) : id;  // (gaps within MSSIM_TRI_TIE are equal)
            id = gmin16i(id);
            // the owner lane and slot of that id
            const int owner = id >= 67 ? 11 + (id - 67) / 3 : (id >= 64 ? 8 + (id - 64) : (id & 15));
            const int slot = id >= 67 ? 1 + (id - 67) % 3 : (id >= 64 ? 1 : (id >> 4));
            const f3 mine = slot == 0 ? X[0] : (slot == 1 ? X[1] : (slot == 2 ? X[2] : X[3]));
            const f3 px = f3{gbc(mine.x, owner), gbc(mine.y, owner), gbc(mine.z, owner)};
            const float psep = gbc(slot == 0 ? Sp[0] : (slot == 1 ? Sp[1] : (slot == 2 ? Sp[2] : Sp[3])), owner);
#pragma unroll
            for (int t2 = 0; t2 < 4; t2++)
              if (t2 == cnt) { mx[t2] = px; msep[t2] = psep; }
            if (c == owner) {
#pragma unroll
              for (int k = 0; k < 4; k++) ok[k] = ok[k] && k != slot;
            }
            cnt++;
          }
          f3 nrm = nf;
          // (no part of A is nearer to the triangle than its lowest point is to the triangle's plane: beyond the offset, no query)
          const bool ask = has && cnt == 0 && s_low < offset;
          if (__any(ask)) {
            // nothing over or under the triangle: the generic query (from the side, or a round shape next to / across it)
            f3 inside;
            {
              float w[3];
              closest_on_triangle(q0 - A.c, q1 - A.c, q2 - A.c, w);
              inside = q0 * w[0] + q1 * w[1] + q2 * w[2];
            }
            manifold_t gq;
            manifold_clear(gq);
            if (ask) collide_mpr_t(A, Tr, offset, gq, sup, true, inside);
            if (ask && gq.count > 0 && !(gq.sep[0] < s_low - 1e-3f)) {  // (an answer far below A's lowest gap over the plane: a ray out of the side, see the oracle)
              const bool face = dot(nf, gq.n) > 0.5f;
              nrm = face ? nf : gq.n;
              mx[0] = gq.x[0]; msep[0] = gq.sep[0];
              cnt = 1;
              if (face && A.type == SH_BOX) {
                // a box face across an edge of the mesh: the query's point is approximate; its foot on the triangle's plane,
                // if in the triangle, is measured against the box exactly (see the oracle)
                const f3 foot = gq.x[0] - nf * dot(gq.x[0] - q0, nf), dd = foot - q0;
                const float d20 = dot(dd, e0), d21 = dot(dd, e1);
                const float v = (d11 * d20 - d01 * d21) / den, w = (d00 * d21 - d01 * d20) / den;
                float t_in;
                const bool above = box_above(foot, t_in);
                if (v < -tol || w < -tol || v + w > 1.f + tol || !above) cnt = 0;
                mx[0] = foot + nf * (0.5f * t_in); msep[0] = t_in;
              }
            }
          }
          if (has && cnt > 0) {
            int off = 0;
            if (c == 0) off = pool_alloc(Lg, cnt);
            off = gbci(off, 0);
            if (off >= 0) {
              if (c == 0) {
                reinterpret_cast<int*>(Lg)[S16_NP_CNT + idx] = cnt;
                reinterpret_cast<int*>(Lg)[S16_NP_OFF + idx] = off;
                float* hn = Lg + S16_NP_HN + 3 * idx;
                hn[0] = nrm.x; hn[1] = nrm.y; hn[2] = nrm.z;
              }
              if (c < cnt) {
                const f3 x = c == 0 ? mx[0] : (c == 1 ? mx[1] : (c == 2 ? mx[2] : mx[3]));
                const float sp = c == 0 ? msep[0] : (c == 1 ? msep[1] : (c == 2 ? msep[2] : msep[3]));
                *reinterpret_cast<float4*>(Lg + S16_NP_POOL + 4 * (off + c)) = float4{x.x, x.y, x.z, sp};
              }
            }
          }
        }
        BSYNC();  // the task lists lie in scratch that stage C uses next; manifolds written for other waves' envs are in place
        PH(31);
      }
      BT_T(22);
      PH(25);
      // ---- stage C: box-box pairs by 16-lane groups, round-robin over the block's list like stage B (the pairs of a wave
      // that has more than 16 of them were done one per lane in stage A); behind them in the same list the (plane, hull)
      // pairs: the hull's vertices over the lanes, the choice of its four points by group reductions
      {
        int TBb = 0, TPb = 0;
#pragma unroll
        for (int j = 0; j < BLK_ENVS; j++) { TBb += blk_nbl[j]; TPb += blk_npl[j]; }
#ifdef EXP_NO_STAGE_C
        TBb = 0; TPb = 0;
#endif
        for (int t = grp; t < TBb + TPb; t += BLK_GRPS) {
          const bool ph = t >= TBb;  // (group-uniform) a (plane, hull) pair
          int ge = 0, k = ph ? t - TBb : t;
#pragma unroll
          for (int j = 0; j < BLK_ENVS - 1; j++) {
            const int nj = ph ? blk_npl[j] : blk_nbl[j];
            const bool past = ge == j && k >= nj;
            k -= past ? nj : 0;
            ge += past ? 1 : 0;
          }
          float* Lg = sm + ge * S16_ENV_FLOATS;
          const int idx = ph ? reinterpret_cast<const unsigned char*>(Lg + S16_NP_PL)[k] : reinterpret_cast<const unsigned char*>(Lg + S16_NP_BL)[k];
          const int pk = reinterpret_cast<const int*>(Lg)[S16_NP_HIT + idx];
          const shape_t A = shape_from_table(M, Lg + S16_NP_SHP + S16_SHP * ((pk >> 16) & 0xFF));
          const shape_t B = shape_from_table(M, Lg + S16_NP_SHP + S16_SHP * ((pk >> 24) & 0xFF));
          // (scratch of this GROUP: with two rows the env's other row borrows the body-pair keys' words, idle until the patch pass;
          // with four the three other rows have words of their own behind the records)
          float* const xscr = NR == 4 ? L + S16_XSCR + 44 * (r - 1) : L + S16_NP_KEY;
          float* out = lead ? L + S16_NP_BOUT : xscr + 24;
          if (ph) collide_plane_hull_coop(A, B, M.contact_offset, out, c, g);
          else collide_box_box_coop(A, B, M.contact_offset, lead ? L + S16_NP_BSCR : xscr, out, c, g);
          const int cnt = __float_as_int(out[0]);  // (one wave: the LDS writes of the group's lanes are complete)
          if (cnt > 0) {
            int off = 0;
            if (c == 0) off = pool_alloc(Lg, cnt);
            off = gbci(off, 0);
            if (off >= 0) {
              if (c == 0) {
                reinterpret_cast<int*>(Lg)[S16_NP_CNT + idx] = cnt;
                reinterpret_cast<int*>(Lg)[S16_NP_OFF + idx] = off;
              }
              if (c < 3) Lg[S16_NP_HN + 3 * idx + c] = out[1 + c];
              if (c < cnt) *reinterpret_cast<float4*>(Lg + S16_NP_POOL + 4 * (off + c)) = float4{out[4 + 4 * c], out[5 + 4 * c], out[6 + 4 * c], out[7 + 4 * c]};
            }
          }
        }
      }
      BSYNC();  // the manifolds of this wave's envs may have been written by other waves
      (void)bcum; (void)TB;
      if (__any(pool_over) && pool_over && live) atomicOr(&S.overflow[e], MSSIM_OVERFLOW_RAW);  // (recorded for the group's own env: a reported condition either way)
      WSYNC();
      PH(24);
      BT_T0b;
      // ---- contact patches (include/mssim.h, MSSIM_PATCH_COS): manifolds of one body pair with normals inside a
      // cone of the first of them are one patch, cut to its 4 most significant points. Lane c looks after the
      // manifolds c, c + 16, ..: (1) body-pair key, (2) anchor = first manifold of the same key within the cone,
      // (3) the anchor's lane selects the patch's points when there are more than 4, (4) records are written in
      // pair order from what survives.
      {
        int* const cnt_ = reinterpret_cast<int*>(L + S16_NP_CNT);
        const int* const off_ = reinterpret_cast<const int*>(L + S16_NP_OFF);
        int* const keep_ = reinterpret_cast<int*>(L + S16_NP_KEEP);
        int* const key_ = reinterpret_cast<int*>(L + S16_NP_KEY);
        const int* const hit_ = reinterpret_cast<const int*>(L + S16_NP_HIT);
        for (int i = c; i < nh; i += 16) {
          const int pk = hit_[i];
          const unsigned pa = __float_as_uint(L[S16_NP_SHP + S16_SHP * ((pk >> 16) & 0xFF) + 14]), pb = __float_as_uint(L[S16_NP_SHP + S16_SHP * ((pk >> 24) & 0xFF) + 14]);
          // bit 16: the manifold carries a torsional friction row (either shape has a patch radius)
          const bool tors = fmaxf(L[S16_NP_SHP + S16_SHP * ((pk >> 16) & 0xFF) + 19], L[S16_NP_SHP + S16_SHP * ((pk >> 24) & 0xFF) + 19]) > 0.f;
          key_[i] = (int)(((pa >> 10) & 31u) | (((pb >> 10) & 31u) << 8)) | (tors ? 1 << 16 : 0);
        }
        WSYNC();
        // sleeping free bodies: a "disturber" is an articulation link or a free body that is awake and not calm. A
        // sleeping body touched by a disturber wakes; the manifolds of a body that stays asleep are dropped; whether a
        // disturber touched it feeds its sleep counter at the end of the substep.
        {
          unsigned dist = 0u;  // bit b: free body b has a manifold with a disturber
          auto free_of = [&](int slot1) { return (slot1 > S16_PT_FREE && slot1 <= S16_PT_FREE + S16_MAX_FREE) ? slot1 - 1 - S16_PT_FREE : -1; };
          auto disturber = [&](int slot1) {
            if (slot1 > S16_PT_LINK && slot1 <= S16_PT_FREE) return true;  // pose slots 1..16: articulation links
            const int b = free_of(slot1);
            return b >= 0 && fw_at(b) > 0.f && !fc_at(b);
          };
          for (int i = c; i < nh; i += 16) {
            if (cnt_[i] <= 0) continue;
            const int ky = key_[i];
            const int s1a = ky & 31, s1b = (ky >> 8) & 31;
            const int ba = free_of(s1a), bb = free_of(s1b);
            if (ba >= 0 && disturber(s1b)) dist |= 1u << ba;
            if (bb >= 0 && disturber(s1a)) dist |= 1u << bb;
          }
          dist |= __shfl_xor(dist, 8, 16); dist |= __shfl_xor(dist, 4, 16); dist |= __shfl_xor(dist, 2, 16); dist |= __shfl_xor(dist, 1, 16);
          if constexpr (NR > 1) dist = (unsigned)env_bci((int)dist, 0);  // (every lane of the env keeps the sleep counters)
          fdist = dist;
#pragma unroll
          for (int b = 0; b < S16_MAX_FREE; b++)
            if (b < nf && fwake[b] <= 0.f && ((dist >> b) & 1u)) fwake[b] = MSSIM_WAKE_TIME;
          for (int i = c; i < nh; i += 16) {
            const int ky = key_[i];
            const int ba = free_of(ky & 31), bb = free_of((ky >> 8) & 31);
            const bool sa_ = ba >= 0 && fw_at(ba) <= 0.f, sb_ = bb >= 0 && fw_at(bb) <= 0.f;
            if (sa_ || sb_) cnt_[i] = 0;
          }
          WSYNC();
        }
        BT_T(8);
        bool any_big = false;
        for (int i = c; i < nh; i += 16) {
          int anchor = i;
          const int ci = cnt_[i];
          if (ci > 0) {
            const int ky = key_[i] & 0xFFFF;
            const f3 ni = f3{L[S16_NP_HN + 3 * i], L[S16_NP_HN + 3 * i + 1], L[S16_NP_HN + 3 * i + 2]};
            for (int k = 0; k < i; k++) {
              if (cnt_[k] > 0 && (key_[k] & 0xFFFF) == ky && dot(f3{L[S16_NP_HN + 3 * k], L[S16_NP_HN + 3 * k + 1], L[S16_NP_HN + 3 * k + 2]}, ni) >= MSSIM_PATCH_COS) { anchor = k; break; }
            }
          }
          keep_[i] = (anchor << 4) | ((1 << ci) - 1);
        }
        WSYNC();
        BT_T(25);
        // pass 1, the anchor's lane: size of its patch and whether it carries a torsional record
        for (int a = c; a < nh; a += 16) {
          if ((keep_[a] >> 4) != a || cnt_[a] == 0) continue;
          int total = 0, tflag = 0;
          for (int i = a; i < nh; i++) {
            const bool mem = (keep_[i] >> 4) == a;
            total += mem ? cnt_[i] : 0;
            tflag |= (mem && cnt_[i] > 0) ? (key_[i] >> 16) & 1 : 0;
          }
          // bit 17 (anchors): the patch carries a torsional friction record; bit 18: more than 4 points, to be cut
          key_[a] |= (tflag << 17) | (total > 4 ? 1 << 18 : 0);
          any_big = any_big || total > 4;
        }
        WSYNC();
        // pass 2, the whole group on one patch at a time: its candidate points in (manifold, point) order are spread over
        // the lanes (position p in lane p % 16), every scan of the selection rule is a group reduction -- the extremum,
        // then the lowest position within the tie tolerance of it ("first candidate wins", as the oracle's sequential scans)
        if (__any(any_big)) {
          unsigned short* const cl = reinterpret_cast<unsigned short*>(L + S16_NP_ML);  // (the stage B / C lists are dead) candidate: id | pool index << 8
          auto b16 = [&](bool v) __attribute__((always_inline)) { return (unsigned)(__ballot(v) >> (16 * g)) & 0xFFFFu; };
          for (int a = 0; a < nh; a++) {  // (group-uniform)
            if (!((key_[a] >> 18) & 1)) continue;
            const f3 na = f3{L[S16_NP_HN + 3 * a], L[S16_NP_HN + 3 * a + 1], L[S16_NP_HN + 3 * a + 2]};
            int n = 0;
            for (int base = a; base < nh; base += 16) {
              const int i = base + c;
              const bool mem = i < nh && (keep_[i] >> 4) == a && cnt_[i] > 0;
              const int ni = mem ? cnt_[i] : 0;
              const unsigned b0 = b16(ni & 1), b1 = b16(ni & 2), b2 = b16(ni & 4), lt = (1u << c) - 1u;
              const int pre = n + __popc(b0 & lt) + 2 * __popc(b1 & lt) + 4 * __popc(b2 & lt);
              if (mem) {
                const int of = off_[i];
                for (int q = 0; q < ni; q++)
                  if (pre + q < 64) cl[pre + q] = (unsigned short)((4 * i + q) | ((of + q) << 8));
              }
              n += __popc(b0) + 2 * __popc(b1) + 4 * __popc(b2);
            }
            WSYNC();
            int i0 = -1, i1 = -1, i2 = -1, i3 = -1;
            if (n <= 64) {
              float4 P[4];
              bool has[4];
#pragma unroll
              for (int k = 0; k < 4; k++) {
                has[k] = c + 16 * k < n;
                const int w = has[k] ? cl[c + 16 * k] : 0;
                P[k] = *reinterpret_cast<const float4*>(L + S16_NP_POOL + 4 * (w >> 8));
              }
              auto point_at = [&](int pos) __attribute__((always_inline)) { return *reinterpret_cast<const float4*>(L + S16_NP_POOL + 4 * (cl[pos] >> 8)); };
              // first position whose value passes `ok` (lowest over the group), -1 if none
              auto first_pos = [&](const bool (&ok)[4]) __attribute__((always_inline)) {
                int p = 1 << 20;
#pragma unroll
                for (int k = 3; k >= 0; k--) p = ok[k] ? c + 16 * k : p;
                p = gmin16i(p);
                return p == (1 << 20) ? -1 : p;
              };
              float v[4];
              bool ok[4];
              // deepest point
              float best = 3e38f;
#pragma unroll
              for (int k = 0; k < 4; k++) best = fminf(best, has[k] ? P[k].w : 3e38f);
              best = -gmax16(-best);
#pragma unroll
              for (int k = 0; k < 4; k++) ok[k] = has[k] && P[k].w <= best + MSSIM_PATCH_TIE_SEP;
              const int q0 = first_pos(ok);
              {  // a patch resting on three or more points: the ones well above them do not compete (MSSIM_PATCH_SLACK)
                int near_ = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) near_ += __popc(b16(has[k] && P[k].w <= best + MSSIM_PATCH_SLACK));
                if (near_ >= 3) {
#pragma unroll
                  for (int k = 0; k < 4; k++) has[k] = has[k] && P[k].w <= best + MSSIM_PATCH_SLACK;
                }
              }
              const float4 P0 = point_at(q0);
              const f3 p0 = f3{P0.x, P0.y, P0.z};
              // farthest from it
              best = -1.f;
#pragma unroll
              for (int k = 0; k < 4; k++) {
                const f3 d = f3{P[k].x, P[k].y, P[k].z} - p0;
                v[k] = dot(d, d);
                has[k] = has[k] && c + 16 * k != q0;
                best = fmaxf(best, has[k] ? v[k] : -1.f);
              }
              best = gmax16(best);
#pragma unroll
              for (int k = 0; k < 4; k++) ok[k] = has[k] && v[k] >= best - MSSIM_PATCH_TIE_REL * best;
              const int q1 = first_pos(ok);
              const float4 P1 = point_at(q1);
              const f3 ed = f3{P1.x, P1.y, P1.z} - p0;
              // largest area on either side of that edge
              best = -1.f;
#pragma unroll
              for (int k = 0; k < 4; k++) {
                v[k] = dot(cross(ed, f3{P[k].x, P[k].y, P[k].z} - p0), na);
                has[k] = has[k] && c + 16 * k != q1;
                best = fmaxf(best, has[k] ? fabsf(v[k]) : -1.f);
              }
              best = gmax16(best);
#pragma unroll
              for (int k = 0; k < 4; k++) ok[k] = has[k] && fabsf(v[k]) >= best - MSSIM_PATCH_TIE_REL * best;
              const int q2 = first_pos(ok);
              const float4 P2 = point_at(q2);
              const float sgn2 = dot(cross(ed, f3{P2.x, P2.y, P2.z} - p0), na);
              best = 0.f;
#pragma unroll
              for (int k = 0; k < 4; k++) {
                v[k] = sgn2 >= 0.f ? -v[k] : v[k];
                has[k] = has[k] && c + 16 * k != q2;
                best = fmaxf(best, has[k] ? v[k] : 0.f);
              }
              best = gmax16(best);
              int q3 = -1;
              if (best > MSSIM_PATCH_TIE_REL * fabsf(sgn2)) {  // (the 4th point has to add area: not one on the edge i0-i1 up to rounding)
#pragma unroll
                for (int k = 0; k < 4; k++) ok[k] = has[k] && v[k] >= best - MSSIM_PATCH_TIE_REL * best;
                q3 = first_pos(ok);
              }
              i0 = cl[q0] & 255; i1 = cl[q1] & 255; i2 = cl[q2] & 255; i3 = q3 >= 0 ? (cl[q3] & 255) : -1;
              for (int i = a + c; i < nh; i += 16) {
                if ((keep_[i] >> 4) != a) continue;
                int mk = 0;
                if ((i0 >> 2) == i) mk |= 1 << (i0 & 3);
                if ((i1 >> 2) == i) mk |= 1 << (i1 & 3);
                if ((i2 >> 2) == i) mk |= 1 << (i2 & 3);
                if (i3 >= 0 && (i3 >> 2) == i) mk |= 1 << (i3 & 3);
                keep_[i] = (a << 4) | mk;
              }
            } else if (c == 0) {
              // more than 64 candidates in one patch (a heap of bodies on one another): the sequential scans, one lane
          const f3 na = f3{L[S16_NP_HN + 3 * a], L[S16_NP_HN + 3 * a + 1], L[S16_NP_HN + 3 * a + 2]};
              // point id = manifold * 4 + point; "first candidate wins" in (manifold, point) order, as the oracle
              auto scan0 = [&](auto&& f) __attribute__((always_inline)) {
                for (int i = a; i < nh; i++) {
                  if ((keep_[i] >> 4) != a) continue;
                  const int cn = cnt_[i], of = off_[i];
                  for (int q = 0; q < cn; q++) f(4 * i + q, *reinterpret_cast<const float4*>(L + S16_NP_POOL + 4 * (of + q)));
                }
              };
              // every scan: the extremum, then the first candidate within the tie tolerance of it (include/mssim.h)
              int i0 = -1, i1 = -1, i2 = -1, i3 = -1;
              float best = 3e38f;
              f3 p0 = f3{0, 0, 0}, p1 = f3{0, 0, 0};
              scan0([&](int, float4 P) { best = fminf(best, P.w); });
              scan0([&](int id, float4 P) { if (i0 < 0 && P.w <= best + MSSIM_PATCH_TIE_SEP) { i0 = id; p0 = f3{P.x, P.y, P.z}; } });
              // (MSSIM_PATCH_SLACK: with three or more points near the deepest, the ones well above them are passed over)
              int near_ = 0;
              const float low = best + MSSIM_PATCH_SLACK;
              scan0([&](int, float4 P) { near_ += P.w <= low ? 1 : 0; });
              const float lim = near_ >= 3 ? low : 3e38f;
              auto scan_all = scan0;
              auto scan = [&](auto&& f) __attribute__((always_inline)) {
                scan_all([&](int id, float4 P) { if (P.w <= lim) f(id, P); });
              };
              best = -1.f;
              scan([&](int id, float4 P) {
                const f3 d = f3{P.x, P.y, P.z} - p0;
                if (id != i0) best = fmaxf(best, dot(d, d));
              });
              scan([&](int id, float4 P) {
                const f3 d = f3{P.x, P.y, P.z} - p0;
                if (i1 < 0 && id != i0 && dot(d, d) >= best - MSSIM_PATCH_TIE_REL * best) { i1 = id; p1 = f3{P.x, P.y, P.z}; }
              });
              const f3 ed = p1 - p0;
              best = -1.f;
              scan([&](int id, float4 P) {
                if (id != i0 && id != i1) best = fmaxf(best, fabsf(dot(cross(ed, f3{P.x, P.y, P.z} - p0), na)));
              });
              float sgn2 = 0.f;
              scan([&](int id, float4 P) {
                const float ar = dot(cross(ed, f3{P.x, P.y, P.z} - p0), na);
                if (i2 < 0 && id != i0 && id != i1 && fabsf(ar) >= best - MSSIM_PATCH_TIE_REL * best) { i2 = id; sgn2 = ar; }
              });
              best = 0.f;
              scan([&](int id, float4 P) {
                const float ar = dot(cross(ed, f3{P.x, P.y, P.z} - p0), na);
                if (id != i0 && id != i1 && id != i2) best = fmaxf(best, sgn2 >= 0.f ? -ar : ar);
              });
              if (best > MSSIM_PATCH_TIE_REL * fabsf(sgn2))
                scan([&](int id, float4 P) {
                  const float ar = dot(cross(ed, f3{P.x, P.y, P.z} - p0), na);
                  const float v = sgn2 >= 0.f ? -ar : ar;
                  if (i3 < 0 && id != i0 && id != i1 && id != i2 && v >= best - MSSIM_PATCH_TIE_REL * best) i3 = id;
                });
              for (int i = a; i < nh; i++) {
                if ((keep_[i] >> 4) != a) continue;
                int mk = 0;
                if ((i0 >> 2) == i) mk |= 1 << (i0 & 3);
                if ((i1 >> 2) == i) mk |= 1 << (i1 & 3);
                if ((i2 >> 2) == i) mk |= 1 << (i2 & 3);
                if (i3 >= 0 && (i3 >> 2) == i) mk |= 1 << (i3 & 3);
                keep_[i] = (a << 4) | mk;
              }

            }
            WSYNC();
          }
        }
        BT_T(23);
        (void)0;
        WSYNC();
        // records in solver order: patch by patch (patches in the order of their anchors), inside a patch manifold by
        // manifold (a manifold belongs to one patch: the points of a shape pair stay together), then the patch's
        // torsional friction record if one of its shapes carries a patch radius (one solver block each,
        // include/mssim.h shape_material). The lane of manifold i writes its points, the lane of an anchor the
        // torsional record of its patch.
        // slot layout, the whole group on one patch at a time (patches in anchor order): a member manifold's points start at
        // the running count plus the points of the members in front of it (prefix over the lanes by ballots), the patch's
        // torsional record follows its points. Lane c keeps the results of its manifolds c, c + 16, c + 32, c + 48.
        int offs[4] = {0, 0, 0, 0}, offe[4] = {0, 0, 0, 0};
        float trr[4] = {-1.f, -1.f, -1.f, -1.f};  // (anchors) torsional radius of the patch, < 0: no torsional record
        int tot = 0;
        {
          auto b16 = [&](bool v) __attribute__((always_inline)) { return (unsigned)(__ballot(v) >> (16 * g)) & 0xFFFFu; };
          for (int a = 0; a < nh; a++) {  // (group-uniform)
            if ((keep_[a] >> 4) != a || cnt_[a] == 0) continue;
            const bool tpatch = (key_[a] >> 17) & 1;
            float tr = 0.f;
            for (int base = a & ~15; base < nh; base += 16) {
              const int i = base + c;
              const int wi = i < nh ? keep_[i] : 0;
              const bool mem = i >= a && i < nh && (wi >> 4) == a;
              const int ni = mem ? __popc(wi & 15) : 0;
              const unsigned b0 = b16(ni & 1), b1 = b16(ni & 2), b2 = b16(ni & 4), lt = (1u << c) - 1u;
              const int pre = tot + __popc(b0 & lt) + 2 * __popc(b1 & lt) + 4 * __popc(b2 & lt);
#pragma unroll
              for (int k = 0; k < 4; k++)
                if (mem && (base >> 4) == k) offs[k] = pre;
              tot += __popc(b0) + 2 * __popc(b1) + 4 * __popc(b2);
              if (tpatch) {
                float t = 0.f;
                if (mem && cnt_[i] > 0 && ((key_[i] >> 16) & 1)) {
                  const int pkj = hit_[i];
                  t = fmaxf(L[S16_NP_SHP + S16_SHP * ((pkj >> 16) & 0xFF) + 19], L[S16_NP_SHP + S16_SHP * ((pkj >> 24) & 0xFF) + 19]);
                }
                tr = fmaxf(tr, gmax16(t));
              }
            }
            if (tpatch) {
#pragma unroll
              for (int k = 0; k < 4; k++)
                if (c + 16 * k == a) { offe[k] = tot; trr[k] = tr; }
              tot++;
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int i = c + 16 * k;
          if (i >= nh) continue;
          const int wi = keep_[i];
          const int mk = wi & 15, a = wi >> 4;
          const bool is_anchor = a == i && cnt_[i] > 0;
          if (mk == 0 && !is_anchor) continue;
          int off = offs[k];
          const int off_end = offe[k];
          const bool my_tors = is_anchor && trr[k] >= 0.f;
          const float my_tr = trr[k];
          const int pk = hit_[i];
          const int sa = (pk >> 16) & 0xFF, sb = (pk >> 24) & 0xFF;
          const float mu = 0.5f * (L[S16_NP_SHP + S16_SHP * sa + 15] + L[S16_NP_SHP + S16_SHP * sb + 15]);
          const int slA = (int)((__float_as_uint(L[S16_NP_SHP + S16_SHP * sa + 14]) >> 10) & 31u) - 1, slB = (int)((__float_as_uint(L[S16_NP_SHP + S16_SHP * sb + 14]) >> 10) & 31u) - 1;
          const int bodies = (int)(slot_lane_mask<NR>(L, slA, n, 0) | (slot_lane_mask<NR>(L, slB, n, 0) << 16));
          const int bodies1 = NR > 1 ? (int)(slot_lane_mask<NR>(L, slA, n, 1) | (slot_lane_mask<NR>(L, slB, n, 1) << 16)) : 0;
          const int bodies2 = NR > 2 ? (int)(slot_lane_mask<NR>(L, slA, n, 2) | (slot_lane_mask<NR>(L, slB, n, 2) << 16)) : 0;
          const int bodies3 = NR > 2 ? (int)(slot_lane_mask<NR>(L, slA, n, 3) | (slot_lane_mask<NR>(L, slB, n, 3) << 16)) : 0;
          const f3 nn = f3{L[S16_NP_HN + 3 * i], L[S16_NP_HN + 3 * i + 1], L[S16_NP_HN + 3 * i + 2]};
          const int of = off_[i];
          for (int q = 0; q < 4; q++) {
            if (!((mk >> q) & 1)) continue;
            if (off < MAXC) {
              const float4 P = *reinterpret_cast<const float4*>(L + S16_NP_POOL + 4 * (of + q));
              float* r = L + S16_REC + S16_REC_LEN * off;
              r[0] = nn.x; r[1] = nn.y; r[2] = nn.z;
              r[3] = P.x; r[4] = P.y; r[5] = P.z;
              r[6] = P.w - M.rest_offset;
              // pair | patch | manifold slot (warm-start key) | bit 27: no key (the manifolds of a mesh pair's triangles share the pair)
              r[7] = __int_as_float((pk & 0xFFFF) | (a << 16) | (q << 24) | ((TRI && (__float_as_uint(L[S16_NP_SHP + S16_SHP * sb + 14]) & 7u) == SH_TRIMESH) ? 1 << 27 : 0));
              r[8] = __int_as_float(bodies);
              r[9] = mu;
              if constexpr (NR > 1) r[10] = __int_as_float(bodies1);
              if constexpr (NR > 2) { r[11] = __int_as_float(bodies2); r[12] = __int_as_float(bodies3); }
            }
            off++;
          }
          if (my_tors && off_end < MAXC) {
            float* r = L + S16_REC + S16_REC_LEN * off_end;
            r[0] = nn.x; r[1] = nn.y; r[2] = nn.z;
            r[3] = 0.f; r[4] = 0.f; r[5] = 0.f; r[6] = 0.f;
            r[7] = __int_as_float((pk & 0xFFFF) | (a << 16) | (1 << 30));  // bit 30: torsional record
            r[8] = __int_as_float(bodies);
            r[9] = mu * my_tr;
            if constexpr (NR > 1) r[10] = __int_as_float(bodies1);
            if constexpr (NR > 2) { r[11] = __int_as_float(bodies2); r[12] = __int_as_float(bodies3); }
          }
        }
        if (tot > MAXC) { if (live && c == 0) atomicOr(&S.overflow[e], MSSIM_OVERFLOW_CONTACTS); tot = MAXC; }
        nc = tot;
        if constexpr (NR > 1) nc = env_bci(nc, 0);  // (the env's other row built none of this)
      }
      WSYNC();
    }
    if (FUSED && last && lead) {
      nold = S.hit_list[e];
#pragma unroll
      for (int k = 0; k < 3; k++)
        if (c + 16 * k < nold) oldp[k] = S.hit_list[(size_t)(1 + c + 16 * k) * N + e];
    }
    BT_T(24);
    PH(21);

    // ================================================================ per-env constants of this lane
    float qt_c = 0.f, qdt_c = 0.f, qf_c = 0.f;
    bool rev_c = false;
    unsigned anc_c = 0u;
    float kp0 = 0.f, kd0 = 0.f, fmax = 3e38f, arm = 0.f, lo_c = -3e38f, hi_c = 3e38f;
    float inert[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool grav_c = false, accel_mode = false;
    pose_t JF_c = pose_t{f3{0, 0, 0}, q4{1, 0, 0, 0}};  // joint frame in the parent body frame
    f3 al_c = f3{0, 0, 0};                            // joint axis in the joint frame
    int par_c = -1;
    if (art) {
      float r[32];  // the joint's 128-byte constant record: 8 x 16-byte loads from one place
      const float4* rp = reinterpret_cast<const float4*>(M.dof_pack + 32 * c);
#pragma unroll
      for (int k = 0; k < 8; k++) { const float4 t = rp[k]; r[4 * k] = t.x; r[4 * k + 1] = t.y; r[4 * k + 2] = t.z; r[4 * k + 3] = t.w; }
      JF_c = pose_t{f3{r[0], r[1], r[2]}, qnormalized(q4{r[3], r[4], r[5], r[6]})};
      al_c = f3{r[7], r[8], r[9]};
      par_c = __float_as_int(r[10]);
      rev_c = __float_as_int(r[11]) == MSSIM_JOINT_REVOLUTE;
      anc_c = __float_as_uint(r[12]);
      kp0 = r[13]; kd0 = r[14]; fmax = r[15];
      accel_mode = (int)r[16] == MSSIM_DRIVE_ACCELERATION;
      arm = r[17];
      lo_c = r[18]; hi_c = r[19];
#pragma unroll
      for (int k = 0; k < 10; k++) inert[k] = r[20 + k];
      grav_c = __float_as_int(r[30]) != 0;
      qt_c = SOA(S.qt, c); qdt_c = SOA(S.qdt, c); qf_c = SOA(S.qf, c);
    }
    float fin[2][10];  // (the inertial parameters of the free bodies of this lane's row)
#pragma unroll
    for (int b = 0; b < 2; b++) {
#pragma unroll
      for (int k = 0; k < 10; k++) fin[b][k] = 0.f;
      if (b < nfr) free_inertial_of(M, N, fb0 + b, e, fin[b]);
    }

    // ================================================================ dynamics: RNEA bias + CRBA
    // body pose from the pose table; world joint axis / anchor from the parent's pose (the arithmetic of the FK below)
    const pose_t root = lds_pose(L + S16_PT);
    pose_t bp_c = root;
    f3 aw_c = f3{0, 0, 0}, an_c = f3{0, 0, 0};
    if (art) {
      bp_c = lds_pose(L + S16_BP + 7 * c);
      pose_t Wp0 = root;
      if (par_c >= 0) Wp0 = lds_pose(L + S16_BP + 7 * par_c);
      const pose_t Jw0 = pmul(Wp0, JF_c);
      aw_c = qrot(Jw0.q, al_c);
      an_c = Jw0.p;
    }
    sv6 S_c = sv6{f3{0, 0, 0}, f3{0, 0, 0}};
    if (art) S_c = rev_c ? sv6{aw_c, cross(an_c - O, aw_c)} : sv6{f3{0, 0, 0}, aw_c};
    if (lead) {  // (the staging tables of the dynamics belong to the articulation's row)
      float* p = L + S16_S + 6 * c;
      p[0] = S_c.w.x; p[1] = S_c.w.y; p[2] = S_c.w.z; p[3] = S_c.v.x; p[4] = S_c.v.y; p[5] = S_c.v.z;
      L[S16_VEC + c] = qd_c;
    }
    WSYNC();
    // V_c = sum over (ancestors + self) of S_i qd_i
    sv6 V = sv6{f3{0, 0, 0}, f3{0, 0, 0}};
    for (int i = 0; i < n; i++) {
      float m = (((anc_c | self_c) >> i) & 1u) ? L[S16_VEC + i] : 0.f;
      const float* p = L + S16_S + 6 * i;
      V.w += f3{p[0], p[1], p[2]} * m;
      V.v += f3{p[3], p[4], p[5]} * m;
    }
    if (lead) {
      sv6 T = crossm(V, S_c);
      float* p = L + S16_T + 6 * c;
      p[0] = T.w.x * qd_c; p[1] = T.w.y * qd_c; p[2] = T.w.z * qd_c; p[3] = T.v.x * qd_c; p[4] = T.v.y * qd_c; p[5] = T.v.z * qd_c;
    }
    WSYNC();
    sv6 Ab = sv6{f3{0, 0, 0}, f3{0, 0, 0}};
    for (int i = 0; i < n; i++) {
      float m = (((anc_c | self_c) >> i) & 1u) ? 1.f : 0.f;
      const float* p = L + S16_T + 6 * i;
      Ab.w += f3{p[0], p[1], p[2]} * m;
      Ab.v += f3{p[3], p[4], p[5]} * m;
    }
    si10 I_c = si10{0.f, f3{0, 0, 0}, s3{0, 0, 0, 0, 0, 0}};
    sf6 Fb_c = sf6{f3{0, 0, 0}, f3{0, 0, 0}};
    if (art) {
      m3 R = qmat(bp_c.q);
      s3 Iw = srotate(R, s3{inert[4], inert[5], inert[6], inert[7], inert[8], inert[9]});
      f3 cm = bp_c.p + mmulv(R, f3{inert[1], inert[2], inert[3]}) - O;
      float m = inert[0], cc = dot(cm, cm);
      Iw.xx += m * (cc - cm.x * cm.x); Iw.yy += m * (cc - cm.y * cm.y); Iw.zz += m * (cc - cm.z * cm.z);
      Iw.xy -= m * cm.x * cm.y; Iw.xz -= m * cm.x * cm.z; Iw.yz -= m * cm.y * cm.z;
      I_c = si10{m, cm * m, Iw};
      sf6 f1 = imul(I_c, Ab);
      sf6 f2 = crossf(V, imul(I_c, V));
      Fb_c = sf6{f1.n + f2.n, f1.f + f2.f};
      if (grav_c) { Fb_c.f -= g3 * m; Fb_c.n -= cross(I_c.h, g3); }
    }
    if (lead) {
      float* p = L + S16_F + 6 * c;
      p[0] = Fb_c.n.x; p[1] = Fb_c.n.y; p[2] = Fb_c.n.z; p[3] = Fb_c.f.x; p[4] = Fb_c.f.y; p[5] = Fb_c.f.z;
      float* qI = L + S16_IC + 10 * c;
      qI[0] = I_c.m; qI[1] = I_c.h.x; qI[2] = I_c.h.y; qI[3] = I_c.h.z;
      qI[4] = I_c.I.xx; qI[5] = I_c.I.yy; qI[6] = I_c.I.zz; qI[7] = I_c.I.xy; qI[8] = I_c.I.xz; qI[9] = I_c.I.yz;
#pragma unroll
      for (int k = 0; k < 16; k++) L[S16_MAT + 16 * c + k] = 0.f;
    }
    WSYNC();
    // composite force / inertia: sum over (descendants + self)
    sf6 Fc = sf6{f3{0, 0, 0}, f3{0, 0, 0}};
    si10 Icc = si10{0.f, f3{0, 0, 0}, s3{0, 0, 0, 0, 0, 0}};
    for (int k = 0; k < n; k++) {
      const unsigned ak = reinterpret_cast<const unsigned*>(L)[S16_ANC + k];
      float m = (art && ((ak >> c) & 1u)) ? 1.f : 0.f;
      const float* p = L + S16_F + 6 * k;
      Fc.n += f3{p[0], p[1], p[2]} * m;
      Fc.f += f3{p[3], p[4], p[5]} * m;
      const float* qI = L + S16_IC + 10 * k;
      Icc.m += qI[0] * m;
      Icc.h += f3{qI[1], qI[2], qI[3]} * m;
      Icc.I.xx += qI[4] * m; Icc.I.yy += qI[5] * m; Icc.I.zz += qI[6] * m;
      Icc.I.xy += qI[7] * m; Icc.I.xz += qI[8] * m; Icc.I.yz += qI[9] * m;
    }
    const float bias_c = sdot(S_c, Fc);
    {
      sf6 Fcol = imul(Icc, S_c);
      for (int i = 0; i < n; i++) {
        if (art && (((anc_c | self_c) >> i) & 1u)) {
          const float* p = L + S16_S + 6 * i;
          float v = dot(f3{p[0], p[1], p[2]}, Fcol.n) + dot(f3{p[3], p[4], p[5]}, Fcol.f);
          L[S16_MAT + 16 * c + i] = v;
          L[S16_MAT + 16 * i + c] = v;
        }
      }
    }
    WSYNC();
    float Mrow[16], qdv[16];
    ld16(L + S16_MAT + 16 * c, Mrow);
    ld16(L + S16_VEC, qdv);
    const float Mdiag = L[S16_MAT + 16 * c + c];
    float kp = kp0, kd = kd0;
    if (accel_mode) { kp *= Mdiag; kd *= Mdiag; }
    const float tau0 = kp * (qt_c - q_c) + kd * qdt_c;
    float Dj = dt * kd + dt * dt * kp;
    float mv = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) mv += Mrow[k] * qdv[k];
    // tendons: torque into rhs, implicit stiffness into the matrix row
    float tau_t = 0.f;
    float Trow[16];
#pragma unroll
    for (int k = 0; k < 16; k++) Trow[k] = 0.f;
    for (int t = 0; t < M.n_tendon; t++) {
      const int a = M.tendon_dof[2 * t], b = M.tendon_dof[2 * t + 1];
      const float* tp = M.tendon_param + 5 * t;
      const float ca = tp[0], cb = tp[1];
      const float cval = ca * gbc(q_c, a) + cb * gbc(q_c, b) - tp[2];
      const float w = dt * dt * tp[3] + dt * tp[4];
      const float cj = c == a ? ca : (c == b ? cb : 0.f);
      tau_t -= tp[3] * cval * cj;
#pragma unroll
      for (int k = 0; k < 16; k++) Trow[k] += w * cj * (k == a ? ca : (k == b ? cb : 0.f));
    }
    // (the drive torque is kept apart from the rest of the right-hand side: a saturated joint swaps it for the
    // constant limit torque, and with a target far away -- an IK step near a singular pose asks for 1e8 rad --
    // `rhs += dt * (limit - tau0)` would cancel 1e9 against 1e9 in f32)
    const float rhs0_c = art ? mv + dt * (tau_t - bias_c + qf_c) : 0.f;
    float rhs_c = art ? rhs0_c + dt * tau0 : 0.f;
    PH(1);

    // ================================================================ A^-1 by Gauss-Jordan (row per lane)
    float Irow[16];
    float vstar = 0.f;
    // (a Sherman-Morrison downdate per saturated joint instead of the second elimination was tried:
    // 1 - D_j a_jj is ~0.01..0.1 for these drives and the cancellation costs ~3 digits in f32)
    // (joint count known at compile time: the elimination only touches the articulation's NA columns,
    // rounded up to whole 16-byte LDS words; A^-1 is block diagonal, the rest of the row stays zero)
    constexpr int NA = NDOF ? ((NDOF + 3) & ~3) : 16;
#pragma unroll
    for (int k = 0; k < 16; k++) Irow[k] = 0.f;
    EXP_DUP(DUP_GJ)
    for (int pass = 0; pass < 2; pass++) {
      float Arow[NA];
#pragma unroll
      for (int k = 0; k < NA; k++) {
        Arow[k] = art ? (Mrow[k] + Trow[k] + (k == c ? Dj + arm : 0.f)) : (k == c ? 1.f : 0.f);
        Irow[k] = k == c ? 1.f : 0.f;
      }
      // (division-free elimination: the pivot row stays unscaled and is broadcast inside the env's 16-lane row by
      // DPP row_newbcast -- one v_fmac per element, no LDS round trip --; the columns left of the pivot are already zero
      // in A and the ones right of it still zero in I, so a pivot costs NA multiply-adds; each row is divided by its
      // own diagonal at the end.  Measured: the LDS-broadcast version of this loop was 25 % of a fresh PickCube launch)
      float diag = 1.f;
      gj_eliminate<0, NA>(Arow, Irow, diag, c, n);
#pragma unroll
      for (int j = 0; j < NA; j++) Irow[j] *= diag;
      WSYNC();
      if (lead) L[S16_VEC + 16 + c] = rhs_c;
      WSYNC();
      float rv[16];
      ld16(L + S16_VEC + 16, rv);
      vstar = 0.f;
#pragma unroll
      for (int k = 0; k < NA; k++) vstar += Irow[k] * rv[k];
      if (pass == 1) break;
      // drive force limit: saturated joints get the constant limit torque, lose their implicit terms
      const float td = kp * (qt_c - q_c - dt * vstar) + kd * (qdt_c - vstar);
      const bool sat = art && fmax < 1e30f && fabsf(td) > fmax;
      if (!__any(sat)) break;
      if (sat) {
        rhs_c = rhs0_c + dt * (td > 0.f ? fmax : -fmax);
        Dj = 0.f;
      }
    }
    if (!art) {
#pragma unroll
      for (int k = 0; k < 16; k++) Irow[k] = 0.f;
    }
    PH(2);

    // ================================================================ free bodies
    v_c = art ? vstar : 0.f;
    f3 mycom = f3{0, 0, 0};
#pragma unroll
    for (int bl = 0; bl < 2; bl++) {  // (the free bodies of this lane's row; a row without any skips the loop)
      if (bl >= nfr) break;
      const int b = fb0 + bl;
      const float* in = fin[bl];
      const pose_t P = lds_pose(L + S16_PT + 7 * (S16_PT_FREE + b));
      m3 R = qmat(P.q);
      s3 Iw = srotate(R, s3{in[4], in[5], in[6], in[7], in[8], in[9]});
      s3 Ii = sinverse(Iw);
      const float minv = rcp_f(in[0]);
      f3 com = P.p + mmulv(R, f3{in[1], in[2], in[3]});
      const int base = fcol0 + 6 * bl;
      f3 v0 = f3{gbc(vfree_c, base), gbc(vfree_c, base + 1), gbc(vfree_c, base + 2)};
      f3 w0 = clamp_norm(f3{gbc(vfree_c, base + 3), gbc(vfree_c, base + 4), gbc(vfree_c, base + 5)}, MSSIM_MAX_ANGULAR_VELOCITY);
      f3 acc = f3{gbc(fforce_c, base), gbc(fforce_c, base + 1), gbc(fforce_c, base + 2)} * minv;
      if (M.free_gravity[b]) acc += g3;
      f3 vv = v0 + acc * dt;
      f3 ww = w0 - smulv(Ii, cross(w0, smulv(Iw, w0))) * dt;
      const float ld = 1.f - dt * M.free_damping[2 * b], ad = 1.f - dt * M.free_damping[2 * b + 1];
      vv = vv * (ld > 0.f ? ld : 0.f);
      ww = ww * (ad > 0.f ? ad : 0.f);
      if (c == 0) { L[S16_COM + 3 * b] = com.x; L[S16_COM + 3 * b + 1] = com.y; L[S16_COM + 3 * b + 2] = com.z; }
      const bool asleep_b = fw_at(b) <= 0.f;  // at rest and out of the solver (none of its manifolds was kept)
      if (freel && fb_id == b) {
        mycom = com;
        v_c = asleep_b ? 0.f : (fk < 3 ? comp(vv, fk) : comp(ww, fk - 3));
        const f3 irow = fk == 3 ? f3{Ii.xx, Ii.xy, Ii.xz} : (fk == 4 ? f3{Ii.xy, Ii.yy, Ii.yz} : f3{Ii.xz, Ii.yz, Ii.zz});
#pragma unroll
        for (int k = 0; k < 16; k++) {
          float val = 0.f;
          if (fk < 3) val = k == c ? minv : 0.f;
          else val = k == fbase + 3 ? irow.x : (k == fbase + 4 ? irow.y : (k == fbase + 5 ? irow.z : 0.f));
          Irow[k] = asleep_b ? 0.f : val;
        }
      }
    }
    // this lane's velocity component as an axis (+ origin for rotation-like components), see the row build
    f3 jaxis_c = f3{0, 0, 0}, jorigin_c = f3{0, 0, 0};
    bool jrot_c = false;
    if (art) { jaxis_c = aw_c; jorigin_c = an_c; jrot_c = rev_c; }
    else if (freel) { jaxis_c = f3{(fk % 3) == 0 ? 1.f : 0.f, (fk % 3) == 1 ? 1.f : 0.f, (fk % 3) == 2 ? 1.f : 0.f}; jorigin_c = mycom; jrot_c = fk >= 3; }
    fforce_c = 0.f;  // an applied force acts during one substep only
    WSYNC();  // the dynamics staging area is dead from here on: rows overlay it
    PH(3);

    // ================================================================ rows
    // joint limits: row j for joint j. J = side_j e_j, W = side_j * column j of A^-1 (LDS [j][16]);
    // the scalars and the multiplier stay in lane j's registers.
    float lim_inv = 0.f, lim_bpos = 0.f, lim_bvel = 0.f, lim_side = 0.f;
    lim_lam = 0.f;
    {
      const bool has = art && (lo_c > -1e30f || hi_c < 1e30f);
      const float dlo = q_c - lo_c, dhi = hi_c - q_c;
      const float C = dlo <= dhi ? dlo : dhi;
      lim_side = has ? (dlo <= dhi ? 1.f : -1.f) : 0.f;
      float dself = 0.f;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        if (j >= n) break;
        const float sj = gbc(lim_side, j);
        if (lead) L[S16_LIMW + 16 * j + c] = sj * Irow[j];
        if (c == j) dself = Irow[j];
      }
      lim_inv = (has && dself > 1e-12f) ? rcp_f(dself) : 0.f;
      lim_bpos = C >= 0.f ? C * inv_dt : fmaxf(M.erp * C * inv_dt, -M.max_depen);
      lim_bvel = C >= 0.f ? C * inv_dt : 0.f;
    }
    // A^-1 row in rotated order for W = A^-1 J^T of the contact rows: Irot[k] = Ainv[c][src_k], src_k = the
    // lane a DPP row rotation by k delivers to lane c (taken from the rotation itself, so no convention
    // about its direction enters). Per-lane dynamic indexing goes through 16 private LDS words.
    float Irot[16];
    EXP_DUP(DUP_IROT) {
      float* tmp = L + S16_JW + 16 * cl;
#pragma unroll
      for (int j = 0; j < 16; j += 4) *reinterpret_cast<float4*>(tmp + j) = float4{Irow[j], Irow[j + 1], Irow[j + 2], Irow[j + 3]};
      rot_gather<0>(tmp, c, Irot);
    }
    // contacts: a block of 3 rows each (normal, t1, t2), built from the LDS records
    int max_nc = nc;
#pragma unroll
    for (int o = 8 * S16_ENVS_PER_BLOCK; o >= 16; o >>= 1) max_nc = max(max_nc, __shfl_xor(max_nc, o));
    max_nc = __builtin_amdgcn_readfirstlane(max_nc);  // wave-uniform: scalar branches instead of exec masks
    // J3 / W3: this lane's Jacobian and A^-1 J^T entries of the three rows; block scalars go to the LDS
    // table. Padding slots (this env has fewer contacts than the wave's longest) become all-zero blocks:
    // the solver sweeps the maximum over the wave's envs and their updates then move nothing.
    auto build_contact = [&](int i, bool ck, float (&J3)[3], float (&W3)[3], float (&lam3)[3]) __attribute__((always_inline)) {
      const float* rec = L + S16_REC + S16_REC_LEN * (ck ? i : 0);
      const f3 nrm = f3{rec[0], rec[1], rec[2]};
      const f3 x = f3{rec[3], rec[4], rec[5]};
      const float sep = rec[6];
      const int pw = __float_as_int(rec[7]);
      const bool tors = ck && ((pw >> 30) & 1);  // torsional friction block of the patch that ends just before it
      const int p = pw & 0xFFFF;
      const bool first = i == 0 || ((__float_as_int(L[S16_REC + S16_REC_LEN * (i - 1) + 7]) ^ pw) & 0x3F0000) != 0;  // first block of its patch
      const int bodies = __float_as_int((NR > 1 && !lead) ? rec[9 + r] : rec[8]);  // (lane masks of this lane's row)
      const float mu = tors ? 0.f : rec[9];
      // (one cross product and one normalisation: the helper axis is selected, not the result)
      const bool use_x = fabsf(nrm.x) < 0.57735f;
      const f3 t1 = normalized(cross(nrm, f3{use_x ? 1.f : 0.f, use_x ? 0.f : 1.f, 0.f}));
      const f3 t2 = cross(nrm, t1);
      // does this lane's component move with side A / side B of the pair? (lane masks in the record)
      const float sgn = (float)((bodies >> c) & 1) - (float)((bodies >> (16 + c)) & 1);
      // Every lane's Jacobian entry is axis . d (translation-like: prismatic joint, free linear) or
      // axis . ((x - origin) x d) (rotation-like: revolute joint about its anchor, free angular about the
      // centre of mass): one code path for articulation and free-body lanes
      const f3 rx = x - jorigin_c;
#pragma unroll
      for (int dk = 0; dk < 3; dk++) {
        const f3 d = dk == 0 ? nrm : (dk == 1 ? t1 : t2);
        const float J = sgn * dot(jaxis_c, sel3(jrot_c, cross(rx, d), d));
        J3[dk] = (ck && !tors) ? J : 0.f;
      }
      // torsional block: row 0 = relative angular velocity about the normal (rotation-like components only), rows 1, 2 empty
      if (tors) J3[0] = jrot_c ? sgn * dot(jaxis_c, nrm) : 0.f;
      // W = A^-1 J^T for the three directions: 16 DPP row rotations of J against the pre-rotated row
      rot_fma3(Irot, J3, W3);
      // warm start: the block's initial impulses act on the velocities before the first sweep
      lam3[0] = L[S16_CS + 16 * i + 9]; lam3[1] = L[S16_CS + 16 * i + 10]; lam3[2] = L[S16_CS + 16 * i + 11];
      v_c = fmaf(W3[2], lam3[2], fmaf(W3[1], lam3[1], fmaf(W3[0], lam3[0], v_c)));
      // diagonal and the Delassus cross terms with the earlier rows of this contact (block Gauss-Seidel)
      float d0 = J3[0] * W3[0], d1 = J3[1] * W3[1], d2 = J3[2] * W3[2];
      float g10 = J3[1] * W3[0], g20 = J3[2] * W3[0], g21 = J3[2] * W3[1];
      gsumx3<NR>(d0, d1, d2);
      gsumx3<NR>(g10, g20, g21);
      if (c == 0 && lead) {
        const float i0 = d0 > 1e-12f ? rcp_f(d0) : 0.f, i1 = d1 > 1e-12f ? rcp_f(d1) : 0.f, i2 = d2 > 1e-12f ? rcp_f(d2) : 0.f;
        float4* cs = reinterpret_cast<float4*>(L + S16_CS + 16 * i);
        const bool pt = ck && !tors;
        cs[0] = float4{i0, pt ? (sep >= 0.f ? sep * inv_dt : fmaxf(M.erp * sep * inv_dt, -M.max_depen)) : 0.f, pt ? (sep >= 0.f ? sep * inv_dt : 0.f) : 0.f, pt ? mu : 0.f};
        cs[1] = float4{i1, g10 * i1, i2, g20 * i2};
        L[S16_CS + 16 * i + 8] = g21 * i2;  // (words 9..11: the warm-start multipliers, already in place)
        // torsional bound mu r (0 = contact block) | carry factor of the manifold's normal-multiplier sum (0 = first block)
        cs[3] = float4{__int_as_float(pw), 0.f, tors ? rec[9] : 0.f, (ck && !first) ? 1.f : 0.f};
      }
    };
    // Warm start (include/mssim.h): the multipliers a (shape pair, manifold slot) carried in the previous substep, one
    // 16-byte load per contact, all contacts of the env in flight at once; they land in the multiplier words of the block
    // table (zeros for new contacts, torsional blocks and padding blocks) and are applied as the blocks are built.
    for (int i = lead ? c : max_nc; i < max_nc; i += 16) {
      float4 w = float4{0.f, 0.f, 0.f, 0.f};
      bool ok = false;
      if (i < nc && live) {
        const int pw = __float_as_int(L[S16_REC + S16_REC_LEN * i + 7]);
        if (!((pw >> 30) & 1) && !(TRI && ((pw >> 27) & 1))) {
          w = *reinterpret_cast<const float4*>(S.warm + (((size_t)(4 * (pw & 0xFFFF) + ((pw >> 24) & 3))) * N + e) * 4);
          ok = __float_as_int(w.w) == pcm_tick - 1;
        }
      }
      L[S16_CS + 16 * i + 9] = ok ? w.x : 0.f;
      L[S16_CS + 16 * i + 10] = ok ? w.y : 0.f;
      L[S16_CS + 16 * i + 11] = ok ? w.z : 0.f;
    }
    WSYNC();
    float* const grow = S.rows + (size_t)e * ((size_t)S16_ROWS_GLB * S16_ROWLEN);
    float Jr[S16_REGC][3], Wr[S16_REGC][3], lamr[S16_REGC][3];
#pragma unroll
    for (int k = 0; k < S16_REGC; k++) {
#pragma unroll
      for (int dk = 0; dk < 3; dk++) { Jr[k][dk] = 0.f; Wr[k][dk] = 0.f; lamr[k][dk] = 0.f; }
      if (k < max_nc) {  // wave-uniform
        float J3[3], W3[3], l3[3];
        build_contact(k, k < nc, J3, W3, l3);
#pragma unroll
        for (int dk = 0; dk < 3; dk++) { Jr[k][dk] = J3[dk]; Wr[k][dk] = W3[dk]; lamr[k][dk] = l3[dk]; }
      }
    }
    for (int i = S16_REGC; i < max_nc; i++) {
      float J3[3], W3[3], l3[3];
      build_contact(i, i < nc, J3, W3, l3);
      // (LDS and global rows are written by separate code: one pointer for both would be a flat pointer)
      if (i < S16_REGC + S16_LDSC) {
        float* row = L + S16_JW + S16_JWLEN * (i - S16_REGC);
#pragma unroll
        for (int dk = 0; dk < 3; dk++) { row[2 * GW * dk + cl] = J3[dk]; row[2 * GW * dk + GW + cl] = W3[dk]; }
      } else if (live) {
        float* row = grow + (size_t)S16_JWLEN * (i - S16_REGC - S16_LDSC);
#pragma unroll
        for (int dk = 0; dk < 3; dk++) { row[2 * GW * dk + cl] = J3[dk]; row[2 * GW * dk + GW + cl] = W3[dk]; }
      }
    }
    nrow_con = nc;
    PH_ADD(29, max_nc);
    PH_ADD(30, __shfl(nc, 0) + __shfl(nc, 16) + __shfl(nc, 32) + __shfl(nc, 48));
    WSYNC();
    PH(4);

    // ================================================================ projected Gauss-Seidel
    const int nc_reg = nc < S16_REGC ? nc : S16_REGC;
    const int nc_lds = nc - nc_reg < S16_LDSC ? nc - nc_reg : S16_LDSC;
    const int nc_glb = nc - nc_reg - nc_lds;
    int max_creg = nc_reg, max_clds = nc_lds, max_cglb = nc_glb;
#pragma unroll
    for (int o = 8 * S16_ENVS_PER_BLOCK; o >= 16; o >>= 1) {
      max_creg = max(max_creg, __shfl_xor(max_creg, o));
      max_clds = max(max_clds, __shfl_xor(max_clds, o));
      max_cglb = max(max_cglb, __shfl_xor(max_cglb, o));
    }
    max_creg = __builtin_amdgcn_readfirstlane(max_creg);
    max_clds = __builtin_amdgcn_readfirstlane(max_clds);
    max_cglb = __builtin_amdgcn_readfirstlane(max_cglb);
    // one contact = block of 3 rows. The three J.v reductions are independent (issued back to back);
    // the sequential Gauss-Seidel dependence inside the block is carried by the Delassus cross terms
    // (k10 = J1.W0 / d1, k20 = J2.W0 / d2, k21 = J2.W1 / d2) in scalar arithmetic. The dependent chain
    // per contact is jv0 -> nl0 -> nl1 -> nl2 -> v; everything that does not depend on the previous
    // multiplier of the block (a1, a2, the first two W updates) is computed off the chain.
    // Torsional friction blocks (row 0 = spin about the normal, bounded by +-tmu * the sum of the normal multipliers of
    // the manifold that ends just before the block) ride the same code: the clamp of row 0 is a median of three with
    // the bounds [0, inf) for a contact and [-tmu acc, tmu acc] for a torsional block, `acc` the running sum of the
    // current manifold (restarted by its first block: accmul = 0) -- nothing is added to the dependent chain.
    float acc_n = 0.f;
    auto con_solve = [&](float J0, float W0, float J1, float W1, float J2, float W2, float& lam0, float& lam1, float& lam2,
                         float4 s0, float4 sk, float k21, float tmu, float accmul, bool use_bias) __attribute__((always_inline)) {
      float jv0 = J0 * v_c, jv1 = J1 * v_c, jv2 = J2 * v_c;
      gsumx3<NR>(jv0, jv1, jv2);
      const float a1 = fmaf(-jv1, sk.x, lam1);
      const float a2 = fmaf(-jv2, sk.z, lam2);
      const bool tors = tmu > 0.f;
      const float tb = tmu * acc_n;
      const float nl0 = __builtin_amdgcn_fmed3f(fmaf(-(jv0 + (use_bias ? s0.y : s0.z)), s0.x, lam0), -tb, tors ? tb : 3e38f);
      acc_n = fmaf(acc_n, accmul, tors ? 0.f : nl0);
      const float dl0 = nl0 - lam0;
      const float h = s0.w * nl0;
      const float nl1 = fminf(fmaxf(fmaf(-sk.y, dl0, a1), -h), h);
      const float dl1 = nl1 - lam1;
      const float nl2 = fminf(fmaxf(fmaf(-k21, dl1, fmaf(-sk.w, dl0, a2)), -h), h);
      const float dl2 = nl2 - lam2;
      v_c = fmaf(W2, dl2, fmaf(W1, dl1, fmaf(W0, dl0, v_c)));
      lam0 = nl0; lam1 = nl1; lam2 = nl2;
    };
    // streamed contacts: J | W rows from LDS or global (read-only), scalars and multipliers in the LDS table
    struct ConRec {
      float J0, W0, J1, W1, J2, W2;
      float4 s0, sk, kl;  // (1/d0, bias pos, bias vel, mu), (1/d1, k10, 1/d2, k20), (k21, lam0, lam1, lam2)
      float2 tq;          // (torsional bound factor, carry factor of the manifold's normal-multiplier sum)
    };
    auto con_load = [&](const float* row, int ci, ConRec& R) __attribute__((always_inline)) {
      R.J0 = row[cl]; R.W0 = row[GW + cl];
      R.J1 = row[2 * GW + cl]; R.W1 = row[3 * GW + cl];
      R.J2 = row[4 * GW + cl]; R.W2 = row[5 * GW + cl];
      const float4* cs = reinterpret_cast<const float4*>(L + S16_CS + 16 * ci);
      R.s0 = cs[0]; R.sk = cs[1]; R.kl = cs[2];
      R.tq = *reinterpret_cast<const float2*>(L + S16_CS + 16 * ci + 14);
    };
    auto con_apply = [&](ConRec& R, int ci, bool use_bias) __attribute__((always_inline)) {
      con_solve(R.J0, R.W0, R.J1, R.W1, R.J2, R.W2, R.kl.y, R.kl.z, R.kl.w, R.s0, R.sk, R.kl.x, R.tq.x, R.tq.y, use_bias);
      if (c == 0 && lead) *reinterpret_cast<float4*>(L + S16_CS + 16 * ci + 8) = R.kl;  // (padding blocks rewrite their zeros)
    };
#ifdef EXP_ITERS
    const int n_iters = EXP_ITERS;
#else
    const int n_iters = M.pos_iters + M.vel_iters;
#endif
    // early exit of the position sweeps, per env (include/mssim.h MSSIM_PGS_EXIT_TOLERANCE): `settled` is group-uniform;
    // a sweep runs under the exec mask of the envs that are not, and not at all once the wave's four are
    bool settled = false;
    for (int it = 0; it <= n_iters; it++) {
      if (it == M.pos_iters) {
        q_c += dt * fminf(fmaxf(v_c, -MSSIM_MAX_JOINT_VELOCITY), MSSIM_MAX_JOINT_VELOCITY);
#pragma unroll
        for (int bl = 0; bl < 2; bl++) {
          if (bl >= nfr) break;
          const int b = fb0 + bl;
          const float* in = fin[bl];
          const int base = fcol0 + 6 * bl;
          f3 vv = f3{gbc(v_c, base), gbc(v_c, base + 1), gbc(v_c, base + 2)};
          f3 ww = clamp_norm(f3{gbc(v_c, base + 3), gbc(v_c, base + 4), gbc(v_c, base + 5)}, MSSIM_MAX_ANGULAR_VELOCITY);
          f3 com = f3{L[S16_COM + 3 * b], L[S16_COM + 3 * b + 1], L[S16_COM + 3 * b + 2]} + vv * dt;
          float* pt = L + S16_PT + 7 * (S16_PT_FREE + b);
          q4 qq = qnormalized(q4{pt[3], pt[4], pt[5], pt[6]});
          q4 dq = qmul(q4{0.f, ww.x, ww.y, ww.z}, qq);
          qq = qnormalized(q4{qq.w + 0.5f * dt * dq.w, qq.x + 0.5f * dt * dq.x, qq.y + 0.5f * dt * dq.y, qq.z + 0.5f * dt * dq.z});
          f3 pp = com - qrot(qq, f3{in[1], in[2], in[3]});
          WSYNC();
          if (fw_at(b) <= 0.f) {  // asleep: the pose stays bit for bit
            pp = f3{pt[0], pt[1], pt[2]};
            qq = q4{pt[3], pt[4], pt[5], pt[6]};
          }
          if (c == 0) lds_pose_store(pt, pose_t{pp, qq});
          if (c == 0 && live && last) {
            SOA(S.free_s, 13 * b) = pp.x; SOA(S.free_s, 13 * b + 1) = pp.y; SOA(S.free_s, 13 * b + 2) = pp.z;
            SOA(S.free_s, 13 * b + 3) = qq.w; SOA(S.free_s, 13 * b + 4) = qq.x; SOA(S.free_s, 13 * b + 5) = qq.y; SOA(S.free_s, 13 * b + 6) = qq.z;
          }
          WSYNC();
        }
      }
      if (it == n_iters) break;
      PH(17);
      const bool use_bias = it < M.pos_iters;
      const bool run = !(settled && use_bias);
      if (!__any(run)) { it = M.pos_iters - 1; continue; }  // (wave-uniform) every env of the wave has settled: on to the integration
      const float v_before = v_c;
      BT_ADD(9, 1);  // (debug builds: sweeps this wave ran, limit-row visits)
      if (run) {
      // block scalars of contact k + 1 are read from the LDS table before the dependent chain of contact k
      // (one wave per SIMD: nothing else hides the LDS latency); slot k + 1 always exists in the table
      float4 ns0 = *reinterpret_cast<const float4*>(L + S16_CS), nsk = *reinterpret_cast<const float4*>(L + S16_CS + 4);
      float nk21 = L[S16_CS + 8];
      float2 ntq = *reinterpret_cast<const float2*>(L + S16_CS + 14);
      acc_n = 0.f;
#pragma unroll
      for (int k = 0; k < S16_REGC; k++) {
        if (k < max_creg) {  // wave-uniform
          const float4 s0 = ns0, sk = nsk;
          const float k21 = nk21;
          const float2 tq = ntq;
          if (k + 1 < S16_REGC) {
            const float4* cs = reinterpret_cast<const float4*>(L + S16_CS + 16 * (k + 1));
            ns0 = cs[0]; nsk = cs[1];
            nk21 = L[S16_CS + 16 * (k + 1) + 8];
            ntq = *reinterpret_cast<const float2*>(L + S16_CS + 16 * (k + 1) + 14);
          }
          float l0 = lamr[k][0], l1 = lamr[k][1], l2 = lamr[k][2];
          con_solve(Jr[k][0], Wr[k][0], Jr[k][1], Wr[k][1], Jr[k][2], Wr[k][2], l0, l1, l2, s0, sk, k21, tq.x, tq.y, use_bias);
          lamr[k][0] = l0; lamr[k][1] = l1; lamr[k][2] = l2;
        }
      }
      PH(19);
      if (max_clds > 0) {
        // next block is loaded before the dependent chain of the current one (index clamped: in range)
        const float* jw = L + S16_JW;
        ConRec A, B;
        con_load(jw, S16_REGC, A);
        int k = 0;
        while (true) {
          con_load(jw + S16_JWLEN * min(k + 1, max_clds - 1), S16_REGC + min(k + 1, max_clds - 1), B);
          con_apply(A, S16_REGC + k, use_bias);
          if (++k >= max_clds) break;
          con_load(jw + S16_JWLEN * min(k + 1, max_clds - 1), S16_REGC + min(k + 1, max_clds - 1), A);
          con_apply(B, S16_REGC + k, use_bias);
          if (++k >= max_clds) break;
        }
      }
      if (max_cglb > 0) {
        // contacts beyond the LDS capacity: their J | W rows stream (read-only) from the per-env global
        // scratch, three blocks ahead -- an L2 round trip is ~3 block solves long
        auto gload = [&](int k, ConRec& R) __attribute__((always_inline)) {
          const int kk = min(k, max_cglb - 1);
          const float* row = grow + (size_t)S16_JWLEN * kk;
          R.J0 = row[cl]; R.W0 = row[GW + cl];
          R.J1 = row[2 * GW + cl]; R.W1 = row[3 * GW + cl];
          R.J2 = row[4 * GW + cl]; R.W2 = row[5 * GW + cl];
        };
        auto gapply = [&](ConRec& R, int k) __attribute__((always_inline)) {
          const int ci = S16_REGC + S16_LDSC + k;
          const float4* cs = reinterpret_cast<const float4*>(L + S16_CS + 16 * ci);
          R.s0 = cs[0]; R.sk = cs[1]; R.kl = cs[2];
          R.tq = *reinterpret_cast<const float2*>(L + S16_CS + 16 * ci + 14);
          con_apply(R, ci, use_bias);
        };
        ConRec A, B, C;
        gload(0, A); gload(1, B); gload(2, C);
        int k = 0;
        while (true) {
          gapply(A, k); gload(k + 3, A);
          if (++k >= max_cglb) break;
          gapply(B, k); gload(k + 3, B);
          if (++k >= max_cglb) break;
          gapply(C, k); gload(k + 3, C);
          if (++k >= max_cglb) break;
        }
      }
      PH(20);
      // joint-limit rows, after the contacts of the sweep (an articulation's internal constraints are solved after
      // its contacts, as in PhysX); exact sequential Gauss-Seidel semantics over the joints in order: lane j evaluates
      // its own row against the current v (J is +-1 at lane j, no reduction), d(lambda) reaches the env's lanes by DPP
      // row_newbcast:j and each applies its entry of W. A row that does not change contributes d(lambda) = 0 exactly, so
      // every joint is visited -- ~7 VALU instructions each, no LDS or ballot on the dependent chain (the previous
      // version walked the changing rows only, ~800 cycles per visit: a third of the time of the slowest waves) --
      // and the pass is skipped when no row of the wave would change.
#ifndef EXP_NO_LIMROWS  // (timing experiments only)
      {
        const float bl = use_bias ? lim_bpos : lim_bvel;
        if (__any(fmaxf(lim_lam - (lim_side * v_c + bl) * lim_inv, 0.f) != lim_lam)) {
          BT_ADD(31, 1);
          constexpr int NL = NDOF ? NDOF : 16;
          float wj[NL];
#pragma unroll
          for (int j = 0; j < NL; j++) wj[j] = (lead && j < n) ? L[S16_LIMW + 16 * j + c] : 0.f;  // (W of a limit row is zero outside the articulation's block)
          limit_rows_pass<0, NL>(wj, v_c, lim_lam, lim_side, lim_inv, bl, c, n);
        }
      }
#endif
      }  // run
      if (use_bias) settled = settled || benv(fabsf(v_c - v_before) > MSSIM_PGS_EXIT_TOLERANCE) == 0u;
      PH(18);
    }
    if (c == 0 && lead) {
#pragma unroll
      for (int k = 0; k < S16_REGC; k++)
        if (k < nc_reg) { L[S16_CS + 16 * k + 9] = lamr[k][0]; L[S16_CS + 16 * k + 10] = lamr[k][1]; L[S16_CS + 16 * k + 11] = lamr[k][2]; }
    }
    WSYNC();
    // the multipliers of this substep, keyed by (shape pair, manifold slot) and stamped: the next substep's warm start
    for (int i = lead ? c : nc; i < nc; i += 16) {
      const int pw = __float_as_int(L[S16_REC + S16_REC_LEN * i + 7]);
      if (((pw >> 30) & 1) || !live || (TRI && ((pw >> 27) & 1))) continue;
      *reinterpret_cast<float4*>(S.warm + (((size_t)(4 * (pw & 0xFFFF) + ((pw >> 24) & 3))) * N + e) * 4) =
          float4{L[S16_CS + 16 * i + 9], L[S16_CS + 16 * i + 10], L[S16_CS + 16 * i + 11], __int_as_float(pcm_tick)};
    }
    PH(5);

    // ================================================================ contact impulses per pair (last substep)
    if (last) {
      // FUSED keeps the dense pair_cnt / pair_imp arrays valid with sparse updates: pairs of the
      // previous step's hit list that are no longer in contact are zeroed, the new list is written
      int* const newl = reinterpret_cast<int*>(L + S16_VEC);  // up to MAXC pair ids
      int nnew = 0;
      int prev_p = -1, run = 0;
      f3 acc = f3{0, 0, 0};
      auto flush = [&]() __attribute__((always_inline)) {
        if (prev_p >= 0 && live) {
          if (c == 0 && lead) { SOA(S.pair_imp, 3 * prev_p) = acc.x; SOA(S.pair_imp, 3 * prev_p + 1) = acc.y; SOA(S.pair_imp, 3 * prev_p + 2) = acc.z; }
          if (FUSED && c == 0 && lead) { S.pair_cnt[(size_t)prev_p * N + e] = run; S.hit_list[(size_t)(1 + nnew) * N + e] = prev_p; }
        }
        if (FUSED && prev_p >= 0) { if (c == 0 && lead) newl[nnew] = prev_p; nnew++; }
      };
      for (int i = 0; i < nc; i++) {
        const float l0 = L[S16_CS + 16 * i + 9], l1 = L[S16_CS + 16 * i + 10], l2 = L[S16_CS + 16 * i + 11];
        const float* rec = L + S16_REC + S16_REC_LEN * i;
        const int pw = __float_as_int(rec[7]);
        if ((pw >> 30) & 1) continue;  // torsional block: a pure torque, no part of the pair's force
        const int p = pw & 0xFFFF;
        const f3 nrm = f3{rec[0], rec[1], rec[2]};
        const f3 t1 = fabsf(nrm.x) < 0.57735f ? normalized(cross(nrm, f3{1, 0, 0})) : normalized(cross(nrm, f3{0, 1, 0}));
        const f3 t2 = cross(nrm, t1);
        const f3 imp = nrm * l0 + t1 * l1 + t2 * l2;
        if (p != prev_p) {
          flush();
          acc = f3{0, 0, 0};
          run = 0;
          prev_p = p;
        }
        acc += imp;
        run++;
      }
      flush();
      if (FUSED) {
        WSYNC();
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int po = oldp[k];
          bool still = po < 0;
          for (int j = 0; j < nnew; j++) still = still || (newl[j] == po);
          if (!still && live) S.pair_cnt[(size_t)po * N + e] = 0;
        }
        if (c == 0 && lead && live) S.hit_list[e] = nnew;
        WSYNC();
      }
    }
    PH(6);

    // ================================================================ FK at the new state
    // FK by pointer jumping: lane c starts from its joint-local transform T_c (parent body -> body c)
    // and composes with its ancestor's partial product, doubling the covered chain length per round
    // (4 rounds cover depth 16); sincos and the products run in all lanes at once instead of a
    // 9-long sequential chain. Same transforms as the sequential product, different association.
    WSYNC();
    const pose_t rootf = lds_pose(L + S16_PT);
    pose_t nb = rootf;
    f3 naw = f3{0, 0, 0}, nan = f3{0, 0, 0};
    EXP_DUP(DUP_FK) {
      pose_t T = pose_t{f3{0, 0, 0}, q4{1, 0, 0, 0}};
      int up = -1;
      if (art) {
        if (rev_c) T = pose_t{JF_c.p, qnormalized(qmul(JF_c.q, qaxis_angle(al_c, q_c)))};
        else T = pose_t{JF_c.p + qrot(JF_c.q, al_c) * q_c, JF_c.q};
        up = par_c;
      }
      int* const upv = reinterpret_cast<int*>(L + S16_VEC + 48);
      float* const mine = L + S16_BP + 7 * c;
      for (int round = 0; round < 4; round++) {
        if (!__any(up >= 0)) break;
        if (art) lds_pose_store(mine, T);
        if (lead) upv[c] = up;
        WSYNC();
        if (up >= 0) {
          const pose_t Tu = lds_pose(L + S16_BP + 7 * up);
          up = upv[up];
          T = pmul(Tu, T);
        }
        WSYNC();
      }
      nb = pmul(rootf, T);
      if (art) lds_pose_store(mine, nb);
      WSYNC();
      pose_t Wp = rootf;
      if (par_c >= 0) Wp = lds_pose(L + S16_BP + 7 * par_c);
      const pose_t Jw = pmul(Wp, JF_c);
      naw = qrot(Jw.q, al_c);
      nan = Jw.p;
    }
    // carry the state into the next substep (joint velocities within MSSIM_MAX_JOINT_VELOCITY)
    {
      const float vj = art ? fminf(fmaxf(v_c, -MSSIM_MAX_JOINT_VELOCITY), MSSIM_MAX_JOINT_VELOCITY) : 0.f;
      qacc_c = (vj - qd_c) * inv_dt;
      qd_c = vj;
    }
    vfree_c = freel ? v_c : 0.f;
    // sleep counters: run down while the body is calm and no disturber touches it, restart otherwise. Whether a body is calm is
    // seen by its own row; every lane of the env keeps the counters
    bool calm_l[2] = {false, false};
#pragma unroll
    for (int bl = 0; bl < 2; bl++) {
      if (bl >= nfr) break;
      const int base = fcol0 + 6 * bl;
      const f3 vv = f3{gbc(v_c, base), gbc(v_c, base + 1), gbc(v_c, base + 2)}, ww = f3{gbc(v_c, base + 3), gbc(v_c, base + 4), gbc(v_c, base + 5)};
      const float* pt = L + S16_PT + 7 * (S16_PT_FREE + fb0 + bl);
      calm_l[bl] = norm_energy(fin[bl], vv, ww, q4{pt[3], pt[4], pt[5], pt[6]}) < M.sleep_threshold;
    }
#pragma unroll
    for (int b = 0; b < S16_MAX_FREE; b++) {
      if (b >= nf) break;
      bool calm = (b & 1) ? calm_l[1] : calm_l[0];
      if constexpr (NR > 1) calm = env_bci(calm ? 1 : 0, 16 * (1 + (b >> 1))) != 0;  // (from the row that holds body b)
      if (fwake[b] > 0.f) {
        fwake[b] = (calm && M.sleep_threshold > 0.f && !((fdist >> b) & 1u)) ? fwake[b] - dt : MSSIM_WAKE_TIME;
        if (fwake[b] <= 0.f) {
          fwake[b] = 0.f;
          if (freel && fb_id == b) { vfree_c = 0.f; v_c = 0.f; }
        }
      }
      fcalm[b] = calm;
    }

    PH(15);
      // ================================================================ write back (last substep)
    if (last) {
      if (art && live) {
        SOA(S.qacc, c) = qacc_c;
        SOA(S.q, c) = q_c;
        SOA(S.qd, c) = qd_c;
      }
      if (freel && live) {
        SOA(S.free_s, 13 * fb_id + 7 + fk) = v_c;
        if (fk < 3) SOA(S.free_force, 3 * fb_id + fk) = 0.f;
      }
      if (c < nf && lead && live) SOA(S.free_wake, c) = fw_at(c);
      if (c == 0 && lead && live) S.pcm_tick[e] = pcm_tick;
      // body velocities about O with the new subspaces
      sv6 nS = sv6{f3{0, 0, 0}, f3{0, 0, 0}};
      if (art) nS = rev_c ? sv6{naw, cross(nan - O, naw)} : sv6{f3{0, 0, 0}, naw};
      WSYNC();
      if (lead) {
        float* p = L + S16_S + 6 * c;
        p[0] = nS.w.x; p[1] = nS.w.y; p[2] = nS.w.z; p[3] = nS.v.x; p[4] = nS.v.y; p[5] = nS.v.z;
        L[S16_VEC + c] = qd_c;
      }
      WSYNC();
      sv6 nV = sv6{f3{0, 0, 0}, f3{0, 0, 0}};
      for (int i = 0; i < n; i++) {
        float m = (((anc_c | self_c) >> i) & 1u) ? L[S16_VEC + i] : 0.f;
        const float* p = L + S16_S + 6 * i;
        nV.w += f3{p[0], p[1], p[2]} * m;
        nV.v += f3{p[3], p[4], p[5]} * m;
      }
      PH(16);
      if (art && live) {
        pose_store_soa(S.bodypose, 7 * c, N, e, nb);
        float* o = S.bodyvel + (size_t)(6 * c) * N + e;
        o[0] = nV.w.x; o[(size_t)N] = nV.w.y; o[2 * (size_t)N] = nV.w.z;
        o[3 * (size_t)N] = nV.v.x; o[4 * (size_t)N] = nV.v.y; o[5 * (size_t)N] = nV.v.z;
        float* a = S.bodyaux + (size_t)(6 * c) * N + e;
        a[0] = naw.x; a[(size_t)N] = naw.y; a[2 * (size_t)N] = naw.z;
        a[3 * (size_t)N] = nan.x; a[4 * (size_t)N] = nan.y; a[5 * (size_t)N] = nan.z;
      }
    }
  }

  (void)nrow_con;
  PH(7);
  // ================================================================ copy-out + task epilogue of this env
  if (FUSED && TASK > 0) {
    // the state written above is read back below by other lanes of this wave, through the CU's own L1 (which its
    // stores keep current): a block-scope fence = wait for the stores. (A device-scope fence would write back and
    // invalidate caches -- tens of microseconds per wave here.)
    __threadfence_block();
    if (live) {
      const int R = M.n_link + M.n_free + M.n_kin;
      for (int row = c; row < R; row += 16) fetch_row(M, S, S.tail_buf, S.tail_fetch, e, row);
      if (art) fetch_art_joint(M, S, S.tail_buf, S.tail_fetch, e, c);
    }
    __threadfence_block();
    if (live && c == 0) {
      if (TASK == 1) task_pick_env(M, S, S.tail_buf, S.tail_task.pick, S.tail_pairs, S.tail_npairs, S.tail_obs, S.tail_reward, S.tail_flags, e);
      if (TASK == 2) task_push_env(M, S, S.tail_buf, S.tail_task.push, S.tail_obs, S.tail_reward, S.tail_flags, e);
      if (TASK == 3) task_peg_env(M, S, S.tail_buf, S.tail_task.peg, S.tail_pairs, S.tail_npairs, S.tail_obs, S.tail_reward, S.tail_flags, S.tail_head, e);
    }
    PH(8);
  }
  PH_FLUSH
}
