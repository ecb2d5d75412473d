/*
 * hoqp.h -- C ABI of the MI355X-native batched hierarchical-QP (HO-QP) solver.
 *
 * Drop-in boundary for the solve step of control::HoQpProblem / control::HoQpController
 * of bernhardpg/quadruped_locomotion.  The reference has no FFI layer for this path: the
 * boundary it exposes is the C++ class API (include/control/ho_qp/ho_qp_problem.hpp:23-43,
 * include/control/ho_qp/ho_qp_controller.hpp:27-44).  Each entry point below cites the
 * reference interface it replaces (paths relative to the reference root).  The C++ shim
 * include/hoqp.hpp re-creates TaskDefinition / HoQpProblem / HoQpController on top of it.
 *
 * All pointers inside *_batch / *_inputs / *_outputs structs are DEVICE pointers
 * (hipMalloc memory) unless stated otherwise.  Matrices are dense, row-major, fp64.
 * Batched arrays are instance-major: the block of one instance is contiguous, so the
 * one-workgroup-per-instance kernels read them with fully coalesced loads.
 *
 * Every function returns 0 on success, <0 on an argument / dimension / HIP error
 * (hoqp_last_error_string gives the text); per-instance solver outcomes go to the
 * `status` output array, never to abort()/assert (the reference only asserts,
 * src/control/ho_qp/ho_qp_problem.cpp:348).
 */
#ifndef HOQP_H
#define HOQP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HOQP_MAX_LEVELS 8   /* priority levels per hierarchy */
#define HOQP_MAX_VARS 32    /* decision variables n (ANYmal: 18 + 3 n_c <= 30) */
#define HOQP_MAX_EQ_ROWS 32 /* equality rows of one level */
#define HOQP_MAX_INEQ_ROWS 64 /* inequality rows summed over all levels */

/* Semantics switches: the reference's behavioural quirks. */
#define HOQP_Q1_FRICTION_COLS 0x01 /* friction pyramid on columns 6+3*leg, ho_qp_controller.cpp:321-324 */
#define HOQP_Q2_GATE_PRIOR_INEQ 0x02 /* prior inequality rows only if the level has its own, ho_qp_problem.cpp:297-298 */
#define HOQP_Q3_SLACK_ORDER 0x04 /* rows newest-first vs slacks oldest-first, ho_qp_problem.cpp:83-101 */
#define HOQP_Q5_DROP_BASE_POS_GAIN 0x10 /* stray ';' at ho_qp_controller.cpp:236 */
#define HOQP_Q6_ZERO_KERNEL_COL 0x20 /* Eigen kernel() of a full-column-rank matrix (shim level only) */
#define HOQP_SEMANTICS_REFERENCE 0x37
#define HOQP_SEMANTICS_PAPER 0x00

/* per-instance status bits */
#define HOQP_STATUS_OK 0
#define HOQP_STATUS_MAX_ITER 0x01	 /* active-set iteration cap hit at some level (iterate is feasible) */
#define HOQP_STATUS_INFEASIBLE_START 0x02 /* a prior (hard) row was violated when a level started */
#define HOQP_STATUS_WORKSPACE 0x04	 /* working set outgrew the LDS workspace */
#define HOQP_STATUS_NONFINITE 0x08	 /* NaN/Inf in the final x, a slack, a level residual, a validity residual or tau */
#define HOQP_STATUS_NEAR_SINGULAR 0x10 /* reference semantics: a level's projected block is so close to rank loss
					  (eps > 3e-8 d^2, d = smallest pivot of its factorization ~ sigma_min) that its
					  eps-regularised solve must be replayed literally in the reference's LU bases,
					  and the replay could not be done (a level with more than 12 independent rows,
					  a QP level with equality rows before it, rank decisions of LU and QR differ).
					  x then carries the first-order eps correction in the solver's own orthonormal
					  basis: it deviates from the reference's x by ~ eps / sigma_min^2 of the level's
					  step, which can exceed the 1e-6 parity bar.  At level 0 (identity basis on
					  both sides) the bit means eps > 1e-4 d^2.  Informational: x is the solver's
					  best answer, every other output is consistent with it. */
#define HOQP_STATUS_BAD_EXTENT 0x20	 /* n_inst / m_inst / p_inst beyond the storage extent: clamped to it */
#define HOQP_STATUS_BASE_RANK 0x40	 /* hoqp_solve_wbc_batched (either semantics; compact LDS layouts): [M_fb, -J_c,fb^T] (the floating-base
					  * block of ho_qp_controller.cpp:351-366) has rank < 6 -- not a physical mass matrix.
					  * The state is not solved (x, tau, slacks = NaN); hoqp_solve_wbc_unfused takes any state */

typedef struct hoqp_handle hoqp_handle;

typedef struct {
	int device;	/* HIP device ordinal the handle is bound to */
	int semantics;	/* HOQP_Q* bitmask */
	double eps;	/* eps_ of ho_qp_problem.hpp:56 (1e-12) */
	int max_iter;	/* active-set iterations per level */
	/* HoQpController constants, ho_qp_controller.hpp:54-56 and the gains of
	 * ho_qp_controller.cpp:191-192, 224-225, 259-260 */
	double mu, tau_max, tau_min;
	double kp_swing, kv_swing, kp_base, kv_base, kv_rot;
	int options;	/* HOQP_OPT_* bitmask, 0 = bit-compatible default */
} hoqp_config;

/* Skip a level that has only inequality rows when nothing can observe its result: under Q2
 * (ho_qp_problem.cpp:297-298) no later level looks at its rows, and the caller passed v == NULL, so its
 * slacks are not read either.  Such a level only moves x inside the null space of the levels above it; when
 * the hierarchy ends with an empty null space (the WBC hierarchy) the final x, tau and validity flags are
 * the same to ~1e-7 relative (the eps_ bias of the later levels depends on where in that null space x sits),
 * while x_levels of the skipped level keep the previous level's x.  In the
 * reference's WBC hierarchy this removes the torque / friction level's QP (about 30 % of the solve).
 * Honoured by hoqp_solve_wbc_batched / hoqp_solve_wbc_host only: a generic hierarchy
 * (hoqp_solve_tasks_batched, and the unfused / allocation-free WBC entries built on it) need not end with an
 * empty null space, so the option is ignored there. */
#define HOQP_OPT_SKIP_UNREAD_SLACK_LEVELS 0x01

/* Replaces: the compile-time constants of HoQpController / HoQpProblem. */
void hoqp_default_config(hoqp_config *cfg);

/* Replaces: HoQpController::HoQpController (ho_qp_controller.cpp:5).  One handle per host
 * thread and GPU; handles share no mutable state except the scratch array of
 * hoqp_solve_wbc_noalloc.  hoqp_solve_wbc_batched and hoqp_solve_tasks_batched keep no state in the
 * handle and may be issued to any number of streams; the host-buffer entries and
 * hoqp_solve_wbc_unfused use handle-owned staging and must be issued one at a time per handle.
 * Every entry point makes the handle's device current for its own duration and restores the
 * caller's current device before it returns. */
int hoqp_create(const hoqp_config *cfg, hoqp_handle **out);
int hoqp_destroy(hoqp_handle *h);
const char *hoqp_last_error_string(const hoqp_handle *h);

/*
 * A batch of task hierarchies: level k of instance i is TaskDefinition{A,b,D,f}
 * (include/control/ho_qp/task_definition.hpp:5-11).  Storage of level k:
 *   A[k]: [batch][m[k]][n]   b[k]: [batch][m[k]]   D[k]: [batch][p[k]][n]   f[k]: [batch][p[k]]
 * m[k] / p[k] / n are the STORAGE extents (0 = level has no such block).  If n_inst /
 * m_inst / p_inst are non-NULL, instance i uses only the leading n_inst[i] columns and
 * m_inst[i*num_levels+k] / p_inst[...] rows (variable constraint counts per instance).
 */
typedef struct {
	int batch, n, num_levels;
	int m[HOQP_MAX_LEVELS], p[HOQP_MAX_LEVELS];
	const double *A[HOQP_MAX_LEVELS], *b[HOQP_MAX_LEVELS];
	const double *D[HOQP_MAX_LEVELS], *f[HOQP_MAX_LEVELS];
	const int *n_inst, *m_inst, *p_inst;
} hoqp_task_batch;

/* Any pointer may be NULL (that output is skipped).  P = sum_k p[k]. */
typedef struct {
	double *x;	   /* [batch][n]   HoQpProblem::GetSolution of the last level (ho_qp_problem.cpp:46-51) */
	double *x_levels;  /* [batch][L][n] GetSolution of every level */
	double *v;	   /* [batch][P]   GetStackedSlackSolutions, oldest level first (:58-61, 88-101) */
	double *level_res; /* [batch][L]   |A_k x_k - b_k|_2 at the level's own optimum */
	double *Z_levels;  /* [batch][L][n][n] GetStackedNullspaceMatrix (leading nz columns, orthonormal basis) */
	int *nz_levels;	   /* [batch][L]   columns of Z_k */
	int *iters;	   /* [batch][L]   active-set iterations */
	int *status;	   /* [batch]      HOQP_STATUS_* */
	/* CheckSolutionValid at the FINAL x (include/helper_functions.hpp:63-91): */
	int *flags;	   /* [batch][L] 1 = "Solution valid" (|Ax-b| <= 1e-7 and Dx-f <= 0) */
	double *eq_res_max;    /* [batch][L] max |A_k x - b_k| */
	double *ineq_viol_max; /* [batch][L] max (D_k x - f_k) */
	unsigned long long *active_mask; /* [batch] bit r: stacked inequality row r has D x - f >= -1e-7 */
} hoqp_result_batch;

/*
 * Replaces: the chain of HoQpProblem constructors built by
 * HoQpController::ConstructOptProblems (ho_qp_controller.cpp:120-141), i.e. for every
 * level the pipeline of HoQpProblem::HoQpProblem (ho_qp_problem.cpp:11-25), including
 * drake::solvers::Solve (:339) and CalcNullSpaceProjMatrix (math.hpp:99-104).
 * Asynchronous on `stream` (a hipStream_t, may be NULL).
 */
int hoqp_solve_tasks_batched(hoqp_handle *h, const hoqp_task_batch *tasks, const hoqp_result_batch *out,
			     void *stream);

/*
 * What HoQpController::UpdateModelDynamics (ho_qp_controller.cpp:89-114), SetBaseCmd
 * (:27-36) and SetLegCmd (:38-65) hand to the solve, per instance:
 *   M [18][18], h [18], Jfoot [4][3][18] (translational foot Jacobians, legs LF,RF,LH,RH),
 *   Jdotu [4][3], footpos [4][3], Jb [6][18] (rows 0-2 rot, 3-5 pos; NULL = [I6 0]),
 *   q [19], u [18], base_cmd [9] (pos,vel,acc), leg_cmd [36] (pos,vel,acc x 12),
 *   contact [4] ints (1 = contact, 0 = swing).
 */
typedef struct {
	int batch;
	const double *M, *h, *Jfoot, *Jdotu, *footpos, *Jb, *q, *u, *base_cmd, *leg_cmd;
	const int *contact;
} hoqp_wbc_inputs;

typedef struct {
	double *x;	/* [batch][30] solution_, zero padded beyond n = 18 + 3 n_c */
	double *qdd;	/* [batch][12] GetJointAccelerationCmd (ho_qp_controller.cpp:23, 80-83) */
	double *tau;	/* [batch][12] GetJointTorqueCmd (:67-78) */
	double *v;	/* [batch][40] slack of the torque/friction level */
	int *flags;	/* [batch][5]  the five CheckSolutionValid results of CalcJointCmd (:18-22) */
	double *eq_res_max, *ineq_viol_max, *level_res; /* [batch][5] */
	unsigned long long *active_mask;		/* [batch] */
	int *status, *iters, *nz_levels;		/* [batch], [batch][5], [batch][5] */
	double *x_levels;				/* [batch][5][30] */
} hoqp_wbc_outputs;

/*
 * Replaces: HoQpController::CalcJointCmd after UpdateModelDynamics
 * (ho_qp_controller.cpp:7-25): ConstructTasks (:147-184) with its nine builders
 * (:186-400), ConstructOptProblems (:120-141), the five CheckSolutionValid calls and
 * CalcJointTorquesCmd (:67-73).  Mixed contact patterns are handled per instance.
 */
int hoqp_solve_wbc_batched(hoqp_handle *h, const hoqp_wbc_inputs *in, const hoqp_wbc_outputs *out, void *stream);

/*
 * Replaces: the same span of HoQpController::CalcJointCmd as hoqp_solve_wbc_batched, as three
 * kernels instead of one: ConstructTasks (ho_qp_controller.cpp:147-184) materialised as dense task
 * blocks in a handle-owned, grow-only workspace, the generic level chain on them
 * (ConstructOptProblems, :120-141), then CalcJointTorquesCmd (:67-73).  Same results; kept as the
 * cross-check of the fused kernel.  One stream at a time per handle (the workspace is shared).
 */
int hoqp_solve_wbc_unfused(hoqp_handle *h, const hoqp_wbc_inputs *in, const hoqp_wbc_outputs *out, void *stream);

/*
 * No counterpart in the reference.  Allocation-free variant of hoqp_solve_wbc_unfused for callers
 * that cannot allocate: code under HIP graph capture, tight control / serving loops, harnesses that
 * forbid hipMalloc.  It calls no hipMalloc / hipFree / event creation; the task blocks go to a
 * file-scope device array of HOQP_NOALLOC_MAX_BATCH * 21,868 bytes (5.6 MB) that every GPU the
 * library runs on holds until the process ends.  batch > HOQP_NOALLOC_MAX_BATCH is rejected (-4).
 * All calls on one GPU share that array: calls that overlap on one GPU (two streams, two handles on
 * the same device) race; serialise them.  Results are those of hoqp_solve_wbc_unfused.
 */
#define HOQP_NOALLOC_MAX_BATCH 256
int hoqp_solve_wbc_noalloc(hoqp_handle *h, const hoqp_wbc_inputs *in, const hoqp_wbc_outputs *out, void *stream);

/*
 * Replaces: HoQpController::ConstructTasks (ho_qp_controller.cpp:147-184) alone -- the five
 * TaskDefinitions of every instance written to caller-owned device storage `tasks` (a
 * hoqp_task_batch with n = 30, num_levels = 5, m = {6,0,12,18,12}, p = {0,40,0,0,0}, A[0], b[0], D[1],
 * f[1], A[2..4], b[2..4] and n_inst / m_inst / p_inst all non-NULL and writable).  Rows beyond an
 * instance's own count are zeroed.  The result can be handed to hoqp_solve_tasks_batched as is.
 */
int hoqp_wbc_build_tasks_batched(hoqp_handle *h, const hoqp_wbc_inputs *in, const hoqp_task_batch *tasks, void *stream);

/*
 * Host-buffer variants: every pointer in the structs is a HOST pointer; the library stages
 * through device memory, runs the same kernels and blocks until the results are back.
 * This is the form a drop-in for the single-instance reference API needs: HoQpProblem /
 * HoQpController hand over host Eigen matrices (ho_qp_problem.hpp:23-27,
 * ho_qp_controller.hpp:27-41).  PCIe-inclusive; not the benchmarked path.
 */
int hoqp_solve_tasks_host(hoqp_handle *h, const hoqp_task_batch *tasks, const hoqp_result_batch *out);
int hoqp_solve_wbc_host(hoqp_handle *h, const hoqp_wbc_inputs *in, const hoqp_wbc_outputs *out);

/*
 * Dynamics-terms producer: what HoQpController::UpdateModelDynamics
 * (ho_qp_controller.cpp:89-114) pulls from the reference's Dynamics object -- GetMassMatrix,
 * GetBiasVector, GetStackedContactJacobianInW, GetStackedContactAccInW, foot positions,
 * GetBaseSpatialJacobian (dynamics.cpp:75-124, 130-152, 179-228) -- batched on the device for a
 * quadruped with 3 joints per leg (HAA about body x, HFE and KFE about y).  The reference gets
 * these from Drake's MultibodyPlant on Drake's ANYmal-C URDF; neither exists here, so the model
 * parameters are this library's own (defaults: an approximate 50 kg ANYmal-C, documented in
 * quadruped_locomotion_amd/anymal_model.py) and parity of the MODEL is unpinned.
 * Offsets and centres of mass are given for the LF leg; x is mirrored for hind legs (HAA offset
 * only), y for right legs.  Inertias are principal moments in the body frame.
 */
typedef struct {
	double base_mass, hip_mass, thigh_mass, shank_mass;
	double base_inertia[3], hip_inertia[3], thigh_inertia[3], shank_inertia[3];
	double haa_off[3], hfe_off[3], kfe_off[3], foot_off[3];
	double hip_com[3], thigh_com[3], shank_com[3];
	double gravity;
} hoqp_quadruped_model;

void hoqp_default_quadruped_model(hoqp_quadruped_model *model);

/* q [batch][19] = quat wxyz, base position, joints; u [batch][18] = omega_W, v_W, joint rates
 * (anymal_plugin.cpp:266-290).  Outputs in the layout of hoqp_wbc_inputs: M [batch][18][18],
 * hvec [batch][18] (gravity included iff include_gravity; the reference's CalcBiasTerm has none,
 * dynamics.cpp:82-87), Jfoot [batch][4][3][18], Jdotu [batch][4][3], footpos [batch][4][3],
 * Jb [batch][6][18] (may be NULL).  model == NULL: defaults.  Asynchronous on `stream`. */
int hoqp_wbc_dynamics_batched(hoqp_handle *h, const hoqp_quadruped_model *model, int batch, const double *q,
			      const double *u, int include_gravity, double *M, double *hvec, double *Jfoot,
			      double *Jdotu, double *footpos, double *Jb, void *stream);

/* Host-buffer variant of hoqp_wbc_dynamics_batched (all pointers are host pointers; blocks). */
int hoqp_wbc_dynamics_host(hoqp_handle *h, const hoqp_quadruped_model *model, int batch, const double *q,
			   const double *u, int include_gravity, double *M, double *hvec, double *Jfoot, double *Jdotu,
			   double *footpos, double *Jb);

/* Post-solve step of the whole-body controller, batched and device resident:
 * the two forward-Euler integrations that turn the joint acceleration command into velocity and
 * position commands (WholeBodyController::SetJointCmdFromIntegrators,
 * whole_body_controller.cpp:148-155; Integrator::Integrate, integrator.cpp:31-36) followed by the
 * joint PD + feed-forward torque (JointController::CalcJointTorques, joint_controller.cpp:103-110):
 *     qd_cmd += dt * qdd;   q_cmd += dt * qd_cmd;                 (in place, the integrator state)
 *     tau_cmd = kp * (q_cmd - q_j) + kd * (qd_cmd - qd_j) + tau_ff
 * All arrays are [batch][12] doubles on the device.  `qdd` and `tau_ff` are the outputs `qdd` and
 * `tau` of hoqp_solve_wbc_batched; q_j / qd_j are the measured joint positions / velocities
 * (q[7:19], u[6:18] of the WBC inputs).  The reference takes dt from ros::Time (wall clock);
 * here it is an argument.  tau_cmd may be NULL (integration only).  Asynchronous on `stream`. */
int hoqp_wbc_poststep_batched(hoqp_handle *h, int batch, double dt, double kp, double kd, const double *qdd,
			      const double *tau_ff, const double *q_j, const double *qd_j, double *q_cmd,
			      double *qd_cmd, double *tau_cmd, void *stream);

/*
 * SURVEY 8f N3 -- the command source as a batch sampler.
 * hoqp_gait replaces GaitSequence (include/planner/gait_sequence.hpp:3-9) plus the two constants
 * of the swing trajectory.  kind 0 = MotionPlanner::CreateSimpleSequence, 1 = CreateCrawlSequence
 * (motion_planner.cpp:536-575).
 */
#define HOQP_MAX_GAIT_STEPS 32
typedef struct {
	int n_steps;		  /* <= HOQP_MAX_GAIT_STEPS */
	double duration;	  /* [s]; step_time = duration / n_steps */
	unsigned contact_bits[4]; /* bit s of leg l = contact_schedule(l, s); legs LF, RF, LH, RH */
	double apex_height;	  /* z_apex_height of LegPlanner::CreateZLegTrajectory (leg_planner.cpp:241) */
	int fix_derivatives;	  /* 0: as the reference, velocity and acceleration commands evaluate the
				   * POSITION curves (leg_planner.cpp:337-341, 358-362); 1: their derivatives */
} hoqp_gait;
int hoqp_default_gait(int kind, hoqp_gait *g);

/* Replaces: LegPlanner::PlanLegsMotion (leg_planner.cpp:3-12) for a plan made at time 0 from the foot
 * positions stance0 [batch][4][2] under velocity command vel_cmd [batch][2], then MotionPlanner::
 * UpdateWalkLegCmd (motion_planner.cpp:75-81) at time[batch]: leg_cmd [batch][36] = GetStackedLeg
 * {Pos,Vel,Acc}AtT, contact [batch][4] = GetContactPatternAtT.  Device pointers; asynchronous. */
int hoqp_gait_sample_batched(hoqp_handle *h, const hoqp_gait *g, int batch, const double *time, const double *vel_cmd,
			     const double *stance0, double *leg_cmd, int *contact, void *stream);

/* Replaces: BasePlanner::PlanBaseStandupMotion + EvalStandup{Pos,Vel,Acc}TrajAtT
 * (base_planner.cpp:272-317) and MotionPlanner::UpdateStandupCmd (motion_planner.cpp:53-59):
 * base_cmd [batch][9] = (pos, vel, acc) of the ramp from curr_pose [batch][3] to the target height,
 * contact [batch][4] = all ones (may be NULL). */
int hoqp_standup_sample_batched(hoqp_handle *h, int batch, const double *time, double seconds_to_standup,
				double target_height, const double *curr_pose, double *base_cmd, int *contact,
				void *stream);

/* SURVEY 8f N4.  Replaces: BasePlanner::PlanBaseWalkMotion (base_planner.cpp:7-19: the spline QP of
 * FormulateOptimizationProblem, :162-266, n_segments quintic 2-D segments of unit duration from
 * curr_pos [batch][2] to final_pos [batch][2] = the centroid of the last support polygon, :254-256) and
 * EvalWalkTraj{Pos,Vel,Acc}AtT (:34-47, 100-142) at time [batch]: base_cmd [batch][9].
 * fix_layout 0: as the reference, the solution is read back with rows and columns of the coefficient
 * matrix exchanged (GeneratePolynomials, :77-98); 1: the trajectory that was optimised. */
int hoqp_base_walk_sample_batched(hoqp_handle *h, int batch, const double *time, const double *curr_pos,
				  const double *final_pos, double walking_height, int n_segments, int fix_layout,
				  double *base_cmd, void *stream);

/* Blocks until `stream` has drained; returns <0 if a kernel faulted. */
int hoqp_synchronize(hoqp_handle *h, void *stream);

/* Average device time (ms) of the solve kernel over the launches recorded since the last
 * call with reset != 0; measured with HIP events on the launch stream.  Used by bench.py. */
int hoqp_kernel_time_ms(hoqp_handle *h, int reset, double *avg_ms, int *launches);

/* Device self-test of the wavefront primitives (reductions, broadcasts).  0 = pass. */
int hoqp_selftest(hoqp_handle *h);

/* LDS bytes per workgroup the solve kernel needs for these extents (0 on error). */
size_t hoqp_lds_bytes(int n, int num_levels, const int *m, const int *p);

/*
 * Multi-GPU host driver in ONE process.  No counterpart in the reference, which
 * solves one state per cycle on one thread (whole_body_controller.cpp:329-342): the batch of independent states is
 * cut into contiguous shards, one per GPU; every shard has its own worker thread, HIP stream and solver handle,
 * so the shards' launches, copies and waits never serialise on one host thread.  There is no exchange step during
 * the solve.  After it, optionally, (i) the per-instance results of every shard are copied device-to-device into
 * arrays on a root GPU at the shard's offset (hipMemcpyPeerAsync on the shard's own stream: each peer uses its own
 * xGMI link to the root -- the one-process form of the RCCL gather that bench.py's one-process-per-GPU path does) and
 * (ii) the residual statistics are reduced on each device and combined on the host (MAX / SUM).
 * devices[r] may repeat (several shards on one GPU: how the driver is tested on a one-GPU box).
 */
#define HOQP_MAX_SHARDS 16
typedef struct hoqp_sharded hoqp_sharded;

typedef struct {
	double eq_res_max, ineq_viol_max;		  /* MAX over all instances and levels */
	long long status_nonzero, iters_total, instances; /* SUM */
} hoqp_shard_stats;

typedef struct {
	int root_device;  /* device that holds the collected arrays */
	double *x;	  /* [total][30]  (any pointer may be NULL: not collected) */
	double *tau;	  /* [total][12] */
	double *v;	  /* [total][40] */
	int *status;	  /* [total] */
} hoqp_wbc_gather;

int hoqp_sharded_create(const hoqp_config *cfg, int num_shards, const int *devices, hoqp_sharded **out);
int hoqp_sharded_destroy(hoqp_sharded *s);
const char *hoqp_sharded_last_error(const hoqp_sharded *s);

/*
 * Replaces: nothing in the reference (see above); per shard it is hoqp_solve_wbc_batched.  in[r] / out[r]
 * (r < num_shards) hold device pointers ON devices[r]; in[r].batch is the shard's size (0 allowed).  Instance
 * order of the collected arrays = shard order.  `gather` and `stats` may be NULL; statistics need
 * out[r].eq_res_max, ineq_viol_max, status and iters.  Blocks until every shard, its copies and its statistics
 * have completed.  Returns 0, or the first failing shard's error code (hoqp_sharded_last_error says which).
 */
int hoqp_solve_wbc_sharded(hoqp_sharded *s, const hoqp_wbc_inputs *in, const hoqp_wbc_outputs *out,
			   const hoqp_wbc_gather *gather, hoqp_shard_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* HOQP_H */
