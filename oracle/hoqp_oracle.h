/*
 * hoqp_oracle.h -- interface of the CPU restatement (TEST INFRASTRUCTURE ONLY,
 * PARITY UNPINNED; see hoqp_oracle.c).  Plain C, ctypes-friendly.
 */
#ifndef HOQP_ORACLE_H
#define HOQP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	double eps;		/* eps_ = 1e-12, ho_qp_problem.hpp:56 */
	int gate_prior_ineq;	/* Q2: 1 = reference gating, ho_qp_problem.cpp:297-298 */
	int slack_order_quirk;	/* Q3: 1 = reference ordering, ho_qp_problem.cpp:83-101 */
	int zero_kernel_col;	/* Q6: 1 = Eigen kernel() returns one zero column */
	int nullspace_basis;	/* 0 = FullPivLU kernel (math.hpp:99-104), 1 = Householder */
	int max_iter;
	int solver;		/* 1 = least-squares form in long double (authoritative); 0 = literal H + eps I, double */
} hoqp_oracle_cfg;

typedef struct {
	double mu, tau_max, tau_min;		/* ho_qp_controller.hpp:54-56 */
	double kp_swing, kv_swing;		/* ho_qp_controller.cpp:191-192 */
	double kp_base, kv_base, kv_rot;	/* :224-225, :259-260 */
	int friction_col_quirk;			/* Q1, :321-324 */
	int drop_base_pos_gain;			/* Q5, :235-237 */
} hoqp_oracle_wbc_params;

/* what UpdateModelDynamics (ho_qp_controller.cpp:89-114) and the Set*Cmd calls hand over */
typedef struct {
	double M[18 * 18];	/* GetMassMatrix */
	double h[18];		/* GetBiasVector */
	double Jfoot[4 * 3 * 18]; /* translational foot Jacobians, all four legs LF,RF,LH,RH */
	double Jdotu[4 * 3];	/* J_dot u per foot */
	double footpos[4 * 3];	/* foot positions in W */
	double Jb[6 * 18];	/* base spatial Jacobian, rows 0-2 rot, 3-5 pos */
	double q[19];		/* quat wxyz, pos, joints */
	double u[18];		/* omega, v, joint vel */
	double base_cmd[9];	/* pos, vel, acc */
	double leg_cmd[36];	/* pos(12), vel(12), acc(12), leg-major */
	int contact[4];		/* 1 = contact, 0 = swing (ho_qp_controller.cpp:426-450) */
} hoqp_oracle_wbc_state;

typedef struct {
	int n, nv, status;
	double x[30];
	double x_levels[5 * 30];
	double qdd[12], tau[12];
	double v[64];
	int nz_levels[5], iters[5];
	int flags[5];
	double eq_res_max[5], ineq_viol_max[5], level_res[5];
	unsigned long long active_mask;
} hoqp_oracle_wbc_result;

void hoqp_oracle_default_cfg(hoqp_oracle_cfg *cfg);
void hoqp_oracle_default_wbc_params(hoqp_oracle_wbc_params *prm);

int hoqp_oracle_qp(int N, const double *H, const double *c, int Mc, const double *C, const double *d,
		   double *xi, double *lam, int maxit, int *iters);

int hoqp_oracle_fullpivlu_kernel(int r, int c, const double *A, double *Kout, int zero_col);

int hoqp_oracle_solve_tasks(const hoqp_oracle_cfg *cfg, int n, int K, const int *m, const int *p,
			    const double *A, const double *b, const double *D, const double *f,
			    double *x_levels, int *nz_levels, double *Z_levels, double *v_levels,
			    int *iters, int *status);

int hoqp_oracle_wbc_build(const hoqp_oracle_wbc_params *prm, const hoqp_oracle_wbc_state *s, int *m,
			  int *p, double *A, double *b, double *D, double *f);

int hoqp_oracle_wbc_solve(const hoqp_oracle_cfg *cfg, const hoqp_oracle_wbc_params *prm,
			  const hoqp_oracle_wbc_state *s, hoqp_oracle_wbc_result *out);

int hoqp_oracle_wbc_solve_batch(const hoqp_oracle_cfg *cfg, const hoqp_oracle_wbc_params *prm, int N,
				const hoqp_oracle_wbc_state *s, hoqp_oracle_wbc_result *out, int threads);

/* Post-solve step: Integrator::Integrate twice (integrator.cpp:31-36, called by
 * WholeBodyController::SetJointCmdFromIntegrators, whole_body_controller.cpp:148-155) and
 * JointController::CalcJointTorques (joint_controller.cpp:103-110); arrays [N][12]; q_cmd and
 * qd_cmd are updated in place; dt is an argument (the reference reads the wall clock). */
void hoqp_oracle_wbc_poststep(int N, double dt, double kp, double kd, const double *qdd, const double *tau_ff,
			      const double *q_j, const double *qd_j, double *q_cmd, double *qd_cmd, double *tau_cmd);

/* N3: command / gait sampler (leg_planner.cpp, base_planner.cpp:272-317) */
void hoqp_oracle_gait_sample(int N, int n_steps, double duration, const int *schedule, double apex,
			     int fix_derivatives, const double *time, const double *vel_cmd, const double *stance0,
			     double *leg_cmd, int *contact);
void hoqp_oracle_standup_sample(int N, double seconds, double target_height, const double *time,
				const double *curr_pose, double *base_cmd, int *contact);

#ifdef __cplusplus
}
#endif
#endif
