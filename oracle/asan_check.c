/*
 * asan_check.c -- memory-safety / undefined-behaviour run of the CPU restatement (TEST INFRASTRUCTURE).
 * Built by `make -C oracle asan_check` with -fsanitize=address,undefined (host tests under
 * ASan / UBSan) and run by tests/test_oracle.py::test_oracle_under_sanitizers: every entry point of
 * hoqp_oracle.h on deterministic pseudo-random inputs that walk the edge shapes the GPU tests use -- empty
 * levels, rank-deficient and zero blocks, n = 32 with 64 inequality rows, every contact pattern incl. flags
 * that are neither 0 nor 1, both solvers, both semantics, several OpenMP threads.  Exit status 0 = clean
 * (the sanitizers abort on the first finding).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hoqp_oracle.h"

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static double urand(void)
{
	rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
	return (double)(rng_state >> 11) / 9007199254740992.0 * 2.0 - 1.0;
}

static int finite_all(const double *v, int n)
{
	for (int i = 0; i < n; ++i)
		if (!isfinite(v[i]))
			return 0;
	return 1;
}

static int run_tasks(int n, int K, const int *m, const int *p, int lowrank, int gate, int solver)
{
	int M = 0, P = 0;
	for (int k = 0; k < K; ++k) {
		M += m[k];
		P += p[k];
	}
	double *A = calloc((size_t)(M + 1) * n, sizeof(double)), *b = calloc(M + 1, sizeof(double));
	double *D = calloc((size_t)(P + 1) * n, sizeof(double)), *f = calloc(P + 1, sizeof(double));
	for (int i = 0; i < M * n; ++i)
		A[i] = urand();
	for (int i = 0; i < M; ++i)
		b[i] = urand();
	for (int i = 0; i < P * n; ++i)
		D[i] = urand();
	for (int i = 0; i < P; ++i)
		f[i] = urand() + 1.5;
	if (lowrank && m[0] >= 3) /* third row of the first level = combination of the first two */
		for (int j = 0; j < n; ++j)
			A[2 * n + j] = A[j] - 2.0 * A[n + j];
	double *xl = calloc((size_t)K * n, sizeof(double)), *Zl = calloc((size_t)K * n * n, sizeof(double));
	double *vl = calloc(P + 1, sizeof(double));
	int *nz = calloc(K, sizeof(int)), *it = calloc(K, sizeof(int)), *st = calloc(K, sizeof(int));
	hoqp_oracle_cfg cfg;
	hoqp_oracle_default_cfg(&cfg);
	cfg.gate_prior_ineq = gate;
	cfg.slack_order_quirk = gate;
	cfg.solver = solver;
	if (!gate)
		cfg.eps = 0.0;
	(void)hoqp_oracle_solve_tasks(&cfg, n, K, m, p, A, b, D, f, xl, nz, Zl, vl, it, st);
	int ok = finite_all(xl, K * n) && finite_all(vl, P);
	free(A), free(b), free(D), free(f), free(xl), free(Zl), free(vl), free(nz), free(it), free(st);
	return ok;
}

static void make_state(hoqp_oracle_wbc_state *s, const int *contact)
{
	memset(s, 0, sizeof(*s));
	double L[18 * 18];
	for (int i = 0; i < 18 * 18; ++i)
		L[i] = 0.3 * urand();
	for (int i = 0; i < 18; ++i)
		for (int j = 0; j < 18; ++j) {
			double a = (i == j) ? 2.0 : 0.0;
			for (int k = 0; k < 18; ++k)
				a += L[i * 18 + k] * L[j * 18 + k];
			s->M[i * 18 + j] = a;
		}
	for (int i = 0; i < 18; ++i)
		s->h[i] = 5.0 * urand(), s->u[i] = 0.3 * urand();
	for (int i = 0; i < 216; ++i)
		s->Jfoot[i] = urand();
	for (int l = 0; l < 4; ++l)
		for (int d = 0; d < 3; ++d)
			s->Jfoot[(l * 3 + d) * 18 + 3 + d] = 1.0; /* d foot / d base position = I */
	for (int i = 0; i < 12; ++i)
		s->Jdotu[i] = urand(), s->footpos[i] = urand();
	for (int i = 0; i < 6; ++i)
		s->Jb[i * 18 + i] = 1.0;
	s->q[0] = 1.0;
	for (int i = 0; i < 9; ++i)
		s->base_cmd[i] = 0.5 * urand();
	for (int i = 0; i < 36; ++i)
		s->leg_cmd[i] = 0.5 * urand();
	for (int i = 0; i < 4; ++i)
		s->contact[i] = contact[i];
}

int main(void)
{
	int bad = 0;
	/* generic hierarchies: the shapes of tests/test_gpu_parity.py */
	static const int shapes[][1 + 2 * 5] = {
		{4, 3, 0, 0, 6, 3, 0, 4, 0},	       /* n = 8 */
		{4, 0, 5, 4, 0, 3, 4, 6, 0},	       /* ineq first, mixed level */
		{2, 6, 0, 2, 3},		       /* full-rank first level: empty null space (Q6) */
		{4, 4, 6, 5, 0, 0, 0, 8, 0},	       /* empty level */
		{5, 0, 4, 2, 5, 0, 3, 3, 4, 9, 0},     /* four ineq levels (Q3) */
		{3, 6, 0, 0, 64, 32, 0},	       /* n = 32 limits */
	};
	static const int ns[] = {8, 10, 6, 12, 14, 32};
	for (int c = 0; c < 6; ++c)
		for (int gate = 0; gate < 2; ++gate)
			for (int solver = 0; solver < 2; ++solver)
				for (int lowrank = 0; lowrank < 2; ++lowrank) {
					int K = shapes[c][0], m[5], p[5];
					for (int k = 0; k < K; ++k)
						m[k] = shapes[c][1 + 2 * k], p[k] = shapes[c][2 + 2 * k];
					/* (an infeasible reference-mode hierarchy may return rc != 0: only memory safety is checked there) */
					(void)run_tasks(ns[c], K, m, p, lowrank, gate, solver);
				}
	/* FullPivLU kernel: full rank, rank deficient, zero, square full rank (zero-column quirk) */
	{
		double A[6 * 9], Kout[9 * 10];
		for (int i = 0; i < 54; ++i)
			A[i] = urand();
		bad |= hoqp_oracle_fullpivlu_kernel(6, 9, A, Kout, 1) != 3;
		memcpy(A + 9, A, 9 * sizeof(double));
		bad |= hoqp_oracle_fullpivlu_kernel(6, 9, A, Kout, 1) != 4;
		memset(A, 0, sizeof(A));
		bad |= hoqp_oracle_fullpivlu_kernel(6, 9, A, Kout, 1) != 9;
		double B[9] = {2, 0, 0, 0, 3, 0, 0, 0, 4};
		bad |= hoqp_oracle_fullpivlu_kernel(3, 3, B, Kout, 1) != 1;
		bad |= hoqp_oracle_fullpivlu_kernel(3, 3, B, Kout, 0) != 0;
	}
	/* WBC: every contact pattern class, odd flags, both modes and solvers, batch over OpenMP threads */
	static const int masks[][4] = {{1, 1, 1, 1}, {1, 0, 0, 1}, {0, 1, 1, 1}, {1, 0, 0, 0}, {0, 0, 0, 0}, {1, 2, -1, 0}, {2, 2, 2, 2}};
	enum { NS = 7 * 3 };
	hoqp_oracle_wbc_state *st = calloc(NS, sizeof(*st));
	hoqp_oracle_wbc_result *res = calloc(NS, sizeof(*res));
	for (int i = 0; i < NS; ++i)
		make_state(&st[i], masks[i % 7]);
	for (int gate = 0; gate < 2; ++gate)
		for (int solver = 0; solver < 2; ++solver) {
			hoqp_oracle_cfg cfg;
			hoqp_oracle_wbc_params prm;
			hoqp_oracle_default_cfg(&cfg);
			hoqp_oracle_default_wbc_params(&prm);
			cfg.solver = solver;
			if (!gate) {
				cfg.gate_prior_ineq = cfg.slack_order_quirk = 0;
				cfg.eps = 0.0;
				prm.friction_col_quirk = prm.drop_base_pos_gain = 0;
				prm.mu = 0.2, prm.tau_max = 40.0, prm.tau_min = -40.0;
			}
			(void)hoqp_oracle_wbc_solve_batch(&cfg, &prm, NS, st, res, 3);
			for (int i = 0; i < NS; ++i)
				bad |= !finite_all(res[i].x, 30) || !finite_all(res[i].tau, 12);
			int m[5], p[5];
			double *A = calloc(5 * 32 * 32, sizeof(double)), *b = calloc(5 * 32, sizeof(double));
			double *D = calloc(64 * 32, sizeof(double)), *f = calloc(64, sizeof(double));
			for (int i = 0; i < NS; ++i)
				(void)hoqp_oracle_wbc_build(&prm, &st[i], m, p, A, b, D, f);
			free(A), free(b), free(D), free(f);
		}
	free(st), free(res);
	/* post-solve step, gait and stand-up samplers */
	{
		enum { N = 33 };
		double a[7][N * 12];
		for (int k = 0; k < 7; ++k)
			for (int i = 0; i < N * 12; ++i)
				a[k][i] = urand();
		hoqp_oracle_wbc_poststep(N, 2.5e-3, 50.0, 1.5, a[0], a[1], a[2], a[3], a[4], a[5], a[6]);
		bad |= !finite_all(a[6], N * 12);
		int sched[4 * 8];
		for (int i = 0; i < 32; ++i)
			sched[i] = (i * 7 + 3) % 5 != 0;
		double time[N], vel[N * 2], stance[N * 8], leg_cmd[N * 36], base_cmd[N * 9], pose[N * 3];
		int contact[N * 4];
		for (int i = 0; i < N; ++i) {
			time[i] = 20.0 * urand(); /* negative times included */
			vel[2 * i] = urand(), vel[2 * i + 1] = urand();
			for (int j = 0; j < 8; ++j)
				stance[8 * i + j] = urand();
			for (int j = 0; j < 3; ++j)
				pose[3 * i + j] = urand();
		}
		for (int fix = 0; fix < 2; ++fix)
			hoqp_oracle_gait_sample(N, 8, 8.0, sched, 0.15, fix, time, vel, stance, leg_cmd, contact);
		bad |= !finite_all(leg_cmd, N * 36);
		hoqp_oracle_standup_sample(N, 2.0, 0.5, time, pose, base_cmd, contact);
		bad |= !finite_all(base_cmd, N * 9);
	}
	printf(bad ? "asan_check: FAILED a sanity check\n" : "asan_check: ok\n");
	return bad ? 1 : 0;
}
