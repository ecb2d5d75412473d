/*
 * hoqp_oracle.c -- CPU restatement of the reference HO-QP whole-body-control solve.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (quadruped_locomotion_amd/,
 * include/, the HIP library) may include, link or call this file.  It is the checker
 * used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED: the reference (bernhardpg/quadruped_locomotion) ships no tests,
 * golden vectors or fixtures for this path, and cannot be built here (it needs Drake,
 * Eigen, ROS1, Gazebo, none of which are installed).  The arithmetic the reference
 * delegates to Drake's `Solve` (ho_qp_problem.cpp:339) and Eigen's
 * `FullPivLU::kernel()` (math.hpp:102) is restated from the published algorithms:
 * the QP is solved to its exact KKT point by a dense primal active-set method (Drake
 * would pick an unpinned third-party solver), the kernel by complete-pivoting LU.
 * This restatement is cross-checked against numpy/scipy in tests/ only.
 *
 * Two QP solvers (hoqp_oracle_cfg.solver): 1 (default) = the program in least-squares form with
 * every factorization a Householder QR in long double -- the exact-arithmetic minimiser of the
 * reference's program to ~1e-13, the value parity is asserted against; 0 = H = (AZ)'AZ + eps I
 * formed and factored in double, which is what a double-precision restatement of
 * ho_qp_problem.cpp:231-357 computes (kept as a second opinion and as the timed CPU baseline).
 *
 * Every function names the reference file:line it follows (paths relative to the
 * reference root).  All matrices are dense row-major doubles.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hoqp_oracle.h"

#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))

/* ------------------------------------------------------------------------- */
/* small dense helpers                                                        */
/* ------------------------------------------------------------------------- */

static double *dalloc(size_t n)
{
	double *p = (double *)calloc(n ? n : 1, sizeof(double));
	if (!p) {
		fprintf(stderr, "hoqp_oracle: out of memory\n");
		abort();
	}
	return p;
}

/* C (r x c) = A (r x k) * B (k x c) */
static void matmul(int r, int k, int c, const double *A, const double *B, double *C)
{
	for (int i = 0; i < r; ++i)
		for (int j = 0; j < c; ++j) {
			double s = 0.0;
			for (int t = 0; t < k; ++t)
				s += A[i * k + t] * B[t * c + j];
			C[i * c + j] = s;
		}
}

/* y (r) = A (r x c) x (c) */
static void matvec(int r, int c, const double *A, const double *x, double *y)
{
	for (int i = 0; i < r; ++i) {
		double s = 0.0;
		for (int j = 0; j < c; ++j)
			s += A[i * c + j] * x[j];
		y[i] = s;
	}
}

static double norm_inf(int n, const double *x)
{
	double m = 0.0;
	for (int i = 0; i < n; ++i)
		m = MAX(m, fabs(x[i]));
	return m;
}

/*
 * Householder QR with column pivoting of T (r x c):  T P = Q R.
 * On exit T holds R (columns physically permuted), Q (r x r) is explicit,
 * perm[j] = original column now at position j.  Returns the numerical rank
 * (pivot column norms > rtol * first pivot norm).
 */
static int qr_colpiv(int r, int c, double *T, double *Q, int *perm, double rtol)
{
	for (int i = 0; i < r; ++i)
		for (int j = 0; j < r; ++j)
			Q[i * r + j] = (i == j) ? 1.0 : 0.0;
	for (int j = 0; j < c; ++j)
		perm[j] = j;
	int kmax = MIN(r, c), rank = 0;
	double first = 0.0;
	double *v = dalloc(r);
	for (int k = 0; k < kmax; ++k) {
		int best = -1;
		double bn = -1.0;
		for (int j = k; j < c; ++j) {
			double s = 0.0;
			for (int i = k; i < r; ++i)
				s += T[i * c + j] * T[i * c + j];
			if (s > bn) {
				bn = s;
				best = j;
			}
		}
		double nrm = sqrt(bn);
		if (k == 0)
			first = nrm;
		if (!(nrm > rtol * first) || nrm == 0.0)
			break;
		if (best != k) {
			for (int i = 0; i < r; ++i) {
				double t = T[i * c + k];
				T[i * c + k] = T[i * c + best];
				T[i * c + best] = t;
			}
			int t = perm[k];
			perm[k] = perm[best];
			perm[best] = t;
		}
		double alpha = T[k * c + k];
		double beta = (alpha >= 0.0) ? -nrm : nrm;
		double vn2 = 0.0;
		for (int i = k; i < r; ++i) {
			v[i] = T[i * c + k];
			if (i == k)
				v[i] = alpha - beta;
			vn2 += v[i] * v[i];
		}
		if (vn2 > 0.0) {
			for (int j = k; j < c; ++j) {
				double s = 0.0;
				for (int i = k; i < r; ++i)
					s += v[i] * T[i * c + j];
				s *= 2.0 / vn2;
				for (int i = k; i < r; ++i)
					T[i * c + j] -= s * v[i];
			}
			for (int i = 0; i < r; ++i) {
				double s = 0.0;
				for (int t = k; t < r; ++t)
					s += Q[i * r + t] * v[t];
				s *= 2.0 / vn2;
				for (int t = k; t < r; ++t)
					Q[i * r + t] -= s * v[t];
			}
		}
		T[k * c + k] = beta;
		for (int i = k + 1; i < r; ++i)
			T[i * c + k] = 0.0;
		++rank;
	}
	free(v);
	return rank;
}

/*
 * Solve the symmetric positive SEMI-definite system A x = rhs (A n x n, destroyed)
 * by diagonally pivoted Cholesky; pivots <= rtol * largest initial diagonal are
 * treated as zero and the MINIMUM-NORM solution of the consistent singular system is returned.
 */
static int psd_solve(int n, double *A, const double *rhs, double *x, double rtol, double floor_abs)
{
	int *perm = (int *)malloc(sizeof(int) * (n ? n : 1));
	double *y = dalloc(n);
	for (int i = 0; i < n; ++i)
		perm[i] = i;
	double maxd0 = 0.0;
	for (int i = 0; i < n; ++i)
		maxd0 = MAX(maxd0, A[i * n + i]);
	int rank = 0;
	for (int k = 0; k < n; ++k) {
		int best = k;
		for (int j = k + 1; j < n; ++j)
			if (A[j * n + j] > A[best * n + best])
				best = j;
		if (!(A[best * n + best] > rtol * maxd0) || A[best * n + best] <= floor_abs)
			break;
		if (best != k) {
			for (int j = 0; j < n; ++j) {
				double t = A[k * n + j];
				A[k * n + j] = A[best * n + j];
				A[best * n + j] = t;
			}
			for (int i = 0; i < n; ++i) {
				double t = A[i * n + k];
				A[i * n + k] = A[i * n + best];
				A[i * n + best] = t;
			}
			int t = perm[k];
			perm[k] = perm[best];
			perm[best] = t;
		}
		double l = sqrt(A[k * n + k]);
		A[k * n + k] = l;
		for (int i = k + 1; i < n; ++i)
			A[i * n + k] /= l;
		for (int i = k + 1; i < n; ++i)
			for (int j = k + 1; j <= i; ++j) {
				A[i * n + j] -= A[i * n + k] * A[j * n + k];
				A[j * n + i] = A[i * n + j];
			}
		++rank;
	}
	if (rank == n) {
		for (int i = 0; i < rank; ++i) {
			double s = rhs[perm[i]];
			for (int j = 0; j < i; ++j)
				s -= A[i * n + j] * y[j];
			y[i] = s / A[i * n + i];
		}
		for (int i = rank - 1; i >= 0; --i) {
			double s = y[i];
			for (int j = i + 1; j < rank; ++j)
				s -= A[j * n + i] * y[j];
			y[i] = s / A[i * n + i];
		}
		for (int i = 0; i < n; ++i)
			x[perm[i]] = y[i];
	} else {
		/* P'AP = L L' with L (n x rank) = lower trapezoid of A.  Minimum-norm solution
		 * x = L (L'L)^-1 (L'L)^-1 L' rhs: this is the eps -> 0 limit of the reference's
		 * (H + eps I) solve (ho_qp_problem.cpp:244-246). */
		double *G = dalloc((size_t)rank * rank), *t = dalloc(rank), *a = dalloc(rank);
		for (int i = 0; i < rank; ++i)
			for (int j = 0; j <= i; ++j) {
				double s = 0.0;
				for (int k = i; k < n; ++k) /* L[k][i] = 0 for k < i */
					s += A[k * n + i] * A[k * n + j];
				G[i * rank + j] = G[j * rank + i] = s;
			}
		for (int j = 0; j < rank; ++j) {
			double s = 0.0;
			for (int k = j; k < n; ++k)
				s += A[k * n + j] * rhs[perm[k]];
			t[j] = s;
		}
		/* Cholesky of G (SPD, rank x rank) */
		for (int k = 0; k < rank; ++k) {
			double l = sqrt(G[k * rank + k]);
			G[k * rank + k] = l;
			for (int i = k + 1; i < rank; ++i)
				G[i * rank + k] /= l;
			for (int i = k + 1; i < rank; ++i)
				for (int j = k + 1; j <= i; ++j)
					G[i * rank + j] -= G[i * rank + k] * G[j * rank + k];
		}
		for (int pass = 0; pass < 2; ++pass) {
			for (int i = 0; i < rank; ++i) {
				double s = t[i];
				for (int j = 0; j < i; ++j)
					s -= G[i * rank + j] * a[j];
				a[i] = s / G[i * rank + i];
			}
			for (int i = rank - 1; i >= 0; --i) {
				double s = a[i];
				for (int j = i + 1; j < rank; ++j)
					s -= G[j * rank + i] * t[j];
				t[i] = s / G[i * rank + i];
			}
		}
		for (int k = 0; k < n; ++k) {
			double s = 0.0;
			for (int j = 0; j < rank && j <= k; ++j)
				s += A[k * n + j] * t[j];
			x[perm[k]] = s;
		}
		free(G);
		free(t);
		free(a);
	}
	free(perm);
	free(y);
	return rank;
}

/* ------------------------------------------------------------------------- */
/* Convex QP,  min 1/2 xi'H xi + c'xi  s.t.  C xi <= d,  H PSD                 */
/* stands in for drake::solvers::Solve(prog_), ho_qp_problem.cpp:337-357       */
/* ------------------------------------------------------------------------- */

#define QP_OK 0
#define QP_MAXIT 1
#define QP_UNBOUNDED 2
#define QP_INFEASIBLE 4

/*
 * Primal active-set method (null-space form) from a feasible start `xi`.
 * Reduced Hessians may be singular: the EQP step is then a basic solution of the
 * consistent singular system; an inconsistent system yields a zero-curvature
 * descent ray that is followed to the first blocking constraint.
 */
static int qp_primal_active_set(int N, const double *H, const double *c, int Mc, const double *C,
				const double *d, double *xi, double *lam, int maxit, int *iters_out, double piv_floor)
{
	int status = QP_MAXIT;
	int *W = (int *)malloc(sizeof(int) * (N + 1));
	char *inW = (char *)calloc(Mc ? Mc : 1, 1);
	int w = 0;
	double *g = dalloc(N), *p = dalloc(N), *T = dalloc((size_t)N * (N + 1));
	double *Q = dalloc((size_t)N * N), *HZ = dalloc((size_t)N * N), *Hr = dalloc((size_t)N * N);
	double *gr = dalloc(N), *q = dalloc(N), *Hr0 = dalloc((size_t)N * N), *tmp = dalloc(N);
	int *perm = (int *)malloc(sizeof(int) * (N + 1));
	int it = 0, degenerate_run = 0;
	for (int i = 0; i < Mc; ++i)
		lam[i] = 0.0;

	for (it = 0; it < maxit; ++it) {
		/* gradient */
		matvec(N, N, H, xi, g);
		for (int i = 0; i < N; ++i)
			g[i] += c[i];
		/* null space of the working set: QR of C_W' (N x w) */
		int rw = 0;
		if (w > 0) {
			for (int i = 0; i < N; ++i)
				for (int j = 0; j < w; ++j)
					T[i * w + j] = C[W[j] * N + i];
			rw = qr_colpiv(N, w, T, Q, perm, 1e-12);
		} else {
			for (int i = 0; i < N; ++i)
				for (int j = 0; j < N; ++j)
					Q[i * N + j] = (i == j) ? 1.0 : 0.0;
		}
		int nf = N - rw;
		int ray = 0;
		for (int i = 0; i < N; ++i)
			p[i] = 0.0;
		if (nf > 0) {
			/* Zw = Q[:, rw:];  Hr = Zw' H Zw;  gr = Zw' g */
			for (int i = 0; i < N; ++i)
				for (int j = 0; j < nf; ++j) {
					double s = 0.0;
					for (int t = 0; t < N; ++t)
						s += H[i * N + t] * Q[t * N + rw + j];
					HZ[i * nf + j] = s;
				}
			for (int i = 0; i < nf; ++i)
				for (int j = 0; j < nf; ++j) {
					double s = 0.0;
					for (int t = 0; t < N; ++t)
						s += Q[t * N + rw + i] * HZ[t * nf + j];
					Hr[i * nf + j] = s;
				}
			for (int i = 0; i < nf; ++i)
				for (int j = i + 1; j < nf; ++j) {
					double a = 0.5 * (Hr[i * nf + j] + Hr[j * nf + i]);
					Hr[i * nf + j] = Hr[j * nf + i] = a;
				}
			memcpy(Hr0, Hr, sizeof(double) * nf * nf);
			for (int i = 0; i < nf; ++i) {
				double s = 0.0;
				for (int t = 0; t < N; ++t)
					s += Q[t * N + rw + i] * g[t];
				gr[i] = -s;
			}
			psd_solve(nf, Hr, gr, q, 1e-10, piv_floor);
			/* consistency: r = Hr q - (-gr') must vanish for a bounded EQP */
			double rmax = 0.0, gmax = norm_inf(nf, gr);
			for (int i = 0; i < nf; ++i) {
				double s = -gr[i];
				for (int j = 0; j < nf; ++j)
					s += Hr0[i * nf + j] * q[j];
				tmp[i] = s;
				rmax = MAX(rmax, fabs(s));
			}
			if (rmax > 1e-6 * (gmax + 1e-300) && rmax > 1e-9) {
				/* zero-curvature descent ray: follow -r */
				ray = 1;
				for (int i = 0; i < nf; ++i)
					q[i] = -tmp[i];
			}
			for (int i = 0; i < N; ++i) {
				double s = 0.0;
				for (int j = 0; j < nf; ++j)
					s += Q[i * N + rw + j] * q[j];
				p[i] = s;
			}
		}
		double pn = norm_inf(N, p), xn = norm_inf(N, xi);
		if (pn <= 1e-11 * (1.0 + xn)) {
			/* stationary on the working set: multipliers from C_W' lam = -g */
			if (w == 0) {
				status = QP_OK;
				break;
			}
			/* T = R (N x w, permuted), Q explicit:  lam_perm = -R1^{-1} Q1' g */
			double *rhs = tmp;
			for (int j = 0; j < rw; ++j) {
				double s = 0.0;
				for (int t = 0; t < N; ++t)
					s += Q[t * N + j] * g[t];
				rhs[j] = -s;
			}
			for (int j = rw - 1; j >= 0; --j) {
				double s = rhs[j];
				for (int t = j + 1; t < rw; ++t)
					s -= T[j * w + t] * q[t];
				q[j] = s / T[j * w + j];
			}
			double gn = norm_inf(N, g);
			int drop = -1;
			double worst = -1e-9 * (1.0 + gn);
			for (int j = 0; j < rw; ++j) {
				int wi = perm[j];
				if (degenerate_run > 2 * N) { /* Bland: lowest index */
					if (q[j] < -1e-9 * (1.0 + gn) && (drop < 0 || W[wi] < W[drop]))
						drop = wi;
				} else if (q[j] < worst) {
					worst = q[j];
					drop = wi;
				}
			}
			if (drop < 0) {
				for (int j = 0; j < rw; ++j)
					lam[W[perm[j]]] = q[j];
				status = QP_OK;
				break;
			}
				inW[W[drop]] = 0;
			for (int j = drop; j < w - 1; ++j)
				W[j] = W[j + 1];
			--w;
			continue;
		}
		/* ratio test */
		double alpha = ray ? INFINITY : 1.0;
		int block = -1;
		for (int i = 0; i < Mc; ++i) {
			if (inW[i])
				continue;
			double cp = 0.0, cx = 0.0, cn = 0.0;
			for (int j = 0; j < N; ++j) {
				cp += C[i * N + j] * p[j];
				cx += C[i * N + j] * xi[j];
				cn = MAX(cn, fabs(C[i * N + j]));
			}
			if (cp > 1e-12 * cn * pn) {
				double a = (d[i] - cx) / cp;
				if (a < 0.0)
					a = 0.0;
				if (a < alpha || (a == alpha && block >= 0 && degenerate_run > 2 * N && i < block)) {
					alpha = a;
					block = i;
				}
			}
		}
		if (isinf(alpha)) {
			status = QP_UNBOUNDED;
			break;
		}
		for (int i = 0; i < N; ++i)
			xi[i] += alpha * p[i];
		degenerate_run = (alpha <= 1e-14) ? degenerate_run + 1 : 0;
		if (block >= 0) {
			W[w++] = block;
			inW[block] = 1;
		}
	}
	if (iters_out)
		*iters_out = it;
	free(W);
	free(inW);
	free(g);
	free(p);
	free(T);
	free(Q);
	free(HZ);
	free(Hr);
	free(gr);
	free(q);
	free(Hr0);
	free(tmp);
	free(perm);
	return status;
}

/*
 * Full solve: phase 1 (only if the supplied start violates C xi <= d) then phase 2.
 * Phase 1 minimises 1/2 |s|^2 s.t. C xi - s <= d, which is itself a PSD QP with the
 * trivially feasible start s = max(0, C xi - d).
 */
static int qp_solve(int N, const double *H, const double *c, int Mc, const double *C, const double *d,
		    double *xi, double *lam, int maxit, int *iters_out, double piv_floor)
{
	double viol = 0.0, scale = 1.0;
	for (int i = 0; i < Mc; ++i) {
		double s = -d[i];
		for (int j = 0; j < N; ++j)
			s += C[i * N + j] * xi[j];
		viol = MAX(viol, s);
		scale = MAX(scale, fabs(d[i]));
	}
	int it1 = 0, st = QP_OK;
	if (viol > 1e-9 * scale) {
		int N1 = N + Mc;
		double *H1 = dalloc((size_t)N1 * N1), *c1 = dalloc(N1), *C1 = dalloc((size_t)Mc * N1);
		double *x1 = dalloc(N1), *l1 = dalloc(Mc);
		for (int i = 0; i < Mc; ++i) {
			H1[(N + i) * N1 + N + i] = 1.0;
			for (int j = 0; j < N; ++j)
				C1[i * N1 + j] = C[i * N + j];
			C1[i * N1 + N + i] = -1.0;
		}
		memcpy(x1, xi, sizeof(double) * N);
		for (int i = 0; i < Mc; ++i) {
			double s = -d[i];
			for (int j = 0; j < N; ++j)
				s += C[i * N + j] * xi[j];
			x1[N + i] = MAX(0.0, s);
		}
		st = qp_primal_active_set(N1, H1, c1, Mc, C1, d, x1, l1, maxit, &it1, piv_floor);
		memcpy(xi, x1, sizeof(double) * N);
		double smax = norm_inf(Mc, x1 + N);
		free(H1);
		free(c1);
		free(C1);
		free(x1);
		free(l1);
		if (smax > 1e-9 * scale) {
			if (iters_out)
				*iters_out = it1;
			return st | QP_INFEASIBLE;
		}
	}
	int it2 = 0;
	st |= qp_primal_active_set(N, H, c, Mc, C, d, xi, lam, maxit, &it2, piv_floor);
	if (iters_out)
		*iters_out = it1 + it2;
	return st;
}

/* ------------------------------------------------------------------------- */
/* The same QP in LEAST-SQUARES form, extended precision (solver = 1)          */
/* ------------------------------------------------------------------------- */
/*
 * The reference's H and c (ConstructHMatrix / ConstructCVector, ho_qp_problem.cpp:231-285) are
 * H = B'B, c = -B'beta with  B = [A Z, 0 ; sqrt(eps) I, 0 ; 0, I],  beta = [b - A x ; 0 ; 0],
 * so the QP of :291-300 is   min 1/2 |B xi - beta|^2   s.t.  C xi <= d.
 * Forming H squares the condition number of A Z: with sigma_min(A Z) ~ 1e-3 (a leg near a singular
 * posture) a double-precision Cholesky of H is good to ~1e-6 only, which is the tolerance of the
 * whole parity claim.  This solver never forms H: every equality-constrained step is a
 * minimum-norm least-squares solve by Householder QR with column pivoting in long double
 * (64-bit mantissa), so its answer is the exact-arithmetic minimiser of the reference's program to
 * ~1e-13 -- the authoritative value the HIP path is compared with.  The literal H form above stays
 * as solver = 0 (what a double-precision restatement of the reference computes).
 */
typedef long double ldbl;

static ldbl *lalloc(size_t n)
{
	ldbl *p = (ldbl *)calloc(n ? n : 1, sizeof(ldbl));
	if (!p) {
		fprintf(stderr, "hoqp_oracle: out of memory\n");
		abort();
	}
	return p;
}

/* T (r x c) P = Q R, long double; as qr_colpiv */
static int qr_colpiv_ld(int r, int c, ldbl *T, ldbl *Q, int *perm, ldbl rtol)
{
	for (int i = 0; i < r; ++i)
		for (int j = 0; j < r; ++j)
			Q[i * r + j] = (i == j) ? 1.0L : 0.0L;
	for (int j = 0; j < c; ++j)
		perm[j] = j;
	int kmax = MIN(r, c), rank = 0;
	ldbl first = 0.0L;
	ldbl *v = lalloc(r);
	for (int k = 0; k < kmax; ++k) {
		int best = -1;
		ldbl bn = -1.0L;
		for (int j = k; j < c; ++j) {
			ldbl s = 0.0L;
			for (int i = k; i < r; ++i)
				s += T[i * c + j] * T[i * c + j];
			if (s > bn) {
				bn = s;
				best = j;
			}
		}
		ldbl nrm = sqrtl(bn);
		if (k == 0)
			first = nrm;
		if (!(nrm > rtol * first) || nrm == 0.0L)
			break;
		if (best != k) {
			for (int i = 0; i < r; ++i) {
				ldbl t = T[i * c + k];
				T[i * c + k] = T[i * c + best];
				T[i * c + best] = t;
			}
			int t = perm[k];
			perm[k] = perm[best];
			perm[best] = t;
		}
		ldbl alpha = T[k * c + k];
		ldbl beta = (alpha >= 0.0L) ? -nrm : nrm;
		ldbl vn2 = 0.0L;
		for (int i = k; i < r; ++i) {
			v[i] = T[i * c + k];
			if (i == k)
				v[i] = alpha - beta;
			vn2 += v[i] * v[i];
		}
		if (vn2 > 0.0L) {
			for (int j = k; j < c; ++j) {
				ldbl s = 0.0L;
				for (int i = k; i < r; ++i)
					s += v[i] * T[i * c + j];
				s *= 2.0L / vn2;
				for (int i = k; i < r; ++i)
					T[i * c + j] -= s * v[i];
			}
			for (int i = 0; i < r; ++i) {
				ldbl s = 0.0L;
				for (int t = k; t < r; ++t)
					s += Q[i * r + t] * v[t];
				s *= 2.0L / vn2;
				for (int t = k; t < r; ++t)
					Q[i * r + t] -= s * v[t];
			}
		}
		T[k * c + k] = beta;
		for (int i = k + 1; i < r; ++i)
			T[i * c + k] = 0.0L;
		++rank;
	}
	free(v);
	return rank;
}

/* q = minimum-norm minimiser of |B q - beta| (B is r x c, any rank); complete orthogonal decomposition */
static void lsq_minnorm_ld(int r, int c, const ldbl *B, const ldbl *beta, ldbl *q, ldbl rtol)
{
	for (int j = 0; j < c; ++j)
		q[j] = 0.0L;
	if (r == 0 || c == 0)
		return;
	ldbl *T = lalloc((size_t)r * c), *Q = lalloc((size_t)r * r), *c1 = lalloc(r), *y = lalloc(c);
	int *perm = (int *)malloc(sizeof(int) * c);
	memcpy(T, B, sizeof(ldbl) * r * c);
	int k = qr_colpiv_ld(r, c, T, Q, perm, rtol);
	for (int j = 0; j < k; ++j) {
		ldbl s = 0.0L;
		for (int i = 0; i < r; ++i)
			s += Q[i * r + j] * beta[i];
		c1[j] = s;
	}
	if (k == c) {
		for (int i = k - 1; i >= 0; --i) {
			ldbl s = c1[i];
			for (int j = i + 1; j < k; ++j)
				s -= T[i * c + j] * y[j];
			y[i] = s / T[i * c + i];
		}
	} else if (k > 0) {
		/* [R11 R12] y = c1, minimum norm:  (R')(c x k) P2 = Q2 R2  =>  y = Q2[:, :k] R2^-T P2' c1 */
		ldbl *Tt = lalloc((size_t)c * k), *Q2 = lalloc((size_t)c * c), *w = lalloc(k);
		int *p2 = (int *)malloc(sizeof(int) * k);
		for (int i = 0; i < k; ++i)
			for (int j = 0; j < c; ++j)
				Tt[j * k + i] = T[i * c + j];
		int k2 = qr_colpiv_ld(c, k, Tt, Q2, p2, 0.0L);
		for (int i = 0; i < k2; ++i) { /* R2' w = P2' c1 (forward substitution) */
			ldbl s = c1[p2[i]];
			for (int j = 0; j < i; ++j)
				s -= Tt[j * k + i] * w[j];
			w[i] = s / Tt[i * k + i];
		}
		for (int j = 0; j < c; ++j) {
			ldbl s = 0.0L;
			for (int i = 0; i < k2; ++i)
				s += Q2[j * c + i] * w[i];
			y[j] = s;
		}
		free(Tt);
		free(Q2);
		free(w);
		free(p2);
	}
	for (int j = 0; j < c; ++j)
		q[perm[j]] = y[j];
	free(T);
	free(Q);
	free(c1);
	free(y);
	free(perm);
}

/* primal active set on  min 1/2 |B xi - beta|^2  s.t.  C xi <= d  from a feasible xi (B: R x N) */
static int lsq_primal_active_set(int N, int R, const ldbl *B, const ldbl *beta, int Mc, const ldbl *C, const ldbl *d,
				 ldbl *xi, ldbl *lam, int maxit, int *iters_out)
{
	int status = QP_MAXIT;
	int *W = (int *)malloc(sizeof(int) * (N + 1));
	char *inW = (char *)calloc(Mc ? Mc : 1, 1);
	int w = 0;
	ldbl *g = lalloc(N), *p = lalloc(N), *T = lalloc((size_t)N * (N + 1)), *Q = lalloc((size_t)N * N);
	ldbl *BZ = lalloc((size_t)R * N), *res = lalloc(R), *q = lalloc(N), *tmp = lalloc(N);
	int *perm = (int *)malloc(sizeof(int) * (N + 1));
	int it = 0, degenerate_run = 0;
	for (int i = 0; i < Mc; ++i)
		lam[i] = 0.0L;
	for (it = 0; it < maxit; ++it) {
		/* residual beta - B xi and gradient g = -B'(beta - B xi) */
		for (int i = 0; i < R; ++i) {
			ldbl s = beta[i];
			for (int j = 0; j < N; ++j)
				s -= B[i * N + j] * xi[j];
			res[i] = s;
		}
		for (int j = 0; j < N; ++j) {
			ldbl s = 0.0L;
			for (int i = 0; i < R; ++i)
				s -= B[i * N + j] * res[i];
			g[j] = s;
		}
		int rw = 0;
		if (w > 0) {
			for (int i = 0; i < N; ++i)
				for (int j = 0; j < w; ++j)
					T[i * w + j] = C[W[j] * N + i];
			/* rank tolerances follow the precision of the DATA (C and B hold products rounded to
			 * double: dependent rows are dependent to ~1e-16 only), not of the arithmetic */
			rw = qr_colpiv_ld(N, w, T, Q, perm, 1e-12L);
		} else {
			for (int i = 0; i < N; ++i)
				for (int j = 0; j < N; ++j)
					Q[i * N + j] = (i == j) ? 1.0L : 0.0L;
		}
		int nf = N - rw;
		for (int i = 0; i < N; ++i)
			p[i] = 0.0L;
		if (nf > 0) {
			for (int i = 0; i < R; ++i)
				for (int j = 0; j < nf; ++j) {
					ldbl s = 0.0L;
					for (int t = 0; t < N; ++t)
						s += B[i * N + t] * Q[t * N + rw + j];
					BZ[i * nf + j] = s;
				}
			lsq_minnorm_ld(R, nf, BZ, res, q, 1e-12L);
			for (int i = 0; i < N; ++i) {
				ldbl s = 0.0L;
				for (int j = 0; j < nf; ++j)
					s += Q[i * N + rw + j] * q[j];
				p[i] = s;
			}
		}
		ldbl pn = 0.0L, xn = 0.0L, gn = 0.0L;
		for (int i = 0; i < N; ++i) {
			pn = MAX(pn, fabsl(p[i]));
			xn = MAX(xn, fabsl(xi[i]));
			gn = MAX(gn, fabsl(g[i]));
		}
		if (pn <= 1e-13L * (1.0L + xn)) {
			if (w == 0) {
				status = QP_OK;
				break;
			}
			for (int j = 0; j < rw; ++j) {
				ldbl s = 0.0L;
				for (int t = 0; t < N; ++t)
					s += Q[t * N + j] * g[t];
				tmp[j] = -s;
			}
			for (int j = rw - 1; j >= 0; --j) {
				ldbl s = tmp[j];
				for (int t = j + 1; t < rw; ++t)
					s -= T[j * w + t] * q[t];
				q[j] = s / T[j * w + j];
			}
			int drop = -1;
			ldbl worst = -1e-9L * (1.0L + gn);
			for (int j = 0; j < rw; ++j) {
				int wi = perm[j];
				if (degenerate_run > 2 * N) { /* Bland: lowest index */
					if (q[j] < -1e-9L * (1.0L + gn) && (drop < 0 || W[wi] < W[drop]))
						drop = wi;
				} else if (q[j] < worst) {
					worst = q[j];
					drop = wi;
				}
			}
			if (drop < 0) {
				for (int j = 0; j < rw; ++j)
					lam[W[perm[j]]] = q[j];
				status = QP_OK;
				break;
			}
			inW[W[drop]] = 0;
			for (int j = drop; j < w - 1; ++j)
				W[j] = W[j + 1];
			--w;
			continue;
		}
		ldbl alpha = 1.0L;
		int block = -1;
		for (int i = 0; i < Mc; ++i) {
			if (inW[i])
				continue;
			ldbl cp = 0.0L, cx = 0.0L, cn = 0.0L;
			for (int j = 0; j < N; ++j) {
				cp += C[i * N + j] * p[j];
				cx += C[i * N + j] * xi[j];
				cn = MAX(cn, fabsl(C[i * N + j]));
			}
			if (cp > 1e-14L * cn * pn) {
				ldbl a = (d[i] - cx) / cp;
				if (a < 0.0L)
					a = 0.0L;
				if (a < alpha || (a == alpha && block >= 0 && degenerate_run > 2 * N && i < block)) {
					alpha = a;
					block = i;
				}
			}
		}
		for (int i = 0; i < N; ++i)
			xi[i] += alpha * p[i];
		degenerate_run = (alpha <= 1e-16L) ? degenerate_run + 1 : 0;
		if (block >= 0) {
			W[w++] = block;
			inW[block] = 1;
		}
	}
	if (iters_out)
		*iters_out = it;
	free(W);
	free(inW);
	free(g);
	free(p);
	free(T);
	free(Q);
	free(BZ);
	free(res);
	free(q);
	free(tmp);
	free(perm);
	return status;
}

/* phase 1 (only if xi violates C xi <= d): min 1/2 |s|^2 s.t. C xi - s <= d, then phase 2 */
static int lsq_qp_solve(int N, int R, const ldbl *B, const ldbl *beta, int Mc, const ldbl *C, const ldbl *d, ldbl *xi,
			ldbl *lam, int maxit, int *iters_out)
{
	ldbl viol = 0.0L, scale = 1.0L;
	for (int i = 0; i < Mc; ++i) {
		ldbl s = -d[i];
		for (int j = 0; j < N; ++j)
			s += C[i * N + j] * xi[j];
		viol = MAX(viol, s);
		scale = MAX(scale, fabsl(d[i]));
	}
	int it1 = 0, st = QP_OK;
	if (viol > 1e-9L * scale) {
		int N1 = N + Mc;
		ldbl *B1 = lalloc((size_t)Mc * N1), *b1 = lalloc(Mc), *C1 = lalloc((size_t)Mc * N1);
		ldbl *x1 = lalloc(N1), *l1 = lalloc(Mc);
		for (int i = 0; i < Mc; ++i) {
			B1[i * N1 + N + i] = 1.0L;
			for (int j = 0; j < N; ++j)
				C1[i * N1 + j] = C[i * N + j];
			C1[i * N1 + N + i] = -1.0L;
		}
		memcpy(x1, xi, sizeof(ldbl) * N);
		for (int i = 0; i < Mc; ++i) {
			ldbl s = -d[i];
			for (int j = 0; j < N; ++j)
				s += C[i * N + j] * xi[j];
			x1[N + i] = MAX(0.0L, s);
		}
		st = lsq_primal_active_set(N1, Mc, B1, b1, Mc, C1, d, x1, l1, maxit, &it1);
		memcpy(xi, x1, sizeof(ldbl) * N);
		ldbl smax = 0.0L;
		for (int i = 0; i < Mc; ++i)
			smax = MAX(smax, fabsl(x1[N + i]));
		free(B1);
		free(b1);
		free(C1);
		free(x1);
		free(l1);
		if (smax > 1e-9L * scale) {
			if (iters_out)
				*iters_out = it1;
			return st | QP_INFEASIBLE;
		}
	}
	int it2 = 0;
	st |= lsq_primal_active_set(N, R, B, beta, Mc, C, d, xi, lam, maxit, &it2);
	if (iters_out)
		*iters_out = it1 + it2;
	return st;
}

/* exported for unit tests of the QP stand-in */
int hoqp_oracle_qp(int N, const double *H, const double *c, int Mc, const double *C, const double *d,
		   double *xi, double *lam, int maxit, int *iters)
{
	return qp_solve(N, H, c, Mc, C, d, xi, lam, maxit, iters, 0.0);
}

/* ------------------------------------------------------------------------- */
/* null-space bases                                                           */
/* ------------------------------------------------------------------------- */

/*
 * Restatement of Eigen's FullPivLU::kernel() as used by CalcNullSpaceProjMatrix
 * (math.hpp:99-104): LU with complete pivoting P A Q = L U, rank = #pivots with
 * |U_ii| > eps * min(r,c) * |max pivot|, kernel = Q [ -U11^{-1} U12 ; I ].
 * A kernel of dimension 0 is returned as ONE all-zero column (quirk Q6) when
 * zero_col is set, else as 0 columns.  A is r x c; Kout is c x (return value).
 */
int hoqp_oracle_fullpivlu_kernel(int r, int c, const double *A, double *Kout, int zero_col)
{
	double *lu = dalloc((size_t)r * c);
	memcpy(lu, A, sizeof(double) * r * c);
	int *colperm = (int *)malloc(sizeof(int) * (c ? c : 1));
	for (int j = 0; j < c; ++j)
		colperm[j] = j;
	int size = MIN(r, c), nonzero = size;
	double maxpivot = 0.0;
	for (int k = 0; k < size; ++k) {
		/* biggest |coefficient| of the bottom-right corner; among equal ones the first in column-major
		 * order, as Eigen's maxCoeff visitor walks a column-major matrix (strict '>' keeps the first) */
		int bi = k, bj = k;
		double best = 0.0;
		for (int j = k; j < c; ++j)
			for (int i = k; i < r; ++i)
				if (fabs(lu[i * c + j]) > best) {
					best = fabs(lu[i * c + j]);
					bi = i;
					bj = j;
				}
		if (best == 0.0) {
			nonzero = k;
			break;
		}
		maxpivot = MAX(maxpivot, best);
		if (bi != k)
			for (int j = 0; j < c; ++j) {
				double t = lu[k * c + j];
				lu[k * c + j] = lu[bi * c + j];
				lu[bi * c + j] = t;
			}
		if (bj != k) {
			for (int i = 0; i < r; ++i) {
				double t = lu[i * c + k];
				lu[i * c + k] = lu[i * c + bj];
				lu[i * c + bj] = t;
			}
			int t = colperm[k];
			colperm[k] = colperm[bj];
			colperm[bj] = t;
		}
		for (int i = k + 1; i < r; ++i)
			lu[i * c + k] /= lu[k * c + k];
		for (int i = k + 1; i < r; ++i)
			for (int j = k + 1; j < c; ++j)
				lu[i * c + j] -= lu[i * c + k] * lu[k * c + j];
	}
	double thr = maxpivot * (2.220446049250313e-16 * (double)size);
	int rank = 0;
	for (int k = 0; k < nonzero; ++k)
		if (fabs(lu[k * c + k]) > thr)
			++rank;
	int dimker = c - rank;
	if (dimker == 0) {
		free(lu);
		free(colperm);
		if (zero_col) {
			for (int i = 0; i < c; ++i)
				Kout[i] = 0.0;
			return 1;
		}
		return 0;
	}
	/* With complete pivoting the significant pivots lead: U11 = lu[0:rank,0:rank]. */
	double *X = dalloc((size_t)rank * dimker); /* X = U11^{-1} U12 */
	for (int k = 0; k < dimker; ++k)
		for (int i = rank - 1; i >= 0; --i) {
			double s = lu[i * c + rank + k];
			for (int j = i + 1; j < rank; ++j)
				s -= lu[i * c + j] * X[j * dimker + k];
			X[i * dimker + k] = s / lu[i * c + i];
		}
	for (int i = 0; i < c * dimker; ++i)
		Kout[i] = 0.0;
	for (int i = 0; i < rank; ++i)
		for (int k = 0; k < dimker; ++k)
			Kout[colperm[i] * dimker + k] = -X[i * dimker + k];
	for (int k = 0; k < dimker; ++k)
		Kout[colperm[rank + k] * dimker + k] = 1.0;
	free(X);
	free(lu);
	free(colperm);
	return dimker;
}

/* Orthonormal kernel basis from a column-pivoted Householder QR of A' ("paper"/build basis). */
static int householder_kernel(int r, int c, const double *A, double *Kout)
{
	double *T = dalloc((size_t)c * r), *Q = dalloc((size_t)c * c);
	int *perm = (int *)malloc(sizeof(int) * (r ? r : 1));
	for (int i = 0; i < c; ++i)
		for (int j = 0; j < r; ++j)
			T[i * r + j] = A[j * c + i];
	int rank = qr_colpiv(c, r, T, Q, perm, 2.220446049250313e-16 * (double)MAX(r, c));
	int dimker = c - rank;
	for (int i = 0; i < c; ++i)
		for (int k = 0; k < dimker; ++k)
			Kout[i * dimker + k] = Q[i * c + rank + k];
	free(T);
	free(Q);
	free(perm);
	return dimker;
}

/* ------------------------------------------------------------------------- */
/* HoQpProblem chain                                                          */
/* ------------------------------------------------------------------------- */

void hoqp_oracle_default_cfg(hoqp_oracle_cfg *cfg)
{
	cfg->eps = 1e-12;	  /* ho_qp_problem.hpp:56 */
	cfg->gate_prior_ineq = 1; /* Q2, ho_qp_problem.cpp:297-298 */
	cfg->slack_order_quirk = 1; /* Q3, ho_qp_problem.cpp:83-101, 224-225 */
	cfg->zero_kernel_col = 1; /* Q6, Eigen kernel() of a full-column-rank matrix */
	cfg->nullspace_basis = 0; /* FullPivLU, math.hpp:99-104 */
	cfg->max_iter = 2000;
	cfg->solver = 1; /* authoritative: least-squares form in long double; 0 = the literal H + eps I form in double */
}

/*
 * The chain of HoQpProblem constructors built by HoQpController::ConstructOptProblems
 * (ho_qp_controller.cpp:120-141); each level is the constructor pipeline of
 * ho_qp_problem.cpp:11-25.
 */
int hoqp_oracle_solve_tasks(const hoqp_oracle_cfg *cfg, int n, int K, const int *m, const int *p,
			    const double *A, const double *b, const double *D, const double *f,
			    double *x_levels, int *nz_levels, double *Z_levels, double *v_levels,
			    int *iters, int *status)
{
	int Ptot = 0;
	for (int k = 0; k < K; ++k)
		Ptot += p[k];
	/* InitPrevProblemValuesToDefault, ho_qp_problem.cpp:119-128 */
	int nz = n;
	double *Z = dalloc((size_t)n * (n + 1));
	for (int i = 0; i < n; ++i)
		Z[i * nz + i] = 1.0;
	double *x = dalloc(n);
	/* stacked_tasks_.D / .f are kept newest-first (StackTasks, :83-86 with
	 * ConcatenateTasks(curr, prev), task_definition.hpp:13-30); slacks oldest-first
	 * (StackSlackSolutions, :88-101). */
	double *stackD = dalloc((size_t)Ptot * n), *stackf = dalloc(Ptot), *stackv = dalloc(Ptot);
	int P = 0;
	const double *Ak = A, *bk = b, *Dk = D, *fk = f;
	int voff = 0, rc = 0;

	for (int k = 0; k < K; ++k) {
		int mk = m[k], nv = p[k]; /* InitTaskVariables, :68-81 */
		int has_eq = mk > 0, has_ineq = nv > 0;
		int N = nz + nv;
		double *AZ = dalloc((size_t)mk * nz), *H = dalloc((size_t)N * N), *c = dalloc(N);
		double *z = dalloc(nz + 1), *v = dalloc(nv + 1);
		double *tmpm = dalloc(mk + 1);
		int st = QP_OK, it = 0;

		/* ConstructHMatrix, :231-258 */
		if (has_eq && nz > 0) {
			matmul(mk, n, nz, Ak, Z, AZ);
			for (int i = 0; i < nz; ++i)
				for (int j = 0; j < nz; ++j) {
					double s = 0.0;
					for (int t = 0; t < mk; ++t)
						s += AZ[t * nz + i] * AZ[t * nz + j];
					H[i * N + j] = s + ((i == j) ? cfg->eps : 0.0);
				}
		}
		for (int i = 0; i < nv; ++i)
			H[(nz + i) * N + nz + i] = 1.0;
		/* ConstructCVector, :261-285 */
		if (has_eq) {
			matvec(mk, n, Ak, x, tmpm);
			for (int i = 0; i < mk; ++i)
				tmpm[i] -= bk[i];
			for (int j = 0; j < nz; ++j) {
				double s = 0.0;
				for (int t = 0; t < mk; ++t)
					s += AZ[t * nz + j] * tmpm[t];
				c[j] = s;
			}
		}
		/* AddIneqConstraints only if the CURRENT level has inequalities (Q2), :297-298 */
		int use_con = cfg->gate_prior_ineq ? has_ineq : (has_ineq || P > 0);
		int Mc = use_con ? 2 * nv + P : 0;
		double *Dh = dalloc((size_t)Mc * N), *fh = dalloc(Mc), *lam = dalloc(Mc);
		double *xi = dalloc(N);
		if (use_con) {
			/* ConstructDMatrix, :180-205:  [0 -I ; Dprev Z 0 ; D Z -I] */
			for (int i = 0; i < nv; ++i)
				Dh[i * N + nz + i] = -1.0;
			for (int i = 0; i < P; ++i)
				for (int j = 0; j < nz; ++j) {
					double s = 0.0;
					for (int t = 0; t < n; ++t)
						s += stackD[i * n + t] * Z[t * nz + j];
					Dh[(nv + i) * N + j] = s;
				}
			for (int i = 0; i < nv; ++i) {
				for (int j = 0; j < nz; ++j) {
					double s = 0.0;
					for (int t = 0; t < n; ++t)
						s += Dk[i * n + t] * Z[t * nz + j];
					Dh[(nv + P + i) * N + j] = s;
				}
				Dh[(nv + P + i) * N + nz + i] = -1.0;
			}
			/* ConstructFVector, :207-229 */
			for (int i = 0; i < P; ++i) {
				double s = stackf[i];
				for (int t = 0; t < n; ++t)
					s -= stackD[i * n + t] * x[t];
				fh[nv + i] = s + stackv[i];
			}
			for (int i = 0; i < nv; ++i) {
				double s = fk[i];
				for (int t = 0; t < n; ++t)
					s -= Dk[i * n + t] * x[t];
				fh[nv + P + i] = s;
				/* feasible start for the stand-in solver: z = 0, v = max(0, D x - f) */
				xi[nz + i] = MAX(0.0, -s);
			}
		}
		/* SolveQp, :337-357 */
		if (N > 0 && cfg->solver == 0) {
			/* literal form.  Pivots <= 100*eps_ are indistinguishable from the eps_*I term: the solve is
			 * the eps_ -> 0 (minimum-norm) limit of the reference's regularised QP */
			st = qp_solve(N, H, c, Mc, Dh, fh, xi, lam, cfg->max_iter, &it, 100.0 * cfg->eps);
		} else if (N > 0) {
			/* least-squares form (see lsq_primal_active_set): B = [A Z, 0 ; sqrt(eps) I, 0 ; 0, I] */
			const int has_reg = has_eq && nz > 0 && cfg->eps > 0.0;
			const int mrows = (has_eq && nz > 0) ? mk : 0;
			const int R = mrows + (has_reg ? nz : 0) + nv;
			ldbl *B = lalloc((size_t)R * N), *beta = lalloc(R), *Cl = lalloc((size_t)Mc * N), *dl = lalloc(Mc);
			ldbl *xl = lalloc(N), *ll = lalloc(Mc);
			for (int i = 0; i < mrows; ++i) {
				for (int j = 0; j < nz; ++j)
					B[i * N + j] = AZ[i * nz + j];
				ldbl r = bk[i];
				for (int t = 0; t < n; ++t)
					r -= (ldbl)Ak[i * n + t] * (ldbl)x[t];
				beta[i] = r;
			}
			if (has_reg) {
				const ldbl se = sqrtl((ldbl)cfg->eps);
				for (int j = 0; j < nz; ++j)
					B[(mrows + j) * N + j] = se;
			}
			for (int i = 0; i < nv; ++i)
				B[(R - nv + i) * N + nz + i] = 1.0L;
			for (int i = 0; i < Mc * N; ++i)
				Cl[i] = Dh[i];
			for (int i = 0; i < Mc; ++i)
				dl[i] = fh[i];
			for (int i = 0; i < N; ++i)
				xl[i] = xi[i];
			st = lsq_qp_solve(N, R, B, beta, Mc, Cl, dl, xl, ll, cfg->max_iter, &it);
			for (int i = 0; i < N; ++i)
				xi[i] = (double)xl[i];
			free(B);
			free(beta);
			free(Cl);
			free(dl);
			free(xl);
			free(ll);
		}
		memcpy(z, xi, sizeof(double) * nz);
		memcpy(v, xi + nz, sizeof(double) * nv);
		/* GetSolution, :46-51 */
		for (int i = 0; i < n; ++i) {
			double s = x[i];
			for (int j = 0; j < nz; ++j)
				s += Z[i * nz + j] * z[j];
			x[i] = s;
		}
		/* ConstructStackedNullspaceMatrix, :164-178 */
		if (has_eq && nz > 0) {
			double *Kn = dalloc((size_t)nz * (nz + 1));
			int dk;
			if (cfg->nullspace_basis == 0)
				dk = hoqp_oracle_fullpivlu_kernel(mk, nz, AZ, Kn, cfg->zero_kernel_col);
			else
				dk = householder_kernel(mk, nz, AZ, Kn);
			double *Zn = dalloc((size_t)n * (dk + 1));
			matmul(n, nz, dk, Z, Kn, Zn);
			memcpy(Z, Zn, sizeof(double) * n * dk);
			nz = dk;
			free(Kn);
			free(Zn);
		}
		/* StackTasks (newest first), :83-86 */
		if (nv > 0) {
			memmove(stackD + (size_t)nv * n, stackD, sizeof(double) * P * n);
			memmove(stackf + nv, stackf, sizeof(double) * P);
			memcpy(stackD, Dk, sizeof(double) * nv * n);
			memcpy(stackf, fk, sizeof(double) * nv);
		}
		/* StackSlackSolutions (oldest first), :88-101.  With the quirk OFF the slack
		 * stack is kept in the same (newest-first) order as the rows it relaxes. */
		if (cfg->slack_order_quirk) {
			memcpy(stackv + P, v, sizeof(double) * nv);
		} else {
			memmove(stackv + nv, stackv, sizeof(double) * P);
			memcpy(stackv, v, sizeof(double) * nv);
		}
		P += nv;
		memcpy(v_levels + voff, v, sizeof(double) * nv);
		voff += nv;
		memcpy(x_levels + (size_t)k * n, x, sizeof(double) * n);
		nz_levels[k] = nz;
		if (Z_levels) {
			double *Zk = Z_levels + (size_t)k * n * n;
			memset(Zk, 0, sizeof(double) * n * n);
			for (int i = 0; i < n; ++i)
				for (int j = 0; j < nz; ++j)
					Zk[i * n + j] = Z[i * nz + j];
		}
		if (iters)
			iters[k] = it;
		if (status)
			status[k] = st;
		rc |= st;
		Ak += (size_t)mk * n;
		bk += mk;
		Dk += (size_t)nv * n;
		fk += nv;
		free(AZ);
		free(H);
		free(c);
		free(z);
		free(v);
		free(tmpm);
		free(Dh);
		free(fh);
		free(lam);
		free(xi);
	}
	free(Z);
	free(x);
	free(stackD);
	free(stackf);
	free(stackv);
	return rc;
}

/* ------------------------------------------------------------------------- */
/* HoQpController: task construction, torque map, validity flags              */
/* ------------------------------------------------------------------------- */

void hoqp_oracle_default_wbc_params(hoqp_oracle_wbc_params *prm)
{
	prm->mu = 0.5;		/* ho_qp_controller.hpp:56 */
	prm->tau_max = 100.0;	/* :54 */
	prm->tau_min = -100.0;	/* :55 */
	prm->kp_swing = 1.0;	/* ho_qp_controller.cpp:191 */
	prm->kv_swing = 1.0;	/* :192 */
	prm->kp_base = 1.0;	/* :224 */
	prm->kv_base = 1.0;	/* :225 */
	prm->kv_rot = 1.0;	/* :260 */
	prm->friction_col_quirk = 1; /* Q1, :321-324 */
	prm->drop_base_pos_gain = 1; /* Q5, stray ';' at :236 */
}

#define NU 18 /* kNumGenVels, anymal_constants.hpp:7 */
#define NJ 12 /* kNumJoints, :5 */
#define NB 6  /* kNumTwistCoords, :4 */

/*
 * HoQpController::ConstructTasks (ho_qp_controller.cpp:147-184) and the nine builders
 * (:186-400), after SetLegCmd (:38-65).  Output: 5 levels, concatenated row-major.
 * Returns n = 18 + 3 n_c.
 */
int hoqp_oracle_wbc_build(const hoqp_oracle_wbc_params *prm, const hoqp_oracle_wbc_state *s, int *m,
			  int *p, double *A, double *b, double *D, double *f)
{
	int cl[4], sl[4], nc = 0, ns = 0;
	for (int l = 0; l < 4; ++l) { /* GetContactLegs / GetSwingLegs, :426-450 */
		if (s->contact[l] == 1)
			cl[nc++] = l;
		if (s->contact[l] == 0)
			sl[ns++] = l;
	}
	int n = NU + 3 * nc; /* :64 */
	const double *M = s->M, *h = s->h;
	m[0] = NB;
	p[0] = 0;
	m[1] = 0;
	p[1] = 2 * NJ + 4 * nc;
	m[2] = 3 * nc;
	p[2] = 0;
	m[3] = 6 + 3 * ns;
	p[3] = 0;
	m[4] = 3 * nc;
	p[4] = 0;
	int mt = 0, pt = 0;
	for (int k = 0; k < 5; ++k) {
		mt += m[k];
		pt += p[k];
	}
	memset(A, 0, sizeof(double) * mt * n);
	memset(b, 0, sizeof(double) * mt);
	memset(D, 0, sizeof(double) * pt * n);
	memset(f, 0, sizeof(double) * pt);
	/* J_c (3nc x 18) rows of contact legs, GetStackedContactJacobianInW, dynamics.cpp:195-210 */
#define JC(i, c) (s->Jfoot[(cl[(i) / 3] * 3 + (i) % 3) * NU + (c)])
#define JS(i, c) (s->Jfoot[(sl[(i) / 3] * 3 + (i) % 3) * NU + (c)])
	double *Ak = A, *bk = b;
	/* level 0: ConstructFloatingBaseEomTask, :351-366 */
	for (int r = 0; r < NB; ++r) {
		for (int c = 0; c < NU; ++c)
			Ak[r * n + c] = M[r * NU + c];
		for (int i = 0; i < 3 * nc; ++i)
			Ak[r * n + NU + i] = -JC(i, r);
		bk[r] = -h[r];
	}
	Ak += NB * n;
	bk += NB;
	/* level 1: ConstructJointTorqueTask :332-349 ++ ConstructFrictionConeTask :299-330 */
	for (int r = 0; r < NJ; ++r) {
		for (int c = 0; c < NU; ++c) {
			D[r * n + c] = M[(NB + r) * NU + c];
			D[(NJ + r) * n + c] = -M[(NB + r) * NU + c];
		}
		for (int i = 0; i < 3 * nc; ++i) {
			D[r * n + NU + i] = -JC(i, NB + r);
			D[(NJ + r) * n + NU + i] = JC(i, NB + r);
		}
		f[r] = prm->tau_max - h[NB + r];
		f[NJ + r] = -(prm->tau_min - h[NB + r]);
	}
	for (int i = 0; i < nc; ++i) {
		int col = prm->friction_col_quirk ? NB + cl[i] * 3 : NU + 3 * i;
		double *Dr = D + (size_t)(2 * NJ + 4 * i) * n;
		Dr[0 * n + col + 0] = 1.0;
		Dr[0 * n + col + 2] = -prm->mu;
		Dr[1 * n + col + 0] = -1.0;
		Dr[1 * n + col + 2] = -prm->mu;
		Dr[2 * n + col + 1] = 1.0;
		Dr[2 * n + col + 2] = -prm->mu;
		Dr[3 * n + col + 1] = -1.0;
		Dr[3 * n + col + 2] = -prm->mu;
	}
	/* level 2: ConstructNoContactMotionTask, :283-297 */
	for (int i = 0; i < 3 * nc; ++i) {
		for (int c = 0; c < NU; ++c)
			Ak[i * n + c] = JC(i, c);
		bk[i] = -s->Jdotu[cl[i / 3] * 3 + i % 3];
	}
	Ak += 3 * nc * n;
	bk += 3 * nc;
	/* level 3: base pos :217-247, base rot :249-281, swing legs :186-215 */
	for (int r = 0; r < 3; ++r) {
		for (int c = 0; c < NU; ++c)
			Ak[r * n + c] = s->Jb[(3 + r) * NU + c]; /* J_b_pos_ = rows 3..5, :113 */
		bk[r] = s->base_cmd[6 + r] + prm->kv_base * (s->base_cmd[3 + r] - s->u[3 + r]);
		if (!prm->drop_base_pos_gain)
			bk[r] += prm->kp_base * (s->base_cmd[r] - s->q[4 + r]);
	}
	for (int r = 0; r < 3; ++r) {
		for (int c = 0; c < NU; ++c)
			Ak[(3 + r) * n + c] = s->Jb[r * NU + c]; /* J_b_rot_ = rows 0..2, :112 */
		bk[3 + r] = prm->kv_rot * (0.0 - s->u[r]);
	}
	for (int i = 0; i < 3 * ns; ++i) {
		double vel = 0.0;
		for (int c = 0; c < NU; ++c) {
			Ak[(6 + i) * n + c] = JS(i, c);
			vel += JS(i, c) * s->u[c];
		}
		int li = sl[i / 3] * 3 + i % 3; /* GetSwingLegCmd, :452-466 */
		bk[6 + i] = s->leg_cmd[24 + li] + prm->kp_swing * (s->leg_cmd[li] - s->footpos[li]) +
			    prm->kv_swing * (s->leg_cmd[12 + li] - vel);
	}
	Ak += (6 + 3 * ns) * n;
	bk += 6 + 3 * ns;
	/* level 4: ConstructForceMinimizationTask, :385-400 */
	for (int i = 0; i < 3 * nc; ++i)
		Ak[i * n + NU + i] = 1.0;
#undef JC
#undef JS
	return n;
}

/*
 * HoQpController::CalcJointCmd, ho_qp_controller.cpp:7-25: tasks -> chain -> x ->
 * five CheckSolutionValid flags (helper_functions.hpp:63-91) -> q_j_ddot -> tau
 * (CalcJointTorquesCmd, :67-73).
 */
int hoqp_oracle_wbc_solve(const hoqp_oracle_cfg *cfg, const hoqp_oracle_wbc_params *prm,
			  const hoqp_oracle_wbc_state *s, hoqp_oracle_wbc_result *out)
{
	int m[5], p[5];
	double *A = dalloc(64 * 30), *b = dalloc(64), *D = dalloc(64 * 30), *f = dalloc(64);
	int n = hoqp_oracle_wbc_build(prm, s, m, p, A, b, D, f);
	double xl[5 * 30], vl[64];
	int nzl[5], it[5], stl[5];
	int rc = hoqp_oracle_solve_tasks(cfg, n, 5, m, p, A, b, D, f, xl, nzl, NULL, vl, it, stl);
	memset(out, 0, sizeof(*out));
	out->n = n;
	out->status = rc;
	const double *x = xl + 4 * n;
	memcpy(out->x, x, sizeof(double) * n);
	memcpy(out->x_levels, xl, sizeof(double) * 5 * n);
	memcpy(out->v, vl, sizeof(double) * p[1]);
	out->nv = p[1];
	for (int k = 0; k < 5; ++k) {
		out->nz_levels[k] = nzl[k];
		out->iters[k] = it[k];
	}
	for (int j = 0; j < NJ; ++j)
		out->qdd[j] = x[NB + j]; /* :23 */
	/* tau = [M_j, -J_c_j^T] x + c_j : row r of D level 1 (first 12 rows) */
	for (int r = 0; r < NJ; ++r) {
		double t = s->h[NB + r];
		for (int c = 0; c < n; ++c)
			t += D[r * n + c] * x[c];
		out->tau[r] = t;
	}
	/* CheckSolutionValid on tasks[0..4] at the FINAL x, helper_functions.hpp:63-91 */
	const double *Ak = A, *bk = b, *Dk = D, *fk = f;
	for (int k = 0; k < 5; ++k) {
		double er = 0.0, iv = -INFINITY;
		int ok = 1;
		for (int i = 0; i < m[k]; ++i) {
			double r = -bk[i];
			for (int c = 0; c < n; ++c)
				r += Ak[i * n + c] * x[c];
			er = MAX(er, fabs(r));
			if (fabs(r) > 1e-7)
				ok = 0;
		}
		for (int i = 0; i < p[k]; ++i) {
			double r = -fk[i];
			for (int c = 0; c < n; ++c)
				r += Dk[i * n + c] * x[c];
			iv = MAX(iv, r);
			if (r > 0)
				ok = 0;
			if (i < 64 && r >= -1e-7)
				out->active_mask |= (1ull << i);
		}
		out->flags[k] = ok;
		out->eq_res_max[k] = er;
		out->ineq_viol_max[k] = p[k] ? iv : 0.0;
		/* per-level optimal residual |A_k x_k - b_k| (basis-invariant) */
		double rn = 0.0;
		const double *xk = xl + (size_t)k * n;
		for (int i = 0; i < m[k]; ++i) {
			double r = -bk[i];
			for (int c = 0; c < n; ++c)
				r += Ak[i * n + c] * xk[c];
			rn += r * r;
		}
		out->level_res[k] = sqrt(rn);
		Ak += (size_t)m[k] * n;
		bk += m[k];
		Dk += (size_t)p[k] * n;
		fk += p[k];
	}
	free(A);
	free(b);
	free(D);
	free(f);
	return rc;
}

int hoqp_oracle_wbc_solve_batch(const hoqp_oracle_cfg *cfg, const hoqp_oracle_wbc_params *prm, int N,
				const hoqp_oracle_wbc_state *s, hoqp_oracle_wbc_result *out, int threads)
{
	int rc = 0;
	(void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : 1) reduction(| : rc)
#endif
	for (int i = 0; i < N; ++i)
		rc |= hoqp_oracle_wbc_solve(cfg, prm, s + i, out + i);
	return rc;
}

/* ---------------------------------------------------------------------------
 * Post-solve step.  q_j_ddot_cmd_integrator_.Integrate(q_j_ddot_cmd_) makes the
 * velocity command, q_j_dot_cmd_integrator_.Integrate(q_j_dot_cmd_) the position command
 * (whole_body_controller.cpp:148-155; integral_ += dt_ * vec, integrator.cpp:31-36); the joint
 * controller then forms tau = kp (q* - q) + kd (qd* - qd) + tau_ff (joint_controller.cpp:103-110).
 * --------------------------------------------------------------------------- */
void hoqp_oracle_wbc_poststep(int N, double dt, double kp, double kd, const double *qdd, const double *tau_ff,
			      const double *q_j, const double *qd_j, double *q_cmd, double *qd_cmd, double *tau_cmd)
{
	for (long i = 0; i < (long)N * 12; ++i) {
		qd_cmd[i] += dt * qdd[i];
		q_cmd[i] += dt * qd_cmd[i];
		if (tau_cmd)
			tau_cmd[i] = kp * (q_cmd[i] - q_j[i]) + kd * (qd_cmd[i] - qd_j[i]) + tau_ff[i];
	}
}

/* ---------------------------------------------------------------------------
 * N3: the command source of the WBC as a batch sampler.
 *
 * Gait: LegPlanner::PlanLegsMotion (leg_planner.cpp:3-12) = GenerateStanceSequence (:105-151, a swing
 * leg's foothold advances by vel_cmd * step_time per gait step), CreateLegMotionForLeg (:264-299,
 * lift-off / touch-down = the schedule's 1->0 / 0->1 transitions scanned from step 1, start / end
 * position = the stance of the step before), CreateXYLegTrajectory (:209-228, first-order hold) and
 * CreateZLegTrajectory (:230-255, 0 -> apex -> 0 at lift-off, mid-swing, touch-down).  The z curve is
 * Drake's PiecewisePolynomial::CubicWithContinuousSecondDerivatives on three knots (Drake absent;
 * version unpinned, CMakeLists.txt:22): with three knots its not-a-knot end condition degenerates and
 * Drake sets the cubic coefficients of both segments to zero, i.e. the curve is the parabola through
 * the three points -- restated from the published implementation, parity unpinned.
 * Evaluation: EvalLegPos/Vel/AccAtT (:302-366) with time_rel = fmod(t, duration); outside
 * [t_liftoff, t_touchdown] everything is zero.  The reference evaluates the POSITION polynomials for
 * velocity and acceleration as well (:337-341, :358-362); fix_derivatives = 1 evaluates the
 * derivatives instead.  Contact pattern: GetContactPatternAtT (:55-59).
 * leg_cmd = [pos(12); vel(12); acc(12)], legs LF, RF, LH, RH.
 * --------------------------------------------------------------------------- */
void hoqp_oracle_gait_sample(int N, int n_steps, double duration, const int *schedule /* [4][n_steps] */,
			     double apex, int fix_derivatives, const double *time, const double *vel_cmd /* [N][2] */,
			     const double *stance0 /* [N][4][2] */, double *leg_cmd /* [N][36] */, int *contact /* [N][4] */)
{
	const double step_time = duration / (double)n_steps;
	for (int i = 0; i < N; ++i) {
		/* negative times (never produced by the reference's ros::Time differences) count as 0 */
		const double t_rel = fmax(0.0, fmod(time[i], duration));
		int step = (int)(t_rel / step_time);
		if (step > n_steps - 1)
			step = n_steps - 1;
		if (step < 0)
			step = 0;
		for (int leg = 0; leg < 4; ++leg) {
			const int *sch = schedule + leg * n_steps;
			double pos[2] = {stance0[(i * 4 + leg) * 2], stance0[(i * 4 + leg) * 2 + 1]};
			double t_lo = -1.0, t_td = -1.0, start[2] = {0, 0}, end[2] = {0, 0};
			int last = sch[0];
			for (int g = 1; g < n_steps; ++g) {
				const double prev[2] = {pos[0], pos[1]};
				if (sch[g] != 1) {
					pos[0] += vel_cmd[i * 2] * step_time;
					pos[1] += vel_cmd[i * 2 + 1] * step_time;
				}
				if (sch[g] != last) {
					if (last == 1) {
						t_lo = step_time * g;
						last = 0;
						start[0] = prev[0];
						start[1] = prev[1];
					} else {
						t_td = step_time * g;
						last = 1;
						end[0] = prev[0];
						end[1] = prev[1];
					}
				}
			}
			double p[3] = {0, 0, 0}, v[3] = {0, 0, 0}, a[3] = {0, 0, 0};
			if (!(t_rel < t_lo || t_rel > t_td)) {
				const double T = t_td - t_lo, half = 0.5 * fabs(T);
				const double w = (t_rel - t_lo) / T;
				const double u = (t_rel - (t_lo + half)) / half;
				p[0] = start[0] + w * (end[0] - start[0]);
				p[1] = start[1] + w * (end[1] - start[1]);
				p[2] = apex * (1.0 - u * u);
				if (fix_derivatives) {
					v[0] = (end[0] - start[0]) / T;
					v[1] = (end[1] - start[1]) / T;
					v[2] = -2.0 * apex * u / half;
					a[2] = -2.0 * apex / (half * half);
				} else {
					for (int d = 0; d < 3; ++d)
						v[d] = a[d] = p[d];
				}
			}
			for (int d = 0; d < 3; ++d) {
				leg_cmd[i * 36 + leg * 3 + d] = p[d];
				leg_cmd[i * 36 + 12 + leg * 3 + d] = v[d];
				leg_cmd[i * 36 + 24 + leg * 3 + d] = a[d];
			}
			contact[i * 4 + leg] = sch[step];
		}
	}
}

/* Stand-up ramp: BasePlanner::PlanBaseStandupMotion / EvalStandup{Pos,Vel,Acc}TrajAtT
 * (base_planner.cpp:272-317): first-order hold from the current pose to (x, y, target height) over
 * `seconds`, held afterwards; velocity = the constant slope, acceleration = 0, both zero after the ramp;
 * all legs in contact (motion_planner.cpp:53-59).  base_cmd = [pos(3); vel(3); acc(3)]. */
void hoqp_oracle_standup_sample(int N, double seconds, double target_height, const double *time,
				const double *curr_pose /* [N][3] */, double *base_cmd /* [N][9] */, int *contact)
{
	for (int i = 0; i < N; ++i) {
		const double *c = curr_pose + i * 3;
		const double tgt[3] = {c[0], c[1], target_height};
		const double t = fmax(time[i], 0.0); /* Drake's FirstOrderHold::value clamps to the first break */
		for (int d = 0; d < 3; ++d) {
			const double slope = (tgt[d] - c[d]) / seconds;
			base_cmd[i * 9 + d] = (t > seconds) ? tgt[d] : c[d] + slope * t;
			base_cmd[i * 9 + 3 + d] = (t > seconds) ? 0.0 : slope;
			base_cmd[i * 9 + 6 + d] = 0.0;
		}
		for (int leg = 0; leg < 4; ++leg)
			contact[i * 4 + leg] = 1;
	}
}
