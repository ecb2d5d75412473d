// hoqp_kernels.hip -- MI355X (gfx950) kernels of the batched HO-QP solve and the C ABI.
//
// Path and boundary: see include/hoqp.h.  Algorithm: per instance
// one wavefront walks the priority levels with all state in LDS:
//   x_k, an orthonormal null-space basis Z_k (Householder), and every inequality row seen
//   so far projected into the current null space (G = D Z, g = f + v* - D x).
// A level is solved in the rotated coordinates y = Q'z of the complete orthogonal
// decomposition (A Z) Q = [L 0]:
//   min 1/2 |L y1 - r|^2 + 1/2 |(G_soft y - g_soft)_+|^2   s.t.  G_hard y <= g_hard
// by a primal active-set method on the piecewise quadratic (slacks eliminated:
// v = (G y - g)_+), started at the always-feasible y = 0.  This is the QP of
// ho_qp_problem.cpp:180-285 with xi = [z; v]; the redundant rows -v <= 0 are implied.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hoqp.h"
#include "hoqp_device.hpp"

namespace hoqp {

// ---------------------------------------------------------------------------
// kernel arguments and LDS layout
// ---------------------------------------------------------------------------
struct SolveArgs {
	int batch, n, L;
	int m[HOQP_MAX_LEVELS], p[HOQP_MAX_LEVELS];
	const double *A[HOQP_MAX_LEVELS], *b[HOQP_MAX_LEVELS], *D[HOQP_MAX_LEVELS], *f[HOQP_MAX_LEVELS];
	const int *n_inst, *m_inst, *p_inst;
	int gate_prior, slack_quirk, max_iter;
	double eps;
	hoqp_result_batch out;
	int Ptot;   // storage extent of stacked inequality rows
	int mmax;   // max_k m[k]
	int ld;	    // odd leading dimension >= n for Z, G, T
	int ncol;   // columns of the QR workspace T
	int chunk;  // rows staged per pass
};

struct Lds {
	double *Z, *x, *G, *g, *s, *c, *T, *S, *BL, *r0, *rho, *a0, *tau, *rdiag, *rhs, *chol, *tmp;
	int *piv, *cpos, *W, *colrow, *lsrow;
};

__host__ __device__ inline size_t lds_layout(const SolveArgs &a, char *base, Lds *l)
{
	size_t off = 0;
	auto take = [&](size_t bytes) {
		char *p = base ? base + off : nullptr;
		off += (bytes + 7) & ~size_t(7);
		return p;
	};
	const int n = a.n, ld = a.ld, P = a.Ptot, mm = a.mmax;
	double *Z = (double *)take(sizeof(double) * n * ld);
	double *x = (double *)take(sizeof(double) * ld);
	double *G = (double *)take(sizeof(double) * (P ? P : 1) * ld);
	double *g = (double *)take(sizeof(double) * (P + 1));
	double *s = (double *)take(sizeof(double) * (P + 1));
	double *c = (double *)take(sizeof(double) * (P + 1));
	double *T = (double *)take(sizeof(double) * ld * a.ncol);
	double *S = (double *)take(sizeof(double) * a.chunk * n);
	double *BL = (double *)take(sizeof(double) * (mm ? mm : 1) * ld);
	double *r0 = (double *)take(sizeof(double) * (mm + 1));
	double *rho = (double *)take(sizeof(double) * (mm + 1));
	double *a0 = (double *)take(sizeof(double) * (mm + 1));
	double *tau = (double *)take(sizeof(double) * ld);
	double *rdiag = (double *)take(sizeof(double) * ld);
	double *rhs = (double *)take(sizeof(double) * a.ncol);
	double *chol = (double *)take(sizeof(double) * ld * ld);
	double *tmp = (double *)take(sizeof(double) * 2 * kWave);
	int *piv = (int *)take(sizeof(int) * ld);
	int *cpos = (int *)take(sizeof(int) * a.ncol);
	int *W = (int *)take(sizeof(int) * ld);
	int *colrow = (int *)take(sizeof(int) * a.ncol);
	int *lsrow = (int *)take(sizeof(int) * (mm + 1));
	if (l) {
		*l = Lds{Z, x, G, g, s, c, T, S, BL, r0, rho, a0, tau, rdiag, rhs, chol, tmp, piv, cpos, W, colrow, lsrow};
	}
	return off;
}

// ---------------------------------------------------------------------------
// staging and projection:  rows of A or D (global, [rows][nstore]) -> LDS S ([rows][n])
// ---------------------------------------------------------------------------
__device__ inline void stage_rows(const double *src, int rows, int nstore, int n, double *S)
{
	const int total = rows * nstore;
	for (int idx = lane_id(); idx < total; idx += kWave) {
		int r = idx / nstore, l = idx - r * nstore;
		if (l < n)
			S[r * n + l] = src[idx];
	}
	__syncthreads();
}

// out(r, j) = sum_l S[r][l] Z[l][j]  for r < rows, j < nz.  colmajor: out[j + (r0+r)*ldo]
// (a column of T per row), else out[(r0+r)*ldo + j] (a row of G per row).
__device__ inline void project_rows(const double *S, int rows, int n, const double *Z, int ldz, int nz, double *out,
				    int ldo, int r0, bool colmajor)
{
	const int total = rows * nz;
	for (int idx = lane_id(); idx < total; idx += kWave) {
		int r = idx / nz, j = idx - r * nz;
		const double *sr = S + r * n;
		double acc = 0.0;
		for (int l = 0; l < n; ++l)
			acc = fma(sr[l], Z[l * ldz + j], acc);
		if (colmajor)
			out[(size_t)(r0 + r) * ldo + j] = acc;
		else
			out[(size_t)(r0 + r) * ldo + j] = acc;
	}
}

// rhs[r0 + r] = sgn * (v[r] - S[r] . x)
__device__ inline void residual_rows(const double *S, int rows, int n, const double *x, const double *v, double *out,
				     int r0)
{
	for (int r = lane_id(); r < rows; r += kWave) {
		const double *sr = S + r * n;
		double acc = v[r];
		for (int l = 0; l < n; ++l)
			acc = fma(-sr[l], x[l], acc);
		out[r0 + r] = acc;
	}
}

// ---------------------------------------------------------------------------
// least-squares solve on a finished factorization.
// The B block of T holds, in pivot positions [k1, k2), the factor R_B (rb = k2-k1 rows)
// of the nb "B" columns listed in bcols[0..nb): columns with k1 <= cpos < k2 are pivoted,
// the rest (cpos < 0) are dependent/overdetermined.  Solves min |R_B' q - rhs| (rhs per B
// column, indexed like bcols) and returns q_j in lane j (j < rb).  If eps > 0 the result
// is corrected to first order for the reference's eps*I Tikhonov term.
// ---------------------------------------------------------------------------
__device__ inline double ls_solve(const double *T, int ldt, const int *piv, const int *cpos, const double *rdiag,
				  int k1, int k2, int cb, int ce, const double *rhs, double *chol, int ldc,
				  double *tmp, double eps)
{
	const int lane = lane_id();
	const int rb = k2 - k1;
	if (rb <= 0)
		return 0.0;
	// is any B column unpivoted?
	int unp = 0;
	for (int c = cb + lane; c < ce; c += kWave)
		unp |= (cpos[c] < 0);
	const bool square = (__ballot(unp) == 0ull);
	double q = 0.0;
	if (square && eps == 0.0) {
		// forward substitution R_B' q = rhs, column oriented; lane j owns position k1 + j
		double acc = (lane < rb) ? rhs[piv[k1 + lane]] : 0.0;
		const double *mycol = (lane < rb) ? T + (size_t)piv[k1 + lane] * ldt : T;
		for (int i = 0; i < rb; ++i) {
			double qi = readlane_d(acc, i) / rdiag[k1 + i];
			if (lane == i)
				q = qi;
			if (lane > i && lane < rb)
				acc = fma(-mycol[k1 + i], qi, acc);
		}
		return q;
	}
	// corrected semi-normal equations:  S = R_B R_B',  S q = R_B rhs, one refinement step
	auto rb_entry = [&](int i, int c) -> double {  // R_B(i, column c), i in [0, rb)
		int cp = cpos[c];
		if (cp >= k1 && cp < k2) {
			int j = cp - k1;
			return (i < j) ? T[(size_t)c * ldt + k1 + i] : (i == j ? rdiag[cp] : 0.0);
		}
		return (cp < 0) ? T[(size_t)c * ldt + k1 + i] : 0.0;
	};
	for (int idx = lane; idx < rb * rb; idx += kWave) {
		int i = idx / rb, j = idx - i * rb;
		double acc = 0.0;
		for (int c = cb; c < ce; ++c)
			acc = fma(rb_entry(i, c), rb_entry(j, c), acc);
		chol[i * ldc + j] = acc;
	}
	__syncthreads();
	// Cholesky (lower, in place), lane per row
	for (int k = 0; k < rb; ++k) {
		double d = sqrt(chol[k * ldc + k]);
		__syncthreads();
		if (lane > k && lane < rb)
			chol[lane * ldc + k] /= d;
		if (lane == k)
			chol[k * ldc + k] = d;
		__syncthreads();
		if (lane > k && lane < rb) {
			double lik = chol[lane * ldc + k];
			for (int j = k + 1; j <= lane; ++j)
				chol[lane * ldc + j] = fma(-lik, chol[j * ldc + k], chol[lane * ldc + j]);
		}
		__syncthreads();
	}
	auto chol_solve = [&](double t) -> double {  // lane j holds t_j, returns (L L')^-1 t in lane j
		double acc = t, y = 0.0;
		for (int i = 0; i < rb; ++i) {
			double yi = readlane_d(acc, i) / chol[i * ldc + i];
			if (lane == i)
				y = yi;
			if (lane > i && lane < rb)
				acc = fma(-chol[lane * ldc + i], yi, acc);
		}
		acc = y;
		double z = 0.0;
		for (int i = rb - 1; i >= 0; --i) {
			double zi = readlane_d(acc, i) / chol[i * ldc + i];
			if (lane == i)
				z = zi;
			if (lane < i)
				acc = fma(-chol[i * ldc + lane], zi, acc);
		}
		return z;
	};
	// t = R_B rhs
	double t = 0.0;
	if (lane < rb)
		for (int c = cb; c < ce; ++c)
			t = fma(rb_entry(lane, c), rhs[c], t);
	q = chol_solve(t);
	// refinement: res = rhs - R_B' q ; dq = S^-1 R_B res
	if (lane < rb)
		tmp[lane] = q;
	__syncthreads();
	double t2 = 0.0;
	if (lane < rb) {
		for (int c = cb; c < ce; ++c) {
			double res = rhs[c];
			for (int i = 0; i < rb; ++i)
				res = fma(-rb_entry(i, c), tmp[i], res);
			t2 = fma(rb_entry(lane, c), res, t2);
		}
	}
	q += chol_solve(t2);
	if (eps != 0.0)
		q -= eps * chol_solve(q);
	__syncthreads();
	return (lane < rb) ? q : 0.0;
}

// ---------------------------------------------------------------------------
// the per-level solve
// ---------------------------------------------------------------------------
struct LevelState {
	int n, nz, P;  // decision vars, current null-space dimension, inequality rows stored
	int status;
};

// Equality-constrained least-squares step of the active-set method, from scratch:
//   min |B (y+p) - ...|  s.t.  G_W p = 0, minimum norm.  Columns of T: [W rows | LS rows | V rows].
// Returns p_i in lane i; k1 = rank of the working set (reflectors [0,k1) stay valid in T).
__device__ inline double eqp_step(const SolveArgs &a, const Lds &l, int nz, int rank, int mrows, int w,
				  unsigned long long vmask, int nsoft0, int *k1_out, bool *overflow)
{
	const int lane = lane_id();
	const int ld = a.ld;
	const int nV = __popcll(vmask);
	const int nb = mrows + nV;
	const int ncol = w + nb;
	*k1_out = 0;
	if (ncol > a.ncol) {
		*overflow = true;
		return 0.0;
	}
	// column -> source row map
	for (int c = lane; c < ncol; c += kWave) {
		int src;
		if (c < w)
			src = l.W[c];  // G row
		else if (c < w + mrows)
			src = -1 - (c - w);  // LS row j
		else {
			// (c - w - mrows)-th set bit of vmask
			int t = c - w - mrows;
			unsigned long long mk = vmask;
			for (int q = 0; q < t; ++q)
				mk &= mk - 1;
			src = __ffsll((long long)mk) - 1;
		}
		l.colrow[c] = src;
		l.cpos[c] = -1;
	}
	__syncthreads();
	for (int c = lane; c < ncol; c += kWave) {
		int src = l.colrow[c];
		double *col = l.T + (size_t)c * ld;
		if (src >= 0) {
			const double *row = l.G + (size_t)src * ld;
			for (int i = 0; i < nz; ++i)
				col[i] = row[i];
			l.rhs[c] = (c < w) ? 0.0 : -l.s[src];
		} else {
			int j = -1 - src;
			const double *row = l.BL + (size_t)j * ld;
			for (int i = 0; i < nz; ++i)
				col[i] = (i < rank) ? row[i] : 0.0;
			l.rhs[c] = -l.rho[j];
		}
	}
	__syncthreads();
	(void)nsoft0;
	int k1 = hh_factor(l.T, ld, nz, 0, 0, w, ncol, 1e-12, l.cpos, l.piv, l.tau, l.rdiag);
	int k2 = hh_factor(l.T, ld, nz, k1, w, ncol, ncol, 1e-11, l.cpos, l.piv, l.tau, l.rdiag);
	*k1_out = k1;
	double q = ls_solve(l.T, ld, l.piv, l.cpos, l.rdiag, k1, k2, w, ncol, l.rhs, l.chol, ld, l.tmp, 0.0);
	// u = [0 (k1) ; q ; 0],  p = Q u
	double u = 0.0;
	{
		// move q from lane j to lane k1 + j
		int srcl = lane - k1;
		double qq = __shfl(q, srcl < 0 ? 0 : srcl);
		u = (srcl >= 0 && srcl < k2 - k1) ? qq : 0.0;
	}
	return hh_apply_vec(u, nz, l.T, ld, l.piv, l.tau, 0, k2, false);
}

__device__ inline void solve_level(const SolveArgs &a, const Lds &l, LevelState &st, int inst, int k, int m, int p)
{
	const int lane = lane_id();
	const int n = st.n, ld = a.ld;
	const int nz = st.nz, P = st.P;
	const int nstore = a.n;
	const bool consider = (nz > 0) && (a.gate_prior ? (p > 0) : (p > 0 || P > 0));
	int iters = 0;

	// ---- 1. project the task into the current null space (ConstructHMatrix/CVector/
	//         DMatrix/FVector of ho_qp_problem.cpp:180-285, in factored form) ----
	for (int c = lane; c < a.ncol; c += kWave)
		l.cpos[c] = -1;
	for (int r0 = 0; r0 < m; r0 += a.chunk) {
		int rows = min(a.chunk, m - r0);
		stage_rows(a.A[k] + ((size_t)inst * a.m[k] + r0) * nstore, rows, nstore, n, l.S);
		project_rows(l.S, rows, n, l.Z, ld, nz, l.T, ld, r0, true);
		residual_rows(l.S, rows, n, l.x, a.b[k] + (size_t)inst * a.m[k] + r0, l.rhs, r0);
		__syncthreads();
	}
	for (int r0 = 0; r0 < p; r0 += a.chunk) {
		int rows = min(a.chunk, p - r0);
		stage_rows(a.D[k] + ((size_t)inst * a.p[k] + r0) * nstore, rows, nstore, n, l.S);
		project_rows(l.S, rows, n, l.Z, ld, nz, l.G, ld, P + r0, false);
		residual_rows(l.S, rows, n, l.x, a.f[k] + (size_t)inst * a.p[k] + r0, l.g, P + r0);
		__syncthreads();
	}
	const int Pn = P + p;

	// ---- 2. complete orthogonal decomposition of A Z (CalcNullSpaceProjMatrix, math.hpp:99-104,
	//         with an orthonormal Householder basis instead of the FullPivLU kernel) ----
	int rank = 0;
	if (m > 0 && nz > 0)
		rank = hh_factor(l.T, ld, nz, 0, 0, m, m, kEpsMach * (double)max(m, nz), l.cpos, l.piv, l.tau, l.rdiag);

	// ---- 3. rotate Z and every stored inequality row: Z <- Z Q, G <- G Q ----
	if (rank > 0) {
		hh_apply_right_rows(l.Z, ld, n, nz, l.T, ld, l.piv, l.tau, 0, rank);
		hh_apply_right_rows(l.G, ld, Pn, nz, l.T, ld, l.piv, l.tau, 0, rank);
	}
	__syncthreads();

	// LS rows in rotated coordinates: BL[j][0..rank) for j < m (pivoted rows first), r0 likewise
	double yi = 0.0;  // lane i holds y_i
	if (m > 0) {
		// row order: pivot positions 0..rank-1, then unpivoted columns in index order
		if (lane == 0) {
			int t = rank;
			for (int c = 0; c < m; ++c)
				if (l.cpos[c] < 0)
					l.lsrow[t++] = c;
			for (int j = 0; j < rank; ++j)
				l.lsrow[j] = l.piv[j];
		}
		__syncthreads();
		for (int j = lane; j < m; j += kWave) {
			int c = l.lsrow[j];
			double *row = l.BL + (size_t)j * ld;
			for (int i = 0; i < rank; ++i) {
				double v;
				if (j < rank)
					v = (i < j) ? l.T[(size_t)c * ld + i] : (i == j ? l.rdiag[j] : 0.0);
				else
					v = l.T[(size_t)c * ld + i];
				row[i] = v;
			}
			l.r0[j] = l.rhs[c];
		}
		__syncthreads();
	}

	if (!consider) {
		// ---- 4a. unconstrained level: y1 = argmin |R' y1 - r0| (+ eps term), y2 = 0 ----
		if (rank > 0) {
			double q = ls_solve(l.T, ld, l.piv, l.cpos, l.rdiag, 0, rank, 0, m, l.rhs, l.chol, ld, l.tmp, a.eps);
			yi = q;
		}
	} else {
		// ---- 4b. primal active set on the piecewise quadratic ----
		// s_i = G_i y - g_i (all rows), rho_j = B0_j y - r0_j
		for (int i = lane; i < Pn; i += kWave)
			l.s[i] = -l.g[i];
		for (int j = lane; j < m; j += kWave)
			l.rho[j] = -l.r0[j];
		__syncthreads();
		const unsigned long long softmask = (p >= 64) ? ~0ull : (((1ull << p) - 1ull) << P);
		const unsigned long long hardmask = (P >= 64) ? ~0ull : ((1ull << P) - 1ull);
		unsigned long long wmask = 0ull;
		int w = 0;
		// rows that start violated although they are hard: flag (no phase 1 here)
		{
			int bad = 0;
			if (lane < P)
				bad = l.s[lane] > 1e-9 * (1.0 + fabs(l.g[lane]));
			if (__ballot(bad))
				st.status |= HOQP_STATUS_INFEASIBLE_START;
		}
		bool done = false;
		for (iters = 0; iters < a.max_iter && !done; ++iters) {
			// violated soft rows
			int isV = 0;
			if (lane < Pn && ((softmask >> lane) & 1ull))
				isV = l.s[lane] > 1e-11 * (1.0 + fabs(l.g[lane]));
			const unsigned long long vmask = __ballot(isV);
			int k1 = 0;
			bool overflow = false;
			double pi = eqp_step(a, l, nz, rank, m, w, vmask, P, &k1, &overflow);
			if (overflow) {
				st.status |= HOQP_STATUS_WORKSPACE;
				break;
			}
			double pn = wave_max(fabs(pi));
			double yn = wave_max(fabs(yi));
			if (!(pn > 1e-10 * (1.0 + yn))) {
				if (w == 0) {
					done = true;
					break;
				}
				// multipliers: G_W' mu = -grad,  grad = B0' rho + G_V' s_V
				double gi = 0.0;
				if (lane < nz) {
					if (lane < rank)
						for (int j = 0; j < m; ++j)
							gi = fma(l.BL[(size_t)j * ld + lane], l.rho[j], gi);
					unsigned long long mk = vmask;
					while (mk) {
						int r = __ffsll((long long)mk) - 1;
						mk &= mk - 1;
						gi = fma(l.G[(size_t)r * ld + lane], l.s[r], gi);
					}
				}
				double gn = wave_max(fabs(gi));
				double ti = hh_apply_vec(-gi, nz, l.T, ld, l.piv, l.tau, 0, k1, true);
				// back substitution R_C mu = t (k1 x k1, pivot positions)
				double acc = ti, mu = 0.0;
				for (int i = k1 - 1; i >= 0; --i) {
					double mi = readlane_d(acc, i) / l.rdiag[i];
					if (lane == i)
						mu = mi;
					if (lane < i)
						acc = fma(-l.T[(size_t)l.piv[i] * ld + lane], mi, acc);
				}
				int wi = (lane < k1) ? l.piv[lane] : -1;  // column index == position in W
				int drop;
				double worst = wave_argmin((lane < k1) ? mu : 1e300, wi, &drop);
				if (!(worst < -1e-9 * (1.0 + gn)) || drop < 0) {
					done = true;
					break;
				}
				wmask &= ~(1ull << l.W[drop]);
				__syncthreads();
				if (lane == 0)
					for (int j = drop; j < w - 1; ++j)
						l.W[j] = l.W[j + 1];
				--w;
				__syncthreads();
				continue;
			}
			// c_i = G_i p for every row, a0_j = B0_j p
			if (lane < nz)
				l.tmp[lane] = pi;
			__syncthreads();
			double ci = 0.0, si = 0.0;
			if (lane < Pn) {
				const double *row = l.G + (size_t)lane * ld;
				for (int i = 0; i < nz; ++i)
					ci = fma(row[i], l.tmp[i], ci);
				si = l.s[lane];
				l.c[lane] = ci;
			}
			double a0 = 0.0, rj = 0.0;
			if (lane < m) {
				const double *row = l.BL + (size_t)lane * ld;
				for (int i = 0; i < rank; ++i)
					a0 = fma(row[i], l.tmp[i], a0);
				rj = l.rho[lane];
			}
			__syncthreads();
			const double q0 = wave_sum(a0 * rj), q1 = wave_sum(a0 * a0);
			// ratio test on hard rows outside the working set
			const bool isHard = lane < P && ((hardmask >> lane) & 1ull) && !((wmask >> lane) & 1ull);
			double ah = 1e300;
			if (isHard && ci > 1e-12 * pn)
				ah = fmax(0.0, -si / ci);
			int blk;
			double t_hard = wave_argmin(ah, lane, &blk);
			if (!(t_hard < 1e299))
				blk = -1;
			// exact line search on the piecewise quadratic along p
			const bool isSoft = lane < Pn && ((softmask >> lane) & 1ull);
			const double tolv = 1e-11 * (1.0 + fabs(isSoft ? l.g[lane] : 0.0));
			double tb = 1e300;
			if (isSoft && ci != 0.0 && ((si > tolv && ci < 0.0) || (si <= tolv && ci > 0.0))) {
				double t = -si / ci;
				if (t > 0.0)
					tb = t;
			}
			auto dphi = [&](double t) -> double {  // uniform t: lanes add their own row
				double sv = si + t * ci;
				double mine = (isSoft && sv > 0.0) ? ci * sv : 0.0;
				return q0 + q1 * t + wave_sum(mine);
			};
			// each lane evaluates d(phi)/dt at its own breakpoint (rows read from LDS)
			double dmine = -1.0;
			if (tb < 1e299) {
				double acc = q0 + q1 * tb;
				unsigned long long mk = softmask;
				while (mk) {
					int r = __ffsll((long long)mk) - 1;
					mk &= mk - 1;
					double cr = l.c[r];
					double sv = l.s[r] + tb * cr;
					if (sv > 0.0 && r != lane)
						acc = fma(cr, sv, acc);
				}
				dmine = acc;
			}
			double t_hi = wave_min((tb < 1e299 && dmine >= 0.0) ? tb : 1e300);
			double t_lo = wave_max((tb < t_hi) ? tb : 0.0);
			double tm = (t_hi < 1e299) ? 0.5 * (t_lo + t_hi) : t_lo + 1.0;
			double slope;
			{
				double sv = si + tm * ci;
				slope = q1 + wave_sum((isSoft && sv > 0.0) ? ci * ci : 0.0);
			}
			double d_lo = dphi(t_lo);
			double t_star;
			if (d_lo >= 0.0)
				t_star = t_lo;
			else if (slope > 0.0)
				t_star = fmin(t_lo - d_lo / slope, t_hi);
			else
				t_star = (t_hi < 1e299) ? t_hi : t_lo;
			double t = fmin(t_star, t_hard);
			yi = fma(t, pi, yi);
			if (lane < Pn)
				l.s[lane] = fma(t, ci, si);
			if (lane < m)
				l.rho[lane] = fma(t, a0, rj);
			if (t_hard <= t_star && blk >= 0) {
				if (w < ld) {
					if (lane == 0)
						l.W[w] = blk;
					wmask |= 1ull << blk;
					++w;
				} else {
					st.status |= HOQP_STATUS_WORKSPACE;
					break;
				}
			}
			__syncthreads();
		}
		if (!done)
			st.status |= HOQP_STATUS_MAX_ITER;
	}

	// ---- 5. x_k = x_{k-1} + (Z Q) y  (GetSolution, ho_qp_problem.cpp:46-51); slack bookkeeping ----
	__syncthreads();
	if (lane < ld)
		l.tmp[lane] = (lane < nz) ? yi : 0.0;
	__syncthreads();
	if (lane < n) {
		const double *row = l.Z + (size_t)lane * ld;
		double acc = l.x[lane];
		for (int i = 0; i < nz; ++i)
			acc = fma(row[i], l.tmp[i], acc);
		l.x[lane] = acc;
	}
	if (lane < Pn) {
		const double *row = l.G + (size_t)lane * ld;
		double gi = l.g[lane];
		for (int i = 0; i < nz; ++i)
			gi = fma(-row[i], l.tmp[i], gi);
		if (lane >= P) {
			// v* = (D x - f)_+ ; the row becomes hard with rhs f + v* - D x  (ConstructFVector, :207-229)
			double v = fmax(0.0, -gi);
			if (a.out.v)
				a.out.v[(size_t)inst * a.Ptot + lane] = v;
			gi = fmax(gi, 0.0);
		}
		l.g[lane] = gi;
	}
	// level residual |A_k x_k - b_k| = |B0 y - r0|
	{
		double rj = 0.0;
		if (lane < m) {
			const double *row = l.BL + (size_t)lane * ld;
			double acc = -l.r0[lane];
			for (int i = 0; i < rank; ++i)
				acc = fma(row[i], l.tmp[i], acc);
			rj = acc;
		}
		double rn = sqrt(wave_sum(rj * rj));
		if (lane == 0) {
			if (a.out.level_res)
				a.out.level_res[(size_t)inst * a.L + k] = rn;
			if (a.out.iters)
				a.out.iters[(size_t)inst * a.L + k] = iters;
			if (a.out.nz_levels)
				a.out.nz_levels[(size_t)inst * a.L + k] = nz - rank;
		}
	}
	__syncthreads();

	// ---- 6. Z_k = Z_{k-1} Q2, G <- G Q2 : drop the first `rank` rotated coordinates ----
	if (rank > 0) {
		for (int r = lane; r < n + Pn; r += kWave) {
			double *row = (r < n) ? l.Z + (size_t)r * ld : l.G + (size_t)(r - n) * ld;
			for (int i = rank; i < nz; ++i)
				row[i - rank] = row[i];
		}
	}
	st.nz = nz - rank;
	st.P = Pn;
	__syncthreads();

	if (a.out.x_levels && lane < a.n)
		a.out.x_levels[((size_t)inst * a.L + k) * a.n + lane] = (lane < n) ? l.x[lane] : 0.0;
	if (a.out.Z_levels) {
		double *Zo = a.out.Z_levels + ((size_t)inst * a.L + k) * a.n * a.n;
		for (int idx = lane; idx < a.n * a.n; idx += kWave) {
			int r = idx / a.n, c = idx - r * a.n;
			Zo[idx] = (r < n && c < st.nz) ? l.Z[(size_t)r * ld + c] : 0.0;
		}
	}
}

__global__ void __launch_bounds__(kWave) hoqp_solve_kernel(const SolveArgs a)
{
	extern __shared__ __align__(16) char smem[];
	Lds l;
	lds_layout(a, smem, &l);
	const int lane = lane_id();
	const int inst = blockIdx.x;
	if (inst >= a.batch)
		return;
	LevelState st;
	st.n = a.n_inst ? a.n_inst[inst] : a.n;
	st.nz = st.n;
	st.P = 0;
	st.status = 0;
	const int n = st.n, ld = a.ld;
	// InitPrevProblemValuesToDefault, ho_qp_problem.cpp:119-128:  Z = I, x = 0
	for (int idx = lane; idx < n * ld; idx += kWave) {
		int r = idx / ld, c = idx - r * ld;
		l.Z[idx] = (r == c) ? 1.0 : 0.0;
	}
	if (lane < ld)
		l.x[lane] = 0.0;
	if (a.out.v)
		for (int i = lane; i < a.Ptot; i += kWave)
			a.out.v[(size_t)inst * a.Ptot + i] = 0.0;
	__syncthreads();

	for (int k = 0; k < a.L; ++k) {
		int m = a.m_inst ? a.m_inst[(size_t)inst * a.L + k] : a.m[k];
		int p = a.p_inst ? a.p_inst[(size_t)inst * a.L + k] : a.p[k];
		solve_level(a, l, st, inst, k, m, p);
	}

	// ---- final x and the CheckSolutionValid flags (helper_functions.hpp:63-91) ----
	double xi = (lane < n) ? l.x[lane] : 0.0;
	if (__ballot(!isfinite(xi)))
		st.status |= HOQP_STATUS_NONFINITE;
	if (a.out.x && lane < a.n)
		a.out.x[(size_t)inst * a.n + lane] = xi;
	if (a.out.flags || a.out.eq_res_max || a.out.ineq_viol_max || a.out.active_mask) {
		unsigned long long amask = 0ull;
		int prow = 0;
		for (int k = 0; k < a.L; ++k) {
			int m = a.m_inst ? a.m_inst[(size_t)inst * a.L + k] : a.m[k];
			int p = a.p_inst ? a.p_inst[(size_t)inst * a.L + k] : a.p[k];
			double er = 0.0, iv = -1e300;
			for (int r0 = 0; r0 < m; r0 += a.chunk) {
				int rows = min(a.chunk, m - r0);
				stage_rows(a.A[k] + ((size_t)inst * a.m[k] + r0) * a.n, rows, a.n, n, l.S);
				residual_rows(l.S, rows, n, l.x, a.b[k] + (size_t)inst * a.m[k] + r0, l.rhs, 0);
				__syncthreads();
				double e = (lane < rows) ? fabs(l.rhs[lane]) : 0.0;
				er = fmax(er, wave_max(e));
				__syncthreads();
			}
			for (int r0 = 0; r0 < p; r0 += a.chunk) {
				int rows = min(a.chunk, p - r0);
				stage_rows(a.D[k] + ((size_t)inst * a.p[k] + r0) * a.n, rows, a.n, n, l.S);
				residual_rows(l.S, rows, n, l.x, a.f[k] + (size_t)inst * a.p[k] + r0, l.rhs, 0);
				__syncthreads();
				double e = (lane < rows) ? -l.rhs[lane] : -1e300;  // D x - f
				iv = fmax(iv, wave_max(e));
				unsigned long long act = __ballot(lane < rows && e >= -1e-7);
				if (prow + r0 < 64)
					amask |= act << (prow + r0);
				__syncthreads();
			}
			prow += p;
			if (lane == 0) {
				size_t o = (size_t)inst * a.L + k;
				if (a.out.flags)
					a.out.flags[o] = (!(er > 1e-7) && !(p > 0 && iv > 0.0)) ? 1 : 0;
				if (a.out.eq_res_max)
					a.out.eq_res_max[o] = er;
				if (a.out.ineq_viol_max)
					a.out.ineq_viol_max[o] = p ? iv : 0.0;
			}
		}
		if (lane == 0 && a.out.active_mask)
			a.out.active_mask[inst] = amask;
	}
	if (lane == 0 && a.out.status)
		a.out.status[inst] = st.status;
}

// ---------------------------------------------------------------------------
// WBC task construction (HoQpController::ConstructTasks, ho_qp_controller.cpp:147-184)
// ---------------------------------------------------------------------------
struct WbcArgs {
	hoqp_wbc_inputs in;
	int friction_quirk, drop_base_gain;
	double mu, tau_max, tau_min, kp_swing, kv_swing, kp_base, kv_base, kv_rot;
	// task storage (handle workspace), extents of hoqp_wbc_extents()
	double *A0, *b0, *D1, *f1, *A2, *b2, *A3, *b3, *A4, *b4;
	int *n_inst, *m_inst, *p_inst;
};

constexpr int kNU = 18, kNJ = 12, kNB = 6, kNX = 30;

__global__ void __launch_bounds__(kWave) wbc_build_kernel(const WbcArgs a)
{
	const int lane = lane_id();
	const int inst = blockIdx.x;
	if (inst >= a.in.batch)
		return;
	const double *M = a.in.M + (size_t)inst * kNU * kNU;
	const double *h = a.in.h + (size_t)inst * kNU;
	const double *Jf = a.in.Jfoot + (size_t)inst * 12 * kNU;
	const double *Jdu = a.in.Jdotu + (size_t)inst * 12;
	const double *fp = a.in.footpos + (size_t)inst * 12;
	const double *Jb = a.in.Jb ? a.in.Jb + (size_t)inst * 6 * kNU : nullptr;
	const double *q = a.in.q + (size_t)inst * 19;
	const double *u = a.in.u + (size_t)inst * kNU;
	const double *bc = a.in.base_cmd + (size_t)inst * 9;
	const double *lc = a.in.leg_cmd + (size_t)inst * 36;
	const int *ct = a.in.contact + (size_t)inst * 4;
	// GetContactLegs / GetSwingLegs, ho_qp_controller.cpp:426-450
	int cl[4], sl[4], nc = 0, ns = 0;
	for (int g = 0; g < 4; ++g) {
		if (ct[g] == 1)
			cl[nc++] = g;
		if (ct[g] == 0)
			sl[ns++] = g;
	}
	for (int g = nc; g < 4; ++g)
		cl[g] = 0;
	for (int g = ns; g < 4; ++g)
		sl[g] = 0;
	const int n = kNU + 3 * nc;  // :64
	auto JC = [&](int i, int c) { return Jf[(cl[i / 3] * 3 + i % 3) * kNU + c]; };
	auto JS = [&](int i, int c) { return Jf[(sl[i / 3] * 3 + i % 3) * kNU + c]; };
	auto JB = [&](int r, int c) { return Jb ? Jb[r * kNU + c] : (r == c ? 1.0 : 0.0); };

	if (lane == 0) {
		a.n_inst[inst] = n;
		int *mi = a.m_inst + (size_t)inst * 5, *pi = a.p_inst + (size_t)inst * 5;
		mi[0] = kNB;
		mi[1] = 0;
		mi[2] = 3 * nc;
		mi[3] = 6 + 3 * ns;
		mi[4] = 3 * nc;
		pi[0] = 0;
		pi[1] = 2 * kNJ + 4 * nc;
		pi[2] = pi[3] = pi[4] = 0;
	}
	// level 0: ConstructFloatingBaseEomTask, :351-366
	{
		double *A = a.A0 + (size_t)inst * kNB * kNX;
		for (int idx = lane; idx < kNB * kNX; idx += kWave) {
			int r = idx / kNX, c = idx - r * kNX;
			double v = 0.0;
			if (c < kNU)
				v = M[r * kNU + c];
			else if (c < n)
				v = -JC(c - kNU, r);
			A[idx] = v;
		}
		if (lane < kNB)
			a.b0[(size_t)inst * kNB + lane] = -h[lane];
	}
	// level 1: ConstructJointTorqueTask :332-349 ++ ConstructFrictionConeTask :299-330
	{
		double *D = a.D1 + (size_t)inst * 40 * kNX;
		double *f = a.f1 + (size_t)inst * 40;
		for (int idx = lane; idx < 40 * kNX; idx += kWave) {
			int r = idx / kNX, c = idx - r * kNX;
			double v = 0.0;
			if (r < 2 * kNJ) {
				int rr = (r < kNJ) ? r : r - kNJ;
				double sg = (r < kNJ) ? 1.0 : -1.0;
				if (c < kNU)
					v = sg * M[(kNB + rr) * kNU + c];
				else if (c < n)
					v = -sg * JC(c - kNU, kNB + rr);
			} else if (r < 2 * kNJ + 4 * nc) {
				int i = (r - 2 * kNJ) / 4, t = (r - 2 * kNJ) % 4;
				int col = a.friction_quirk ? kNB + cl[i] * 3 : kNU + 3 * i;
				int d = c - col;
				if (d == 2)
					v = -a.mu;
				else if (d == 0 && t < 2)
					v = (t == 0) ? 1.0 : -1.0;
				else if (d == 1 && t >= 2)
					v = (t == 2) ? 1.0 : -1.0;
			}
			D[idx] = v;
		}
		if (lane < 40) {
			double v = 0.0;
			if (lane < kNJ)
				v = a.tau_max - h[kNB + lane];
			else if (lane < 2 * kNJ)
				v = -(a.tau_min - h[kNB + lane - kNJ]);
			f[lane] = v;
		}
	}
	// level 2: ConstructNoContactMotionTask, :283-297
	{
		double *A = a.A2 + (size_t)inst * 12 * kNX;
		for (int idx = lane; idx < 12 * kNX; idx += kWave) {
			int r = idx / kNX, c = idx - r * kNX;
			A[idx] = (r < 3 * nc && c < kNU) ? JC(r, c) : 0.0;
		}
		if (lane < 12)
			a.b2[(size_t)inst * 12 + lane] = (lane < 3 * nc) ? -Jdu[cl[lane / 3] * 3 + lane % 3] : 0.0;
	}
	// level 3: base pos :217-247, base rot :249-281, swing legs :186-215
	{
		double *A = a.A3 + (size_t)inst * 18 * kNX;
		for (int idx = lane; idx < 18 * kNX; idx += kWave) {
			int r = idx / kNX, c = idx - r * kNX;
			double v = 0.0;
			if (c < kNU) {
				if (r < 3)
					v = JB(3 + r, c);
				else if (r < 6)
					v = JB(r - 3, c);
				else if (r < 6 + 3 * ns)
					v = JS(r - 6, c);
			}
			A[idx] = v;
		}
		if (lane < 18) {
			double v = 0.0;
			if (lane < 3) {
				v = bc[6 + lane] + a.kv_base * (bc[3 + lane] - u[3 + lane]);
				if (!a.drop_base_gain)
					v += a.kp_base * (bc[lane] - q[4 + lane]);
			} else if (lane < 6) {
				v = a.kv_rot * (0.0 - u[lane - 3]);
			} else if (lane < 6 + 3 * ns) {
				int i = lane - 6;
				double vel = 0.0;
				for (int c = 0; c < kNU; ++c)
					vel = fma(JS(i, c), u[c], vel);
				int li = sl[i / 3] * 3 + i % 3;
				v = lc[24 + li] + a.kp_swing * (lc[li] - fp[li]) + a.kv_swing * (lc[12 + li] - vel);
			}
			a.b3[(size_t)inst * 18 + lane] = v;
		}
	}
	// level 4: ConstructForceMinimizationTask, :385-400
	{
		double *A = a.A4 + (size_t)inst * 12 * kNX;
		for (int idx = lane; idx < 12 * kNX; idx += kWave) {
			int r = idx / kNX, c = idx - r * kNX;
			A[idx] = (r < 3 * nc && c == kNU + r) ? 1.0 : 0.0;
		}
		if (lane < 12)
			a.b4[(size_t)inst * 12 + lane] = 0.0;
	}
}

// q_j_ddot = x[6:18] (ho_qp_controller.cpp:23); tau = [M_j, -J_c_j^T] x + c_j (:67-73)
__global__ void wbc_finalize_kernel(hoqp_wbc_inputs in, const double *x, double *qdd, double *tau)
{
	int t = blockIdx.x * blockDim.x + threadIdx.x;
	int inst = t / kNJ, r = t - inst * kNJ;
	if (inst >= in.batch)
		return;
	const double *xi = x + (size_t)inst * kNX;
	if (qdd)
		qdd[(size_t)inst * kNJ + r] = xi[kNB + r];
	if (tau) {
		const double *M = in.M + (size_t)inst * kNU * kNU + (size_t)(kNB + r) * kNU;
		const double *Jf = in.Jfoot + (size_t)inst * 12 * kNU;
		const int *ct = in.contact + (size_t)inst * 4;
		double acc = in.h[(size_t)inst * kNU + kNB + r];
		for (int c = 0; c < kNU; ++c)
			acc = fma(M[c], xi[c], acc);
		int i = 0;
		for (int g = 0; g < 4; ++g)
			if (ct[g] == 1) {
				for (int d = 0; d < 3; ++d)
					acc = fma(-Jf[(g * 3 + d) * kNU + kNB + r], xi[kNU + 3 * i + d], acc);
				++i;
			}
		tau[(size_t)inst * kNJ + r] = acc;
	}
}

// ---------------------------------------------------------------------------
// device self-test of the cross-lane primitives
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kWave) selftest_kernel(int *fail)
{
	const int lane = lane_id();
	int bad = 0;
	double v = 1.0 + lane * 0.5;
	double s = wave_sum(v);
	if (fabs(s - (64.0 + 0.5 * 2016.0)) > 1e-12)
		bad |= 1;
	double mx = wave_max((lane == 37) ? 100.0 : (double)lane);
	if (mx != 100.0)
		bad |= 2;
	int idx;
	double am = wave_argmax((lane == 21 || lane == 50) ? 7.0 : -(double)lane, lane + 1000, &idx);
	if (am != 7.0 || idx != 1021)
		bad |= 4;
	double mn = wave_argmin((lane == 44) ? -3.0 : (double)lane, lane, &idx);
	if (mn != -3.0 || idx != 44)
		bad |= 8;
	double rl = readlane_d(v, 9);
	if (rl != 1.0 + 4.5)
		bad |= 16;
	am = wave_argmax(-1.0, -1, &idx);
	if (idx != -1)
		bad |= 32;
	if (bad)
		atomicOr(fail, bad);
}

}  // namespace hoqp

// ===========================================================================
// host side: handle, launches, C ABI
// ===========================================================================
using namespace hoqp;

struct hoqp_handle {
	hoqp_config cfg;
	std::string err;
	int max_lds = 0;
	// WBC task workspace (grow-only)
	int ws_batch = 0;
	double *ws_d = nullptr;
	int *ws_i = nullptr;
	// kernel timing
	bool timing = false;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};

static int fail(hoqp_handle *h, int code, const std::string &msg)
{
	if (h)
		h->err = msg;
	return code;
}

#define HIP_OK(h, call)                                                                       \
	do {                                                                                  \
		hipError_t e_ = (call);                                                       \
		if (e_ != hipSuccess)                                                         \
			return fail(h, -3, std::string(#call) + ": " + hipGetErrorString(e_)); \
	} while (0)

extern "C" void hoqp_default_config(hoqp_config *cfg)
{
	std::memset(cfg, 0, sizeof(*cfg));
	cfg->device = 0;
	cfg->semantics = HOQP_SEMANTICS_REFERENCE;
	cfg->eps = 1e-12;
	cfg->max_iter = 200;
	cfg->mu = 0.5;
	cfg->tau_max = 100.0;
	cfg->tau_min = -100.0;
	cfg->kp_swing = cfg->kv_swing = cfg->kp_base = cfg->kv_base = cfg->kv_rot = 1.0;
}

extern "C" int hoqp_create(const hoqp_config *cfg, hoqp_handle **out)
{
	if (!cfg || !out)
		return -1;
	hoqp_handle *h = new hoqp_handle();
	h->cfg = *cfg;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) {
		delete h;
		return -2;
	}
	if (hipSetDevice(cfg->device) != hipSuccess) {
		delete h;
		return -3;
	}
	int lds = 0;
	if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device) != hipSuccess)
		lds = 65536;
	h->max_lds = lds;
	*out = h;
	return 0;
}

extern "C" int hoqp_destroy(hoqp_handle *h)
{
	if (!h)
		return -1;
	(void)hipSetDevice(h->cfg.device);
	for (auto &e : h->ev) {
		(void)hipEventDestroy(e.first);
		(void)hipEventDestroy(e.second);
	}
	if (h->ws_d)
		(void)hipFree(h->ws_d);
	if (h->ws_i)
		(void)hipFree(h->ws_i);
	delete h;
	return 0;
}

extern "C" const char *hoqp_last_error_string(const hoqp_handle *h) { return h ? h->err.c_str() : "null handle"; }

static int fill_args(hoqp_handle *h, const hoqp_task_batch *t, const hoqp_result_batch *out, SolveArgs *a)
{
	if (!t || !out)
		return fail(h, -1, "null argument");
	if (t->batch < 0 || t->n < 1 || t->n > HOQP_MAX_VARS)
		return fail(h, -1, "n must be in [1, HOQP_MAX_VARS]");
	if (t->num_levels < 1 || t->num_levels > HOQP_MAX_LEVELS)
		return fail(h, -1, "num_levels must be in [1, HOQP_MAX_LEVELS]");
	std::memset(a, 0, sizeof(*a));
	a->batch = t->batch;
	a->n = t->n;
	a->L = t->num_levels;
	int P = 0, mm = 0, nineq = 0;
	for (int k = 0; k < t->num_levels; ++k) {
		if (t->m[k] < 0 || t->m[k] > HOQP_MAX_EQ_ROWS || t->p[k] < 0)
			return fail(h, -1, "level row counts out of range");
		if ((t->m[k] > 0 && (!t->A[k] || !t->b[k])) || (t->p[k] > 0 && (!t->D[k] || !t->f[k])))
			return fail(h, -1, "null task block for a level with rows");
		a->m[k] = t->m[k];
		a->p[k] = t->p[k];
		a->A[k] = t->A[k];
		a->b[k] = t->b[k];
		a->D[k] = t->D[k];
		a->f[k] = t->f[k];
		P += t->p[k];
		nineq += t->p[k] > 0;
		mm = std::max(mm, t->m[k]);
	}
	if (P > HOQP_MAX_INEQ_ROWS)
		return fail(h, -1, "more than HOQP_MAX_INEQ_ROWS inequality rows in the hierarchy");
	if ((t->m_inst != nullptr) != (t->p_inst != nullptr))
		return fail(h, -1, "m_inst and p_inst must be given together");
	a->n_inst = t->n_inst;
	a->m_inst = t->m_inst;
	a->p_inst = t->p_inst;
	a->gate_prior = (h->cfg.semantics & HOQP_Q2_GATE_PRIOR_INEQ) ? 1 : 0;
	a->slack_quirk = (h->cfg.semantics & HOQP_Q3_SLACK_ORDER) ? 1 : 0;
	if (a->slack_quirk && nineq >= 3)
		return fail(h, -4, "Q3 slack pairing with >= 3 inequality levels is not supported");
	a->max_iter = h->cfg.max_iter > 0 ? h->cfg.max_iter : 200;
	a->eps = h->cfg.eps;
	a->out = *out;
	a->Ptot = P;
	a->mmax = mm;
	a->ld = t->n | 1;
	a->ncol = std::min(128, std::max(mm + P, 1));
	a->chunk = 8;
	return 0;
}

extern "C" size_t hoqp_lds_bytes(int n, int num_levels, const int *m, const int *p)
{
	if (n < 1 || n > HOQP_MAX_VARS || num_levels < 1 || num_levels > HOQP_MAX_LEVELS || !m || !p)
		return 0;
	SolveArgs a;
	std::memset(&a, 0, sizeof(a));
	a.n = n;
	int P = 0, mm = 0;
	for (int k = 0; k < num_levels; ++k) {
		P += p[k];
		mm = std::max(mm, m[k]);
	}
	a.Ptot = P;
	a.mmax = mm;
	a.ld = n | 1;
	a.ncol = std::min(128, std::max(mm + P, 1));
	a.chunk = 8;
	return lds_layout(a, nullptr, nullptr);
}

static int launch_solve(hoqp_handle *h, SolveArgs &a, hipStream_t s)
{
	a.ld = a.n | 1;
	size_t lds = lds_layout(a, nullptr, nullptr);
	if ((int)lds > h->max_lds)
		return fail(h, -5, "LDS workspace " + std::to_string(lds) + " B exceeds the device limit");
	if (lds > 48 * 1024)
		HIP_OK(h, hipFuncSetAttribute((const void *)hoqp_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
					       (int)lds));
	if (a.batch == 0)
		return 0;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	if (h->timing && h->ev.size() < 100000) {
		HIP_OK(h, hipEventCreate(&e0));
		HIP_OK(h, hipEventCreate(&e1));
		HIP_OK(h, hipEventRecord(e0, s));
	}
	hipLaunchKernelGGL(hoqp_solve_kernel, dim3(a.batch), dim3(kWave), lds, s, a);
	HIP_OK(h, hipGetLastError());
	if (e0) {
		HIP_OK(h, hipEventRecord(e1, s));
		h->ev.emplace_back(e0, e1);
	}
	return 0;
}

extern "C" int hoqp_solve_tasks_batched(hoqp_handle *h, const hoqp_task_batch *tasks, const hoqp_result_batch *out,
					void *stream)
{
	if (!h)
		return -1;
	HIP_OK(h, hipSetDevice(h->cfg.device));
	SolveArgs a;
	int rc = fill_args(h, tasks, out, &a);
	if (rc)
		return rc;
	return launch_solve(h, a, (hipStream_t)stream);
}

// storage extents of the WBC hierarchy (n = 30; rows at n_c = 4 / n_s = 4 maxima)
static const int kWbcM[5] = {6, 0, 12, 18, 12};
static const int kWbcP[5] = {0, 40, 0, 0, 0};

extern "C" int hoqp_solve_wbc_batched(hoqp_handle *h, const hoqp_wbc_inputs *in, const hoqp_wbc_outputs *out,
				      void *stream)
{
	if (!h)
		return -1;
	if (!in || !out)
		return fail(h, -1, "null argument");
	if (in->batch < 0)
		return fail(h, -1, "negative batch");
	if (!in->M || !in->h || !in->Jfoot || !in->Jdotu || !in->footpos || !in->q || !in->u || !in->base_cmd ||
	    !in->leg_cmd || !in->contact)
		return fail(h, -1, "null WBC input array");
	if (!out->x)
		return fail(h, -1, "hoqp_wbc_outputs.x is required");
	HIP_OK(h, hipSetDevice(h->cfg.device));
	hipStream_t s = (hipStream_t)stream;
	const int B = in->batch;
	if (B == 0)
		return 0;
	// per-instance doubles: A0 180 + b0 6 + D1 1200 + f1 40 + A2 360 + b2 12 + A3 540 + b3 18 + A4 360 + b4 12
	const size_t per_d = 180 + 6 + 1200 + 40 + 360 + 12 + 540 + 18 + 360 + 12;
	if (B > h->ws_batch) {
		HIP_OK(h, hipStreamSynchronize(s));
		if (h->ws_d)
			HIP_OK(h, hipFree(h->ws_d));
		if (h->ws_i)
			HIP_OK(h, hipFree(h->ws_i));
		h->ws_d = nullptr;
		h->ws_i = nullptr;
		h->ws_batch = 0;
		HIP_OK(h, hipMalloc(&h->ws_d, sizeof(double) * per_d * B));
		HIP_OK(h, hipMalloc(&h->ws_i, sizeof(int) * 11 * B));
		h->ws_batch = B;
	}
	WbcArgs w;
	std::memset(&w, 0, sizeof(w));
	w.in = *in;
	w.friction_quirk = (h->cfg.semantics & HOQP_Q1_FRICTION_COLS) ? 1 : 0;
	w.drop_base_gain = (h->cfg.semantics & HOQP_Q5_DROP_BASE_POS_GAIN) ? 1 : 0;
	w.mu = h->cfg.mu;
	w.tau_max = h->cfg.tau_max;
	w.tau_min = h->cfg.tau_min;
	w.kp_swing = h->cfg.kp_swing;
	w.kv_swing = h->cfg.kv_swing;
	w.kp_base = h->cfg.kp_base;
	w.kv_base = h->cfg.kv_base;
	w.kv_rot = h->cfg.kv_rot;
	double *d = h->ws_d;
	w.A0 = d;
	d += (size_t)180 * B;
	w.b0 = d;
	d += (size_t)6 * B;
	w.D1 = d;
	d += (size_t)1200 * B;
	w.f1 = d;
	d += (size_t)40 * B;
	w.A2 = d;
	d += (size_t)360 * B;
	w.b2 = d;
	d += (size_t)12 * B;
	w.A3 = d;
	d += (size_t)540 * B;
	w.b3 = d;
	d += (size_t)18 * B;
	w.A4 = d;
	d += (size_t)360 * B;
	w.b4 = d;
	w.n_inst = h->ws_i;
	w.m_inst = h->ws_i + B;
	w.p_inst = h->ws_i + (size_t)6 * B;
	hipLaunchKernelGGL(wbc_build_kernel, dim3(B), dim3(kWave), 0, s, w);
	HIP_OK(h, hipGetLastError());

	hoqp_task_batch t;
	std::memset(&t, 0, sizeof(t));
	t.batch = B;
	t.n = kNX;
	t.num_levels = 5;
	for (int k = 0; k < 5; ++k) {
		t.m[k] = kWbcM[k];
		t.p[k] = kWbcP[k];
	}
	t.A[0] = w.A0;
	t.b[0] = w.b0;
	t.D[1] = w.D1;
	t.f[1] = w.f1;
	t.A[2] = w.A2;
	t.b[2] = w.b2;
	t.A[3] = w.A3;
	t.b[3] = w.b3;
	t.A[4] = w.A4;
	t.b[4] = w.b4;
	t.n_inst = w.n_inst;
	t.m_inst = w.m_inst;
	t.p_inst = w.p_inst;
	hoqp_result_batch r;
	std::memset(&r, 0, sizeof(r));
	r.x = out->x;
	r.x_levels = out->x_levels;
	r.v = out->v;
	r.level_res = out->level_res;
	r.nz_levels = out->nz_levels;
	r.iters = out->iters;
	r.status = out->status;
	r.flags = out->flags;
	r.eq_res_max = out->eq_res_max;
	r.ineq_viol_max = out->ineq_viol_max;
	r.active_mask = out->active_mask;
	SolveArgs a;
	int rc = fill_args(h, &t, &r, &a);
	if (rc)
		return rc;
	rc = launch_solve(h, a, s);
	if (rc)
		return rc;
	if (out->qdd || out->tau) {
		int threads = B * kNJ;
		hipLaunchKernelGGL(wbc_finalize_kernel, dim3((threads + 255) / 256), dim3(256), 0, s, *in, out->x, out->qdd,
				   out->tau);
		HIP_OK(h, hipGetLastError());
	}
	return 0;
}

// ---------------------------------------------------------------------------
// host-buffer convenience entry points (single-instance drop-in use: the reference calls
// CalcJointCmd once per 2.5 ms cycle with host Eigen data, whole_body_controller.cpp:97-107)
// ---------------------------------------------------------------------------
namespace {
struct DevBuf {
	std::vector<void *> ptrs;
	~DevBuf()
	{
		for (void *p : ptrs)
			(void)hipFree(p);
	}
	template <typename T>
	hipError_t up(const T *host, size_t count, const T **dev)
	{
		*dev = nullptr;
		if (!host || count == 0)
			return hipSuccess;
		void *d = nullptr;
		hipError_t e = hipMalloc(&d, sizeof(T) * count);
		if (e != hipSuccess)
			return e;
		ptrs.push_back(d);
		*dev = (const T *)d;
		return hipMemcpy(d, host, sizeof(T) * count, hipMemcpyHostToDevice);
	}
	template <typename T>
	hipError_t out(T *host, size_t count, T **dev)
	{
		*dev = nullptr;
		if (!host || count == 0)
			return hipSuccess;
		void *d = nullptr;
		hipError_t e = hipMalloc(&d, sizeof(T) * count);
		if (e != hipSuccess)
			return e;
		ptrs.push_back(d);
		*dev = (T *)d;
		return hipMemset(d, 0, sizeof(T) * count);
	}
};
template <typename T>
hipError_t down(T *host, const T *dev, size_t count)
{
	if (!host || !dev || count == 0)
		return hipSuccess;
	return hipMemcpy(host, dev, sizeof(T) * count, hipMemcpyDeviceToHost);
}
}  // namespace

extern "C" int hoqp_solve_tasks_host(hoqp_handle *h, const hoqp_task_batch *t, const hoqp_result_batch *o)
{
	if (!h)
		return -1;
	if (!t || !o)
		return fail(h, -1, "null argument");
	if (t->batch < 0 || t->n < 1 || t->n > HOQP_MAX_VARS || t->num_levels < 1 || t->num_levels > HOQP_MAX_LEVELS)
		return fail(h, -1, "bad extents");
	HIP_OK(h, hipSetDevice(h->cfg.device));
	DevBuf buf;
	hoqp_task_batch dt = *t;
	const size_t B = t->batch, n = t->n, L = t->num_levels;
	size_t P = 0;
	for (size_t k = 0; k < L; ++k) {
		if (t->m[k] < 0 || t->p[k] < 0)
			return fail(h, -1, "negative row count");
		HIP_OK(h, buf.up(t->A[k], B * t->m[k] * n, &dt.A[k]));
		HIP_OK(h, buf.up(t->b[k], B * t->m[k], &dt.b[k]));
		HIP_OK(h, buf.up(t->D[k], B * t->p[k] * n, &dt.D[k]));
		HIP_OK(h, buf.up(t->f[k], B * t->p[k], &dt.f[k]));
		P += t->p[k];
	}
	HIP_OK(h, buf.up(t->n_inst, B, &dt.n_inst));
	HIP_OK(h, buf.up(t->m_inst, B * L, &dt.m_inst));
	HIP_OK(h, buf.up(t->p_inst, B * L, &dt.p_inst));
	hoqp_result_batch d;
	std::memset(&d, 0, sizeof(d));
	HIP_OK(h, buf.out(o->x, B * n, &d.x));
	HIP_OK(h, buf.out(o->x_levels, B * L * n, &d.x_levels));
	HIP_OK(h, buf.out(o->v, B * P, &d.v));
	HIP_OK(h, buf.out(o->level_res, B * L, &d.level_res));
	HIP_OK(h, buf.out(o->Z_levels, B * L * n * n, &d.Z_levels));
	HIP_OK(h, buf.out(o->nz_levels, B * L, &d.nz_levels));
	HIP_OK(h, buf.out(o->iters, B * L, &d.iters));
	HIP_OK(h, buf.out(o->status, B, &d.status));
	HIP_OK(h, buf.out(o->flags, B * L, &d.flags));
	HIP_OK(h, buf.out(o->eq_res_max, B * L, &d.eq_res_max));
	HIP_OK(h, buf.out(o->ineq_viol_max, B * L, &d.ineq_viol_max));
	HIP_OK(h, buf.out(o->active_mask, B, &d.active_mask));
	int rc = hoqp_solve_tasks_batched(h, &dt, &d, nullptr);
	if (rc)
		return rc;
	HIP_OK(h, hipStreamSynchronize(nullptr));
	HIP_OK(h, down(o->x, d.x, B * n));
	HIP_OK(h, down(o->x_levels, d.x_levels, B * L * n));
	HIP_OK(h, down(o->v, d.v, B * P));
	HIP_OK(h, down(o->level_res, d.level_res, B * L));
	HIP_OK(h, down(o->Z_levels, d.Z_levels, B * L * n * n));
	HIP_OK(h, down(o->nz_levels, d.nz_levels, B * L));
	HIP_OK(h, down(o->iters, d.iters, B * L));
	HIP_OK(h, down(o->status, d.status, B));
	HIP_OK(h, down(o->flags, d.flags, B * L));
	HIP_OK(h, down(o->eq_res_max, d.eq_res_max, B * L));
	HIP_OK(h, down(o->ineq_viol_max, d.ineq_viol_max, B * L));
	HIP_OK(h, down(o->active_mask, d.active_mask, B));
	return 0;
}

extern "C" int hoqp_solve_wbc_host(hoqp_handle *h, const hoqp_wbc_inputs *in, const hoqp_wbc_outputs *o)
{
	if (!h)
		return -1;
	if (!in || !o)
		return fail(h, -1, "null argument");
	if (in->batch < 0)
		return fail(h, -1, "negative batch");
	HIP_OK(h, hipSetDevice(h->cfg.device));
	DevBuf buf;
	const size_t B = in->batch;
	hoqp_wbc_inputs di = *in;
	HIP_OK(h, buf.up(in->M, B * 324, &di.M));
	HIP_OK(h, buf.up(in->h, B * 18, &di.h));
	HIP_OK(h, buf.up(in->Jfoot, B * 216, &di.Jfoot));
	HIP_OK(h, buf.up(in->Jdotu, B * 12, &di.Jdotu));
	HIP_OK(h, buf.up(in->footpos, B * 12, &di.footpos));
	HIP_OK(h, buf.up(in->Jb, B * 108, &di.Jb));
	HIP_OK(h, buf.up(in->q, B * 19, &di.q));
	HIP_OK(h, buf.up(in->u, B * 18, &di.u));
	HIP_OK(h, buf.up(in->base_cmd, B * 9, &di.base_cmd));
	HIP_OK(h, buf.up(in->leg_cmd, B * 36, &di.leg_cmd));
	HIP_OK(h, buf.up(in->contact, B * 4, &di.contact));
	hoqp_wbc_outputs d;
	std::memset(&d, 0, sizeof(d));
	double *xtmp = nullptr;
	HIP_OK(h, buf.out(o->x ? o->x : (double *)1, B * 30, &xtmp));
	d.x = xtmp;
	HIP_OK(h, buf.out(o->qdd, B * 12, &d.qdd));
	HIP_OK(h, buf.out(o->tau, B * 12, &d.tau));
	HIP_OK(h, buf.out(o->v, B * 40, &d.v));
	HIP_OK(h, buf.out(o->flags, B * 5, &d.flags));
	HIP_OK(h, buf.out(o->eq_res_max, B * 5, &d.eq_res_max));
	HIP_OK(h, buf.out(o->ineq_viol_max, B * 5, &d.ineq_viol_max));
	HIP_OK(h, buf.out(o->level_res, B * 5, &d.level_res));
	HIP_OK(h, buf.out(o->active_mask, B, &d.active_mask));
	HIP_OK(h, buf.out(o->status, B, &d.status));
	HIP_OK(h, buf.out(o->iters, B * 5, &d.iters));
	HIP_OK(h, buf.out(o->nz_levels, B * 5, &d.nz_levels));
	HIP_OK(h, buf.out(o->x_levels, B * 150, &d.x_levels));
	int rc = hoqp_solve_wbc_batched(h, &di, &d, nullptr);
	if (rc)
		return rc;
	HIP_OK(h, hipStreamSynchronize(nullptr));
	HIP_OK(h, down(o->x, (const double *)d.x, B * 30));
	HIP_OK(h, down(o->qdd, (const double *)d.qdd, B * 12));
	HIP_OK(h, down(o->tau, (const double *)d.tau, B * 12));
	HIP_OK(h, down(o->v, (const double *)d.v, B * 40));
	HIP_OK(h, down(o->flags, (const int *)d.flags, B * 5));
	HIP_OK(h, down(o->eq_res_max, (const double *)d.eq_res_max, B * 5));
	HIP_OK(h, down(o->ineq_viol_max, (const double *)d.ineq_viol_max, B * 5));
	HIP_OK(h, down(o->level_res, (const double *)d.level_res, B * 5));
	HIP_OK(h, down(o->active_mask, (const unsigned long long *)d.active_mask, B));
	HIP_OK(h, down(o->status, (const int *)d.status, B));
	HIP_OK(h, down(o->iters, (const int *)d.iters, B * 5));
	HIP_OK(h, down(o->nz_levels, (const int *)d.nz_levels, B * 5));
	HIP_OK(h, down(o->x_levels, (const double *)d.x_levels, B * 150));
	return 0;
}

extern "C" int hoqp_synchronize(hoqp_handle *h, void *stream)
{
	if (!h)
		return -1;
	HIP_OK(h, hipSetDevice(h->cfg.device));
	HIP_OK(h, hipStreamSynchronize((hipStream_t)stream));
	return 0;
}

extern "C" int hoqp_kernel_time_ms(hoqp_handle *h, int reset, double *avg_ms, int *launches)
{
	if (!h)
		return -1;
	HIP_OK(h, hipSetDevice(h->cfg.device));
	double tot = 0.0;
	int cnt = 0;
	for (auto &e : h->ev) {
		float ms = 0.f;
		HIP_OK(h, hipEventSynchronize(e.second));
		HIP_OK(h, hipEventElapsedTime(&ms, e.first, e.second));
		tot += ms;
		++cnt;
	}
	if (avg_ms)
		*avg_ms = cnt ? tot / cnt : 0.0;
	if (launches)
		*launches = cnt;
	if (reset) {
		for (auto &e : h->ev) {
			(void)hipEventDestroy(e.first);
			(void)hipEventDestroy(e.second);
		}
		h->ev.clear();
		h->timing = (reset == 1);
	}
	return 0;
}

extern "C" int hoqp_selftest(hoqp_handle *h)
{
	if (!h)
		return -1;
	HIP_OK(h, hipSetDevice(h->cfg.device));
	int *d = nullptr, host = 0;
	HIP_OK(h, hipMalloc(&d, sizeof(int)));
	HIP_OK(h, hipMemset(d, 0, sizeof(int)));
	hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(kWave), 0, 0, d);
	HIP_OK(h, hipGetLastError());
	HIP_OK(h, hipMemcpy(&host, d, sizeof(int), hipMemcpyDeviceToHost));
	HIP_OK(h, hipFree(d));
	if (host)
		return fail(h, host, "wavefront primitive self-test failed, mask " + std::to_string(host));
	return 0;
}
