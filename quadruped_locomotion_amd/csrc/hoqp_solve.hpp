// hoqp_solve.hpp -- the per-instance HO-QP solve (device code, gfx950).
//
// One workgroup == one wavefront == one instance.  Master copies of the small matrices live
// in LDS; every O(n^2)/O(n^3) loop runs on REGISTER-resident rows/columns (one row or column
// per lane, compile-time unrolled over NR = 32 entries), so the inner loops are pure fp64
// FMAs with broadcast LDS reads of the shared operand (Householder vector, staged task row).
//
// Algorithm: DESIGN.md section 3.  Reference: src/control/ho_qp/ho_qp_problem.cpp:11-357.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/hoqp.h"
#include "hoqp_device.hpp"

namespace hoqp {

constexpr int NR = 32;	  // register-array length, >= HOQP_MAX_VARS
constexpr int LDV = 34;	  // leading dimension of the reflector store (even: 16-byte aligned rows)
constexpr int SROW = 32;  // row stride of the task staging buffer
constexpr int XROWS = 16; // dumped unpivoted columns (over-determined least squares); aliases S
constexpr int CHUNK = 16; // task rows staged per pass

static_assert(NR >= HOQP_MAX_VARS, "register arrays must cover n");

struct SolveArgs {
	int batch, n, L;
	int m[HOQP_MAX_LEVELS], p[HOQP_MAX_LEVELS];
	const double *A[HOQP_MAX_LEVELS], *b[HOQP_MAX_LEVELS], *D[HOQP_MAX_LEVELS], *f[HOQP_MAX_LEVELS];
	const int *n_inst, *m_inst, *p_inst;
	int gate_prior, slack_quirk, max_iter;
	double eps;
	hoqp_result_batch out;
	int Ptot;  // storage extent of stacked inequality rows (<= 64)
	int mmax;  // max_k m[k] (<= 32)
	int ld;	   // odd leading dimension >= n of Z, G, BL, X
	int use_bl;	   // some level needs its equality rows inside the active-set loop
	int last_ineq_use; // last level that looks at stored inequality rows (rows are dropped after it)
	unsigned long long *dbg;  // HOQP_PROFILE builds only: [batch][8] cycle stamps per section
};

// byte offsets of the LDS arrays (shared by host sizing and device addressing)
struct LdsOff {
	int Z, x, G, g, s, c, TV, S, BL, r0, rho, tau, rdiag, rhs, X, tmp, pivlane, W, xlane, total;
};

__host__ __device__ inline LdsOff lds_offsets(const SolveArgs &a)
{
	LdsOff o;
	int off = 0;
	auto take = [&](int bytes) {
		int p = off;
		off += (bytes + 15) & ~15;
		return p;
	};
	const int n = a.n, ld = a.ld, P = a.Ptot, mm = a.mmax;
	const int tv = (LDV * n > ld * mm) ? LDV * n : ld * mm;
	o.Z = take(8 * n * ld);
	o.x = take(8 * NR);
	o.G = take(8 * (P ? P : 1) * ld);
	o.g = take(8 * (P + 1));
	o.s = take(8 * (P + 1));
	o.c = take(8 * (P + 1));
	o.TV = take(8 * tv);
	o.S = take(8 * ((CHUNK * SROW > XROWS * ld) ? CHUNK * SROW : XROWS * ld));
	o.BL = take(8 * ((mm && a.use_bl) ? mm : 1) * ld);
	o.r0 = take(8 * (mm + 1));
	o.rho = take(8 * (mm + 1));
	o.tau = take(8 * NR);
	o.rdiag = take(8 * NR);
	o.rhs = take(8 * kWave);
	o.X = o.S;  // the staging buffer is dead whenever X is live
	o.tmp = take(8 * kWave);
	o.pivlane = take(4 * NR);
	o.W = take(4 * NR);
	o.xlane = take(4 * XROWS);
	o.total = off + 256;  // readable tail: full-width (NR) row reads may run past the last array
	return o;
}

#ifdef __HIP_DEVICE_COMPILE__
struct Lds {
	ldsd *Z, *x, *G, *g, *s, *c, *TV, *S, *BL, *r0, *rho, *tau, *rdiag, *rhs, *X, *tmp;
	ldsi *pivlane, *W, *xlane;
};

__device__ inline Lds lds_bind(const SolveArgs &a, HOQP_LDS char *base)
{
	const LdsOff o = lds_offsets(a);
	Lds l;
	l.Z = (ldsd *)(base + o.Z);
	l.x = (ldsd *)(base + o.x);
	l.G = (ldsd *)(base + o.G);
	l.g = (ldsd *)(base + o.g);
	l.s = (ldsd *)(base + o.s);
	l.c = (ldsd *)(base + o.c);
	l.TV = (ldsd *)(base + o.TV);
	l.S = (ldsd *)(base + o.S);
	l.BL = (ldsd *)(base + o.BL);
	l.r0 = (ldsd *)(base + o.r0);
	l.rho = (ldsd *)(base + o.rho);
	l.tau = (ldsd *)(base + o.tau);
	l.rdiag = (ldsd *)(base + o.rdiag);
	l.rhs = (ldsd *)(base + o.rhs);
	l.X = (ldsd *)(base + o.X);
	l.tmp = (ldsd *)(base + o.tmp);
	l.pivlane = (ldsi *)(base + o.pivlane);
	l.W = (ldsi *)(base + o.W);
	l.xlane = (ldsi *)(base + o.xlane);
	return l;
}

// Loads are issued unconditionally and masked afterwards: a load behind a condition becomes a
// branch + load + wait per element (measured: the guarded form serialised every LDS read).
// Callers keep the ADDRESS valid (clamped index); LDS has a readable tail pad for rows that
// are read to their full NR extent.
__device__ __forceinline__ double sel(bool on, double v) { return on ? v : 0.0; }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---------------------------------------------------------------------------
// Static-index row loops.  Register arrays may only be indexed by compile-time constants, but
// the active row range [k, nr) is run-time (wave-uniform).  rows_from2 walks the rows in
// blocks of 8 (pairs inside a block): blocks entirely outside [k, nr) are skipped by a scalar
// branch, blocks entirely inside run unmasked, only the block containing k carries
// per-element masks.  body(i, on0, on1) handles rows i, i+1; onX == false means "row < k".
// ---------------------------------------------------------------------------
template <typename Body>
__device__ __forceinline__ void rows_from2(int k, int nr, Body body)
{
#pragma unroll
	for (int b = 0; b < NR / 8; ++b) {
		if (8 * b + 8 <= k || 8 * b >= nr)
			continue;
		if (8 * b >= k) {
#pragma unroll
			for (int j = 0; j < 8; j += 2)
				body(8 * b + j, true, true);
		} else {
#pragma unroll
			for (int j = 0; j < 8; j += 2)
				body(8 * b + j, 8 * b + j >= k, 8 * b + j + 1 >= k);
		}
	}
}

// rows [0, nr) in pairs, whole blocks of 8 skipped beyond nr (entries in [nr, 8*ceil) must be benign)
template <typename Body>
__device__ __forceinline__ void rows_upto2(int nr, Body body)
{
#pragma unroll
	for (int b = 0; b < NR / 8; ++b) {
		if (8 * b >= nr)
			continue;
#pragma unroll
		for (int j = 0; j < 8; j += 2)
			body(8 * b + j);
	}
}

// ---------------------------------------------------------------------------
// Householder QR with column pivoting, ONE COLUMN PER LANE IN REGISTERS.
//   col[0..nr)  : this lane's column (entries >= nr must be 0)
//   mypos       : -1 = active (unpivoted) column, >= 0 = pivoted at that step, -2 = no column
//   cand        : this lane's column may be chosen as a pivot in this call
// Step k: the pivot lane dumps its column into VR[k*LDV + i]; rows >= k of that row are the
// Householder vector (v_k at i == k), rows < k the R entries R(i, k); rdiag[k] = R(k, k),
// tau[k] = 2 / v'v, pivlane[k] = pivot lane.  Stops when the largest remaining candidate
// column norm is <= rtol * (first pivot norm of this call).  Returns the new step count.
// ---------------------------------------------------------------------------
__device__ inline int hh_factor_reg(double (&col)[NR], int &mypos, bool cand, int nr, int k0, double rtol,
				    const Lds &l)
{
	const int lane = lane_id();
	ldsd *VR = l.TV;
	int k = k0;
	double first = -1.0;
	while (k < nr) {
		double s0 = 0.0, s1 = 0.0;
		rows_from2(k, nr, [&](int i, bool on0, bool on1) {
			const double a = on0 ? col[i] : 0.0;
			const double b = on1 ? col[i + 1] : 0.0;
			s0 = fma(a, a, s0);
			s1 = fma(b, b, s1);
		});
		const bool mine = (mypos == -1) && cand;
		int pc;
		double best = wave_argmax(mine ? (s0 + s1) : -1.0, mine ? lane : -1, &pc);
		if (pc < 0 || !(best > 0.0))
			break;
		const double nrm = sqrt(best);
		if (first < 0.0)
			first = nrm;
		if (!(nrm > rtol * first))
			break;
		ldsd *vr = VR + k * LDV;
		if (lane == pc) {
			ldsd2 *w2 = (ldsd2 *)vr;
			rows_upto2(nr, [&](int i) {
				d2v t;
				t.x = col[i];
				t.y = col[i + 1];
				w2[i >> 1] = t;
			});
			mypos = k;
		}
		__syncthreads();
		const double alpha = vr[k];
		const double beta = (alpha >= 0.0) ? -nrm : nrm;
		const double v0 = alpha - beta;
		const double tk = 1.0 / (nrm * (nrm + fabs(alpha)));
		if (mypos == -1) {
			double vv[NR];
			const ldsd2 *vr2 = (const ldsd2 *)vr;
			double d0 = 0.0, d1 = 0.0;
			rows_from2(k, nr, [&](int i, bool on0, bool on1) {
				const d2v t = vr2[i >> 1];
				vv[i] = on0 ? ((i == k) ? v0 : t.x) : 0.0;
				vv[i + 1] = on1 ? ((i + 1 == k) ? v0 : t.y) : 0.0;
				d0 = fma(vv[i], col[i], d0);
				d1 = fma(vv[i + 1], col[i + 1], d1);
			});
			const double d = (d0 + d1) * tk;
			rows_from2(k, nr, [&](int i, bool, bool) {
				col[i] = fma(-d, vv[i], col[i]);
				col[i + 1] = fma(-d, vv[i + 1], col[i + 1]);
			});
		}
		if (lane == 0) {
			vr[k] = v0;  // program order: after every lane's read of vr[k] above
			l.tau[k] = tk;
			l.rdiag[k] = beta;
			l.pivlane[k] = pc;
		}
		++k;
	}
	__syncthreads();
	return k;
}

// rows <- rows * H_ka ... H_(kb-1): reflectors [ka, kb) applied from the right to row-major
// rows (one row per lane, in registers).  Rows r < n0 come from A0, the rest from A1.
__device__ inline void hh_apply_right_rows_reg(ldsd *A0, int n0, ldsd *A1, int n1, int ldr, int nr, int ka,
					       int kb, const Lds &l)
{
	const ldsd *VR = l.TV;
	for (int r = lane_id(); r < n0 + n1; r += kWave) {
		ldsd *rowp = (r < n0) ? A0 + r * ldr : A1 + (r - n0) * ldr;
		double row[NR];
#pragma unroll
		for (int i = 0; i < NR; ++i)
			row[i] = sel(i < nr, rowp[i]);
		for (int k = ka; k < kb; ++k) {
			const ldsd2 *vr2 = (const ldsd2 *)(VR + k * LDV);
			double vv[NR];
			double d0 = 0.0, d1 = 0.0;
			rows_from2(k, nr, [&](int i, bool on0, bool on1) {
				const d2v t = vr2[i >> 1];
				vv[i] = on0 ? t.x : 0.0;
				vv[i + 1] = on1 ? t.y : 0.0;
				d0 = fma(vv[i], row[i], d0);
				d1 = fma(vv[i + 1], row[i + 1], d1);
			});
			const double d = (d0 + d1) * l.tau[k];
			rows_from2(k, nr, [&](int i, bool, bool) {
				row[i] = fma(-d, vv[i], row[i]);
				row[i + 1] = fma(-d, vv[i + 1], row[i + 1]);
			});
		}
#pragma unroll
		for (int i = 0; i < NR; ++i)
			if (i < nr)
				rowp[i] = row[i];
	}
	__syncthreads();
}

// One entry per lane.  forward: vec <- H_(kb-1) ... H_ka vec ; else vec <- H_ka ... H_(kb-1) vec.
__device__ inline double hh_apply_vec(double vi, int nr, int ka, int kb, bool forward, const Lds &l)
{
	const int lane = lane_id();
	for (int t = 0; t < kb - ka; ++t) {
		int k = forward ? ka + t : kb - 1 - t;
		double v = sel(lane >= k && lane < nr, l.TV[k * LDV + (lane & (NR - 1))]);
		double s = wave_sum(v * vi) * l.tau[k];
		vi = fma(-s, v, vi);
	}
	return vi;
}

// ---------------------------------------------------------------------------
// Least squares on a finished factorization.  Pivot positions [k1, k2) hold the factor R_B
// (rb = k2 - k1 rows) of the "B" columns; `rhs_c` is this lane's right-hand side (lane ==
// column); `isB` marks lanes that own a B column, mypos their pivot position (or -1).
// Solves min |R_B' q - rhs| and returns q_j in lane j (j < rb); *res_norm2 = |residual|^2.
// eps != 0 adds the first order effect of the reference's eps*I term
// (ho_qp_problem.cpp:244-246):  q <- q - eps (R_B R_B')^-1 q.
// Unpivoted B columns (over-determined case) must have been dumped by the caller to
// l.X[u*ld + i] with their lanes listed in l.xlane[0..nunp).
// ---------------------------------------------------------------------------
__device__ inline double ls_solve_reg(int k1, int k2, double rhs_c, bool isB, int mypos, int nunp, double eps,
				      int ld, double *res_norm2, const Lds &l)
{
	const int lane = lane_id();
	const int rb = k2 - k1;
	const ldsd *VR = l.TV;
	if (rb <= 0) {
		if (res_norm2)
			*res_norm2 = wave_sum(isB ? rhs_c * rhs_c : 0.0);
		return 0.0;
	}
	__syncthreads();
	l.rhs[lane] = rhs_c;
	__syncthreads();
	const bool act = lane < rb;
	const int pl = l.pivlane[k1 + (act ? lane : 0)];
	const double rp = sel(act, l.rhs[pl]);  // rhs by pivot position
	const double inv = sel(act, 1.0 / l.rdiag[k1 + (act ? lane : 0)]);
	// row `lane` of L = R_B' (pivoted part): L(j, i) = R(i, j) = VR[(k1+j)*LDV + k1+i], i < j
	double Lrow[NR];
	{
		const ldsd *vr = VR + (k1 + (act ? lane : 0)) * LDV + k1;
#pragma unroll
		for (int i = 0; i < NR; ++i)
			Lrow[i] = sel(act && i < lane, vr[i]);
	}
	auto fwd = [&](double t) -> double {  // L y = t
		double acc = t, y = 0.0;
#pragma unroll
		for (int i = 0; i < NR; ++i) {
			if (i >= rb)
				continue;
			const double yi = readlane_d(acc * inv, i);
			y = (lane == i) ? yi : y;
			acc = fma(-Lrow[i], yi, acc);
		}
		return y;
	};
	if (nunp == 0) {
		double q = fwd(rp);
		if (eps != 0.0) {
			// row `lane` of U = R_B: U(i, j) = R(i, j) = VR[(k1+j)*LDV + k1+i], j > i
			double Urow[NR];
#pragma unroll
			for (int j = 0; j < NR; ++j)
				Urow[j] = sel(act && j > lane && j < rb, VR[(k1 + (j < rb ? j : 0)) * LDV + k1 + (act ? lane : 0)]);
			double acc = q, z = 0.0;  // U z = q
#pragma unroll
			for (int i = NR - 1; i >= 0; --i) {
				if (i >= rb)
					continue;
				const double zi = readlane_d(acc * inv, i);
				z = (lane == i) ? zi : z;
				acc = fma(-Urow[i], zi, acc);
			}
			q -= eps * fwd(z);  // (B'B)^-1 q0 = L^-1 U^-1 q0 with B = L, B' = U
			// residual of the eps-regularised solution: L (q0 - q) = eps z
			if (res_norm2)
				*res_norm2 = eps * eps * wave_sum(act ? z * z : 0.0);
		} else if (res_norm2) {
			*res_norm2 = 0.0;
		}
		return act ? q : 0.0;
	}

	// ---- over-determined / rank-deficient: corrected semi-normal equations ----
	const ldsd *Xk = l.X + k1;
	// S = R_B R_B' (rb x rb), row `lane` in registers
	double Srow[NR];
#pragma unroll
	for (int j = 0; j < NR; ++j)
		Srow[j] = 0.0;
	for (int p = 0; p < rb; ++p) {  // pivoted column at position p: e_i = R(i,p), i < p ; R(p,p)
		const ldsd *vr = VR + (k1 + p) * LDV + k1;
		const double dp = l.rdiag[k1 + p];
		const double ei = (lane < p) ? vr[lane] : (lane == p ? dp : 0.0);
#pragma unroll
		for (int j = 0; j < NR; ++j) {
			if (j > p)
				continue;
			const double ej = (j < p) ? vr[j] : dp;
			Srow[j] = fma(ei, ej, Srow[j]);
		}
	}
	for (int u = 0; u < nunp; ++u) {
		const ldsd *xr = Xk + u * ld;
		const double ei = act ? xr[lane] : 0.0;
#pragma unroll
		for (int j = 0; j < NR; ++j) {
			if (j >= rb)
				continue;
			Srow[j] = fma(ei, xr[j], Srow[j]);
		}
	}
	// Cholesky S = C C' (C lower), row `lane` of C in registers
	double Crow[NR];
#pragma unroll
	for (int j = 0; j < NR; ++j)
		Crow[j] = 0.0;
	double cinv = 0.0;
#pragma unroll
	for (int k = 0; k < NR; ++k) {
		if (k >= rb)
			continue;
		const double dk = sqrt(readlane_d(Srow[k], k));
		const double rdk = 1.0 / dk;
		const double lik = (act && lane > k) ? Srow[k] * rdk : 0.0;
		Crow[k] = (lane == k) ? dk : lik;
		cinv = (lane == k) ? rdk : cinv;
#pragma unroll
		for (int j = 0; j < NR; ++j) {
			if (j >= rb)
				continue;
			if (j > k)
				Srow[j] = fma(-lik, readlane_d(lik, j), Srow[j]);
		}
	}
	auto chol_solve = [&](double t) -> double {  // (C C')^-1 t, t_j / result in lane j
		double acc = t, y = 0.0;
#pragma unroll
		for (int i = 0; i < NR; ++i) {
			if (i >= rb)
				continue;
			const double yi = readlane_d(acc * cinv, i);
			y = (lane == i) ? yi : y;
			acc = (lane > i) ? fma(-Crow[i], yi, acc) : acc;
		}
		acc = y;  // C' z = y: lane l needs C(i, l), which lane i holds -> transpose through LDS
		double z = 0.0;
#pragma unroll
		for (int i = NR - 1; i >= 0; --i) {
			if (i >= rb)
				continue;
			const double zi = readlane_d(acc * cinv, i);
			z = (lane == i) ? zi : z;
			__syncthreads();
			if (lane == i) {
#pragma unroll
				for (int j = 0; j < NR; ++j)
					if (j < i)
						l.tmp[32 + j] = Crow[j];
			}
			__syncthreads();
			if (lane < i)
				acc = fma(-l.tmp[32 + lane], zi, acc);
		}
		return z;
	};
	auto RB_vec = [&]() -> double {  // (R_B w)_lane with w per column in l.rhs[column lane]
		double acc = 0.0;
		if (act) {
			for (int p = lane; p < rb; ++p) {
				const double e = (p == lane) ? l.rdiag[k1 + p] : VR[(k1 + p) * LDV + k1 + lane];
				acc = fma(e, l.rhs[l.pivlane[k1 + p]], acc);
			}
			for (int u = 0; u < nunp; ++u)
				acc = fma(Xk[u * ld + lane], l.rhs[l.xlane[u]], acc);
		}
		return acc;
	};
	int myu = -1;
	for (int t = 0; t < nunp; ++t)
		myu = (l.xlane[t] == lane) ? t : myu;
	auto residual = [&](double rc) -> double {  // rc - (R_B' q)_column, q by position in l.tmp[0..rb)
		if (!isB)
			return 0.0;
		double acc = rc;
		if (mypos >= k1 && mypos < k2) {
			const int p = mypos - k1;
			const ldsd *vr = VR + mypos * LDV + k1;
			for (int i = 0; i < p; ++i)
				acc = fma(-vr[i], l.tmp[i], acc);
			acc = fma(-l.rdiag[mypos], l.tmp[p], acc);
		} else if (myu >= 0) {
			for (int i = 0; i < rb; ++i)
				acc = fma(-Xk[myu * ld + i], l.tmp[i], acc);
		}
		return acc;
	};
	double q = chol_solve(RB_vec());  // l.rhs holds rhs per column
	__syncthreads();
	if (act)
		l.tmp[lane] = q;
	__syncthreads();
	double res = residual(rhs_c);
	__syncthreads();
	l.rhs[lane] = res;
	__syncthreads();
	q += chol_solve(RB_vec());  // one refinement step
	if (eps != 0.0)
		q -= eps * chol_solve(q);
	__syncthreads();
	if (act)
		l.tmp[lane] = q;
	__syncthreads();
	if (res_norm2) {
		res = residual(rhs_c);
		*res_norm2 = wave_sum(res * res);
	}
	__syncthreads();
	return act ? q : 0.0;
}


// ---------------------------------------------------------------------------
// Task projection.  Lane j < nz owns column j of Z (zc[l] = Z[l][j]); lane 63 owns x.
// For every staged task row a_r:  lane j gets (a_r Z)_j, lane 63 gets a_r x.
//   to_T : (A Z)' -> TV[r*ld + j] (column r of the QR workspace), rhs -> l.rhs[r] = b_r - a_r x
//   else : D Z   -> G[(P+r)*ld + j],                              g[P+r]       = f_r - d_r x
// ---------------------------------------------------------------------------
__device__ inline void project_task(const double *__restrict__ rows_g, const double *__restrict__ rhs_g, int rows,
				    int nstore, int n, int nz, int ld, const double (&zc)[NR], bool to_T, int P,
				    const Lds &l)
{
	const int lane = lane_id();
	// staging map: half a wavefront per task row (lane & 31 = column, lane >> 5 = row parity);
	// the next chunk's global loads are in flight while the current chunk is multiplied
	const int c = lane & 31, h = lane >> 5;
	const int cc = (c < nstore) ? c : 0;
	double pre[CHUNK / 2];
	double rv_next;
	auto fetch = [&](int r0) {
		const int nr = min(CHUNK, rows - r0);
#pragma unroll
		for (int t = 0; t < CHUNK / 2; ++t) {
			const int r = 2 * t + h;
			// clamped, never masked: rows >= nr are not read back, columns >= nstore meet zc == 0
			pre[t] = rows_g[(size_t)(r0 + (r < nr ? r : 0)) * nstore + cc];
		}
		rv_next = rhs_g[r0 + (lane < nr ? lane : 0)];
	};
	fetch(0);
	for (int r0 = 0; r0 < rows; r0 += CHUNK) {
		const int nr = min(CHUNK, rows - r0);
		__syncthreads();  // readers of the previous chunk are done with S
#pragma unroll
		for (int t = 0; t < CHUNK / 2; ++t)
			l.S[(2 * t + h) * SROW + c] = pre[t];
		const double rv = rv_next;
		__syncthreads();
		if (r0 + CHUNK < rows)
			fetch(r0 + CHUNK);
		for (int r = 0; r < nr; ++r) {
			const ldsd2 *sr2 = (const ldsd2 *)(l.S + r * SROW);
			double a0 = 0.0, a1 = 0.0;
			rows_upto2(n, [&](int i) {
				const d2v t = sr2[i >> 1];
				a0 = fma(t.x, zc[i], a0);
				a1 = fma(t.y, zc[i + 1], a1);
			});
			const double acc = a0 + a1;
			const double rr = readlane_d(rv, r);
			if (to_T) {
				if (lane < nz)
					l.TV[(r0 + r) * ld + lane] = acc;
				if (lane == 63)
					l.rhs[r0 + r] = rr - acc;
			} else {
				if (lane < nz)
					l.G[(P + r0 + r) * ld + lane] = acc;
				if (lane == 63)
					l.g[P + r0 + r] = rr - acc;
			}
		}
	}
	__syncthreads();
}

#ifdef HOQP_PROFILE
#define HOQP_STAMP(slot)                                                              \
	do {                                                                          \
		unsigned long long now_ = __builtin_amdgcn_s_memtime();               \
		if (a.dbg && lane_id() == 0)                                          \
			a.dbg[(size_t)blockIdx.x * 8 + (slot)] += now_ - st.t_last;   \
		st.t_last = __builtin_amdgcn_s_memtime();                             \
	} while (0)
#else
#define HOQP_STAMP(slot) \
	do {             \
	} while (0)
#endif

struct LevelState {
	int n, nz;
	int P;	   // live stored inequality rows
	int Pout;  // inequality rows emitted so far (offset of this level's slacks in out.v)
	int status;
	unsigned long long t_last;
};

// ---------------------------------------------------------------------------
// Equality-constrained least-squares step of the active-set method, factored from scratch:
//   min |B (y+p) - ...|  s.t.  G_W p = 0, minimum norm.  Lane c owns column c of
//   [ G_W' | B0' | G_V' ]  (w + m + |V| <= 64 columns).
// Returns p_i in lane i; *k1_out = rank of the working set (reflectors [0,k1) stay in TV).
// ---------------------------------------------------------------------------
__device__ inline double eqp_step(const SolveArgs &a, const Lds &l, int nz, int rank, int m, int w,
				  unsigned long long vmask, int *k1_out, bool *overflow)
{
	const int lane = lane_id();
	const int ld = a.ld;
	const int nV = __popcll(vmask);
	const int nb = m + nV;
	const int ncol = w + nb;
	*k1_out = 0;
	if (ncol > kWave) {
		*overflow = true;
		return 0.0;
	}
	double col[NR];
	double rhs_c = 0.0;
	const bool has = lane < ncol;
	const bool isC = lane < w;
	{
		const ldsd *src = l.G;
		int lim = nz;
		if (has) {
			if (isC) {
				src = l.G + l.W[lane] * ld;
			} else if (lane < w + m) {
				int j = lane - w;
				src = l.BL + j * ld;
				lim = rank;
				rhs_c = -l.rho[j];
			} else {
				int t = lane - w - m;
				unsigned long long mk = vmask;
				for (int q = 0; q < t; ++q)
					mk &= mk - 1;
				int r = __ffsll((long long)mk) - 1;
				src = l.G + r * ld;
				rhs_c = -l.s[r];
			}
		} else {
			lim = 0;
		}
#pragma unroll
		for (int i = 0; i < NR; ++i)
			col[i] = sel(i < lim, src[i]);
	}
	__syncthreads();
	int mypos = has ? -1 : -2;
	const int k1 = uni(hh_factor_reg(col, mypos, isC, nz, 0, 1e-12, l));
	const int k2 = uni(hh_factor_reg(col, mypos, has && !isC, nz, k1, 1e-11, l));
	*k1_out = k1;
	// unpivoted B columns -> X
	const bool unp = has && !isC && mypos == -1;
	const unsigned long long um = __ballot(unp);
	const int nunp = __popcll(um);
	if (nunp > XROWS) {
		*overflow = true;
		return 0.0;
	}
	if (unp) {
		int u = __popcll(um & ((1ull << lane) - 1ull));
		ldsd *xr = l.X + u * ld;
#pragma unroll
		for (int i = 0; i < NR; ++i)
			if (i < nz)
				xr[i] = col[i];
		l.xlane[u] = lane;
	}
	__syncthreads();
	double q = ls_solve_reg(k1, k2, rhs_c, has && !isC, mypos, nunp, 0.0, ld, nullptr, l);
	// u = [0 (k1) ; q ; 0],  p = Q u
	int srcl = lane - k1;
	double qq = __shfl(q, srcl < 0 ? 0 : srcl);
	double u = (srcl >= 0 && srcl < k2 - k1) ? qq : 0.0;
	return hh_apply_vec(u, nz, 0, k2, false, l);
}

// ---------------------------------------------------------------------------
// one priority level (HoQpProblem constructor pipeline, ho_qp_problem.cpp:11-25)
// ---------------------------------------------------------------------------
__device__ inline void solve_level(const SolveArgs &a, const Lds &l, LevelState &st, int inst, int k, int m, int p)
{
	const int lane = lane_id();
	const int n = uni(st.n), ld = a.ld;
	const int nz = uni(st.nz), P = uni(st.P);
	const int nstore = a.n;
	m = uni(m);
	p = uni(p);
	const int Pn = P + p;
	const bool consider = (nz > 0) && (a.gate_prior ? (p > 0) : (p > 0 || P > 0));
	// stored rows are still needed by a later level?  (else they are neither rotated nor kept)
	const bool keep_rows = k < a.last_ineq_use;
	const int Pg = (consider || keep_rows) ? Pn : 0;  // rows that take part in rotation / update
	int iters = 0;
	double level_res2 = 0.0;

	// ---- 1. project the task into the current null space (ConstructH/C/D/FVector,
	//         ho_qp_problem.cpp:180-285, in factored form) ----
	{
		double zc[NR];
		const ldsd *src = (lane < nz) ? l.Z + lane : l.x;
		const int stride = (lane < nz) ? ld : 1;
		const bool live = (lane < nz) || (lane == 63);
#pragma unroll
		for (int i = 0; i < NR; ++i)
			zc[i] = sel(live && i < n, src[(i < n ? i : 0) * stride]);
		if (m > 0)
			project_task(a.A[k] + (size_t)inst * a.m[k] * nstore, a.b[k] + (size_t)inst * a.m[k], m, nstore, n,
				     nz, ld, zc, true, 0, l);
		if (p > 0)
			project_task(a.D[k] + (size_t)inst * a.p[k] * nstore, a.f[k] + (size_t)inst * a.p[k], p, nstore, n,
				     nz, ld, zc, false, P, l);
	}

	HOQP_STAMP(1);
	// ---- 2. complete orthogonal decomposition of A Z (CalcNullSpaceProjMatrix, math.hpp:99-104,
	//         with an orthonormal Householder basis instead of the FullPivLU kernel) ----
	int rank = 0;
	double yi = 0.0;  // lane i holds y_i (rotated coordinates)
	if (m > 0 && nz > 0) {
		double col[NR];
		const bool has = lane < m;
		{
			const ldsd *src = l.TV + (has ? lane : 0) * ld;
#pragma unroll
			for (int i = 0; i < NR; ++i)
				col[i] = sel(has && i < nz, src[i]);
		}
		const double rhs_c = has ? l.rhs[lane] : 0.0;
		__syncthreads();  // every column is in registers before TV is reused as the reflector store
		int mypos = has ? -1 : -2;
		rank = uni(hh_factor_reg(col, mypos, has, nz, 0, kEpsMach * (double)max(m, nz), l));
		HOQP_STAMP(2);
		const bool unp = has && mypos == -1;
		const unsigned long long um = __ballot(unp);
		const int nunp = __popcll(um);
		const int myu = __popcll(um & ((1ull << lane) - 1ull));
		if (consider) {
			// LS rows in rotated coordinates for the active-set loop: BL[j][0..rank), r0[j];
			// row order: pivot positions first, then the unpivoted (dependent) rows
			if (has) {
				int j = unp ? rank + myu : mypos;
				ldsd *row = l.BL + j * ld;
				if (unp) {
#pragma unroll
					for (int i = 0; i < NR; ++i)
						if (i < rank)
							row[i] = col[i];
				} else {
					const ldsd *vr = l.TV + mypos * LDV;
#pragma unroll
					for (int i = 0; i < NR; ++i)
						if (i < rank)
							row[i] = (i < mypos) ? vr[i] : (i == mypos ? l.rdiag[mypos] : 0.0);
				}
				l.r0[j] = rhs_c;
			}
		} else if (rank > 0) {
			// ---- 4a. unconstrained level: y1 = argmin |R' y1 - r| (+ eps term), y2 = 0 ----
			if (nunp > XROWS) {
				st.status |= HOQP_STATUS_WORKSPACE;
			} else {
				if (unp) {
					ldsd *xr = l.X + myu * ld;
#pragma unroll
					for (int i = 0; i < NR; ++i)
						if (i < nz)
							xr[i] = col[i];
					l.xlane[myu] = lane;
				}
				__syncthreads();
				yi = ls_solve_reg(0, rank, rhs_c, has, mypos, nunp, a.eps, ld, &level_res2, l);
			}
		} else {
			level_res2 = wave_sum(has ? rhs_c * rhs_c : 0.0);
		}
		__syncthreads();
		HOQP_STAMP(3);
		// ---- 3. rotate Z and every stored inequality row: Z <- Z Q, G <- G Q ----
		if (rank > 0)
			hh_apply_right_rows_reg(l.Z, n, l.G, Pg, ld, nz, 0, rank, l);
		HOQP_STAMP(4);
	} else if (m > 0) {
		// no free directions left: the residual is what it is
		double r = (lane < m) ? l.rhs[lane] : 0.0;
		level_res2 = wave_sum(r * r);
	}

	if (consider) {
		// ---- 4b. primal active set on the piecewise quadratic ----
		if (lane < Pn)
			l.s[lane] = -l.g[lane];
		if (lane < m)
			l.rho[lane] = -l.r0[lane];
		__syncthreads();
		const unsigned long long softmask = (p >= 64) ? ~0ull : (((1ull << p) - 1ull) << P);
		const unsigned long long hardmask = (P >= 64) ? ~0ull : ((1ull << P) - 1ull);
		unsigned long long wmask = 0ull;
		int w = 0;
		{
			int bad = 0;
			if (lane < P)
				bad = l.s[lane] > 1e-9 * (1.0 + fabs(l.g[lane]));
			if (__ballot(bad))
				st.status |= HOQP_STATUS_INFEASIBLE_START;
		}
		const double gl = (lane < Pn) ? l.g[lane] : 0.0;
		const bool isSoft = lane < Pn && ((softmask >> lane) & 1ull);
		const double tolv = 1e-11 * (1.0 + fabs(gl));
		bool done = false;
		for (iters = 0; iters < a.max_iter && !done; ++iters) {
			const double si = (lane < Pn) ? l.s[lane] : 0.0;
			const double rj = (lane < m) ? l.rho[lane] : 0.0;
			const unsigned long long vmask = __ballot(isSoft && si > tolv);
			int k1 = 0;
			bool overflow = false;
			const double pi = eqp_step(a, l, nz, rank, m, w, vmask, &k1, &overflow);
			if (overflow) {
				st.status |= HOQP_STATUS_WORKSPACE;
				break;
			}
			const double pn = wave_max(fabs(pi));
			const double yn = wave_max(fabs(yi));
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
							gi = fma(l.BL[j * ld + lane], l.rho[j], gi);
					unsigned long long mk = vmask;
					while (mk) {
						int r = __ffsll((long long)mk) - 1;
						mk &= mk - 1;
						gi = fma(l.G[r * ld + lane], l.s[r], gi);
					}
				}
				const double gn = wave_max(fabs(gi));
				const double ti = hh_apply_vec(-gi, nz, 0, k1, true, l);
				// back substitution R_C mu = t (k1 x k1, pivot positions)
				double acc = ti, mu = 0.0;
				for (int i = k1 - 1; i >= 0; --i) {
					double mi = readlane_d(acc, i) / l.rdiag[i];
					if (lane == i)
						mu = mi;
					if (lane < i)
						acc = fma(-l.TV[i * LDV + lane], mi, acc);
				}
				int wi = (lane < k1) ? l.pivlane[lane] : -1;  // pivot lane == position in W
				int drop;
				double worst = wave_argmin((lane < k1) ? mu : 1e300, wi, &drop);
				if (!(worst < -1e-9 * (1.0 + gn)) || drop < 0) {
					done = true;
					break;
				}
				wmask &= ~(1ull << uni(l.W[drop]));
				__syncthreads();
				if (lane == 0)
					for (int j = drop; j < w - 1; ++j)
						l.W[j] = l.W[j + 1];
				--w;
				__syncthreads();
				continue;
			}
			// c_i = G_i p for every row, a0_j = B0_j p   (p broadcast through LDS)
			__syncthreads();
			l.tmp[lane] = (lane < nz) ? pi : 0.0;
			__syncthreads();
			double c0 = 0.0, c1 = 0.0, b0 = 0.0, b1 = 0.0;
			{
				// this lane's inequality row and LS row (reloaded per iteration: keeping them
				// live across eqp_step would spill)
				double grow[NR], brow[NR];
				const ldsd *gp = l.G + (lane < Pn ? lane : 0) * ld;
				const ldsd *bp = l.BL + (lane < m ? lane : 0) * ld;
#pragma unroll
				for (int i = 0; i < NR; ++i) {
					grow[i] = sel(lane < Pn && i < nz, gp[i]);
					brow[i] = sel(lane < m && i < rank, bp[i]);
				}
				const ldsd2 *p2 = (const ldsd2 *)l.tmp;
#pragma unroll
				for (int i = 0; i < NR; i += 2) {
					if (i >= nz)
						continue;
					d2v t = p2[i >> 1];
					c0 = fma(grow[i], t.x, c0);
					c1 = fma(grow[i + 1], t.y, c1);
					b0 = fma(brow[i], t.x, b0);
					b1 = fma(brow[i + 1], t.y, b1);
				}
			}
			const double ci = c0 + c1, a0 = b0 + b1;
			if (lane < Pn)
				l.c[lane] = ci;
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
			double tb = 1e300;
			if (isSoft && ci != 0.0 && ((si > tolv && ci < 0.0) || (si <= tolv && ci > 0.0))) {
				double t = -si / ci;
				if (t > 0.0)
					tb = t;
			}
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
			const double t_hi = wave_min((tb < 1e299 && dmine >= 0.0) ? tb : 1e300);
			const double t_lo = wave_max((tb < t_hi) ? tb : 0.0);
			const double tm = (t_hi < 1e299) ? 0.5 * (t_lo + t_hi) : t_lo + 1.0;
			double slope, d_lo;
			{
				double sv = si + tm * ci;
				slope = q1 + wave_sum((isSoft && sv > 0.0) ? ci * ci : 0.0);
				sv = si + t_lo * ci;
				d_lo = q0 + q1 * t_lo + wave_sum((isSoft && sv > 0.0) ? ci * sv : 0.0);
			}
			double t_star;
			if (d_lo >= 0.0)
				t_star = t_lo;
			else if (slope > 0.0)
				t_star = fmin(t_lo - d_lo / slope, t_hi);
			else
				t_star = (t_hi < 1e299) ? t_hi : t_lo;
			const double t = fmin(t_star, t_hard);
			yi = fma(t, pi, yi);
			__syncthreads();
			if (lane < Pn)
				l.s[lane] = fma(t, ci, si);
			if (lane < m)
				l.rho[lane] = fma(t, a0, rj);
			if (t_hard <= t_star && blk >= 0) {
				if (w < NR) {
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
		{
			double rj = (lane < m) ? l.rho[lane] : 0.0;
			level_res2 = wave_sum(rj * rj);
		}
	}

	HOQP_STAMP(5);
	// ---- 5./6. x_k = x_{k-1} + (Z Q) y (GetSolution, ho_qp_problem.cpp:46-51); slack bookkeeping
	//            (ConstructFVector, :207-229); Z_k = Z_{k-1} Q2, G <- G Q2 (:172-178) ----
	__syncthreads();
	l.tmp[lane] = (lane < nz) ? yi : 0.0;
	__syncthreads();
	for (int r = lane; r < n + Pg; r += kWave) {
		const bool isZ = r < n;
		ldsd *rowp = isZ ? l.Z + r * ld : l.G + (r - n) * ld;
		double row[NR];
#pragma unroll
		for (int i = 0; i < NR; ++i)
			row[i] = sel(i < nz, rowp[i]);
		double d0 = 0.0, d1 = 0.0;
		const ldsd2 *y2 = (const ldsd2 *)l.tmp;
#pragma unroll
		for (int i = 0; i < NR; i += 2) {
			if (i >= nz)
				continue;
			d2v t = y2[i >> 1];
			d0 = fma(row[i], t.x, d0);
			d1 = fma(row[i + 1], t.y, d1);
		}
		const double d = d0 + d1;
		if (isZ) {
			l.x[r] += d;
		} else {
			const int gr = r - n;
			double gi = l.g[gr] - d;
			if (gr >= P) {
				// v* = (D x - f)_+ ; the row becomes hard with rhs f + v* - D x
				if (a.out.v)
					a.out.v[(size_t)inst * a.Ptot + st.Pout + (gr - P)] = fmax(0.0, -gi);
				gi = fmax(gi, 0.0);
			}
			l.g[gr] = gi;
		}
		if (rank > 0) {
#pragma unroll
			for (int i = 0; i < NR; ++i)
				if (i >= rank && i < nz)
					rowp[i - rank] = row[i];
		}
	}
	if (lane == 0) {
		if (a.out.level_res)
			a.out.level_res[(size_t)inst * a.L + k] = sqrt(level_res2);
		if (a.out.iters)
			a.out.iters[(size_t)inst * a.L + k] = iters;
		if (a.out.nz_levels)
			a.out.nz_levels[(size_t)inst * a.L + k] = nz - rank;
	}
	if (Pg == 0 && p > 0 && lane < p && a.out.v)  // no free direction left: v* = (D x - f)_+ as is
		a.out.v[(size_t)inst * a.Ptot + st.Pout + lane] = fmax(0.0, -l.g[P + lane]);
	st.nz = nz - rank;
	st.P = keep_rows ? Pn : 0;
	st.Pout += p;
	__syncthreads();
	HOQP_STAMP(6);

	if (a.out.x_levels && lane < a.n)
		a.out.x_levels[((size_t)inst * a.L + k) * a.n + lane] = (lane < n) ? l.x[lane] : 0.0;
	if (a.out.Z_levels) {
		double *Zo = a.out.Z_levels + ((size_t)inst * a.L + k) * a.n * a.n;
		for (int idx = lane; idx < a.n * a.n; idx += kWave) {
			int r = idx / a.n, c = idx - r * a.n;
			Zo[idx] = (r < n && c < st.nz) ? l.Z[r * ld + c] : 0.0;
		}
	}
}

#endif  // __HIP_DEVICE_COMPILE__

__global__ void __launch_bounds__(kWave) hoqp_solve_kernel(const SolveArgs a)
{
#ifdef __HIP_DEVICE_COMPILE__
	extern __shared__ __align__(16) char smem[];
	const Lds l = lds_bind(a, (HOQP_LDS char *)smem);
	const int lane = lane_id();
	const int inst = blockIdx.x;
	if (inst >= a.batch)
		return;
	LevelState st;
	st.n = uni(a.n_inst ? a.n_inst[inst] : a.n);
	st.nz = st.n;
	st.P = 0;
	st.Pout = 0;
	st.status = 0;
	st.t_last = __builtin_amdgcn_s_memtime();
	const int n = st.n, ld = a.ld;
	// InitPrevProblemValuesToDefault, ho_qp_problem.cpp:119-128:  Z = I, x = 0
	for (int idx = lane; idx < n * ld; idx += kWave) {
		int r = idx / ld, c = idx - r * ld;
		l.Z[idx] = (r == c) ? 1.0 : 0.0;
	}
	if (lane < NR)
		l.x[lane] = 0.0;
	if (a.out.v)
		for (int i = lane; i < a.Ptot; i += kWave)
			a.out.v[(size_t)inst * a.Ptot + i] = 0.0;
	__syncthreads();

	HOQP_STAMP(0);
	for (int k = 0; k < a.L; ++k) {
		int m = uni(a.m_inst ? a.m_inst[(size_t)inst * a.L + k] : a.m[k]);
		int p = uni(a.p_inst ? a.p_inst[(size_t)inst * a.L + k] : a.p[k]);
		solve_level(a, l, st, inst, k, m, p);
	}

	// ---- final x and the CheckSolutionValid flags (helper_functions.hpp:63-91) ----
	double xi = (lane < n) ? l.x[lane] : 0.0;
	if (__ballot(!isfinite(xi)))
		st.status |= HOQP_STATUS_NONFINITE;
	if (a.out.x && lane < a.n)
		a.out.x[(size_t)inst * a.n + lane] = xi;
	if (a.out.flags || a.out.eq_res_max || a.out.ineq_viol_max || a.out.active_mask) {
		// one task row per lane; x broadcast from LDS
		double xr[NR];
#pragma unroll
		for (int i = 0; i < NR; ++i)
			xr[i] = sel(i < n, l.x[i]);
		unsigned long long amask = 0ull;
		int prow = 0;
		for (int k = 0; k < a.L; ++k) {
			const int m = uni(a.m_inst ? a.m_inst[(size_t)inst * a.L + k] : a.m[k]);
			const int p = uni(a.p_inst ? a.p_inst[(size_t)inst * a.L + k] : a.p[k]);
			double er = 0.0, iv = -1e300;
			if (m > 0) {
				double e = 0.0;
				if (lane < m) {
					const double *ar = a.A[k] + ((size_t)inst * a.m[k] + lane) * a.n;
					double acc = -a.b[k][(size_t)inst * a.m[k] + lane];
					double t[NR];
#pragma unroll
					for (int i = 0; i < NR; ++i)
						t[i] = ar[i < n ? i : 0];  // all loads in flight at once
#pragma unroll
					for (int i = 0; i < NR; ++i)
						acc = fma(t[i], xr[i], acc);  // xr[i >= n] == 0
					e = fabs(acc);
				}
				er = wave_max(e);
			}
			if (p > 0) {
				double e = -1e300;
				if (lane < p) {
					const double *dr = a.D[k] + ((size_t)inst * a.p[k] + lane) * a.n;
					double acc = -a.f[k][(size_t)inst * a.p[k] + lane];
					double t[NR];
#pragma unroll
					for (int i = 0; i < NR; ++i)
						t[i] = dr[i < n ? i : 0];
#pragma unroll
					for (int i = 0; i < NR; ++i)
						acc = fma(t[i], xr[i], acc);
					e = acc;  // D x - f
				}
				iv = wave_max(e);
				unsigned long long act = __ballot(lane < p && e >= -1e-7);
				if (prow < 64)
					amask |= act << prow;
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
	HOQP_STAMP(7);
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace hoqp
