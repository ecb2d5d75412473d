// hoqp_solve.hpp -- the per-instance HO-QP solve (device code, gfx950).
//
// One workgroup == one wavefront == one instance.  Master copies of the small matrices live
// in LDS.  The O(n^2)/O(n^3) loops run on register tiles with a 2-D lane map: a lane is
// (s, c) = (lane & 3, lane >> 2); a "panel" holds up to 64 vectors of length <= 32, vector
// q = c + 16*slot owned by the four lanes of quad c, lane s holding entries s, s+4, s+8, ...
// (NT = 8 per vector).  Dot products are 8 FMAs + a two-step DPP quad reduction, and the
// entry range [k, nr) of Householder step k shrinks uniformly over the four lanes.
//
// Algorithm: DESIGN.md section 3.  Reference: src/control/ho_qp/ho_qp_problem.cpp:11-357.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include "../../include/hoqp.h"
#include "hoqp_device.hpp"
#include "wbc_tasks.hpp"

namespace hoqp {

constexpr int NR = 32;	  // vector length capacity, >= HOQP_MAX_VARS
constexpr int NT = NR / 4; // entries of one vector held by one lane
constexpr int NSP = 4;	  // panel slots: 64 columns  (working set + LS rows + violated rows)
constexpr int NSR = 6;	  // row slots: 96 rows >= HOQP_MAX_VARS + HOQP_MAX_INEQ_ROWS
constexpr int LDV = 34;	  // leading dimension of the reflector store
constexpr int SROW = 36;  // row stride of the task staging buffer of the n_z == 32 projection (16-byte
			  // aligned rows); the matrix-core projection loads its operand from global memory
constexpr int XROWS = 16; // dumped unpivoted columns (rank-deficient least squares); aliases S
// Compact LDS layout of the fused WBC kernel under reference semantics (SolveArgs::compact; 19,824 B per
// workgroup: eight workgroups per CU -- two waves on every SIMD -- instead of four).  It rests on these facts of
// that hierarchy: level 0 never stores Z = I (its rotation generates the rows) and leaves n - 6 <= 24 free
// coordinates, so from level 1 on every vector handled has at most 24 entries -- Z, G, the dependent-column
// store and the projected task rows take a leading dimension of 25, the reflector store 25 (level 0, whose six
// reflectors have 30 entries: 34); of the 40 inequality rows of level 1 only the 12 torque rows are stored, the
// negated copies and the friction rows (two non-zeros each) are formed from G and Z on the fly (vrow); the
// level-1 QP has no least-squares rows and no working set (Q2), so BL / r0 / rho / a0 are stubs; at most 8
// dependent columns are kept for a rank-deficient least squares (more: HOQP_STATUS_WORKSPACE); and the
// reference replay updates its basis chain in place inside G + TV + S.
// A state whose floating-base block [M_fb, -J_c,fb^T] has rank < 6 (impossible for a physical mass matrix) does
// not fit and is reported as HOQP_STATUS_BASE_RANK instead of being solved.
// The fused WBC kernel under paper semantics (SolveArgs::compact == 2, kPaperDims below, 25,472 B: six workgroups
// per CU) uses the same strides and virtual rows; its stored rows live on as hard rows of levels 2..4, so it adds
// the least-squares rows of a level inside the active-set loop and 16 dependent columns (solve_level: CP).
constexpr int kCLd = 25;	 // Z, G, X, projected rows of levels >= 1
constexpr int kCLd0 = 31;	 // projected rows of level 0 (6 x n)
constexpr int kCLdv = 25;	 // reflector store, levels >= 1
constexpr int kCXrows = 8;	 // dependent-column store
constexpr int kCTv = 24 * kCLdv + 8;	 // >= 6 * LDV, >= 18 * kCLd: reflector store / projected rows
constexpr int kCGrows = 12;	 // stored rows of G (= kNJ torque rows)
constexpr int kCReplayZ = 720;	 // replay: doubles for the basis (30 x 24, updated in place), rest: M and its factors
constexpr int CHUNK = 16; // task rows per projection pass (one 16 x K matrix-core operand tile)
constexpr int kDbgSlots = 56;	  // HOQP_PROFILE builds: cycle-stamp slots per instance
// the reference replay runs when eps / (smallest pivot)^2 of an unconstrained level exceeds this
constexpr double kReplayTau = 3e-8;

static_assert(NR >= HOQP_MAX_VARS, "register tiles must cover n");
static_assert(16 * NSR >= HOQP_MAX_VARS + HOQP_MAX_INEQ_ROWS, "row slots must cover Z and G");

struct SolveArgs {
	int batch, n, L;
	int m[HOQP_MAX_LEVELS], p[HOQP_MAX_LEVELS];
	const double *A[HOQP_MAX_LEVELS], *b[HOQP_MAX_LEVELS], *D[HOQP_MAX_LEVELS], *f[HOQP_MAX_LEVELS];
	const int *n_inst, *m_inst, *p_inst;
	int gate_prior, slack_quirk, max_iter;
	int skip_unread;  // HOQP_OPT_SKIP_UNREAD_SLACK_LEVELS
	double eps;
	hoqp_result_batch out;
	int Ptot;  // storage extent of stacked inequality rows (<= 64)
	int mmax;  // max_k m[k] (<= 32)
	int ld;	   // odd leading dimension >= n of Z, G, BL, X
	int use_bl;	   // some level needs its equality rows inside the active-set loop
	int compact;	   // compact LDS layout (fused WBC entry under reference semantics only)
	int any_hard;	   // some level looks at rows stored by an earlier level
	int last_ineq_use; // last level that looks at stored inequality rows (rows are dropped after it)
	// fused WBC entry (hoqp_solve_kernel<.., .., true>): the task blocks are generated from the
	// dynamics terms (wbc_tasks.hpp) instead of being read from A / b / D / f
	hoqp_wbc_inputs win;
	WbcPrm wprm;
	double *qdd, *tau;  // [batch][12] outputs of the fused tail (may be null)
	unsigned long long *dbg;  // HOQP_PROFILE builds only: [batch][16] cycle stamps per section
};

// Leading dimension (in doubles) of the row-major LDS matrices Z, G, BL, X and of the projected task rows.
inline int lds_leading_dim(int n)
{
#ifdef HOQP_LD36
	(void)n;
	return 36;
#else
	return n | 1;
#endif
}

// byte offsets of the LDS arrays (shared by host sizing and device addressing)
struct LdsOff {
	int Z, x, G, g, vs, s, c, TV, S, BL, r0, rho, a0, tau, rdiag, rhs, X, tmp, dump, b3, pivlane, W, xlane, colpos, vlist, total;
};

// the extents the layout depends on
struct LdsDims {
	int n, ld, Ptot, mmax, use_bl, compact;
};
// the fused WBC entry under reference semantics (compact layout): n = 30 storage columns, 40 inequality rows.
// In the kernel these are compile-time constants, so every LDS array sits at a literal offset.
constexpr LdsDims kCompactDims = {30, 0, 40, 0, 0, 1};
// the fused WBC entry under paper semantics: the same strides and virtual rows, plus what hard rows need -- the
// least-squares rows of a level inside the active-set loop (kPBlRows) and 16 dependent columns
// (The Householder routines read whole rows of the reflector store at their compiled vector length: behind level 0
// the paper kernel uses the 24-entry forms -- panel_factor<2 / 3>, rows_apply_reflectors_t<.., 6> -- so that the
// store can keep the stride 25 of the reference kernel; with the 32-entry forms a row's tail is the next row's
// head, which cost an afternoon.)
constexpr int kPBlRows = 18, kPXrows = 16, kPLdv = kCLdv, kPTv = kCTv;
constexpr LdsDims kPaperDims = {30, 0, 40, kPBlRows, 1, 2};

__host__ __device__ inline LdsOff lds_offsets(const LdsDims &a)
{
	LdsOff o;
	int off = 0;
	auto take = [&](int bytes) {
		int p = off;
		off += (bytes + 15) & ~15;
		return p;
	};
	const int n = a.n, ld = a.compact ? kCLd : a.ld, P = a.Ptot, mm = a.mmax;
	const int tv = (LDV * n > ld * mm) ? LDV * n : ld * mm;
	o.Z = take(8 * n * ld);
	o.x = take(8 * NR);
	if (a.compact) {
		// G, TV and S are contiguous: the reference replay uses them as one scratch region, and reads
		// of a masked-off tail of a TV row (tall levels: 32 rows x ld) stay inside the allocation
		o.G = take(8 * kCGrows * ld);
		o.TV = take(8 * (a.compact == 2 ? kPTv : kCTv));
		o.S = take(8 * (a.compact == 2 ? kPXrows : kCXrows) * ld);
	}
	o.g = take(8 * (P + 1));
	o.vs = take(8 * (P + 1));  // optimal slack v* of every stored inequality row (Q3)
	o.s = take(8 * (P + 1));
	o.c = take(8 * (P + 1));
	if (!a.compact) {
		o.G = take(8 * (P ? P : 1) * ld);
		o.TV = take(8 * tv);
		o.S = take(8 * ((CHUNK * SROW > XROWS * ld) ? CHUNK * SROW : XROWS * ld));
	}
	o.BL = take(a.compact == 1 ? 16 : 8 * ((mm && a.use_bl) ? mm : 1) * ld);
	o.r0 = take(a.compact == 1 ? 16 : 8 * (mm + 1));
	o.rho = take(a.compact == 1 ? 16 : 8 * (mm + 1));
	o.a0 = take(a.compact == 1 ? 16 : 8 * (mm + 1));
	o.tau = take(8 * NR);
	o.rdiag = take(8 * NR);
	o.rhs = take(8 * kWave);
	o.X = o.S;  // the staging buffer is dead whenever X is live
	o.tmp = take(8 * kWave);
	o.dump = take(8 * kWave);  // per-lane sink of predicated-off stores
	o.b3 = take(8 * 24);	   // fused WBC entry: right-hand side of the tracking level
	o.pivlane = take(4 * NR);
	o.W = take(4 * NR);
	o.xlane = take(4 * (a.compact == 1 ? kCXrows : XROWS));
	o.colpos = take(4 * kWave);
	o.vlist = take(4 * kWave);
	o.total = off + 256;  // readable tail: full-width (NR) vector reads may run past the last array
	return o;
}

__host__ __device__ inline LdsOff lds_offsets(const SolveArgs &a)
{
	if (a.compact)
		return lds_offsets(a.compact == 2 ? kPaperDims : kCompactDims);
	const LdsDims d = {a.n, a.ld, a.Ptot, a.mmax, a.use_bl, 0};
	return lds_offsets(d);
}

#ifdef __HIP_DEVICE_COMPILE__
struct Lds {
	ldsd *Z, *x, *G, *g, *vs, *s, *c, *TV, *S, *BL, *r0, *rho, *a0, *tau, *rdiag, *rhs, *X, *tmp, *dump, *b3;
	ldsi *pivlane, *W, *xlane, *colpos, *vlist;
	int ldv;    // leading dimension of the reflector store (LDV; the compact layout: per level)
	int xrows;  // capacity of the dependent-column store X
	// scratch of the reference replay: the chain of bases, and the projected block with its kernel matrix
	ldsd *rpZ, *rpM;
	int rpZcap, rpMcap;
	unsigned long long *dbgp;  // HOQP_PROFILE builds: this instance's cycle-stamp slots (else null)
};

template <int CK = 0>  // CK: 1 / 2 = the kernels of the compact layouts (offsets are literals), 0 = from the arguments
__device__ __forceinline__ Lds lds_bind(const SolveArgs &a, HOQP_LDS char *base)
{
	const LdsOff o = CK == 1 ? lds_offsets(kCompactDims) : (CK == 2 ? lds_offsets(kPaperDims) : lds_offsets(a));
	Lds l;
	l.Z = (ldsd *)(base + o.Z);
	l.x = (ldsd *)(base + o.x);
	l.G = (ldsd *)(base + o.G);
	l.g = (ldsd *)(base + o.g);
	l.vs = (ldsd *)(base + o.vs);
	l.s = (ldsd *)(base + o.s);
	l.c = (ldsd *)(base + o.c);
	l.TV = (ldsd *)(base + o.TV);
	l.S = (ldsd *)(base + o.S);
	l.BL = (ldsd *)(base + o.BL);
	l.r0 = (ldsd *)(base + o.r0);
	l.rho = (ldsd *)(base + o.rho);
	l.a0 = (ldsd *)(base + o.a0);
	l.tau = (ldsd *)(base + o.tau);
	l.rdiag = (ldsd *)(base + o.rdiag);
	l.rhs = (ldsd *)(base + o.rhs);
	l.X = (ldsd *)(base + o.X);
	l.tmp = (ldsd *)(base + o.tmp);
	l.dump = (ldsd *)(base + o.dump);
	l.b3 = (ldsd *)(base + o.b3);
	l.pivlane = (ldsi *)(base + o.pivlane);
	l.W = (ldsi *)(base + o.W);
	l.xlane = (ldsi *)(base + o.xlane);
	l.colpos = (ldsi *)(base + o.colpos);
	l.vlist = (ldsi *)(base + o.vlist);
	l.ldv = LDV;
	const int cmode = CK ? CK : a.compact;
	l.xrows = cmode == 1 ? kCXrows : (cmode == 2 ? kPXrows : XROWS);
	if (cmode) {
		l.rpZ = l.G;
		l.rpZcap = kCReplayZ;
		l.rpM = l.G + kCReplayZ;
		l.rpMcap = kCGrows * kCLd + (cmode == 2 ? kPTv : kCTv) + l.xrows * kCLd - kCReplayZ;
	} else {
		l.rpZ = l.G;
		l.rpZcap = (a.Ptot ? a.Ptot : 1) * a.ld;
		l.rpM = l.TV;
		l.rpMcap = (LDV * a.n > a.ld * a.mmax) ? LDV * a.n : a.ld * a.mmax;
	}
#ifdef HOQP_PROFILE
	l.dbgp = a.dbg ? a.dbg + (size_t)blockIdx.x * kDbgSlots : nullptr;
#else
	l.dbgp = nullptr;
#endif
	return l;
}

// Loads are issued unconditionally and masked afterwards: a load behind a condition becomes a
// branch + load + wait per element (measured: the guarded form serialised every LDS read).
// Callers keep the ADDRESS valid (clamped index); LDS has a readable tail pad for vectors
// that are read to their full NR extent.
__device__ __forceinline__ double sel(bool on, double v) { return on ? v : 0.0; }

// Predicated LDS store without touching EXEC: a store behind `if (on)` costs an exec save /
// restore pair around every ds_write (measured ~60 cycles each); instead the address of a
// masked-off lane is redirected to that lane's private sink slot.
__device__ __forceinline__ void st_if(bool on, ldsd *p, ldsd *sink, double v)
{
	ldsd *q = on ? p : sink;
	*q = v;
}

// Pins a loaded value in a register.  Without it the compiler turns `cond ? load : 0` back
// into a branch around the load (select-with-load -> branch in CodeGenPrepare), which
// serialises the loads: exec-mask juggling plus one s_waitcnt per element.  Always load a
// whole group first, then pin the group (the waits then come in issue order).
__device__ __forceinline__ void pin(double &v) { asm volatile("" : "+v"(v)); }

// u[t] = p[4 t], t < NT: one lane's entries of a vector stored with unit stride
// (p already includes the lane's entry offset s)
__device__ __forceinline__ void ld8(double (&u)[NT], const ldsd *p)
{
#pragma unroll
	for (int t = 0; t < NT; ++t)
		u[t] = p[4 * t];
}

__device__ __forceinline__ void pin8(double (&u)[NT])
{
#pragma unroll
	for (int t = 0; t < NT; ++t)
		pin(u[t]);
}

// single group: loads then pins.  Several groups: ld8 every group first, then pin8 every group
// (a pin waits for its load, so pinning group 0 before loading group 1 costs a round trip).
__device__ __forceinline__ void load8(double (&u)[NT], const ldsd *p)
{
	ld8(u, p);
	pin8(u);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// sum over the four lanes of a quad (quad_perm [1,0,3,2] then [2,3,0,1]); all four get it
__device__ __forceinline__ double quad_sum(double v)
{
	v += dpp_d<0xB1>(v);
	v += dpp_d<0x4E>(v);
	return v;
}

// ---------------------------------------------------------------------------
// Static-index entry loops.  Lane s holds entries i = s + 4t (t = 0..NT-1) of a vector in
// registers x[t]; register arrays may only be indexed by compile-time constants, the active
// entry range [k, nr) is run-time.  The loops are STRAIGHT-LINE over all NT entries with a
// per-lane mask `on`: a version that skipped inactive blocks with wave-uniform branches was
// measured 3x slower (a taken scalar branch costs far more than the 8 masked FMAs it saves).
// Entries >= nr of every register vector are kept at zero.
// ---------------------------------------------------------------------------
template <typename Body>
__device__ __forceinline__ void t_from(int k, int nr, int s, Body body)
{
	(void)nr;
#pragma unroll
	for (int t = 0; t < NT; ++t)
		body(t, s + 4 * t >= k, true);
}

// entries [0, nr): `on` == false for entries >= nr
template <typename Body>
__device__ __forceinline__ void t_upto(int nr, int s, Body body)
{
#pragma unroll
	for (int t = 0; t < NT; ++t)
		body(t, s + 4 * t < nr);
}

// ---------------------------------------------------------------------------
// Householder QR with column pivoting on a register panel.
//   x[sl][t]   : entry s+4t of column q = c + 16*sl (entries >= nr must be 0)
//   pos[sl]    : -1 = active (unpivoted) column, >= 0 = pivoted at that step, -2 = no column
//   [qlo, qhi) : columns that may be chosen as pivots in this call
// Step k: the pivot quad dumps its column into VR[k*LDV + i]; entries >= k of that row are the
// Householder vector (v_k at i == k), entries < k the R entries R(i, k); rdiag[k] = R(k, k),
// tau[k] = 2 / v'v, pivlane[k] = pivot column, colpos[column] = k.  Stops when the largest
// remaining candidate column norm is <= rtol * (first pivot norm of this call).  Returns the
// new step count.
// ---------------------------------------------------------------------------
#ifdef HOQP_PROFILE
#define PF_T(i)                                                       \
	do {                                                          \
		unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
		pt[i] += n_ - t0_;                                    \
		t0_ = n_;                                             \
	} while (0)
#else
#define PF_T(i) \
	do {    \
	} while (0)
#endif

// NTA: register blocks of four entries that can be non-zero (vector length nr <= 4 NTA): the loops over a
// vector's registers stop there -- at the tracking level of the WBC (n_z = 12) that is 3 of 8.
// (Natural column order -- no pivot search -- was tried for levels 0 and 2 of the fused reference kernel: 0.7 %,
// the search is not where a step waits; see DESIGN.md, round 3 negatives.)
template <int NS, int NTA>
__device__ __forceinline__ int panel_factor_t(double (&x)[NSP][NT], int (&pos)[NSP], int qlo, int qhi, int nr, int k0,
					      double rtol, const Lds &l, unsigned long long *dbgp)
{
#ifdef HOQP_PROFILE
	unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0};
	unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
	const int lane = lane_id(), s = lane & 3, c = lane >> 2;
	ldsd *VR = l.TV;
	int k = k0;
	double first = -1.0;
	while (k < nr) {
		double best = -1.0;
		int bq = -1;
#pragma unroll
		for (int sl = 0; sl < NS; ++sl) {
			double a0 = 0.0, a1 = 0.0;
#pragma unroll
			for (int t = 0; t < NTA; ++t) {
				const double v = (s + 4 * t >= k) ? x[sl][t] : 0.0;
				if (t & 1)
					a1 = fma(v, v, a1);
				else
					a0 = fma(v, v, a0);
			}
			const double ss = quad_sum(a0 + a1);
			const int q = c + 16 * sl;
			const bool mine = pos[sl] == -1 && q >= qlo && q < qhi;
			const bool take = mine && ss > best;
			best = take ? ss : best;
			bq = take ? q : bq;
		}
		PF_T(0);
		// pivot = column with the largest key (top 32 bits of the squared norm: any column within
		// 2^-20 of the largest is as good a pivot); its exact squared norm is read back
		const int wl = wave_argmax_key(bq >= 0 ? (unsigned)__double2hiint(best) : 0u);
		if (wl < 0)
			break;
		const int pq = __builtin_amdgcn_readlane(bq, wl);
		const double bmax = readlane_d(best, wl);
		if (pq < 0 || !(bmax > 1e-290))
			break;
		PF_T(1);
		const double rn = fast_rsqrt(bmax);  // 1 / norm
		double nrm = bmax * rn;
		nrm = fma(fma(-nrm, nrm, bmax), 0.5 * rn, nrm);
		if (first < 0.0)
			first = nrm;
		if (!(nrm > rtol * first))
			break;
		PF_T(2);
		const int pc = pq & 15, ps = pq >> 4;
		ldsd *vr = VR + k * l.ldv;
		if (c == pc) {
#pragma unroll
			for (int sl = 0; sl < NS; ++sl) {
				if (NS > 1 && sl != ps)
					continue;
				// the whole row of the store is written (zeros beyond nr): its readers take full rows
				t_upto(nr, s, [&](int t, bool) { vr[s + 4 * t] = x[sl][t]; });
				pos[sl] = k;
			}
		}
		wave_sync();
		double alpha = vr[k];
		double vv[NT];
#pragma unroll
		for (int t = 0; t < NTA; ++t)
			vv[t] = vr[s + 4 * t];
		pin(alpha);
#pragma unroll
		for (int t = 0; t < NTA; ++t)
			pin(vv[t]);
		PF_T(3);
		const double beta = (alpha >= 0.0) ? -nrm : nrm;
		const double v0 = alpha - beta;
		const double tk = rn * fast_rcp(nrm + fabs(alpha));  // 2 / v'v = 1 / (nrm (nrm + |alpha|))
#pragma unroll
		for (int t = 0; t < NTA; ++t) {
			const double u = (s + 4 * t == k) ? v0 : vv[t];
			vv[t] = (s + 4 * t >= k) ? u : 0.0;
		}
		PF_T(4);
#pragma unroll
		for (int sl = 0; sl < NS; ++sl) {
			double d0 = 0.0, d1 = 0.0;
#pragma unroll
			for (int t = 0; t < NTA; ++t) {
				if (t & 1)
					d1 = fma(vv[t], x[sl][t], d1);
				else
					d0 = fma(vv[t], x[sl][t], d0);
			}
			double d = quad_sum(d0 + d1) * tk;
			d = (pos[sl] == -1) ? d : 0.0;
#pragma unroll
			for (int t = 0; t < NTA; ++t)
				x[sl][t] = fma(-d, vv[t], x[sl][t]);
		}
		if (lane == 0) {
			vr[k] = v0;  // program order: after every lane's read of vr[k] above
			l.tau[k] = tk;
			l.rdiag[k] = beta;
			l.pivlane[k] = pq;
			l.colpos[pq] = k;
		}
		PF_T(5);
		++k;
	}
	wave_sync();
#ifdef HOQP_PROFILE
	if (dbgp && lane == 0) {
		for (int i = 0; i < 6; ++i)
			dbgp[16 + i] += pt[i];
		dbgp[22] += k - k0;
	}
#endif
	return k;
}

// slot count and vector length class fixed at compile time inside the step loop (a per-slot scalar branch in
// every step cost more than the slot's arithmetic); one dispatch per factorization
// VL: dispatch on the vector length class as well (the specialised level copies of the fused reference kernel;
// in the general kernels the three extra copies of every step cost more than they save: paper semantics
// 5.2 -> 4.4 M solves/s)
// VL (int): 0 = full length always; 1 = dispatch on 12 / 24 / 32; 2 = 24 or 32; 3 = 24 always (the caller knows that
// nr <= 24: the levels behind level 0 of the paper-semantics WBC kernel, whose reflector store has stride 25)
template <int NS, int VL>
__device__ __forceinline__ int panel_factor_n(double (&x)[NSP][NT], int (&pos)[NSP], int qlo, int qhi, int nr, int k0,
					      double rtol, const Lds &l, unsigned long long *dbgp)
{
	if (VL == 3)
		return panel_factor_t<NS, 6>(x, pos, qlo, qhi, nr, k0, rtol, l, dbgp);
	if (VL == 1 && nr <= 12)
		return panel_factor_t<NS, 3>(x, pos, qlo, qhi, nr, k0, rtol, l, dbgp);
	if (VL >= 1 && nr <= 24)
		return panel_factor_t<NS, 6>(x, pos, qlo, qhi, nr, k0, rtol, l, dbgp);
	return panel_factor_t<NS, NT>(x, pos, qlo, qhi, nr, k0, rtol, l, dbgp);
}

template <int VL = 0>
__device__ __forceinline__ int panel_factor(double (&x)[NSP][NT], int (&pos)[NSP], int nslot, int qlo, int qhi, int nr, int k0,
				   double rtol, const Lds &l, unsigned long long *dbgp = nullptr)
{
	if (nslot <= 1)
		return panel_factor_n<1, VL>(x, pos, qlo, qhi, nr, k0, rtol, l, dbgp);
	if (nslot == 2)
		return panel_factor_n<2, VL>(x, pos, qlo, qhi, nr, k0, rtol, l, dbgp);
	return panel_factor_n<NSP, VL>(x, pos, qlo, qhi, nr, k0, rtol, l, dbgp);
}

// Z = I (n x n, leading dimension ld): InitPrevProblemValuesToDefault, ho_qp_problem.cpp:119-128
__device__ __forceinline__ void lds_set_identity(ldsd *Z, int n, int ld)
{
	const int lane = lane_id();
	// (row, column) of entry idx advance by 64 entries per trip without a division per trip
	const int qstep = kWave / ld, rstep = kWave - qstep * ld;
	int r = lane / ld, c = lane - r * ld;
	for (int idx = lane; idx < n * ld; idx += kWave) {
		Z[idx] = (r == c) ? 1.0 : 0.0;
		r += qstep;
		c += rstep;
		if (c >= ld) {
			c -= ld;
			++r;
		}
	}
}

// rows <- rows * H_ka ... H_(kb-1): reflectors [ka, kb) applied from the right to row-major
// rows.  Row r = c + 16*slot is a vector owned by quad c; rows r < n0 come from A0, the rest
// from A1.
// IDENT: the rows of A0 are the rows of the identity and are not read (level 0 of a hierarchy starts from
// Z = I: the rotation then writes Z = Q without Z ever having been initialised).
// UPDATE: the level update in the same pass (the level's y, rotated coordinates, is already in l.tmp): after
// the rotation the row is still in registers, so d_r = row_r . y -> l.S[r] and the row is stored without its
// first kb coordinates (Z_k = Z_{k-1} Q_2) -- level_update_t without a second trip of every row through LDS.
// NTA: register blocks of four entries that can be non-zero (nr <= 4 NTA), as in panel_factor_t.
template <int NS, bool IDENT = false, bool UPDATE = false, int NTA = NT>
__device__ __forceinline__ void rows_apply_reflectors_t(ldsd *A0, int n0, ldsd *A1, int n1, int ldr, int nr, int ka,
							int kb, const Lds &l)
{
	const int lane = lane_id(), s = lane & 3, c = lane >> 2;
	const int R = n0 + n1;
	const ldsd *VR = l.TV;
	if (R <= 0 || kb <= ka) {
		wave_sync();
		return;
	}
	double x[NSR][NT];
	ldsd *rowp[NSR];
	bool valid[NSR];
#pragma unroll
	for (int sl = 0; sl < NS; ++sl) {
		const int r = c + 16 * sl;
		valid[sl] = r < R;
		const int rr = valid[sl] ? r : 0;
		rowp[sl] = (rr < n0) ? A0 + rr * ldr : A1 + (rr - n0) * ldr;
		if (!IDENT) {
#pragma unroll
			for (int t = 0; t < NTA; ++t)
				x[sl][t] = rowp[sl][s + 4 * t];
		}
	}
	if (IDENT) {
#pragma unroll
		for (int sl = 0; sl < NS; ++sl) {
			const int r = c + 16 * sl;
#pragma unroll
			for (int t = 0; t < NTA; ++t)
				x[sl][t] = (s + 4 * t < nr && valid[sl] && s + 4 * t == r) ? 1.0 : 0.0;
		}
	} else {
#pragma unroll
		for (int sl = 0; sl < NS; ++sl) {
#pragma unroll
			for (int t = 0; t < NTA; ++t)
				pin(x[sl][t]);
		}
#pragma unroll
		for (int sl = 0; sl < NS; ++sl) {
#pragma unroll
			for (int t = 0; t < NTA; ++t)
				x[sl][t] = sel(s + 4 * t < nr && valid[sl], x[sl][t]);
		}
	}
	for (int k = ka; k < kb; ++k) {
		const ldsd *vr = VR + k * l.ldv;
		double tk = l.tau[k];
		double vv[NT];
#pragma unroll
		for (int t = 0; t < NTA; ++t)
			vv[t] = vr[s + 4 * t];
		pin(tk);
#pragma unroll
		for (int t = 0; t < NTA; ++t)
			pin(vv[t]);
#pragma unroll
		for (int t = 0; t < NTA; ++t)
			vv[t] = sel(s + 4 * t >= k, vv[t]);
#pragma unroll
		for (int sl = 0; sl < NS; ++sl) {
			double d0 = 0.0, d1 = 0.0;
#pragma unroll
			for (int t = 0; t < NTA; ++t) {
				if (t & 1)
					d1 = fma(vv[t], x[sl][t], d1);
				else
					d0 = fma(vv[t], x[sl][t], d0);
			}
			const double d = quad_sum(d0 + d1) * tk;
#pragma unroll
			for (int t = 0; t < NTA; ++t)
				x[sl][t] = fma(-d, vv[t], x[sl][t]);
		}
	}
	if (UPDATE) {
		double yv[NT];
#pragma unroll
		for (int t = 0; t < NTA; ++t)
			yv[t] = l.tmp[s + 4 * t];
#pragma unroll
		for (int t = 0; t < NTA; ++t)
			pin(yv[t]);
#pragma unroll
		for (int t = 0; t < NTA; ++t)
			yv[t] = sel(s + 4 * t < nr, yv[t]);
#pragma unroll
		for (int sl = 0; sl < NS; ++sl) {
			double d0 = 0.0, d1 = 0.0;
#pragma unroll
			for (int t = 0; t < NTA; ++t) {
				if (t & 1)
					d1 = fma(x[sl][t], yv[t], d1);
				else
					d0 = fma(x[sl][t], yv[t], d0);
			}
			const double d = quad_sum(d0 + d1);
			if (valid[sl] && s == 0)
				l.S[c + 16 * sl] = d;
		}
		if (kb < nr) {
#pragma unroll
			for (int sl = 0; sl < NS; ++sl) {
#pragma unroll
				for (int t = 0; t < NTA; ++t) {
					const int i = s + 4 * t;
					st_if(i >= kb && i < nr && valid[sl], rowp[sl] + (i - kb), l.dump + lane, x[sl][t]);
				}
			}
		}
		wave_sync();
		return;
	}
#pragma unroll
	for (int sl = 0; sl < NS; ++sl) {
#pragma unroll
		for (int t = 0; t < NTA; ++t)
			st_if(s + 4 * t < nr && valid[sl], rowp[sl] + (s + 4 * t), l.dump + lane, x[sl][t]);
	}
	wave_sync();
}

// slot count fixed at compile time (see panel_factor)
__device__ __forceinline__ void rows_apply_reflectors(ldsd *A0, int n0, ldsd *A1, int n1, int ldr, int nr, int ka, int kb,
					     const Lds &l)
{
	const int R = n0 + n1;
	if (R <= 32)
		rows_apply_reflectors_t<2>(A0, n0, A1, n1, ldr, nr, ka, kb, l);
	else if (R <= 64)
		rows_apply_reflectors_t<4>(A0, n0, A1, n1, ldr, nr, ka, kb, l);
	else
		rows_apply_reflectors_t<NSR>(A0, n0, A1, n1, ldr, nr, ka, kb, l);
}

// out[r] = rows[r][0..nr) . vec[0..nr) for r < R (row r owned by quad r & 15, slot r >> 4)
template <int NS>
__device__ __forceinline__ void rows_dot(const ldsd *A, int R, int ldr, int nr, const ldsd *vec, ldsd *out)
{
	const int lane = lane_id(), s = lane & 3, c = lane >> 2;
	if (R <= 0)
		return;
	double vv[NT];
	double ev[NS][NT];
	ld8(vv, vec + s);
#pragma unroll
	for (int sl = 0; sl < NS; ++sl) {
		const int r = c + 16 * sl;
		ld8(ev[sl], A + (r < R ? r : 0) * ldr + s);
	}
	pin8(vv);
#pragma unroll
	for (int sl = 0; sl < NS; ++sl)
		pin8(ev[sl]);
	t_upto(nr, s, [&](int t, bool on) { vv[t] = sel(on, vv[t]); });
#pragma unroll
	for (int sl = 0; sl < NS; ++sl) {
		const int r = c + 16 * sl;
		double d0 = 0.0, d1 = 0.0;
		t_upto(nr, s, [&](int t, bool on) {
			const double e = sel(on, ev[sl][t]);
			if (t & 1)
				d1 = fma(e, vv[t], d1);
			else
				d0 = fma(e, vv[t], d0);
		});
		const double d = quad_sum(d0 + d1);
		if (r < R && s == 0)
			out[r] = d;
	}
}

// ---------------------------------------------------------------------------
// Virtual rows of the torque / friction block (level 1 of the fused reference kernel, compact layout).  Of its
// 24 + 4 n_c projected rows only the 12 torque rows [M_j, -J_c,j^T] Z are stored (l.G); the others are
//   rows 12..23 : the negated rows 0..11                         (ConstructJointTorqueTask, [D; -D])
//   rows 24..   : +-Z[col + (t >> 1)] - mu Z[col + 2]            (ConstructFrictionConeTask, two non-zeros per row)
// i.e. every row is  s0 * p0[j] + s1 * p1[j]  over rows p0, p1 of l.G or l.Z -- formed on the fly with one
// multiplication by +-1 (exact) and one FMA, bit-identical to the rows wbc_project_level1 used to store.
// Z is not touched while level 1 runs (no equality rows: no rotation), so the rows stay valid.
// ---------------------------------------------------------------------------
struct VRowCtx {
	const ldsd *G, *Z;
	int ld, clp, nc, quirk;
	double mu;
};
struct VRow {
	const ldsd *p0, *p1;
	double s0, s1;
};
__device__ __forceinline__ VRow vrow(const VRowCtx &v, int r)
{
	VRow o;
	const int rr = (r < kNJ) ? r : r - kNJ;
	const int f = (r >= 2 * kNJ) ? r - 2 * kNJ : 0;
	const int i = f >> 2, t = f & 3;
	const int col = v.quirk ? kNB + wbc_pick(v.clp, i) * 3 : kNU + 3 * i;
	const bool tq = r < 2 * kNJ;
	o.p0 = tq ? v.G + (rr < kNJ ? rr : 0) * v.ld : v.Z + (col + (t >> 1)) * v.ld;
	o.p1 = tq ? o.p0 : v.Z + (col + 2) * v.ld;
	o.s0 = tq ? ((r < kNJ) ? 1.0 : -1.0) : ((t & 1) ? -1.0 : 1.0);
	o.s1 = tq ? 0.0 : 0.0 - v.mu;
	return o;
}
__device__ __forceinline__ double vval(const VRow &r, double a, double b) { return fma(r.s1, b, r.s0 * a); }

// rows_dot over the virtual rows: out[r] = row_r[0..nr) . vec[0..nr) for r < R
template <int NS>
__device__ __forceinline__ void rows_dot_v(const VRowCtx &vc, int R, int nr, const ldsd *vec, ldsd *out)
{
	const int lane = lane_id(), s = lane & 3, c = lane >> 2;
	if (R <= 0)
		return;
	double vv[NT];
	double ea[NS][NT], eb[NS][NT];
	VRow vr[NS];
	ld8(vv, vec + s);
#pragma unroll
	for (int sl = 0; sl < NS; ++sl) {
		const int r = c + 16 * sl;
		vr[sl] = vrow(vc, r < R ? r : 0);
		ld8(ea[sl], vr[sl].p0 + s);
		ld8(eb[sl], vr[sl].p1 + s);
	}
	pin8(vv);
#pragma unroll
	for (int sl = 0; sl < NS; ++sl) {
		pin8(ea[sl]);
		pin8(eb[sl]);
	}
	t_upto(nr, s, [&](int t, bool on) { vv[t] = sel(on, vv[t]); });
#pragma unroll
	for (int sl = 0; sl < NS; ++sl) {
		const int r = c + 16 * sl;
		double d0 = 0.0, d1 = 0.0;
		t_upto(nr, s, [&](int t, bool on) {
			const double e = sel(on, vval(vr[sl], ea[sl][t], eb[sl][t]));
			if (t & 1)
				d1 = fma(e, vv[t], d1);
			else
				d0 = fma(e, vv[t], d0);
		});
		const double d = quad_sum(d0 + d1);
		if (r < R && s == 0)
			out[r] = d;
	}
}

// One entry per lane.  forward: vec <- H_(kb-1) ... H_ka vec ; else vec <- H_ka ... H_(kb-1) vec.
__device__ __forceinline__ double hh_apply_vec(double vi, int nr, int ka, int kb, bool forward, const Lds &l)
{
	const int lane = lane_id();
	for (int t = 0; t < kb - ka; ++t) {
		int k = forward ? ka + t : kb - 1 - t;
		double v = sel(lane >= k && lane < nr, l.TV[k * l.ldv + (lane & (NR - 1))]);
		double s = wave_sum(v * vi) * l.tau[k];
		vi = fma(-s, v, vi);
	}
	return vi;
}

// ---------------------------------------------------------------------------
// Triangular solves on the reflector store, one row per lane.  Pivot positions [k1, k1+rb)
// hold R (rb x rb, upper): R(i, j) = VR[(k1+j)*LDV + k1+i] for i < j, R(j, j) = rdiag[k1+j].
//   Lrow: row `lane` of L = R'  (L(lane, i) = R(i, lane), i < lane; contiguous in VR)
//   Urow: row `lane` of U = R   (U(lane, j) = R(lane, j), j > lane; strided in VR)
// The substitution loops are static over NR in blocks of 8 (one scalar branch per block).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load_Lrow(double (&Lrow)[NR], int k1, int rb, const Lds &l)
{
	const int lane = lane_id();
	const bool act = lane < rb;
	const ldsd *vr = l.TV + (k1 + (act ? lane : 0)) * l.ldv + k1;
	// only the blocks of 8 below rb are loaded and masked (one scalar branch per block)
#pragma unroll
	for (int b = 0; b < NR / 8; ++b) {
		if (8 * b >= rb) {
#pragma unroll
			for (int i = 8 * b; i < 8 * b + 8; ++i)
				Lrow[i] = 0.0;
			continue;
		}
#pragma unroll
		for (int i = 8 * b; i < 8 * b + 8; ++i)
			Lrow[i] = vr[i];
#pragma unroll
		for (int i = 8 * b; i < 8 * b + 8; ++i)
			pin(Lrow[i]);
#pragma unroll
		for (int i = 8 * b; i < 8 * b + 8; ++i)
			Lrow[i] = sel(act && i < lane, Lrow[i]);
	}
}

__device__ __forceinline__ void load_Urow(double (&Urow)[NR], int k1, int rb, const Lds &l)
{
	const int lane = lane_id();
	const bool act = lane < rb;
	const int ldv = l.ldv;
	const ldsd *vc = l.TV + k1 * ldv + k1 + (act ? lane : 0);
#pragma unroll
	for (int b = 0; b < NR / 8; ++b) {
		if (8 * b >= rb) {
#pragma unroll
			for (int j = 8 * b; j < 8 * b + 8; ++j)
				Urow[j] = 0.0;
			continue;
		}
#pragma unroll
		for (int j = 8 * b; j < 8 * b + 8; ++j)
			Urow[j] = vc[(j < rb ? j : 0) * ldv];
#pragma unroll
		for (int j = 8 * b; j < 8 * b + 8; ++j)
			pin(Urow[j]);
#pragma unroll
		for (int j = 8 * b; j < 8 * b + 8; ++j)
			Urow[j] = sel(act && j > lane && j < rb, Urow[j]);
	}
}

// L y = t  (inv = 1 / L(lane, lane), 0 for lanes >= rb); returns y_lane
__device__ __forceinline__ double tri_fwd(const double (&Lrow)[NR], double inv, double t, int rb)
{
	// Lrow[i] is zero for i >= lane, so a lane's accumulator is final after step lane - 1 and
	// y_lane = acc * inv at the end (no per-step capture)
	double acc = t;
#pragma unroll
	for (int b = 0; b < NR / 8; ++b) {
		if (8 * b >= rb)
			continue;
#pragma unroll
		for (int i = 8 * b; i < 8 * b + 8; ++i) {
			const double yi = readlane_d(acc * inv, i);
			acc = fma(-Lrow[i], yi, acc);
		}
	}
	return acc * inv;
}

// U z = t; returns z_lane
__device__ __forceinline__ double tri_bwd(const double (&Urow)[NR], double inv, double t, int rb)
{
	double acc = t;  // Urow[i] is zero for i <= lane: z_lane = acc * inv at the end
#pragma unroll
	for (int b = NR / 8 - 1; b >= 0; --b) {
		if (8 * b >= rb)
			continue;
#pragma unroll
		for (int i = 8 * b + 7; i >= 8 * b; --i) {
			const double zi = readlane_d(acc * inv, i);
			acc = fma(-Urow[i], zi, acc);
		}
	}
	return acc * inv;
}

// Solves the (<= 8 x 8) symmetric positive definite system M a = g by Gauss-Jordan without
// pivoting.  Lane (i, j) = (lane >> 3, lane & 7) holds M(i, j) in mval and g_i in gval; rows
// beyond the system size hold the identity / 0.  Returns a_(lane >> 3).
__device__ __forceinline__ double gauss_jordan8(double mval, double gval)
{
	const int lane = lane_id(), i = lane >> 3, j = lane & 7;
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const double rpv = fast_rcp(readlane_d(mval, 9 * k));
		const double mik = __shfl(mval, i * 8 + k);
		const double mkj = __shfl(mval, k * 8 + j) * rpv;  // normalised pivot row
		const double gk = readlane_d(gval, 8 * k) * rpv;
		mval = (i == k) ? mkj : fma(-mik, mkj, mval);
		gval = (i == k) ? gk : fma(-mik, gk, gval);
	}
	return gval;
}

// The same with the smallest pivot reported (conditioning guard of the normal-equations shortcut).
__device__ __forceinline__ double gauss_jordan8_piv(double mval, double gval, double *minpiv)
{
	const int lane = lane_id(), i = lane >> 3, j = lane & 7;
	double mp = 1e300;
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const double pv = readlane_d(mval, 9 * k);
		mp = fmin(mp, pv);
		const double rpv = fast_rcp(pv);
		const double mik = __shfl(mval, i * 8 + k);
		const double mkj = __shfl(mval, k * 8 + j) * rpv;
		const double gk = readlane_d(gval, 8 * k) * rpv;
		mval = (i == k) ? mkj : fma(-mik, mkj, mval);
		gval = (i == k) ? gk : fma(-mik, gk, gval);
	}
	*minpiv = mp;
	return gval;
}

// ---------------------------------------------------------------------------
// Least squares on a finished factorization.  Pivot positions [k1, k2) hold the factor R_B
// (rb = k2 - k1 rows) of the "B" columns [qB0, qB1); the right-hand side is in l.rhs[column],
// the pivot position of a column in l.colpos[column] (-1 = unpivoted).
// Solves min |R_B' q - rhs| and returns q_j in lane j (j < rb); *res_norm2 = |residual|^2.
// eps != 0 adds the first order effect of the reference's eps*I term
// (ho_qp_problem.cpp:244-246):  q <- q - eps (R_B R_B')^-1 q.
// Unpivoted B columns (over-determined case) must have been dumped by the caller to
// l.X[u*ld + i] with their lanes listed in l.xlane[0..nunp).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double ls_solve_reg(int k1, int k2, int qB0, int qB1, int nunp, double eps, int ld,
				      double *res_norm2, const Lds &l, double *dmin_out = nullptr)
{
	const int lane = lane_id();
	const int rb = k2 - k1;
	const ldsd *VR = l.TV;
	wave_sync();
	const bool isB = lane >= qB0 && lane < qB1;  // lane == column here
	const double rhs_c = l.rhs[lane];
	const int mypos = l.colpos[lane];
	if (rb <= 0) {
		if (res_norm2)
			*res_norm2 = wave_sum(isB ? rhs_c * rhs_c : 0.0);
		return 0.0;
	}
	const bool act = lane < rb;
	const int pl = l.pivlane[k1 + (act ? lane : 0)];
	const double rp = sel(act, l.rhs[pl]);  // rhs by pivot position
	const double rdg = l.rdiag[k1 + (act ? lane : 0)];
	const double inv = sel(act, fast_rcp(rdg));
	if (nunp == 0) {
		double Lrow[NR];
		load_Lrow(Lrow, k1, rb, l);
		double q = tri_fwd(Lrow, inv, rp, rb);
		// The eps term moves q by about eps / sigma_min(R)^2 relative; with the smallest pivot as
		// the estimate of sigma_min the two extra triangular solves are skipped while that is
		// below 1e-10 (tolerance of the whole path: 1e-6).
		const double dmin = wave_min(act ? fabs(rdg) : 1e300);
		if (dmin_out)
			*dmin_out = dmin;
		if (eps != 0.0 && eps > 1e-10 * dmin * dmin) {
			double Urow[NR];
			load_Urow(Urow, k1, rb, l);
			const double z = tri_bwd(Urow, inv, q, rb);  // U z = q
			q -= eps * tri_fwd(Lrow, inv, z, rb);	     // (B'B)^-1 q0 = L^-1 U^-1 q0 with B = L, B' = U
			// residual of the eps-regularised solution: L (q0 - q) = eps z
			if (res_norm2)
				*res_norm2 = eps * eps * wave_sum(act ? z * z : 0.0);
		} else if (res_norm2) {
			*res_norm2 = 0.0;
		}
		return act ? q : 0.0;
	}

	if (nunp <= 8) {
		// ---- rank-deficient B with few dependent columns: the pivoted block R1 (rb x rb) is
		// nonsingular, the nu unpivoted columns are R1 w_u.  With u = R1' q the problem is
		//   min |u - d1|^2 + |W'u - d2|^2   =>   u = d1 + W a,  (I + W'W) a = d2 - W'd1,
		// a nu x nu SPD system (pivots >= 1) solved by an 8 x 8 Gauss-Jordan, one entry per lane.
		double Lrow[NR], Urow[NR];
		load_Lrow(Lrow, k1, rb, l);
		load_Urow(Urow, k1, rb, l);
		ldsd *Wk = l.X + k1;
		for (int u = 0; u < nunp; ++u) {  // w_u = R1^-1 r2_u, in place
			const double r2 = sel(act, Wk[u * ld + (act ? lane : 0)]);
			const double wv = tri_bwd(Urow, inv, r2, rb);
			if (act)
				Wk[u * ld + lane] = wv;
		}
		wave_sync();
		const int gi = lane >> 3, gj = lane & 7;
		double mval = (gi == gj) ? 1.0 : 0.0, gval = 0.0;
		for (int u = 0; u < nunp; ++u) {
			const double wu = sel(act, Wk[u * ld + (act ? lane : 0)]);
			const double gu = l.rhs[l.xlane[u]] - wave_sum(wu * rp);
			gval = (gi == u) ? gu : gval;
			for (int v = 0; v <= u; ++v) {
				const double wv = sel(act, Wk[v * ld + (act ? lane : 0)]);
				const double mv = wave_sum(wu * wv) + ((u == v) ? 1.0 : 0.0);
				mval = (lane == u * 8 + v || lane == v * 8 + u) ? mv : mval;
			}
		}
		const double av = gauss_jordan8(mval, gval);  // a_(lane >> 3)
		double e = 0.0;
		for (int u = 0; u < nunp; ++u)
			e = fma(readlane_d(av, 8 * u), sel(act, Wk[u * ld + (act ? lane : 0)]), e);
		double q = tri_fwd(Lrow, inv, rp + e, rb);
		if (res_norm2)  // |u - d1|^2 + |W'u - d2|^2 = |e|^2 + |a|^2
			*res_norm2 = wave_sum(e * e) + wave_sum(gj == 0 ? av * av : 0.0);
		if (eps != 0.0) {
			// q <- q - eps (R_B R_B')^-1 q,  (R_B R_B')^-1 = R1^-T (I - W (I + W'W)^-1 W') R1^-1
			double z = tri_bwd(Urow, inv, q, rb);
			double hval = 0.0;
			for (int u = 0; u < nunp; ++u) {
				const double hu = wave_sum(sel(act, Wk[u * ld + (act ? lane : 0)]) * z);
				hval = (gi == u) ? hu : hval;
			}
			const double bv = gauss_jordan8(mval, hval);
			for (int u = 0; u < nunp; ++u)
				z = fma(-readlane_d(bv, 8 * u), sel(act, Wk[u * ld + (act ? lane : 0)]), z);
			q -= eps * tri_fwd(Lrow, inv, z, rb);
		}
		wave_sync();
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
		const ldsd *vr = VR + (k1 + p) * l.ldv + k1;
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
			wave_sync();
			if (lane == i) {
#pragma unroll
				for (int j = 0; j < NR; ++j)
					if (j < i)
						l.tmp[32 + j] = Crow[j];
			}
			wave_sync();
			if (lane < i)
				acc = fma(-l.tmp[32 + lane], zi, acc);
		}
		return z;
	};
	auto RB_vec = [&]() -> double {  // (R_B w)_lane with w per column in l.rhs[column lane]
		double acc = 0.0;
		if (act) {
			for (int p = lane; p < rb; ++p) {
				const double e = (p == lane) ? l.rdiag[k1 + p] : VR[(k1 + p) * l.ldv + k1 + lane];
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
			const ldsd *vr = VR + mypos * l.ldv + k1;
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
	wave_sync();
	if (act)
		l.tmp[lane] = q;
	wave_sync();
	double res = residual(rhs_c);
	wave_sync();
	l.rhs[lane] = res;
	wave_sync();
	q += chol_solve(RB_vec());  // one refinement step
	if (eps != 0.0)
		q -= eps * chol_solve(q);
	wave_sync();
	if (act)
		l.tmp[lane] = q;
	wave_sync();
	if (res_norm2) {
		res = residual(rhs_c);
		*res_norm2 = wave_sum(res * res);
	}
	wave_sync();
	return act ? q : 0.0;
}



// ---------------------------------------------------------------------------
// Task projection.  Two task rows per pass: lanes 0-31 take row 2i, lanes 32-63 row 2i+1
// (one row per pass in the rare case nz == 32).  Within a half, lane j < nz owns column j
// of Z (zc[i] = Z[i][j]) and the spare lane (31 of the half) owns x.  For a staged task
// row a_r:  lane j gets (a_r Z)_j, the spare lane gets a_r x.
//   to_T : (A Z)' -> TV[r*ld + j] (vector r of the QR panel), rhs -> l.rhs[r] = b_r - a_r x
//   else : D Z   -> G[(P+r)*ld + j],                           g[P+r]       = f_r - d_r x
// ---------------------------------------------------------------------------
struct ProjLane {
	bool dual, isZ, isX;
	int j;
};

__device__ __forceinline__ ProjLane proj_lane(int nz)
{
	const int lane = lane_id();
	ProjLane p;
	p.dual = nz < 32;
	p.j = p.dual ? (lane & 31) : lane;
	p.isZ = p.j < nz;
	p.isX = p.dual ? ((lane & 31) == 31) : (lane == 63);
	return p;
}

__device__ __forceinline__ void project_task(const double *__restrict__ rows_g, const double *__restrict__ rhs_g, int rows,
				    int nstore, int n, int nz, int ld, const double (&zc)[NR], bool to_T, int P,
				    const Lds &l)
{
	const int lane = lane_id();
	const ProjLane pl = proj_lane(nz);
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
			// clamped, never masked: rows >= nr are not used, columns >= nstore meet zc == 0
			pre[t] = rows_g[(size_t)(r0 + (r < nr ? r : 0)) * nstore + cc];
		}
		rv_next = rhs_g[r0 + (lane < nr ? lane : 0)];
	};
	fetch(0);
	for (int r0 = 0; r0 < rows; r0 += CHUNK) {
		const int nr = min(CHUNK, rows - r0);
		wave_sync();  // readers of the previous chunk are done with S
#pragma unroll
		for (int t = 0; t < CHUNK / 2; ++t)
			l.S[(2 * t + h) * SROW + c] = sel(c < n, pre[t]);  // storage columns >= n are not part of the problem
		const double rv = rv_next;
		wave_sync();
		if (r0 + CHUNK < rows)
			fetch(r0 + CHUNK);
		const int npass = pl.dual ? (nr + 1) >> 1 : nr;
		for (int i = 0; i < npass; ++i) {
			const int r = pl.dual ? 2 * i + h : i;  // this lane's row (r == nr: odd tail, discarded)
			const ldsd2 *sr2 = (const ldsd2 *)(l.S + r * SROW);
			double a0 = 0.0, a1 = 0.0;
			// all 16 reads are issued back to back (one LDS round trip per row pair); entries
			// >= n multiply zc == 0
			double te[NR];
#pragma unroll
			for (int e = 0; e < NR; e += 2) {
				const d2v t = sr2[e >> 1];
				te[e] = t.x;
				te[e + 1] = t.y;
			}
#pragma unroll
			for (int e = 0; e < NR; ++e)
				pin(te[e]);
#pragma unroll
			for (int e = 0; e < NR; e += 2) {
				a0 = fma(te[e], zc[e], a0);
				a1 = fma(te[e + 1], zc[e + 1], a1);
			}
			const double acc = a0 + a1;
			const int ra = pl.dual ? 2 * i : i;
			const double rrA = readlane_d(rv, ra), rrB = readlane_d(rv, (ra + 1) & 63);
			const double rr = (pl.dual && h) ? rrB : rrA;
			const bool rok = r < nr;
			// one store per lane: the projected entry, the right-hand side, or the sink
			ldsd *dst = l.dump + lane;
			if (to_T) {
				dst = (pl.isZ && rok) ? l.TV + (r0 + r) * ld + pl.j : dst;
				dst = (pl.isX && rok) ? l.rhs + (r0 + r) : dst;
			} else {
				dst = (pl.isZ && rok) ? l.G + (P + r0 + r) * ld + pl.j : dst;
				dst = (pl.isX && rok) ? l.g + (P + r0 + r) : dst;
			}
			*dst = pl.isX ? rr - acc : acc;
		}
	}
	wave_sync();
}

// ---------------------------------------------------------------------------
// Task projection on the matrix cores (n_z <= 31).  A chunk of 16 task rows is one
// 16 x K operand tile of v_mfma_f64_16x16x4_f64; [Z | x] (n x (n_z + 1)) is the other operand,
// kept in registers for the whole level:  bm[kb][cb] = B[k = 4 kb + lane/16][j = 16 cb + lane%16]
// with B[k][j] = Z[k][j] (j < n_z), x[k] (j == n_z), 0 otherwise / for k >= n.
// Operand layout of the instruction on gfx950 (probed, tools/mfma/mfma_layout.hip):
//   A[i = lane%16][k = lane/16],  B[k = lane/16][j = lane%16],  D[i = lane/16 + 4 r][j = lane%16].
// Outputs as in project_task: (A Z) rows -> TV / G, right-hand side minus (a_r x) -> rhs / g.
// ---------------------------------------------------------------------------
typedef double d4v __attribute__((ext_vector_type(4)));

// B operand [Zm | xv] from any LDS matrix Zm (n x nz, leading dimension ldz) and vector xv; `ident`:
// Zm is not read and stands for the identity.  (An explicit flag, never a null test: the solver's Z
// sits at LDS offset 0, and offset 0 is what a null test of an LDS pointer compares with.)
template <int NKB = NR / 4>
__device__ __forceinline__ void load_bm_from(double (&bm)[NR / 4][2], int n, int nz, int ldz, const ldsd *Zm, const ldsd *xv,
					     bool ident)
{
	const int lane = lane_id(), j16 = lane & 15, q4 = lane >> 4;
#pragma unroll
	for (int kb = 0; kb < NKB; ++kb) {
		const int k = 4 * kb + q4;
		const int kc = (k < n) ? k : 0;
#pragma unroll
		for (int cb = 0; cb < 2; ++cb) {
			const int j = 16 * cb + j16;
			const ldsd *src = (j < nz && !ident) ? Zm + kc * ldz + j : xv + kc;
			bm[kb][cb] = *src;
		}
	}
#pragma unroll
	for (int kb = 0; kb < NKB; ++kb) {
		pin(bm[kb][0]);
		pin(bm[kb][1]);
	}
#pragma unroll
	for (int kb = 0; kb < NKB; ++kb) {
		const int k = 4 * kb + q4;
#pragma unroll
		for (int cb = 0; cb < 2; ++cb) {
			const int j = 16 * cb + j16;
			double v = sel(k < n && j <= nz, bm[kb][cb]);
			if (ident && j < nz)
				v = (k == j && k < n) ? 1.0 : 0.0;
			bm[kb][cb] = v;
		}
	}
}

template <int NKB = NR / 4>
__device__ __forceinline__ void load_bm(double (&bm)[NR / 4][2], int n, int nz, int ld, const Lds &l)
{
	load_bm_from<NKB>(bm, n, nz, ld, l.Z, l.x, false);
}

// destination of a projection: rows of (A Z) to M[row * ldm + j], right-hand sides to rhs[row]
struct ProjDst {
	ldsd *M;
	int ldm;
	ldsd *rhs;
};

// Where the matrix-core operand of a task chunk comes from.  fetch() issues the loads of one
// chunk of 16 rows (the next chunk's loads are in flight during the multiply), take() waits for
// them and forms the operand: A[i = lane % 16][k = lane / 16 + 4 kb] and the right-hand sides of
// rows lane / 16 + 4 r.
//
// TaskRowsSrc: the rows of a stored task block (generic entry), each lane loading its own entries
// straight from global memory (16 rows x 32 contiguous bytes per load instruction).
struct TaskRowsSrc {
	static constexpr int NKB = NR / 4;
	const double *__restrict__ rows_g;
	const double *__restrict__ rhs_g;
	int nstore, n;
	double pre[NR / 4], rpre[4];
	__device__ __forceinline__ void fetch(int r0, int rows)
	{
		const int lane = lane_id(), j16 = lane & 15, q4 = lane >> 4;
		const int nr = min(CHUNK, rows - r0);
		const double *row = rows_g + (size_t)(r0 + (j16 < nr ? j16 : 0)) * nstore;
#pragma unroll
		for (int kb = 0; kb < NR / 4; ++kb) {
			const int k = 4 * kb + q4;
			pre[kb] = row[k < nstore ? k : 0];
		}
#pragma unroll
		for (int r = 0; r < 4; ++r)
			rpre[r] = rhs_g[r0 + (q4 + 4 * r < nr ? q4 + 4 * r : 0)];
	}
	__device__ __forceinline__ void take(double (&av)[NR / 4], double (&rvr)[4])
	{
		const int q4 = lane_id() >> 4;
#pragma unroll
		for (int kb = 0; kb < NR / 4; ++kb)
			av[kb] = pre[kb];
#pragma unroll
		for (int r = 0; r < 4; ++r)
			rvr[r] = rpre[r];
#pragma unroll
		for (int kb = 0; kb < NR / 4; ++kb)
			pin(av[kb]);
#pragma unroll
		for (int r = 0; r < 4; ++r)
			pin(rvr[r]);
		// columns >= n of the storage are not part of the instance's problem: whatever they hold
		// (NaN included) must not reach the product
#pragma unroll
		for (int kb = 0; kb < NR / 4; ++kb)
			av[kb] = sel(4 * kb + q4 < n, av[kb]);
	}
};

// WbcEntrySrc: level LEVEL of the WBC hierarchy through the general generator wbc_entry<LEVEL> (any level,
// any basis: used by the reference replay; the fast path has cheaper special forms, below).
template <int LEVEL>
struct WbcEntrySrc {
	static constexpr int NKB = NR / 4;
	const WbcCtx &x;
	const WbcPrm &prm;
	const Lds &l;
	const int (&jc)[NR / 4];  // wbc_jcoff of this lane's columns
	double lv[NR / 4], sc[NR / 4], ad[NR / 4], rl[4], rs[4], ra[4];
	__device__ __forceinline__ WbcEntrySrc(const WbcCtx &x_, const WbcPrm &prm_, const Lds &l_, const int (&jc_)[NR / 4])
		: x(x_), prm(prm_), l(l_), jc(jc_)
	{
	}
	__device__ __forceinline__ void fetch(int r0, int rows)
	{
		const int lane = lane_id(), j16 = lane & 15, q4 = lane >> 4;
		const int nr = min(CHUNK, rows - r0);
		const int r = r0 + (j16 < nr ? j16 : 0);
#pragma unroll
		for (int kb = 0; kb < NR / 4; ++kb) {
			const Entry e = wbc_entry<LEVEL>(x, prm, r, 4 * kb + q4, jc[kb]);
			lv[kb] = *e.p;
			sc[kb] = e.scale;
			ad[kb] = e.add;
		}
#pragma unroll
		for (int rr = 0; rr < 4; ++rr) {
			const int ri = r0 + (q4 + 4 * rr < nr ? q4 + 4 * rr : 0);
			if (LEVEL == 3) {
				rl[rr] = l.b3[ri];
				rs[rr] = 1.0;
				ra[rr] = 0.0;
			} else {
				const Entry e = wbc_rhs_entry<LEVEL>(x, prm, ri);
				rl[rr] = *e.p;
				rs[rr] = e.scale;
				ra[rr] = e.add;
			}
		}
	}
	__device__ __forceinline__ void take(double (&av)[NR / 4], double (&rvr)[4])
	{
#pragma unroll
		for (int kb = 0; kb < NR / 4; ++kb)
			pin(lv[kb]);
#pragma unroll
		for (int r = 0; r < 4; ++r)
			pin(rl[r]);
#pragma unroll
		for (int kb = 0; kb < NR / 4; ++kb)
			av[kb] = (sc[kb] != 0.0) ? sc[kb] * lv[kb] : ad[kb];
#pragma unroll
		for (int r = 0; r < 4; ++r)
			rvr[r] = fma(rs[r], rl[r], ra[r]);
	}
};

// WbcLevelSrc: the equality blocks of levels 2 and 3 of the WBC hierarchy generated from the dynamics terms
// (fused entry).  Every row of these blocks is a row of J_foot or J_b (or a unit row) padded with zeros
// beyond column 17, so a lane needs one row address per chunk and loads its entries at fixed offsets;
// only the first five blocks of four columns can be non-zero (NKB = 5: the other three take no part in
// the product).  Same values as wbc_entry<LEVEL> / wbc_rhs_entry<LEVEL> of wbc_tasks.hpp, which remain
// the definition (tests/test_gpu_fused.py compares the two).  Levels 0, 1 and 4 do not come through here.
constexpr int kWbcKB = 5;
template <int LEVEL>
struct WbcLevelSrc {
	static constexpr int NKB = kWbcKB;
	const WbcCtx &x;
	const Lds &l;
	double lv[NR / 4], rl[4];
	int rb;	 // level 3 without J_b: the unit row's column (-1: a Jacobian row)
	__device__ __forceinline__ WbcLevelSrc(const WbcCtx &x_, const Lds &l_) : x(x_), l(l_), rb(-1) {}
	__device__ __forceinline__ void fetch(int r0, int rows)
	{
		const int lane = lane_id(), j16 = lane & 15, q4 = lane >> 4;
		const int nr = min(CHUNK, rows - r0);
		const int r = r0 + (j16 < nr ? j16 : 0);
		const double *row;
		if (LEVEL == 2) {
			row = x.Jf + wbc_crow(x, r) * kNU;  // ConstructNoContactMotionTask: A = [J_c 0]
		} else {
			// base position rows (J_b rows 3..5), base rotation rows (0..2), swing feet
			const int br = (r < 3) ? 3 + r : r - 3;
			const double *jb = x.Jb ? x.Jb + (r < 6 ? br : 0) * kNU : x.Jf;
			const double *js = x.Jf + wbc_srow(x, r < 6 ? 0 : r - 6) * kNU;
			row = (r < 6) ? jb : js;
			rb = (r < 6 && !x.Jb) ? br : -1;
		}
#pragma unroll
		for (int kb = 0; kb < NKB; ++kb) {
			const int c = 4 * kb + q4;
			lv[kb] = row[(kb < 4 || c < kNU) ? c : 0];
		}
#pragma unroll
		for (int rr = 0; rr < 4; ++rr) {
			const int ri = r0 + (q4 + 4 * rr < nr ? q4 + 4 * rr : 0);
			if (LEVEL == 3)
				rl[rr] = l.b3[ri];
			else
				rl[rr] = x.Jdu[wbc_crow(x, ri)];  // b = -J_c_dot u
		}
	}
	__device__ __forceinline__ void take(double (&av)[NR / 4], double (&rvr)[4])
	{
		const int q4 = lane_id() >> 4;
#pragma unroll
		for (int kb = 0; kb < NKB; ++kb)
			pin(lv[kb]);
#pragma unroll
		for (int r = 0; r < 4; ++r)
			pin(rl[r]);
#pragma unroll
		for (int kb = 0; kb < NKB; ++kb) {
			const int c = 4 * kb + q4;
			double v = (kb < 4 || c < kNU) ? lv[kb] : 0.0;
			if (LEVEL == 3)
				v = (rb >= 0) ? ((c == rb) ? 1.0 : 0.0) : v;
			av[kb] = v;
		}
#pragma unroll
		for (int r = 0; r < 4; ++r)
			rvr[r] = (LEVEL == 3) ? rl[r] : 0.0 - rl[r];
	}
};

#ifdef HOQP_PROFILE
#define dbgp_proj (l.dbgp)
#endif
template <typename Src>
__device__ __forceinline__ void project_task_mfma(Src &src, int rows, int nz, const double (&bm)[NR / 4][2],
						  const ProjDst dstm, const Lds &l)
{
	const int lane = lane_id();
	const int j16 = lane & 15, q4 = lane >> 4;
	const bool two = nz + 1 > 16;  // second block of 16 output columns needed
#ifdef HOQP_PROFILE
	unsigned long long pt[4] = {0, 0, 0, 0};
	unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
	src.fetch(0, rows);
	PF_T(0);
	for (int r0 = 0; r0 < rows; r0 += CHUNK) {
		const int nr = min(CHUNK, rows - r0);
		double av[NR / 4], rvr[4];
		src.take(av, rvr);
		PF_T(1);
		if (r0 + CHUNK < rows)
			src.fetch(r0 + CHUNK, rows);
		PF_T(0);
		d4v acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
		if (two) {
#pragma unroll
			for (int kb = 0; kb < Src::NKB; ++kb) {
				acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kb], bm[kb][0], acc0, 0, 0, 0);
				acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kb], bm[kb][1], acc1, 0, 0, 0);
			}
		} else {
#pragma unroll
			for (int kb = 0; kb < Src::NKB; ++kb)
				acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kb], bm[kb][0], acc0, 0, 0, 0);
		}
		// D[i = q4 + 4 r][j = j16 (+16)]: one store per held entry (projected entry, right-hand
		// side, or the lane's sink)
#pragma unroll
		for (int cb = 0; cb < 2; ++cb) {
			if (cb == 1 && !two)
				continue;
			const int j = 16 * cb + j16;
			const bool isZ = j < nz, isX = j == nz;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int i = q4 + 4 * r;
				const bool rok = i < nr;
				const double v = cb ? acc1[r] : acc0[r];
				ldsd *dst = l.dump + lane;
				dst = (isZ && rok) ? dstm.M + (r0 + i) * dstm.ldm + j : dst;
				dst = (isX && rok) ? dstm.rhs + (r0 + i) : dst;
				*dst = isX ? rvr[r] - v : v;
			}
		}
		PF_T(2);
	}
	wave_sync();
#ifdef HOQP_PROFILE
	if (dbgp_proj && lane == 0)
		for (int i = 0; i < 3; ++i)
			dbgp_proj[39 + i] += pt[i];
#endif
}

// Level 0 of the fused entry (ConstructFloatingBaseEomTask, ho_qp_controller.cpp:351-366): Z = I and x = 0
// (InitPrevProblemValuesToDefault, ho_qp_problem.cpp:119-128), so the projected block is the block itself:
// A = [M_fb, -J_c,fb^T], r = b = -h_fb.  One column per lane, three rows each.
__device__ __forceinline__ void wbc_project_level0(const WbcCtx &wx, int ld, const Lds &l, ldsd *dstM, ldsd *dstr)
{
	const int lane = lane_id(), c = lane & 31, r2 = lane >> 5;
	const int jc0 = wbc_jcoff(wx, c);
	const bool ism = c < kNU;
	double v[3];
#pragma unroll
	for (int j = 0; j < 3; ++j) {
		const int r = r2 + 2 * j;
		const double *p = ism ? wx.M + r * kNU + c : wx.Jf + (jc0 >= 0 ? jc0 : 0) + r;
		v[j] = *p;
	}
	const double hv = wx.h[lane < kNB ? lane : 0];
#pragma unroll
	for (int j = 0; j < 3; ++j)
		pin(v[j]);
#pragma unroll
	for (int j = 0; j < 3; ++j) {
		const int r = r2 + 2 * j;
		st_if(c < wx.n, dstM + r * ld + c, l.dump + lane, ism ? v[j] : 0.0 - v[j]);
	}
	st_if(lane < kNB, dstr + lane, l.dump + lane, 0.0 - hv);
	wave_sync();
}

// Level 1 of the fused entry: ConstructJointTorqueTask (ho_qp_controller.cpp:332-349) ++
// ConstructFrictionConeTask (:299-330), 24 + 4 n_c inequality rows, of which only 12 are multiplied:
//   rows  0..11  [M_j, -J_c,j^T]              one 16-row tile on the matrix cores
//   rows 12..23  the negated rows 0..11        written as the negated results (exact)
//   rows 24..    friction pyramid, two non-zeros per row: +-Z[col + (t >> 1)] - mu Z[col + 2], straight from Z
// Same values as the generated rows of wbc_entry<1> pushed through project_task_mfma.
// VG: only the 12 torque rows and every right-hand side are stored; the other rows are virtual (see vrow).
template <bool VG = false>
__device__ __forceinline__ void wbc_project_level1(const WbcCtx &wx, const WbcPrm &prm, const int (&jc)[NR / 4], int nz, int ld,
						   const double (&bm)[NR / 4][2], const ProjDst dst, const Lds &l)
{
	const int lane = lane_id(), j16 = lane & 15, q4 = lane >> 4;
	const bool two = nz + 1 > 16;
	const int rr = j16 < kNJ ? j16 : 0;
	const double *mrow = wx.M + (kNB + rr) * kNU;
	const double *jrow = wx.Jf + kNB + rr;
	double lv[NR / 4], hv[3];
#pragma unroll
	for (int kb = 0; kb < NR / 4; ++kb) {
		const int c = 4 * kb + q4;
		const double *pj = jrow + (jc[kb] >= 0 ? jc[kb] : 0);
		const double *p = (kb < 4) ? mrow + c : (kb == 4 ? (c < kNU ? mrow + c : pj) : pj);
		lv[kb] = *p;
	}
#pragma unroll
	for (int r = 0; r < 3; ++r)
		hv[r] = wx.h[kNB + q4 + 4 * r];
	// friction rows: lane = (column j, half hf); contact i: rows 24 + 4 i + 2 hf + {0, 1}
	const int fj = lane & 31, hf = lane >> 5;
	const bool fz = fj < nz, fx = fj == nz;
	double fa[4], fc[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int ii = i < wx.nc ? i : 0;
		// Q1 (:321-324): the pyramid lands on columns 6 + 3 leg (joint accelerations)
		const int col = prm.friction_quirk ? kNB + wbc_pick(wx.clp, ii) * 3 : kNU + 3 * ii;
		const ldsd *pa = fz ? l.Z + (col + hf) * ld + fj : l.x + (col + hf);
		const ldsd *pc = fz ? l.Z + (col + 2) * ld + fj : l.x + (col + 2);
		fa[i] = *pa;
		fc[i] = *pc;
	}
#pragma unroll
	for (int kb = 0; kb < NR / 4; ++kb)
		pin(lv[kb]);
#pragma unroll
	for (int r = 0; r < 3; ++r)
		pin(hv[r]);
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		pin(fa[i]);
		pin(fc[i]);
	}
	double av[NR / 4];
#pragma unroll
	for (int kb = 0; kb < NR / 4; ++kb) {
		const int c = 4 * kb + q4;
		const double vj = (jc[kb] >= 0) ? 0.0 - lv[kb] : 0.0;
		av[kb] = (kb < 4) ? lv[kb] : (kb == 4 ? (c < kNU ? lv[kb] : vj) : vj);
	}
	d4v acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
	if (two) {
#pragma unroll
		for (int kb = 0; kb < NR / 4; ++kb) {
			acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kb], bm[kb][0], acc0, 0, 0, 0);
			acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kb], bm[kb][1], acc1, 0, 0, 0);
		}
	} else {
#pragma unroll
		for (int kb = 0; kb < NR / 4; ++kb)
			acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kb], bm[kb][0], acc0, 0, 0, 0);
	}
	// friction rows while the matrix cores work
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const bool on = i < wx.nc && (VG ? fx : (fz || fx));
		const double vp = fma(-prm.mu, fc[i], fa[i]), vm = fma(-prm.mu, fc[i], 0.0 - fa[i]);
		const int row = 2 * kNJ + 4 * i + 2 * hf;
		ldsd *d0 = fz ? dst.M + row * dst.ldm + fj : dst.rhs + row;
		ldsd *d1 = fz ? dst.M + (row + 1) * dst.ldm + fj : dst.rhs + row + 1;
		st_if(on, d0, l.dump + lane, fz ? vp : 0.0 - vp);
		st_if(on, d1, l.dump + lane, fz ? vm : 0.0 - vm);
	}
	// D[i = q4 + 4 r][j = j16 (+16)], r < 3: rows i and 12 + i
#pragma unroll
	for (int cb = 0; cb < 2; ++cb) {
		if (cb == 1 && !two)
			continue;
		const int j = 16 * cb + j16;
		const bool isZ = j < nz, isX = j == nz;
#pragma unroll
		for (int r = 0; r < 3; ++r) {
			const int i = q4 + 4 * r;
			const double v = cb ? acc1[r] : acc0[r];
			const double fu = fma(-1.0, hv[r], prm.tau_max);   // tau_max - h_j, :343
			const double fl = fma(1.0, hv[r], -prm.tau_min);  // -(tau_min - h_j), :344
			ldsd *du = isZ ? dst.M + i * dst.ldm + j : dst.rhs + i;
			ldsd *dl = isZ ? dst.M + (kNJ + i) * dst.ldm + j : dst.rhs + kNJ + i;
			st_if(isZ || isX, du, l.dump + lane, isX ? fu - v : v);
			st_if(VG ? isX : (isZ || isX), dl, l.dump + lane, isX ? fl + v : 0.0 - v);
		}
	}
	wave_sync();
}

#ifdef HOQP_PROFILE
#define HOQP_STAMP(slot)                                                              \
	do {                                                                          \
		unsigned long long now_ = __builtin_amdgcn_s_memtime();               \
		if (a.dbg && lane_id() == 0)                                          \
			a.dbg[(size_t)blockIdx.x * kDbgSlots + (slot)] += now_ - st.t_last;  \
		st.t_last = __builtin_amdgcn_s_memtime();                             \
	} while (0)
#else
#define HOQP_STAMP(slot) \
	do {             \
	} while (0)
#endif

#ifdef HOQP_PROFILE
#define HOQP_DBGP (a.dbg ? a.dbg + (size_t)blockIdx.x * kDbgSlots : nullptr)
#else
#define HOQP_DBGP nullptr
#endif

struct LevelState {
	int n, nz;
	int P;	   // live stored inequality rows
	int Pout;  // inequality rows emitted so far (offset of this level's slacks in out.v)
	int status;
	int replay_ok;	// every level so far can be replayed literally (see ref_replay)
	int bail;	// compact layout: the state does not fit (HOQP_STATUS_BASE_RANK), no further level is run
	unsigned long long t_last;
};

// number the quads' unpivoted columns of one slot: u = base + (rank of this quad among them)
__device__ __forceinline__ int unp_index(bool unp, int &base)
{
	const int lane = lane_id();
	const unsigned long long bm = __ballot(unp && (lane & 3) == 0);
	const int u = base + __popcll(bm & ((1ull << (lane & ~3)) - 1ull));
	base += __popcll(bm);
	return u;
}

// ---------------------------------------------------------------------------
// Equality-constrained least-squares step of the active-set method, factored from scratch:
//   min |B (y+p) - ...|  s.t.  G_W p = 0, minimum norm.  The panel holds the columns of
//   [ G_W' | B0' | G_V' ]  (w + m + |V| <= 64 columns, rows of V listed in l.vlist).
// Returns p_i in lane i; *k1_out = rank of the working set (reflectors [0,k1) stay in TV).
// ---------------------------------------------------------------------------
// VG: the rows of G are virtual (vrow; the fused reference kernel's level 1: no working set, no LS rows)
template <bool HARD, int VL = 0, bool VG = false>
__device__ __forceinline__ double eqp_step(const SolveArgs &a, const Lds &l, LevelState &st, int ld, int nz, int rank, int m,
				  int w_in, int nV, int *k1_out, bool *overflow, const VRowCtx &vc)
{
	const int w = HARD ? w_in : 0;  // no stored hard rows: the working set is always empty
	const int lane = lane_id(), s = lane & 3, c = lane >> 2;
	const int nb = m + nV;
	const int ncol = w + nb;
	*k1_out = 0;
	if (ncol > kWave) {
		*overflow = true;
		return 0.0;
	}
	const int nslot = (ncol + 15) >> 4;
	double x[NSP][NT];
	double xb[VG ? NSP : 1][NT];
	VRow vrs[VG ? NSP : 1];
	int pos[NSP], lims[NSP];
	wave_sync();
#pragma unroll
	for (int sl = 0; sl < NSP; ++sl) {
		pos[sl] = -2;
		lims[sl] = 0;
		if (sl >= nslot)
			continue;
		const int q = c + 16 * sl;
		const bool has = q < ncol;
		const bool isW = q < w, isL = q >= w && q < w + m;
		const int wq = l.W[isW ? q : 0];
		const int vq = l.vlist[(has && !isW && !isL) ? q - w - m : 0];
		const int grow = has ? (isW ? wq : (isL ? 0 : vq)) : 0;
		lims[sl] = has ? (isL ? rank : nz) : 0;
		if (VG) {
			vrs[sl] = vrow(vc, grow);
			if (HARD && isL) {  // a least-squares row of the level: stored (BL), taken as it is
				vrs[sl].p0 = l.BL + (q - w) * ld;
				vrs[sl].p1 = vrs[sl].p0;
				vrs[sl].s0 = 1.0;
				vrs[sl].s1 = 0.0;
			}
			ld8(x[sl], vrs[sl].p0 + s);
			ld8(xb[sl], vrs[sl].p1 + s);
		} else {
			const ldsd *src = isL ? l.BL + (q - w) * ld : l.G + grow * ld;
			ld8(x[sl], src + s);
		}
		pos[sl] = has ? -1 : -2;
	}
#pragma unroll
	for (int sl = 0; sl < NSP; ++sl) {
		if (sl >= nslot)
			continue;
		pin8(x[sl]);
		if (VG) {
			pin8(xb[sl]);
#pragma unroll
			for (int t = 0; t < NT; ++t)
				x[sl][t] = vval(vrs[sl], x[sl][t], xb[sl][t]);
		}
		t_upto(nz, s, [&](int t, bool) { x[sl][t] = sel(s + 4 * t < lims[sl], x[sl][t]); });
	}
	{
		// right-hand side and pivot map per column (lane == column)
		double rc = 0.0;
		if (lane >= w && lane < w + m)
			rc = -l.rho[lane - w];
		else if (lane >= w + m && lane < ncol)
			rc = -l.s[l.vlist[lane - w - m]];
		l.rhs[lane] = rc;
		l.colpos[lane] = -1;
	}
	wave_sync();
	const int k1 = HARD ? uni(panel_factor<VL>(x, pos, nslot, 0, w, nz, 0, 1e-12, l)) : 0;
	HOQP_STAMP(8);
	const int k2 = uni(panel_factor<VL>(x, pos, nslot, w, ncol, nz, k1, 1e-11, l));
	HOQP_STAMP(9);
	*k1_out = k1;
	// unpivoted B columns -> X
	int nunp = 0;
	if (k2 - k1 < nb) {
#pragma unroll
		for (int sl = 0; sl < NSP; ++sl) {
			if (sl >= nslot)
				continue;
			const int q = c + 16 * sl;
			const bool unp = pos[sl] == -1 && q >= w;
			const int u = unp_index(unp, nunp);
			if (unp && u < l.xrows) {
				ldsd *xr = l.X + u * ld;
				t_upto(nz, s, [&](int t, bool on) {
					if (on)
						xr[s + 4 * t] = x[sl][t];
				});
				if (s == 0)
					l.xlane[u] = q;
			}
		}
		nunp = uni(nunp);
		if (nunp > l.xrows) {
			*overflow = true;
			return 0.0;
		}
	}
	wave_sync();
#ifdef HOQP_PROFILE
	if (a.dbg && lane == 0) {
		a.dbg[(size_t)blockIdx.x * kDbgSlots + 14] += 1;           // EQP calls
		a.dbg[(size_t)blockIdx.x * kDbgSlots + 15] += (nunp > 0);  // ... that needed the semi-normal path
		a.dbg[(size_t)blockIdx.x * kDbgSlots + 23] += nunp;
		a.dbg[(size_t)blockIdx.x * kDbgSlots + 24] += (nunp > 4);
		a.dbg[(size_t)blockIdx.x * kDbgSlots + 25] += (nunp > 8);
		a.dbg[(size_t)blockIdx.x * kDbgSlots + 26] += (nunp > 0 && k2 - k1 + nunp > 32);
		a.dbg[(size_t)blockIdx.x * kDbgSlots + 51] += (nb <= 16);
		a.dbg[(size_t)blockIdx.x * kDbgSlots + 52] += (nb <= 16 && nunp == 0);
		a.dbg[(size_t)blockIdx.x * kDbgSlots + 53] += nb;
		a.dbg[(size_t)blockIdx.x * kDbgSlots + 54] += (nb <= 8);
	}
#endif
	double q = ls_solve_reg(k1, k2, w, ncol, nunp, 0.0, ld, nullptr, l);
	HOQP_STAMP(10);
	// u = [0 (k1) ; q ; 0],  p = Q u
	int srcl = lane - k1;
	double qq = __shfl(q, srcl < 0 ? 0 : srcl);
	double u = (srcl >= 0 && srcl < k2 - k1) ? qq : 0.0;
	const double pi = hh_apply_vec(u, nz, 0, k2, false, l);
	HOQP_STAMP(11);
	return pi;
}

// Equality-constrained step of an all-soft level (no least-squares rows, no working set) with at most 8
// violated rows V, through the small side: p = -G_V' (G_V G_V')^-1 s_V is the minimum-norm step that takes
// the rows of V to zero -- the same p as the factorization of the 8 columns of G_V' gives, at the price of
// an 8 x 8 Gram matrix (one entry per lane), an 8 x 8 Gauss-Jordan and one combination of the rows.
// Squaring the condition number is harmless while it is small: accepted only if the smallest pivot is above
// 1e-4 of the largest diagonal entry (kappa(G_V) < 100); otherwise (dependent or nearly dependent rows)
// the caller factors as usual.  Returns false in that case.
template <bool VG = false>
__device__ __forceinline__ bool eqp_dual8(const Lds &l, int nz, int nV, int ld, double *p_out, const VRowCtx &vc)
{
	const int lane = lane_id(), gi = lane >> 3, gj = lane & 7;
	const int ri = l.vlist[gi < nV ? gi : 0], rj = l.vlist[gj < nV ? gj : 0];
	const ldsd *ra = l.G + (VG ? 0 : ri) * ld, *rb = l.G + (VG ? 0 : rj) * ld;
	VRow va = {}, vb = {};
	if (VG) {
		va = vrow(vc, ri);
		vb = vrow(vc, rj);
	}
	double g0 = 0.0, g1 = 0.0;
	for (int k0 = 0; k0 < nz; k0 += 8) {
		double av[8], bv[8];
		if (VG) {
			double a1[8], b1[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const int kk = (k0 + u < nz) ? k0 + u : 0;
				av[u] = va.p0[kk];
				a1[u] = va.p1[kk];
				bv[u] = vb.p0[kk];
				b1[u] = vb.p1[kk];
			}
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				pin(av[u]);
				pin(a1[u]);
				pin(bv[u]);
				pin(b1[u]);
			}
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				av[u] = vval(va, av[u], a1[u]);
				bv[u] = vval(vb, bv[u], b1[u]);
			}
		} else {
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const int kk = (k0 + u < nz) ? k0 + u : 0;
				av[u] = ra[kk];
				bv[u] = rb[kk];
			}
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			pin(av[u]);
			pin(bv[u]);
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const double t = sel(k0 + u < nz, av[u]);
			if (u & 1)
				g1 = fma(t, bv[u], g1);
			else
				g0 = fma(t, bv[u], g0);
		}
	}
	const double gg = g0 + g1;
	const bool inb = gi < nV && gj < nV;
	const double dmax = wave_max((inb && gi == gj) ? gg : 0.0);
	const double mval = inb ? gg : ((gi == gj) ? 1.0 : 0.0);
	const double gval = (gi < nV) ? 0.0 - l.s[ri] : 0.0;
	double minpiv;
	const double lam = gauss_jordan8_piv(mval, gval, &minpiv);
	if (!(minpiv > 1e-4 * dmax && dmax > 0.0))
		return false;
	// p_k = sum_i G[v_i][k] lam_i, one coordinate per lane
	const int kk = lane < nz ? lane : 0;
	double gk[8];
	if (VG) {
		double g1[8];
		VRow vr[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			vr[i] = vrow(vc, __builtin_amdgcn_readlane(ri, 8 * i));
			gk[i] = vr[i].p0[kk];
			g1[i] = vr[i].p1[kk];
		}
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			pin(gk[i]);
			pin(g1[i]);
		}
#pragma unroll
		for (int i = 0; i < 8; ++i)
			gk[i] = vval(vr[i], gk[i], g1[i]);
	} else {
#pragma unroll
		for (int i = 0; i < 8; ++i)
			gk[i] = l.G[__builtin_amdgcn_readlane(ri, 8 * i) * ld + kk];
	}
#pragma unroll
	for (int i = 0; i < 8; ++i)
		pin(gk[i]);
	double p = 0.0;
#pragma unroll
	for (int i = 0; i < 8; ++i)
		p = (i < nV) ? fma(gk[i], readlane_d(lam, 8 * i), p) : p;
	*p_out = (lane < nz) ? p : 0.0;
	return true;
}

// ---------------------------------------------------------------------------
// The same step for up to 16 rows, on the matrix cores (fused reference kernel, level 1: every row is a virtual
// row).  Two things make the small side work where eqp_dual8 gives up:
//  * Structural dependence is removed first.  The four pyramid rows of one contact, [+-1, 0, -mu] and
//    [0, +-1, -mu] on three coordinates e = (a, b, c) of x, have orthogonal columns (B'B = diag(2, 2, 4 mu^2)); when
//    all four are violated their least-squares term sum_i (s_i + B_i e)^2 is minimised by the UNIQUE e* =
//    -(B'B)^-1 B's, so the four dependent rows are replaced by the three rows of Z that pick a, b, c, with targets
//    e* = (-(s1 - s2) / 2, -(s3 - s4) / 2, (s1 + s2 + s3 + s4) / (4 mu)).  The compacted rows G~ span the same
//    row space, and p = G~'(G~ G~')^-1 t is the minimum-norm minimiser of the same model -- what the
//    factorization path computes with its rank-deficient least squares (67 % of the factored steps).
//  * The Gram matrix is one 16 x 16 tile: six v_mfma_f64_16x16x4_f64 with the SAME register as both operands
//    (A[i][k] and B[k][j] = G~[j][k] sit in the same lane), then a Gauss-Jordan on the tile's own register layout
//    (K[i = lane / 16 + 4 r][j = lane % 16]): the pivot column comes by DPP row broadcast, the pivot row by one
//    cross-row permute.  Rows are scaled to unit norm first, so the acceptance test (smallest pivot above 1e-4)
//    measures conditioning, not the scale difference between torque rows and friction rows.
// Returns false (caller factors as usual) for more than 16 compacted rows or an ill-conditioned set.
// ---------------------------------------------------------------------------
__device__ __forceinline__ VRow crow(const VRowCtx &v, int code)
{
	// code < 64: row `code` of the level's block; 64 + 3 c + j: row of Z that picks coordinate j of contact c
	const int z = code >= 64 ? code - 64 : 0;
	const int c = (z * 11) >> 5, j = z - 3 * c;
	const int col = v.quirk ? kNB + wbc_pick(v.clp, c) * 3 : kNU + 3 * c;
	VRow o = vrow(v, code < 64 ? code : 0);
	if (code >= 64) {
		o.p0 = v.Z + (col + j) * v.ld;
		o.p1 = o.p0;
		o.s0 = 1.0;
		o.s1 = 0.0;
	}
	return o;
}

template <int K>
__device__ __forceinline__ void gj16_steps(double (&M)[4], double (&gv)[4], int nR, double &minpiv)
{
	if constexpr (K < 16) {
		if (K < nR) {
			const int lane = lane_id(), j16 = lane & 15, q4 = lane >> 4;
			constexpr int kr = K >> 2, kq = K & 3;
			const double piv = readlane_d(M[kr], kq * 16 + K);
			minpiv = fmin(minpiv, piv);
			const double rpv = fast_rcp(piv);
			const double rowk = __shfl(M[kr], kq * 16 + j16) * rpv;  // M[K][j] / piv
			const double gk = readlane_d(gv[kr], kq * 16) * rpv;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const double cik = dpp_d<0x150 + K>(M[r]);  // row_newbcast: M[i][K], i = q4 + 4 r
				const bool isk = (r == kr) && (q4 == kq);
				M[r] = isk ? rowk : fma(-cik, rowk, M[r]);
				gv[r] = isk ? gk : fma(-cik, gk, gv[r]);
			}
			gj16_steps<K + 1>(M, gv, nR, minpiv);
		}
	}
}

__device__ __forceinline__ bool eqp_dual16(const SolveArgs &a, LevelState &st, const Lds &l, const VRowCtx &vc, int nz,
					   unsigned long long vmask, double si, double *p_out)
{
	(void)a;
	(void)st;
	HOQP_STAMP(13);
	const int lane = lane_id(), j16 = lane & 15, q4 = lane >> 4;
	// contacts whose four pyramid rows are all violated
	unsigned long long vm1 = vmask;
	int ng = 0;
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const bool full = c < vc.nc && ((vmask >> (2 * kNJ + 4 * c)) & 0xFull) == 0xFull;
		vm1 = full ? (vm1 & ~(0xFull << (2 * kNJ + 4 * c))) : vm1;
		ng += full;
	}
	const int n1 = __popcll(vm1), nR = n1 + 3 * ng;
	if (nR < 1 || nR > 16 || nR > nz)
		return false;
	wave_sync();
	if ((vm1 >> lane) & 1ull) {
		const int idx = __popcll(vm1 & ((1ull << lane) - 1ull));
		l.vlist[idx] = lane;
		l.rhs[idx] = 0.0 - si;
	}
	{
		int g = 0;
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			const bool full = c < vc.nc && ((vmask >> (2 * kNJ + 4 * c)) & 0xFull) == 0xFull;
			const int r0 = 2 * kNJ + 4 * c;
			if (full) {
				const double s1 = readlane_d(si, r0), s2 = readlane_d(si, r0 + 1), s3 = readlane_d(si, r0 + 2),
					     s4 = readlane_d(si, r0 + 3);
				if (lane < 3) {
					const double t = (lane == 0) ? -0.5 * (s1 - s2)
								     : ((lane == 1) ? -0.5 * (s3 - s4) : ((s1 + s2) + (s3 + s4)) / (4.0 * vc.mu));
					l.vlist[n1 + 3 * g + lane] = 64 + 3 * c + lane;
					l.rhs[n1 + 3 * g + lane] = t;
				}
				++g;
			}
		}
	}
	wave_sync();
	HOQP_STAMP(8);
	// operand: G~[i = j16][k = 4 kb + q4]
	const bool rowon = j16 < nR;
	const VRow vr = crow(vc, l.vlist[rowon ? j16 : 0]);
	double a0[6], a1[6];
#pragma unroll
	for (int kb = 0; kb < 6; ++kb) {
		const int k = 4 * kb + q4;
		a0[kb] = vr.p0[k < nz ? k : 0];
		a1[kb] = vr.p1[k < nz ? k : 0];
	}
	double t_i[4];
#pragma unroll
	for (int r = 0; r < 4; ++r)
		t_i[r] = l.rhs[(q4 + 4 * r < nR) ? q4 + 4 * r : 0];
#pragma unroll
	for (int kb = 0; kb < 6; ++kb) {
		pin(a0[kb]);
		pin(a1[kb]);
	}
	d4v acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
	for (int kb = 0; kb < 6; ++kb) {
		const double v = (rowon && 4 * kb + q4 < nz) ? vval(vr, a0[kb], a1[kb]) : 0.0;
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
	}
	// Jacobi scaling: d_j = K[j][j] sits in lane (j & 3) * 16 + j, register j >> 2
	double M[4], gv[4];
	double dj = 0.0;
#pragma unroll
	for (int r = 0; r < 4; ++r) {
		M[r] = acc[r];
		dj = (q4 + 4 * r == j16) ? acc[r] : dj;  // my own diagonal entry, if I hold one
	}
	dj += __shfl_xor(dj, 16);
	dj += __shfl_xor(dj, 32);  // every lane: d_(j16)
	const bool colon = j16 < nR;
	if (__ballot(colon && !(dj > 1e-290)) != 0ull)
		return false;  // a zero row
	const double rsj = colon ? fast_rsqrt(dj) : 1.0;
#pragma unroll
	for (int r = 0; r < 4; ++r) {
		const int i = q4 + 4 * r;
		const double rsi = __shfl(rsj, q4 * 16 + i);  // (lane q4 * 16 + i of my row holds column i's scale)
		const bool on = i < nR && colon;
		M[r] = on ? M[r] * rsi * rsj : ((i == j16) ? 1.0 : 0.0);
		gv[r] = (i < nR) ? t_i[r] * rsi : 0.0;
	}
	double minpiv = 1e300;
	HOQP_STAMP(9);
	gj16_steps<0>(M, gv, nR, minpiv);
	HOQP_STAMP(10);
	if (!(minpiv > 1e-4))
		return false;
	// lambda_i = gv (scaled system) * rs_i;  p = G~' lambda.  The operand registers still hold G~[i = j16][k = 4 kb + q4]:
	// scale each by lambda_(j16), sum over the 16 lanes of the row (DPP), and hand coordinate k = 4 kb + q4 to
	// lane k -- no second trip through LDS.
	double lamsrc = gv[0];
#pragma unroll
	for (int r = 1; r < 4; ++r)
		lamsrc = ((j16 >> 2) == r) ? gv[r] : lamsrc;  // lane (i & 3) * 16 + i holds lambda~_i here
	const double lam = rowon ? __shfl(lamsrc, (j16 & 3) * 16 + j16) * rsj : 0.0;
	double red[6];
#pragma unroll
	for (int kb = 0; kb < 6; ++kb) {
		double v = (rowon && 4 * kb + q4 < nz) ? vval(vr, a0[kb], a1[kb]) * lam : 0.0;
		v += dpp_d<0xB1>(v);
		v += dpp_d<0x4E>(v);
		v += dpp_d<0x141>(v);
		v += dpp_d<0x140>(v);
		red[kb] = v;  // p_(4 kb + q4) in every lane of row q4
	}
	double psrc = red[0];
#pragma unroll
	for (int kb = 1; kb < 6; ++kb)
		psrc = (j16 == kb) ? red[kb] : psrc;  // lane q4 * 16 + kb offers p_(4 kb + q4)
	const double p = __shfl(psrc, (lane & 3) * 16 + ((lane >> 2) < 6 ? (lane >> 2) : 0));
	*p_out = (lane < nz) ? p : 0.0;
	HOQP_STAMP(11);
	return true;
}

// Level update on the rows of Z and G (row r = c + 16*slot):  d_r = row_r . y  -> l.S[r], then
// the rows lose their first `rank` coordinates (shifted store).  y is in l.tmp.
template <int NS>
__device__ __forceinline__ void level_update_t(const Lds &l, int n, int Pg, int nz, int rank, int ld,
					       unsigned long long *dbgp = nullptr)
{
	const int lane = lane_id(), s = lane & 3, c = lane >> 2;
#ifdef HOQP_PROFILE
	unsigned long long pt[4] = {0, 0, 0, 0};
	unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
	const int R = n + Pg;
	double yv[NT];
	ld8(yv, l.tmp + s);
	double x[NS][NT];
	ldsd *rowp[NS];
	bool valid[NS];
	// every load of every slot precedes the first shifted store (single wavefront, LDS in order)
#pragma unroll
	for (int sl = 0; sl < NS; ++sl) {
		const int r = c + 16 * sl;
		valid[sl] = r < R;
		const int rr = valid[sl] ? r : 0;
		rowp[sl] = (rr < n) ? l.Z + rr * ld : l.G + (rr - n) * ld;
		ld8(x[sl], rowp[sl] + s);
	}
	PF_T(0);
	pin8(yv);
#pragma unroll
	for (int sl = 0; sl < NS; ++sl)
		pin8(x[sl]);
	PF_T(1);
	t_upto(nz, s, [&](int t, bool on) { yv[t] = sel(on, yv[t]); });
#pragma unroll
	for (int sl = 0; sl < NS; ++sl) {
		double d0 = 0.0, d1 = 0.0;
		t_upto(nz, s, [&](int t, bool on) {
			x[sl][t] = sel(on, x[sl][t]);
			if (t & 1)
				d1 = fma(x[sl][t], yv[t], d1);
			else
				d0 = fma(x[sl][t], yv[t], d0);
		});
		const double d = quad_sum(d0 + d1);
		if (valid[sl] && s == 0)
			l.S[c + 16 * sl] = d;
	}
	PF_T(2);
	if (rank > 0 && rank < nz) {
#pragma unroll
		for (int sl = 0; sl < NS; ++sl) {
			t_from(rank, nz, s, [&](int t, bool on, bool) {
				st_if(on && valid[sl] && s + 4 * t < nz, rowp[sl] + (s + 4 * t - rank), l.dump + lane, x[sl][t]);
			});
		}
	}
	PF_T(3);
#ifdef HOQP_PROFILE
	if (dbgp && lane == 0)
		for (int i = 0; i < 4; ++i)
			dbgp[28 + i] += pt[i];
#endif
}

// ---------------------------------------------------------------------------
// Reference replay.  The reference regularises every level in the coordinates of ITS null-space
// basis -- min |A Z z - r|^2 + eps |z|^2 with Z the FullPivLU kernel chain (ho_qp_problem.cpp:
// 172-178, 244-246, math.hpp:99-104) -- and |z| in those coordinates is the norm of a SUBSET of
// the entries of the step x_k - x_{k-1} (the LU's free columns), not its Euclidean norm.  The fast
// path keeps orthonormal bases and reproduces the eps bias to first order in the Euclidean
// metric; the difference is ~ (eps / sigma_min^2) * O(1..10) relative, invisible (<= 1e-7) on a
// well-conditioned level and above the 1e-6 parity tolerance once sigma_min(A Z) < ~3e-3 (a leg
// near a singular posture at the tracking level of the WBC).  When an unconstrained level's
// smallest pivot says so (eps > kReplayTau * dmin^2), the levels 0..k are recomputed literally as
// the reference does them -- complete-pivoting LU kernels in Eigen's pivot order, regularised
// least squares by Householder QR of the stacked system (never forming H) -- and x_k is replaced
// by the result; the orthonormal chain goes on unchanged (the two bases span the same space).
// Slow and simple on purpose (one LDS access at a time): it runs for ~0.01 % of stance states and
// ~10 % of three-contact states.  It needs the inequality workspace G, the reflector store and the
// staging buffer as scratch, so it only runs when no later level looks at stored rows; levels that
// solved a QP with equality rows cannot be replayed (the fast path's answer stands).
// ---------------------------------------------------------------------------
// Complete-pivoting LU of M (r x c, stride c) in place, Eigen's FullPivLU: the pivot is the largest
// |coefficient| of the remaining corner, among equal ones the first in column-major order.  Returns
// the rank (pivots above maxpivot * eps_mach * min(r, c)); cperm[j] = original column at position j.
__device__ __forceinline__ int lds_fullpivlu(ldsd *M, int r, int c, ldsi *cperm)
{
	const int lane = lane_id();
	if (lane < c)
		cperm[lane] = lane;
	wave_sync();
	const int size = min(r, c);
	double maxpivot = 0.0;
	int nonzero = size;
	constexpr int EPL = 8;  // corner entries per lane: 8 * 64 >= 18 * 27 of the WBC; larger corners loop
	// e / w for 0 <= e < 2^15, 1 <= w <= 32 without an integer division: (e + 0.5) / w is at least 0.5 / w
	// away from an integer, far more than the rounding of the float product
	auto idiv = [](int e, float rw) { return (int)(((float)e + 0.5f) * rw); };
	for (int k = 0; k < size; ++k) {
		// ---- pivot search over the corner [k, r) x [k, c), entries spread over all lanes so that the
		//      loads of a lane are independent and in flight together ----
		const int cw = c - k, ne = (r - k) * cw;
		const float rcw = 1.0f / (float)cw;
		double best = -1.0;
		int bkey = 0x7fffffff;  // column-major position j * 64 + i of the lane's best entry
		for (int e0 = 0; e0 < ne; e0 += EPL * kWave) {
			double v[EPL];
			int key[EPL];
#pragma unroll
			for (int t = 0; t < EPL; ++t) {
				const int e = e0 + lane + kWave * t;
				const int ec = (e < ne) ? e : 0;
				const int qi = idiv(ec, rcw);
				const int i = k + qi, j = k + ec - qi * cw;
				v[t] = M[i * c + j];
				key[t] = (e < ne) ? j * 64 + i : 0x7fffffff;
			}
#pragma unroll
			for (int t = 0; t < EPL; ++t) {
				const double av = fabs(v[t]);
				const bool valid = key[t] != 0x7fffffff;
				const bool take = valid && (av > best || (av == best && key[t] < bkey));
				best = take ? av : best;
				bkey = take ? key[t] : bkey;
			}
		}
		const double mx = wave_max(best);
		if (!(mx > 0.0)) {
			nonzero = k;
			break;
		}
		// among the lanes holding the maximum: the first entry in column-major order (one lane, usually)
		const unsigned long long cand = __ballot(best == mx);
		int pk;
		if (__popcll(cand) == 1)
			pk = __builtin_amdgcn_readlane(bkey, __ffsll((long long)cand) - 1);
		else
			pk = (int)wave_min((best == mx) ? (double)bkey : 1e18);
		pk = uni(pk);
		const int bj = pk >> 6, bi = pk & 63;
		maxpivot = fmax(maxpivot, mx);
		wave_sync();
		if (bi != k && lane < c) {
			const double t0 = M[k * c + lane], t1 = M[bi * c + lane];
			M[k * c + lane] = t1;
			M[bi * c + lane] = t0;
		}
		wave_sync();
		if (bj != k) {
			if (lane < r) {
				const double t0 = M[lane * c + k], t1 = M[lane * c + bj];
				M[lane * c + k] = t1;
				M[lane * c + bj] = t0;
			}
			if (lane == 0) {
				const int t0 = cperm[k], t1 = cperm[bj];
				cperm[k] = t1;
				cperm[bj] = t0;
			}
		}
		wave_sync();
		const double piv = M[k * c + k];
		if (lane > k && lane < r)
			M[lane * c + k] = M[lane * c + k] / piv;
		wave_sync();
		// ---- rank-1 update of the corner (k, r) x (k, c), again spread over all lanes ----
		const int cw1 = c - k - 1, ne1 = (r - k - 1) * cw1;
		const float rcw1 = 1.0f / (float)(cw1 > 0 ? cw1 : 1);
		for (int e0 = 0; e0 < ne1; e0 += EPL * kWave) {
			double mij[EPL], lik[EPL], ukj[EPL];
			int adr[EPL];
#pragma unroll
			for (int t = 0; t < EPL; ++t) {
				const int e = e0 + lane + kWave * t;
				const int ec = (e < ne1) ? e : 0;
				const int qi = idiv(ec, rcw1);
				const int i = k + 1 + qi, j = k + 1 + ec - qi * cw1;
				adr[t] = (e < ne1) ? i * c + j : -1;
				mij[t] = M[i * c + j];
				lik[t] = M[i * c + k];
				ukj[t] = M[k * c + j];
			}
#pragma unroll
			for (int t = 0; t < EPL; ++t)
				if (adr[t] >= 0)
					M[adr[t]] = fma(-lik[t], ukj[t], mij[t]);
		}
		wave_sync();
	}
	const double thr = maxpivot * (kEpsMach * (double)size);
	const bool sig = lane < nonzero && fabs(M[(lane < nonzero ? lane : 0) * c + (lane < nonzero ? lane : 0)]) > thr;
	return __popcll(__ballot(sig));
}

// The same factorization with the matrix in registers: one column per lane (c <= 64), its r <= R entries in
// col[0..R).  Columns are not moved; each lane tracks the POSITION of its column (pos), which is what Eigen's
// physical swaps change, and rows are swapped in the registers of every lane.  Same pivot rule (largest
// |coefficient| of the corner, ties: lowest position, then lowest row), same arithmetic per entry (true
// division for the multipliers, one FMA per update) as lds_fullpivlu; ~4 x fewer cycles per step, because the
// pivot search is R compares per lane and one wave maximum instead of passes over LDS.  On return M holds U
// in position order as lds_fullpivlu leaves it (the strictly lower part is NOT written: nobody reads L).
template <int R>
__device__ __forceinline__ int reg_fullpivlu(ldsd *M, int r, int c, ldsi *cperm)
{
	const int lane = lane_id();
	const bool isc = lane < c;
	double col[R];
	{
		const ldsd *src = M + (isc ? lane : 0);
#pragma unroll
		for (int i = 0; i < R; ++i)
			col[i] = src[(i < r ? i : 0) * c];
#pragma unroll
		for (int i = 0; i < R; ++i)
			col[i] = (isc && i < r) ? col[i] : 0.0;
	}
	int pos = lane;
	const int size = min(r, c);
	double maxpivot = 0.0;
	int nonzero = size;
#pragma unroll
	for (int k = 0; k < R; ++k) {
		if (k < size && nonzero == size) {  // (wave-uniform)
			double best = -1.0;
			int bi = k;
#pragma unroll
			for (int i = k; i < R; ++i) {
				const double av = fabs(col[i]);
				const bool take = i < r && av > best;
				best = take ? av : best;
				bi = take ? i : bi;
			}
			const bool cand = isc && pos >= k;
			const double mx = wave_max(cand ? best : -1.0);
			if (!(mx > 0.0)) {
				nonzero = k;
			} else {
				// among the columns holding the maximum: the lowest position
				const unsigned long long tie = __ballot(cand && best == mx);
				int pl = __ffsll((long long)tie) - 1;
				if (__popcll(tie) > 1) {
					const int pmin = (int)wave_min((cand && best == mx) ? (double)pos : 1e18);
					pl = __ffsll((long long)__ballot(cand && best == mx && pos == pmin)) - 1;
				}
				pl = uni(pl);
				const int prow = __builtin_amdgcn_readlane(bi, pl);
				const int oldpos = __builtin_amdgcn_readlane(pos, pl);
				maxpivot = fmax(maxpivot, mx);
				// row swap k <-> prow (every lane), column swap k <-> oldpos (positions only)
#pragma unroll
				for (int i = k + 1; i < R; ++i) {
					const bool sw = i == prow;
					const double t = col[i];
					col[i] = sw ? col[k] : col[i];
					col[k] = sw ? t : col[k];
				}
				pos = (pos == k) ? oldpos : ((lane == pl) ? k : pos);
				const double piv = readlane_d(col[k], pl);
#pragma unroll
				for (int i = k + 1; i < R; ++i) {
					const double li = readlane_d(col[i] / piv, pl);
					col[i] = (pos > k && i < r) ? fma(-li, col[k], col[i]) : col[i];
				}
			}
		}
	}
	// U in position order (rows i <= position), the column permutation, the rank
	wave_sync();
	if (isc) {
		cperm[pos] = lane;
#pragma unroll
		for (int i = 0; i < R; ++i)
			if (i < r && i <= pos)
				M[i * c + pos] = col[i];
	}
	double ukk = 0.0;
#pragma unroll
	for (int i = 0; i < R; ++i)
		ukk = (pos == i) ? col[i] : ukk;
	const double thr = maxpivot * (kEpsMach * (double)size);
	const bool sig = isc && pos < nonzero && fabs(ukk) > thr;
	wave_sync();
	return __popcll(__ballot(sig));
}

// K (c x dk, stride dk, dk = c - rank) = FullPivLU::kernel() from the factors above:
// rows of the pivot columns = -U11^-1 U12, rows of the free columns = identity.
__device__ __forceinline__ void lds_lu_kernel(const ldsd *M, int c, int rank, const ldsi *cperm, ldsd *K)
{
	const int lane = lane_id();
	const int dk = c - rank;
	// rows in pivot POSITION order first (X = U11^-1 U12 by back substitution, one kernel column per
	// lane, the loads of a row's terms in flight together), scattered to the original rows at the end
	if (rank <= 12) {
		// (as in lds_basis_update: the lane's column of X in registers, a row of U11 read as a block)
		double xc[12];
#pragma unroll
		for (int i = 0; i < 12; ++i)
			xc[i] = 0.0;
		const int cl = rank + (lane < dk ? lane : 0);
#pragma unroll
		for (int i = 11; i >= 0; --i) {
			if (i >= rank)
				continue;
			double u[12];
#pragma unroll
			for (int j = 0; j < 12; ++j)
				u[j] = M[i * c + ((j >= i && j < rank) ? j : i)];
			double sacc = M[i * c + cl];
#pragma unroll
			for (int j = 0; j < 12; ++j)
				pin(u[j]);
#pragma unroll
			for (int j = 0; j < 12; ++j)
				sacc = (j > i && j < rank) ? fma(-u[j], xc[j], sacc) : sacc;
			xc[i] = sacc / u[i];
		}
#pragma unroll
		for (int i = 0; i < 12; ++i)
			if (i < rank && lane < dk)
				K[i * dk + lane] = xc[i];
	} else if (lane < dk) {
		for (int i = rank - 1; i >= 0; --i) {
			double sacc = M[i * c + rank + lane];
			for (int j0 = i + 1; j0 < rank; j0 += 4) {
				double u[4], xv[4];
#pragma unroll
				for (int t = 0; t < 4; ++t) {
					const int j = (j0 + t < rank) ? j0 + t : i + 1;
					u[t] = M[i * c + j];
					xv[t] = K[j * dk + lane];
				}
#pragma unroll
				for (int t = 0; t < 4; ++t)
					sacc = (j0 + t < rank) ? fma(-u[t], xv[t], sacc) : sacc;
			}
			K[i * dk + lane] = sacc / M[i * c + i];
		}
	}
	wave_sync();
	// K currently holds X in rows [0, rank); build the kernel in place: row cperm[i] = -X_i, row of the
	// q-th free column = e_q.  Rows are permuted, so go through registers: every lane reads its entries
	// of X first, then all lanes write.
	{
		const int total = c * dk;  // entries of the final K
		constexpr int PASS = 16;   // 16 * 64 >= 30 * 24
		double val[PASS];
		int dst[PASS];
#pragma unroll
		for (int t = 0; t < PASS; ++t) {
			const int e = lane + kWave * t;
			const bool on = e < total;
			const int ec = on ? e : 0;
			const int pos = (int)(((float)ec + 0.5f) * (1.0f / (float)dk)), q = ec - pos * dk;  // position, kernel column
			const int orig = cperm[pos < c ? pos : 0];
			double v;
			if (pos < rank)
				v = -K[pos * dk + q];
			else
				v = (pos - rank == q) ? 1.0 : 0.0;
			val[t] = v;
			dst[t] = on ? orig * dk + q : -1;
		}
		wave_sync();
#pragma unroll
		for (int t = 0; t < PASS; ++t)
			if (dst[t] >= 0)
				K[dst[t]] = val[t];
	}
	wave_sync();
}

// Z (n x nzr, stride nzr) <- Z K in place (n x dk, stride dk), K = FullPivLU::kernel() of the factors in LU
// (r x nzr, stride nzr; rank pivots): rows of the pivot columns = -U11^-1 U12, rows of the free columns = identity.
// X = U11^-1 U12 overwrites U12; K itself is never formed: Z K is summed over t = 0 .. nzr-1 in natural order
// with K[t][q] looked up through the inverse column permutation -- the same terms in the same order as the
// explicit product.  One row of Z per lane (n <= 64), the row's dk <= MAXQ results in registers until every
// lane has read its row.  ipos: scratch for nzr ints.
template <int MAXQ>
__device__ __forceinline__ void lds_basis_update(ldsd *Z, int n, int nzr, int rank, const ldsi *cperm, ldsd *LU, ldsi *ipos)
{
	const int lane = lane_id();
	const int dk = nzr - rank;
	if (lane < nzr)
		ipos[cperm[lane]] = lane;
	if (rank <= 12) {
		// one column of X per lane, held in registers; row i of U11 is read as a block (uniform addresses,
		// the loads of a row in flight together) -- the same terms in the same order as the loop below
		double xc[12];
#pragma unroll
		for (int i = 0; i < 12; ++i)
			xc[i] = 0.0;
		const int cl = rank + (lane < dk ? lane : 0);
#pragma unroll
		for (int i = 11; i >= 0; --i) {
			if (i >= rank)
				continue;
			double u[12];
#pragma unroll
			for (int j = 0; j < 12; ++j)
				u[j] = LU[i * nzr + ((j >= i && j < rank) ? j : i)];
			double sacc = LU[i * nzr + cl];
#pragma unroll
			for (int j = 0; j < 12; ++j)
				pin(u[j]);
#pragma unroll
			for (int j = 0; j < 12; ++j)
				sacc = (j > i && j < rank) ? fma(-u[j], xc[j], sacc) : sacc;
			xc[i] = sacc / u[i];
		}
		wave_sync();
#pragma unroll
		for (int i = 0; i < 12; ++i)
			if (i < rank && lane < dk)
				LU[i * nzr + rank + lane] = xc[i];
	} else if (lane < dk) {
		for (int i = rank - 1; i >= 0; --i) {
			double sacc = LU[i * nzr + rank + lane];
			for (int j = i + 1; j < rank; ++j)
				sacc = fma(-LU[i * nzr + j], LU[j * nzr + rank + lane], sacc);
			LU[i * nzr + rank + lane] = sacc / LU[i * nzr + i];
		}
	}
	wave_sync();
	double out[MAXQ];
#pragma unroll
	for (int q = 0; q < MAXQ; ++q)
		out[q] = 0.0;
	const ldsd *zrow = Z + (lane < n ? lane : 0) * nzr;
	for (int t = 0; t < nzr; ++t) {
		const double zt = zrow[t];
		const int pt = uni(ipos[t]);
		if (pt < rank) {
			const ldsd *xr = LU + pt * nzr + rank;
			double kq[MAXQ];
#pragma unroll
			for (int q = 0; q < MAXQ; ++q)
				kq[q] = xr[q < dk ? q : 0];
#pragma unroll
			for (int q = 0; q < MAXQ; ++q)
				out[q] = fma(zt, 0.0 - kq[q], out[q]);
		} else {
#pragma unroll
			for (int q = 0; q < MAXQ; ++q)
				out[q] = fma(zt, (pt - rank == q) ? 1.0 : 0.0, out[q]);
		}
	}
	wave_sync();  // every lane has read its row
	if (lane < n) {
#pragma unroll
		for (int q = 0; q < MAXQ; ++q)
			if (q < dk)
				Z[lane * dk + q] = out[q];
	}
	wave_sync();
}

constexpr int kRegC = 12;  // columns of the stacked system held in registers (WBC: <= 12)
// Householder steps K.. of the lane-per-row QR (recursion = full unrolling with the step index a
// compile-time constant, so that the dot products of one step are reduced together, wave_sum_n)
template <int K, int RC>
__device__ __forceinline__ void reg_qr_steps(double (&w)[RC + 1], int C, int R)
{
	if constexpr (K < RC) {
		if (K < C) {
			const int lane = lane_id();
			const bool on = lane >= K && lane < R;
			double v = on ? w[K] : 0.0;
			const double alpha = readlane_d(v, K);
			const double nrm = sqrt(wave_sum(v * v));
			if (nrm > 0.0) {
				const double beta = (alpha >= 0.0) ? -nrm : nrm;
				v = (lane == K) ? alpha - beta : v;
				const double tk = 1.0 / (nrm * (nrm + fabs(alpha)));
				double d[RC - K];  // columns K+1 .. RC-1 and the right-hand side (slot RC)
#pragma unroll
				for (int j = K + 1; j <= RC; ++j)
					d[j - K - 1] = v * w[j];
				wave_sum_n(d);
#pragma unroll
				for (int j = K + 1; j <= RC; ++j)
					w[j] = fma(-d[j - K - 1] * tk, v, w[j]);
				w[K] = (lane == K) ? beta : w[K];
			}
			reg_qr_steps<K + 1, RC>(w, C, R);
		}
	}
}

// zout (nz) = argmin |M z - r|^2 + eps |z|^2, M (m x nz, stride nz) and r in LDS, by Householder QR,
// one row per lane, the row held in registers (C = min(m, nz) <= RC columns plus the right-hand side;
// the column loops are unrolled over RC with wave-uniform guards):
//   m > nz:  the stacked system [M; sqrt(eps) I] z = [r; 0].
//   m <= nz: z lies in range(M') = range(Q) of the QR M' = Q R, so z = Q y = M' R^-1 y with
//            y = argmin |R' y - r|^2 + eps |y|^2, again a stacked system [R'; sqrt(eps) I] y = [r; 0] (2m x m).
//            (Round 2 solved [M'; sqrt(eps) I] w = [0; r / sqrt(eps)], z = M' w: the same minimiser on paper, but a
//            least-squares problem whose residual is ~ |r| / sqrt(eps) = 1e6 |r|, and the error of a QR solve
//            grows with kappa^2 x residual: 2e-5 relative at sigma_min(M) = 1e-6.  Both stacked systems here
//            have residuals of the size of r: errors ~ u kappa.)
// T: scratch for the upper triangle of an m x m matrix, packed by rows (tri_at; m (m + 1) / 2 doubles; m <= nz case).
__device__ __forceinline__ int tri_at(int i, int j, int m) { return i * m - ((i * (i - 1)) >> 1) + (j - i); }  // j >= i

template <int RC>
__device__ __forceinline__ void lds_reg_solve(const ldsd *M, const ldsd *r, int m, int nz, double eps, ldsd *wv, ldsd *zout,
					      ldsd *T)
{
	const int lane = lane_id();
	const bool dual = m <= nz;
	const int C = dual ? m : nz;
	const double se = sqrt(eps);
	double w[RC + 1];
	if (dual) {
		// ---- R of M' = Q R: lane q < nz holds row q of M' ----
		{
			double ld_[RC];
#pragma unroll
			for (int j = 0; j < RC; ++j)
				ld_[j] = M[((j < C) ? j : 0) * nz + (lane < nz ? lane : 0)];
#pragma unroll
			for (int j = 0; j < RC; ++j)
				w[j] = (j < C && lane < nz) ? ld_[j] : 0.0;
			w[RC] = 0.0;
		}
		reg_qr_steps<0, RC>(w, C, nz);
		// T(i, j) = R(i, j), j >= i: lane i < m holds row i of R in w[i..m)
#pragma unroll
		for (int j = 0; j < RC; ++j)
			if (j < C && lane < m && j >= lane)
				T[tri_at(lane, j, m)] = w[j];
		wave_sync();
		// ---- stacked system [R'; sqrt(eps) I] y = [r; 0]: lane i < m holds row i of R' ----
		{
			double ld_[RC + 1];
#pragma unroll
			for (int j = 0; j < RC; ++j)  // R'(lane, j) = R(j, lane), zero for j > lane
				ld_[j] = T[(j < C && lane < m && j <= lane) ? tri_at(j, lane, m) : 0];
			ld_[RC] = r[lane < m ? lane : 0];
#pragma unroll
			for (int j = 0; j < RC; ++j)
				w[j] = (j < C) ? ((lane < m) ? ((j <= lane) ? ld_[j] : 0.0) : ((lane - m == j) ? se : 0.0)) : 0.0;
			w[RC] = (lane < m) ? ld_[RC] : 0.0;
		}
		reg_qr_steps<0, RC>(w, C, 2 * m);
	} else {
		// the lane's row: all loads first (clamped addresses), then the selects
		double ld_[RC + 1];
#pragma unroll
		for (int j = 0; j <= RC; ++j)
			ld_[j] = (j < C) ? M[(lane < m ? lane : 0) * nz + ((j < C) ? j : 0)] : r[lane < m ? lane : 0];
#pragma unroll
		for (int j = 0; j < RC; ++j)
			w[j] = (j < C) ? ((lane < m) ? ld_[j] : ((lane - m == j && lane < m + nz) ? se : 0.0)) : 0.0;
		w[RC] = (lane < m) ? ld_[RC] : 0.0;
		reg_qr_steps<0, RC>(w, C, m + nz);
	}
	// back substitution: lane i holds row i of the triangular factor and the transformed right-hand side
	double acc = w[RC];
#pragma unroll
	for (int i = RC - 1; i >= 0; --i) {
		if (i >= C)
			continue;
		const double si = readlane_d(acc / w[i], i);
		acc = fma(-w[i], si, acc);
		if (lane == 0)
			wv[i] = si;
	}
	wave_sync();
	if (dual) {
		// w = R^-1 y (R in T, one row per lane), then z = M' w
		double acc2 = wv[lane < m ? lane : 0];
		const int lr = lane < m ? lane : 0;
		const double rdiag = T[tri_at(lr, lr, m)];
		for (int i = m - 1; i >= 0; --i) {
			const double si = readlane_d(acc2 / rdiag, i);
			acc2 = fma(-((i >= lr) ? T[tri_at(lr, i >= lr ? i : lr, m)] : 0.0), si, acc2);
			wave_sync();
			if (lane == 0)
				wv[i] = si;
		}
		wave_sync();
	}
	if (lane < nz) {
		double z = 0.0;
		if (dual) {
			for (int i = 0; i < m; ++i)
				z = fma(M[i * nz + lane], wv[i], z);
		} else {
			z = wv[lane];
		}
		zout[lane] = z;
	}
	wave_sync();
}

// Returns true (and replaces l.x by the reference's x_k) if levels 0..k could be replayed.
template <bool WBC, int MAXQ = NR>
__device__ __forceinline__ bool ref_replay(const SolveArgs &a, const Lds &l, int inst, int k, int n, int nz_entry,
					   const WbcCtx &wx, const int (&jc)[NR / 4], double *res2_out)
{
	const int lane = lane_id();
	// scratch: ZB holds the chain of bases, MK the projected block M and its kernel matrix K (lds_bind)
	ldsd *ZB = l.rpZ, *MK = l.rpM;
	const int Gcap = l.rpZcap, tvcap = l.rpMcap;
	ldsd *xr = l.tmp, *zv = l.tmp + NR, *wv = l.tau;
#ifdef HOQP_REPLAY_DEBUG
#define RDBG(...) do { if (lane == 0) printf(__VA_ARGS__); } while (0)
#else
#define RDBG(...) do { } while (0)
#endif
	RDBG("[replay] inst %d k %d n %d nz_entry %d Gcap %d tvcap %d\n", inst, k, n, nz_entry, Gcap, tvcap);
#ifdef HOQP_PROFILE
	unsigned long long rt_ = __builtin_amdgcn_s_memtime();
#define RSTAMP(slot)                                                                          \
	do {                                                                                  \
		const unsigned long long n_ = __builtin_amdgcn_s_memtime();                   \
		if (a.dbg && lane == 0)                                                       \
			a.dbg[(size_t)blockIdx.x * kDbgSlots + (slot)] += n_ - rt_;           \
		rt_ = __builtin_amdgcn_s_memtime();                                           \
	} while (0)
	if (a.dbg && lane == 0)
		a.dbg[(size_t)blockIdx.x * kDbgSlots + 32] += 1;
#else
#define RSTAMP(slot) \
	do {         \
	} while (0)
#endif
	if (n >= NR)
		return false;
	wave_sync();
	// x_{k-1}: the fast path's own (its earlier levels were either well conditioned -- the eps term is below
	// 3e-8 of the step there, whatever the basis -- or replayed themselves); the level update has just added
	// the step l.S to it.  Only the BASIS of the earlier levels has to be redone in the reference's
	// coordinates, not their solves.
	if (lane < NR)
		xr[lane] = (lane < n) ? l.x[lane] - l.S[lane] : 0.0;
	wave_sync();
	const ldsd *Zr = ZB;
	bool zident = true;  // Zr is the identity so far (no null tests on LDS pointers: offset 0 is a valid address)
	int nzr = n;
	for (int j = 0; j <= k; ++j) {
		int mj, pj;
		if (WBC) {
			mj = wbc_m(wx, j);
			pj = wbc_p(wx, j);
		} else {
			mj = min(max(uni(a.m_inst ? a.m_inst[(size_t)inst * a.L + j] : a.m[j]), 0), a.m[j]);
			pj = min(max(uni(a.p_inst ? a.p_inst[(size_t)inst * a.L + j] : a.p[j]), 0), a.p[j]);
		}
		if (nzr == 0)
			break;
		if (pj > 0 || mj == 0)	// an inequality-only level leaves the null space as it is
			continue;
		if (j == k && nzr != nz_entry)
			return false;  // the LU's rank decisions differ from the QR's: leave the fast answer
		const int cmin = min(mj, nzr);
		RDBG("[replay]  level %d m %d nzr %d\n", j, mj, nzr);
		// MK: M (mj x nzr; for j < k its LU factors, then U11^-1 U12 inside them) and, at j == k, the m x m factor of
		// lds_reg_solve.  The bases live in ZB: from the identity the kernel matrix IS the new basis, afterwards
		// the basis is updated in place (lds_basis_update).
		if (mj * nzr + (j < k ? 0 : (cmin * (cmin + 1)) / 2) > tvcap || cmin > kRegC || mj + nzr > kWave || nzr - cmin > MAXQ)
			return false;
		if (WBC && j == 0 && zident && j < k) {
			// from the identity the projected block is the block itself (the right-hand side is not used: j < k)
			wbc_project_level0(wx, nzr, l, MK, l.rhs);
		} else {
			// M = A_j Zr -> TV (stride nzr),  r = b_j - A_j xr -> l.rhs
			double bm[NR / 4][2];
			load_bm_from(bm, n, nzr, nzr, Zr, xr, zident);
			const ProjDst dst = {MK, nzr, l.rhs};
			if (WBC) {
				// levels 2 and 3: the row sources of the fast path (rows of J_foot / J_b, zero beyond column 17)
				if (j == 0) {
					WbcEntrySrc<0> src(wx, a.wprm, l, jc);
					project_task_mfma(src, mj, nzr, bm, dst, l);
				} else if (j == 2) {
					WbcLevelSrc<2> src(wx, l);
					project_task_mfma(src, mj, nzr, bm, dst, l);
				} else if (j == 3) {
					WbcLevelSrc<3> src(wx, l);
					project_task_mfma(src, mj, nzr, bm, dst, l);
				} else {
					WbcEntrySrc<4> src(wx, a.wprm, l, jc);
					project_task_mfma(src, mj, nzr, bm, dst, l);
				}
			} else {
				TaskRowsSrc src = {a.A[j] + (size_t)inst * a.m[j] * a.n, a.b[j] + (size_t)inst * a.m[j], a.n, n};
				project_task_mfma(src, mj, nzr, bm, dst, l);
			}
		}
		RSTAMP(33);
		if (j == k) {
		if (cmin <= 6)  // (the tracking level of a stance state: half the columns, half the reductions)
			lds_reg_solve<6>(MK, l.rhs, mj, nzr, a.eps, wv, zv, MK + mj * nzr);
		else
			lds_reg_solve<kRegC>(MK, l.rhs, mj, nzr, a.eps, wv, zv, MK + mj * nzr);
		RSTAMP(34);
		{  // |A_k x_k - b_k|^2 at the regularised optimum
			double ri = 0.0;
			if (lane < mj) {
				ri = l.rhs[lane];
				for (int q = 0; q < nzr; ++q)
					ri = fma(-MK[lane * nzr + q], zv[q], ri);
			}
			*res2_out = wave_sum(ri * ri);
		}
		if (lane < n) {
			double sx = xr[lane];
			if (zident) {
				sx += zv[lane];
			} else {
				for (int q0 = 0; q0 < nzr; q0 += 8) {
					double za[8], zb[8];
#pragma unroll
					for (int u = 0; u < 8; ++u) {
						const int q = (q0 + u < nzr) ? q0 + u : 0;
						za[u] = Zr[lane * nzr + q];
						zb[u] = zv[q];
					}
#pragma unroll
					for (int u = 0; u < 8; ++u)
						sx = (q0 + u < nzr) ? fma(za[u], zb[u], sx) : sx;
				}
			}
			xr[lane] = sx;
		}
		wave_sync();
		RSTAMP(35);
		}
		if (j < k) {
			// (blocks of up to 12 rows -- every level of the WBC hierarchy a later level can replay through
			// except a tracking level with swing legs -- are factored in registers)
			const int rank = uni(mj <= 6 ? reg_fullpivlu<6>(MK, mj, nzr, l.colpos)
						      : (mj <= 12 ? reg_fullpivlu<12>(MK, mj, nzr, l.colpos)
								  : lds_fullpivlu(MK, mj, nzr, l.colpos)));
			RSTAMP(36);
			const int dk = nzr - rank;
			RDBG("[replay]  level %d rank %d\n", j, rank);
			if (dk == 0) {
				nzr = 0;
				continue;
			}
			if (n * dk > Gcap || dk > MAXQ)
				return false;
			if (zident) {
				lds_lu_kernel(MK, nzr, rank, l.colpos, ZB);  // Zn = I K (nzr == n)
				RSTAMP(37);
			} else {
				RSTAMP(37);
				if (MAXQ > 12 && dk <= 12)  // (the WBC: 12 free coordinates are left behind the contact level)
					lds_basis_update<12>(ZB, n, nzr, rank, l.colpos, MK, l.pivlane);
				else
					lds_basis_update<MAXQ>(ZB, n, nzr, rank, l.colpos, MK, l.pivlane);
			}
			wave_sync();
			RSTAMP(38);
			Zr = ZB;
			zident = false;
			nzr = dk;
		}
	}
	wave_sync();
	RDBG("[replay] done inst %d\n", inst);
	if (lane < n)
		l.x[lane] = xr[lane];
	wave_sync();
	return true;
}

// ---------------------------------------------------------------------------
// one priority level (HoQpProblem constructor pipeline, ho_qp_problem.cpp:11-25)
// ---------------------------------------------------------------------------
// GATE: Q2 of the reference (levels without inequalities of their own ignore the stored rows,
// ho_qp_problem.cpp:297-298) as a compile-time switch: the gated kernel carries none of the code
// that only ungated hierarchies reach.
// HARD: some level of the hierarchy looks at rows stored by an earlier level (false for a single
// inequality level under GATE, i.e. the reference's WBC hierarchy: no hard rows, no working set,
// no phase 1, no multipliers -- all of that code is compiled out).
// WBC: the task rows come from the generator of wbc_tasks.hpp (fused entry) instead of a.A / a.D.
// KC: the level index as a compile-time constant (-1: run-time, k_rt).  The fused WBC kernel of the reference
// semantics runs its five levels as five specialised copies: with the level known, each copy carries only its
// own projection and only the solver paths its shape can reach (the QP code exists in level 1 alone), and the
// level loop has no far backward jump (measured: ~2 k cycles of instruction fetch per back-edge of the
// 380 KB loop body).
template <bool GATE, bool HARD, bool WBC, int KC = -1>
__device__ __forceinline__ void solve_level(const SolveArgs &a, const Lds &l_in, LevelState &st, int inst, int k_rt, int m, int p,
					    const WbcCtx &wx, const int (&jc)[NR / 4])
{
	const int k = (KC >= 0) ? KC : k_rt;
	HOQP_STAMP(45);
	const int lane = lane_id(), s = lane & 3, c = lane >> 2;
	// the specialised level copies of the fused reference kernel run on the compact LDS layout (hoqp_solve.hpp
	// head): compile-time leading dimensions; ldt: the projected rows of this level in TV
	constexpr bool COMPACT = WBC && GATE && !HARD && KC >= 0;
	// CP: the fused WBC entry under paper semantics (run-time level loop, hard rows): the same compact strides and
	// virtual inequality rows; CV: either of the two
	constexpr bool CP = WBC && !GATE && HARD;
	constexpr bool CV = COMPACT || CP;
	const int n = uni(st.n), ld = CV ? kCLd : a.ld;
	const int ldt = (CV && k == 0) ? kCLd0 : ld;
	Lds l = l_in;
	if (CV) {
		l.ldv = (k == 0) ? LDV : (CP ? kPLdv : kCLdv);
		l.xrows = CP ? kPXrows : kCXrows;
	}
	// the virtual rows of level 1 (compact layout; unused otherwise)
	const VRowCtx vc = {l.G, l.Z, ld, wx.clp, wx.nc, a.wprm.friction_quirk, a.wprm.mu};
	const int nz = uni(st.nz), P = HARD ? uni(st.P) : 0;
	const int nstore = a.n;
	if (WBC && KC >= 0) {
		// a specialised copy: m and p are expressions of the (uniform) contact counts, and the shape facts
		// of the WBC hierarchy are told to the compiler so that each copy keeps only the slot counts and
		// solver paths it can reach (n = 18 + 3 n_c; level 0 starts from Z = I)
		__builtin_assume(n >= kNU && n <= kNX);
		__builtin_assume(nz >= 0 && nz <= n);
		if (KC == 0)
			__builtin_assume(nz == n);
	} else {
		m = uni(m);
		p = uni(p);
	}
	const int Pn = P + p;
	const bool consider = (nz > 0) && (GATE ? (p > 0) : (p > 0 || P > 0));
	// stored rows are still needed by a later level?  (else they are neither rotated nor kept)
	const bool keep_rows = k < a.last_ineq_use;
	const int Pg = (consider || keep_rows) ? Pn : 0;  // rows that take part in rotation / update
	int iters = 0;
	double level_res2 = 0.0;
	double dmin_level = 1e300;  // smallest pivot of an unconstrained solve (conditioning estimate)

	int rank = 0;
	bool updated = false;	 // the level update already happened in the rotation pass
	double yi = 0.0;	 // lane i holds y_i (rotated coordinates)
	double res_const = 0.0;	 // part of the level residual that no y can change (tall levels)
	int mB = m;		 // LS rows seen by the active-set loop
	// A level with more equality rows than free coordinates is first tried "tall": QR of A Z
	// itself; with full column rank that is the whole answer (unique y, empty null space).
	// A rank-deficient tall level falls back to the general path (second pass of the loop).
	// (fused WBC: level 0 is 6 x n and level 2 is 3 n_c x (n_z >= 12 + 3 n_c): never tall)
	bool tall = (WBC && (KC == 0 || KC == 2)) ? false : (m > nz && nz > 0);
	// A level without inequality rows of its own that has to respect stored rows (paper
	// semantics) first tries its unconstrained optimum: if no stored row objects, that is the
	// answer and the active-set loop is skipped.
	const bool tryfast = !GATE && consider && p == 0;
	bool have_y0 = false;
	// ---- 1. project the task into the current null space (ConstructH/C/D/FVector,
	//         ho_qp_problem.cpp:180-285, in factored form) ----
	auto project = [&](bool doA, bool doD) {
		const ProjDst dstT = {l.TV, ldt, l.rhs}, dstG = {l.G + P * ld, ld, l.g + P};
		if (WBC && k == 4) {
			// ConstructForceMinimizationTask: A = [0 I], b = 0, so A Z is rows 18.. of Z and A x the
			// contact forces -- copied, not multiplied (the products are exact either way)
			if (m > 0 && doA) {
#ifdef HOQP_PROFILE
				const unsigned long long tb0_ = __builtin_amdgcn_s_memtime();
#endif
				for (int idx = lane; idx < m * nz; idx += kWave) {
					const int r = idx / nz, j = idx - r * nz;
					l.TV[r * ldt + j] = l.Z[(kNU + r) * ld + j];
				}
				if (lane < m)
					l.rhs[lane] = 0.0 - l.x[kNU + lane];
				wave_sync();
#ifdef HOQP_PROFILE
				if (l.dbgp && lane == 0)
					l.dbgp[43] += __builtin_amdgcn_s_memtime() - tb0_;
#endif
			}
		} else if (WBC && k == 0) {
			if (m > 0 && doA)
				wbc_project_level0(wx, ldt, l, l.TV, l.rhs);
		} else if (WBC) {  // fused entry (n <= 30: always the matrix-core path)
			double bm[NR / 4][2];
#ifdef HOQP_PROFILE
			const unsigned long long tb0_ = __builtin_amdgcn_s_memtime();
#endif
			if (k == 1) {
				if (p > 0 && doD) {
					load_bm(bm, n, nz, ld, l);
					wbc_project_level1<CV>(wx, a.wprm, jc, nz, ld, bm, dstG, l);
				}
			} else if (m > 0 && doA) {
				load_bm<kWbcKB>(bm, n, nz, ld, l);
				if (k == 2) {
					WbcLevelSrc<2> src(wx, l);
					project_task_mfma(src, m, nz, bm, dstT, l);
				} else {  // k == 3 (level 4 is handled above)
					WbcLevelSrc<3> src(wx, l);
					project_task_mfma(src, m, nz, bm, dstT, l);
				}
			}
#ifdef HOQP_PROFILE
			if (l.dbgp && lane == 0)
				l.dbgp[42] += __builtin_amdgcn_s_memtime() - tb0_;
#endif
		} else if (nz < 32) {  // matrix-core path
			double bm[NR / 4][2];
			load_bm(bm, n, nz, ld, l);
			if (m > 0 && doA) {
				TaskRowsSrc src = {a.A[k] + (size_t)inst * a.m[k] * nstore, a.b[k] + (size_t)inst * a.m[k], nstore, n};
				project_task_mfma(src, m, nz, bm, dstT, l);
			}
			if (p > 0 && doD) {
				TaskRowsSrc src = {a.D[k] + (size_t)inst * a.p[k] * nstore, a.f[k] + (size_t)inst * a.p[k], nstore, n};
				project_task_mfma(src, p, nz, bm, dstG, l);
			}
		} else {
			double zc[NR];
			const ProjLane pl = proj_lane(nz);
			const ldsd *src = pl.isZ ? l.Z + pl.j : l.x;
			const int stride = pl.isZ ? ld : 1;
			const bool live = pl.isZ || pl.isX;
#pragma unroll
			for (int i = 0; i < NR; ++i)
				zc[i] = src[(i < n ? i : 0) * stride];
#pragma unroll
			for (int i = 0; i < NR; ++i)
				pin(zc[i]);
#pragma unroll
			for (int i = 0; i < NR; ++i)
				zc[i] = sel(live && i < n, zc[i]);
			if (m > 0 && doA)
				project_task(a.A[k] + (size_t)inst * a.m[k] * nstore, a.b[k] + (size_t)inst * a.m[k], m,
					     nstore, n, nz, ld, zc, true, 0, l);
			if (p > 0 && doD)
				project_task(a.D[k] + (size_t)inst * a.p[k] * nstore, a.f[k] + (size_t)inst * a.p[k], p,
					     nstore, n, nz, ld, zc, false, P, l);
		}
	};
	HOQP_STAMP(44);
	project(true, true);
	HOQP_STAMP(1);
	const bool factorA = m > 0 && nz > 0;
	if (!factorA && m > 0) {
		// no free directions left: the residual is what it is
		const double r = sel(lane < m, l.rhs[lane]);
		level_res2 = wave_sum(r * r);
	}
	bool ne_done = false;
	if (factorA && tall && !consider && nz <= 8) {
		// ---- 2n. a tall level with at most 8 free coordinates and nothing to respect (the force
		//          level of the WBC: 12 x 6): the normal equations ((AZ)'AZ + eps I) y = (AZ)'r -- in
		//          an orthonormal basis literally the reference's system (ho_qp_problem.cpp:238-252) --
		//          by an 8 x 8 Gauss-Jordan, one matrix entry per lane, instead of n_z Householder
		//          steps.  Squaring the condition number is harmless while it is small: accepted only if
		//          the smallest pivot is above 1e-4 of the largest diagonal entry (kappa(AZ) < 100).
		const int gi = lane >> 3, gj = lane & 7;
		const int ci = gi < nz ? gi : 0, cj = gj < nz ? gj : 0;
		double g0 = 0.0, c0 = 0.0;
		for (int r0 = 0; r0 < m; r0 += 4) {
			double ai[4], aj[4], rr[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const int r = (r0 + u < m) ? r0 + u : 0;
				ai[u] = l.TV[r * ldt + ci];
				aj[u] = l.TV[r * ldt + cj];
				rr[u] = l.rhs[r];
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const bool on = r0 + u < m;
				g0 = on ? fma(ai[u], aj[u], g0) : g0;
				c0 = on ? fma(ai[u], rr[u], c0) : c0;
			}
		}
		const bool inb = gi < nz && gj < nz;
		const double dmax = wave_max((inb && gi == gj) ? g0 : 0.0);
		const double mval = inb ? g0 + ((gi == gj) ? a.eps : 0.0) : ((gi == gj) ? 1.0 : 0.0);
		const double gval = (gi < nz) ? c0 : 0.0;
		double minpiv;
		const double av = gauss_jordan8_piv(mval, gval, &minpiv);
		if (minpiv > 1e-4 * dmax && dmax > 0.0) {
			ne_done = true;
			rank = nz;
			yi = sel(lane < nz, __shfl(av, (8 * lane) & (kWave - 1)));
			// residual of the level at its optimum
			const int rc = lane < m ? lane : 0;
			double tv[8];
#pragma unroll
			for (int i = 0; i < 8; ++i)
				tv[i] = l.TV[rc * ldt + (i < nz ? i : 0)];
			double res = l.rhs[rc];
#pragma unroll
			for (int i = 0; i < 8; ++i)
				res = (i < nz) ? fma(-tv[i], readlane_d(av, 8 * i), res) : res;
			level_res2 = wave_sum(lane < m ? res * res : 0.0);
			dmin_level = sqrt(minpiv);
			have_y0 = true;
			HOQP_STAMP(2);
		}
	}
	if (factorA && tall && !ne_done) {
		double x[NSP][NT];
		int pos[NSP];
		l.colpos[lane] = -1;
		{
			// ---- 2t. QR with column pivoting of A Z (m x nz): panel column j < nz is
			//          coordinate j over the task rows, column nz carries the right-hand side ----
			const int nslot = (nz + 1 + 15) >> 4;
#pragma unroll
			for (int sl = 0; sl < NSP; ++sl) {
				pos[sl] = -2;
				if (sl >= nslot)
					continue;
				const int q = c + 16 * sl;
				const bool has = q <= nz;
				const ldsd *src = (q < nz) ? l.TV + q : l.rhs;
				const int stride = (q < nz) ? ldt : 1;
#pragma unroll
				for (int t = 0; t < NT; ++t)
					x[sl][t] = src[(s + 4 * t) * stride];
				pos[sl] = has ? -1 : -2;
			}
#pragma unroll
			for (int sl = 0; sl < NSP; ++sl) {
				if (sl >= nslot)
					continue;
				pin8(x[sl]);
				t_upto(m, s, [&](int t, bool on) { x[sl][t] = sel(on && pos[sl] == -1, x[sl][t]); });
			}
			wave_sync();
			const int rT = uni(panel_factor<(COMPACT ? 1 : (CP ? 3 : 0))>(x, pos, nslot, 0, nz, m, 0, kEpsMach * (double)max(m, nz), l, HOQP_DBGP));
			HOQP_STAMP(2);
			if (rT < nz) {
				tall = false;  // rank-deficient: project again, then the general path
			} else {
			// transformed right-hand side: c1 = entries [0, nz), the rest is the residual
			if (c == (nz & 15)) {
#pragma unroll
				for (int sl = 0; sl < NSP; ++sl) {
					if (sl != (nz >> 4))
						continue;
#pragma unroll
					for (int t = 0; t < NT; ++t)
						l.tmp[s + 4 * t] = x[sl][t];
				}
			}
			wave_sync();
			const double cl = sel(lane < NR, l.tmp[lane & (NR - 1)]);
			res_const = wave_sum((lane >= nz && lane < m) ? cl * cl : 0.0);
			rank = nz;
			if (consider) {
				// the active-set loop sees the nz compressed rows T Pi' (current coordinates):
				// BL[p][j] = T(p, pos_j), r0[p] = c1[p]
				mB = nz;
#pragma unroll
				for (int sl = 0; sl < 2; ++sl) {
					if (16 * sl >= nz)
						continue;
					const int pr = c + 16 * sl;
					const bool hasr = pr < nz;
					int pj[NT];
#pragma unroll
					for (int t = 0; t < NT; ++t)
						pj[t] = l.colpos[(s + 4 * t < nz) ? s + 4 * t : 0];
					double dg = l.rdiag[hasr ? pr : 0];
					double e[NT];
#pragma unroll
					for (int t = 0; t < NT; ++t)
						e[t] = l.TV[(pj[t] > 0 ? pj[t] : 0) * l.ldv + (hasr ? pr : 0)];
#pragma unroll
					for (int t = 0; t < NT; ++t)
						pin(e[t]);
					pin(dg);
					ldsd *row = l.BL + (hasr ? pr : 0) * ld;
					t_upto(nz, s, [&](int t, bool on) {
						const double v = (pj[t] > pr) ? e[t] : ((pj[t] == pr) ? dg : 0.0);
						if (on && hasr)
							row[s + 4 * t] = v;
					});
				}
				if (lane < nz)
					l.r0[lane] = cl;
			}
			if (!consider || tryfast) {
				// ---- 4t. T z = c1, y = Pi z (+ eps term: z <- z - eps (T'T)^-1 z) ----
				const bool act = lane < nz;
				double inv = fast_rcp(l.rdiag[act ? lane : 0]);
				inv = sel(act, inv);
				double Urow[NR];
				load_Urow(Urow, 0, nz, l);
				double z = tri_bwd(Urow, inv, sel(act, cl), nz);
				const double dmin = wave_min(act ? fabs(l.rdiag[lane]) : 1e300);
				dmin_level = dmin;
				if (a.eps != 0.0 && a.eps > 1e-10 * dmin * dmin) {  // see ls_solve_reg
					double Lrow[NR];
					load_Lrow(Lrow, 0, nz, l);
					const double wv = tri_fwd(Lrow, inv, z, nz);
					z -= a.eps * tri_bwd(Urow, inv, wv, nz);
				}
				wave_sync();
				l.tmp[lane] = z;
				wave_sync();
				yi = sel(act, l.tmp[l.colpos[act ? lane : 0] & (kWave - 1)]);
				level_res2 = res_const;
				have_y0 = true;
			}
			wave_sync();
			HOQP_STAMP(3);
			}
		}
		if (!tall)
			project(true, false);
	}
	if (factorA && !tall) {
		double x[NSP][NT];
		int pos[NSP];
		l.colpos[lane] = -1;
		// ---- 2. complete orthogonal decomposition of A Z (CalcNullSpaceProjMatrix, math.hpp:99-104,
		//         with an orthonormal Householder basis instead of the FullPivLU kernel) ----
		const int nslot = (m + 15) >> 4;
#pragma unroll
		for (int sl = 0; sl < NSP; ++sl) {
			pos[sl] = -2;
			if (sl >= nslot)
				continue;
			const int q = c + 16 * sl;
			const bool has = q < m;
			const ldsd *src = l.TV + (has ? q : 0) * ldt;
			ld8(x[sl], src + s);
			pos[sl] = has ? -1 : -2;
		}
#pragma unroll
		for (int sl = 0; sl < NSP; ++sl) {
			if (sl >= nslot)
				continue;
			pin8(x[sl]);
			t_upto(nz, s, [&](int t, bool on) { x[sl][t] = sel(on && pos[sl] == -1, x[sl][t]); });
		}
		wave_sync();  // every column is in registers before TV is reused as the reflector store
		rank = uni(panel_factor<(COMPACT ? 1 : (CP ? 2 : 0))>(x, pos, nslot, 0, m, nz, 0, kEpsMach * (double)max(m, nz), l, HOQP_DBGP));
		HOQP_STAMP(2);
		if (CV && k == 0 && rank != kNB) {
			// [M_fb, -J_c,fb^T] without full row rank (not a physical mass matrix): more than 24 free
			// coordinates would be left, which the compact layout has no room for
			st.status |= HOQP_STATUS_BASE_RANK;
			st.bail = 1;
			return;
		}
		int nunp = 0;
		int uidx[NSP];
#pragma unroll
		for (int sl = 0; sl < NSP; ++sl) {
			uidx[sl] = 0;
			if (sl >= nslot || rank == m)
				continue;
			uidx[sl] = unp_index(pos[sl] == -1, nunp);
		}
		nunp = uni(nunp);
		if (consider) {
			// LS rows in rotated coordinates for the active-set loop: BL[j][0..rank), r0[j];
			// row order: pivot positions first, then the unpivoted (dependent) rows
#pragma unroll
			for (int sl = 0; sl < NSP; ++sl) {
				if (sl >= nslot)
					continue;
				const int q = c + 16 * sl;
				const bool hasc = pos[sl] != -2;
				const bool unp = pos[sl] == -1;
				const int mp = pos[sl];
				const int j = hasc ? (unp ? rank + uidx[sl] : mp) : 0;
				ldsd *row = l.BL + j * ld;
				const double dg = l.rdiag[mp > 0 ? mp : 0];
				const double rq = l.rhs[q];
				t_upto(rank, s, [&](int t, bool on) {
					const int i = s + 4 * t;
					const double v = unp ? x[sl][t] : (i < mp ? x[sl][t] : (i == mp ? dg : 0.0));
					if (on && hasc)
						row[i] = v;
				});
				if (s == 0 && hasc)
					l.r0[j] = rq;
			}
		}
		if ((!consider || tryfast) && rank > 0) {
			// ---- 4a. unconstrained level: y1 = argmin |R' y1 - r| (+ eps term), y2 = 0 ----
			if (nunp > l.xrows) {
				if (!consider)
					st.status |= HOQP_STATUS_WORKSPACE;
			} else {
				if (nunp > 0) {
#pragma unroll
					for (int sl = 0; sl < NSP; ++sl) {
						if (sl >= nslot)
							continue;
						const bool unp = pos[sl] == -1;
						ldsd *xr = l.X + (unp ? uidx[sl] : 0) * ld;
						t_upto(nz, s, [&](int t, bool on) {
							if (on && unp)
								xr[s + 4 * t] = x[sl][t];
						});
						if (s == 0 && unp)
							l.xlane[uidx[sl]] = c + 16 * sl;
					}
				}
				wave_sync();
				yi = ls_solve_reg(0, rank, 0, m, nunp, a.eps, ld, &level_res2, l, &dmin_level);
				have_y0 = true;
			}
		} else if (!consider) {
			const double r = sel(lane < m, l.rhs[lane]);
			level_res2 = wave_sum(r * r);
		}
		wave_sync();
		HOQP_STAMP(3);
		// ---- 3. rotate Z and every stored inequality row: Z <- Z Q, G <- G Q ----
		if (WBC && GATE && !HARD && KC >= 0 && KC != 1) {
			// a specialised unconstrained level (no stored rows, y final): rotation and level update in one
			// pass over the rows of Z; at level 0 Z = I has not been written (kernel main) and the rotation
			// generates its rows
			if (rank > 0) {
				wave_sync();
				l.tmp[lane] = (lane < nz) ? yi : 0.0;
				wave_sync();
				if (KC == 0)
					rows_apply_reflectors_t<2, true, true>(l.Z, n, l.G, 0, ld, nz, 0, rank, l);
				else if (nz <= 12)
					rows_apply_reflectors_t<2, false, true, 3>(l.Z, n, l.G, 0, ld, nz, 0, rank, l);
				else if (nz <= 24)
					rows_apply_reflectors_t<2, false, true, 6>(l.Z, n, l.G, 0, ld, nz, 0, rank, l);
				else
					rows_apply_reflectors_t<2, false, true>(l.Z, n, l.G, 0, ld, nz, 0, rank, l);
				updated = true;
			} else if (KC == 0) {
				lds_set_identity(l.Z, n, ld);
				wave_sync();
			}
		} else if (CP && k == 0) {
			// (as the specialised level 0 above: Z = I is neither stored nor read -- a row of the compact layout
			// has no room for 30 columns; rank == kNB here, the other case has bailed out)
			wave_sync();
			l.tmp[lane] = (lane < nz) ? yi : 0.0;
			wave_sync();
			rows_apply_reflectors_t<2, true, true>(l.Z, n, l.G, 0, ld, nz, 0, rank, l);
			updated = true;
		} else if (CP) {
			// the 12 stored rows beside the rows of Z: at most 42 rows = three row slots, vectors of at most 24
			if (rank > 0)
				rows_apply_reflectors_t<3, false, false, 6>(l.Z, n, l.G, min(Pg, kCGrows), ld, nz, 0, rank, l);
		} else if (rank > 0)
			rows_apply_reflectors(l.Z, n, l.G, Pg, ld, nz, 0, rank, l);
		HOQP_STAMP(4);
	}

	// Q3 (ho_qp_problem.cpp:83-101, 224-225): the reference stacks the prior rows newest level
	// first but their optimal slacks oldest level first and adds the two element-wise.  Stored
	// row r (oldest first here) of a level block [off, off + pj) sits at position
	// P - off - pj + (r - off) of the reference's row stack and is therefore relaxed by the slack
	// of stored row number <that position>.  delta = that slack - the row's own.
	const bool q3 = HARD && a.slack_quirk && consider && P > 0;
	auto q3_delta = [&]() {
		int off = 0, myoff = 0, myp = 0;
		for (int j = 0; j < k; ++j) {
			const int pj = WBC ? wbc_p(wx, j) : uni(a.p_inst ? a.p_inst[(size_t)inst * a.L + j] : a.p[j]);
			if (lane >= off && lane < off + pj) {
				myoff = off;
				myp = pj;
			}
			off += pj;
		}
		const int src = P - myoff - myp + (lane - myoff);
		const bool on = lane < P;
		const double vsrc = l.vs[on ? src : 0], vown = l.vs[on ? lane : 0];
		return on ? vsrc - vown : 0.0;
	};
	if (q3) {
		const double dq = q3_delta();
		if (lane < P)
			l.g[lane] += dq;
		wave_sync();
	}
	bool skip_qp = false;
	double y0i = 0.0;
	bool y0_ok = false;
	if (tryfast && have_y0) {
		// does any stored row object to the unconstrained optimum?
		wave_sync();
		l.tmp[lane] = (lane < nz) ? yi : 0.0;
		wave_sync();
		if (CP)
			rows_dot_v<3>(vc, P, nz, l.tmp, l.c);
		else
			rows_dot<4>(l.G, P, ld, nz, l.tmp, l.c);
		wave_sync();
		const double gl = sel(lane < P, l.g[lane < P ? lane : 0]);
		const double cl = sel(lane < P, l.c[lane < P ? lane : 0]);
		skip_qp = __ballot(lane < P && cl - gl > 1e-11 * (1.0 + fabs(gl))) == 0ull;
		if (!skip_qp) {
			y0i = yi;  // the loop's first step from y = 0 with an empty working set is this vector (below)
			y0_ok = true;
			yi = 0.0;
			level_res2 = 0.0;
		}
	}
	bool qp_ran = false;
	if (consider && !skip_qp) {
		// ---- 4b. primal active set on the piecewise quadratic ----
		qp_ran = true;
		if (lane < Pn)
			l.s[lane] = -l.g[lane];
		if (lane < mB)
			l.rho[lane] = -l.r0[lane];
		wave_sync();
		const unsigned long long softmask_main = (p >= 64) ? ~0ull : (((1ull << p) - 1ull) << P);
		const unsigned long long hardmask_main = (P >= 64) ? ~0ull : ((1ull << P) - 1ull);
		const double gl = sel(lane < Pn, l.g[lane < Pn ? lane : 0]);
		const double tolv = 1e-11 * (1.0 + fabs(gl));
		// Phase 1 (only when a prior row is violated at y = 0, which reference semantics allow:
		// levels without inequalities ignore the prior rows, ho_qp_problem.cpp:325-329): the same
		// loop with the prior rows as the only, soft, rows finds a feasible y; the main phase
		// then starts there.
		bool need_p1 = false;
		if (HARD) {
			int bad = 0;
			if (lane < P)
				bad = l.s[lane] > 1e-9 * (1.0 + fabs(l.g[lane]));
			need_p1 = __ballot(bad) != 0ull;
		}
		const int mB_main = mB;
		iters = 0;
		for (int phase = need_p1 ? 0 : 1; phase < 2; ++phase) {
		const bool p1 = phase == 0;
		const unsigned long long softmask = p1 ? hardmask_main : softmask_main;
		const unsigned long long hardmask = p1 ? 0ull : hardmask_main;
		const int soft_lo = p1 ? 0 : P, soft_hi = p1 ? P : Pn;
		mB = p1 ? 0 : mB_main;
		if (!p1 && need_p1) {
			// feasibility reached?  then rho = B y - r0 at the phase-1 point
			int bad = 0;
			if (lane < P)
				bad = l.s[lane] > 1e-9 * (1.0 + fabs(l.g[lane]));
			if (__ballot(bad))
				st.status |= HOQP_STATUS_INFEASIBLE_START;
			wave_sync();
			l.tmp[lane] = (lane < nz) ? yi : 0.0;
			wave_sync();
			rows_dot<2>(l.BL, mB, ld, rank, l.tmp, l.a0);
			wave_sync();
			if (lane < mB)
				l.rho[lane] = l.a0[lane] - l.r0[lane];
			wave_sync();
		}
		unsigned long long wmask = 0ull;
		int w = 0;
		const bool isSoft = lane < Pn && ((softmask >> lane) & 1ull);
		bool done = false;
		// After a full, unblocked step (t = 1) that leaves the set of violated soft rows unchanged
		// the iterate IS the minimiser of the model the last equality-constrained step was
		// computed for: the next step would be zero, so its factorization is skipped and the
		// multipliers are read from the reflectors still in TV.
		unsigned long long prev_vmask = 0ull;
		bool prev_full = false;
		int k1_keep = 0;
		for (; iters < a.max_iter && !done; ++iters) {
			const double si = sel(lane < Pn, l.s[lane < Pn ? lane : 0]);
			const double rj = sel(lane < mB, l.rho[lane < mB ? lane : 0]);
			const bool viol = isSoft && si > tolv;
			const unsigned long long vmask = __ballot(viol);
			const int nV = __popcll(vmask);
			if (viol)
				l.vlist[__popcll(vmask & ((1ull << lane) - 1ull))] = lane;
			if (w + mB + nV == 0) {  // nothing to minimise, nothing active: y is optimal
				done = true;
				break;
			}
			int k1 = k1_keep;
			bool overflow = false;
			HOQP_STAMP(5);
			double pi = 0.0;
			if (HARD && y0_ok && !need_p1 && iters == 0 && w == 0 && nV == 0) {
				// Nothing active, nothing violated, y = 0: the equality-constrained step is the minimum-norm
				// minimiser of |B0 p - r0| -- the unconstrained optimum y0 that was computed and found infeasible
				// a moment ago.  No factorization of the (triangular) least-squares block for a known answer.
				pi = y0i;
				k1 = 0;
			} else if (!(prev_full && vmask == prev_vmask)) {
				wave_sync();
				bool small = !HARD && mB == 0 && nV >= 1 && nV <= 8 && nV <= nz && eqp_dual8<COMPACT>(l, nz, nV, ld, &pi, vc);
#ifdef HOQP_PROFILE
				if (small && a.dbg && lane == 0)
					a.dbg[(size_t)blockIdx.x * kDbgSlots + 50] += 1;  // steps through the small side
#endif
				if (COMPACT && !small && mB == 0 && nV >= 1) {
					small = eqp_dual16(a, st, l, vc, nz, vmask, si, &pi);
#ifdef HOQP_PROFILE
					if (small && a.dbg && lane == 0)
						a.dbg[(size_t)blockIdx.x * kDbgSlots + 55] += 1;  // steps through the 16 x 16 tile
#endif
					// (eqp_dual16 used l.vlist for its compacted rows: restore the list of violated rows)
					wave_sync();
					if (viol)
						l.vlist[__popcll(vmask & ((1ull << lane) - 1ull))] = lane;
					wave_sync();
				}
				if (!small)
					pi = eqp_step<HARD, (COMPACT ? 1 : (CP ? 3 : 0)), CV>(a, l, st, ld, nz, rank, mB, w, nV, &k1, &overflow, vc);
				if (overflow) {
					st.status |= HOQP_STATUS_WORKSPACE;
					break;
				}
				k1_keep = k1;
			}
			prev_full = false;
			const double pn = wave_max(fabs(pi));
			const double yn = wave_max(fabs(yi));
			if (!(pn > 1e-10 * (1.0 + yn))) {
				if (!HARD || w == 0) {
					done = true;
					break;
				}
				// multipliers: G_W' mu = -grad,  grad = B0' rho + G_V' s_V
				double gi = 0.0;
				if (lane < nz) {
					if (lane < rank)
						for (int j = 0; j < mB; ++j)
							gi = fma(l.BL[j * ld + lane], l.rho[j], gi);
					unsigned long long mk = vmask;
					while (mk) {
						int r = __ffsll((long long)mk) - 1;
						mk &= mk - 1;
						if (CV) {
							const VRow vr = vrow(vc, r);
							gi = fma(vval(vr, vr.p0[lane], vr.p1[lane]), l.s[r], gi);
						} else {
							gi = fma(l.G[r * ld + lane], l.s[r], gi);
						}
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
						acc = fma(-l.TV[i * l.ldv + lane], mi, acc);
				}
				int wi = (lane < k1) ? l.pivlane[lane] : -1;  // pivot column == position in W
				int drop;
				double worst = wave_argmin((lane < k1) ? mu : 1e300, wi, &drop);
				HOQP_STAMP(12);
				if (!(worst < -1e-9 * (1.0 + gn)) || drop < 0) {
					done = true;
					break;
				}
				wmask &= ~(1ull << uni(l.W[drop]));
				wave_sync();
				if (lane == 0)
					for (int j = drop; j < w - 1; ++j)
						l.W[j] = l.W[j + 1];
				--w;
				wave_sync();
				continue;
			}
			// c_i = G_i p for every row, a0_j = B0_j p   (p broadcast through LDS)
			wave_sync();
			l.tmp[lane] = (lane < nz) ? pi : 0.0;
			wave_sync();
			if (CV)
				rows_dot_v<3>(vc, Pn, nz, l.tmp, l.c);  // at most 40 rows, 12 of them stored
			else if (WBC && KC >= 0)
				rows_dot<3>(l.G, Pn, ld, nz, l.tmp, l.c);  // at most 40 rows
			else
				rows_dot<4>(l.G, Pn, ld, nz, l.tmp, l.c);
			rows_dot<2>(l.BL, mB, ld, rank, l.tmp, l.a0);
			wave_sync();
			const double ci = sel(lane < Pn, l.c[lane < Pn ? lane : 0]);
			const double a0 = sel(lane < mB, l.a0[lane < mB ? lane : 0]);
			// ratio test on hard rows outside the working set
			int blk = -1;
			double t_hard = 1e300;
			if (HARD) {
				const bool isHard = lane < P && ((hardmask >> lane) & 1ull) && !((wmask >> lane) & 1ull);
				double ah = 1e300;
				if (isHard && ci > 1e-12 * pn)
					ah = fmax(0.0, -si / ci);
				t_hard = wave_argmin(ah, lane, &blk);
				if (!(t_hard < 1e299))
					blk = -1;
			}
			// The step p minimises the model with the violated set V held fixed, so its directional
			// derivative at t = 1 is zero.  Every s_i + t c_i is linear in t: if at t = 1 no row outside V
			// is violated and no row of V has gone negative (an under-determined step takes the rows of V
			// exactly to zero), the positive part was V all along [0, 1] and t = 1 is the exact minimiser of
			// the convex piecewise quadratic: no search (the usual last step of an instance).
			const double s1 = si + ci;
			const bool bad1 = isSoft && (viol ? (s1 < -tolv) : (s1 > tolv));
			const bool fullstep = __ballot(bad1) == 0ull && !(t_hard < 1.0);
			double t_star = 1.0;
			if (!fullstep) {
			const double q0 = wave_sum(a0 * rj), q1 = wave_sum(a0 * a0);
			// exact line search on the piecewise quadratic along p
			double tb = 1e300;
			if (isSoft && ci != 0.0 && ((si > tolv && ci < 0.0) || (si <= tolv && ci > 0.0))) {
				double t = -si / ci;
				if (t > 0.0)
					tb = t;
			}
			// derivative of the piecewise quadratic just right of this lane's breakpoint; the soft
			// rows [P, Pn) are broadcast from their lanes (no LDS round trip per row)
			double dmine = -1.0;
			{
				// (the lane's own row contributes c (s + tb c) = 0 at its own breakpoint, so it needs no
				// special case; max() instead of a select keeps the accumulation a plain FMA chain)
				double acc = q0 + q1 * tb, acc1 = 0.0;
				int r = soft_lo;
				for (; r + 1 < soft_hi; r += 2) {
					const double c0 = readlane_d(ci, r), s0 = readlane_d(si, r);
					const double c1 = readlane_d(ci, r + 1), s1 = readlane_d(si, r + 1);
					acc = fma(c0, fmax(fma(tb, c0, s0), 0.0), acc);
					acc1 = fma(c1, fmax(fma(tb, c1, s1), 0.0), acc1);
				}
				if (r < soft_hi) {
					const double c0 = readlane_d(ci, r), s0 = readlane_d(si, r);
					acc = fma(c0, fmax(fma(tb, c0, s0), 0.0), acc);
				}
				acc += acc1;
				dmine = (tb < 1e299) ? acc : -1.0;
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
			if (d_lo >= 0.0)
				t_star = t_lo;
			else if (slope > 0.0)
				t_star = fmin(t_lo - d_lo / slope, t_hi);
			else
				t_star = (t_hi < 1e299) ? t_hi : t_lo;
			}
			const double t = fmin(t_star, t_hard);
			prev_full = fabs(t - 1.0) <= 1e-9 && !(t_hard <= t_star && blk >= 0);
			prev_vmask = vmask;
			yi = fma(t, pi, yi);
			wave_sync();
			if (lane < Pn)
				l.s[lane] = fma(t, ci, si);
			if (lane < mB)
				l.rho[lane] = fma(t, a0, rj);
			if (HARD && t_hard <= t_star && blk >= 0) {
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
			wave_sync();
			HOQP_STAMP(13);
		}
		if (!done) {
			st.status |= HOQP_STATUS_MAX_ITER;
			break;
		}
		}  // phase
		mB = mB_main;

		{
			double rj = sel(lane < mB, l.rho[lane < mB ? lane : 0]);
			level_res2 = wave_sum(rj * rj) + res_const;
		}
	}

	HOQP_STAMP(5);
	// ---- 5./6. x_k = x_{k-1} + (Z Q) y (GetSolution, ho_qp_problem.cpp:46-51); slack bookkeeping
	//            (ConstructFVector, :207-229); Z_k = Z_{k-1} Q2, G <- G Q2 (:172-178) ----
	wave_sync();
	l.tmp[lane] = (lane < nz) ? yi : 0.0;
	wave_sync();
	{
		// After the active-set loop l.s holds G y - g for every row (it moved with y, step by step), so the
		// products G y need not be formed again; with rank == 0 the stored rows are not shifted either, and
		// the update visits the rows of Z only (the all-soft level of the WBC: 30 rows instead of 70).
		// (CP: G y is never formed again for the virtual rows -- after the loop it is in l.s, after a skipped loop
		// in l.c from the check of the unconstrained optimum; the update shifts the 12 stored rows only)
		const bool from_s = CP ? qp_ran : (qp_ran && rank == 0);
		const bool from_c = CP && !qp_ran && skip_qp;
		if (CP && Pg > 0 && !from_s && !from_c) {  // (not reached by the WBC hierarchy: kept rows, no free coordinate)
			rows_dot_v<3>(vc, Pg, nz, l.tmp, l.c);
			wave_sync();
		}
		const int Pu = CP ? ((Pg > 0 && rank > 0) ? kCGrows : 0) : (from_s ? 0 : Pg);
		const int R = n + Pu;
		if (updated)
			;
		else if (CP)
			level_update_t<3>(l, n, Pu, nz, rank, ld, HOQP_DBGP);
		else if (R <= 32)
			level_update_t<2>(l, n, Pu, nz, rank, ld, HOQP_DBGP);
		else if (R <= 64)
			level_update_t<4>(l, n, Pu, nz, rank, ld, HOQP_DBGP);
		else
			level_update_t<NSR>(l, n, Pu, nz, rank, ld, HOQP_DBGP);
		wave_sync();
		HOQP_STAMP(27);
		// one row per lane: x += (Z y), g -= (G y) with the slack bookkeeping of this level's rows
		if (lane < n)
			l.x[lane] += l.S[lane];
		bool nonfinite = !isfinite(level_res2);
		const double dq = q3 ? q3_delta() : 0.0;  // stored g goes back to each row's own slack
		if (lane < Pg) {
			double gi = (from_s ? 0.0 - l.s[lane] : (CP ? l.g[lane] - l.c[lane] : l.g[lane] - l.S[n + lane])) - dq;
			if (lane >= P) {
				// v* = (D x - f)_+ ; the row becomes hard with rhs f + v* - D x
				const double vopt = fmax(0.0, -gi);
				nonfinite |= !isfinite(gi);
				l.vs[lane] = vopt;
				if (a.out.v)
					a.out.v[(size_t)inst * a.Ptot + st.Pout + (lane - P)] = vopt;
				gi = fmax(gi, 0.0);
			}
			l.g[lane] = gi;
		}
		if (__ballot(nonfinite))
			st.status |= HOQP_STATUS_NONFINITE;
		// bookkeeping of the reference replay: the null space of a QP level with equality rows is not
		// rebuilt by it (an inequality-only level leaves the null space alone)
		if (consider && m > 0)
			st.replay_ok = 0;
	}
	// (level 0: both sides work in the identity basis, the first-order correction is then exact to (eps / sigma^2)^2
	// and there is nothing to replay; it is only reported when even that is in reach of the parity bar)
	if (__builtin_expect(a.eps != 0.0 && m > 0 && rank > 0 &&
				     a.eps > (k > 0 ? kReplayTau : 1e-4) * dmin_level * dmin_level,
			     0)) {
		bool replayed = false;
#ifndef HOQP_NO_REPLAY
		if (!consider && !keep_rows && st.replay_ok && k > 0) {
			wave_sync();
			double res2 = 0.0;
			replayed = ref_replay<WBC, (COMPACT ? 24 : NR)>(a, l, inst, k, n, nz, wx, jc, &res2);
			if (replayed)
				level_res2 = res2 + res_const;
		}
#endif
		if (!replayed)
			st.status |= HOQP_STATUS_NEAR_SINGULAR;  // first-order eps correction only (include/hoqp.h)
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
	wave_sync();
	HOQP_STAMP(6);
	HOQP_STAMP(46);	 // back-to-back with the previous one: what a stamp itself leaves in the next section

	if (a.out.x_levels && lane < a.n)
		a.out.x_levels[((size_t)inst * a.L + k) * a.n + lane] = (lane < n) ? l.x[lane] : 0.0;
	if (a.out.Z_levels) {
		double *Zo = a.out.Z_levels + ((size_t)inst * a.L + k) * a.n * a.n;
		for (int idx = lane; idx < a.n * a.n; idx += kWave) {
			int r = idx / a.n, c2 = idx - r * a.n;
			Zo[idx] = (r < n && c2 < st.nz) ? l.Z[r * ld + c2] : 0.0;
		}
	}
}

#endif  // __HIP_DEVICE_COMPILE__

#ifdef __HIP_DEVICE_COMPILE__
// ---------------------------------------------------------------------------
// Tail of the fused WBC entry: CalcJointTorquesCmd (ho_qp_controller.cpp:67-73), q_j_ddot (:23) and
// the five CheckSolutionValid results (:18-22, helper_functions.hpp:63-91) at the final x, from the
// dynamics terms themselves instead of the 76 generated task rows:
//   w  = M udd - J_c' lam + h :  rows 0..5 = residual of the floating-base EoM (level 0),
//                                rows 6..17 = tau; the torque rows of level 1 are tau - tau_max, tau_min - tau
//   jv = J_foot udd           :  contact feet + J_c_dot u = level 2, swing feet - b = level 3
// One row per quad (lane s of the quad takes entries s, s + 4, ...), then one row per lane.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wbc_tail(const SolveArgs &a, const Lds &l, const WbcCtx &wx, int inst, int &status)
{
	const int lane = lane_id(), s = lane & 3, c = lane >> 2;
	const WbcPrm &prm = a.wprm;
	ldsd *lamf = l.TV, *wv = l.TV + 16, *jv = l.TV + 40, *jb = l.TV + 56;  // the reflector store is dead here
	// contact force per foot row (zero for swing legs)
	if (lane < 12) {
		const int leg = lane / 3, d = lane - 3 * leg;
		int slot = -1;
#pragma unroll
		for (int i = 0; i < 4; ++i)
			slot = (i < wx.nc && wbc_pick(wx.clp, i) == leg) ? i : slot;
		lamf[lane] = (slot >= 0) ? l.x[kNU + 3 * (slot >= 0 ? slot : 0) + d] : 0.0;
	}
	wave_sync();
	double xq[5], lq[3];
#pragma unroll
	for (int t = 0; t < 5; ++t)
		xq[t] = l.x[(s + 4 * t < kNU) ? s + 4 * t : 0];
#pragma unroll
	for (int t = 0; t < 3; ++t)
		lq[t] = lamf[s + 4 * t];
#pragma unroll
	for (int t = 0; t < 5; ++t)
		pin(xq[t]);
#pragma unroll
	for (int t = 0; t < 3; ++t)
		pin(lq[t]);
#pragma unroll
	for (int t = 0; t < 5; ++t)
		xq[t] = sel(s + 4 * t < kNU, xq[t]);
	// w (18 rows, two passes) and jv (12 rows) and, with a general J_b, J_b udd (6 rows)
	double mv[2][5], jt[2][3], jr[5], br[5], hv[2];
	const int rowj = (c < 12) ? c : 0, rowb = (c < 6) ? c : 0;
#pragma unroll
	for (int ps = 0; ps < 2; ++ps) {
		const int i = c + 16 * ps;
		const int ic = (i < kNU) ? i : 0;
#pragma unroll
		for (int t = 0; t < 5; ++t)
			mv[ps][t] = wx.M[ic * kNU + ((s + 4 * t < kNU) ? s + 4 * t : 0)];
#pragma unroll
		for (int t = 0; t < 3; ++t)
			jt[ps][t] = wx.Jf[(s + 4 * t) * kNU + ic];
		hv[ps] = wx.h[ic];
	}
#pragma unroll
	for (int t = 0; t < 5; ++t)
		jr[t] = wx.Jf[rowj * kNU + ((s + 4 * t < kNU) ? s + 4 * t : 0)];
	if (wx.Jb) {
#pragma unroll
		for (int t = 0; t < 5; ++t)
			br[t] = wx.Jb[rowb * kNU + ((s + 4 * t < kNU) ? s + 4 * t : 0)];
	}
#pragma unroll
	for (int ps = 0; ps < 2; ++ps) {
		const int i = c + 16 * ps;
		double d0 = 0.0, d1 = 0.0;
#pragma unroll
		for (int t = 0; t < 5; ++t) {
			if (t & 1)
				d1 = fma(mv[ps][t], xq[t], d1);  // xq == 0 beyond column 17
			else
				d0 = fma(mv[ps][t], xq[t], d0);
		}
#pragma unroll
		for (int t = 0; t < 3; ++t)
			d1 = fma(-jt[ps][t], lq[t], d1);
		const double w = quad_sum(d0 + d1) + hv[ps];
		if (i < kNU && s == 0)
			wv[i] = w;
	}
	{
		double d0 = 0.0, d1 = 0.0;
#pragma unroll
		for (int t = 0; t < 5; ++t) {
			if (t & 1)
				d1 = fma(jr[t], xq[t], d1);
			else
				d0 = fma(jr[t], xq[t], d0);
		}
		const double d = quad_sum(d0 + d1);
		if (c < 12 && s == 0)
			jv[c] = d;
	}
	if (wx.Jb) {
		double d0 = 0.0, d1 = 0.0;
#pragma unroll
		for (int t = 0; t < 5; ++t) {
			if (t & 1)
				d1 = fma(br[t], xq[t], d1);
			else
				d0 = fma(br[t], xq[t], d0);
		}
		const double d = quad_sum(d0 + d1);
		if (c < 6 && s == 0)
			jb[c] = d;
	} else if (lane < 6) {
		jb[lane] = l.x[lane];  // J_b = [I6 0]
	}
	wave_sync();
	// ---- one row per lane ----
	const double wl = wv[lane < kNU ? lane : 0];
	if (lane >= kNB && lane < kNU) {
		if (a.tau)
			a.tau[(size_t)inst * kNJ + lane - kNB] = wl;
		if (a.qdd)
			a.qdd[(size_t)inst * kNJ + lane - kNB] = l.x[lane];
	}
	if (__ballot(lane < kNU && !isfinite(wl)))
		status |= HOQP_STATUS_NONFINITE;
	if (!(a.out.flags || a.out.eq_res_max || a.out.ineq_viol_max || a.out.active_mask))
		return;
	const int m2 = 3 * wx.nc, m3 = 6 + 3 * wx.ns, p1 = 2 * kNJ + 4 * wx.nc;
	// level 0
	const double er0 = wave_max(lane < kNB ? fabs(wl) : 0.0);
	// level 1: torque rows from tau, friction rows from x (Q1: columns 6 + 3 leg)
	double d1v = -1e300;
	if (lane < kNJ) {
		d1v = wv[kNB + lane] - prm.tau_max;
	} else if (lane < 2 * kNJ) {
		d1v = prm.tau_min - wv[kNB + lane - kNJ];
	} else if (lane < p1) {
		const int i = (lane - 2 * kNJ) >> 2, t = (lane - 2 * kNJ) & 3;
		const int col = prm.friction_quirk ? kNB + wbc_pick(wx.clp, i) * 3 : kNU + 3 * i;
		const double xa = l.x[col + (t >> 1)], xz = l.x[col + 2];
		d1v = fma(-prm.mu, xz, (t & 1) ? -xa : xa);
	}
	const double iv1 = wave_max(lane < p1 ? d1v : -1e300);
	const unsigned long long act = __ballot(lane < p1 && d1v >= -1e-7);
	// level 2
	double e2 = 0.0;
	if (lane < m2) {
		const int fr = wbc_crow(wx, lane);
		e2 = fabs(jv[fr] + wx.Jdu[fr]);
	}
	const double er2 = wave_max(e2);
	// level 3
	double e3 = 0.0;
	if (lane < 6) {
		const int rb = (lane < 3) ? 3 + lane : lane - 3;
		e3 = fabs(jb[rb] - l.b3[lane]);
	} else if (lane < m3) {
		e3 = fabs(jv[wbc_srow(wx, lane - 6)] - l.b3[lane]);
	}
	const double er3 = wave_max(e3);
	// level 4
	const double er4 = wave_max(lane < m2 ? fabs(l.x[kNU + (lane < m2 ? lane : 0)]) : 0.0);
	// The maxima above are fmax reductions, which drop a NaN: an instance whose x is not finite (or that was not
	// solved at all, HOQP_STATUS_BASE_RANK) must never come out "valid" -- its residuals are NaN, its flags 0.
	const bool unsolved = (status & (HOQP_STATUS_NONFINITE | HOQP_STATUS_BASE_RANK)) != 0;
	if (lane < 5) {
		const double er = lane == 0 ? er0 : (lane == 2 ? er2 : (lane == 3 ? er3 : (lane == 4 ? er4 : 0.0)));
		const double iv = lane == 1 ? iv1 : 0.0;
		const size_t o = (size_t)inst * 5 + lane;
		if (a.out.flags)
			a.out.flags[o] = (!unsolved && er <= 1e-7 && (lane != 1 || iv <= 0.0)) ? 1 : 0;
		if (a.out.eq_res_max)
			a.out.eq_res_max[o] = unsolved ? __builtin_nan("") : er;
		if (a.out.ineq_viol_max)
			a.out.ineq_viol_max[o] = unsolved ? __builtin_nan("") : iv;
	}
	if (lane == 0 && a.out.active_mask)
		a.out.active_mask[inst] = unsolved ? 0ull : act;
}
#endif  // __HIP_DEVICE_COMPILE__

template <bool GATE, bool HARD, bool WBC>
__device__ __forceinline__ void hoqp_solve_body(const SolveArgs &a)
{
#ifdef __HIP_DEVICE_COMPILE__
	extern __shared__ __align__(16) char smem[];
	constexpr int CK = (WBC && GATE && !HARD) ? 1 : ((WBC && !GATE && HARD) ? 2 : 0);  // compact layouts
	const Lds l = lds_bind<CK>(a, (HOQP_LDS char *)smem);
	const int lane = lane_id();
	const int inst = blockIdx.x;
	if (inst >= a.batch)
		return;
	LevelState st;
	st.status = 0;
	st.replay_ok = 1;
	st.bail = 0;
	WbcCtx wx = {};
	int jc[NR / 4] = {-1, -1, -1, -1, -1, -1, -1, -1};
	if (WBC) {
		wx = wbc_ctx(a.win, inst);
		wx.clp = uni(wx.clp);
		wx.slp = uni(wx.slp);
		wx.nc = uni(wx.nc);
		wx.ns = uni(wx.ns);
		wx.n = uni(wx.n);
		__builtin_assume(wx.nc >= 0 && wx.nc <= 4);
		__builtin_assume(wx.ns >= 0 && wx.ns <= 4);
		__builtin_assume(wx.n == kNU + 3 * wx.nc);
#pragma unroll
		for (int kb = 4; kb < NR / 4; ++kb)
			jc[kb] = wbc_jcoff(wx, 4 * kb + (lane >> 4));
		st.n = wx.n;
	} else {
		// per-instance extents are device data: clamp them to the storage extents (a larger value
		// would run past the LDS arrays and into the next instance's rows) and report it
		const int ni = uni(a.n_inst ? a.n_inst[inst] : a.n);
		st.n = min(max(ni, 1), a.n);
		if (st.n != ni)
			st.status |= HOQP_STATUS_BAD_EXTENT;
	}
	st.nz = st.n;
	st.P = 0;
	st.Pout = 0;
	st.t_last = __builtin_amdgcn_s_memtime();
	const int n = st.n, ld = a.ld;
	// InitPrevProblemValuesToDefault, ho_qp_problem.cpp:119-128:  Z = I, x = 0  (the specialised level 0 of the
	// fused reference kernel never reads Z = I: its projection is direct and its rotation generates the rows)
	if (CK == 0)
		lds_set_identity(l.Z, n, ld);
	if (lane < NR)
		l.x[lane] = 0.0;
	// Touch every cache line of the instance's M, J_foot, h and J_dot u now (one load per lane, one line
	// of 16 doubles each): the levels read them one after the other, and each first touch would otherwise
	// expose a full HBM round trip (~2 k cycles) to the only wave of the SIMD.
	double warm = 0.0;
	if (WBC) {
		const double *pw;
		if (lane < 21)
			pw = wx.M + min(16 * lane, kNU * kNU - 1);
		else if (lane < 35)
			pw = wx.Jf + min(16 * (lane - 21), 12 * kNU - 1);
		else if (lane < 37)
			pw = wx.h + min(16 * (lane - 35), kNU - 1);
		else if (lane < 39)
			pw = wx.M + ((lane == 37) ? kNU * kNU - 1 : 0);
		else if (lane < 41)
			pw = wx.Jf + ((lane == 39) ? 12 * kNU - 1 : 0);
		else
			pw = wx.Jdu + ((lane & 1) ? 11 : 0);
		warm = *pw;
	}
	if (WBC && lane < 24)
		l.b3[lane] = wbc_rhs3(wx, a.wprm, lane);
	if (a.out.v)
		for (int i = lane; i < a.Ptot; i += kWave)
			a.out.v[(size_t)inst * a.Ptot + i] = 0.0;
	wave_sync();
	asm volatile("" ::"v"(warm));  // the warming loads must be issued; their values are not used

	HOQP_STAMP(0);
	// one level; kc: std::integral_constant<int, level> for a specialised copy, <int, -1> for the run-time loop
	auto run_level = [&](auto kc, int k_rt) __attribute__((always_inline)) {
		constexpr int KC = decltype(kc)::value;
		const int k = (KC >= 0) ? KC : k_rt;
		HOQP_STAMP(48);	 // loop back-edge
		int m, p;
		if (WBC) {
			m = wbc_m(wx, k);
			p = wbc_p(wx, k);
		} else {
			const int mi = uni(a.m_inst ? a.m_inst[(size_t)inst * a.L + k] : a.m[k]);
			const int pi = uni(a.p_inst ? a.p_inst[(size_t)inst * a.L + k] : a.p[k]);
			m = min(max(mi, 0), a.m[k]);
			p = min(max(pi, 0), a.p[k]);
			if (m != mi || p != pi)
				st.status |= HOQP_STATUS_BAD_EXTENT;
		}
		if (GATE && a.skip_unread && !a.out.v && m == 0 && p > 0 && k >= a.last_ineq_use) {
			// HOQP_OPT_SKIP_UNREAD_SLACK_LEVELS: nobody reads this level's rows (Q2) or slacks (v == NULL)
			if (lane == 0) {
				const size_t o = (size_t)inst * a.L + k;
				if (a.out.level_res)
					a.out.level_res[o] = 0.0;
				if (a.out.iters)
					a.out.iters[o] = 0;
				if (a.out.nz_levels)
					a.out.nz_levels[o] = st.nz;
			}
			if (a.out.x_levels && lane < a.n)
				a.out.x_levels[((size_t)inst * a.L + k) * a.n + lane] = (lane < n) ? l.x[lane] : 0.0;
			st.Pout += p;
			wave_sync();
			return;
		}
		solve_level<GATE, HARD, WBC, KC>(a, l, st, inst, k, m, p, wx, jc);
		HOQP_STAMP(49);	 // per-level outputs after the last stamp of the level
	};
	if (WBC && GATE && !HARD) {
		// the reference's WBC hierarchy (fill_args: L == 5): five specialised copies
		run_level(std::integral_constant<int, 0>{}, 0);
		if (!st.bail) {
			run_level(std::integral_constant<int, 1>{}, 1);
			run_level(std::integral_constant<int, 2>{}, 2);
			run_level(std::integral_constant<int, 3>{}, 3);
			run_level(std::integral_constant<int, 4>{}, 4);
		} else {
			// HOQP_STATUS_BASE_RANK: nothing was solved; every output says so
			wave_sync();
			if (lane < NR)
				l.x[lane] = __builtin_nan("");
			if (lane < a.L) {
				const size_t o = (size_t)inst * a.L + lane;
				if (a.out.level_res)
					a.out.level_res[o] = __builtin_nan("");
				if (a.out.iters)
					a.out.iters[o] = 0;
				if (a.out.nz_levels)
					a.out.nz_levels[o] = 0;
			}
			if (a.out.x_levels)
				for (int i = lane; i < a.L * a.n; i += kWave)
					a.out.x_levels[(size_t)inst * a.L * a.n + i] = __builtin_nan("");
			if (a.out.v)
				for (int i = lane; i < a.Ptot; i += kWave)
					a.out.v[(size_t)inst * a.Ptot + i] = __builtin_nan("");
			wave_sync();
		}
	} else {
		for (int k = 0; k < a.L && !st.bail; ++k)
			run_level(std::integral_constant<int, -1>{}, k);
		if (st.bail) {  // (paper-semantics WBC kernel: HOQP_STATUS_BASE_RANK, as above)
			wave_sync();
			if (lane < NR)
				l.x[lane] = __builtin_nan("");
			if (lane < a.L) {
				const size_t o = (size_t)inst * a.L + lane;
				if (a.out.level_res)
					a.out.level_res[o] = __builtin_nan("");
				if (a.out.iters)
					a.out.iters[o] = 0;
				if (a.out.nz_levels)
					a.out.nz_levels[o] = 0;
			}
			if (a.out.x_levels)
				for (int i = lane; i < a.L * a.n; i += kWave)
					a.out.x_levels[(size_t)inst * a.L * a.n + i] = __builtin_nan("");
			if (a.out.v)
				for (int i = lane; i < a.Ptot; i += kWave)
					a.out.v[(size_t)inst * a.Ptot + i] = __builtin_nan("");
			wave_sync();
		}
	}
	// ---- final x and the CheckSolutionValid flags (helper_functions.hpp:63-91) ----
	double xi = (lane < n) ? l.x[lane] : 0.0;
	if (__ballot(!isfinite(xi)))
		st.status |= HOQP_STATUS_NONFINITE;
	if (a.out.x && lane < a.n)
		a.out.x[(size_t)inst * a.n + lane] = xi;
	if (WBC) {
		wbc_tail(a, l, wx, inst, st.status);
	} else if (a.out.flags || a.out.eq_res_max || a.out.ineq_viol_max || a.out.active_mask) {
		// one task row per quad (row = 16 * pass + c, lane s of the quad takes entries s, s+4, ...):
		// every load instruction covers 16 rows x 32 contiguous bytes instead of 64 rows x 8 bytes
		const int s = lane & 3, c = lane >> 2;
		double xq[NT];
#pragma unroll
		for (int t = 0; t < NT; ++t)
			xq[t] = l.x[s + 4 * t];
#pragma unroll
		for (int t = 0; t < NT; ++t)
			pin(xq[t]);
#pragma unroll
		for (int t = 0; t < NT; ++t)
			xq[t] = sel(s + 4 * t < n, xq[t]);
		// rows . x - rhs for `cnt` rows; returns the wave maximum of |.| (equalities) or of the signed
		// value (inequalities) and, for inequalities, the bit mask of rows with value >= -1e-7
		auto check_rows = [&](const double *rows_g, const double *rhs_g, int cnt, bool ineq, unsigned long long *act) {
			double mx = ineq ? -1e300 : 0.0;
			bool bad = false;
			unsigned long long bits = 0ull;
			for (int r0 = 0; r0 < cnt; r0 += 16) {
				const int r = r0 + c;
				const bool valid = r < cnt;
				const double *row = rows_g + (size_t)(valid ? r : 0) * a.n;
				double tv[NT];
#pragma unroll
				for (int t = 0; t < NT; ++t)
					tv[t] = row[(s + 4 * t < a.n) ? s + 4 * t : 0];  // all loads in flight at once
				const double rh = rhs_g[valid ? r : 0];
				double d0 = 0.0, d1 = 0.0;
#pragma unroll
				for (int t = 0; t < NT; t += 2) {
					// storage columns >= n are not part of the problem (they may hold anything)
					d0 = fma(sel(s + 4 * t < n, tv[t]), xq[t], d0);
					d1 = fma(sel(s + 4 * t + 4 < n, tv[t + 1]), xq[t + 1], d1);
				}
				const double d = quad_sum(d0 + d1) - rh;
				const double e = ineq ? d : fabs(d);
				mx = valid ? fmax(mx, e) : mx;
				bad |= valid && !isfinite(e);  // fmax drops a NaN: keep it visible
				if (ineq) {
					const int pred = valid && e >= -1e-7;
					const int mine = __shfl(pred, (lane * 4) & (kWave - 1));  // lane q < 16 <- quad q
					const unsigned long long bm = __ballot(lane < 16 && mine);
					if (r0 < 64)
						bits |= bm << r0;
				}
			}
			if (act)
				*act = bits;
			const double wm = wave_max(mx);
			return __ballot(bad) ? __builtin_nan("") : wm;
		};
		unsigned long long amask = 0ull;
		int prow = 0;
		for (int k = 0; k < a.L; ++k) {
			const int m = min(max(uni(a.m_inst ? a.m_inst[(size_t)inst * a.L + k] : a.m[k]), 0), a.m[k]);
			const int p = min(max(uni(a.p_inst ? a.p_inst[(size_t)inst * a.L + k] : a.p[k]), 0), a.p[k]);
			double er = 0.0, iv = -1e300;
			if (m > 0)
				er = check_rows(a.A[k] + (size_t)inst * a.m[k] * a.n, a.b[k] + (size_t)inst * a.m[k], m, false, nullptr);
			if (p > 0) {
				unsigned long long act = 0ull;
				iv = check_rows(a.D[k] + (size_t)inst * a.p[k] * a.n, a.f[k] + (size_t)inst * a.p[k], p, true, &act);
				if (prow < 64)
					amask |= act << prow;
			}
			prow += p;
			if (lane == 0) {
				size_t o = (size_t)inst * a.L + k;
				// a NaN residual is "violated", not "valid"
				if (a.out.flags)
					a.out.flags[o] = (er <= 1e-7 && !(p > 0 && !(iv <= 0.0))) ? 1 : 0;
				if (a.out.eq_res_max)
					a.out.eq_res_max[o] = er;
				if (a.out.ineq_viol_max)
					a.out.ineq_viol_max[o] = p ? iv : 0.0;
			}
			if (!isfinite(er) || (p > 0 && !isfinite(iv)))
				st.status |= HOQP_STATUS_NONFINITE;
		}
		if (lane == 0 && a.out.active_mask)
			a.out.active_mask[inst] = amask;
	}
	if (lane == 0 && a.out.status)
		a.out.status[inst] = st.status;
	HOQP_STAMP(7);
#endif  // __HIP_DEVICE_COMPILE__
}

// Generic instantiations (stored task blocks; the fused WBC entry under paper semantics): one wave per SIMD.
template <bool GATE, bool HARD, bool WBC>
#ifdef HOQP_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(HOQP_WAVES_PER_EU, HOQP_WAVES_PER_EU)))
#endif
__global__ void __launch_bounds__(kWave) hoqp_solve_kernel(const SolveArgs a)
{
	hoqp_solve_body<GATE, HARD, WBC>(a);
}

// The fused WBC entry under reference semantics (the benchmarked kernel): compact LDS layout (six workgroups
// per CU) and the register budget of two waves per SIMD (256 registers per lane in total; the compiler would
// otherwise take 256 + 45 accumulation registers, which pins the kernel at one wave per SIMD whatever the LDS).
#ifndef HOQP_WBC_REF_WAVES
#define HOQP_WBC_REF_WAVES 2
#endif
// The generic instantiations once more under the register budget of two waves per SIMD, for hierarchies whose
// LDS footprint lets at least six workgroups share a CU (launch_solve picks: <= 27,306 B).  A small hierarchy
// (n = 8 ... 20) runs 1.5 - 1.65 x faster this way; a 30-variable one with stored rows (38 KB) stays at one wave
// per SIMD whatever its registers and would only pay the limit's spills (-7 %), hence two sets.
template <bool GATE, bool HARD>
__attribute__((amdgpu_waves_per_eu(2, 2)))
__global__ void __launch_bounds__(kWave) hoqp_solve_kernel_w2(const SolveArgs a)
{
	hoqp_solve_body<GATE, HARD, false>(a);
}

// The fused WBC entry under paper semantics: the compact strides and virtual rows with hard rows (kPaperDims).
__attribute__((amdgpu_waves_per_eu(HOQP_WBC_REF_WAVES, HOQP_WBC_REF_WAVES)))
__global__ void __launch_bounds__(kWave) hoqp_solve_wbc_paper_kernel(const SolveArgs a)
{
	hoqp_solve_body<false, true, true>(a);
}

__attribute__((amdgpu_waves_per_eu(HOQP_WBC_REF_WAVES, HOQP_WBC_REF_WAVES)))
__global__ void __launch_bounds__(kWave) hoqp_solve_wbc_ref_kernel(const SolveArgs a)
{
	hoqp_solve_body<true, false, true>(a);
}

}  // namespace hoqp
