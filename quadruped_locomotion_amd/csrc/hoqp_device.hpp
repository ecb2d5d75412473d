// hoqp_device.hpp -- wavefront-level building blocks of the HO-QP solve (gfx950, wave64).
//
// One workgroup == one wavefront (64 lanes) == one problem instance; every matrix of the
// instance lives in LDS.  All routines here are called by the whole wavefront with
// wave-uniform control flow.  `__syncthreads()` in a 64-thread workgroup lowers to the
// LDS wait plus a (single-wave) barrier, i.e. it only orders LDS traffic.
#pragma once
#include <hip/hip_runtime.h>

namespace hoqp {

constexpr int kWave = 64;
constexpr double kEpsMach = 2.220446049250313e-16;

__device__ __forceinline__ int lane_id() { return threadIdx.x; }

// ---------------------------------------------------------------------------
// cross-lane primitives (DPP butterflies inside 16-lane rows, readlane across rows)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double readlane_d(double v, int l)
{
	int lo = __double2loint(v), hi = __double2hiint(v);
	lo = __builtin_amdgcn_readlane(lo, l);
	hi = __builtin_amdgcn_readlane(hi, l);
	return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_d(double v)
{
	int lo = __double2loint(v), hi = __double2hiint(v);
	lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
	hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
	return __hiloint2double(hi, lo);
}

// 0xB1 quad_perm[1,0,3,2], 0x4E quad_perm[2,3,0,1], 0x141 row_half_mirror, 0x140 row_mirror:
// after the four butterflies every lane holds the reduction of its 16-lane row.
__device__ __forceinline__ double wave_sum(double v)
{
	v += dpp_d<0xB1>(v);
	v += dpp_d<0x4E>(v);
	v += dpp_d<0x141>(v);
	v += dpp_d<0x140>(v);
	return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

__device__ __forceinline__ double wave_max(double v)
{
	v = fmax(v, dpp_d<0xB1>(v));
	v = fmax(v, dpp_d<0x4E>(v));
	v = fmax(v, dpp_d<0x141>(v));
	v = fmax(v, dpp_d<0x140>(v));
	return fmax(fmax(readlane_d(v, 0), readlane_d(v, 16)), fmax(readlane_d(v, 32), readlane_d(v, 48)));
}

__device__ __forceinline__ double wave_min(double v) { return -wave_max(-v); }

// argmax over lanes: returns the max value, *idx = payload of the lowest lane holding it
// (-1 if every lane passed v < 0, which callers use as "no candidate").
__device__ __forceinline__ double wave_argmax(double v, int payload, int *idx)
{
	double m = wave_max(v);
	unsigned long long b = __ballot(v == m);
	int l = b ? (__ffsll((long long)b) - 1) : 0;
	*idx = __builtin_amdgcn_readlane(payload, l);
	return m;
}

__device__ __forceinline__ double wave_argmin(double v, int payload, int *idx)
{
	return -wave_argmax(-v, payload, idx);
}

// ---------------------------------------------------------------------------
// Householder QR with column pivoting on an LDS matrix T (column-major, leading
// dimension ldt, nr rows).  One lane owns one column (strided by 64 beyond that).
//
// Steps start at k0.  Pivots are chosen among columns [cb, ce) that are still
// unpivoted (cpos[c] < 0); each reflector is applied to every unpivoted column of
// [0, ncol).  Reflector k is left in column piv[k], rows k..nr-1 (v_k at row k), the R
// entries above it; rdiag[k] = R_kk, tau[k] = 2 / v'v.  Stops when the largest remaining
// column norm is <= rtol * (first pivot norm of this call).  Returns the new step count.
// ---------------------------------------------------------------------------
__device__ inline int hh_factor(double *T, int ldt, int nr, int k0, int cb, int ce, int ncol, double rtol,
				int *cpos, int *piv, double *tau, double *rdiag)
{
	const int lane = lane_id();
	int k = k0;
	double first = -1.0;
	while (k < nr) {
		double best = -1.0;
		int bc = -1;
		for (int c = cb + lane; c < ce; c += kWave) {
			if (cpos[c] >= 0)
				continue;
			const double *col = T + (size_t)c * ldt;
			double s = 0.0;
			for (int i = k; i < nr; ++i)
				s = fma(col[i], col[i], s);
			if (s > best) {
				best = s;
				bc = c;
			}
		}
		int pc;
		best = wave_argmax(best, bc, &pc);
		if (pc < 0 || !(best > 0.0))
			break;
		double nrm = sqrt(best);
		if (first < 0.0)
			first = nrm;
		if (!(nrm > rtol * first))
			break;
		double *pcol = T + (size_t)pc * ldt;
		double alpha = pcol[k];
		double beta = (alpha >= 0.0) ? -nrm : nrm;
		double tk = 1.0 / (nrm * (nrm + fabs(alpha)));
		__syncthreads();
		if (lane == 0) {
			pcol[k] = alpha - beta;
			piv[k] = pc;
			tau[k] = tk;
			rdiag[k] = beta;
			cpos[pc] = k;
		}
		__syncthreads();
		for (int c = lane; c < ncol; c += kWave) {
			if (cpos[c] >= 0)
				continue;
			double *col = T + (size_t)c * ldt;
			double s = 0.0;
			for (int i = k; i < nr; ++i)
				s = fma(pcol[i], col[i], s);
			s *= tk;
			for (int i = k; i < nr; ++i)
				col[i] = fma(-s, pcol[i], col[i]);
		}
		__syncthreads();
		++k;
	}
	return k;
}

// rows <- rows * H_ka * ... * H_(kb-1): applies reflectors [ka, kb) from the right to
// `nrows` row-major rows (leading dimension ldr, nr columns).  One lane per row.
__device__ inline void hh_apply_right_rows(double *rows, int ldr, int nrows, int nr, const double *T, int ldt,
					   const int *piv, const double *tau, int ka, int kb)
{
	for (int r = lane_id(); r < nrows; r += kWave) {
		double *row = rows + (size_t)r * ldr;
		for (int k = ka; k < kb; ++k) {
			const double *v = T + (size_t)piv[k] * ldt;
			double s = 0.0;
			for (int i = k; i < nr; ++i)
				s = fma(v[i], row[i], s);
			s *= tau[k];
			for (int i = k; i < nr; ++i)
				row[i] = fma(-s, v[i], row[i]);
		}
	}
}

// One entry per lane (lane i holds vec_i, nr <= 64).
// forward = true : vec <- H_(kb-1) ... H_ka vec   (= Q' vec for ka = 0)
// forward = false: vec <- H_ka ... H_(kb-1) vec   (= Q vec)
__device__ inline double hh_apply_vec(double vi, int nr, const double *T, int ldt, const int *piv, const double *tau,
				      int ka, int kb, bool forward)
{
	const int lane = lane_id();
	for (int t = 0; t < kb - ka; ++t) {
		int k = forward ? ka + t : kb - 1 - t;
		double v = (lane >= k && lane < nr) ? T[(size_t)piv[k] * ldt + lane] : 0.0;
		double s = wave_sum(v * vi) * tau[k];
		vi = fma(-s, v, vi);
	}
	return vi;
}

// R(i, j) of the factorization in PIVOT positions, i <= j.
__device__ __forceinline__ double r_entry(const double *T, int ldt, const int *piv, const double *rdiag, int i, int j)
{
	return (i == j) ? rdiag[i] : T[(size_t)piv[j] * ldt + i];
}

}  // namespace hoqp
