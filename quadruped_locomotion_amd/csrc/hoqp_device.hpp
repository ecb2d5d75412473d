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

// LDS-qualified types.  Every LDS pointer in the solver is address_space(3) in its TYPE, so
// the compiler emits ds_read/ds_write; a generic pointer derived through a struct or a
// select is lowered to flat_load/flat_store (measured: 10x slower here).
#define HOQP_LDS __attribute__((address_space(3)))
typedef HOQP_LDS double ldsd;
typedef HOQP_LDS int ldsi;
typedef double d2v __attribute__((ext_vector_type(2)));
typedef HOQP_LDS d2v ldsd2;
constexpr double kEpsMach = 2.220446049250313e-16;

// The lane index is handed out through an empty volatile asm, so the compiler cannot see that two routines
// derive their (s, c) maps, masks and LDS addresses from the same value.  Without it the optimizer shares and
// hoists every lane-derived expression of the one big kernel function: hundreds of values (the comparison masks
// alone are an SGPR pair each) that live from the kernel's head to their last use -- 611 SGPR and 91 VGPR spills,
// 272 B of scratch per lane in the fused reference kernel.  Recomputing them per routine costs a few integer
// instructions and leaves 247 VGPRs, no scratch, 85 SGPR spills: +9 %.
__device__ __forceinline__ int lane_id()
{
	int l = threadIdx.x;
	asm volatile("" : "+v"(l));
	return l;
}

// Ordering point between LDS writes and reads of different lanes.  The workgroup is a single
// wavefront and the LDS pipeline executes one wavefront's instructions in issue order, so
// only the COMPILER must be kept from reordering; __syncthreads() would add a full
// s_waitcnt lgkmcnt(0) drain (one LDS round trip, ~100+ cycles) at every call.
__device__ __forceinline__ void wave_sync()
{
	__builtin_amdgcn_wave_barrier();
	asm volatile("" ::: "memory");
}

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
	// mov_dpp (destination undefined before the move) instead of update_dpp(old = v): every lane of the
	// patterns used here has a valid source, and the tied "old" operand cost a v_mov copy per half
	lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
	hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
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

// N wave sums at once: the same butterflies as wave_sum with the N dependency chains interleaved (a
// single wave_sum is a chain of ~16 dependent operations; with one wave per SIMD nothing else hides it)
template <int N>
__device__ __forceinline__ void wave_sum_n(double (&v)[N])
{
#pragma unroll
	for (int i = 0; i < N; ++i)
		v[i] += dpp_d<0xB1>(v[i]);
#pragma unroll
	for (int i = 0; i < N; ++i)
		v[i] += dpp_d<0x4E>(v[i]);
#pragma unroll
	for (int i = 0; i < N; ++i)
		v[i] += dpp_d<0x141>(v[i]);
#pragma unroll
	for (int i = 0; i < N; ++i)
		v[i] += dpp_d<0x140>(v[i]);
#pragma unroll
	for (int i = 0; i < N; ++i)
		v[i] = (readlane_d(v[i], 0) + readlane_d(v[i], 16)) + (readlane_d(v[i], 32) + readlane_d(v[i], 48));
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
// cheap pivot search and reciprocal (square) roots for the Householder steps
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v)
{
	return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}

// Lane holding the largest 32-bit key, keys identical within a quad, 0 = "not a candidate";
// -1 if every key is 0.  Ties go to the lowest lane.  Two DPP stages (row_ror:4, row_ror:8)
// cover the four quads of a 16-lane row, the four rows are merged on the scalar unit.
__device__ __forceinline__ int wave_argmax_key(unsigned key)
{
	unsigned m = key;
	unsigned t = dpp_u<0x124>(m);
	m = m > t ? m : t;
	t = dpp_u<0x128>(m);
	m = m > t ? m : t;
	const unsigned m0 = __builtin_amdgcn_readlane(m, 0), m1 = __builtin_amdgcn_readlane(m, 16);
	const unsigned m2 = __builtin_amdgcn_readlane(m, 32), m3 = __builtin_amdgcn_readlane(m, 48);
	const unsigned a = m0 > m1 ? m0 : m1, b = m2 > m3 ? m2 : m3;
	const unsigned mx = a > b ? a : b;
	if (mx == 0)
		return -1;
	const unsigned long long bal = __ballot(key == mx);
	return __ffsll((long long)bal) - 1;
}

// 1/sqrt(b) for normal positive b: hardware estimate (~2^-23) + two Newton steps
__device__ __forceinline__ double fast_rsqrt(double b)
{
	double y = __builtin_amdgcn_rsq(b);
	const double h = 0.5 * b;
	y = y * fma(-h * y, y, 1.5);
	y = y * fma(-h * y, y, 1.5);
	return y;
}

// 1/d for normal d: hardware estimate + two Newton steps
__device__ __forceinline__ double fast_rcp(double d)
{
	double r = __builtin_amdgcn_rcp(d);
	r = fma(fma(-d, r, 1.0), r, r);
	r = fma(fma(-d, r, 1.0), r, r);
	return r;
}

}  // namespace hoqp
