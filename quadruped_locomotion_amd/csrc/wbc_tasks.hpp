// wbc_tasks.hpp -- the five WBC task blocks as a GENERATOR (device code, gfx950).
//
// HoQpController::ConstructTasks (ho_qp_controller.cpp:147-184) with its builders (:186-400)
// writes the hierarchy into dense matrices; here every entry of every block is a pure function
// of (level, row, column) and the per-instance dynamics terms, so the solve kernel forms the
// matrix-core operand straight from M / h / J (fused entry, no task round trip through HBM),
// and wbc_build_kernel (task dump, allocation-free entry) writes the very same values.
//
// An entry is +-(one input value) or a constant: Entry{address, scale, add}, value =
// scale != 0 ? scale * *address : add.  The address is ALWAYS readable, so the load is issued
// unconditionally (a load behind a condition costs a branch and a wait per entry).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/hoqp.h"

namespace hoqp {

constexpr int kNU = 18, kNJ = 12, kNB = 6, kNX = 30;

// HoQpController constants and switches (ho_qp_controller.hpp:54-56, gains of the builders)
struct WbcPrm {
	int friction_quirk, drop_base_gain;
	double mu, tau_max, tau_min, kp_swing, kv_swing, kp_base, kv_base, kv_rot;
};

// storage extents of the WBC hierarchy (n = 30; rows at n_c = 4 / n_s = 4 maxima)
constexpr int kWbcMmax[5] = {6, 0, 12, 18, 12};
constexpr int kWbcPmax[5] = {0, 40, 0, 0, 0};

struct Entry {
	const double *p;  // always a readable address (the load is issued unconditionally)
	double scale;	  // 0: the entry is the constant `add`
	double add;
};

// One instance's inputs and contact pattern (wave-uniform).  GetContactLegs / GetSwingLegs
// (ho_qp_controller.cpp:426-450) as 2-bit lists: leg of the i-th contact = (clp >> 2 i) & 3.
struct WbcCtx {
	const double *M, *h, *Jf, *Jdu, *fp, *Jb, *q, *u, *bc, *lc;
	int clp, slp, nc, ns, n;
};

__device__ __forceinline__ int wbc_pick(int packed, int i) { return (packed >> (2 * (i & 3))) & 3; }

__device__ __forceinline__ WbcCtx wbc_ctx(const hoqp_wbc_inputs &in, int inst)
{
	WbcCtx x;
	x.M = in.M + (size_t)inst * kNU * kNU;
	x.h = in.h + (size_t)inst * kNU;
	x.Jf = in.Jfoot + (size_t)inst * 12 * kNU;
	x.Jdu = in.Jdotu + (size_t)inst * 12;
	x.fp = in.footpos + (size_t)inst * 12;
	x.Jb = in.Jb ? in.Jb + (size_t)inst * 6 * kNU : nullptr;
	x.q = in.q + (size_t)inst * 19;
	x.u = in.u + (size_t)inst * kNU;
	x.bc = in.base_cmd + (size_t)inst * 9;
	x.lc = in.leg_cmd + (size_t)inst * 36;
	const int *ct = in.contact + (size_t)inst * 4;
	int clp = 0, slp = 0, nc = 0, ns = 0;
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		const int c = ct[g];
		if (c == 1) {
			clp |= g << (2 * nc);
			++nc;
		}
		if (c == 0) {
			slp |= g << (2 * ns);
			++ns;
		}
	}
	x.clp = clp;
	x.slp = slp;
	x.nc = nc;
	x.ns = ns;
	x.n = kNU + 3 * nc;  // ho_qp_controller.cpp:64
	return x;
}

// rows of level k for this contact pattern (ConstructTasks, :147-184)
__device__ __forceinline__ int wbc_m(const WbcCtx &x, int k)
{
	return k == 0 ? kNB : (k == 2 || k == 4) ? 3 * x.nc : (k == 3 ? 6 + 3 * x.ns : 0);
}
__device__ __forceinline__ int wbc_p(const WbcCtx &x, int k) { return k == 1 ? 2 * kNJ + 4 * x.nc : 0; }

// offset into Jfoot of the foot row that column c >= 18 of x (a contact-force component)
// multiplies, -1 for c < 18 or c >= n.  Depends on the column only.
__device__ __forceinline__ int wbc_jcoff(const WbcCtx &x, int c)
{
	const int i = c - kNU;
	const int ci = (i * 11) >> 5;  // i / 3 for 0 <= i < 32
	const int d = i - 3 * ci;
	const int off = (wbc_pick(x.clp, ci) * 3 + d) * kNU;
	return (c >= kNU && c < x.n) ? off : -1;
}

// foot row (0..11) of row i of the stacked contact / swing Jacobian
__device__ __forceinline__ int wbc_crow(const WbcCtx &x, int i)
{
	const int ci = (i * 11) >> 5;
	return wbc_pick(x.clp, ci) * 3 + (i - 3 * ci);
}
__device__ __forceinline__ int wbc_srow(const WbcCtx &x, int i)
{
	const int ci = (i * 11) >> 5;
	return wbc_pick(x.slp, ci) * 3 + (i - 3 * ci);
}

// Entry (r, c) of the A block (levels 0, 2, 3, 4) or the D block (level 1).  r may be any row
// below the storage extent (rows beyond the instance's own count give zeros), 0 <= c < 32;
// jcoff = wbc_jcoff(x, c).
template <int LEVEL>
__device__ __forceinline__ Entry wbc_entry(const WbcCtx &x, const WbcPrm &prm, int r, int c, int jcoff)
{
	Entry e = {x.M, 0.0, 0.0};
	if (LEVEL == 0) {
		// ConstructFloatingBaseEomTask, :351-366:  A = [M_fb, -J_c,fb^T]
		if (c < kNU)
			e = {x.M + r * kNU + c, 1.0, 0.0};
		else if (jcoff >= 0)
			e = {x.Jf + jcoff + r, -1.0, 0.0};
		if (r >= kNB)
			e = {x.M, 0.0, 0.0};
	} else if (LEVEL == 1) {
		// ConstructJointTorqueTask :332-349 ++ ConstructFrictionConeTask :299-330
		if (r < 2 * kNJ) {
			const int rr = (r < kNJ) ? r : r - kNJ;
			const double sg = (r < kNJ) ? 1.0 : -1.0;
			if (c < kNU)
				e = {x.M + (kNB + rr) * kNU + c, sg, 0.0};
			else if (jcoff >= 0)
				e = {x.Jf + jcoff + kNB + rr, -sg, 0.0};
		} else if (r < 2 * kNJ + 4 * x.nc) {
			const int i = (r - 2 * kNJ) >> 2, t = (r - 2 * kNJ) & 3;
			// Q1 (:321-324): the pyramid lands on columns 6 + 3 leg (joint accelerations)
			const int col = prm.friction_quirk ? kNB + wbc_pick(x.clp, i) * 3 : kNU + 3 * i;
			const int d = c - col;
			if (d == 2)
				e.add = -prm.mu;
			else if (d == 0 && t < 2)
				e.add = (t == 0) ? 1.0 : -1.0;
			else if (d == 1 && t >= 2)
				e.add = (t == 2) ? 1.0 : -1.0;
		}
	} else if (LEVEL == 2) {
		// ConstructNoContactMotionTask, :283-297:  A = [J_c 0]
		if (r < 3 * x.nc && c < kNU)
			e = {x.Jf + wbc_crow(x, r) * kNU + c, 1.0, 0.0};
	} else if (LEVEL == 3) {
		// base pos :217-247 (J_b rows 3..5), base rot :249-281 (rows 0..2), swing legs :186-215
		if (c < kNU) {
			if (r < 6) {
				const int rb = (r < 3) ? 3 + r : r - 3;
				if (x.Jb)
					e = {x.Jb + rb * kNU + c, 1.0, 0.0};
				else
					e.add = (rb == c) ? 1.0 : 0.0;
			} else if (r < 6 + 3 * x.ns) {
				e = {x.Jf + wbc_srow(x, r - 6) * kNU + c, 1.0, 0.0};
			}
		}
	} else {
		// ConstructForceMinimizationTask, :385-400:  A = [0 I]
		e.add = (r < 3 * x.nc && c == kNU + r) ? 1.0 : 0.0;
	}
	return e;
}

__device__ __forceinline__ double entry_value(const Entry &e, double loaded) { return (e.scale != 0.0) ? e.scale * loaded : e.add; }

// Right-hand side b_r / f_r of levels 0, 1, 2, 4 in the same form (level 3 needs the swing-foot
// velocities J_swing u, see wbc_rhs3).
template <int LEVEL>
__device__ __forceinline__ Entry wbc_rhs_entry(const WbcCtx &x, const WbcPrm &prm, int r)
{
	Entry e = {x.h, 0.0, 0.0};
	if (LEVEL == 0) {
		if (r < kNB)
			e = {x.h + r, -1.0, 0.0};  // b = -h_fb, :362
	} else if (LEVEL == 1) {
		// value = add + scale * load here (both parts used), see wbc_rhs_value
		if (r < kNJ)
			e = {x.h + kNB + r, -1.0, prm.tau_max};  // tau_max - h_j, :343
		else if (r < 2 * kNJ)
			e = {x.h + kNB + r - kNJ, 1.0, -prm.tau_min};  // -(tau_min - h_j), :344
	} else if (LEVEL == 2) {
		if (r < 3 * x.nc)
			e = {x.Jdu + wbc_crow(x, r), -1.0, 0.0};  // b = -J_c_dot u, :293
	}
	return e;
}
// (fma with scale = +-1 is exact: the same value as the subtraction in the reference's builders)
__device__ __forceinline__ double rhs_value(const Entry &e, double loaded) { return fma(e.scale, loaded, e.add); }

// b_r of the tracking level (3): base position :226-237 (Q5: the stray ';' at :236 drops the
// position gain), base rotation :261-271, swing legs :197-207.  One row per calling lane.
__device__ __forceinline__ double wbc_rhs3(const WbcCtx &x, const WbcPrm &prm, int r)
{
	double v = 0.0;
	if (r < 3) {
		v = x.bc[6 + r] + prm.kv_base * (x.bc[3 + r] - x.u[3 + r]);
		if (!prm.drop_base_gain)
			v += prm.kp_base * (x.bc[r] - x.q[4 + r]);
	} else if (r < 6) {
		v = prm.kv_rot * (0.0 - x.u[r - 3]);
	} else if (r < 6 + 3 * x.ns) {
		const int li = wbc_srow(x, r - 6);
		const double *J = x.Jf + li * kNU;
		double vel = 0.0;
		for (int c = 0; c < kNU; ++c)
			vel = fma(J[c], x.u[c], vel);
		v = x.lc[24 + li] + prm.kp_swing * (x.lc[li] - x.fp[li]) + prm.kv_swing * (x.lc[12 + li] - vel);
	}
	return v;
}

}  // namespace hoqp
