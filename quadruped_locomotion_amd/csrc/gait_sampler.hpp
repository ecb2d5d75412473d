// the command source of the whole-body controller as a batch sampler -- contact
// pattern and swing-leg commands of LegPlanner (leg_planner.cpp:105-151, 209-299, 302-366) under a
// gait schedule of MotionPlanner (motion_planner.cpp:536-575), and the stand-up ramp of BasePlanner
// (base_planner.cpp:272-317).  One thread per (state, leg): a scan over <= 32 gait steps and a few
// polynomial evaluations; the outputs are exactly the solve's `leg_cmd`, `base_cmd` and `contact`.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/hoqp.h"

namespace hoqp {

__global__ void gait_sample_kernel(hoqp_gait g, int batch, const double *__restrict__ time,
				   const double *__restrict__ vel_cmd, const double *__restrict__ stance0,
				   double *__restrict__ leg_cmd, int *__restrict__ contact)
{
	const int tid = blockIdx.x * blockDim.x + threadIdx.x;
	const int i = tid >> 2, leg = tid & 3;
	if (i >= batch)
		return;
	const double step_time = g.duration / (double)g.n_steps;
	// time is device data: a negative time (the reference's ros::Time differences never are) is taken
	// as 0 instead of giving a negative gait step (a shift by a negative count below)
	const double t_rel = fmax(0.0, fmod(time[i], g.duration));
	int step = (int)(t_rel / step_time);  // GetGaitStepFromTime, leg_planner.cpp:393-399
	step = step > g.n_steps - 1 ? g.n_steps - 1 : (step < 0 ? 0 : step);
	const unsigned bits = g.contact_bits[leg];
	const double vx = vel_cmd[2 * i] * step_time, vy = vel_cmd[2 * i + 1] * step_time;
	double px = stance0[(4 * i + leg) * 2], py = stance0[(4 * i + leg) * 2 + 1];
	double t_lo = -1.0, t_td = -1.0, sx = 0.0, sy = 0.0, ex = 0.0, ey = 0.0;
	int last = bits & 1u;
	for (int s = 1; s < g.n_steps; ++s) {
		const int cur = (bits >> s) & 1u;
		const double qx = px, qy = py;  // stance of the step before
		if (!cur) {                     // GenerateStanceForNextTimestep, :124-151
			px += vx;
			py += vy;
		}
		if (cur != last) {  // CreateLegMotionForLeg, :264-299
			if (last) {
				t_lo = step_time * s;
				sx = qx;
				sy = qy;
			} else {
				t_td = step_time * s;
				ex = qx;
				ey = qy;
			}
			last = cur;
		}
	}
	double p[3] = {0.0, 0.0, 0.0}, v[3] = {0.0, 0.0, 0.0}, a[3] = {0.0, 0.0, 0.0};
	if (!(t_rel < t_lo || t_rel > t_td)) {
		const double T = t_td - t_lo, half = 0.5 * fabs(T);
		const double w = (t_rel - t_lo) / T;
		const double u = (t_rel - (t_lo + half)) / half;
		p[0] = sx + w * (ex - sx);  // first-order hold, :209-228
		p[1] = sy + w * (ey - sy);
		p[2] = g.apex_height * (1.0 - u * u);  // three-knot spline == parabola, :230-255
		if (g.fix_derivatives) {
			v[0] = (ex - sx) / T;
			v[1] = (ey - sy) / T;
			v[2] = -2.0 * g.apex_height * u / half;
			a[2] = -2.0 * g.apex_height / (half * half);
		} else {  // the reference evaluates the position polynomials again, :337-341, :358-362
			for (int d = 0; d < 3; ++d)
				v[d] = a[d] = p[d];
		}
	}
	double *o = leg_cmd + (size_t)i * 36 + leg * 3;
	for (int d = 0; d < 3; ++d) {
		o[d] = p[d];
		o[12 + d] = v[d];
		o[24 + d] = a[d];
	}
	contact[4 * i + leg] = (bits >> step) & 1u;  // GetContactPatternAtT, :55-59
}

__global__ void standup_sample_kernel(int batch, double seconds, double target_height, const double *__restrict__ time,
				      const double *__restrict__ curr_pose, double *__restrict__ base_cmd,
				      int *__restrict__ contact)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= batch)
		return;
	const double t = fmax(time[i], 0.0);  // Drake's FirstOrderHold::value clamps to the first break
	for (int d = 0; d < 3; ++d) {
		const double c = curr_pose[3 * i + d];
		const double tgt = (d == 2) ? target_height : c;
		const double slope = (tgt - c) / seconds;
		base_cmd[9 * i + d] = (t > seconds) ? tgt : c + slope * t;
		base_cmd[9 * i + 3 + d] = (t > seconds) ? 0.0 : slope;
		base_cmd[9 * i + 6 + d] = 0.0;
	}
	if (contact)
		for (int leg = 0; leg < 4; ++leg)
			contact[4 * i + leg] = 1;
}

// BasePlanner::PlanBaseWalkMotion + EvalWalkTraj{Pos,Vel,Acc}AtT (base_planner.cpp:7-19,
// 100-142, 162-266).  The reference's program -- n quintic 2-D segments of unit duration, sampled
// acceleration cost, position and velocity continuity, fixed end points -- has a zero-cost feasible
// point, the straight line from the current position to the target at constant speed, and its KKT
// matrix is non-singular, so that line IS the solution (the tests solve the literal KKT system on the
// CPU and compare).  No per-instance factorization is left: segment k has
// x(t) = p0x + (k + t) dx, y(t) = p0y + (k + t) dy in the program's variable order
// alpha_k = [x coefficients; y coefficients].  The reference then reads the solution back as a
// column-major 2 x 6 matrix (GeneratePolynomials, :77-98): C(0, d) = alpha[2d], C(1, d) = alpha[2d+1],
// which turns the line into x = ax0 + ay0 t^3, y = ax1 + ay1 t^3; fix_layout = 1 evaluates the line.
__global__ void base_walk_sample_kernel(int batch, int n_segments, int fix_layout, double walking_height,
					const double *__restrict__ time, const double *__restrict__ curr_pos,
					const double *__restrict__ final_pos, double *__restrict__ base_cmd)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= batch)
		return;
	const double t = time[i];
	int k = 0;
	while (((double)k + 1.0 < t) && (k + 1 < n_segments))  // EvalWalkTrajAtT, :103-106
		++k;
	const double tin = t - (double)k;
	const double p0x = curr_pos[2 * i], p0y = curr_pos[2 * i + 1];
	const double dx = (final_pos[2 * i] - p0x) / (double)n_segments, dy = (final_pos[2 * i + 1] - p0y) / (double)n_segments;
	const double ax0 = p0x + (double)k * dx, ax1 = dx, ay0 = p0y + (double)k * dy, ay1 = dy;
	double *o = base_cmd + (size_t)i * 9;
	if (fix_layout) {
		o[0] = ax0 + ax1 * tin;
		o[1] = ay0 + ay1 * tin;
		o[3] = ax1;
		o[4] = ay1;
		o[6] = 0.0;
		o[7] = 0.0;
	} else {
		const double t2 = tin * tin;
		o[0] = ax0 + ay0 * t2 * tin;
		o[1] = ax1 + ay1 * t2 * tin;
		o[3] = 3.0 * ay0 * t2;
		o[4] = 3.0 * ay1 * t2;
		o[6] = 6.0 * ay0 * tin;
		o[7] = 6.0 * ay1 * tin;
	}
	o[2] = walking_height;  // :118, 123, 128
	o[5] = 0.0;
	o[8] = 0.0;
}

}  // namespace hoqp
