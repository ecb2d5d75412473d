// hoqp.hpp -- C++ host side above the C ABI (include/hoqp.h): the reference's own class API
// for the HO-QP solve step, so that code written against
//   struct TaskDefinition                     include/control/ho_qp/task_definition.hpp:5-11
//   ConcatenateTasks                          include/control/ho_qp/task_definition.hpp:13-41
//   control::HoQpProblem                      include/control/ho_qp/ho_qp_problem.hpp:20-43
//   control::HoQpController                   include/control/ho_qp/ho_qp_controller.hpp:21-44
// keeps compiling and now solves on the GPU.  Header-only, needs only libhoqp_hip.so.
// Eigen is used when available (<Eigen/Core>), otherwise the minimal column-major
// hoqp::MatrixXd / hoqp::VectorXd below stand in (Eigen is not part of this image).
#pragma once

#include <algorithm>
#include <cassert>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "hoqp.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && !defined(HOQP_NO_EIGEN)
#include <Eigen/Core>
#define HOQP_HAVE_EIGEN 1
#endif
#endif

namespace hoqp {

#ifdef HOQP_HAVE_EIGEN
using MatrixXd = Eigen::MatrixXd;
using VectorXd = Eigen::VectorXd;
#else
// Column-major dense matrix with the few members of Eigen::MatrixXd the path uses.
class MatrixXd {
public:
	MatrixXd() = default;
	MatrixXd(int r, int c) : r_(r), c_(c), d_((size_t)r * c, 0.0) {}
	static MatrixXd Zero(int r, int c) { return MatrixXd(r, c); }
	static MatrixXd Identity(int r, int c)
	{
		MatrixXd m(r, c);
		for (int i = 0; i < r && i < c; ++i)
			m(i, i) = 1.0;
		return m;
	}
	int rows() const { return r_; }
	int cols() const { return c_; }
	double &operator()(int i, int j) { return d_[(size_t)j * r_ + i]; }
	double operator()(int i, int j) const { return d_[(size_t)j * r_ + i]; }
	double *data() { return d_.data(); }
	const double *data() const { return d_.data(); }
	void resize(int r, int c)
	{
		r_ = r;
		c_ = c;
		d_.assign((size_t)r * c, 0.0);
	}

private:
	int r_ = 0, c_ = 0;
	std::vector<double> d_;
};

class VectorXd {
public:
	VectorXd() = default;
	explicit VectorXd(int n) : d_(n, 0.0) {}
	static VectorXd Zero(int n) { return VectorXd(n); }
	int rows() const { return (int)d_.size(); }
	int cols() const { return 1; }
	int size() const { return (int)d_.size(); }
	double &operator()(int i) { return d_[i]; }
	double operator()(int i) const { return d_[i]; }
	double &operator[](int i) { return d_[i]; }
	double operator[](int i) const { return d_[i]; }
	double *data() { return d_.data(); }
	const double *data() const { return d_.data(); }
	void resize(int n) { d_.assign(n, 0.0); }

private:
	std::vector<double> d_;
};
#endif

// One lazily created handle per host thread (the reference is single-threaded on this
// path, whole_body_controller.cpp:329-342); handles share no state.
struct Handle {
	hoqp_handle *h = nullptr;
	hoqp_config cfg;
	explicit Handle(const hoqp_config &c) : cfg(c)
	{
		int rc = hoqp_create(&cfg, &h);
		if (rc != 0)
			throw std::runtime_error("hoqp_create failed: " + std::to_string(rc));
	}
	~Handle()
	{
		if (h)
			hoqp_destroy(h);
	}
	Handle(const Handle &) = delete;
	Handle &operator=(const Handle &) = delete;
};

inline hoqp_config &default_config()
{
	static thread_local hoqp_config cfg = [] {
		hoqp_config c;
		hoqp_default_config(&c);
		return c;
	}();
	return cfg;
}

inline std::shared_ptr<Handle> &thread_handle()
{
	static thread_local std::shared_ptr<Handle> h;
	if (!h)
		h = std::make_shared<Handle>(default_config());
	return h;
}

// Re-create the thread's handle with other semantics / constants (e.g. HOQP_SEMANTICS_PAPER).
inline void configure(const hoqp_config &cfg)
{
	default_config() = cfg;
	thread_handle() = std::make_shared<Handle>(cfg);
}

inline void check(Handle &h, int rc, const char *what)
{
	if (rc != 0)
		throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " +
					 hoqp_last_error_string(h.h));
}

inline std::vector<double> row_major(const MatrixXd &m)
{
	std::vector<double> out((size_t)m.rows() * m.cols());
	for (int i = 0; i < m.rows(); ++i)
		for (int j = 0; j < m.cols(); ++j)
			out[(size_t)i * m.cols() + j] = m(i, j);
	return out;
}

}  // namespace hoqp

// ---------------------------------------------------------------------------
// task_definition.hpp:5-41
// ---------------------------------------------------------------------------
struct TaskDefinition {
	hoqp::MatrixXd A;
	hoqp::VectorXd b;
	hoqp::MatrixXd D;
	hoqp::VectorXd f;
};

namespace hoqp {
// ConcatenateMatrices (math.hpp:22-37): a 0-column operand is skipped.
inline MatrixXd ConcatenateMatrices(const MatrixXd &m1, const MatrixXd &m2)
{
	if (m1.cols() == 0)
		return m2;
	if (m2.cols() == 0)
		return m1;
	assert(m1.cols() == m2.cols());
	MatrixXd r(m1.rows() + m2.rows(), m1.cols());
	for (int j = 0; j < m1.cols(); ++j) {
		for (int i = 0; i < m1.rows(); ++i)
			r(i, j) = m1(i, j);
		for (int i = 0; i < m2.rows(); ++i)
			r(m1.rows() + i, j) = m2(i, j);
	}
	return r;
}
// ConcatenateVectors (math.hpp:39-54)
inline VectorXd ConcatenateVectors(const VectorXd &v1, const VectorXd &v2)
{
	VectorXd r((int)(v1.rows() + v2.rows()));
	for (int i = 0; i < v1.rows(); ++i)
		r(i) = v1(i);
	for (int i = 0; i < v2.rows(); ++i)
		r((int)v1.rows() + i) = v2(i);
	return r;
}
}  // namespace hoqp

inline TaskDefinition ConcatenateTasks(const TaskDefinition &t1, const TaskDefinition &t2)
{
	return TaskDefinition{hoqp::ConcatenateMatrices(t1.A, t2.A), hoqp::ConcatenateVectors(t1.b, t2.b),
			      hoqp::ConcatenateMatrices(t1.D, t2.D), hoqp::ConcatenateVectors(t1.f, t2.f)};
}

inline TaskDefinition ConcatenateTasks(const std::vector<TaskDefinition> &tasks)
{
	TaskDefinition res = tasks[0];
	for (size_t i = 1; i < tasks.size(); ++i)
		res = ConcatenateTasks(res, tasks[i]);
	return res;
}

namespace control {

// ---------------------------------------------------------------------------
// ho_qp_problem.hpp:20-43.  The reference solves in the constructor; here a level solves its whole
// ancestor chain in ONE kernel launch (the chain state never leaves LDS) the first time one of its
// results is asked for.  HoQpController::ConstructOptProblems (ho_qp_controller.cpp:120-141) builds
// the chain of five and only reads the last problem (:17), so the cycle costs one upload and one
// launch instead of five of each (an eager solve per constructor would redo every ancestor:
// O(L^2) work).  Getters of an intermediate level still work: they solve that level's own chain.
// ---------------------------------------------------------------------------
class HoQpProblem {
public:
	explicit HoQpProblem(TaskDefinition &new_task) : HoQpProblem(new_task, std::shared_ptr<HoQpProblem>(nullptr)) {}

	HoQpProblem(TaskDefinition &new_task, std::shared_ptr<HoQpProblem> higher_pri_problem)
		: curr_task_(new_task), higher_pri_problem_(higher_pri_problem)
	{
		// The solve itself is lazy (below), but malformed input fails HERE, where the reference's constructor
		// would have failed (dimension asserts of math.hpp:31, 48 and ho_qp_problem.cpp:108-114), not at some
		// later getter on another thread.
		const TaskDefinition &t = curr_task_;
		if (t.A.rows() != t.b.rows() || t.D.rows() != t.f.rows())
			throw std::invalid_argument("HoQpProblem: A/b or D/f row counts differ");
		if (t.A.rows() > HOQP_MAX_EQ_ROWS || t.D.rows() > HOQP_MAX_INEQ_ROWS)
			throw std::invalid_argument("HoQpProblem: more rows than HOQP_MAX_EQ_ROWS / HOQP_MAX_INEQ_ROWS");
		if (t.A.rows() > 0 && t.D.rows() > 0 && t.A.cols() != t.D.cols())
			throw std::invalid_argument("HoQpProblem: A and D have different column counts");
		const int n = (int)std::max(t.A.cols(), t.D.cols());
		if (n > HOQP_MAX_VARS)
			throw std::invalid_argument("HoQpProblem: more than HOQP_MAX_VARS decision variables");
		if (higher_pri_problem_)
			chain_ = higher_pri_problem_->chain_;
		int P = (int)t.D.rows();
		for (auto &c : chain_) {
			const int nc = (int)std::max(c.A.cols(), c.D.cols());
			if (n > 0 && nc > 0 && nc != n)
				throw std::invalid_argument("HoQpProblem: column count differs from the higher-priority problem's");
			P += (int)c.D.rows();
		}
		if (P > HOQP_MAX_INEQ_ROWS)
			throw std::invalid_argument("HoQpProblem: more than HOQP_MAX_INEQ_ROWS inequality rows in the chain");
		chain_.push_back(curr_task_);
		if ((int)chain_.size() > HOQP_MAX_LEVELS)
			throw std::invalid_argument("HoQpProblem: more than HOQP_MAX_LEVELS levels");
	}

	TaskDefinition GetStackedTasks() const
	{
		// StackTasks: ConcatenateTasks(curr, prev) -- newest first (ho_qp_problem.cpp:83-86)
		TaskDefinition st = chain_.back();
		for (int k = (int)chain_.size() - 2; k >= 0; --k)
			st = ConcatenateTasks(st, chain_[k]);
		return st;
	}
	hoqp::MatrixXd GetStackedNullspaceMatrix() const { return Solved().stacked_Z_; }
	int GetStackedNumSlackVars() const
	{
		int P = 0;  // known without solving
		for (auto &t : chain_)
			P += (int)t.D.rows();
		return P;
	}
	hoqp::VectorXd GetSolution() const { return Solved().x_; }
	hoqp::VectorXd GetSlackSolutions() const { return Solved().slack_; }
	hoqp::VectorXd GetStackedSlackSolutions() const { return Solved().stacked_slack_; }
	int GetStatus() const { return Solved().status_; }

private:
	// (the results are a cache of a pure function of the chain: mutable, so that the getters stay const)
	const HoQpProblem &Solved() const
	{
		if (!solved_) {
			Solve();
			solved_ = true;
		}
		return *this;
	}

	void Solve() const
	{
		auto hd = hoqp::thread_handle();
		const int L = (int)chain_.size();
		if (L > HOQP_MAX_LEVELS)
			throw std::runtime_error("HoQpProblem: more than HOQP_MAX_LEVELS levels");
		int n = 0;
		for (auto &t : chain_)
			n = std::max(n, (int)std::max(t.A.cols(), t.D.cols()));  // ho_qp_problem.cpp:112-114
		hoqp_task_batch tb{};
		tb.batch = 1;
		tb.n = n;
		tb.num_levels = L;
		std::vector<std::vector<double>> keep;
		int P = 0;
		for (int k = 0; k < L; ++k) {
			const TaskDefinition &t = chain_[k];
			tb.m[k] = (int)t.A.rows();
			tb.p[k] = (int)t.D.rows();
			if (tb.m[k]) {
				keep.push_back(hoqp::row_major(t.A));
				tb.A[k] = keep.back().data();
				tb.b[k] = t.b.data();
			}
			if (tb.p[k]) {
				keep.push_back(hoqp::row_major(t.D));
				tb.D[k] = keep.back().data();
				tb.f[k] = t.f.data();
			}
			P += tb.p[k];
		}
		std::vector<double> xl((size_t)L * n), v(P + 1), Zl((size_t)L * n * n);
		std::vector<int> nz(L);
		hoqp_result_batch rb{};
		rb.x_levels = xl.data();
		rb.v = v.data();
		rb.Z_levels = Zl.data();
		rb.nz_levels = nz.data();
		rb.status = &status_;
		hoqp::check(*hd, hoqp_solve_tasks_host(hd->h, &tb, &rb), "hoqp_solve_tasks_host");
		x_.resize(n);
		for (int i = 0; i < n; ++i)
			x_(i) = xl[(size_t)(L - 1) * n + i];
		int nzk = nz[L - 1];
		const bool has_eq = chain_.back().A.rows() > 0;
		if (nzk == 0 && has_eq && (hd->cfg.semantics & HOQP_Q6_ZERO_KERNEL_COL)) {
			stacked_Z_ = hoqp::MatrixXd::Zero(n, 1);  // Eigen kernel() of a trivial kernel (Q6)
		} else {
			stacked_Z_ = hoqp::MatrixXd::Zero(n, nzk);
			const double *Zk = Zl.data() + (size_t)(L - 1) * n * n;
			for (int i = 0; i < n; ++i)
				for (int j = 0; j < nzk; ++j)
					stacked_Z_(i, j) = Zk[(size_t)i * n + j];
		}
		const int pk = tb.p[L - 1];
		slack_.resize(pk);
		stacked_slack_.resize(P);
		for (int i = 0; i < P; ++i)
			stacked_slack_(i) = v[i];  // oldest first (ho_qp_problem.cpp:88-101)
		for (int i = 0; i < pk; ++i)
			slack_(i) = v[P - pk + i];
		num_stacked_slack_ = P;
	}

	TaskDefinition curr_task_;
	std::shared_ptr<HoQpProblem> higher_pri_problem_;
	std::vector<TaskDefinition> chain_;
	mutable hoqp::VectorXd x_, slack_, stacked_slack_;
	mutable hoqp::MatrixXd stacked_Z_;
	mutable int num_stacked_slack_ = 0;
	mutable int status_ = 0;
	mutable bool solved_ = false;
};

// ---------------------------------------------------------------------------
// ho_qp_controller.hpp:21-44.  The Dynamics member (Drake MultibodyPlant, out of scope)
// is replaced by SetModelDynamics, which takes exactly what UpdateModelDynamics
// (ho_qp_controller.cpp:89-114) pulls from it.
// ---------------------------------------------------------------------------
class HoQpController {
public:
	HoQpController() : base_cmd_(9, 0.0), leg_cmd_(36, 0.0), q_(19, 0.0), u_(18, 0.0)
	{
		for (int i = 0; i < 4; ++i)
			contact_[i] = 1;
	}

	void SetBaseCmd(const hoqp::VectorXd &base_cmd, const hoqp::VectorXd &base_dot_cmd,
			const hoqp::VectorXd &base_ddot_cmd)
	{
		for (int i = 0; i < 3; ++i) {
			base_cmd_[i] = base_cmd(i);
			base_cmd_[3 + i] = base_dot_cmd(i);
			base_cmd_[6 + i] = base_ddot_cmd(i);
		}
	}

	void SetLegCmd(const hoqp::VectorXd &leg_cmd, const hoqp::VectorXd &leg_dot_cmd,
		       const hoqp::VectorXd &leg_ddot_cmd, const hoqp::VectorXd &contact_pattern_cmd)
	{
		for (int i = 0; i < 12; ++i) {
			leg_cmd_[i] = leg_cmd(i);
			leg_cmd_[12 + i] = leg_dot_cmd(i);
			leg_cmd_[24 + i] = leg_ddot_cmd(i);
		}
		for (int i = 0; i < 4; ++i)  // == 1 contact, == 0 swing (ho_qp_controller.cpp:426-450)
			contact_[i] = contact_pattern_cmd(i) == 1 ? 1 : (contact_pattern_cmd(i) == 0 ? 0 : -1);
	}

	// Row-major host arrays: M[18*18], h[18], Jfoot[4*3*18], Jdotu[12], footpos[12], Jb[6*18] (or null).
	void SetModelDynamics(const double *M, const double *h, const double *Jfoot, const double *Jdotu,
			      const double *footpos, const double *Jb)
	{
		M_.assign(M, M + 324);
		h_.assign(h, h + 18);
		Jfoot_.assign(Jfoot, Jfoot + 216);
		Jdotu_.assign(Jdotu, Jdotu + 12);
		footpos_.assign(footpos, footpos + 12);
		if (Jb)
			Jb_.assign(Jb, Jb + 108);
		else
			Jb_.clear();
	}

	// Instead of SetModelDynamics: let CalcJointCmd produce M, h, J, ... from (q, u) itself, as
	// the reference's UpdateModelDynamics does (ho_qp_controller.cpp:89-114), with the library's
	// quadruped model (hoqp_wbc_dynamics_host; model == nullptr: the default parameters).
	// include_gravity < 0 (default): follow the handle's semantics -- under HOQP_SEMANTICS_REFERENCE
	// h is the reference's CalcBiasTerm, which carries NO gravity (dynamics.cpp:82-87); under any other
	// semantics gravity is part of h.  0 / 1 force it off / on.
	void UseBuiltinDynamics(const hoqp_quadruped_model *model = nullptr, int include_gravity = -1)
	{
		builtin_ = true;
		if (include_gravity < 0)
			include_gravity_ = hoqp::thread_handle()->cfg.semantics != HOQP_SEMANTICS_REFERENCE;
		else
			include_gravity_ = include_gravity != 0;
		if (model)
			model_ = *model;
		else
			hoqp_default_quadruped_model(&model_);
	}

	void CalcJointCmd(const hoqp::VectorXd &q, const hoqp::VectorXd &u)
	{
		for (int i = 0; i < 19; ++i)
			q_[i] = q(i);
		for (int i = 0; i < 18; ++i)
			u_[i] = u(i);
		auto hd = hoqp::thread_handle();
		if (builtin_) {
			M_.resize(324), h_.resize(18), Jfoot_.resize(216), Jdotu_.resize(12), footpos_.resize(12), Jb_.resize(108);
			hoqp::check(*hd, hoqp_wbc_dynamics_host(hd->h, &model_, 1, q_.data(), u_.data(), include_gravity_ ? 1 : 0,
							       M_.data(), h_.data(), Jfoot_.data(), Jdotu_.data(),
							       footpos_.data(), Jb_.data()),
				    "hoqp_wbc_dynamics_host");
		}
		if (M_.empty())
			throw std::runtime_error("HoQpController: neither SetModelDynamics nor UseBuiltinDynamics was called");
		hoqp_wbc_inputs in{};
		in.batch = 1;
		in.M = M_.data();
		in.h = h_.data();
		in.Jfoot = Jfoot_.data();
		in.Jdotu = Jdotu_.data();
		in.footpos = footpos_.data();
		in.Jb = Jb_.empty() ? nullptr : Jb_.data();
		in.q = q_.data();
		in.u = u_.data();
		in.base_cmd = base_cmd_.data();
		in.leg_cmd = leg_cmd_.data();
		in.contact = contact_;
		double x[30], qdd[12], tau[12];
		hoqp_wbc_outputs out{};
		out.x = x;
		out.qdd = qdd;
		out.tau = tau;
		out.flags = flags_;
		out.status = &status_;
		hoqp::check(*hd, hoqp_solve_wbc_host(hd->h, &in, &out), "hoqp_solve_wbc_host");
		int nc = 0;
		for (int i = 0; i < 4; ++i)
			nc += contact_[i] == 1;
		solution_.resize(18 + 3 * nc);
		for (int i = 0; i < 18 + 3 * nc; ++i)
			solution_(i) = x[i];
		q_j_ddot_cmd_.resize(12);
		tau_cmd_.resize(12);
		for (int i = 0; i < 12; ++i) {
			q_j_ddot_cmd_(i) = qdd[i];
			tau_cmd_(i) = tau[i];
		}
	}

	hoqp::VectorXd GetJointAccelerationCmd() { return q_j_ddot_cmd_; }
	hoqp::VectorXd GetJointTorqueCmd() { return tau_cmd_; }
	hoqp::VectorXd GetSolution() { return solution_; }
	// the five CheckSolutionValid outcomes of CalcJointCmd (ho_qp_controller.cpp:18-22)
	const int *GetSolutionValidFlags() const { return flags_; }
	int GetStatus() const { return status_; }

private:
	std::vector<double> M_, h_, Jfoot_, Jdotu_, footpos_, Jb_, base_cmd_, leg_cmd_, q_, u_;
	int contact_[4];
	bool builtin_ = false, include_gravity_ = true;
	hoqp_quadruped_model model_{};
	int flags_[5] = {0, 0, 0, 0, 0};
	int status_ = 0;
	hoqp::VectorXd solution_, q_j_ddot_cmd_, tau_cmd_;
};

}  // namespace control
