// shim_check.cpp -- drives include/hoqp.hpp the way the reference drives its own classes
// (ConstructOptProblems, ho_qp_controller.cpp:120-141; CalcJointCmd, :7-25).  Used by
// tests/test_gpu_shim.py: reads a text fixture, prints results as text.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "hoqp.hpp"

using hoqp::MatrixXd;
using hoqp::VectorXd;

static MatrixXd read_mat(std::istream &in, int r, int c)
{
	MatrixXd m(r, c);
	for (int i = 0; i < r; ++i)
		for (int j = 0; j < c; ++j) {
			double v;
			in >> v;
			m(i, j) = v;
		}
	return m;
}
static VectorXd read_vec(std::istream &in, int n)
{
	VectorXd v(n);
	for (int i = 0; i < n; ++i) {
		double t;
		in >> t;
		v(i) = t;
	}
	return v;
}
static void print_vec(const char *name, const VectorXd &v)
{
	std::printf("%s %d", name, (int)v.rows());
	for (int i = 0; i < v.rows(); ++i)
		std::printf(" %.17g", v(i));
	std::printf("\n");
}

static void print_mat(const char *name, const MatrixXd &m)
{
	std::printf("%s %d %d", name, (int)m.rows(), (int)m.cols());
	for (int i = 0; i < m.rows(); ++i)
		for (int j = 0; j < m.cols(); ++j)
			std::printf(" %.17g", m(i, j));
	std::printf("\n");
}

int main(int argc, char **argv)
{
	if (argc < 2)
		return 2;
	if (std::string(argv[1]) == "--validate") {
		// constructor-time validation of include/hoqp.hpp (no GPU needed: nothing is solved)
		int caught = 0;
		auto expect_throw = [&](TaskDefinition t, std::shared_ptr<control::HoQpProblem> prev) {
			try {
				control::HoQpProblem p(t, prev);
			} catch (const std::invalid_argument &) {
				++caught;
			}
		};
		TaskDefinition ok;
		ok.A = MatrixXd::Identity(3, 8);
		ok.b = VectorXd::Zero(3);
		std::shared_ptr<control::HoQpProblem> first(new control::HoQpProblem(ok));
		TaskDefinition t1 = ok;
		t1.b = VectorXd::Zero(2);  // A / b rows differ
		expect_throw(t1, nullptr);
		TaskDefinition t2;
		t2.D = MatrixXd::Identity(4, 8);
		t2.f = VectorXd::Zero(5);  // D / f rows differ
		expect_throw(t2, nullptr);
		TaskDefinition t3;
		t3.A = MatrixXd::Identity(3, HOQP_MAX_VARS + 1);  // too many variables
		t3.b = VectorXd::Zero(3);
		expect_throw(t3, nullptr);
		TaskDefinition t4;
		t4.A = MatrixXd::Identity(3, 9);  // column count differs from the parent's
		t4.b = VectorXd::Zero(3);
		expect_throw(t4, first);
		TaskDefinition t5;
		t5.D = MatrixXd::Identity(HOQP_MAX_INEQ_ROWS + 1, 8);  // too many inequality rows
		t5.f = VectorXd::Zero(HOQP_MAX_INEQ_ROWS + 1);
		expect_throw(t5, nullptr);
		const control::HoQpProblem &cref = *first;  // getters are const
		std::printf("validate caught %d of 5, stacked slack vars %d\n", caught, cref.GetStackedNumSlackVars());
		return caught == 5 ? 0 : 1;
	}
	std::ifstream in(argv[1]);
	std::string kind;
	int semantics;
	in >> kind >> semantics;
	hoqp_config cfg;
	hoqp_default_config(&cfg);
	cfg.semantics = semantics;
	if (semantics == HOQP_SEMANTICS_PAPER)
		cfg.eps = 0.0;
	hoqp::configure(cfg);
	if (kind == "tasks") {
		int n, L;
		in >> n >> L;
		std::vector<TaskDefinition> tasks(L);
		for (int k = 0; k < L; ++k) {
			int m, p;
			in >> m >> p;
			if (m) {
				tasks[k].A = read_mat(in, m, n);
				tasks[k].b = read_vec(in, m);
			}
			if (p) {
				tasks[k].D = read_mat(in, p, n);
				tasks[k].f = read_vec(in, p);
			}
		}
		// exactly ConstructOptProblems, ho_qp_controller.cpp:120-141
		std::vector<std::shared_ptr<control::HoQpProblem>> opt_problems(tasks.size());
		opt_problems[0] = std::shared_ptr<control::HoQpProblem>(new control::HoQpProblem(tasks[0]));
		for (size_t i = 1; i < tasks.size(); ++i)
			opt_problems[i] = std::shared_ptr<control::HoQpProblem>(
				new control::HoQpProblem(tasks[i], opt_problems[i - 1]));
		for (size_t i = 0; i < tasks.size(); ++i) {
			std::printf("level %zu nz %d nslack %d status %d\n", i,
				    (int)opt_problems[i]->GetStackedNullspaceMatrix().cols(),
				    opt_problems[i]->GetStackedNumSlackVars(), opt_problems[i]->GetStatus());
			print_vec("x", opt_problems[i]->GetSolution());
			print_vec("v", opt_problems[i]->GetStackedSlackSolutions());
		}
		// GetStackedTasks by VALUE: StackTasks puts the current task first (ho_qp_problem.cpp:83-86 with
		// ConcatenateTasks(curr, prev), task_definition.hpp:13-30)
		TaskDefinition st = opt_problems.back()->GetStackedTasks();
		print_mat("stackedA", st.A);
		print_vec("stackedb", st.b);
		print_mat("stackedD", st.D);
		print_vec("stackedf", st.f);
		print_vec("vlast", opt_problems.back()->GetSlackSolutions());
		std::printf("stacked %d %d\n", (int)st.A.rows(), (int)st.D.rows());
	} else {
		std::vector<double> M(324), h(18), Jf(216), Jdu(12), fp(12), Jb(108);
		for (auto *v : {&M, &h, &Jf, &Jdu, &fp, &Jb})
			for (auto &e : *v)
				in >> e;
		VectorXd q = read_vec(in, 19), u = read_vec(in, 18);
		VectorXd bc = read_vec(in, 3), bd = read_vec(in, 3), bdd = read_vec(in, 3);
		VectorXd lc = read_vec(in, 12), ld = read_vec(in, 12), ldd = read_vec(in, 12), cp = read_vec(in, 4);
		control::HoQpController ctl;
		ctl.SetModelDynamics(M.data(), h.data(), Jf.data(), Jdu.data(), fp.data(), Jb.data());
		ctl.SetBaseCmd(bc, bd, bdd);
		ctl.SetLegCmd(lc, ld, ldd, cp);
		ctl.CalcJointCmd(q, u);
		print_vec("qdd", ctl.GetJointAccelerationCmd());
		print_vec("tau", ctl.GetJointTorqueCmd());
		print_vec("x", ctl.GetSolution());
		const int *fl = ctl.GetSolutionValidFlags();
		std::printf("flags %d %d %d %d %d status %d\n", fl[0], fl[1], fl[2], fl[3], fl[4], ctl.GetStatus());
		// the same cycle with the dynamics terms produced from (q, u) by the library
		control::HoQpController ctl2;
		ctl2.UseBuiltinDynamics(nullptr, 1);  // the fixture's h includes gravity
		ctl2.SetBaseCmd(bc, bd, bdd);
		ctl2.SetLegCmd(lc, ld, ldd, cp);
		ctl2.CalcJointCmd(q, u);
		print_vec("tau_builtin", ctl2.GetJointTorqueCmd());
		// single-state drop-in latency: the reference's control loop calls CalcJointCmd once per
		// 2.5 ms cycle (whole_body_controller.cpp:97-107, :609); argv[2] = repetitions
		const int reps = argc > 2 ? std::atoi(argv[2]) : 0;
		if (reps > 0) {
			std::vector<double> us(reps), us2(reps), us3(reps);
			for (int w = 0; w < 200; ++w)
				ctl.CalcJointCmd(q, u);
			for (int i = 0; i < reps; ++i) {
				auto t0 = std::chrono::steady_clock::now();
				ctl.CalcJointCmd(q, u);
				us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
			}
			for (int i = 0; i < reps; ++i) {  // with the dynamics terms produced from (q, u) as well
				auto t0 = std::chrono::steady_clock::now();
				ctl2.CalcJointCmd(q, u);
				us2[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
			}
			// the reference's own call pattern on the task API: a chain of HoQpProblem constructors, one GetSolution
			hoqp_config cur = hoqp::default_config();
			std::vector<TaskDefinition> tk(3);
			tk[0].A = MatrixXd::Identity(6, 30);
			tk[0].b = VectorXd::Zero(6);
			tk[1].D = MatrixXd::Identity(8, 30);
			tk[1].f = VectorXd::Zero(8);
			tk[2].A = MatrixXd::Identity(30, 30);
			tk[2].b = VectorXd::Zero(30);
			(void)cur;
			for (int i = 0; i < reps; ++i) {
				auto t0 = std::chrono::steady_clock::now();
				std::shared_ptr<control::HoQpProblem> pr(new control::HoQpProblem(tk[0]));
				for (size_t k = 1; k < tk.size(); ++k)
					pr = std::shared_ptr<control::HoQpProblem>(new control::HoQpProblem(tk[k], pr));
				volatile double x0 = pr->GetSolution()(0);
				(void)x0;
				us3[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
			}
			auto pct = [](std::vector<double> v, double p) {
				std::sort(v.begin(), v.end());
				return v[(size_t)(p * (v.size() - 1))];
			};
			std::printf("latency_us calcjointcmd median %.2f p99 %.2f max %.2f builtin_dynamics median %.2f p99 %.2f "
				    "problem_chain median %.2f p99 %.2f reps %d\n",
				    pct(us, 0.5), pct(us, 0.99), pct(us, 1.0), pct(us2, 0.5), pct(us2, 0.99), pct(us3, 0.5),
				    pct(us3, 0.99), reps);
		}
	}
	return 0;
}
