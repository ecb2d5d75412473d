"""CPU tests of the oracle (oracle/hoqp_oracle.c) -- known-answer tests K1..K10 of
SURVEY.md section 8c, cross-checks against numpy/scipy, and the committed golden fixtures.
The reference has no tests of its own: parity is unpinned (see oracle header)."""
import ast
import os

import numpy as np
import pytest
import scipy.linalg as sla
import scipy.optimize as sopt

import proto_levelsolver as pl
from oracle import anymal_model as am

HERE = os.path.dirname(os.path.abspath(__file__))


def _proj(Z):
    if Z.shape[1] == 0:
        return np.zeros((Z.shape[0], Z.shape[0]))
    Q, _ = np.linalg.qr(Z)
    return Q @ Q.T


def test_k1_single_equality_level(oracle):
    """K1: x = A^+ b, Z = null(A) (compare projectors; basis is FullPivLU, not orthonormal)."""
    rng = np.random.default_rng(1)
    A, b = rng.normal(size=(4, 9)), rng.normal(size=4)
    r = oracle.solve_tasks([dict(A=A, b=b)], 9)
    assert r["rc"] == 0 and r["nz_levels"][0] == 5
    assert np.allclose(r["x_levels"][0], np.linalg.pinv(A) @ b, atol=1e-9)
    assert np.allclose(A @ r["Z_levels"][0], 0, atol=1e-12)
    assert np.linalg.norm(_proj(r["Z_levels"][0]) - _proj(sla.null_space(A))) < 1e-8
    rh = oracle.solve_tasks([dict(A=A, b=b)], 9, oracle.default_cfg("reference", nullspace="householder"))
    assert np.linalg.norm(_proj(rh["Z_levels"][0]) - _proj(r["Z_levels"][0])) < 1e-8


def test_fullpivlu_kernel_matches_null_space(oracle):
    rng = np.random.default_rng(2)
    for (r, c, rank) in [(3, 7, 3), (6, 6, 4), (8, 5, 5), (5, 9, 2)]:
        A = rng.normal(size=(r, rank)) @ rng.normal(size=(rank, c))
        K = oracle.fullpivlu_kernel(A)
        if rank == c:
            assert K.shape == (c, 1) and np.all(K == 0)  # Q6: one zero column
            assert oracle.fullpivlu_kernel(A, zero_col=False).shape == (c, 0)
        else:
            assert K.shape == (c, c - rank)
            assert np.abs(A @ K).max() < 1e-10
            assert np.linalg.matrix_rank(K) == c - rank


def test_k2_conflicting_equalities(oracle):
    """K2: level-2 residual equals the projection residual, level-1 residual unchanged."""
    rng = np.random.default_rng(3)
    A1, b1 = rng.normal(size=(3, 5)), rng.normal(size=3)
    A2, b2 = rng.normal(size=(4, 5)), rng.normal(size=4)
    r = oracle.solve_tasks([dict(A=A1, b=b1), dict(A=A2, b=b2)], 5)
    x1, x2 = r["x_levels"]
    assert np.abs(A1 @ x1 - b1).max() < 1e-10 and np.abs(A1 @ x2 - b1).max() < 1e-10
    Z = sla.null_space(A1)
    z = np.linalg.lstsq(A2 @ Z, b2 - A2 @ x1, rcond=None)[0]
    assert np.allclose(np.linalg.norm(A2 @ x2 - b2), np.linalg.norm(A2 @ (x1 + Z @ z) - b2), atol=1e-9)
    assert np.allclose(x2, x1 + Z @ z, atol=1e-8)  # full column rank at the end -> unique


def test_k3_single_inequality_level(oracle):
    """K3: infeasible pair x <= -1, -x <= -1 -> v* = (1, 1) at x = 0; feasible case -> v* = 0."""
    D = np.array([[1.0], [-1.0]])
    r = oracle.solve_tasks([dict(D=D, f=np.array([-1.0, -1.0]))], 1)
    assert np.allclose(r["v_levels"][0], [1.0, 1.0], atol=1e-10)
    assert abs(r["x_levels"][0][0]) < 1e-10
    r = oracle.solve_tasks([dict(D=D, f=np.array([2.0, 1.0]))], 1)
    assert np.allclose(r["v_levels"][0], 0.0, atol=1e-12)


def test_k4_paper_vs_reference_gating(oracle):
    """K4 / Q2: an inequality level followed by an equality level that wants to violate it."""
    D, f = np.array([[1.0, 0.0]]), np.array([1.0])          # x0 <= 1
    A, b = np.array([[1.0, 0.0], [0.0, 1.0]]), np.array([3.0, 2.0])  # wants x = (3, 2)
    tasks = [dict(D=D, f=f), dict(A=A, b=b)]
    ref = oracle.solve_tasks(tasks, 2, oracle.default_cfg("reference"))
    pap = oracle.solve_tasks(tasks, 2, oracle.default_cfg("paper"))
    assert np.allclose(ref["x_levels"][-1], [3.0, 2.0], atol=1e-8)   # prior row silently dropped (Q2)
    assert np.allclose(pap["x_levels"][-1], [1.0, 2.0], atol=1e-8)   # prior row enforced


def test_k5_two_inequality_levels_with_slack(oracle):
    """K5: two inequality levels with non-zero slacks; paper mode against scipy SLSQP."""
    rng = np.random.default_rng(11)
    n = 4
    D1, f1 = rng.normal(size=(5, n)), rng.normal(size=5) - 1.0
    D2, f2 = rng.normal(size=(4, n)), rng.normal(size=4) - 1.0
    A3, b3 = rng.normal(size=(4, n)), rng.normal(size=4)
    r = oracle.solve_tasks([dict(D=D1, f=f1), dict(D=D2, f=f2), dict(A=A3, b=b3)], n, oracle.default_cfg("paper"))
    assert r["rc"] == 0
    # level 1: min |(D1 x - f1)_+|^2
    res1 = sopt.minimize(lambda x: 0.5 * np.sum(np.maximum(D1 @ x - f1, 0) ** 2), np.zeros(n),
                         jac=lambda x: D1.T @ np.maximum(D1 @ x - f1, 0), method="BFGS", options=dict(gtol=1e-12))
    v1 = np.maximum(D1 @ res1.x - f1, 0)
    assert np.allclose(r["v_levels"][0], v1, atol=1e-6)
    # level 2 subject to level 1's relaxed rows
    cons = [dict(type="ineq", fun=lambda x: f1 + r["v_levels"][0] - D1 @ x, jac=lambda x: -D1)]
    res2 = sopt.minimize(lambda x: 0.5 * np.sum(np.maximum(D2 @ x - f2, 0) ** 2), r["x_levels"][0],
                         jac=lambda x: D2.T @ np.maximum(D2 @ x - f2, 0), constraints=cons, method="SLSQP",
                         options=dict(ftol=1e-15, maxiter=500))
    v2 = np.maximum(D2 @ res2.x - f2, 0)
    assert abs(np.sum(v2 ** 2) - np.sum(r["v_levels"][1] ** 2)) < 1e-6
    assert np.all(D1 @ r["x_levels"][2] - f1 - r["v_levels"][0] < 1e-8)
    assert np.all(D2 @ r["x_levels"][2] - f2 - r["v_levels"][1] < 1e-8)


def test_k6_zero_dimensional_kernel(oracle):
    """K6 / Q6: square full-rank A Z -> next level has n_z = 1 (reference) / 0 (paper), same x."""
    rng = np.random.default_rng(6)
    A1, b1 = rng.normal(size=(4, 4)), rng.normal(size=4)
    A2, b2 = rng.normal(size=(2, 4)), rng.normal(size=2)
    ref = oracle.solve_tasks([dict(A=A1, b=b1), dict(A=A2, b=b2)], 4, oracle.default_cfg("reference"))
    pap = oracle.solve_tasks([dict(A=A1, b=b1), dict(A=A2, b=b2)], 4, oracle.default_cfg("paper"))
    assert ref["nz_levels"][0] == 1 and pap["nz_levels"][0] == 0
    assert np.allclose(ref["x_levels"][1], pap["x_levels"][1], atol=1e-9)
    assert np.allclose(ref["x_levels"][1], np.linalg.solve(A1, b1), atol=1e-9)


def test_qp_standin_against_scipy(oracle):
    """The Drake `Solve` stand-in on random strictly convex QPs, against SLSQP."""
    rng = np.random.default_rng(8)
    for _ in range(5):
        N, Mc = 6, 8
        L = rng.normal(size=(N, N))
        H = L @ L.T + 0.1 * np.eye(N)
        c = rng.normal(size=N)
        C, d = rng.normal(size=(Mc, N)), rng.uniform(0.1, 1.0, Mc)
        xi, lam, st, it = oracle.qp(H, c, C, d, np.zeros(N))
        assert st == 0
        res = sopt.minimize(lambda x: 0.5 * x @ H @ x + c @ x, np.zeros(N), jac=lambda x: H @ x + c,
                            constraints=[dict(type="ineq", fun=lambda x: d - C @ x, jac=lambda x: -C)],
                            method="SLSQP", options=dict(ftol=1e-14, maxiter=300))
        assert np.allclose(xi, res.x, atol=1e-6)
        assert np.all(lam >= -1e-9) and np.abs(H @ xi + c + C.T @ lam).max() < 1e-8  # KKT


def test_k7_wbc_standing(oracle):
    """K7: symmetric stance, u = 0, zero commands: ud = 0, sum lambda_z = m g, tau = -Jc_j^T lambda + h_j."""
    inp = am.make_wbc_inputs(am.standing_state())
    out, rc = oracle.wbc_solve_batch(inp)
    assert rc == 0
    x = out["x"][0]
    assert np.abs(x[:18]).max() < 1e-6
    lam = x[18:30].reshape(4, 3)
    assert abs(lam[:, 2].sum() - am.total_mass() * am.GRAV) < 1e-6
    Jf = inp["Jfoot"][0].reshape(12, 18)
    tau = -Jf[:, 6:].T @ x[18:30] + inp["h"][0][6:]
    assert np.allclose(out["tau"][0], tau, atol=1e-6)
    assert list(out["nz_levels"][0]) == [24, 24, 12, 6, 1]


def test_k8_row_counts_per_contact_pattern(oracle):
    for contact, n, m, p in [((1, 1, 1, 1), 30, [6, 0, 12, 6, 12], 40), ((1, 1, 0, 1), 27, [6, 0, 9, 9, 9], 36),
                             ((1, 0, 0, 1), 24, [6, 0, 6, 12, 6], 32)]:
        st = am.sample_states(1, 3, "stance")
        st["contact"][:] = np.array(contact, dtype=np.int32)
        inp = am.make_wbc_inputs(st)
        tasks, nn = oracle.wbc_build(inp, 0)
        assert nn == n
        assert [0 if "A" not in t else t["A"].shape[0] for t in tasks] == m
        assert tasks[1]["D"].shape == (p, n)
        # Q1: friction rows sit on columns 6 + 3*leg in reference mode, 18 + 3*i in paper mode
        legs = [l for l in range(4) if contact[l]]
        Dq = tasks[1]["D"][24:]
        for i, leg in enumerate(legs):
            assert np.count_nonzero(Dq[4 * i:4 * i + 4, 6 + 3 * leg:9 + 3 * leg]) == 8
        tasks_p, _ = oracle.wbc_build(inp, 0, oracle.default_wbc_params("paper"))
        for i in range(len(legs)):
            assert np.count_nonzero(tasks_p[1]["D"][24 + 4 * i:28 + 4 * i, 18 + 3 * i:21 + 3 * i]) == 8


def test_q5_base_position_gain(oracle):
    st = am.sample_states(1, 5, "stance")
    st["base_cmd"][0, 0:3] = [0.3, -0.2, 0.7]
    inp = am.make_wbc_inputs(st)
    t_ref, _ = oracle.wbc_build(inp, 0)
    t_pap, _ = oracle.wbc_build(inp, 0, oracle.default_wbc_params("paper"))
    diff = t_pap[3]["b"][:3] - t_ref[3]["b"][:3]
    assert np.allclose(diff, inp["base_cmd"][0, 0:3] - inp["q"][0, 4:7])


def test_k9_k11_paper_mode_kkt_properties(oracle):
    """K9/K11: tightened limits in paper mode: hard rows hold at the final x, slacks match their
    definition, higher-priority residuals are not degraded by lower levels."""
    kw = dict(mu=0.2, tau_max=40.0, tau_min=-40.0)
    inp = am.make_wbc_inputs(am.sample_states(24, 4567, "three"))
    prm = oracle.default_wbc_params("paper", **kw)
    out, rc = oracle.wbc_solve_batch(inp, oracle.default_cfg("paper"), prm, threads=4)
    assert rc == 0
    nact = 0
    for i in range(24):
        tasks, n = oracle.wbc_build(inp, i, prm)
        x = out["x"][i][:n]
        v = out["v"][i][:tasks[1]["D"].shape[0]]
        marg = tasks[1]["D"] @ x - tasks[1]["f"] - v
        assert marg.max() < 1e-7
        nact += int((marg > -1e-7).sum())
        xl = out["x_levels"][i].reshape(-1)[:5 * n].reshape(5, n)
        for k in (0, 2):  # equality levels stay exactly satisfied
            assert np.abs(tasks[k]["A"] @ x - tasks[k]["b"]).max() < 1e-7
        r3_own = np.linalg.norm(tasks[3]["A"] @ xl[3] - tasks[3]["b"])
        r3_fin = np.linalg.norm(tasks[3]["A"] @ x - tasks[3]["b"])
        assert abs(r3_own - r3_fin) < 1e-7 * max(1.0, r3_own)
    assert nact > 24  # rows really go active


def test_k10_rank_deficient_contact_jacobian(oracle):
    """K10: duplicated contact rows (two legs share one Jacobian) exercise the rank threshold."""
    st = am.sample_states(2, 9, "stance")
    inp = am.make_wbc_inputs(st)
    inp["Jfoot"][:, 1] = inp["Jfoot"][:, 0]
    inp["Jdotu"][:, 1] = inp["Jdotu"][:, 0]
    out, rc = oracle.wbc_solve_batch(inp)
    assert rc == 0
    assert list(out["nz_levels"][0][:3]) == [24, 24, 15]  # J_c has rank 9, not 12
    for i in range(2):
        tasks, n = oracle.wbc_build(inp, i)
        ch = pl.solve_tasks(tasks, n)
        assert np.allclose(ch.x_levels[2] @ tasks[2]["A"].T, out["x_levels"][i][2 * n:3 * n] @ tasks[2]["A"].T, atol=1e-6)


def test_prototype_agrees_with_oracle(oracle):
    """Two independent implementations (primal active set on [z; v] with normal equations vs
    piecewise-quadratic active set with orthogonal factorizations) agree on invariants."""
    for kind, mode, seed in [("stance", "reference", 1), ("trot", "reference", 2), ("three", "paper", 3)]:
        inp = am.make_wbc_inputs(am.sample_states(12, seed, kind))
        prm = oracle.default_wbc_params(mode)
        out, rc = oracle.wbc_solve_batch(inp, oracle.default_cfg(mode), prm, threads=4)
        assert rc == 0
        for i in range(12):
            tasks, n = oracle.wbc_build(inp, i, prm)
            ch = pl.solve_tasks(tasks, n, gate_prior_ineq=(mode == "reference"), eps=1e-12 if mode == "reference" else 0.0)
            xo = out["x"][i][:n]
            assert np.abs(ch.x - xo).max() <= 1e-7 * max(1.0, np.abs(xo).max())
            assert np.abs(ch.v_levels[1] - out["v"][i][:len(ch.v_levels[1])]).max() < 1e-8


@pytest.mark.parametrize("fname,count", [("wbc_golden.npz", 5), ("wbc_offenders.npz", 6)])
def test_golden_fixtures(oracle, fname, count):
    """Oracle against the committed vectors (tests/golden/make_golden.py): the authoritative solver
    reproduces them, and the literal double-precision solver agrees with them wherever the level
    matrices are well conditioned (on the committed ill-conditioned states it is the literal
    solver that is off, by the amount recorded with each state)."""
    g = np.load(os.path.join(HERE, "golden", fname))
    names = sorted({k.split("/")[0] for k in g.files})
    assert len(names) == count
    for name in names:
        mode, kw = g[f"{name}/meta"]
        kw = ast.literal_eval(str(kw))
        inp = {k.split("/")[2]: g[k] for k in g.files if k.startswith(f"{name}/in/")}
        out, rc = oracle.wbc_solve_batch(inp, oracle.default_cfg(str(mode)), oracle.default_wbc_params(str(mode), **kw))
        assert rc == 0
        for key in ("x", "qdd", "tau", "v", "level_res"):
            ref = g[f"{name}/out/{key}"]
            assert np.all(np.abs(out[key] - ref) <= 1e-9 * np.maximum(1.0, np.abs(ref))), (name, key)
        assert np.array_equal(out["nz_levels"], g[f"{name}/out/nz_levels"])
        clear = (np.abs(g[f"{name}/out/eq_res_max"] - 1e-7) > 1e-8) & (np.abs(g[f"{name}/out/ineq_viol_max"]) > 1e-8)
        assert np.all((out["flags"] == g[f"{name}/out/flags"]) | ~clear)
        lit, rc = oracle.wbc_solve_batch(inp, oracle.default_cfg(str(mode), solver="literal"),
                                         oracle.default_wbc_params(str(mode), **kw))
        assert rc == 0
        tol = 1e-7 if fname == "wbc_golden.npz" else max(1e-7, 3.0 * float(g[f"{name}/cond"][6]))
        ref = g[f"{name}/out/x"]
        assert np.all(np.abs(lit["x"] - ref) <= tol * np.maximum(1.0, np.abs(ref))), name


def test_poststep_known_answer(oracle):
    """N2: two Euler integrations then PD + feed-forward (whole_body_controller.cpp:148-155,
    joint_controller.cpp:103-110) on hand-computed numbers."""
    one = np.ones((1, 12))
    q, qd, tau = oracle.wbc_poststep(0.5, 10.0, 2.0, 4.0 * one, 1.0 * one, 0.25 * one, 0.5 * one, 1.0 * one, 3.0 * one)
    assert np.array_equal(qd, 5.0 * one)                 # 3 + 0.5 * 4
    assert np.array_equal(q, 3.5 * one)                  # 1 + 0.5 * 5
    assert np.array_equal(tau, (10 * 3.25 + 2 * 4.5 + 1) * one)


# the two gait schedules of the reference (motion_planner.cpp:543-547, 564-568), rows LF RF LH RH
SIMPLE = np.array([[1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1],
                   [1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1],
                   [1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1],
                   [1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]])
CRAWL = np.array([[1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1],
                  [1, 1, 1, 1, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1],
                  [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1, 1, 1, 1],
                  [1, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]])
STANCE = np.array([[0.3, 0.2], [0.3, -0.2], [-0.3, 0.2], [-0.3, -0.2]])


def test_gait_sampler_known_answers(oracle):
    """N3: gait schedules as in the reference; swing of RH in the simple gait (step 1 of 12 over 20 s)
    evaluated by hand; the reference's velocity/acceleration commands repeat the position (its
    evaluation bug), the corrected mode gives the derivatives; periodic in the gait duration."""
    assert np.array_equal(oracle.GAITS["simple"][2], SIMPLE) and oracle.GAITS["simple"][:2] == (12, 20.0)
    assert np.array_equal(oracle.GAITS["crawl"][2], CRAWL) and oracle.GAITS["crawl"][:2] == (20, 10.0)
    st = 20.0 / 12.0
    t = np.array([0.5 * st, st, 1.25 * st, 1.5 * st, 2.0 * st, 2.0 * st + 1e-9, 1.5 * st + 40.0])
    N = t.size
    vel = np.tile([0.12, -0.06], (N, 1))
    lc, ct = oracle.gait_sample("simple", t, vel, np.tile(STANCE, (N, 1, 1)))
    assert np.array_equal(ct[:, 3], [1, 0, 0, 0, 1, 1, 0]) and np.all(ct[:, :3] == 1)
    assert np.all(lc[:, :9] == 0) and np.all(lc[0] == 0) and np.all(lc[5] == 0)      # stance legs / outside the swing
    start, end = STANCE[3], STANCE[3] + st * vel[0]
    for row, w in ((1, 0.0), (2, 0.25), (3, 0.5), (4, 1.0), (6, 0.5)):
        want = np.r_[start + w * (end - start), 0.2 * (1.0 - (2.0 * w - 1.0) ** 2)]
        assert np.allclose(lc[row, 9:12], want, atol=1e-12), row
        assert np.allclose(lc[row, 21:24], want, atol=1e-12) and np.allclose(lc[row, 33:36], want, atol=1e-12)
    lf, _ = oracle.gait_sample("simple", t, vel, np.tile(STANCE, (N, 1, 1)), fix_derivatives=True)
    assert np.allclose(lf[:, :12], lc[:, :12], atol=0)
    assert np.allclose(lf[3, 21:24], [0.12, -0.06, 0.0], atol=1e-12)                   # apex: no vertical speed
    assert np.allclose(lf[2, 21:24], [0.12, -0.06, 0.2 * 2.0 * 0.5 / (0.5 * st)], atol=1e-12)
    assert np.allclose(lf[3, 33:36], [0.0, 0.0, -2.0 * 0.2 / (0.5 * st) ** 2], atol=1e-12)
    # crawl: five-step swings (LF and LH overlap for two steps), the foothold advances five steps' worth
    tc = np.linspace(0.0, 9.99, 200)
    lcc, ctc = oracle.gait_sample("crawl", tc, np.tile([0.1, 0.0], (200, 1)), np.tile(STANCE, (200, 1, 1)))
    assert np.array_equal(ctc, CRAWL[:, (tc / 0.5).astype(int)].T) and ctc.sum(axis=1).min() == 2
    sw = ctc[:, 0] == 0
    assert np.all(lcc[sw, 2] > 0) and np.all(lcc[sw, 0] >= 0.3 - 1e-12) and np.all(lcc[sw, 0] <= 0.3 + 0.25 + 1e-12)
    # a negative time counts as time 0 (never a negative gait step); t == duration wraps to 0
    te = np.array([-0.75, 0.0, 10.0, -1e-9, -25.0])
    le, ce = oracle.gait_sample("crawl", te, np.tile([0.1, 0.0], (5, 1)), np.tile(STANCE, (5, 1, 1)))
    assert np.array_equal(ce, np.tile(CRAWL[:, 0], (5, 1))) and np.all(le == le[1])


def test_standup_sampler_known_answers(oracle):
    bc, ct = oracle.standup_sample(np.array([0.0, 1.0, 2.0, 2.5]), 2.0, 0.5, np.tile([0.1, -0.2, 0.1], (4, 1)))
    assert np.allclose(bc[:, 0], 0.1) and np.allclose(bc[:, 1], -0.2) and np.allclose(bc[:, 2], [0.1, 0.3, 0.5, 0.5])
    assert np.allclose(bc[:, 5], [0.2, 0.2, 0.2, 0.0]) and np.all(bc[:, 3:5] == 0) and np.all(bc[:, 6:] == 0)
    assert np.all(ct == 1)
    # before the ramp starts the position holds its first value (Drake's FirstOrderHold clamps)
    bn, _ = oracle.standup_sample(np.array([-1.0]), 2.0, 0.5, np.array([[0.1, -0.2, 0.1]]))
    assert np.allclose(bn[0, :3], [0.1, -0.2, 0.1]) and np.allclose(bn[0, 5], 0.2)


def test_base_walk_plan_is_the_straight_line():
    """N4: the reference's spline QP (KKT solve of the literal program) has the constant-speed
    straight line as its unique solution; the reference's read-back swaps the coefficient layout."""
    from oracle import base_planner_oracle as bp

    p0, pf = np.array([0.1, -0.2]), np.array([0.6, 0.3])
    alpha = bp.base_walk_plan(p0, pf)
    d = (pf - p0) / 10.0
    for k in range(10):
        want = np.zeros(12)
        want[[0, 1, 6, 7]] = [p0[0] + k * d[0], d[0], p0[1] + k * d[1], d[1]]
        assert np.abs(alpha[k] - want).max() < 1e-12
    fixed = bp.base_walk_eval(alpha, 2.5, 0.45, fix_layout=True)
    assert np.allclose(fixed, np.r_[p0 + 2.5 * d, 0.45, d, 0.0, 0.0, 0.0, 0.0], atol=1e-12)
    ref = bp.base_walk_eval(alpha, 2.5, 0.45)
    ax0, ay0 = p0 + 2 * d
    assert np.allclose(ref[:2], [ax0 + ay0 * 0.125, d[0] + d[1] * 0.125], atol=1e-12)
    assert np.allclose(bp.base_walk_eval(alpha, 25.0, 0.45, fix_layout=True)[:2], p0 + 25.0 * d, atol=1e-12)  # last segment extrapolates


def test_oracle_under_sanitizers():
    """SURVEY.md section 5 (host tests under ASan / UBSan): every entry point of the restatement on edge shapes,
    built with -fsanitize=address,undefined (oracle/Makefile: asan_check) -- the sanitizers abort on a finding."""
    import os
    import subprocess

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.check_call(["make", "-s", "-C", root, "asan_check"])
    res = subprocess.run([os.path.join(root, "asan_check")], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "asan_check: ok" in res.stdout
