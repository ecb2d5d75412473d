"""ctypes binding of oracle/libhoqp_oracle.so (TEST INFRASTRUCTURE ONLY, parity unpinned).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libhoqp_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "hoqp_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libhoqp_oracle.so"])
    return _LIB


class Cfg(C.Structure):
    _fields_ = [("eps", C.c_double), ("gate_prior_ineq", C.c_int), ("slack_order_quirk", C.c_int),
                ("zero_kernel_col", C.c_int), ("nullspace_basis", C.c_int), ("max_iter", C.c_int),
                ("solver", C.c_int)]


class WbcParams(C.Structure):
    _fields_ = [("mu", C.c_double), ("tau_max", C.c_double), ("tau_min", C.c_double),
                ("kp_swing", C.c_double), ("kv_swing", C.c_double), ("kp_base", C.c_double),
                ("kv_base", C.c_double), ("kv_rot", C.c_double), ("friction_col_quirk", C.c_int),
                ("drop_base_pos_gain", C.c_int)]


class WbcState(C.Structure):
    _fields_ = [("M", C.c_double * 324), ("h", C.c_double * 18), ("Jfoot", C.c_double * 216),
                ("Jdotu", C.c_double * 12), ("footpos", C.c_double * 12), ("Jb", C.c_double * 108),
                ("q", C.c_double * 19), ("u", C.c_double * 18), ("base_cmd", C.c_double * 9),
                ("leg_cmd", C.c_double * 36), ("contact", C.c_int * 4)]


class WbcResult(C.Structure):
    _fields_ = [("n", C.c_int), ("nv", C.c_int), ("status", C.c_int), ("x", C.c_double * 30),
                ("x_levels", C.c_double * 150), ("qdd", C.c_double * 12), ("tau", C.c_double * 12),
                ("v", C.c_double * 64), ("nz_levels", C.c_int * 5), ("iters", C.c_int * 5),
                ("flags", C.c_int * 5), ("eq_res_max", C.c_double * 5),
                ("ineq_viol_max", C.c_double * 5), ("level_res", C.c_double * 5),
                ("active_mask", C.c_ulonglong)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise RuntimeError("oracle library not built: run `make -C oracle` (or __graft_entry__.build())")
        _lib = C.CDLL(_LIB)
        _lib.hoqp_oracle_fullpivlu_kernel.restype = C.c_int
        _lib.hoqp_oracle_solve_tasks.restype = C.c_int
        _lib.hoqp_oracle_qp.restype = C.c_int
        _lib.hoqp_oracle_wbc_build.restype = C.c_int
        _lib.hoqp_oracle_wbc_solve_batch.restype = C.c_int
    return _lib


_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def default_cfg(mode: str = "reference", nullspace: str | None = None, solver: str = "lsq") -> Cfg:
    """solver "lsq": every QP in least-squares form, Householder QR in long double (authoritative);
    "literal": H = (AZ)'AZ + eps I formed and factored in double, as a double-precision restatement of
    ho_qp_problem.cpp:231-357 computes it (used as the CPU baseline and as a second opinion)."""
    cfg = Cfg()
    lib().hoqp_oracle_default_cfg(C.byref(cfg))
    cfg.solver = {"lsq": 1, "literal": 0}[solver]
    if mode == "paper":
        cfg.gate_prior_ineq = 0
        cfg.slack_order_quirk = 0
        cfg.zero_kernel_col = 0
        cfg.nullspace_basis = 1
        # "paper" = the algorithm without the reference's eps_*I regulariser: minimum-norm least squares
        # (the literal solver needs a tiny positive value for its pivot floor 100*eps; 1e-15 is below
        # fp64 resolution here)
        cfg.eps = 0.0 if cfg.solver == 1 else 1e-15
    elif mode != "reference":
        raise ValueError(mode)
    if nullspace == "householder":
        cfg.nullspace_basis = 1
    elif nullspace == "fullpivlu":
        cfg.nullspace_basis = 0
    return cfg


def default_wbc_params(mode: str = "reference", **kw) -> WbcParams:
    prm = WbcParams()
    lib().hoqp_oracle_default_wbc_params(C.byref(prm))
    if mode == "paper":
        prm.friction_col_quirk = 0
        prm.drop_base_pos_gain = 0
    for k, v in kw.items():
        setattr(prm, k, v)
    return prm


def qp(H, c, Cm, d, xi0, maxit=2000):
    H = np.ascontiguousarray(H, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    N = H.shape[0]
    Cm = np.ascontiguousarray(Cm, dtype=np.float64).reshape(-1, N)
    d = np.ascontiguousarray(d, dtype=np.float64)
    xi = np.array(xi0, dtype=np.float64).copy()
    lam = np.zeros(max(1, Cm.shape[0]))
    it = C.c_int(0)
    st = lib().hoqp_oracle_qp(N, _dp(H), _dp(c), Cm.shape[0], _dp(Cm), _dp(d), _dp(xi), _dp(lam),
                              maxit, C.byref(it))
    return xi, lam[:Cm.shape[0]], st, it.value


def fullpivlu_kernel(A, zero_col=True):
    A = np.ascontiguousarray(A, dtype=np.float64)
    r, c = A.shape
    K = np.zeros((c, max(c, 1)))
    buf = np.zeros(c * max(c, 1))
    dk = lib().hoqp_oracle_fullpivlu_kernel(r, c, _dp(A), _dp(buf), int(zero_col))
    return buf[: c * dk].reshape(c, dk).copy()


def solve_tasks(tasks, n, cfg=None):
    """tasks: list of dicts with optional A (m,n), b (m), D (p,n), f (p).  One instance."""
    cfg = cfg or default_cfg()
    K = len(tasks)
    m = np.array([0 if t.get("A") is None else np.asarray(t["A"]).reshape(-1, n).shape[0] for t in tasks],
                 dtype=np.int32)
    p = np.array([0 if t.get("D") is None else np.asarray(t["D"]).reshape(-1, n).shape[0] for t in tasks],
                 dtype=np.int32)

    def cat(key, rows):
        parts = [np.asarray(t[key], dtype=np.float64).reshape(-1) for t, r in zip(tasks, rows) if r > 0]
        return np.ascontiguousarray(np.concatenate(parts) if parts else np.zeros(1))

    A, b, D, f = cat("A", m), cat("b", m), cat("D", p), cat("f", p)
    xl = np.zeros((K, n))
    nzl = np.zeros(K, dtype=np.int32)
    Zl = np.zeros((K, n, n))
    vl = np.zeros(max(1, int(p.sum())))
    it = np.zeros(K, dtype=np.int32)
    st = np.zeros(K, dtype=np.int32)
    rc = lib().hoqp_oracle_solve_tasks(C.byref(cfg), n, K, _ip(m), _ip(p), _dp(A), _dp(b), _dp(D), _dp(f),
                                       _dp(xl), _ip(nzl), _dp(Zl), _dp(vl), _ip(it), _ip(st))
    vs, off = [], 0
    for pk in p:
        vs.append(vl[off:off + pk].copy())
        off += pk
    return dict(x_levels=xl, nz_levels=nzl, Z_levels=[Zl[k][:, :nzl[k]].copy() for k in range(K)],
                v_levels=vs, iters=it, status=st, rc=rc, m=m, p=p)


def pack_states(inp) -> "C.Array":
    """dict of batched arrays (oracle/anymal_model.make_wbc_inputs) -> array of WbcState."""
    N = inp["M"].shape[0]
    arr = (WbcState * N)()
    raw = np.frombuffer(arr, dtype=np.uint8).reshape(N, C.sizeof(WbcState))
    off = 0
    for name, cnt in (("M", 324), ("h", 18), ("Jfoot", 216), ("Jdotu", 12), ("footpos", 12), ("Jb", 108),
                      ("q", 19), ("u", 18), ("base_cmd", 9), ("leg_cmd", 36)):
        a = np.ascontiguousarray(inp[name], dtype=np.float64).reshape(N, cnt)
        raw[:, off:off + 8 * cnt] = a.view(np.uint8).reshape(N, 8 * cnt)
        off += 8 * cnt
    a = np.ascontiguousarray(inp["contact"], dtype=np.int32).reshape(N, 4)
    raw[:, off:off + 16] = a.view(np.uint8).reshape(N, 16)
    return arr


def wbc_build(inp, i=0, prm=None):
    prm = prm or default_wbc_params()
    st = pack_states({k: v[i:i + 1] for k, v in inp.items()})
    m = np.zeros(5, dtype=np.int32)
    p = np.zeros(5, dtype=np.int32)
    A = np.zeros(64 * 30)
    b = np.zeros(64)
    D = np.zeros(64 * 30)
    f = np.zeros(64)
    n = lib().hoqp_oracle_wbc_build(C.byref(prm), C.byref(st[0]), _ip(m), _ip(p), _dp(A), _dp(b), _dp(D), _dp(f))
    tasks, ao, do = [], 0, 0
    for k in range(5):
        t = {}
        if m[k]:
            t["A"] = A[ao * n:(ao + m[k]) * n].reshape(m[k], n).copy()
            t["b"] = b[ao:ao + m[k]].copy()
        if p[k]:
            t["D"] = D[do * n:(do + p[k]) * n].reshape(p[k], n).copy()
            t["f"] = f[do:do + p[k]].copy()
        ao += m[k]
        do += p[k]
        tasks.append(t)
    return tasks, n


def wbc_solve_batch(inp, cfg=None, prm=None, threads=1):
    cfg = cfg or default_cfg()
    prm = prm or default_wbc_params()
    N = inp["M"].shape[0]
    st = pack_states(inp)
    res = (WbcResult * N)()
    rc = lib().hoqp_oracle_wbc_solve_batch(C.byref(cfg), C.byref(prm), N, st, res, threads)
    dt = np.dtype([("n", "i4"), ("nv", "i4"), ("status", "i4"), ("_pad", "i4"), ("x", "f8", 30),
                   ("x_levels", "f8", 150), ("qdd", "f8", 12), ("tau", "f8", 12), ("v", "f8", 64),
                   ("nz_levels", "i4", 5), ("iters", "i4", 5), ("flags", "i4", 5), ("_pad2", "i4"),
                   ("eq_res_max", "f8", 5), ("ineq_viol_max", "f8", 5), ("level_res", "f8", 5),
                   ("active_mask", "u8")])
    assert dt.itemsize == C.sizeof(WbcResult), (dt.itemsize, C.sizeof(WbcResult))
    out = np.frombuffer(res, dtype=dt).copy()
    return out, rc


def wbc_poststep(dt, kp, kd, qdd, tau_ff, q_j, qd_j, q_cmd, qd_cmd):
    """Post-solve step (N2): returns (q_cmd, qd_cmd, tau_cmd) as new arrays; inputs [N, 12]."""
    N = qdd.shape[0]
    arrs = [np.ascontiguousarray(a, dtype=np.float64).copy() for a in (qdd, tau_ff, q_j, qd_j, q_cmd, qd_cmd)]
    tau = np.zeros((N, 12))
    f = lib().hoqp_oracle_wbc_poststep
    f.restype = None
    f.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double] + [C.POINTER(C.c_double)] * 7
    f(N, dt, kp, kd, *[_dp(a) for a in arrs], _dp(tau))
    return arrs[4], arrs[5], tau


# gait schedules of MotionPlanner::CreateSimpleSequence / CreateCrawlSequence (motion_planner.cpp:536-575):
# (n_steps, duration, [4][n_steps] contact schedule, legs LF RF LH RH)
def _schedule(n, swings):
    sch = np.ones((4, n), dtype=np.int32)
    for leg, steps in enumerate(swings):
        sch[leg, list(steps)] = 0
    return sch


GAITS = {
    "simple": (12, 20.0, _schedule(12, [(10,), (7,), (4,), (1,)])),
    "crawl": (20, 10.0, _schedule(20, [range(14, 19), range(4, 9), range(11, 16), range(1, 6)])),
}


def gait_sample(gait, time, vel_cmd, stance0, apex=0.2, fix_derivatives=False):
    """N3: (leg_cmd [N,36], contact [N,4]) at the given times; gait = key of GAITS or (n, duration, schedule)."""
    n, dur, sch = GAITS[gait] if isinstance(gait, str) else gait
    sch = np.ascontiguousarray(sch, dtype=np.int32)
    time = np.ascontiguousarray(time, dtype=np.float64)
    N = time.shape[0]
    vel = np.ascontiguousarray(vel_cmd, dtype=np.float64).reshape(N, 2)
    st0 = np.ascontiguousarray(stance0, dtype=np.float64).reshape(N, 4, 2)
    leg_cmd = np.zeros((N, 36))
    contact = np.zeros((N, 4), dtype=np.int32)
    f = lib().hoqp_oracle_gait_sample
    f.restype = None
    f.argtypes = [C.c_int, C.c_int, C.c_double, _IP, C.c_double, C.c_int, _DP, _DP, _DP, _DP, _IP]
    f(N, n, dur, _ip(sch), apex, int(fix_derivatives), _dp(time), _dp(vel), _dp(st0), _dp(leg_cmd), _ip(contact))
    return leg_cmd, contact


def standup_sample(time, seconds, target_height, curr_pose):
    time = np.ascontiguousarray(time, dtype=np.float64)
    N = time.shape[0]
    pose = np.ascontiguousarray(curr_pose, dtype=np.float64).reshape(N, 3)
    base_cmd = np.zeros((N, 9))
    contact = np.zeros((N, 4), dtype=np.int32)
    f = lib().hoqp_oracle_standup_sample
    f.restype = None
    f.argtypes = [C.c_int, C.c_double, C.c_double, _DP, _DP, _DP, _IP]
    f(N, seconds, target_height, _dp(time), _dp(pose), _dp(base_cmd), _ip(contact))
    return base_cmd, contact
