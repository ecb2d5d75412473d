"""ctypes binding of the C ABI in include/hoqp.h (libhoqp_hip.so, gfx950 HIP kernels).

PyTorch is used only as the owner of device memory and streams; every solve goes
through the C ABI with raw device pointers.  There is no CPU fallback: if the HIP
library is missing or no GPU is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HOQP_LIB", os.path.join(_HERE, "libhoqp_hip.so"))  # override: profiling build

MAX_LEVELS = 8
MAX_VARS = 32
MAX_EQ_ROWS = 32
MAX_INEQ_ROWS = 64

Q1_FRICTION_COLS = 0x01
Q2_GATE_PRIOR_INEQ = 0x02
Q3_SLACK_ORDER = 0x04
Q5_DROP_BASE_POS_GAIN = 0x10
Q6_ZERO_KERNEL_COL = 0x20
SEMANTICS_REFERENCE = 0x37
SEMANTICS_PAPER = 0x00

STATUS_MAX_ITER = 0x01
STATUS_INFEASIBLE_START = 0x02
STATUS_WORKSPACE = 0x04
STATUS_NONFINITE = 0x08
STATUS_NEAR_SINGULAR = 0x10
STATUS_BAD_EXTENT = 0x20
STATUS_BASE_RANK = 0x40
NOALLOC_MAX_BATCH = 256
OPT_SKIP_UNREAD_SLACK_LEVELS = 0x01

EXPORTS = (
    "hoqp_default_config", "hoqp_create", "hoqp_destroy", "hoqp_last_error_string",
    "hoqp_solve_tasks_batched", "hoqp_solve_wbc_batched", "hoqp_solve_tasks_host", "hoqp_solve_wbc_host",
    "hoqp_solve_wbc_unfused", "hoqp_solve_wbc_noalloc", "hoqp_wbc_build_tasks_batched",
    "hoqp_default_quadruped_model", "hoqp_wbc_dynamics_batched", "hoqp_wbc_dynamics_host", "hoqp_wbc_poststep_batched",
    "hoqp_default_gait", "hoqp_gait_sample_batched", "hoqp_standup_sample_batched",
    "hoqp_base_walk_sample_batched", "hoqp_synchronize",
    "hoqp_kernel_time_ms", "hoqp_selftest", "hoqp_lds_bytes",
    "hoqp_sharded_create", "hoqp_sharded_destroy", "hoqp_sharded_last_error", "hoqp_solve_wbc_sharded",
)
MAX_SHARDS = 16

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_ulonglong)


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("semantics", C.c_int), ("eps", C.c_double), ("max_iter", C.c_int),
                ("mu", C.c_double), ("tau_max", C.c_double), ("tau_min", C.c_double),
                ("kp_swing", C.c_double), ("kv_swing", C.c_double), ("kp_base", C.c_double),
                ("kv_base", C.c_double), ("kv_rot", C.c_double), ("options", C.c_int)]


class QuadrupedModel(C.Structure):
    _fields_ = [("base_mass", C.c_double), ("hip_mass", C.c_double), ("thigh_mass", C.c_double),
                ("shank_mass", C.c_double)] + [(n, C.c_double * 3) for n in (
                    "base_inertia", "hip_inertia", "thigh_inertia", "shank_inertia", "haa_off", "hfe_off", "kfe_off",
                    "foot_off", "hip_com", "thigh_com", "shank_com")] + [("gravity", C.c_double)]


class Gait(C.Structure):
    """hoqp_gait (GaitSequence of the reference as bit masks per leg + swing apex + derivative switch)."""
    _fields_ = [("n_steps", C.c_int), ("duration", C.c_double), ("contact_bits", C.c_uint * 4),
                ("apex_height", C.c_double), ("fix_derivatives", C.c_int)]


class TaskBatch(C.Structure):
    _fields_ = [("batch", C.c_int), ("n", C.c_int), ("num_levels", C.c_int),
                ("m", C.c_int * MAX_LEVELS), ("p", C.c_int * MAX_LEVELS),
                ("A", C.c_void_p * MAX_LEVELS), ("b", C.c_void_p * MAX_LEVELS),
                ("D", C.c_void_p * MAX_LEVELS), ("f", C.c_void_p * MAX_LEVELS),
                ("n_inst", C.c_void_p), ("m_inst", C.c_void_p), ("p_inst", C.c_void_p)]


class ResultBatch(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_levels", C.c_void_p), ("v", C.c_void_p), ("level_res", C.c_void_p),
                ("Z_levels", C.c_void_p), ("nz_levels", C.c_void_p), ("iters", C.c_void_p),
                ("status", C.c_void_p), ("flags", C.c_void_p), ("eq_res_max", C.c_void_p),
                ("ineq_viol_max", C.c_void_p), ("active_mask", C.c_void_p)]


class WbcInputs(C.Structure):
    _fields_ = [("batch", C.c_int), ("M", C.c_void_p), ("h", C.c_void_p), ("Jfoot", C.c_void_p),
                ("Jdotu", C.c_void_p), ("footpos", C.c_void_p), ("Jb", C.c_void_p), ("q", C.c_void_p),
                ("u", C.c_void_p), ("base_cmd", C.c_void_p), ("leg_cmd", C.c_void_p), ("contact", C.c_void_p)]


class WbcOutputs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("qdd", C.c_void_p), ("tau", C.c_void_p), ("v", C.c_void_p),
                ("flags", C.c_void_p), ("eq_res_max", C.c_void_p), ("ineq_viol_max", C.c_void_p),
                ("level_res", C.c_void_p), ("active_mask", C.c_void_p), ("status", C.c_void_p),
                ("iters", C.c_void_p), ("nz_levels", C.c_void_p), ("x_levels", C.c_void_p)]


class ShardStats(C.Structure):
    _fields_ = [("eq_res_max", C.c_double), ("ineq_viol_max", C.c_double), ("status_nonzero", C.c_longlong),
                ("iters_total", C.c_longlong), ("instances", C.c_longlong)]


class WbcGather(C.Structure):
    _fields_ = [("root_device", C.c_int), ("x", C.c_void_p), ("tau", C.c_void_p), ("v", C.c_void_p),
                ("status", C.c_void_p)]


_lib = None


def load_library():
    """dlopen libhoqp_hip.so and declare prototypes.  Raises if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with __graft_entry__.build() "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # torch bundles its own libamdhip64.so.7; it must be in the process first so that this
        # library binds to the SAME HIP runtime (two runtimes in one process do not share devices,
        # streams or allocations).
        import torch  # noqa: F401

        lib = C.CDLL(LIB_PATH)
        lib.hoqp_default_config.argtypes = [C.POINTER(Config)]
        lib.hoqp_default_config.restype = None
        lib.hoqp_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        lib.hoqp_destroy.argtypes = [C.c_void_p]
        lib.hoqp_last_error_string.argtypes = [C.c_void_p]
        lib.hoqp_last_error_string.restype = C.c_char_p
        lib.hoqp_solve_tasks_batched.argtypes = [C.c_void_p, C.POINTER(TaskBatch), C.POINTER(ResultBatch), C.c_void_p]
        lib.hoqp_solve_wbc_batched.argtypes = [C.c_void_p, C.POINTER(WbcInputs), C.POINTER(WbcOutputs), C.c_void_p]
        lib.hoqp_solve_wbc_unfused.argtypes = lib.hoqp_solve_wbc_batched.argtypes
        lib.hoqp_solve_wbc_noalloc.argtypes = lib.hoqp_solve_wbc_batched.argtypes
        lib.hoqp_solve_wbc_host.argtypes = [C.c_void_p, C.POINTER(WbcInputs), C.POINTER(WbcOutputs)]
        lib.hoqp_solve_tasks_host.argtypes = [C.c_void_p, C.POINTER(TaskBatch), C.POINTER(ResultBatch)]
        lib.hoqp_wbc_build_tasks_batched.argtypes = [C.c_void_p, C.POINTER(WbcInputs), C.POINTER(TaskBatch), C.c_void_p]
        lib.hoqp_synchronize.argtypes = [C.c_void_p, C.c_void_p]
        lib.hoqp_default_quadruped_model.argtypes = [C.POINTER(QuadrupedModel)]
        lib.hoqp_default_quadruped_model.restype = None
        lib.hoqp_wbc_dynamics_batched.argtypes = [C.c_void_p, C.POINTER(QuadrupedModel), C.c_int, C.c_void_p, C.c_void_p,
                                                  C.c_int] + [C.c_void_p] * 7
        lib.hoqp_default_gait.argtypes = [C.c_int, C.POINTER(Gait)]
        lib.hoqp_gait_sample_batched.argtypes = [C.c_void_p, C.POINTER(Gait), C.c_int] + [C.c_void_p] * 6
        lib.hoqp_standup_sample_batched.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double] + [C.c_void_p] * 4
        lib.hoqp_base_walk_sample_batched.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                                      C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.hoqp_wbc_poststep_batched.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double] + [C.c_void_p] * 8
        lib.hoqp_kernel_time_ms.argtypes = [C.c_void_p, C.c_int, _dp, _ip]
        lib.hoqp_selftest.argtypes = [C.c_void_p]
        lib.hoqp_lds_bytes.argtypes = [C.c_int, C.c_int, _ip, _ip]
        lib.hoqp_lds_bytes.restype = C.c_size_t
        lib.hoqp_sharded_create.argtypes = [C.POINTER(Config), C.c_int, _ip, C.POINTER(C.c_void_p)]
        lib.hoqp_sharded_destroy.argtypes = [C.c_void_p]
        lib.hoqp_sharded_last_error.argtypes = [C.c_void_p]
        lib.hoqp_sharded_last_error.restype = C.c_char_p
        lib.hoqp_solve_wbc_sharded.argtypes = [C.c_void_p, C.POINTER(WbcInputs), C.POINTER(WbcOutputs),
                                               C.POINTER(WbcGather), C.POINTER(ShardStats)]
        _lib = lib
    return _lib


def default_config(mode: str = "reference", device: int = 0, **kw) -> Config:
    cfg = Config()
    load_library().hoqp_default_config(C.byref(cfg))
    cfg.device = device
    if mode == "paper":
        cfg.semantics = SEMANTICS_PAPER
        cfg.eps = 0.0
    elif mode != "reference":
        raise ValueError(mode)
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def default_gait(kind: str = "simple", **kw) -> Gait:
    """MotionPlanner::CreateSimpleSequence ("simple") / CreateCrawlSequence ("crawl")."""
    g = Gait()
    if load_library().hoqp_default_gait({"simple": 0, "crawl": 1}[kind], C.byref(g)) != 0:
        raise ValueError(kind)
    for k, v in kw.items():
        setattr(g, k, v)
    return g


def lds_bytes(n, m, p):
    L = len(m)
    return load_library().hoqp_lds_bytes(n, L, (C.c_int * L)(*m), (C.c_int * L)(*p))


WBC_M = (6, 0, 12, 18, 12)
WBC_P = (0, 40, 0, 0, 0)


class HoQpSolver:
    """One handle == one GPU == one host thread (include/hoqp.h)."""

    def __init__(self, cfg: Config | None = None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("HoQpSolver needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.lib = load_library()
        self.cfg = cfg or default_config()
        self.torch = torch
        self.device = torch.device("cuda", self.cfg.device)
        h = C.c_void_p()
        rc = self.lib.hoqp_create(C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"hoqp_create failed: {rc}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.hoqp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.hoqp_last_error_string(self.h)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def selftest(self):
        self._check(self.lib.hoqp_selftest(self.h), "hoqp_selftest")

    def synchronize(self):
        self._check(self.lib.hoqp_synchronize(self.h, self._stream()), "hoqp_synchronize")

    def timing(self, enable: bool):
        self._check(self.lib.hoqp_kernel_time_ms(self.h, 1 if enable else 2, None, None), "hoqp_kernel_time_ms")

    def kernel_time_ms(self):
        avg = C.c_double(0)
        cnt = C.c_int(0)
        self._check(self.lib.hoqp_kernel_time_ms(self.h, 0, C.byref(avg), C.byref(cnt)), "hoqp_kernel_time_ms")
        return avg.value, cnt.value

    # -- generic hierarchy -------------------------------------------------
    def alloc_task_outputs(self, batch, n, m, p, debug=False):
        t, dev = self.torch, self.device
        L, P = len(m), max(1, int(sum(p)))
        o = dict(x=t.zeros(batch, n, dtype=t.float64, device=dev),
                 x_levels=t.zeros(batch, L, n, dtype=t.float64, device=dev),
                 v=t.zeros(batch, P, dtype=t.float64, device=dev),
                 level_res=t.zeros(batch, L, dtype=t.float64, device=dev),
                 nz_levels=t.zeros(batch, L, dtype=t.int32, device=dev),
                 iters=t.zeros(batch, L, dtype=t.int32, device=dev),
                 status=t.zeros(batch, dtype=t.int32, device=dev),
                 flags=t.zeros(batch, L, dtype=t.int32, device=dev),
                 eq_res_max=t.zeros(batch, L, dtype=t.float64, device=dev),
                 ineq_viol_max=t.zeros(batch, L, dtype=t.float64, device=dev),
                 active_mask=t.zeros(batch, dtype=t.int64, device=dev))
        if debug:
            o["Z_levels"] = t.zeros(batch, L, n, n, dtype=t.float64, device=dev)
        return o

    def solve_tasks(self, tasks, n, out=None, n_inst=None, m_inst=None, p_inst=None, debug=False):
        """tasks: list (priority order) of dicts with optional device tensors
        A [B,m,n], b [B,m], D [B,p,n], f [B,p] (float64, contiguous)."""
        t = self.torch
        L = len(tasks)
        tb = TaskBatch()
        B = None
        ms, ps = [], []
        keep = []
        for k, task in enumerate(tasks):
            A, b, D, f = (task.get(key) for key in ("A", "b", "D", "f"))
            mk = 0 if A is None else A.shape[1]
            pk = 0 if D is None else D.shape[1]
            for arr in (A, b, D, f):
                if arr is not None:
                    assert arr.dtype == t.float64 and arr.is_contiguous() and arr.is_cuda
                    B = arr.shape[0] if B is None else B
                    assert arr.shape[0] == B
                    keep.append(arr)
            tb.m[k], tb.p[k] = mk, pk
            tb.A[k] = A.data_ptr() if mk else None
            tb.b[k] = b.data_ptr() if mk else None
            tb.D[k] = D.data_ptr() if pk else None
            tb.f[k] = f.data_ptr() if pk else None
            ms.append(mk)
            ps.append(pk)
        tb.batch, tb.n, tb.num_levels = B or 0, n, L
        for name, arr in (("n_inst", n_inst), ("m_inst", m_inst), ("p_inst", p_inst)):
            if arr is not None:
                assert arr.dtype == t.int32 and arr.is_contiguous() and arr.is_cuda
                setattr(tb, name, arr.data_ptr())
        if out is None:
            out = self.alloc_task_outputs(B or 0, n, ms, ps, debug)
        rb = ResultBatch()
        for name, _ in ResultBatch._fields_:
            if name in out and out[name] is not None:
                setattr(rb, name, out[name].data_ptr())
        self._check(self.lib.hoqp_solve_tasks_batched(self.h, C.byref(tb), C.byref(rb), self._stream()),
                    "hoqp_solve_tasks_batched")
        return out

    # -- whole-body controller ---------------------------------------------
    def alloc_wbc_outputs(self, batch):
        t, dev = self.torch, self.device
        return dict(x=t.zeros(batch, 30, dtype=t.float64, device=dev),
                    qdd=t.zeros(batch, 12, dtype=t.float64, device=dev),
                    tau=t.zeros(batch, 12, dtype=t.float64, device=dev),
                    v=t.zeros(batch, 40, dtype=t.float64, device=dev),
                    flags=t.zeros(batch, 5, dtype=t.int32, device=dev),
                    eq_res_max=t.zeros(batch, 5, dtype=t.float64, device=dev),
                    ineq_viol_max=t.zeros(batch, 5, dtype=t.float64, device=dev),
                    level_res=t.zeros(batch, 5, dtype=t.float64, device=dev),
                    active_mask=t.zeros(batch, dtype=t.int64, device=dev),
                    status=t.zeros(batch, dtype=t.int32, device=dev),
                    iters=t.zeros(batch, 5, dtype=t.int32, device=dev),
                    nz_levels=t.zeros(batch, 5, dtype=t.int32, device=dev),
                    x_levels=t.zeros(batch, 5, 30, dtype=t.float64, device=dev))

    def to_device(self, inp):
        """dict of numpy arrays (the layout of hoqp_wbc_inputs) -> dict of device tensors."""
        t = self.torch
        return {k: t.as_tensor(v).contiguous().to(self.device) for k, v in inp.items()}

    def wbc_dynamics(self, q, u, include_gravity=True, model=None, with_Jb=True):
        """Dynamics terms of the batch of states (hoqp_wbc_dynamics_batched): q [B, 19], u [B, 18]
        float64 CUDA tensors -> dict with M, h, Jfoot, Jdotu, footpos (, Jb) in the layout solve_wbc takes."""
        t = self.torch
        B = q.shape[0]
        assert q.dtype == t.float64 and q.is_cuda and q.is_contiguous() and tuple(q.shape) == (B, 19)
        assert u.dtype == t.float64 and u.is_cuda and u.is_contiguous() and tuple(u.shape) == (B, 18)
        kw = dict(dtype=t.float64, device=q.device)
        out = {"M": t.empty((B, 18, 18), **kw), "h": t.empty((B, 18), **kw), "Jfoot": t.empty((B, 4, 3, 18), **kw),
               "Jdotu": t.empty((B, 4, 3), **kw), "footpos": t.empty((B, 4, 3), **kw)}
        if with_Jb:
            out["Jb"] = t.empty((B, 6, 18), **kw)
        self._check(self.lib.hoqp_wbc_dynamics_batched(
            self.h, C.byref(model) if model is not None else None, B, q.data_ptr(), u.data_ptr(),
            1 if include_gravity else 0, out["M"].data_ptr(), out["h"].data_ptr(), out["Jfoot"].data_ptr(),
            out["Jdotu"].data_ptr(), out["footpos"].data_ptr(), out["Jb"].data_ptr() if with_Jb else None,
            self._stream()), "hoqp_wbc_dynamics_batched")
        return out

    def wbc_poststep(self, dt, kp, kd, qdd, tau_ff, q_j, qd_j, q_cmd, qd_cmd, tau_cmd=None):
        """In-place integrator update of (q_cmd, qd_cmd) and the PD + feed-forward torque
        (hoqp_wbc_poststep_batched); all arguments are [batch, 12] float64 CUDA tensors."""
        t = self.torch
        B = qdd.shape[0]
        ptrs = []
        for name, arr in (("qdd", qdd), ("tau_ff", tau_ff), ("q_j", q_j), ("qd_j", qd_j), ("q_cmd", q_cmd),
                          ("qd_cmd", qd_cmd), ("tau_cmd", tau_cmd)):
            if arr is None:
                ptrs.append(None)
                continue
            assert arr.dtype == t.float64 and arr.is_contiguous() and arr.is_cuda and tuple(arr.shape) == (B, 12), name
            ptrs.append(C.c_void_p(arr.data_ptr()))
        self._check(self.lib.hoqp_wbc_poststep_batched(self.h, B, C.c_double(dt), C.c_double(kp), C.c_double(kd),
                                                       *ptrs, self._stream()), "hoqp_wbc_poststep_batched")
        return tau_cmd

    # -- command sampler -------------------------------------
    def gait_sample(self, gait, time, vel_cmd, stance0):
        """leg_cmd [B,36] and contact [B,4] of the gait (a Gait, or "simple" / "crawl") at time [B] for
        plans made from stance0 [B,4,2] under vel_cmd [B,2]; float64 CUDA tensors."""
        t = self.torch
        if isinstance(gait, str):
            gait = default_gait(gait)
        B = time.shape[0]
        for name, arr, shape in (("time", time, (B,)), ("vel_cmd", vel_cmd, (B, 2)), ("stance0", stance0, (B, 4, 2))):
            assert arr.dtype == t.float64 and arr.is_contiguous() and arr.is_cuda and tuple(arr.shape) == shape, name
        leg_cmd = t.empty(B, 36, dtype=t.float64, device=self.device)
        contact = t.empty(B, 4, dtype=t.int32, device=self.device)
        self._check(self.lib.hoqp_gait_sample_batched(self.h, C.byref(gait), B, time.data_ptr(), vel_cmd.data_ptr(),
                                                      stance0.data_ptr(), leg_cmd.data_ptr(), contact.data_ptr(),
                                                      self._stream()), "hoqp_gait_sample_batched")
        return leg_cmd, contact

    def standup_sample(self, time, seconds, target_height, curr_pose):
        """base_cmd [B,9] and contact [B,4] of the stand-up ramp at time [B] from curr_pose [B,3]."""
        t = self.torch
        B = time.shape[0]
        assert time.dtype == t.float64 and time.is_cuda and time.is_contiguous()
        assert curr_pose.dtype == t.float64 and curr_pose.is_cuda and curr_pose.is_contiguous() and tuple(curr_pose.shape) == (B, 3)
        base_cmd = t.empty(B, 9, dtype=t.float64, device=self.device)
        contact = t.empty(B, 4, dtype=t.int32, device=self.device)
        self._check(self.lib.hoqp_standup_sample_batched(self.h, B, time.data_ptr(), C.c_double(seconds),
                                                         C.c_double(target_height), curr_pose.data_ptr(),
                                                         base_cmd.data_ptr(), contact.data_ptr(), self._stream()),
                    "hoqp_standup_sample_batched")
        return base_cmd, contact

    def base_walk_sample(self, time, curr_pos, final_pos, walking_height, n_segments=10, fix_layout=False):
        """base_cmd [B,9] of the base walk trajectory at time [B]; curr_pos, final_pos [B,2]."""
        t = self.torch
        B = time.shape[0]
        for name, arr, shape in (("time", time, (B,)), ("curr_pos", curr_pos, (B, 2)), ("final_pos", final_pos, (B, 2))):
            assert arr.dtype == t.float64 and arr.is_contiguous() and arr.is_cuda and tuple(arr.shape) == shape, name
        base_cmd = t.empty(B, 9, dtype=t.float64, device=self.device)
        self._check(self.lib.hoqp_base_walk_sample_batched(self.h, B, time.data_ptr(), curr_pos.data_ptr(),
                                                           final_pos.data_ptr(), C.c_double(walking_height), n_segments,
                                                           int(fix_layout), base_cmd.data_ptr(), self._stream()),
                    "hoqp_base_walk_sample_batched")
        return base_cmd

    def _wbc_inputs(self, dinp):
        t = self.torch
        wi = WbcInputs()
        wi.batch = dinp["M"].shape[0]
        for name in ("M", "h", "Jfoot", "Jdotu", "footpos", "Jb", "q", "u", "base_cmd", "leg_cmd"):
            arr = dinp.get(name)
            if arr is not None:
                assert arr.dtype == t.float64 and arr.is_contiguous() and arr.is_cuda, name
                setattr(wi, name, arr.data_ptr())
        c = dinp["contact"]
        assert c.dtype == t.int32 and c.is_contiguous() and c.is_cuda
        wi.contact = c.data_ptr()
        return wi

    def build_wbc_tasks(self, dinp):
        """The five task blocks of every state as dense device tensors (hoqp_wbc_build_tasks_batched):
        returns (tasks, n_inst, m_inst, p_inst) in the form solve_tasks takes."""
        t = self.torch
        B = dinp["M"].shape[0]
        wi = self._wbc_inputs(dinp)
        kw = dict(dtype=t.float64, device=self.device)
        tasks = []
        tb = TaskBatch()
        tb.batch, tb.n, tb.num_levels = B, 30, 5
        for k, (m, p) in enumerate(zip(WBC_M, WBC_P)):
            task = {}
            tb.m[k], tb.p[k] = m, p
            if m:
                task["A"], task["b"] = t.full((B, m, 30), float("nan"), **kw), t.full((B, m), float("nan"), **kw)
                tb.A[k], tb.b[k] = task["A"].data_ptr(), task["b"].data_ptr()
            if p:
                task["D"], task["f"] = t.full((B, p, 30), float("nan"), **kw), t.full((B, p), float("nan"), **kw)
                tb.D[k], tb.f[k] = task["D"].data_ptr(), task["f"].data_ptr()
            tasks.append(task)
        n_inst = t.zeros(B, dtype=t.int32, device=self.device)
        m_inst = t.zeros(B, 5, dtype=t.int32, device=self.device)
        p_inst = t.zeros(B, 5, dtype=t.int32, device=self.device)
        tb.n_inst, tb.m_inst, tb.p_inst = n_inst.data_ptr(), m_inst.data_ptr(), p_inst.data_ptr()
        self._check(self.lib.hoqp_wbc_build_tasks_batched(self.h, C.byref(wi), C.byref(tb), self._stream()),
                    "hoqp_wbc_build_tasks_batched")
        return tasks, n_inst, m_inst, p_inst

    def solve_wbc(self, dinp, out=None, entry="batched"):
        """entry: "batched" (fused, one kernel), "unfused" (three kernels, handle workspace) or
        "noalloc" (three kernels, static device scratch, batch <= NOALLOC_MAX_BATCH)."""
        B = dinp["M"].shape[0]
        wi = self._wbc_inputs(dinp)
        if out is None:
            out = self.alloc_wbc_outputs(B)
        wo = WbcOutputs()
        for name, _ in WbcOutputs._fields_:
            if out.get(name) is not None:
                setattr(wo, name, out[name].data_ptr())
        fn = {"batched": self.lib.hoqp_solve_wbc_batched, "unfused": self.lib.hoqp_solve_wbc_unfused,
              "noalloc": self.lib.hoqp_solve_wbc_noalloc}[entry]
        self._check(fn(self.h, C.byref(wi), C.byref(wo), self._stream()), "hoqp_solve_wbc_" + entry)
        return out

    def solve_wbc_host(self, inp):
        """hoqp_solve_wbc_host: numpy arrays in host memory in, numpy arrays out (the entry the C++ shim's
        CalcJointCmd uses; one H2D copy, one launch, one D2H copy through the handle's staging arenas;
        returns when the results are in the output arrays)."""
        import numpy as np
        B = inp["M"].shape[0]
        keep = []
        wi = WbcInputs()
        wi.batch = B
        for name in ("M", "h", "Jfoot", "Jdotu", "footpos", "Jb", "q", "u", "base_cmd", "leg_cmd"):
            arr = inp.get(name)
            if arr is not None:
                arr = np.ascontiguousarray(arr, dtype=np.float64)
                keep.append(arr)
                setattr(wi, name, arr.ctypes.data)
        ct = np.ascontiguousarray(inp["contact"], dtype=np.int32)
        keep.append(ct)
        wi.contact = ct.ctypes.data
        shapes = dict(x=(30, np.float64), qdd=(12, np.float64), tau=(12, np.float64), v=(40, np.float64),
                      flags=(5, np.int32), eq_res_max=(5, np.float64), ineq_viol_max=(5, np.float64),
                      level_res=(5, np.float64), active_mask=(None, np.uint64), status=(None, np.int32),
                      iters=(5, np.int32), nz_levels=(5, np.int32))
        out = {k: np.zeros((B,) if w is None else (B, w), dtype=dt) for k, (w, dt) in shapes.items()}
        wo = WbcOutputs()
        for name, arr in out.items():
            setattr(wo, name, arr.ctypes.data)
        self._check(self.lib.hoqp_solve_wbc_host(self.h, C.byref(wi), C.byref(wo)), "hoqp_solve_wbc_host")
        return out


class ShardedSolver:
    """hoqp_sharded_*: the one-process multi-GPU driver (one worker thread + stream + handle per shard).
    `devices`: HIP ordinal of every shard (may repeat)."""

    def __init__(self, cfg: Config, devices):
        import torch

        self.lib = load_library()
        self.torch = torch
        self.devices = list(devices)
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.lib.hoqp_sharded_create(C.byref(cfg), len(self.devices), (C.c_int * len(self.devices))(*self.devices),
                                          C.byref(h))
        if rc != 0:
            raise RuntimeError(f"hoqp_sharded_create failed: {rc}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.hoqp_sharded_destroy(self.h)
            self.h = None

    def solve_wbc(self, shards_in, shards_out, gather=None, want_stats=True):
        """shards_in / shards_out: one dict of device tensors per shard (as HoQpSolver.solve_wbc takes; a shard
        may be None: empty).  gather: dict with "device" and optional x / tau / v / status tensors of the total
        size on that device.  Returns the reduced statistics (or None).  Blocks until done."""
        t = self.torch
        R = len(self.devices)
        assert len(shards_in) == R and len(shards_out) == R
        wi, wo = (WbcInputs * R)(), (WbcOutputs * R)()
        for r in range(R):
            if shards_in[r] is None:
                continue
            d = shards_in[r]
            wi[r].batch = d["M"].shape[0]
            for name in ("M", "h", "Jfoot", "Jdotu", "footpos", "Jb", "q", "u", "base_cmd", "leg_cmd", "contact"):
                arr = d.get(name)
                if arr is not None:
                    assert arr.is_contiguous() and arr.is_cuda and arr.device.index == self.devices[r], name
                    setattr(wi[r], name, arr.data_ptr())
            for name, _ in WbcOutputs._fields_:
                if shards_out[r].get(name) is not None:
                    setattr(wo[r], name, shards_out[r][name].data_ptr())
        g = None
        if gather is not None:
            g = WbcGather()
            g.root_device = gather["device"]
            for name in ("x", "tau", "v", "status"):
                if gather.get(name) is not None:
                    setattr(g, name, gather[name].data_ptr())
        st = ShardStats() if want_stats else None
        t.cuda.synchronize()     # inputs written on torch streams are complete before the worker streams read them
        rc = self.lib.hoqp_solve_wbc_sharded(self.h, wi, wo, C.byref(g) if g is not None else None,
                                             C.byref(st) if st is not None else None)
        if rc != 0:
            msg = self.lib.hoqp_sharded_last_error(self.h)
            raise RuntimeError(f"hoqp_solve_wbc_sharded failed ({rc}): {msg.decode() if msg else ''}")
        return st
