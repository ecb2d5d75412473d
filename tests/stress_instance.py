"""One WBC instance through the generic task entry on both sides, level by level (by hand, GPU box):
    python tests/stress_instance.py mode kind seed N index"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import anymal_model as am, hoqp
from oracle import oracle
mode, kind, seed, N, i = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
np.set_printoptions(linewidth=220, precision=4)
inp = am.make_wbc_inputs(am.sample_states(N, seed, kind))
tasks, n = oracle.wbc_build(inp, i, oracle.default_wbc_params(mode))
cfg = oracle.default_cfg(mode)
r = oracle.solve_tasks(tasks, n, cfg)
s = hoqp.HoQpSolver(hoqp.default_config(mode))
dt = [{k: torch.as_tensor(v[None]).contiguous().cuda() for k, v in t.items()} for t in tasks]
out = s.solve_tasks(dt, n); s.synchronize()
o = {k: v.cpu().numpy() for k, v in out.items()}
print("n", n, "m", r["m"], "p", r["p"], "rc", r["rc"], "nz ref", r["nz_levels"], "gpu", o["nz_levels"][0], "status", o["status"][0])
Zprev = np.eye(n)
for k, t in enumerate(tasks):
    xr = r["x_levels"][k]; xg = o["x_levels"][0, k, :n]
    d = xg - xr
    line = f"level {k}: |x|max={np.abs(xr).max():.3e} max|dx|={np.abs(d).max():.3e} max|dx[:18]|={np.abs(d[:18]).max():.3e}"
    if "A" in t:
        AZ = t["A"] @ Zprev
        sv = np.linalg.svd(AZ, compute_uv=False)
        line += f" sv(AZ) max={sv[0]:.3e} min={sv[min(AZ.shape)-1]:.3e} tail={sv[-3:]}"
        line += f" res ref={np.linalg.norm(t['A']@xr-t['b']):.6e} gpu={np.linalg.norm(t['A']@xg-t['b']):.6e}"
    print(line)
    Zprev = r["Z_levels"][k]
print("dx final", o["x"][0][:n] - r["x_levels"][-1])
print("x ref", r["x_levels"][-1])
