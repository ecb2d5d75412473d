#!/usr/bin/env python3
"""Kernel times of the SURVEY 8f rows (N1 dynamics producer, N2 post-solve step, N3 command sampler,
N4 base walk trajectory) on one GPU: HIP events around K launches on the launch stream, algorithmic
bytes per state / launch time against the HBM peak.  Prints one JSON object.
    python tools/bench_next.py [--batch 65536] [--steps 50]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import anymal_model as am  # noqa: E402
from quadruped_locomotion_amd import hoqp  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timed(fn, steps, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps  # ms per launch (launches back to back on one stream)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    B = args.batch
    s = hoqp.HoQpSolver(hoqp.default_config("reference"))
    st = am.sample_states(B, 5, "stance")
    rng = np.random.default_rng(6)
    dq, du = torch.as_tensor(st["q"]).cuda(), torch.as_tensor(st["u"]).cuda()
    f64 = lambda *shape: torch.as_tensor(rng.normal(size=shape)).cuda()  # noqa: E731
    res = {}

    def row(name, ms, bytes_per_state, note):
        gbs = bytes_per_state * B / (ms * 1e-3) / 1e9
        res[name] = dict(ms_per_launch=ms, states_per_s=B / (ms * 1e-3), alg_bytes_per_state=bytes_per_state,
                         achieved_gbs=gbs, hbm_frac=gbs / HBM_PEAK_GBS, note=note)

    dyn = s.wbc_dynamics(dq, du)
    row("N1 hoqp_wbc_dynamics_batched", timed(lambda: s.wbc_dynamics(dq, du), args.steps),
        8 * (19 + 18 + 324 + 18 + 216 + 12 + 12 + 108),
        "one wavefront per state; includes the allocation of its six output tensors by the Python binding")
    qdd, tau, qj, qdj = f64(B, 12), f64(B, 12), f64(B, 12), f64(B, 12)
    qc, qdc, tc = f64(B, 12), f64(B, 12), torch.empty(B, 12, dtype=torch.float64, device="cuda")
    row("N2 hoqp_wbc_poststep_batched", timed(lambda: s.wbc_poststep(2.5e-3, 40.0, 1.0, qdd, tau, qj, qdj, qc, qdc, tc), args.steps),
        8 * 12 * 9, "streaming: 6 reads + 3 writes per joint")
    g = hoqp.default_gait("crawl")
    time, vel = torch.as_tensor(rng.uniform(0, 10, B)).cuda(), f64(B, 2)
    st0 = dyn["footpos"][:, :, :2].contiguous()
    row("N3 hoqp_gait_sample_batched", timed(lambda: s.gait_sample(g, time, vel, st0), args.steps),
        8 * (1 + 2 + 8 + 36) + 16, "one thread per state and leg; includes the allocation of the two outputs")
    pose = f64(B, 3)
    row("N3 hoqp_standup_sample_batched", timed(lambda: s.standup_sample(time, 2.0, 0.5, pose), args.steps),
        8 * (1 + 3 + 9) + 16, "one thread per state")
    p0, pf = f64(B, 2), f64(B, 2)
    row("N4 hoqp_base_walk_sample_batched", timed(lambda: s.base_walk_sample(time, p0, pf, 0.5), args.steps),
        8 * (1 + 2 + 2 + 9), "one thread per state; closed form of the spline program")
    # the whole device-resident cycle: commands (N3) -> dynamics terms (N1) -> WBC solve -> post-solve step (N2)
    inp = am.make_wbc_inputs(st)
    dinp = s.to_device(inp)
    out = s.alloc_wbc_outputs(B)
    gfix = hoqp.default_gait("crawl", fix_derivatives=1)

    def cycle():
        d = s.wbc_dynamics(dq, du)
        lc, ct = s.gait_sample(gfix, time, vel, d["footpos"][:, :, :2].contiguous())
        for k in ("M", "h", "Jfoot", "Jdotu", "footpos", "Jb"):
            dinp[k] = d[k]
        dinp["leg_cmd"], dinp["contact"] = lc, ct
        s.solve_wbc(dinp, out)
        s.wbc_poststep(2.5e-3, 40.0, 1.0, out["qdd"], out["tau"], qj, qdj, qc, qdc, tc)

    ms = timed(cycle, max(5, args.steps // 5), warmup=2)
    res["cycle N3 -> N1 -> solve -> N2 (crawl gait, reference semantics)"] = dict(
        ms_per_launch=ms, states_per_s=B / (ms * 1e-3),
        note="device-resident control cycle for a batch of states; the solve is %.0f %% of it" % 0.0)
    ms_solve = timed(lambda: s.solve_wbc(dinp, out), max(5, args.steps // 5), warmup=2)
    res["cycle N3 -> N1 -> solve -> N2 (crawl gait, reference semantics)"]["note"] = (
        "device-resident control cycle for a batch of states; the solve alone takes %.3f ms (%.0f %% of it)"
        % (ms_solve, 100.0 * ms_solve / ms))
    print(json.dumps(dict(batch=B, steps=args.steps, hbm_peak_gbs=HBM_PEAK_GBS, rows=res), indent=1))


if __name__ == "__main__":
    main()
