#!/usr/bin/env python3
"""Throughput of the three WBC entry points on one 4096-state stance batch (device-resident, reference semantics):
fused (hoqp_solve_wbc_batched), unfused (task dump + generic solve, hoqp_solve_wbc_unfused) and the generic
solve alone on stored task blocks (hoqp_solve_tasks_batched).  usage: python tools/entry_rates.py [batch]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import anymal_model as am  # noqa: E402
from quadruped_locomotion_amd import hoqp  # noqa: E402


def rate(fn, batch, reps=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return batch * reps / (time.perf_counter() - t0)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    s = hoqp.HoQpSolver(hoqp.default_config("reference"))
    d = s.to_device(am.make_wbc_inputs(am.sample_states(batch, 1234, "stance")))
    out = s.alloc_wbc_outputs(batch)
    out["x_levels"] = None
    res = {"batch": batch}
    res["fused_wbc_entry"] = rate(lambda: s.solve_wbc(d, out), batch)
    res["unfused_wbc_entry"] = rate(lambda: s.solve_wbc(d, out, entry="unfused"), batch)
    tasks, n_inst, m_inst, p_inst = s.build_wbc_tasks(d)
    s.synchronize()
    tout = s.alloc_task_outputs(batch, 30, list(hoqp.WBC_M), list(hoqp.WBC_P), False)
    tout["x_levels"] = None
    res["generic_entry_on_stored_tasks"] = rate(
        lambda: s.solve_tasks(tasks, 30, out=tout, n_inst=n_inst, m_inst=m_inst, p_inst=p_inst), batch)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
