#!/usr/bin/env python3
"""Single-state drop-in latency of the C++ shim on the GPU box: writes one WBC
state as a tests/shim_check fixture and runs `tests/shim_check <fixture> <reps>`; the binary times
HoQpController::CalcJointCmd (host buffers in, host buffers out: one H2D copy, one launch, one D2H
copy), the same with UseBuiltinDynamics, and the five-object HoQpProblem chain.

usage: python tools/shim_latency.py [reps] > gpurun_out/shim_latency.txt
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import anymal_model as am  # noqa: E402


def fmt(a):
    return " ".join(repr(float(v)) for v in np.asarray(a).reshape(-1))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    for contact, seed in (("stance", 11), ("three", 77)):
        inp = am.make_wbc_inputs(am.sample_states(3, seed, contact))
        i = 1
        parts = ["wbc 55"]
        for k in ("M", "h", "Jfoot", "Jdotu", "footpos", "Jb", "q", "u"):
            parts.append(fmt(inp[k][i]))
        bc, lc = inp["base_cmd"][i], inp["leg_cmd"][i]
        parts += [fmt(bc[0:3]), fmt(bc[3:6]), fmt(bc[6:9]), fmt(lc[0:12]), fmt(lc[12:24]), fmt(lc[24:36]),
                  fmt(inp["contact"][i].astype(float))]
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            f.write("\n".join(parts) + "\n")
            path = f.name
        res = subprocess.run([os.path.join(ROOT, "tests", "shim_check"), path, str(reps)], capture_output=True, text=True,
                             timeout=600)
        os.unlink(path)
        if res.returncode != 0:
            print(res.stderr)
            sys.exit(1)
        for line in res.stdout.splitlines():
            if line.startswith("latency_us"):
                print(f"{contact}: {line}")


if __name__ == "__main__":
    main()
