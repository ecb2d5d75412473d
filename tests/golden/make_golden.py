"""Generates tests/golden/wbc_golden.npz.

The reference ships no golden vectors for this path and cannot be run
here, so these fixtures are authored by the build: inputs from the synthetic model, expected
outputs from the CPU oracle (oracle/hoqp_oracle.c), accepted only if the independent numpy
prototype (tests/proto_levelsolver.py: different algorithm, orthogonal factorizations) agrees
to 1e-7 relative.  PARITY UNPINNED with respect to the real reference binary.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as orc  # noqa: E402
from quadruped_locomotion_amd import anymal_model as am  # noqa: E402
import proto_levelsolver as pl  # noqa: E402

CASES = [
    ("c1_standing_reference", "standing", 0, "reference", {}),
    ("c2_stance_reference", "stance", 1234, "reference", {}),
    ("c3_trot_reference", "trot", 2345, "reference", {}),
    ("c3_trot_paper", "trot", 2345, "paper", {}),
    ("c5_three_paper_tight", "three", 4567, "paper", dict(mu=0.2, tau_max=40.0, tau_min=-40.0)),
]


def main():
    orc.build()
    blob = {}
    for name, kind, seed, mode, kw in CASES:
        st = am.standing_state() if kind == "standing" else am.sample_states(6, seed, kind)
        inp = am.make_wbc_inputs(st)
        prm = orc.default_wbc_params(mode, **kw)
        ref, rc = orc.wbc_solve_batch(inp, orc.default_cfg(mode), prm)
        assert rc == 0, name
        for i in range(inp["M"].shape[0]):
            tasks, n = orc.wbc_build(inp, i, prm)
            ch = pl.solve_tasks(tasks, n, gate_prior_ineq=(mode == "reference"),
                                eps=1e-12 if mode == "reference" else 0.0)
            err = np.abs(ch.x - ref["x"][i][:n]).max() / max(1.0, np.abs(ref["x"][i]).max())
            assert err < 1e-7, (name, i, err)
        for k, v in inp.items():
            blob[f"{name}/in/{k}"] = v
        for k in ("x", "qdd", "tau", "v", "flags", "level_res", "nz_levels", "eq_res_max", "ineq_viol_max",
                  "active_mask", "n"):
            blob[f"{name}/out/{k}"] = ref[k]
        blob[f"{name}/meta"] = np.array([mode, repr(kw)])
    np.savez_compressed(os.path.join(HERE, "wbc_golden.npz"), **blob)
    print("wrote", os.path.join(HERE, "wbc_golden.npz"), len(blob), "arrays")


if __name__ == "__main__":
    main()
