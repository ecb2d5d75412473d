"""Generates tests/golden/wbc_golden.npz.

The reference ships no golden vectors for this path and cannot be run
here, so these fixtures are authored by the build: inputs from the synthetic model, expected
outputs from the CPU oracle (oracle/hoqp_oracle.c, least-squares form in long double), accepted only
if two other implementations agree to 1e-7 relative: the oracle's literal H + eps I solver in double
and the independent numpy prototype (tests/proto_levelsolver.py: different algorithm, orthogonal
factorizations).  PARITY UNPINNED with respect to the real reference binary.

wbc_offenders.npz: the states on which round 1's random sweeps (profiles/r01_stress_parity.txt) found
the HIP path and the round-1 oracle more than 1e-6 apart, with the authoritative outputs and the
conditioning figures of the level that causes it (sigma_min of A Z at the tracking level, eps /
sigma_min^2 = relative size of the reference's eps_ bias, u * kappa^2 = what forming H in double costs).
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
from oracle import anymal_model as am  # noqa: E402
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
        lit, rc = orc.wbc_solve_batch(inp, orc.default_cfg(mode, solver="literal"), prm)
        assert rc == 0, name
        assert (np.abs(lit["x"] - ref["x"]) / np.maximum(1.0, np.abs(ref["x"]))).max() < 1e-7, name
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
    offenders()


# (semantics, contact kind, sample_states seed, batch the sweep drew, index in it, what round 1 saw)
OFFENDERS = [
    ("reference", "three", 113, 65536, 12296, 4.3e-6), ("reference", "three", 113, 65536, 21460, 4.3e-6),
    ("paper", "stance", 111, 65536, 28878, 1.76e-6), ("reference", "three", 413, 32768, 27930, 5.2e-6),
    ("paper", "stance", 411, 32768, 13883, 1.1e-6), ("reference", "three", 513, 32768, 3549, 3.6e-6),
]


def offenders():
    blob = {}
    for mode, kind, seed, N, idx, seen in OFFENDERS:
        st = am.sample_states(N, seed, kind)
        inp = am.make_wbc_inputs({k: v[idx:idx + 1] for k, v in st.items()})
        prm = orc.default_wbc_params(mode)
        ref, rc = orc.wbc_solve_batch(inp, orc.default_cfg(mode), prm)
        assert rc == 0
        lit, _ = orc.wbc_solve_batch(inp, orc.default_cfg(mode, solver="literal"), prm)
        tasks, n = orc.wbc_build(inp, 0, prm)
        r = orc.solve_tasks(tasks, n, orc.default_cfg(mode))
        Z, worst = np.eye(n), (0, 1.0, 1.0)
        for k, t in enumerate(tasks):
            if "A" in t and Z.shape[1]:
                sv = np.linalg.svd(t["A"] @ Z, compute_uv=False)
                rk = int((sv > 1e-12 * sv[0]).sum())
                if sv[0] / sv[rk - 1] > worst[1] / worst[2]:
                    worst = (k, sv[0], sv[rk - 1])
            Z = r["Z_levels"][k]
        name = f"{mode}_{kind}_{seed}_{idx}"
        for k, v in inp.items():
            blob[f"{name}/in/{k}"] = v
        for k in ("x", "qdd", "tau", "v", "flags", "level_res", "nz_levels", "eq_res_max", "ineq_viol_max", "active_mask", "n"):
            blob[f"{name}/out/{k}"] = ref[k]
        kap = worst[1] / worst[2]
        lit_err = float((np.abs(lit["x"] - ref["x"]) / np.maximum(1.0, np.abs(ref["x"]))).max())
        blob[f"{name}/meta"] = np.array([mode, "{}"])
        blob[f"{name}/cond"] = np.array([worst[0], worst[1], worst[2], kap, 1.1e-16 * kap ** 2, 1e-12 / worst[2] ** 2, lit_err, seen])
        print(f"{name}: level {worst[0]} sigma_max {worst[1]:.3e} sigma_min {worst[2]:.3e} kappa {kap:.2e} "
              f"u kappa^2 {1.1e-16 * kap ** 2:.1e} eps/sigma_min^2 {1e-12 / worst[2] ** 2:.1e} literal-vs-lsq {lit_err:.1e} "
              f"|x|inf {np.abs(ref['x']).max():.2e}")
    np.savez_compressed(os.path.join(HERE, "wbc_offenders.npz"), **blob)
    print("wrote wbc_offenders.npz", len(blob), "arrays")


if __name__ == "__main__":
    main()
