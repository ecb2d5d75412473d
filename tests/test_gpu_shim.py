"""GPU test of the C++ shim (include/hoqp.hpp): the reference's own class API driven exactly
as HoQpController::ConstructOptProblems / CalcJointCmd drive it, checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle import anymal_model as am

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "shim_check")


def _run(tmp_path, text):
    assert os.path.exists(EXE), "tests/shim_check not built (run __graft_entry__.build())"
    fx = tmp_path / "fixture.txt"
    fx.write_text(text)
    res = subprocess.run([EXE, str(fx)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    return res.stdout.splitlines()


def _fmt(a):
    return " ".join(repr(float(v)) for v in np.asarray(a).reshape(-1))


@pytest.mark.parametrize("mode,sem", [("reference", 0x37), ("paper", 0)])
def test_shim_problem_chain(tmp_path, oracle, mode, sem):
    rng = np.random.default_rng(5)
    n, spec = 9, [(3, 0), (0, 5), (4, 0), (2, 0)]
    tasks, lines = [], [f"tasks {sem}", f"{n} {len(spec)}"]
    for m, p in spec:
        t = {}
        lines.append(f"{m} {p}")
        if m:
            t["A"], t["b"] = rng.normal(size=(m, n)), rng.normal(size=m)
            lines += [_fmt(t["A"]), _fmt(t["b"])]
        if p:
            t["D"], t["f"] = rng.normal(size=(p, n)), rng.normal(size=p) + 0.5
            lines += [_fmt(t["D"]), _fmt(t["f"])]
        tasks.append(t)
    out = _run(tmp_path, "\n".join(lines) + "\n")
    ref = oracle.solve_tasks(tasks, n, oracle.default_cfg(mode))
    assert ref["rc"] == 0
    lv = [l for l in out if l.startswith("level")]
    xs = [np.array(l.split()[2:], dtype=float) for l in out if l.startswith("x ")]
    vs = [np.array(l.split()[2:], dtype=float) for l in out if l.startswith("v ")]
    assert len(lv) == len(spec)
    for k in range(len(spec)):
        f = lv[k].split()
        assert int(f[7]) == 0  # status
        assert int(f[5]) == sum(p for _, p in spec[:k + 1])  # GetStackedNumSlackVars
        nz_ref = ref["nz_levels"][k]
        assert int(f[3]) == nz_ref, (k, f, nz_ref)
        stacked = np.concatenate([ref["v_levels"][j] for j in range(k + 1)]) if vs[k].size else np.zeros(0)
        assert np.allclose(vs[k], stacked, rtol=1e-6, atol=1e-6)
    # last level: 9 equality rows in 9 variables -> unique x
    assert np.allclose(xs[-1], ref["x_levels"][-1], rtol=1e-6, atol=1e-6)
    assert out[-1].split() == ["stacked", str(sum(m for m, _ in spec)), str(sum(p for _, p in spec))]
    # contents and ORDER of the stacked task of the last problem: the current task first, then the previous
    # stack (ho_qp_problem.cpp:83-86, task_definition.hpp:13-30), i.e. newest level first; bit-exact copies
    def mat(name):
        f = [l for l in out if l.startswith(name + " ")][0].split()
        return np.array(f[3:], dtype=float).reshape(int(f[1]), int(f[2]))

    def vec(name):
        f = [l for l in out if l.startswith(name + " ")][0].split()
        return np.array(f[2:], dtype=float)

    newest_first = list(reversed(tasks))
    assert np.array_equal(mat("stackedA"), np.concatenate([t["A"] for t in newest_first if "A" in t]))
    assert np.array_equal(vec("stackedb"), np.concatenate([t["b"] for t in newest_first if "b" in t]))
    assert np.array_equal(mat("stackedD"), np.concatenate([t["D"] for t in newest_first if "D" in t]))
    assert np.array_equal(vec("stackedf"), np.concatenate([t["f"] for t in newest_first if "f" in t]))
    assert vec("vlast").size == spec[-1][1]      # GetSlackSolutions: the last level's own slacks (none here)


def test_shim_controller(tmp_path, oracle):
    inp = am.make_wbc_inputs(am.sample_states(3, 77, "three"))
    i = 1
    parts = ["wbc 55"]
    for k in ("M", "h", "Jfoot", "Jdotu", "footpos", "Jb", "q", "u"):
        parts.append(_fmt(inp[k][i]))
    bc, lc = inp["base_cmd"][i], inp["leg_cmd"][i]
    parts += [_fmt(bc[0:3]), _fmt(bc[3:6]), _fmt(bc[6:9]), _fmt(lc[0:12]), _fmt(lc[12:24]), _fmt(lc[24:36]),
              _fmt(inp["contact"][i].astype(float))]
    out = _run(tmp_path, "\n".join(parts) + "\n")
    ref, rc = oracle.wbc_solve_batch({k: v[i:i + 1] for k, v in inp.items()})
    assert rc == 0
    got = {l.split()[0]: np.array(l.split()[2:], dtype=float) for l in out if l.split()[0] in ("qdd", "tau", "x")}
    for key in ("qdd", "tau"):
        assert np.all(np.abs(got[key] - ref[key][0]) <= 1e-6 * np.maximum(1.0, np.abs(ref[key][0]))), key
    n = int(ref["n"][0])
    assert got["x"].size == n
    assert np.all(np.abs(got["x"] - ref["x"][0][:n]) <= 1e-6 * np.maximum(1.0, np.abs(ref["x"][0][:n])))
    # UseBuiltinDynamics: the library's own producer gives the same cycle (same model as the fixture)
    tb = [np.array(l.split()[2:], dtype=float) for l in out if l.startswith("tau_builtin")][0]
    assert np.all(np.abs(tb - ref["tau"][0]) <= 1e-6 * np.maximum(1.0, np.abs(ref["tau"][0])))
    fl = [l for l in out if l.startswith("flags")][0].split()
    assert int(fl[7]) == 0
    assert [int(v) for v in fl[1:6]][0] == 1


def test_bench_two_rank_rehearsal():
    """The N > 1 path of bench.py end to end on the one-GPU box: two ranks share device 0 (gloo in place of
    RCCL, which refuses two ranks on one device), workload c4 scaled down -- every step = HIP solve of the
    rank's shard + gather of x / tau / v / status to rank 0 + all-reduce of the statistics."""
    import json
    import sys

    import socket

    root = os.path.dirname(HERE)
    with socket.socket() as sk:                       # a free port, not a fixed one
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HOQP_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c4", "--batch", "8191",
           "--steps", "3", "--warmup", "1", "--latency-batches", "5"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["gather"] == "every_step"
    assert d["config"]["batch_total"] == 8191 and d["config"]["batch_per_gpu"] == 4096      # ragged shards
    st = d["config"]["reduced_stats"]
    assert st["instances"] == 8191 and st["status_nonzero"] == 0 and np.isfinite(st["eq_res_max"])
    assert d["value"] > 0 and d["config"]["gather_ms"] > 0
    # the collection is reported separately from the solve, per step, and summed
    c = d["config"]
    assert c["step_solve_ms"] > 0 and c["step_gather_ms"] > 0
    assert abs(c["step_solve_plus_gather_ms"] - (c["step_solve_ms"] + c["step_gather_ms"])) < 1e-9
    assert c["gathered"] == ["x", "tau", "v", "status"] and "x_levels" not in c["outputs"]


def test_bench_pipelined_option():
    """bench.py --pipeline-streams 2: the extra two-batches-in-flight measurement (two handles, two streams,
    two output sets) gives the same x as the single-stream step and is reported beside, not as, the value."""
    import json
    import sys

    root = os.path.dirname(HERE)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--batch", "1024", "--steps", "5",
           "--warmup", "1", "--latency-batches", "5", "--pipeline-streams", "2"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["pipelined"]["streams"] == 2 and d["pipelined"]["x_identical_to_single_stream"] is True
    assert d["pipelined"]["value"] > 0 and d["value"] > 0 and d["n_gpus"] == 1
