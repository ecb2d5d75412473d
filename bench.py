#!/usr/bin/env python3
"""bench.py -- fp64 HO-QP solves/s on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (hoqp_solve_wbc_batched: ONE kernel -- task construction,
5-level HO-QP chain, torque map and validity flags) over one batch of synthetic ANYmal-C-like
states that is already resident in HBM.

Workloads (`config.workload`):
  default   BASELINE.json configs[1]: 4096 randomised 4-contact stance states per GPU, reference
            semantics.  N > 1: the same per rank (weak scaling), one process per GPU; the solve has
            no exchange step; results are collected once after the timed loop (`gather_ms`).
  --workload c4   configs[3]: 1,048,576 randomised stance states (seed 3456) split contiguously over
            the ranks (parallel.shard_bounds), every step = solve of the shard + RCCL gather of
            x / tau / v / status to rank 0 + all-reduce of the residual statistics, all inside the
            timed region (strong scaling).  At N = 1 the whole batch runs on the one GPU.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VEC_PEAK_TF = 78.6    # public spec figure, vector FP64

# which BASELINE.json configuration a --contact choice restates
WORKLOAD_LABEL = {"stance": "configs[1]", "trot": "configs[2] (trot mix)", "three": "configs[4] (swing tracking)",
                  "crawl": "crawl gait of the reference via the command sampler"}


def alg_bytes_per_solve(nc, ns, general_jb=False):
    """SURVEY.md 8(d), fused WBC entry: inputs 8*(324 M + 18 h + 57 per leg + 6 base rhs) + 4 B mask,
    outputs 8*(54 + 7 n_c) + 8.  5,276 B for n_c = 4; a general J_b (6 x 18) adds 864 B."""
    return 8 * (324 + 18 + 57 * (nc + ns) + 6) + 4 + 8 * (54 + 7 * nc) + 8 + (864 if general_jb else 0)


def alg_flops_per_solve(nc, ns, gate):
    """SURVEY.md 8(d) convention: dense textbook counts, Z_-1 = I and x_-1 = 0 exploited, no sparsity,
    zero active-set iterations.  210,984 for n_c = 4 under reference semantics, 318,984 under paper."""
    n = 18 + 3 * nc
    ms = [6, 0, 3 * nc, 6 + 3 * ns, 3 * nc]
    ps = [0, 24 + 4 * nc, 0, 0, 0]
    nz, P, total = n, 0, 0.0
    for k in range(5):
        m, p = ms[k], ps[k]
        looks = (p > 0) if gate else (p > 0 or P > 0)
        rows = (p + P) if looks else 0
        if k > 0:
            total += 2 * m * n * nz + 2 * m * n + 2 * n * nz
            total += 2 * rows * n * nz + 2 * rows * n
        total += 2 * m * nz * nz + 2 * m * nz
        if m > 0:
            total += nz ** 3 / 3 + 2 * nz * nz
            if k < 4:
                total += 2 * m * m * (nz - m / 3) + 4 * n * nz * m
        P += p
        nz -= min(m, nz)
    return total


def cpu_model():
    """model name plus family / model / stepping from /proc/cpuinfo (virtualised hosts report a bare
    "AMD EPYC" as the name)."""
    info = {}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if ":" not in line:
                    if info:
                        break
                    continue
                k, v = (s.strip() for s in line.split(":", 1))
                if k in ("model name", "cpu family", "model", "stepping", "cpu MHz", "vendor_id") and k not in info:
                    info[k] = v
    except OSError:
        pass
    if not info:
        return "unknown"
    return "{} (vendor {}, family {}, model {}, stepping {}, {} MHz; {} logical CPUs visible)".format(
        info.get("model name", "?"), info.get("vendor_id", "?"), info.get("cpu family", "?"), info.get("model", "?"),
        info.get("stepping", "?"), info.get("cpu MHz", "?"), os.cpu_count())


def cpu_baseline(inp, mode, sample, threads, prm_kw, detail=False):
    """The oracle ("port") timed on the host cores over a bounded sample.  Timed with its literal
    double-precision solver (H = (AZ)'AZ + eps I formed and factored in double, as a restatement of the
    reference computes it); the long-double least-squares solver is the parity checker, not a baseline."""
    from oracle import oracle as orc

    orc.build()
    sub = {k: v[:sample] for k, v in inp.items()}
    cfg, prm = orc.default_cfg(mode, solver="literal"), orc.default_wbc_params(mode, **prm_kw)
    orc.wbc_solve_batch({k: v[:8] for k, v in sub.items()}, cfg, prm, threads=threads)  # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        orc.wbc_solve_batch(sub, cfg, prm, threads=threads)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or reps >= 50:
            break
    res = dict(value=sample * reps / dt, unit="solves/s", cores=threads, kind="port",
               sample=f"{reps} passes over the first {sample} states of the same batch, "
                      f"oracle/hoqp_oracle.c literal double solver (-O3, OpenMP {threads} threads); "
                      f"Drake/Eigen reference unbuildable",
               cpu_model=cpu_model())
    if detail:
        # SURVEY.md 8(d) CPU baseline (i): one thread, configs[0] (the nominal standing state),
        # 10^4 repetitions, per-solve latency
        from oracle import anymal_model as am     # (CPU leg: the oracle's side of the fence)

        one = am.make_wbc_inputs(am.standing_state())
        lat = np.empty(10000)
        for i in range(lat.size):
            a = time.perf_counter()
            orc.wbc_solve_batch(one, cfg, prm, threads=1)
            lat[i] = time.perf_counter() - a
        res["single_thread_c1"] = dict(solves_per_s=float(1.0 / np.median(lat)), median_us=float(np.median(lat) * 1e6),
                                       p99_us=float(np.percentile(lat, 99) * 1e6), repetitions=int(lat.size),
                                       note="includes the ctypes call and state packing of one state")
    return res


def feasibility_match(out, inp, mode, sample, threads, prm_kw):
    """north star: "same feasibility and active constraints".  The HIP path's CheckSolutionValid flags
    (helper_functions.hpp:63-91) and active-row masks against the authoritative oracle on the first
    `sample` states: counts of identical entries, and of the differing ones how many sit inside the
    1e-7 band around the threshold they are compared with."""
    from oracle import oracle as orc

    orc.build()
    sub = {k: v[:sample] for k, v in inp.items()}
    prm = orc.default_wbc_params(mode, **prm_kw)
    ref, rc = orc.wbc_solve_batch(sub, orc.default_cfg(mode), prm, threads=threads)
    flags = out["flags"][:sample].cpu().numpy()
    mask = out["active_mask"][:sample].cpu().numpy().astype(np.uint64)
    x = out["x"][:sample].cpu().numpy()
    fdiff = flags != ref["flags"]
    band = (np.abs(ref["eq_res_max"] - 1e-7) <= 1e-7) | (np.abs(ref["ineq_viol_max"]) <= 1e-7)
    mdiff = mask ^ ref["active_mask"].astype(np.uint64)
    nbits = sum(bin(int(v)).count("1") for v in mdiff)
    # a differing active bit is in-band when the oracle's margin of that row is within 1e-7 of -1e-7
    bits_in_band = 0
    for i in np.nonzero(mdiff)[0]:
        tasks, n = orc.wbc_build(sub, int(i), prm)
        marg = tasks[1]["D"] @ ref["x"][i][:n] - tasks[1]["f"]
        bits_in_band += sum(1 for r in range(len(marg)) if (int(mdiff[i]) >> r) & 1 and abs(marg[r] + 1e-7) <= 1e-7)
    xerr = np.abs(x - ref["x"]) / np.maximum(1.0, np.abs(ref["x"]))
    return {"states": int(sample), "oracle_rc": int(rc),
            "flags_equal": int((~fdiff).sum()), "flags_total": int(fdiff.size),
            "flags_differ_in_band": int((fdiff & band).sum()), "flags_differ_outside_band": int((fdiff & ~band).sum()),
            "active_mask_equal": int((mdiff == 0).sum()), "active_mask_total": int(mdiff.size),
            "active_bits_differ": int(nbits), "active_bits_differ_in_band": int(bits_in_band),
            "x_max_rel_err": float(xerr.max()), "x_over_1e-6": int((xerr.max(axis=1) > 1e-6).sum())}


def load_traffic(batch, mode, contact):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass of this same command
    (tools/profile.sh; separate --pmc runs, corrected by the factors tools/calib/fetch_calib.sh measured).
    Only for the profiled workload AND the profiled binary: the file carries the SHA-256 of the library it
    was measured on; any other library gets None and the reason."""
    import hashlib

    from quadruped_locomotion_amd import hoqp

    try:
        with open(hoqp.LIB_PATH, "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None, "library not readable"
    for name in ("r03_pmc_traffic.json",):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                tj = json.load(f)
        except (OSError, ValueError):
            continue
        if not (tj.get("batch") == batch and tj.get("semantics") == mode and tj.get("contact") == contact):
            return None, f"profiles/{name} was measured on another workload"
        if tj.get("library_sha256") != sha:
            return None, (f"profiles/{name} was measured on another build of libhoqp_hip.so "
                          f"(sha256 {str(tj.get('library_sha256'))[:12]}..., loaded {sha[:12]}...)")
        try:
            return float(tj["fetch_bytes_per_launch"] + tj["write_bytes_per_launch"]), tj.get("source")
        except KeyError:
            return None, f"profiles/{name} is incomplete"
    return None, "no PMC traffic file committed"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="default", choices=["default", "c4"])
    ap.add_argument("--batch", type=int, default=None, help="states per GPU per step (default 4096; c4: 1,048,576 in total)")
    ap.add_argument("--mode", default="reference", choices=["reference", "paper"])
    ap.add_argument("--contact", default="stance", choices=["stance", "trot", "three", "crawl"],
                    help="contact patterns: i.i.d. masks (stance / trot / three) or, for crawl, the pattern and swing "
                         "commands of the reference's crawl gait at random phases (hoqp_gait_sample_batched)")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--pipeline-streams", type=int, default=0,
                    help="extra, non-headline measurement after the timed region: this many batches in flight on as "
                         "many streams (default off, so that a rocprofv3 trace of the default command holds no "
                         "overlapped launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=512)
    ap.add_argument("--mu", type=float, default=None, help="friction coefficient (default: the reference's)")
    ap.add_argument("--tau-max", type=float, default=None, help="joint torque limit, symmetric (default: the reference's)")
    ap.add_argument("--cpu-detail", action="store_true",
                    help="also time the oracle single-threaded on configs[0] (10^4 repetitions)")
    ap.add_argument("--no-gather", action="store_true", help="skip the collection of results on rank 0")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="gather solutions to rank 0 inside every timed step (default workload: once, after the loop)")
    ap.add_argument("--latency-batches", type=int, default=1000, help="batches for the p50 / p99 latency sample")
    ap.add_argument("--skip-unread-slack", action="store_true",
                    help="NOT the headline: HOQP_OPT_SKIP_UNREAD_SLACK_LEVELS with v == NULL (the torque / friction "
                         "level's QP is not solved; x, tau and flags unchanged to rounding)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("HOQP_BENCH_REHEARSE"):
            # rehearsal of the N > 1 code path on a one-GPU box: every rank on device 0, gloo instead of
            # RCCL (RCCL refuses two ranks on one device).  Not a measurement.
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from quadruped_locomotion_amd import hoqp
    from quadruped_locomotion_amd import parallel
    from quadruped_locomotion_amd import state_sampler as ss

    torch.cuda.set_device(local_rank)
    prm_kw = {}
    if args.mu is not None:
        prm_kw["mu"] = args.mu
    if args.tau_max is not None:
        prm_kw.update(tau_max=args.tau_max, tau_min=-args.tau_max)
    cfg_kw = dict(prm_kw)
    if args.skip_unread_slack:
        cfg_kw["options"] = hoqp.OPT_SKIP_UNREAD_SLACK_LEVELS
    solver = hoqp.HoQpSolver(hoqp.default_config(args.mode, device=local_rank, **cfg_kw))
    c4 = args.workload == "c4"
    if c4:
        # one global batch (seed 3456), contiguous shards; every rank generates only its own
        # shard (states are i.i.d., so shard r is drawn from seed 3456 + r)
        total = args.batch or 1048576
        seed = 3456 if args.seed is None else args.seed
        lo, hi = parallel.shard_bounds(total, world, rank)
        nloc = hi - lo
        gather_in_step = world > 1 and not args.no_gather
    else:
        nloc = args.batch or 4096
        total = nloc * world
        seed = 1234 if args.seed is None else args.seed
        gather_in_step = world > 1 and args.gather_every_step and not args.no_gather
    kind = "stance" if args.contact == "crawl" else args.contact
    # Only the states and commands come from the host (state_sampler); the dynamics terms M, h, J_foot, J_dot u,
    # foot positions and J_b are produced on the device by hoqp_wbc_dynamics_batched (what the
    # reference's UpdateModelDynamics pulls from Drake).  The CPU legs below read the device arrays back.
    st = ss.sample_states(nloc, seed + rank, kind)
    dq = torch.as_tensor(st["q"]).contiguous().to(solver.device)
    du = torch.as_tensor(st["u"]).contiguous().to(solver.device)
    dinp = solver.wbc_dynamics(dq, du)
    dinp.update(q=dq, u=du, base_cmd=torch.as_tensor(st["base_cmd"]).contiguous().to(solver.device),
                contact=torch.as_tensor(st["contact"].astype(np.int32)).contiguous().to(solver.device))
    lc = torch.as_tensor(st["leg_cmd"]).contiguous().to(solver.device)
    if np.any(st["contact"] == 0):
        lc[:, 0:12] += dinp["footpos"].reshape(-1, 12)   # swing target = current foot + offset
    dinp["leg_cmd"] = lc
    solver.synchronize()
    head = min(nloc, max(args.cpu_sample, 1))
    inp = {k: v[:head].cpu().numpy() for k, v in dinp.items()}
    inp["contact_all"] = st["contact"]
    del st
    if args.contact == "crawl":
        # commands from the gait sampler on the device (2-, 3- and 4-contact states)
        rng = np.random.default_rng(seed + rank + 77)
        t_gait = torch.as_tensor(rng.uniform(0.0, 10.0, nloc)).cuda()
        v_cmd = torch.as_tensor(rng.normal(0.0, 0.1, (nloc, 2))).cuda()
        lc, ct = solver.gait_sample(hoqp.default_gait("crawl", fix_derivatives=1), t_gait, v_cmd,
                                    dinp["footpos"][:, :, :2].contiguous())
        dinp["leg_cmd"], dinp["contact"] = lc, ct
        solver.synchronize()
        inp["leg_cmd"], inp["contact"] = lc[:head].cpu().numpy(), ct[:head].cpu().numpy()
        inp["contact_all"] = ct.cpu().numpy()
    # the per-level intermediate solutions are a diagnostic output (the reference's CalcJointCmd reads the last
    # problem's solution only, ho_qp_controller.cpp:13-16): not requested on the benchmarked path
    def alloc_out():
        o = solver.alloc_wbc_outputs(nloc)
        o["x_levels"] = None
        if args.skip_unread_slack:
            o["v"] = None
        return o

    # collection of results: per-instance x, tau, v*, status to rank 0 -- received straight
    # into preallocated [total, ...] arrays (parallel.ResultCollector: grouped isend / irecv, RCCL) -- and the
    # residual statistics all-reduced.  Inside the step (c4) it runs on a side stream with two output sets, so
    # that the collection of step k overlaps the solve of step k + 1.
    nbuf = 2 if gather_in_step else 1
    outs = [alloc_out() for _ in range(nbuf)]
    out = outs[0]
    payload_keys = [k for k in ("x", "tau", "v", "status") if out.get(k) is not None]
    collector = None
    if world > 1 and not args.no_gather:
        collector = parallel.ResultCollector({k: (tuple(out[k].shape[1:]), out[k].dtype) for k in payload_keys}, total,
                                             solver.device, dst=0, buffers=nbuf)
    side = torch.cuda.Stream() if gather_in_step else None
    solved_ev = [torch.cuda.Event() for _ in range(nbuf)]
    done_ev = [None] * nbuf
    timing_on = [False]
    solve_evs, gather_evs, gather_host = [], [], []

    def collect(o, b):
        reqs = collector.start({k: o[k] for k in payload_keys}, b)
        collector.wait(reqs)
        mx, sm = parallel.local_stats(o)
        return parallel.reduce_stats(mx, sm)

    step_no = [0]

    def step():
        b = step_no[0] % nbuf
        step_no[0] += 1
        main = torch.cuda.current_stream()
        if gather_in_step and done_ev[b] is not None:
            main.wait_event(done_ev[b])       # the collection that read this output set has finished
        if timing_on[0] and gather_in_step:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
        solver.solve_wbc(dinp, outs[b])
        if not gather_in_step:
            return
        if timing_on[0]:
            e1.record(main)
            solve_evs.append((e0, e1))
        solved_ev[b].record(main)
        with torch.cuda.stream(side):
            side.wait_event(solved_ev[b])
            if timing_on[0]:
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record(side)
                h0 = time.perf_counter()
            collect(outs[b], b)
            if timing_on[0]:
                g1.record(side)
                gather_evs.append((g0, g1))
                gather_host.append(time.perf_counter() - h0)
            done_ev[b] = torch.cuda.Event()
            done_ev[b].record(side)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    solver.timing(True)
    timing_on[0] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    timing_on[0] = False
    elapsed = t1 - t0
    kern_ms, kern_n = solver.kernel_time_ms()
    solver.timing(False)
    out = outs[(step_no[0] - 1) % nbuf] if step_no[0] else outs[0]
    step_solve_ms = step_gather_ms = None
    if gather_in_step and solve_evs:
        step_solve_ms = float(np.mean([a.elapsed_time(b) for a, b in solve_evs]))
        # device time of the collection on its own stream (RCCL); on a gloo rehearsal the transfers are host
        # calls, so the host time of the block is what there is to report
        step_gather_ms = (float(np.mean(gather_host)) * 1e3 if dist.get_backend() == "gloo"
                          else float(np.mean([a.elapsed_time(b) for a, b in gather_evs])))

    # p50/p99 batch latency (submit -> results visible), outside the timed region; SURVEY 8d: >= 1000
    # batches regardless of --steps (fewer only for batches that take longer than 10 ms each)
    lat = []
    nlat = args.latency_batches if elapsed / max(args.steps, 1) < 10e-3 else min(args.latency_batches, 100)
    for _ in range(nlat):
        torch.cuda.synchronize()
        a = time.perf_counter()
        solver.solve_wbc(dinp, out)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - a) * 1e3)
    lat = np.array(lat)

    # Two batches in flight (outside the timed region, never `value`): the same step issued alternately on
    # two streams with two handles and two output sets.  4096 states are exactly two per wave slot, so a
    # single launch ends with most slots idle while the slowest instances finish; a second batch in flight
    # fills that tail.  Reported to show what the launch quantisation costs.
    pipelined = None
    if world == 1 and args.pipeline_streams > 1 and not gather_in_step:
        ns = args.pipeline_streams
        streams = [torch.cuda.Stream() for _ in range(ns)]
        solvers = [hoqp.HoQpSolver(solver.cfg) for _ in range(ns)]
        pouts = [alloc_out() for _ in range(ns)]
        torch.cuda.synchronize()
        reps = max(args.steps, 20)
        for timed in (False, True):
            torch.cuda.synchronize()
            p0 = time.perf_counter()
            for i in range(reps):
                with torch.cuda.stream(streams[i % ns]):
                    solvers[i % ns].solve_wbc(dinp, pouts[i % ns])
            torch.cuda.synchronize()
            p1 = time.perf_counter()
        same = all(bool(torch.equal(pouts[j]["x"], out["x"])) for j in range(ns))
        pipelined = {"streams": ns, "steps": reps, "value": nloc * reps / (p1 - p0), "unit": "solves/s",
                     "ms_per_step": (p1 - p0) / reps * 1e3, "x_identical_to_single_stream": same,
                     "note": "batches of the same size issued alternately on %d streams; not the headline" % ns}
        for sv in solvers:
            sv.close()

    gather_ms, stats, gather_error = None, None, None
    if collector is not None:
        # (after the timed region in the default workload: a failure of the collection must not take the
        # measured line with it -- it is reported in config.gather_error instead)
        try:
            collect(out, 0)  # warm-up of the communicator
            barrier()
            g0 = time.perf_counter()
            mx, sm = collect(out, 0)
            barrier()
            gather_ms = (time.perf_counter() - g0) * 1e3
            stats = dict(eq_res_max=float(mx[0]), ineq_viol_max=float(mx[1]), status_nonzero=int(sm[0]),
                         iters_total=int(sm[1]), instances=int(sm[2]))
        except Exception as e:  # noqa: BLE001
            if gather_in_step:
                raise
            gather_error = repr(e)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64,
                            device="cpu" if dist.get_backend() == "gloo" else solver.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    status = out["status"].cpu().numpy()
    iters = out["iters"].cpu().numpy()
    if rank == 0:
        value = total * args.steps / elapsed
        kern_s = kern_ms * 1e-3
        ct = inp.pop("contact_all", inp["contact"])
        nc_i, ns_i = (ct == 1).sum(axis=1), (ct == 0).sum(axis=1)
        gate = args.mode == "reference"
        pats = {(int(a), int(b)) for a, b in zip(nc_i, ns_i)}
        bytes_mean = float(np.mean([alg_bytes_per_solve(a, b, dinp.get("Jb") is not None) for a, b in zip(nc_i, ns_i)]))
        fl = {pt: alg_flops_per_solve(pt[0], pt[1], gate) for pt in pats}
        flops_mean = float(np.mean([fl[(int(a), int(b))] for a, b in zip(nc_i, ns_i)]))
        ach_tf = flops_mean * nloc / kern_s / 1e12 if kern_s > 0 else 0.0
        ach_gbs = bytes_mean * nloc / kern_s / 1e9 if kern_s > 0 else 0.0
        traffic, traffic_src = load_traffic(nloc, args.mode, args.contact)
        res = {
            "metric": "fp64 HO-QP solves/sec (ANYmal-C, 4 contacts)" if args.contact == "stance" else
                      f"fp64 HO-QP solves/sec (ANYmal-C, {args.contact} contact patterns; NOT the 4-contact headline)",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if c4 else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (own ANYmal-C-like model, seed %d)" % seed,
            "config": {"workload": (f"configs[3]: {total} randomised stance states split over {world} GPU(s), "
                                    f"gather of x/tau/v/status + statistics all-reduce inside every step"
                                    if c4 else
                                    f"{WORKLOAD_LABEL[args.contact]}: {nloc} randomised {args.contact} states "
                                    f"per GPU") + f", 5-level WBC hierarchy, {args.mode} semantics",
                       "batch_per_gpu": nloc, "batch_total": total, "semantics": args.mode, "contact": args.contact,
                       "entry": "hoqp_solve_wbc_batched (fused: one kernel per step)" +
                                (" with HOQP_OPT_SKIP_UNREAD_SLACK_LEVELS and v == NULL -- NOT the headline configuration"
                                 if args.skip_unread_slack else ""),
                       "params": prm_kw or "reference defaults",
                       "gather": ("every_step" if gather_in_step else "after_loop")
                       if (world > 1 and not args.no_gather) else "none", "gather_ms": gather_ms,
                       # c4: mean per step of the solve (main stream) and of the collection (side stream, overlapping
                       # the next step's solve); their sum is what a step would take without the overlap
                       "step_solve_ms": step_solve_ms, "step_gather_ms": step_gather_ms,
                       "step_solve_plus_gather_ms": (step_solve_ms + step_gather_ms) if step_solve_ms is not None else None,
                       "outputs": sorted(k for k, v in out.items() if v is not None),
                       "gathered": payload_keys if collector is not None else [], "reduced_stats": stats,
                       "gather_error": gather_error},
            # the fused entry's arithmetic intensity (~40 FLOP/B) is above the FP64 ridge, so the
            # nominal bound is FP64 vector throughput; HBM is reported beside it.  The kernel is in fact bound
            # by dependent-chain latency at one wave per SIMD.
            "roofline": {"bound": "fp64_vector", "achieved": ach_tf, "peak": FP64_VEC_PEAK_TF, "unit": "TFLOP/s",
                         "frac": ach_tf / FP64_VEC_PEAK_TF, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "hoqp_solve_wbc_ref_kernel" if args.mode == "reference"
                                   else "hoqp_solve_kernel<false,true,true>", "kernel_ms": kern_ms,
                         "launches_timed": kern_n, "alg_flops_per_solve": flops_mean,
                         "alg_bytes_per_solve": bytes_mean,
                         "hbm": {"achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": ach_gbs / HBM_PEAK_GBS},
                         "note": "FLOPs and bytes: SURVEY.md 8(d) formulas for the fused WBC entry, averaged over the "
                                 "batch's contact patterns; active-set iterations are not counted as FLOPs"},
            "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                           "max": float(lat.max()), "batches": int(lat.size)},
            "pipelined": pipelined,
            "solver": {"status_nonzero": int((status != 0).sum()), "iters_level1_mean": float(iters[:, 1].mean()),
                       "iters_max": int(iters.max()),
                       # active-set steps per instance, summed over the levels: histogram 0,1,2,...
                       "iters_hist": np.bincount(iters.sum(axis=1).astype(np.int64)).tolist()},
        }
        if not args.no_cpu_baseline and world == 1:
            threads = min(16, os.cpu_count() or 1)
            sample = min(args.cpu_sample, nloc)
            res["cpu_baseline"] = cpu_baseline(inp, args.mode, sample, threads, prm_kw, detail=args.cpu_detail)
            res["feasibility_match"] = feasibility_match(out, inp, args.mode, sample, threads, prm_kw)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
