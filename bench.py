#!/usr/bin/env python3
"""bench.py -- fp64 HO-QP solves/s on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (hoqp_solve_wbc_batched: task construction +
5-level HO-QP chain + torque map + validity flags) over one batch of synthetic
ANYmal-C-like states that is already resident in HBM.  N = 1 workload: BASELINE.json
configs[1] (4096 randomised 4-contact stance states, reference semantics).  N > 1: the
batch dimension is sharded, one process per GPU, the same 4096 states per rank (weak
scaling); the only collective is the RCCL gather of per-instance solutions and the
all-reduce of residual statistics to rank 0 (the path's one exchange step).

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
# algorithmic bytes per solve of the dominant kernel (hoqp_solve_kernel reads the generic
# task blocks): SURVEY.md section 8(d), generic task entry, n_c = 4: 8*76*31 in + 664 out
ALG_BYTES_PER_SOLVE = 8 * 76 * 31 + 664
# algorithmic flops per solve, reference mode, zero active-set iterations
ALG_FLOPS_PER_SOLVE = 210_984


# which BASELINE.json configuration a --contact choice restates
WORKLOAD_LABEL = {"stance": "configs[1]", "trot": "configs[2] (trot mix)", "three": "configs[4] (swing tracking)",
                  "crawl": "crawl gait of the reference via the command sampler"}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(inp, mode, sample, threads, prm_kw, detail=False):
    """The oracle ("port") timed on the host cores over a bounded sample."""
    from oracle import oracle as orc

    orc.build()
    sub = {k: v[:sample] for k, v in inp.items()}
    cfg, prm = orc.default_cfg(mode), orc.default_wbc_params(mode, **prm_kw)
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
                      f"oracle/hoqp_oracle.c (-O3, OpenMP {threads} threads); Drake/Eigen reference unbuildable",
               cpu_model=cpu_model())
    if detail:
        # SURVEY.md 8(d) CPU baseline (i): one thread, configs[0] (the nominal standing state),
        # 10^4 repetitions, per-solve latency
        from quadruped_locomotion_amd import anymal_model as am

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="states per GPU per step")
    ap.add_argument("--mode", default="reference", choices=["reference", "paper"])
    ap.add_argument("--contact", default="stance", choices=["stance", "trot", "three", "crawl"],
                    help="contact patterns: i.i.d. masks (stance / trot / three) or, for crawl, the pattern and swing "
                         "commands of the reference's crawl gait at random phases (hoqp_gait_sample_batched)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=512)
    ap.add_argument("--mu", type=float, default=None, help="friction coefficient (default: the reference's)")
    ap.add_argument("--tau-max", type=float, default=None, help="joint torque limit, symmetric (default: the reference's)")
    ap.add_argument("--cpu-detail", action="store_true",
                    help="also time the oracle single-threaded on configs[0] (10^4 repetitions)")
    ap.add_argument("--no-gather", action="store_true", help="skip the collection of results on rank 0")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="gather solutions to rank 0 inside every timed step (default: once, after the timed loop)")
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

    from quadruped_locomotion_amd import anymal_model as am
    from quadruped_locomotion_amd import hoqp

    torch.cuda.set_device(local_rank)
    prm_kw = {}
    if args.mu is not None:
        prm_kw["mu"] = args.mu
    if args.tau_max is not None:
        prm_kw.update(tau_max=args.tau_max, tau_min=-args.tau_max)
    solver = hoqp.HoQpSolver(hoqp.default_config(args.mode, device=local_rank, **prm_kw))
    inp = am.make_wbc_inputs(am.sample_states(args.batch, args.seed + rank,
                                              "stance" if args.contact == "crawl" else args.contact))
    dinp = solver.to_device(inp)
    if args.contact == "crawl":
        # commands from the gait sampler on the device (2-, 3- and 4-contact states)
        rng = np.random.default_rng(args.seed + rank + 77)
        t_gait = torch.as_tensor(rng.uniform(0.0, 10.0, args.batch)).cuda()
        v_cmd = torch.as_tensor(rng.normal(0.0, 0.1, (args.batch, 2))).cuda()
        lc, ct = solver.gait_sample(hoqp.default_gait("crawl", fix_derivatives=1), t_gait, v_cmd,
                                    dinp["footpos"][:, :, :2].contiguous())
        dinp["leg_cmd"], dinp["contact"] = lc, ct
        solver.synchronize()
        inp["leg_cmd"], inp["contact"] = lc.cpu().numpy(), ct.cpu().numpy()
    out = solver.alloc_wbc_outputs(args.batch)
    from quadruped_locomotion_amd import parallel

    def collect():
        # collection of results: per-instance solutions to rank 0 (RCCL
        # gather), residual statistics all-reduced.  The solve itself has no exchange step, so by
        # default this runs once after the timed loop and is reported as `gather_ms`.
        parallel.gather_solutions(dict(x=out["x"], tau=out["tau"]), args.batch * world, dst=0)
        mx, sm = parallel.local_stats(out)
        parallel.reduce_stats(mx, sm)

    def step():
        solver.solve_wbc(dinp, out)
        if world > 1 and args.gather_every_step and not args.no_gather:
            collect()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    solver.timing(True)
    lat = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kern_ms, kern_n = solver.kernel_time_ms()
    solver.timing(False)

    # p50/p99 batch latency (submit -> results visible), measured outside the timed region
    # >= 1000 batches for the percentiles on a default run (fewer on short runs)
    for _ in range(1000 if args.steps >= 200 else args.steps):
        torch.cuda.synchronize()
        a = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - a) * 1e3)
    lat = np.array(lat)

    gather_ms = None
    if world > 1 and not args.no_gather:
        collect()  # warm-up of the communicator
        barrier()
        g0 = time.perf_counter()
        collect()
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64,
                            device="cpu" if dist.get_backend() == "gloo" else solver.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    status = out["status"].cpu().numpy()
    iters = out["iters"].cpu().numpy()
    if rank == 0:
        total = args.batch * world * args.steps
        value = total / elapsed
        kern_s = kern_ms * 1e-3
        achieved_gbs = ALG_BYTES_PER_SOLVE * args.batch / kern_s / 1e9 if kern_s > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from the committed PMC pass of this same
        # command (tools/profile.sh; separate --pmc runs).  Only reported for the profiled
        # workload; FETCH_SIZE is uncalibrated for 8-byte-per-lane loads on gfx950.
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")) as f:
                tj = json.load(f)
            if tj.get("batch") == args.batch and tj.get("semantics") == args.mode and tj.get("contact") == args.contact:
                traffic = float(tj["fetch_bytes_per_launch"] + tj["write_bytes_per_launch"])
                traffic_src = tj.get("source")
        except (OSError, ValueError, KeyError):
            pass
        res = {
            "metric": "fp64 HO-QP solves/sec (ANYmal-C, 4 contacts)",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (own ANYmal-C-like model, seed %d)" % args.seed,
            "config": {"workload": f"{WORKLOAD_LABEL[args.contact]}: {args.batch} randomised {args.contact} states "
                                   f"per GPU, 5-level WBC hierarchy, {args.mode} semantics",
                       "batch_per_gpu": args.batch, "semantics": args.mode, "contact": args.contact,
                       "entry": "hoqp_solve_wbc_batched", "params": prm_kw or "reference defaults",
                       "gather": ("every_step" if args.gather_every_step else "after_loop")
                       if (world > 1 and not args.no_gather) else "none", "gather_ms": gather_ms},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": "hoqp_solve_kernel", "kernel_ms": kern_ms, "launches_timed": kern_n,
                         "alg_bytes_per_solve": ALG_BYTES_PER_SOLVE,
                         "fp64_vector": {"achieved_tflops": ALG_FLOPS_PER_SOLVE * args.batch / kern_s / 1e12
                                         if kern_s > 0 else 0.0, "peak_tflops": FP64_VEC_PEAK_TF,
                                         "alg_flops_per_solve": ALG_FLOPS_PER_SOLVE},
                         "note": "latency/LDS-bound small factorizations; both classical fractions are < 1 %"},
            "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                           "max": float(lat.max()), "batches": int(lat.size)},
            "solver": {"status_nonzero": int((status != 0).sum()), "iters_level1_mean": float(iters[:, 1].mean()),
                       "iters_max": int(iters.max()),
                       # active-set steps per instance, summed over the levels: histogram 0,1,2,...
                       "iters_hist": np.bincount(iters.sum(axis=1).astype(np.int64)).tolist()},
        }
        if not args.no_cpu_baseline and world == 1:
            threads = min(16, os.cpu_count() or 1)
            res["cpu_baseline"] = cpu_baseline(inp, args.mode, min(args.cpu_sample, args.batch), threads, prm_kw,
                                               detail=args.cpu_detail)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
