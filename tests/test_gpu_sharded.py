"""GPU test of the one-process multi-GPU driver (hoqp_sharded_*, include/hoqp.h): contiguous shards, one worker
thread + stream + handle per shard, device-to-device collection into the root's arrays, statistics reduced per
device and combined on the host.  The box has one GPU, so every shard is pointed at device 0 (the driver does
not care which device a shard names); results must be bit-identical to one hoqp_solve_wbc_batched call over the
whole batch, for ragged shards and an empty one."""
import numpy as np
import pytest

from oracle import anymal_model as am

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sizes", [(1000, 1000, 700), (513, 0, 1), (4096,)])
def test_sharded_driver_matches_single_call(solver_factory, sizes):
    import torch
    from quadruped_locomotion_amd import hoqp

    total = sum(sizes)
    inp = am.make_wbc_inputs(am.sample_states(total, 61, "trot"))
    s = solver_factory("reference")
    dall = s.to_device(inp)
    ref = {k: v.cpu().numpy() for k, v in s.solve_wbc(dall).items()}
    s.synchronize()
    drv = hoqp.ShardedSolver(hoqp.default_config("reference"), [0] * len(sizes))
    try:
        shards_in, shards_out, lo = [], [], 0
        for n in sizes:
            shards_in.append({k: v[lo:lo + n].contiguous() for k, v in dall.items()} if n else None)
            shards_out.append(s.alloc_wbc_outputs(n))
            lo += n
        kw = dict(dtype=torch.float64, device="cuda:0")
        gather = dict(device=0, x=torch.full((total, 30), float("nan"), **kw), tau=torch.full((total, 12), float("nan"), **kw),
                      v=torch.full((total, 40), float("nan"), **kw),
                      status=torch.full((total,), -1, dtype=torch.int32, device="cuda:0"))
        for rep in range(2):      # the second call reuses workers, streams and accumulators
            st = drv.solve_wbc(shards_in, shards_out, gather)
            for key in ("x", "tau", "v", "status"):
                assert np.array_equal(gather[key].cpu().numpy(), ref[key]), (key, rep)
            assert st.instances == total and st.status_nonzero == int((ref["status"] != 0).sum())
            assert st.iters_total == int(ref["iters"].sum())
            assert st.eq_res_max == ref["eq_res_max"].max() and st.ineq_viol_max == ref["ineq_viol_max"].max()
        # per-shard outputs are the slices of the single call
        lo = 0
        for n, o in zip(sizes, shards_out):
            assert np.array_equal(o["x"].cpu().numpy(), ref["x"][lo:lo + n])
            assert np.array_equal(o["flags"].cpu().numpy(), ref["flags"][lo:lo + n])
            lo += n
        # no collection, no statistics
        assert drv.solve_wbc(shards_in, shards_out, None, want_stats=False) is None
    finally:
        drv.close()


def test_sharded_driver_rejects_bad_arguments():
    from quadruped_locomotion_amd import hoqp
    import ctypes as C

    lib = hoqp.load_library()
    h = C.c_void_p()
    cfg = hoqp.default_config("reference")
    assert lib.hoqp_sharded_create(C.byref(cfg), 0, (C.c_int * 1)(0), C.byref(h)) == -1
    assert lib.hoqp_sharded_create(C.byref(cfg), hoqp.MAX_SHARDS + 1, (C.c_int * 17)(*([0] * 17)), C.byref(h)) == -1
    assert lib.hoqp_sharded_create(C.byref(cfg), 1, (C.c_int * 1)(99), C.byref(h)) != 0      # no such device
    assert lib.hoqp_solve_wbc_sharded(None, None, None, None, None) == -1
