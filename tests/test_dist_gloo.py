"""Multi-process (world_size 2, gloo, CPU) test of the sharded path of SURVEY §8e: row-shard the
batch, evaluate the local block, all-reduce the CAReduce partials — must equal the unsharded
result.  The per-rank evaluation uses the oracle here (no GPU in this tier); on the GPU box the
same ShardedFunction wraps the HIP executor (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, case_name, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import interp
        from golden_util import CASES, case_inputs, case_plan
        from aesara_amd.dist import ShardedFunction, plan_split_outputs, shard_rows

        c = next(c for c in CASES if c["name"] == case_name)
        plan, ins = case_plan(c), case_inputs(c)
        n = ins[0].shape[0]
        lo, hi = shard_rows(n, world, rank)
        if case_name == "cfg5_logistic":
            local = [ins[0][lo:hi], ins[1], ins[2], ins[3][lo:hi]]
        else:
            local = [ins[0][lo:hi]] + ins[1:]
        kinds = plan_split_outputs(plan, 0)
        fn = ShardedFunction(
            lambda *a: [torch.from_numpy(np.array(o, dtype=np.float64)) for o in interp.run_plan(plan, a)],
            kinds)
        outs = fn(*local)
        if rank == 0:
            q.put((kinds, [o.numpy() for o in outs]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case_name", ["cfg2_gauss_sum", "cfg5_logistic"])
def test_row_sharded_allreduce_matches_unsharded(case_name):
    from golden_util import CASES, case_expected
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case_name, q)) for r in range(2)]
    for p in procs:
        p.start()
    kinds, outs = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    c = next(c for c in CASES if c["name"] == case_name)
    exp = case_expected(c)
    assert all(k == "allreduce" for k in kinds)
    for o, e in zip(outs, exp):
        np.testing.assert_allclose(o, e.astype(np.float64), rtol=1e-4, atol=1e-3)


def test_shard_rows_partition():
    from aesara_amd.dist import shard_rows
    for n in (0, 1, 7, 4096, 16777216):
        for w in (1, 2, 3, 8):
            blocks = [shard_rows(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert max(h - l for l, h in blocks) - min(h - l for l, h in blocks) <= 1
