"""Multi-GPU sharding of the hot path: one process per GPU, RCCL over xGMI.

The reference has no data-parallel execution at all (its only communication code is the
optional mpi4py point-to-point ops of tensor/io.py:108-262), so there is nothing to translate:

* independent evaluations / independent graph outputs shard embarrassingly (replicas);
* a batch (row) axis that has been split across ranks needs exactly one exchange: a ``CAReduce``
  over the split axis becomes local-reduce + ``all_reduce(SUM)`` of the partial — a few bytes
  (config 2: one f64; config 5: 258 values), i.e. latency-bound on xGMI, so it is issued
  asynchronously and consecutive evals pipeline;
* a ``Gemv`` on ``X.T`` (reduction over the split axis) is local GEMV + all-reduce of the D-vector.

``torch.distributed`` (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests) is the transport.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_rows(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block [lo, hi) of rank ``rank``: the first ``n_rows % world`` ranks get
    one extra row, every row is owned by exactly one rank."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def plan_split_outputs(plan, split_input: int) -> List[str]:
    """For a plan whose input ``split_input`` is row-sharded, classify each output:
    ``"allreduce"`` (a full CAReduce{add} / Gemv over the split axis: sum the per-rank results)
    or ``"local"`` (row-wise result: stays sharded).  Conservative: anything else raises."""
    producers = {}
    for n in plan.nodes:
        for o in n.outputs:
            producers[o] = n
    kinds = []
    for o in plan.outputs:
        n = producers.get(o)
        if n is not None and n.op == "CAReduce" and n.params["scalar_op"] == "add" and \
                (n.params["axis"] is None or 0 in n.params["axis"]):
            kinds.append("allreduce")
        elif n is not None and n.op == "Gemv":
            a = producers.get(n.inputs[2])
            if a is not None and a.op == "DimShuffle" and a.params["new_order"] == [1, 0]:
                kinds.append("allreduce")  # X.T @ r: contraction over the split axis
            else:
                kinds.append("local")
        else:
            kinds.append("local")
    return kinds


class ShardedFunction:
    """Evaluate a plan on this rank's row block and combine outputs across ranks.

    ``executor`` maps the local inputs to local outputs (the HIP executor on a GPU; any callable
    in tests).  Outputs tagged "allreduce" are summed over the process group; the handles of the
    asynchronous collectives are returned so that callers can pipeline evaluations."""

    def __init__(self, executor, kinds: Sequence[str], group=None):
        self.executor = executor
        self.kinds = list(kinds)
        self.group = group

    def __call__(self, *local_inputs, async_op=False):
        import torch.distributed as dist

        outs = list(self.executor(*local_inputs))
        handles = []
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            for o, k in zip(outs, self.kinds):
                if k == "allreduce":
                    h = dist.all_reduce(o, op=dist.ReduceOp.SUM, group=self.group,
                                        async_op=async_op)
                    if async_op:
                        handles.append(h)
        return (outs, handles) if async_op else outs
