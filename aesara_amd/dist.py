"""Multi-GPU execution of a lowered plan: one process per GPU, RCCL over xGMI.

The reference has no data-parallel execution at all (its only communication code is the
optional mpi4py point-to-point ops of tensor/io.py:108-262), so there is nothing to translate.
Two graph-level transformations, both decided on the *plan* (never on run-time values):

1. **Batch-axis split** (:func:`shard_plan`, :class:`ShardedPlan`).  Some plan inputs are declared
   split along an axis (``X`` and ``y`` of BASELINE config 5 along the rows).  A sharding state is
   propagated through every node:

   * ``rep``      — the same value on every rank;
   * ``split(a)`` — the rank holds its block of the value along axis ``a``;
   * ``partial(op)`` — the true value is ``op`` over the ranks' values (``add`` / ``maximum`` /
     ``minimum``): what a ``CAReduce`` over the split axis, or a contraction over it
     (``Gemv(X.T, r)``, ``Dot22(A.T, B)``), leaves on each rank;
   * ``extent``   — ``Shape_i`` of a split value along its split axis (local length; the global
     length is the sum).

   Basic indexing that leaves the split axis whole keeps the split (``Subtensor`` /
   ``IncSubtensor``); a ``Scan`` whose sequences and initial states are split along their batch
   axis and whose step is row-local (same analysis on the inner plan) runs on every rank's rows
   with no exchange (:func:`_shard_scan`).

   A ``partial`` (or an ``extent`` read by anything but an allocation) has to be combined before
   it is read: the plan is cut into *rounds*; round k runs locally, then ONE packed
   ``all_reduce`` per (reduction op, exchange dtype) sums / maxes all of round k's partials, and
   round k+1 continues on the combined values.  Floating-point sums are exchanged in the
   **accumulator dtype** (``CAReduce.acc_dtype``; float64 for the float32 contraction partials)
   and cast to the node's output dtype after the collective (SURVEY §8e "the Sum's f64->f32
   cast happens after the reduce").  Anything that cannot be proven row-local raises
   :class:`ShardingError` — never a silently wrong per-rank partial.

2. **Independent-output placement** (:func:`place_outputs`, :class:`PlacedPlan`).  Outputs whose
   ancestor sets share no computed value are assigned to different ranks (largest-first onto the
   least-loaded rank); each rank runs only its own sub-plan, no communication at all.

``torch.distributed`` (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests) is the transport;
a :class:`LocalGroup` runs k logical shards in one process (single-GPU boxes, tests).
"""
from __future__ import annotations

import copy
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import knobs
from .plan import Node, Plan


class ShardingError(ValueError):
    """The plan cannot be evaluated on row blocks with the declared split."""


def shard_rows(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block [lo, hi) of rank ``rank``: the first ``n_rows % world`` ranks get
    one extra row, every row is owned by exactly one rank."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


# ---------------------------------------------------------------------------------------------
# sharding-state propagation
# ---------------------------------------------------------------------------------------------
REP = ("rep",)
_RED_OPS = {"add": "add", "maximum": "maximum", "minimum": "minimum"}
_VIEW_REP_ONLY = {"AdvancedSubtensor1", "AdvancedIncSubtensor1",
                  "AdvancedSubtensor", "AdvancedIncSubtensor", "Reshape", "Join", "Split",
                  "CumOp", "Argmax", "MaxAndArgmax", "Sort", "ArgSort", "IfElse", "BatchedDot",
                  "MatMul", "Ger"}


def _const_scalar(plan: Plan, vid: int):
    v = plan.vars[vid]
    if v.const is not None and len(v.const.get("data", ())) == 1:
        return float(v.const["data"][0])
    return None


def _ident_scalar(n_in=1):
    return {"n_in": n_in, "nodes": [], "out": [["i", 0]]}


def _cast_scalar(dtype):
    return {"n_in": 1, "nodes": [{"op": "cast", "in": [["i", 0]], "dtype": dtype}], "out": [["t", 0]]}


class ShardedPlanSpec:
    """Result of :func:`shard_plan`: ``rounds`` = list of ``(plan, exchanges)``; the plan of round
    k takes the original inputs followed by every earlier round's carried values; ``exchanges``
    = ``[(output position in the round plan, op, exchange dtype)]`` — those outputs are combined
    across the ranks before the next round reads them.  ``out_state[i]`` is the sharding state of
    original output i (``rep`` after the final combine, or ``split(a)``: stays sharded)."""

    def __init__(self, rounds, carried, out_src, out_state, state):
        self.rounds, self.carried, self.out_src = rounds, carried, out_src
        self.out_state, self.state = out_state, state

    @property
    def n_exchange_rounds(self):
        return sum(1 for _, ex in self.rounds if ex)


def shard_plan(plan: Plan, split_inputs: Dict[int, int]) -> ShardedPlanSpec:
    """Propagate sharding states (module docstring) and cut ``plan`` into rounds.

    ``split_inputs``: ``{input position: axis}`` — which plan inputs arrive as per-rank blocks."""
    plan = copy.deepcopy(plan)
    state: Dict[int, tuple] = {}
    rnd: Dict[int, int] = {}            # round in which a var's *usable* (combined) value exists
    for vid, v in plan.vars.items():
        if v.const is not None:
            state[vid], rnd[vid] = REP, 0
    for pos, vid in enumerate(plan.inputs):
        if pos in split_inputs:
            ax = split_inputs[pos]
            if not (0 <= ax < plan.vars[vid].ndim):
                raise ShardingError(f"input {pos}: split axis {ax} out of range")
            state[vid] = ("split", ax)
        else:
            state[vid] = REP
        rnd[vid] = 0

    # var -> (raw var holding the per-rank partial, op, exchange dtype); the var itself names the
    # combined value (available one round later)
    pending: Dict[int, tuple] = {}
    node_round: List[int] = []
    extra_nodes: Dict[int, List[Node]] = {}     # node index -> nodes appended after it (casts)
    post_nodes: Dict[int, List[Node]] = {}      # var -> nodes that finish it after the combine

    def use(vid, as_alloc_dim=False):
        """State under which a consumer sees ``vid`` and the round from which it may read it."""
        st = state[vid]
        if st[0] == "partial":
            return REP, rnd[vid] + 1
        if st[0] == "extent":
            if as_alloc_dim:
                return st, rnd[vid]
            return REP, rnd[vid] + 1      # read through its global twin (see global_extent)
        return st, rnd[vid]

    producer_idx: Dict[int, int] = {}
    extent_twin: Dict[int, int] = {}

    def global_extent(vid):
        """Non-allocation readers of a local extent get the SUM over the ranks: a twin variable
        that is exchanged like any other partial (the local value stays for allocations)."""
        g = extent_twin.get(vid)
        if g is None:
            raw = plan.new_var("int64", [], name=f"v{vid}_local_extent")
            g = plan.new_var("int64", [], name=f"v{vid}_global_extent")
            extra_nodes.setdefault(producer_idx[vid], []).append(
                Node("Elemwise", [vid], [raw], {"scalar": _cast_scalar("int64")}))
            post_nodes[g] = [Node("Elemwise", [raw], [g], {"scalar": _cast_scalar("int64")})]
            pending[g] = (raw, "add", "int64")
            state[g], rnd[g] = ("partial", "add"), rnd[vid]
            state[raw], rnd[raw] = ("partial", "add"), rnd[vid]
            extent_twin[vid] = g
        return g

    def make_partial(node_idx, out, op, acc_dtype):
        """``out`` of node ``node_idx`` is a per-rank partial: exchange it in ``acc_dtype``."""
        v = plan.vars[out]
        xdt = acc_dtype if v.dtype.startswith("float") else v.dtype
        if op == "add" and v.dtype == "float32":
            xdt = "float64" if acc_dtype in (None, "float32", "float64") else acc_dtype
        if op != "add":
            xdt = v.dtype
        raw = out
        if xdt != v.dtype:
            raw = plan.new_var(xdt, list(v.shape), name=(v.name or f"v{out}") + "_partial")
            n = plan.nodes[node_idx]
            if n.op == "CAReduce":
                # the reduction itself delivers the accumulator dtype: no intermediate rounding
                n.outputs[n.outputs.index(out)] = raw
            else:
                extra_nodes.setdefault(node_idx, []).append(
                    Node("Elemwise", [out], [raw], {"scalar": _cast_scalar(xdt)}))
            post_nodes[out] = [Node("Elemwise", [raw], [out], {"scalar": _cast_scalar(v.dtype)})]
        pending[out] = (raw, op, xdt)
        state[out] = ("partial", op)
        state[raw], rnd[raw] = ("partial", op), rnd[out]

    for ni, n in enumerate(plan.nodes):
        op = n.op
        for o in n.outputs:
            producer_idx[o] = ni
        if op in ("AllocEmpty", "Alloc"):
            first_dim = 0 if op == "AllocEmpty" else 1
            sts = [use(i, as_alloc_dim=(k >= first_dim)) for k, i in enumerate(n.inputs)]
        else:
            n.inputs = [global_extent(i) if state[i][0] == "extent" else i for i in n.inputs]
            sts = [use(i) for i in n.inputs]
        r = max([q for _, q in sts], default=0)
        ins = [s for s, _ in sts]
        node_round.append(r)
        for o in n.outputs:
            rnd[o] = r

        def all_rep():
            return all(s == REP for s in ins)

        if all_rep() and op != "Shape_i":
            for o in n.outputs:
                state[o] = REP
            continue

        if op == "Elemwise":
            axes = {s[1] for s in ins if s[0] == "split"}
            if len(axes) != 1:
                raise ShardingError(f"Elemwise mixes split axes {sorted(axes)}")
            ax = axes.pop()
            for vid, s in zip(n.inputs, ins):
                v = plan.vars[vid]
                if s == REP and v.ndim and (ax >= v.ndim or v.shape[ax] != 1):
                    raise ShardingError(
                        f"Elemwise: replicated operand v{vid} spans the split axis {ax} "
                        "(declare it in split_inputs)")
            for o in n.outputs:
                state[o] = ("split", ax)
        elif op == "DimShuffle":
            (s,) = ins
            order = n.params["new_order"]
            if s[1] not in order:
                raise ShardingError("DimShuffle drops the split axis")
            state[n.outputs[0]] = ("split", order.index(s[1]))
        elif op in ("ViewOp", "DeepCopyOp", "SpecifyShape", "Assert", "Unbroadcast"):
            if any(s != REP for s in ins[1:]):
                raise ShardingError(f"{op}: only the first operand may be split")
            state[n.outputs[0]] = ins[0]
        elif op == "Shape_i":
            (s,) = ins
            if s[0] == "split" and s[1] == n.params["i"]:
                state[n.outputs[0]] = ("extent",)
            else:
                state[n.outputs[0]] = REP
        elif op in ("AllocEmpty", "Alloc"):
            dims = ins[first_dim:]
            ext = [k for k, s in enumerate(dims) if s[0] == "extent"]
            if any(s != REP for s in ins[:first_dim]) or len(ext) != 1 or \
                    any(s[0] not in ("rep", "extent") for s in dims):
                raise ShardingError(f"{op}: needs a replicated value and exactly one split extent")
            state[n.outputs[0]] = ("split", ext[0])
        elif op == "CAReduce":
            (s,) = ins
            axis = n.params["axis"]
            nd = plan.vars[n.inputs[0]].ndim
            axes = list(range(nd)) if axis is None else [a % nd for a in axis]
            if s[1] in axes:
                rop = _RED_OPS.get(n.params["scalar_op"])
                if rop is None:
                    raise ShardingError(
                        f"CAReduce{{{n.params['scalar_op']}}} over the split axis is not combinable")
                make_partial(ni, n.outputs[0], rop, n.params.get("acc_dtype"))
            else:
                state[n.outputs[0]] = ("split", s[1] - sum(1 for a in axes if a < s[1]))
        elif op == "Gemv":
            sy, sal, sA, sx, sbe = ins
            beta = _const_scalar(plan, n.inputs[4])
            if sal != REP or sbe != REP:
                raise ShardingError("Gemv: alpha / beta must be replicated")
            if sA == ("split", 0) and sx == REP and (sy == ("split", 0) or (beta == 0.0 and sy[0] in ("rep", "split"))):
                state[n.outputs[0]] = ("split", 0)                 # row-wise
            elif sA == ("split", 1) and sx == ("split", 0) and beta == 0.0:
                make_partial(ni, n.outputs[0], "add", "float64")   # contraction over the split axis
            else:
                raise ShardingError(f"Gemv with operand states y={sy} A={sA} x={sx} beta={beta}")
        elif op in ("Dot22", "Dot", "Dot22Scalar", "Gemm"):
            if op == "Gemm":
                sC, sal, sA, sB, sbe = ins
                beta = _const_scalar(plan, n.inputs[4])
                if sal != REP or sbe != REP:
                    raise ShardingError("Gemm: alpha / beta must be replicated")
            else:
                sA, sB = ins[0], ins[1]
                sC, beta = None, 0.0
                if any(s != REP for s in ins[2:]):
                    raise ShardingError(f"{op}: scalar must be replicated")
            ndA, ndB = plan.vars[n.inputs[2 if op == "Gemm" else 0]].ndim, \
                plan.vars[n.inputs[3 if op == "Gemm" else 1]].ndim
            kA, kB = ndA - 1, 0
            c_ok = sC is None or beta == 0.0
            if sA[0] == "split" and sA[1] != kA and sB == REP and (c_ok or sC == ("split", 0)):
                state[n.outputs[0]] = ("split", 0)
            elif sA == REP and sB[0] == "split" and sB[1] != kB and ndB == 2 and \
                    (c_ok or sC == ("split", ndA - 1)):
                state[n.outputs[0]] = ("split", ndA - 1)
            elif sA == ("split", kA) and sB == ("split", kB) and c_ok:
                make_partial(ni, n.outputs[0], "add", "float64")
            else:
                raise ShardingError(f"{op} with operand states A={sA} B={sB} C={sC} beta={beta}")
        elif op in ("ScalarFromTensor", "TensorFromScalar"):
            state[n.outputs[0]] = ins[0]
        elif op in ("Subtensor", "IncSubtensor"):
            # basic indexing that leaves the split axis whole is row-local
            idx = n.params["idx_list"]
            n_idx_in = 2 if op == "IncSubtensor" else 1
            sx = ins[0]
            if any(s != REP for s in ins[n_idx_in:]):
                raise ShardingError(f"{op}: index operands must be replicated")
            if sx[0] != "split":
                raise ShardingError(f"{op}: a split value written into / read through a replicated one")
            a = sx[1]
            if a < len(idx) and idx[a].get("slice") != [None, None, None]:
                raise ShardingError(f"{op}: the index touches the split axis {a}")
            sub_axis = a - sum(1 for e in idx[:a] if "index" in e)
            if op == "Subtensor":
                state[n.outputs[0]] = ("split", sub_axis)
            else:
                sy, yv = ins[1], plan.vars[n.inputs[1]]
                sub_nd = plan.vars[n.inputs[0]].ndim - sum(1 for e in idx if "index" in e)
                ya = sub_axis - (sub_nd - yv.ndim)          # right-aligned broadcasting of y
                if sy == REP:
                    if ya >= 0 and yv.shape[ya] != 1:
                        raise ShardingError("IncSubtensor: the replicated value spans the split axis")
                elif sy != ("split", ya):
                    raise ShardingError(f"IncSubtensor: value split on {sy}, target rows on axis {ya}")
                state[n.outputs[0]] = sx
        elif op == "Scan":
            _shard_scan(plan, n, ins, state)
        else:
            raise ShardingError(f"{op}: not provably row-local for a split operand "
                                f"(operand states {ins})")

    # ---- outputs: a partial output needs one more (final) round for its combine + cast --------
    out_state, last = [], 0
    for o in plan.outputs:
        st = state[o]
        if st[0] == "extent":
            raise ShardingError("an output is the local extent of a split axis")
        if st[0] == "partial":
            out_state.append(REP)
            last = max(last, rnd[o] + 1)
        else:
            out_state.append(st)
            last = max(last, rnd[o])
    n_rounds = max(last, max(node_round, default=0)) + 1

    # ---- build one plan per round ----------------------------------------------------------
    producer_round = {}
    for ni, n in enumerate(plan.nodes):
        for o in n.outputs:
            producer_round[o] = node_round[ni]
        for en in extra_nodes.get(ni, []):
            for o in en.outputs:
                producer_round[o] = node_round[ni]
    # combined value of a pending var becomes available in round rnd+1 (as input, raw var slot)
    rounds, carried = [], []          # carried: vars made available to later rounds, in order
    avail: Dict[int, int] = {}        # var -> round it became an input of later rounds
    for k in range(n_rounds):
        nodes_k: List[Node] = []
        # finish values combined after round k-1
        for v, (raw, _op, _xdt) in pending.items():
            if rnd[v] + 1 == k:
                nodes_k.extend(post_nodes.get(v, []))
        for ni, n in enumerate(plan.nodes):
            if node_round[ni] == k:
                nodes_k.append(n)
                nodes_k.extend(extra_nodes.get(ni, []))
        produced = {o for n in nodes_k for o in n.outputs}
        needed_later = set()
        for k2 in range(k + 1, n_rounds):
            for ni, n in enumerate(plan.nodes):
                if node_round[ni] == k2:
                    needed_later.update(n.inputs)
        exchanges, outs_k = [], []
        for v, (raw, rop, xdt) in pending.items():
            if rnd[v] == k:
                exchanges.append((len(outs_k), rop, xdt))
                outs_k.append(raw)
        n_ex = len(outs_k)
        for v in sorted(produced):
            is_raw = any(raw == v for raw, _, _ in pending.values())
            if v in needed_later and not is_raw and state.get(v, REP)[0] != "partial":
                outs_k.append(v)
        for o in plan.outputs:
            fin = rnd[o] + (1 if state[o][0] == "partial" else 0)
            if fin == k and o not in outs_k[n_ex:]:
                outs_k.append(o)
        p_k = Plan(f"{plan.name}_round{k}", plan.vars, list(plan.inputs) + list(carried),
                   list(outs_k), nodes_k)
        rounds.append((p_k, exchanges))
        for v in outs_k:
            if v not in avail and v not in plan.inputs:
                avail[v] = k
                carried.append(v)
    out_src = []
    for o in plan.outputs:
        fin = rnd[o] + (1 if state[o][0] == "partial" else 0)
        out_src.append((fin, rounds[fin][0].outputs.index(o)))
    return ShardedPlanSpec(rounds, carried, out_src, out_state, state)


def _shard_scan(plan: Plan, n: Node, ins, state):
    """A Scan whose sequences / initial states are split along their BATCH axis (axis 1 of the
    ``[T, B, ...]`` arrays): the recurrences of different batch rows are independent when the step
    itself is row-local — checked by running the same analysis on the inner plan with the per-step
    values split on axis 0 — so every rank runs the loop on its rows, no exchange.  Outputs keep the
    batch split (``[T, B, ...]`` on axis 1).  Covers forward Scans and the gradient Scans of
    ``aesara.grad`` (mit-mot groups: every tap of a split buffer is a split per-step value)."""
    p = n.params
    if p.get("as_while") or p.get("n_shared_outs", 0):
        raise ShardingError("Scan: do-while / shared outputs are not sharded")
    n_seqs, n_nit = p["n_seqs"], p["n_nit_sot"]
    mm_in = [list(t) for t in p.get("mit_mot_in_slices", [])]
    mm_out = [list(t) for t in p.get("mit_mot_out_slices", [])]
    taps = mm_in + [list(t) for t in p["mit_sot_in_slices"]] + [list(t) for t in p["sit_sot_in_slices"]]
    n_mm, n_rec = len(mm_in), len(taps)
    if ins[0] != REP or any(s != REP for s in ins[1 + n_seqs + n_rec:]):
        raise ShardingError("Scan: the step count and the non-sequences must be replicated")
    inner = p["inner"]
    split_in, k = {}, 0
    for j in range(n_seqs):
        s = ins[1 + j]
        if s != REP:
            if s[0] != "split" or s[1] < 1:
                raise ShardingError(f"Scan: sequence {j} is split on {s} (the time axis cannot be split)")
            split_in[k] = s[1] - 1
        k += 1
    for r in range(n_rec):
        s = ins[1 + n_seqs + r]
        if s != REP and (s[0] != "split" or s[1] < 1):
            raise ShardingError(f"Scan: initial state {r} is split on {s} (the time axis cannot be split)")
        for _t in taps[r]:
            if s != REP:
                split_in[k] = s[1] - 1
            k += 1
    spec = shard_plan(inner, split_in)            # raises when the step is not row-local
    if spec.n_exchange_rounds:
        raise ShardingError("Scan: the step combines values across the batch rows")
    o = 0
    for r in range(n_rec):
        s_in = ins[1 + n_seqs + r]
        for _ in range(len(mm_out[r]) if r < n_mm else 1):
            s_out = spec.out_state[o]
            o += 1
            if (s_in == REP) != (s_out == REP) or (s_out != REP and s_out != ("split", s_in[1] - 1)):
                raise ShardingError(f"Scan: state {r} enters as {s_in} and leaves the step as {s_out}")
        state[n.outputs[r]] = s_in
    for j in range(n_nit):
        s_out = spec.out_state[o + j]
        if s_out == REP:
            state[n.outputs[n_rec + j]] = REP
        elif s_out[0] == "split":
            state[n.outputs[n_rec + j]] = ("split", s_out[1] + 1)
        else:
            raise ShardingError(f"Scan: per-step output {j} leaves the step as {s_out}")


def plan_split_outputs(plan: Plan, split_input: int, axis: int = 0) -> List[str]:
    """Classify each output of ``plan`` when input ``split_input`` (and every other input the
    analysis needs split along the same rows — vectors of matching length are NOT guessed) is
    row-sharded: ``"allreduce"`` (per-rank partials are summed) or ``"local"`` (stays sharded).
    Raises :class:`ShardingError` for anything not provably one of the two."""
    spec = None
    err = None
    # the row vectors that go with a row-split matrix (y of config 5) must be split too: try the
    # declared input alone first, then together with the 1-d inputs
    one_d = {p: 0 for p, vid in enumerate(plan.inputs)
             if p != split_input and plan.vars[vid].ndim == 1 and plan.vars[vid].shape != [1]}
    cands = [{split_input: axis}] + [{split_input: axis, p: 0} for p in one_d]
    if len(one_d) > 1:
        cands.append({split_input: axis, **one_d})
    for sp in cands:
        try:
            spec = shard_plan(plan, sp)
            break
        except ShardingError as e:
            err = e
    if spec is None:
        raise err
    return ["local" if st[0] == "split" else
            ("allreduce" if spec.state[o][0] == "partial" else "replicated")
            for o, st in zip(plan.outputs, spec.out_state)]


# ---------------------------------------------------------------------------------------------
# process groups
# ---------------------------------------------------------------------------------------------
class LocalGroup:
    """k logical ranks inside ONE process (a single-GPU box, unit tests): ``all_reduce`` is
    handed the list of every rank's buffer and combines them in place, in rank order."""

    def __init__(self, world: int):
        self.world = world

    def all_reduce_many(self, bufs: Sequence, op: str):
        import torch
        acc = bufs[0].clone()
        for b in bufs[1:]:
            if op == "add":
                acc += b
            elif op == "maximum":
                torch.maximum(acc, b, out=acc)
            else:
                torch.minimum(acc, b, out=acc)
        for b in bufs:
            b.copy_(acc)


class _Done:
    """Handle of a collective that is ordered by the launch stream (nothing to wait for)."""

    def wait(self):
        return True


class HipComm:
    """RCCL communicator owned by the C-ABI (``ahip_comm_*``, include/aesara_hip.h): the
    all-reduce of an exchange round is enqueued by the shim on the LAUNCH stream — between the
    kernels of the rounds around it, no second stream, no event hand-shake, and (while a launch
    list is being recorded) as a list entry that ``ahip_list_run`` replays.  One process per GPU.

    Bootstrap: rank 0 draws the 128-byte id and every other rank receives it through
    ``torch.distributed`` (any initialised backend; gloo is enough) — or pass ``unique_id``
    yourself.  ``world == 1`` needs no bootstrap (and still runs RCCL: the single-GPU test of the
    path)."""

    def __init__(self, world=None, rank=None, unique_id=None, bootstrap_group=None):
        import ctypes as C
        import os
        import torch
        import torch.distributed as dist
        from ._lib import COMM_ID_BYTES, check, lib
        if world is None:
            world = dist.get_world_size(bootstrap_group) if dist.is_initialized() else 1
            rank = dist.get_rank(bootstrap_group) if dist.is_initialized() else 0
        self.world, self.rank = int(world), int(rank or 0)
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(bundled) and not knobs.get("RCCL"):
            check(lib.ahip_comm_set_library(bundled.encode()))     # the copy torch itself uses
        if unique_id is None:
            buf = C.create_string_buffer(COMM_ID_BYTES)
            if self.rank == 0:
                check(lib.ahip_comm_unique_id(buf, COMM_ID_BYTES))
            if self.world > 1:
                box = [buf.raw if self.rank == 0 else None]
                dist.broadcast_object_list(box, src=0, group=bootstrap_group)
                buf = C.create_string_buffer(box[0], COMM_ID_BYTES)
            unique_id = buf.raw
        self._h = C.c_void_p()
        check(lib.ahip_comm_init_rank(C.create_string_buffer(unique_id, COMM_ID_BYTES), self.world,
                                      self.rank, C.byref(self._h)))

    def all_reduce(self, buf, op="add"):
        """In-place all-reduce of a contiguous device tensor on the current (launch) stream."""
        import ctypes as C
        import torch
        from ._lib import RED_OPS, check, lib
        from .device import dtype_code
        if not buf.is_contiguous():
            raise ValueError("HipComm.all_reduce needs a contiguous buffer")
        name = str(buf.dtype).replace("torch.", "")
        ptr = C.c_void_p(buf.data_ptr())
        check(lib.ahip_allreduce(self._h, dtype_code(name), RED_OPS[op], ptr, ptr, buf.numel(),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return _Done()

    def close(self, force=False):
        """Destroy the communicator.  Launch lists that recorded an all-reduce hold its raw
        handle (``REC_ALLREDUCE``): while any is alive (``recorded`` > 0: a ``ShardedPlan`` that
        replays single lists) the communicator is kept unless ``force``."""
        from ._lib import lib
        if self._h and (force or not self.recorded):
            lib.ahip_comm_destroy(self._h)
            self._h = None

    def abort(self):
        """``ncclCommAbort``: give the communicator up even if a collective is stuck on the device
        (a watchdog's way out; ``close`` would wait for it).  Launch lists that recorded this
        communicator must not be replayed afterwards."""
        from ._lib import lib
        if self._h:
            lib.ahip_comm_abort(self._h)
            self._h = None

    recorded = 0        # launch lists holding this communicator's handle

    def __del__(self):
        try:
            self.close()
        except Exception:           # noqa: BLE001  (interpreter shutdown)
            pass


def _dist_reduce(buf, op, group, async_op=False):
    if isinstance(group, HipComm):
        return group.all_reduce(buf, op)        # stream-ordered: sync and async are the same
    import torch.distributed as dist
    rop = {"add": dist.ReduceOp.SUM, "maximum": dist.ReduceOp.MAX, "minimum": dist.ReduceOp.MIN}[op]
    return dist.all_reduce(buf, op=rop, group=group, async_op=async_op)


_TORCH = None


def _torch_dtype(name):
    import torch
    return {"float64": torch.float64, "float32": torch.float32, "int64": torch.int64,
            "int32": torch.int32, "int16": torch.int16, "int8": torch.int8, "uint8": torch.uint8,
            "bool": torch.bool}[name]


class ShardedPlan:
    """Evaluate a plan on this rank's blocks of the split inputs (module docstring, 1.).

    ``executor_factory(plan, use_graph)`` builds the per-round evaluator (default: the HIP
    :class:`~aesara_amd.executor.PlanExecutor`); tests on CPU pass an oracle-backed factory.
    Per evaluation: round 0 replay, then per exchange round ONE ``all_reduce`` per (op, dtype)
    class over a persistent packed buffer that the producing kernels write into directly, then
    the next round's replay reading views of that buffer."""

    def __init__(self, plan: Plan, split_inputs: Dict[int, int], group=None, use_graph=False,
                 executor_factory: Optional[Callable] = None, device=None, borrow=False,
                 force_collectives=False):
        # force_collectives: issue the all-reduces also in a world of ONE rank (an identity that
        # still goes through RCCL: how a single-GPU box exercises the collective path)
        self.force_collectives = force_collectives
        self.spec = shard_plan(plan, split_inputs)
        self.borrow = borrow or not use_graph
        self.group = group
        self.plan = plan
        if executor_factory is None:
            from .executor import PlanExecutor

            def executor_factory(p, use_graph=use_graph):
                # rounds hand their results to the next round / the packed exchange buffer by
                # pointer: borrowed internally, the final outputs are copied out below
                return PlanExecutor(p, use_graph=use_graph, device=device, borrow=True)
        self.execs = [executor_factory(p) for p, _ in self.spec.rounds]
        self._packs: Dict[tuple, list] = {}
        # ONE launch list per sharded evaluation (use_graph + the C-ABI communicator): round 0's
        # kernels, the packed all-reduce (a REC_ALLREDUCE entry), round 1's kernels, ... recorded
        # once per input signature and replayed with one host call — no Python between the rounds
        self.single_list = bool(use_graph) and isinstance(group, HipComm)
        self._lists: Dict[tuple, tuple] = {}
        self._no_list = set()       # signatures that cannot be served from one list
        self.replays = 0            # evaluations served by a single-list replay (tests, bench)

    # -- world size of the group this instance communicates over --------------------------------
    def _world(self):
        if isinstance(self.group, (LocalGroup, HipComm)):
            return self.group.world
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def _pack_buffers(self, k, shapes, like):
        """Persistent packed exchange buffers of round k: one per (op, dtype) class, plus the
        per-output views into them (so pointers stay stable across replays)."""
        import torch
        if True:
            _, exchanges = self.spec.rounds[k]
            classes: Dict[tuple, list] = {}
            for (pos, rop, xdt), shp in zip(exchanges, shapes):
                classes.setdefault((rop, xdt), []).append((pos, shp))
            bufs, views = {}, {}
            for cls, items in classes.items():
                total = sum(int(np.prod(s)) if len(s) else 1 for _, s in items)
                dev = like.device if hasattr(like, "device") else None
                buf = torch.zeros(total, dtype=_torch_dtype(cls[1]), device=dev)
                off = 0
                for pos, shp in items:
                    n = int(np.prod(shp)) if len(shp) else 1
                    views[pos] = buf[off:off + n].view(tuple(shp))
                    off += n
                bufs[cls] = buf
        return bufs, views

    def __call__(self, *local_inputs, async_op=False):
        """Evaluate on this rank's blocks.  ``async_op``: the LAST round's collectives are
        issued asynchronously and their handles returned next to the outputs (which are only
        valid after ``wait()``), so that consecutive evaluations pipeline."""
        import torch
        if isinstance(self.group, LocalGroup):
            raise RuntimeError("logical shards of one process run through run_local_shards()")
        spec = self.spec
        n_orig = len(self.plan.inputs)
        carried_vals: Dict[int, object] = {}
        results: List[list] = []
        handles = []
        world = self._world()
        sig = tuple(tuple(getattr(x, "shape", ())) for x in local_inputs)
        if self.single_list and not async_op:
            done = self._call_single_list(local_inputs, sig, world)
            if done is not None:
                return done
        for k, ((p_k, exchanges), ex) in enumerate(zip(spec.rounds, self.execs)):
            ins = list(local_inputs) + [carried_vals[v] for v in p_k.inputs[n_orig:]]
            if not p_k.nodes:        # nothing left to compute: combined values pass through
                outs = [ins[p_k.inputs.index(v)] for v in p_k.outputs]
            elif not exchanges:
                outs = list(ex(*ins))
            else:
                ent = self._packs.get((k, sig))
                if ent is None or not getattr(ex, "accepts_out", False):
                    outs = list(ex(*ins))
                else:
                    outs = list(ex(*ins, out=ent[2]))
                if ent is None:
                    # first evaluation: learn the partial shapes, build the packed buffers
                    shapes = [tuple(getattr(outs[pos], "shape", ())) for pos, _, _ in exchanges]
                    like = next((o for o in outs if isinstance(o, torch.Tensor)), None)
                    if like is None:
                        like = next((x for x in local_inputs if isinstance(x, torch.Tensor)),
                                    torch.zeros(1))
                    bufs, views = self._pack_buffers(k, shapes, like)
                    targets = [views.get(pos) for pos in range(len(p_k.outputs))]
                    ent = self._packs[(k, sig)] = (bufs, views, targets)
                bufs, views, _ = ent
                for pos, _rop, xdt in exchanges:
                    o = outs[pos]
                    if not isinstance(o, torch.Tensor):
                        o = torch.as_tensor(np.asarray(o), dtype=_torch_dtype(xdt))
                    if o.data_ptr() != views[pos].data_ptr():
                        views[pos].copy_(o.reshape(views[pos].shape))
                    outs[pos] = views[pos]
                if world > 1 or self.force_collectives:
                    last = k + 1 == len(spec.rounds) or not any(
                        q.nodes for q, _ in spec.rounds[k + 1:])
                    for (rop, _xdt), buf in bufs.items():
                        h = _dist_reduce(buf, rop, self.group, async_op=async_op and last)
                        if async_op and last:
                            handles.append(h)
            results.append(outs)
            for v, o in zip(p_k.outputs, outs):
                carried_vals[v] = o
        final = [results[r][pos] for r, pos in spec.out_src]
        if not self.borrow and not async_op:
            final = [o.clone() if isinstance(o, torch.Tensor) else o for o in final]
        return (final, handles) if async_op else final


def _call_single_list(self, local_inputs, sig, world):
    """Serve an evaluation from ONE recorded launch list (rounds + all-reduces).  Returns the
    outputs, or None when this signature cannot be served that way (yet): the first evaluation of
    a signature takes the ordinary path (it learns the partial shapes and builds the packed
    exchange buffers), the second records, later ones replay.  Anything that would need work
    outside the list — a host input, an input living at a new address, a round whose partial
    does not land in the packed buffer by itself, data-dependent host control flow — keeps the
    per-round path."""
    import ctypes as C
    import torch
    from ._lib import check, lib
    from .executor import HostReadInReplay
    if not all(isinstance(x, torch.Tensor) and x.is_cuda for x in local_inputs):
        return None
    # the recorded launches address these buffers with this layout: dtype, shape, strides AND
    # base address are the signature (a transposed view / a reinterpretation of the same buffer
    # is another signature), as PlanExecutor._layout_sig keys its own replays
    key = (sig, tuple((x.dtype, tuple(x.shape), x.stride(), x.data_ptr()) for x in local_inputs))
    ent = self._lists.get(key)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if ent is not None:
        self._lists[key] = self._lists.pop(key)               # most recently used last
        check(lib.ahip_list_run(ent[0], stream))
        self.replays += 1
        self._check_after_list()
        final = ent[1]
        return final if self.borrow else [o.clone() if isinstance(o, torch.Tensor) else o for o in final]
    if key in self._no_list:
        return None
    spec = self.spec
    n_orig = len(self.plan.inputs)
    if any(exchanges and (k, sig) not in self._packs for k, (_p, exchanges) in enumerate(spec.rounds)):
        return None                       # packed buffers not built yet: ordinary first evaluation
    if not all(hasattr(ex, "trace_eager") for ex in self.execs):
        return None
    # pass 1 below traces every round EAGERLY on the un-reduced local partials (the collective is
    # issued once, by the list): a round behind an exchange whose HOST behaviour depends on values
    # (the bad-index check of a gather / scatter, an Assert, the trip count of a do-while Scan)
    # would see partial sums, raise spuriously or record another allocation trace than the
    # combined values give — such plans keep the per-round path
    if any(exchanges_before and _value_dependent_host_logic(p_k)
           for exchanges_before, p_k in _rounds_behind_an_exchange(spec.rounds)):
        self._no_list.add(key)
        return None

    def run_rounds(recording):
        carried, results = {}, []
        for k, ((p_k, exchanges), ex) in enumerate(zip(spec.rounds, self.execs)):
            ins = list(local_inputs) + [carried[v] for v in p_k.inputs[n_orig:]]
            if not p_k.nodes:
                outs = [ins[p_k.inputs.index(v)] for v in p_k.outputs]
            else:
                targets = self._packs[(k, sig)][2] if exchanges else None
                outs = ex.record_external() if recording else ex.trace_eager(ins, out=targets)
            if exchanges:
                bufs, views, _ = self._packs[(k, sig)]
                for pos, _rop, _xdt in exchanges:
                    o = outs[pos]
                    if not isinstance(o, torch.Tensor) or o.data_ptr() != views[pos].data_ptr():
                        raise _NoSingleList("round %d: partial %d is not produced in the packed buffer" % (k, pos))
                    outs[pos] = views[pos]
                # pass 1 only learns the allocation trace (shapes never depend on the combined
                # values): the collective is issued ONCE per evaluation, by the list run below —
                # a rank that records and a rank that replays (or falls back to the per-round
                # path) therefore issue the same number of all-reduces and can never mis-pair
                if recording and (world > 1 or self.force_collectives):
                    for (rop, _xdt), buf in bufs.items():
                        self.group.all_reduce(buf, rop)       # while recording: a REC_ALLREDUCE entry
            results.append(outs)
            for v, o in zip(p_k.outputs, outs):
                carried[v] = o
        return [results[r][pos] for r, pos in spec.out_src]

    try:
        run_rounds(False)                                     # pass 1: eager, allocation traces
        lst = C.c_void_p()
        check(lib.ahip_list_begin())
        try:
            final = run_rounds(True)                          # pass 2: everything into ONE list
        finally:
            rc = lib.ahip_list_end(C.byref(lst))
        check(rc)
    except (_NoSingleList, HostReadInReplay, NotImplementedError):
        self._no_list.add(key)
        return None
    check(lib.ahip_list_run(lst, stream))                     # this call's results, from the list
    # what the list addresses must outlive it: the inputs, and EVERY round's planned arena with
    # its bound values (an executor's ``_ext`` slot is overwritten by the next signature that is
    # recorded — the entry owns them, not the executor)
    exts = [ex.take_external() for ex in self.execs if hasattr(ex, "take_external")]
    self._lists[key] = (lst, final, list(local_inputs), exts)
    self.group.recorded += 1                                  # the list holds the communicator's handle
    while len(self._lists) > self.LISTS_MAX:                  # least recently used signature goes
        self._drop_list(next(iter(self._lists)))
    self.replays += 1
    self._check_after_list()
    return final if self.borrow else [o.clone() if isinstance(o, torch.Tensor) else o for o in final]


def _drop_list(self, key):
    from ._lib import lib
    ent = self._lists.pop(key)
    lib.ahip_list_destroy(ent[0])
    self.group.recorded -= 1


_VALUE_DEPENDENT = ("Assert", "AdvancedSubtensor1", "AdvancedSubtensor", "AdvancedIncSubtensor1",
                    "AdvancedIncSubtensor", "Nonzero", "HostCall")


def _value_dependent_host_logic(plan):
    for n in plan.nodes:
        if n.op in _VALUE_DEPENDENT:
            return True
        if n.op == "Scan" and (n.params.get("as_while") or _value_dependent_host_logic(n.params["inner"])):
            return True
    return False


def _rounds_behind_an_exchange(rounds):
    seen = False
    for p_k, exchanges in rounds:
        yield seen, p_k
        seen = seen or bool(exchanges)


def _check_after_list(self):
    """Error words (bad index, persistent-Scan time-out) of every round's executor, examined
    after a single-list run as the per-round path does after each round (``check_indices``:
    True = the blocking read, "deferred" = the copy behind the launches, examined when it lands)."""
    for ex in self.execs:
        mode = getattr(ex, "check_indices", False)
        if mode == "deferred" and hasattr(ex, "_deferred_check"):
            ex._deferred_check()
        elif mode and hasattr(ex, "_raise_bad_index"):
            ex._raise_bad_index()


def _close_lists(self):
    """Destroy every recorded list (they hold the communicator's handle: ``HipComm.close``)."""
    for key in list(self._lists):
        self._drop_list(key)


class _NoSingleList(Exception):
    pass


ShardedPlan._call_single_list = _call_single_list
ShardedPlan._drop_list = _drop_list
ShardedPlan._check_after_list = _check_after_list
ShardedPlan.close = _close_lists
ShardedPlan.LISTS_MAX = 16      # recorded signatures kept per sharded plan (LRU)


def run_local_shards(plan: Plan, split_inputs: Dict[int, int], shard_inputs: Sequence[Sequence],
                     executor_factory=None, use_graph=False):
    """Evaluate ``plan`` over k logical shards in THIS process (one device): every shard runs
    each round, the partials are combined with :class:`LocalGroup` between rounds — the same
    code path as the multi-process run minus the transport.  Returns the per-shard outputs."""
    import torch
    k = len(shard_inputs)
    grp = LocalGroup(k)
    sps = [ShardedPlan(plan, split_inputs, group=grp, use_graph=use_graph,
                       executor_factory=executor_factory) for _ in range(k)]
    spec = sps[0].spec
    carried = [dict() for _ in range(k)]
    results = [[] for _ in range(k)]
    n_orig = len(plan.inputs)
    for r, (p_r, exchanges) in enumerate(spec.rounds):
        outs_all = []
        for s in range(k):
            ins = list(shard_inputs[s]) + [carried[s][v] for v in p_r.inputs[n_orig:]]
            outs_all.append(list(sps[s].execs[r](*ins)))
        for pos, rop, xdt in exchanges:
            bufs = []
            for s in range(k):
                o = outs_all[s][pos]
                if not isinstance(o, torch.Tensor):
                    o = torch.as_tensor(np.asarray(o), dtype=_torch_dtype(xdt))
                bufs.append(o.clone())
            grp.all_reduce_many(bufs, rop)
            for s in range(k):
                outs_all[s][pos] = bufs[s]
        for s in range(k):
            results[s].append(outs_all[s])
            for v, o in zip(p_r.outputs, outs_all[s]):
                carried[s][v] = o
    return [[results[s][r][pos] for r, pos in spec.out_src] for s in range(k)], spec


# ---------------------------------------------------------------------------------------------
# independent-output placement
# ---------------------------------------------------------------------------------------------
def _ancestors(plan: Plan, out: int, producer: Dict[int, int]) -> set:
    seen, stack = set(), [out]
    while stack:
        v = stack.pop()
        ni = producer.get(v)
        if ni is None or ni in seen:
            continue
        seen.add(ni)
        stack.extend(plan.nodes[ni].inputs)
    return seen


def _node_cost(plan: Plan, n: Node) -> int:
    """Static work estimate used only to balance ranks: heavy for contractions and Scan, the
    operand count for streaming ops (shapes are dynamic, so this is a rank, not a time)."""
    if n.op in ("Gemm", "Dot22", "Dot22Scalar", "BatchedDot", "MatMul"):
        return 64
    if n.op == "Scan":
        return 256
    if n.op in ("Gemv", "Dot", "Ger"):
        return 8
    if n.op in ("Elemwise", "CAReduce"):
        return 1 + sum(plan.vars[i].ndim > 0 for i in n.inputs)
    return 1


def place_outputs(plan: Plan, world: int) -> List[List[int]]:
    """Assign plan outputs to ranks so that no computed value is needed on two ranks.

    Outputs whose ancestor node sets intersect are kept together (one component); components
    are placed largest-first on the least-loaded rank.  Returns, per rank, the positions of
    the outputs it computes (plan inputs / constants are replicated, never communicated)."""
    producer = {o: ni for ni, n in enumerate(plan.nodes) for o in n.outputs}
    anc = [_ancestors(plan, o, producer) for o in plan.outputs]
    comp = list(range(len(plan.outputs)))

    def find(a):
        while comp[a] != a:
            comp[a] = comp[comp[a]]
            a = comp[a]
        return a
    for i in range(len(anc)):
        for j in range(i):
            if anc[i] & anc[j]:
                comp[find(i)] = find(j)
    groups: Dict[int, List[int]] = {}
    for i in range(len(anc)):
        groups.setdefault(find(i), []).append(i)
    costs = []
    for root, members in groups.items():
        nodes = set().union(*(anc[m] for m in members))
        costs.append((sum(_node_cost(plan, plan.nodes[ni]) for ni in nodes), members))
    costs.sort(key=lambda t: (-t[0], t[1]))
    load = [0] * world
    placed: List[List[int]] = [[] for _ in range(world)]
    for c, members in costs:
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += max(c, 1)
        placed[r].extend(members)
    return [sorted(p) for p in placed]


def subplan_for_outputs(plan: Plan, positions: Sequence[int]) -> Plan:
    """The part of ``plan`` that computes the outputs at ``positions`` (same inputs)."""
    producer = {o: ni for ni, n in enumerate(plan.nodes) for o in n.outputs}
    keep = set()
    for p in positions:
        keep |= _ancestors(plan, plan.outputs[p], producer)
    return Plan(plan.name + "_outs" + "_".join(map(str, positions)), plan.vars, list(plan.inputs),
                [plan.outputs[p] for p in positions], [n for ni, n in enumerate(plan.nodes)
                                                       if ni in keep])


class PlacedPlan:
    """Independent graph outputs on different GPUs (module docstring, 2.): this rank evaluates
    only the outputs :func:`place_outputs` gave it; ``__call__`` returns a list with ``None`` in
    the positions other ranks own — no communication."""

    def __init__(self, plan: Plan, world: int, rank: int, use_graph=False, executor_factory=None,
                 device=None):
        self.placement = place_outputs(plan, world)
        self.mine = self.placement[rank]
        self.n_out = len(plan.outputs)
        self.exec = None
        if self.mine:
            sub = subplan_for_outputs(plan, self.mine)
            if executor_factory is None:
                from .executor import PlanExecutor
                self.exec = PlanExecutor(sub, use_graph=use_graph, device=device)   # fresh outputs
            else:
                self.exec = executor_factory(sub)

    def owner(self, position: int) -> int:
        return next(r for r, ps in enumerate(self.placement) if position in ps)

    def __call__(self, *inputs):
        res = [None] * self.n_out
        if self.exec is not None:
            for p, o in zip(self.mine, self.exec(*inputs)):
                res[p] = o
        return res


class ShardedFunction:
    """Thin combiner kept for callers that evaluate the local block themselves: outputs tagged
    "allreduce" are summed over the process group (asynchronously on request, so consecutive
    evaluations pipeline behind the latency-bound collective)."""

    def __init__(self, executor, kinds: Sequence[str], group=None):
        self.executor = executor
        self.kinds = list(kinds)
        self.group = group

    def __call__(self, *local_inputs, async_op=False):
        import torch.distributed as dist

        outs = list(self.executor(*local_inputs))
        handles = []
        if isinstance(self.group, HipComm):
            for o, k in zip(outs, self.kinds):
                if k == "allreduce":
                    handles.append(self.group.all_reduce(o, "add"))
            return (outs, handles if async_op else []) if async_op else outs
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            for o, k in zip(outs, self.kinds):
                if k == "allreduce":
                    h = dist.all_reduce(o, op=dist.ReduceOp.SUM, group=self.group,
                                        async_op=async_op)
                    if async_op:
                        handles.append(h)
        return (outs, handles) if async_op else outs
