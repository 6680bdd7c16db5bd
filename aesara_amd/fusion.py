"""Plan-level fusion performed by the HIP linker before execution.

The reference's rewriter already fuses Elemwise chains into ``Composite`` ops
(tensor/rewriting/elemwise.py:523 local_elemwise_fusion_op) but (a) never fuses an Elemwise into
the CAReduce that consumes it for a non-C target (``local_careduce_fusion`` :942 is C-only and
single-input) and (b) leaves tiny broadcast producers (``square(sigma)`` in BASELINE config 2)
as separate nodes.  On a GPU every extra node is a kernel boundary (≈1.5–2 µs, MI355X guide
"boundary" row) and, for Elemwise→Sum, a full HBM round trip of the intermediate — so the linker
fuses them here, where it is legal: it sees adjacent nodes, client lists and output status.

Produces a list of :class:`Step`:

* ``kind == "elemwise"`` — one generated kernel: ``scalar`` over ``inputs`` -> ``outputs``
* ``kind == "reduce"``   — Elemwise producer (possibly identity) + CAReduce in one kernel
* ``kind == "node"``     — any other plan node, executed by its handler
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

from ._lib import AHIP_MAXOPS
from .plan import Node, Plan

MAX_FUSED_OPERANDS = AHIP_MAXOPS - 2


@dataclass
class Step:
    kind: str
    inputs: List[int]
    outputs: List[int]
    scalar: Optional[dict] = None
    out_refs: List[int] = field(default_factory=list)  # scalar["out"] index per stored output
    reduce: Optional[Dict[str, Any]] = None            # scalar_op, axis, acc_dtype, out, ref
    node: Optional[Node] = None
    alive: bool = True


def _inline_constants(step: Step, plan: Plan):
    """Fold size-1 plan constants into the scalar expression as literals."""
    keep, remap = [], {}
    for pos, vid in enumerate(step.inputs):
        v = plan.vars[vid]
        if v.const is not None and len(v.const["data"]) == 1:
            val = v.const["data"][0]
            remap[pos] = ["c", val, v.dtype]
        else:
            if vid in keep:
                remap[pos] = ["i", keep.index(vid)]
            else:
                keep.append(vid)
                remap[pos] = ["i", len(keep) - 1]
    step.scalar = _remap_inputs(step.scalar, remap, len(keep))
    step.inputs = keep


def _remap_inputs(scalar, remap, n_in, t_shift=0):
    def rr(r):
        if r[0] == "i":
            return list(remap[r[1]])
        if r[0] == "t":
            return ["t", r[1] + t_shift]
        return list(r)

    return {"n_in": n_in,
            "nodes": [{"op": n["op"], "dtype": n["dtype"], "in": [rr(r) for r in n["in"]]}
                      for n in scalar["nodes"]],
            "out": [rr(r) for r in scalar["out"]]}


def _merge_producer(cons: Step, prod: Step, vid: int):
    """Inline single-output elemwise ``prod`` (producing ``vid``) into ``cons``."""
    new_inputs = []

    def slot(v):
        if v not in new_inputs:
            new_inputs.append(v)
        return new_inputs.index(v)

    # producer nodes first
    p_remap = {pos: ["i", slot(v)] for pos, v in enumerate(prod.inputs)}
    ps = _remap_inputs(prod.scalar, p_remap, 0)
    p_out = ps["out"][prod.out_refs[prod.outputs.index(vid)]]
    shift = len(ps["nodes"])
    c_remap = {}
    for pos, v in enumerate(cons.inputs):
        c_remap[pos] = list(p_out) if v == vid else ["i", slot(v)]
    cs = _remap_inputs(cons.scalar, c_remap, 0, t_shift=shift)
    cons.scalar = {"n_in": len(new_inputs), "nodes": ps["nodes"] + cs["nodes"], "out": cs["out"]}
    cons.inputs = new_inputs


def build_steps(plan: Plan, fuse: bool = True) -> List[Step]:
    clients = plan.clients()
    out_set = set(plan.outputs)
    steps: List[Step] = []
    producer: Dict[int, Step] = {}

    for ni, node in enumerate(plan.nodes):
        if node.op == "Elemwise":
            sc = copy.deepcopy(node.params["scalar"])
            st = Step("elemwise", list(node.inputs), list(node.outputs), sc,
                      out_refs=list(range(len(node.outputs))))
            _inline_constants(st, plan)
            if fuse:
                changed = True
                while changed:
                    changed = False
                    for vid in list(st.inputs):
                        p = producer.get(vid)
                        if (p is None or not p.alive or p.kind != "elemwise"
                                or len(p.outputs) != 1 or vid in out_set):
                            continue
                        if any(c[0] != ni for c in clients[vid]):
                            continue
                        n_ops = len(set(st.inputs + p.inputs) - {vid}) + len(st.outputs)
                        if n_ops > MAX_FUSED_OPERANDS:
                            continue
                        _merge_producer(st, p, vid)
                        p.alive = False
                        changed = True
                        break
            steps.append(st)
            for o in st.outputs:
                producer[o] = st
        elif node.op == "CAReduce":
            vid = node.inputs[0]
            red = {"scalar_op": node.params["scalar_op"], "axis": node.params["axis"],
                   "acc_dtype": node.params["acc_dtype"], "out": node.outputs[0], "ref": None}
            p = producer.get(vid) if fuse else None
            if p is not None and p.alive and p.kind == "elemwise" and \
                    len(p.inputs) + len(p.outputs) + 1 <= MAX_FUSED_OPERANDS:
                others = [c for c in clients[vid] if c[0] != ni]
                k = p.outputs.index(vid)
                red["ref"] = p.out_refs[k]
                p.kind = "reduce"
                p.reduce = red
                if not others and vid not in out_set:
                    # intermediate never leaves the kernel
                    del p.outputs[k]
                    del p.out_refs[k]
                producer[node.outputs[0]] = p
                # the fused step now also defines the reduce output; keep its position
                continue
            st = Step("reduce", [vid], [], {"n_in": 1, "nodes": [], "out": [["i", 0]]},
                      out_refs=[], reduce=dict(red, ref=0))
            steps.append(st)
            producer[node.outputs[0]] = st
        else:
            st = Step("node", list(node.inputs), list(node.outputs), node=node)
            steps.append(st)
            for o in node.outputs:
                producer[o] = st
    return [s for s in steps if s.alive]
