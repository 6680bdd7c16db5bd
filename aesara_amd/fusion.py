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
* ``kind == "rowdot"``   — D = A . x (a Gemv stripped of its alpha/beta epilogue)
* ``kind == "gemv_epi"`` — chain of row dots + the Elemwise that consumes them, one kernel
* ``kind == "node"``     — any other plan node, executed by its handler

``Gemv(y, alpha, A, x, beta)`` (tensor/blas.py:231) is split into ``D = rowdot(A, x)`` and the
elementwise ``beta*y + alpha*D`` so that the ordinary Elemwise fusion merges Gemv chains
(``y`` itself a Gemv) and neighbouring Composites around the dot products; whatever elementwise
step finally consumes the dots becomes one ``gemv_epi`` kernel (one GRU gate of BASELINE
config 4 = 1 launch instead of 5).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import numpy as np

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
    dots: List[List[int]] = field(default_factory=list)  # gemv_epi: [A var, x var] per dot;
    #                               the scalar's first len(dots) inputs are the dot results
    extra: Dict[str, Any] = field(default_factory=dict)  # rowpass: reds, col_ref, ws vars
    post: List["Step"] = field(default_factory=list)     # rowpass: fold steps run after it
    fallback: List["Step"] = field(default_factory=list)  # rowpass: the unfused original steps


def _inline_constants(step: Step, plan: Plan):
    """Fold size-1 plan constants into the scalar expression as literals."""
    keep, remap = [], {}

    def foldable(vid):
        v = plan.vars[vid]
        return v.const is not None and len(v.const.get("data", ())) == 1
    # a step whose operands are ALL size-1 constants keeps the one of the highest rank as a real
    # operand: the step's shape (and the axes of a fused reduction) come from its operands
    anchor = None
    if step.inputs and all(foldable(v) for v in step.inputs):
        anchor = max(step.inputs, key=lambda v: len(plan.vars[v].shape))
        if not plan.vars[anchor].shape:
            anchor = None
    for pos, vid in enumerate(step.inputs):
        v = plan.vars[vid]
        if vid != anchor and foldable(vid):
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
        gemv_split = fuse and node.op == "Gemv" and _gemv_splittable(plan, node)
        if node.op == "Elemwise" or gemv_split:
            if gemv_split:
                st = _split_gemv(plan, node, steps, producer)
            else:
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
    steps = [s for s in steps if s.alive]
    if fuse:
        steps = _drop_dead(plan, _fuse_dots(plan, steps))
        steps = _fuse_rowpass(plan, steps)
        steps = _fuse_rowchain(plan, steps)
        steps = _fuse_gemm_epi(plan, steps)
        steps = _fuse_xprologue(plan, steps)
    return steps


_PURE_NODES = {"AllocEmpty", "Alloc", "DimShuffle", "Shape_i", "Shape", "ViewOp", "SpecifyShape",
               "Subtensor", "Reshape", "ScalarFromTensor", "TensorFromScalar", "MakeVector",
               "BroadcastTo", "DeepCopyOp"}


def _drop_dead(plan: Plan, steps: List[Step]) -> List[Step]:
    """Remove side-effect-free steps whose outputs nobody reads any more (e.g. the AllocEmpty
    that only fed the ignored ``y`` of a ``beta == 0`` Gemv)."""
    live = set(plan.outputs)
    keep = []
    for s in reversed(steps):
        outs = list(s.outputs) + ([s.reduce["out"]] if s.reduce else [])
        pure = s.kind in ("elemwise", "rowdot", "gemv_epi") or \
            (s.kind == "node" and s.node.op in _PURE_NODES)
        if pure and not any(o in live for o in outs):
            continue
        keep.append(s)
        live.update(s.inputs)
        for d in s.dots:
            live.update(d)
    return list(reversed(keep))


def _gemv_splittable(plan: Plan, node: Node) -> bool:
    y, alpha, A, x, beta = (plan.vars[i] for i in node.inputs)
    return (A.ndim == 2 and x.ndim == 1 and A.dtype in ("float32", "float64")
            and A.dtype == x.dtype == y.dtype)


def _split_gemv(plan: Plan, node: Node, steps, producer) -> Step:
    """Gemv(y, alpha, A, x, beta) -> rowdot step + elementwise step beta*y + alpha*D."""
    yv, av, Av, xv, bv = node.inputs
    dt = plan.vars[node.outputs[0]].dtype
    d_var = plan.new_var(dt, [None], name="rowdot")
    dot = Step("rowdot", [Av, xv], [d_var])
    steps.append(dot)
    producer[d_var] = dot
    beta = plan.vars[bv]
    beta_zero = beta.const is not None and len(beta.const.get("data", ())) == 1 and \
        float(beta.const["data"][0]) == 0.0
    if beta_zero:
        # BLAS semantics: beta == 0 never reads y (it is usually an uninitialised AllocEmpty)
        sc = {"n_in": 2, "nodes": [{"op": "mul", "in": [["i", 0], ["i", 1]], "dtype": dt}],
              "out": [["t", 0]]}
        return Step("elemwise", [av, d_var], list(node.outputs), sc, out_refs=[0])
    if beta.const is not None and len(beta.const.get("data", ())) == 1:
        sc = {"n_in": 4, "nodes": [{"op": "mul", "in": [["i", 3], ["i", 0]], "dtype": dt},
                                   {"op": "mul", "in": [["i", 1], ["i", 2]], "dtype": dt},
                                   {"op": "add", "in": [["t", 0], ["t", 1]], "dtype": dt}],
              "out": [["t", 2]]}
        return Step("elemwise", [yv, av, d_var, bv], list(node.outputs), sc, out_refs=[0])
    # a RUN-TIME beta: the same rule per evaluation — beta == 0 never reads y (tensor/blas.py:236
    # Gemv.perform / BLAS xGEMV: a NaN or inf in y does not reach the result;
    # tests/tensor/test_blas_c.py:146 test_nan_beta_0)
    sc = {"n_in": 4, "nodes": [{"op": "mul", "in": [["i", 3], ["i", 0]], "dtype": dt},
                               {"op": "eq", "in": [["i", 3], ["c", 0.0, dt]], "dtype": "bool"},
                               {"op": "switch", "in": [["t", 1], ["c", 0.0, dt], ["t", 0]], "dtype": dt},
                               {"op": "mul", "in": [["i", 1], ["i", 2]], "dtype": dt},
                               {"op": "add", "in": [["t", 2], ["t", 3]], "dtype": dt}],
          "out": [["t", 4]]}
    return Step("elemwise", [yv, av, d_var, bv], list(node.outputs), sc, out_refs=[0])


def _fuse_dots(plan: Plan, steps: List[Step]) -> List[Step]:
    """Absorb rowdot steps into the elementwise step that consumes them (kind gemv_epi)."""
    from .codegen import AHIP_GV_MAXOPS, AHIP_MAXDOTS

    dot_of = {s.outputs[0]: s for s in steps if s.kind == "rowdot"}
    if not dot_of:
        return steps
    users: Dict[int, List[Step]] = {}
    for s in steps:
        for v in set(s.inputs):
            users.setdefault(v, []).append(s)
    outs = set(plan.outputs)
    for s in steps:
        if s.kind != "elemwise":
            continue
        dvars = [v for v in s.inputs if v in dot_of and len(users[v]) == 1 and v not in outs]
        if not dvars or len(dvars) > AHIP_MAXDOTS:
            continue
        others = [v for v in s.inputs if v not in dvars]
        if len(others) + len(s.outputs) > AHIP_GV_MAXOPS:
            continue
        # reorder the scalar inputs: dot results first, then the other operands
        order = dvars + others
        remap = {pos: ["i", order.index(v)] for pos, v in enumerate(s.inputs)}
        s.scalar = _remap_inputs(s.scalar, remap, len(order))
        s.inputs = others
        s.dots = [list(dot_of[v].inputs) for v in dvars]
        s.kind = "gemv_epi"
        for v in dvars:
            dot_of[v].alive = False
    return [s for s in steps if s.alive]


def split_invariant(plan: Plan, invariant_inputs: List[int]):
    """Hoist loop-invariant nodes out of a Scan inner plan.

    ``invariant_inputs``: ids of the inner inputs that do not change across steps (the
    non-sequences).  Returns ``(pre_plan, loop_plan, hoisted)``: ``pre_plan`` maps the
    invariant inputs to the hoisted values (run once per Scan call), ``loop_plan`` takes the
    original inputs followed by the hoisted values.  Only view/shape ops are hoisted (they are
    free to recompute but hide loop-invariant *layouts* — e.g. the ``W.T`` DimShuffles feeding
    every Gemv of a GRU step — from the executor, which materialises them once)."""
    hoistable = {"DimShuffle", "Shape_i", "Shape", "ViewOp", "SpecifyShape", "Subtensor",
                 "Reshape", "ScalarFromTensor", "TensorFromScalar", "MakeVector"}
    inv = set(invariant_inputs)
    const_ids = {vid for vid, v in plan.vars.items() if v.const is not None}
    pre_nodes, loop_nodes = [], []
    for n in plan.nodes:
        if n.op in hoistable and all(i in inv or i in const_ids for i in n.inputs):
            pre_nodes.append(n)
            inv.update(n.outputs)
        else:
            loop_nodes.append(n)
    hoisted = []
    used = {i for n in loop_nodes for i in n.inputs} | set(plan.outputs)
    for n in pre_nodes:
        hoisted.extend(o for o in n.outputs if o in used)
    if not hoisted:
        return None, plan, []
    pre = Plan(plan.name + "_invariant", plan.vars, list(invariant_inputs), hoisted, pre_nodes)
    loop = Plan(plan.name + "_loop", plan.vars, list(plan.inputs) + hoisted, list(plan.outputs),
                loop_nodes)
    return pre, loop, hoisted


def split_column_slices(plan: Plan, invariant: set) -> Plan:
    """Fused-gate recurrences compute ONE product for all gates and slice it along its last axis::

        pre = b + x_t @ Wx + h @ Wh          # [B, 4H]   (or a vector of length 4H)
        i, f, o, g = pre[:, 0:H], pre[:, H:2H], ...

    (the usual Theano / Aesara LSTM step).  Every consumer sees only its slice, so the chain is
    split per slice with views of the invariant operands — ``x_t @ Wx[:, kH:(k+1)H]`` … — which
    leaves H-wide products with H-wide epilogues: the shape the sequence hoisting and the
    persistent kernels work on.  Handles a chain of ``Dot22`` / ``Gemm`` (matrix state) or ``Gemv``
    (vector state; the slice selects rows of the matrix) with an invariant matrix operand and
    ``Elemwise`` nodes (chain operands and invariant rows ``[1, N]`` / vectors), every value of
    which is read only inside the chain or by ``Subtensor`` nodes ``[..., a:b]`` with constant
    bounds.  Returns the plan unchanged when nothing matches."""
    V = plan.vars
    clients = plan.clients()
    # invariants and views of them computed in the step (the W.T of every Gemv)
    inv = set(invariant)
    for n in plan.nodes:
        if n.op in ("DimShuffle", "ViewOp") and all(i in inv for i in n.inputs):
            inv.update(n.outputs)

    def cint(vid):
        v = V[vid]
        if v.const is not None and len(v.const.get("data", ())) == 1 and v.dtype.startswith("int"):
            return int(v.const["data"][0])
        return None

    def last_slice(n):
        """(a, b) when node n is Subtensor [..., a:b] on the last axis with constant bounds"""
        if n.op != "Subtensor":
            return None
        nd = V[n.inputs[0]].ndim
        idx = n.params["idx_list"]
        if nd not in (1, 2) or len(idx) != nd or any("slice" not in e for e in idx):
            return None
        if nd == 2 and idx[0]["slice"] != [None, None, None]:
            return None
        if idx[-1]["slice"][2] not in (None, 1):
            return None
        extra, ab = list(n.inputs[1:]), []
        for e in idx[-1]["slice"][:2]:
            if e == "in":
                c = cint(extra.pop(0)) if extra else None
                if c is None:
                    return None
                ab.append(c)
            elif isinstance(e, int):
                ab.append(e)
            else:
                return None
        return (ab[0], ab[1]) if 0 <= ab[0] < ab[1] and not extra else None

    producer = {o: (ni, n) for ni, n in enumerate(plan.nodes) for o in n.outputs}

    def chain_of(root):
        """node indices of the chain ending in `root`, or None"""
        nd = V[root].ndim
        chain, todo = set(), [root]
        while todo:
            v = todo.pop()
            if v not in producer:
                return None
            ni, n = producer[v]
            if ni in chain:
                continue
            if nd == 2 and n.op in ("Dot22", "Dot") and len(n.inputs) == 2 and n.inputs[1] in inv \
                    and V[n.inputs[1]].ndim == 2 and V[n.outputs[0]].ndim == 2:
                chain.add(ni)
            elif nd == 2 and n.op == "Gemm" and n.inputs[3] in inv and V[n.inputs[3]].ndim == 2:
                chain.add(ni)
                todo.append(n.inputs[0])
            elif nd == 1 and n.op == "Gemv" and n.inputs[2] in inv and V[n.inputs[2]].ndim == 2:
                chain.add(ni)
                y = n.inputs[0]
                if not (y in producer and producer[y][1].op == "AllocEmpty"):
                    todo.append(y)                 # beta != 0: accumulates onto a chain value
            elif n.op == "Elemwise" and len(n.outputs) == 1 and V[n.outputs[0]].ndim == nd \
                    and V[n.outputs[0]].shape[-1] != 1:
                chain.add(ni)
                for i in n.inputs:
                    if (i in inv or V[i].const is not None) and V[i].ndim == nd and \
                            (nd == 1 or V[i].shape[0] == 1):
                        continue
                    todo.append(i)
            else:
                return None
        return chain

    new_nodes, replaced, removed = {}, {}, set()
    work = Plan(plan.name, dict(plan.vars), list(plan.inputs), list(plan.outputs), [])
    for root, (rni, rn) in list(producer.items()):
        cl = clients.get(root, [])
        if len(cl) < 2 or any(c[0] == "out" for c in cl):
            continue
        slices = [last_slice(plan.nodes[c[0]]) if c[1] == 0 else None for c in cl]
        if any(sl is None for sl in slices):
            continue
        chain = chain_of(root)
        if not chain or chain & removed:
            continue
        nd = V[root].ndim
        chain_vars = {o for ni in chain for o in plan.nodes[ni].outputs}
        if not all(all(c[0] != "out" and (c[0] in chain or (v == root and last_slice(plan.nodes[c[0]])))
                       for c in clients[v]) for v in chain_vars):
            continue
        order = sorted(chain)
        for (cni, _pos), (a, b) in zip(cl, slices):
            m = {}
            sink = new_nodes.setdefault(cni, [])

            def part(v, axis, a=a, b=b, m=m, sink=sink):
                """v[..., a:b] (axis = -1) or v[a:b] (axis = 0: rows of a Gemv matrix, the y of a
                beta = 0 Gemv) of an operand that is not a chain value"""
                key = ("p", v, axis)
                if key not in m:
                    src = V[v]
                    shp = list(src.shape)
                    shp[axis] = None
                    o = work.new_var(src.dtype, shp)
                    idx = [{"slice": [a, b, None]}] if axis == 0 else \
                        [{"slice": [None, None, None]}] * (src.ndim - 1) + [{"slice": [a, b, None]}]
                    sink.append(Node("Subtensor", [v], [o], {"idx_list": idx}))
                    m[key] = o
                return m[key]
            for ni in order:
                n = plan.nodes[ni]
                ov = V[n.outputs[0]]
                out = work.new_var(ov.dtype, list(ov.shape[:-1]) + [None])
                if n.op in ("Dot22", "Dot"):
                    nn = Node("Dot22", [n.inputs[0], part(n.inputs[1], -1)], [out], {})
                elif n.op == "Gemm":
                    nn = Node("Gemm", [m[n.inputs[0]], n.inputs[1], n.inputs[2], part(n.inputs[3], -1),
                                       n.inputs[4]], [out], dict(n.params))
                elif n.op == "Gemv":
                    y = n.inputs[0]
                    nn = Node("Gemv", [m[y] if y in m else part(y, 0), n.inputs[1], part(n.inputs[2], 0),
                                       n.inputs[3], n.inputs[4]], [out], dict(n.params))
                else:
                    # an operand that is broadcast along the sliced axis (static extent 1: a
                    # scalar handed in as [1, 1], a length-1 vector) has no columns a:b to cut —
                    # it stays whole and keeps broadcasting
                    nn = Node("Elemwise", [m[i] if i in m else (i if V[i].shape[-1] == 1 else part(i, -1))
                                           for i in n.inputs], [out], copy.deepcopy(n.params))
                m[n.outputs[0]] = out
                sink.append(nn)
            replaced[plan.nodes[cni].outputs[0]] = m[root]
            removed.add(cni)
        removed |= chain
    if not replaced:
        return plan
    nodes = []
    for ni, n in enumerate(plan.nodes):
        nodes.extend(new_nodes.get(ni, []))
        if ni not in removed:
            nodes.append(n)
    work.nodes = [Node(n.op, [replaced.get(i, i) for i in n.inputs], list(n.outputs), n.params) for n in nodes]
    work.outputs = [replaced.get(o, o) for o in plan.outputs]
    work.name = plan.name + "_cols"
    return _prune_dead_plan(work)


def _prune_dead_plan(plan: Plan) -> Plan:
    plan.nodes = _prune_dead(plan, list(plan.nodes))
    return plan


def push_out_accumulators(plan: Plan) -> Plan:
    """A gradient Scan sums what it computes for every step into sit-sot accumulators —
    ``acc_t = acc_{t-1} + g_t`` for the gradient of a non-sequence such as a bias, with
    ``g_t = sum(delta_t, axis=0)`` for a batch (Scan.L_op, scan/op.py:2379) — although ``delta_t`` is
    a nit-sot output of the same Scan anyway (the weight gradients are GEMMs over it after the loop).
    Such an accumulator is taken out of the loop: its whole buffer is rebuilt after the Scan as
    ``cumsum([init; reduce(delta_1); ...; reduce(delta_T)])`` (the loop's additions, associated as
    the CumOp kernel associates long columns — equal up to float reordering; one CAReduce + one CumOp
    over [T + 1, H]), cut — or, for a buffer longer than T + 1, zero-padded like the reference's
    scan/op.py:2139 — to the buffer length the caller allocated, so every use of the old output —
    whatever row it reads — sees the same values.  The step loses its batch reduction, which is
    what kept batched recurrences with biases off the persistent gradient kernel."""
    def is_add(sc):
        return (sc.get("n_in") == 2 and len(sc["nodes"]) == 1 and sc["nodes"][0]["op"] == "add"
                and sorted(map(tuple, sc["nodes"][0]["in"])) == [("i", 0), ("i", 1)] and sc["out"] == [["t", 0]])

    out_nodes, changed = [], False
    orig = plan
    plan = Plan(plan.name, dict(plan.vars), list(plan.inputs), list(plan.outputs), list(plan.nodes))
    replaced = {}
    for node in plan.nodes:
        if node.op != "Scan" or node.params.get("as_while") or not node.params.get("sit_sot_in_slices"):
            out_nodes.append(node)
            continue
        p = dict(node.params)
        inner = p["inner"]
        n_seqs = p["n_seqs"]
        mm_in = [list(t) for t in p.get("mit_mot_in_slices", [])]
        mm_out = [list(t) for t in p.get("mit_mot_out_slices", [])]
        ms = [list(t) for t in p["mit_sot_in_slices"]]
        ss = [list(t) for t in p["sit_sot_in_slices"]]
        n_mm, n_ms, n_ss, n_nit = len(mm_in), len(ms), len(ss), p["n_nit_sot"]
        n_sh = p.get("n_shared_outs", 0)
        tap0 = n_seqs + sum(len(t) for t in mm_in) + sum(len(t) for t in ms)      # first sit-sot tap
        out0 = sum(len(t) for t in mm_out) + n_ms                                  # first sit-sot output
        nit0 = out0 + n_ss
        iprod = {o: n_ for n_ in inner.nodes for o in n_.outputs}
        iclients = inner.clients()
        drop = []          # (q, nit index j, reduce params or None)
        for q in range(n_ss):
            if ss[q] != [-1]:
                continue
            tq, oq = inner.inputs[tap0 + q], inner.outputs[out0 + q]
            upd = iprod.get(oq)
            if upd is None or upd.op != "Elemwise" or not is_add(upd.params["scalar"]) or tq not in upd.inputs:
                continue
            if [c for c in iclients[tq]] != [c for c in iclients[tq] if c[0] != "out" and inner.nodes[c[0]] is upd] \
                    or len(iclients[tq]) != 1 or len(iclients[oq]) != 1:
                continue
            a = [i for i in upd.inputs if i != tq]
            if len(a) != 1:
                continue
            a, red = a[0], None
            an = iprod.get(a)
            if an is not None and an.op == "CAReduce" and an.params["scalar_op"] == "add" and \
                    an.params["axis"] == [0] and inner.vars[an.inputs[0]].ndim == 2:
                red, a = dict(an.params), an.inputs[0]
            nits = inner.outputs[nit0:nit0 + n_nit]
            if inner.vars[a].ndim != (2 if red else 1) or inner.vars[a].dtype != inner.vars[oq].dtype:
                continue
            if a in nits:
                drop.append((q, nits.index(a), red))
                continue
            # a vector handed out as a row ([1, n]: DimShuffle 'x', 0): summing the size-1 axis of
            # the [T, 1, n] output gives the [T, n] addends back
            via = [o for o in nits if o in iprod and iprod[o].op == "DimShuffle" and iprod[o].inputs[0] == a
                   and iprod[o].params["new_order"] == ["x", 0]]
            if red is None and via:
                drop.append((q, nits.index(via[0]), {"scalar_op": "add", "axis": [0],
                                                     "acc_dtype": inner.vars[a].dtype}))
        if not drop:
            out_nodes.append(node)
            continue
        changed = True
        dq = {q for q, _j, _r in drop}
        base_in = 1 + n_seqs + n_mm + n_ms            # position of the first sit-sot init in node.inputs
        new_inputs = [v for k, v in enumerate(node.inputs) if not (base_in <= k < base_in + n_ss and k - base_in in dq)]
        base_out = n_mm + n_ms
        new_outputs = [v for k, v in enumerate(node.outputs) if not (base_out <= k < base_out + n_ss and k - base_out in dq)]
        in_keep = [v for k, v in enumerate(inner.inputs) if not (tap0 <= k < tap0 + n_ss and k - tap0 in dq)]
        out_keep = [v for k, v in enumerate(inner.outputs) if not (out0 <= k < out0 + n_ss and k - out0 in dq)]
        new_inner = Plan(inner.name + "_noacc", inner.vars, in_keep, out_keep, list(inner.nodes))
        new_inner.nodes = _prune_dead(new_inner, list(new_inner.nodes))
        p["inner"] = new_inner
        p["sit_sot_in_slices"] = [t for q, t in enumerate(ss) if q not in dq]
        out_nodes.append(Node("Scan", new_inputs, new_outputs, p))
        V = plan.vars
        for q, j, red in drop:
            old = node.outputs[base_out + q]
            init_buf = node.inputs[base_in + q]
            vall = node.outputs[n_mm + n_ms + n_ss + j]
            dt = V[old].dtype
            width = V[init_buf].shape[1:]
            r = vall
            if red is not None:
                r = plan.new_var(dt, [None] + list(width))
                out_nodes.append(Node("CAReduce", [vall], [r], dict(red, axis=[1])))
            # rows 0..T of the old buffer: init, init + g_1, (init + g_1) + g_2, ... — ONE cumulative
            # sum over [init; g_1; ...; g_T] (the loop's additions up to the association the CumOp
            # kernel uses for long columns: tolerance-level, like any reordered float sum)
            row = plan.new_var(dt, [1] + list(width))
            out_nodes.append(Node("Subtensor", [init_buf], [row], {"idx_list": [{"slice": [0, 1, None]}]}))
            ax = plan.add_const(0, "int8")
            cat = plan.new_var(dt, [None] + list(width))
            out_nodes.append(Node("Join", [ax, row, r], [cat], {}))
            full0 = plan.new_var(dt, [None] + list(width))
            out_nodes.append(Node("CumOp", [cat], [full0], {"axis": 0, "mode": "add"}))
            # the caller's buffer holds `store` rows.  store <= T + 1 (the usual case, and what
            # Scan's memory-saving rewrites produce): the LAST `store` rows.  store > T + 1
            # (truncated BPTT): rows beyond T stay zero, scan/op.py:2139-2144.
            n1 = plan.new_var("int64", [])
            out_nodes.append(Node("Shape_i", [full0], [n1], {"i": 0}))
            store = plan.new_var("int64", [])
            out_nodes.append(Node("Shape_i", [init_buf], [store], {"i": 0}))

            def clamp_diff(a_, b_):       # max(a - b, 0) on host integers
                d_ = plan.new_var("int64", [])
                sc = {"n_in": 2, "nodes": [{"op": "sub", "in": [["i", 0], ["i", 1]], "dtype": "int64"},
                                           {"op": "maximum", "in": [["t", 0], ["c", 0, "int64"]], "dtype": "int64"}],
                      "out": [["t", 1]]}
                out_nodes.append(Node("Elemwise", [a_, b_], [d_], {"scalar": sc}))
                return d_
            start, npad = clamp_diff(n1, store), clamp_diff(store, n1)
            wdims = []
            for d_ in range(1, len(width) + 1):
                wv = plan.new_var("int64", [])
                out_nodes.append(Node("Shape_i", [init_buf], [wv], {"i": d_}))
                wdims.append(wv)
            tail = plan.new_var(dt, [None] + list(width))
            out_nodes.append(Node("Alloc", [plan.add_const(0, dt), npad] + wdims, [tail], {}))
            full = plan.new_var(dt, [None] + list(width))
            out_nodes.append(Node("Join", [ax, full0, tail], [full], {}))
            new = plan.new_var(dt, list(V[old].shape))
            out_nodes.append(Node("Subtensor", [full, start], [new], {"idx_list": [{"slice": ["in", None, None]}]}))
            replaced[old] = new
    if not changed:
        return orig
    plan.nodes = [Node(n_.op, [replaced.get(i, i) for i in n_.inputs], list(n_.outputs), n_.params) for n_ in out_nodes]
    plan.outputs = [replaced.get(o, o) for o in plan.outputs]
    return plan


def zero_filled_vars(plan: Plan) -> set:
    """Variables of ``plan`` that are ``Alloc`` of a constant 0 (``zeros_like`` of the front end)."""
    out = set()
    for n_ in plan.nodes:
        if n_.op == "Alloc":
            c = plan.vars[n_.inputs[0]].const
            if c is not None and "data" in c and len(c["data"]) == 1 and float(c["data"][0]) == 0.0:
                out.add(n_.outputs[0])
    return out


def read_last_row_only(plan: Plan) -> set:
    """Variables of ``plan`` whose every use is ``v[-1]`` (what the caller of a Scan accumulator reads)."""
    uses = {}
    for n_ in plan.nodes:
        for k, i in enumerate(n_.inputs):
            uses.setdefault(i, []).append(n_.op == "Subtensor" and k == 0 and n_.params.get("idx_list") == [{"index": -1}])
    return {v for v, u in uses.items() if all(u) and v not in plan.outputs}


def push_out_product_accumulators(plan: Plan, need_one_row: list = None, need_equal: list = None,
                                  zeros: set = None, last_only: set = None, need_le2: list = None) -> Plan:
    """The weight gradient a gradient Scan accumulates IN the loop — ``acc_t = acc_{t-1} + a_t^T @ g_t``
    (``Gemm(acc, alpha, A_t, B_t, 1)``; for a vector state the outer product ``Ger(acc, alpha, x_t,
    y_t)``) in a sit-sot output — is what the reference's ``PushOutDot1`` (scan/rewriting.py) turns
    into ONE product over the stacked operands behind the loop.  That rewrite matches the gradient
    of a one-tap recurrence; for taps [-1, -2] (mit-mot [0, 2, 1] -> [2, 1]) it leaves T small
    products inside the step, which also keeps the step off the persistent kernels.  Done here for
    every such accumulator whose buffer holds ONE row (only the final sum is kept: what Scan's
    memory-saving rewrite leaves when the caller reads ``acc[-1]``):
    ``acc_T = acc_0 + alpha * sum_t A_t @ B_t = acc_0 + alpha * [A_1 .. A_T] @ [B_1; ..; B_T]``,
    the per-step operands taken from the sequences the Scan already receives or handed out as one
    more nit-sot output.  Same sum in another association (one long-K GEMM): equal to the loop's up
    to float reordering, like PushOutDot1's.

    The buffer length is rarely static (``Alloc`` of a shape expression): with ``need_one_row`` (a
    list) the static test is skipped and the variables that must turn out to hold one row are
    appended to it — ``ScanMixin._op_Scan`` applies the rewrite to the single Scan node at hand
    and runs the rewritten form when the buffers it is handed have one row.  ``need_equal``
    collects pairs of host integers that must be equal then: a per-step operand that already IS a
    nit-sot output is taken from there when that output keeps all ``n_steps`` rows."""
    def cval(pl, vid):
        v = pl.vars[vid]
        if v.const is not None and len(v.const.get("data", ())) == 1:
            return float(v.const["data"][0])
        return None

    orig = plan
    plan = Plan(plan.name, dict(plan.vars), list(plan.inputs), list(plan.outputs), list(plan.nodes))
    out_nodes, replaced, changed = [], {}, False
    for node in plan.nodes:
        if node.op != "Scan" or node.params.get("as_while") or not node.params.get("sit_sot_in_slices"):
            out_nodes.append(node)
            continue
        p = dict(node.params)
        inner = p["inner"]
        n_seqs = p["n_seqs"]
        mm_in = [list(t) for t in p.get("mit_mot_in_slices", [])]
        mm_out = [list(t) for t in p.get("mit_mot_out_slices", [])]
        ms = [list(t) for t in p["mit_sot_in_slices"]]
        ss = [list(t) for t in p["sit_sot_in_slices"]]
        n_mm, n_ms, n_ss, n_nit = len(mm_in), len(ms), len(ss), p["n_nit_sot"]
        n_sh = p.get("n_shared_outs", 0)
        tap0 = n_seqs + sum(len(t) for t in mm_in) + sum(len(t) for t in ms)      # first sit-sot tap
        out0 = sum(len(t) for t in mm_out) + n_ms                                  # first sit-sot output
        nit0 = out0 + n_ss
        base_in = 1 + n_seqs + n_mm + n_ms            # position of the first sit-sot init in node.inputs
        base_out = n_mm + n_ms
        iprod = {o: n_ for n_ in inner.nodes for o in n_.outputs}
        iclients = inner.clients()
        V = plan.vars
        found = []         # (q, alpha, "gemm" | "ger", a var, b var)
        zero_set = zero_filled_vars(plan) if zeros is None else zeros
        last_set = read_last_row_only(plan) if last_only is None else last_only
        n_var_ = len(inner.inputs) - p["n_non_seqs"]
        inner_zero = {iv for iv, ov in zip(inner.inputs[n_var_:], node.inputs[len(node.inputs) - p["n_non_seqs"]:])
                      if ov in zero_set} if p["n_non_seqs"] else set()

        def is_acc(v, tq_):
            """v is the accumulator tap itself, or ``tap + zeros`` (what the gradient of a lookup
            leaves: acc + inc_subtensor(zeros[i], g) rewritten to inc_subtensor((acc + zeros)[i], g))"""
            if v == tq_:
                return True
            pn_ = iprod.get(v)
            if pn_ is None or pn_.op != "Elemwise" or len(pn_.inputs) != 2 or len(iclients[v]) != 1:
                return False
            sc_ = pn_.params["scalar"]
            return (tq_ in pn_.inputs and any(i in inner_zero for i in pn_.inputs) and len(sc_["nodes"]) == 1
                    and sc_["nodes"][0]["op"] == "add"
                    and sorted(map(tuple, sc_["nodes"][0]["in"])) == [("i", 0), ("i", 1)] and sc_["out"] == [["t", 0]]
                    and inner.vars[v].shape == inner.vars[tq_].shape)
        for q in range(n_ss):
            if ss[q] != [-1]:
                continue
            tq, oq = inner.inputs[tap0 + q], inner.outputs[out0 + q]
            upd = iprod.get(oq)
            init_buf = node.inputs[base_in + q]
            scatter = upd is not None and upd.op in ("IncSubtensor", "AdvancedIncSubtensor1")
            if upd is None or len(iclients[tq]) != 1 or len(iclients[oq]) != 1 or \
                    inner.vars[oq].dtype not in ("float32", "float64"):
                continue
            if scatter and node.outputs[base_out + q] not in last_set and need_le2 is None:
                continue          # (a scatter accumulator: only its LAST row may be read, or — decided by
                #                    the caller at run time — the buffer holds at most two rows)
            if not scatter and need_one_row is None and V[init_buf].shape[0] != 1:
                continue
            if upd.op == "Gemm" and upd.inputs[0] == tq and cval(inner, upd.inputs[1]) is not None and \
                    cval(inner, upd.inputs[4]) == 1.0 and tq not in upd.inputs[1:]:
                found.append((q, cval(inner, upd.inputs[1]), "gemm", upd.inputs[2], upd.inputs[3]))
            elif upd.op == "Ger" and upd.inputs[0] == tq and cval(inner, upd.inputs[1]) is not None and \
                    tq not in upd.inputs[1:]:
                found.append((q, cval(inner, upd.inputs[1]), "ger", upd.inputs[2], upd.inputs[3]))
            elif upd.op == "IncSubtensor" and is_acc(upd.inputs[0], tq) and not upd.params["set_instead_of_inc"] and \
                    upd.params["idx_list"] == [{"index": "in"}] and len(upd.inputs) == 3 and tq not in upd.inputs[1:] \
                    and inner.vars[upd.inputs[1]].ndim == inner.vars[tq].ndim - 1:
                # acc[i_t] += v_t (the gradient of a table whose rows the step looks up itself,
                # subtensor.py:1419 IncSubtensor): acc_T = acc_0 with rows v_1 .. v_T added at i_1 .. i_T
                found.append((q, 1.0, "scatter", upd.inputs[1], upd.inputs[2]))
            elif upd.op == "AdvancedIncSubtensor1" and is_acc(upd.inputs[0], tq) and not upd.params["set_instead_of_inc"] \
                    and tq not in upd.inputs[1:] and inner.vars[upd.inputs[2]].ndim == 1 and \
                    inner.vars[upd.inputs[1]].ndim == inner.vars[tq].ndim:
                found.append((q, 1.0, "scatter1", upd.inputs[1], upd.inputs[2]))
        if not found:
            out_nodes.append(node)
            continue
        # where the per-step operands come from: a sequence the Scan receives (rows 0 .. n_steps - 1
        # of it), or a value of the step, handed out as one more nit-sot output of n_steps rows
        n_steps_v = node.inputs[0]
        new_nits = []                  # inner vars handed out additionally
        post = []                      # nodes behind the Scan (appended after it)
        new_nit_outs = {}              # inner var -> outer var of its rows

        def rows_of(v):
            """outer variable holding [T, ...] = the value of inner variable v at every step (None:
            not a per-step value we can take out)"""
            pn = iprod.get(v)
            if pn is not None and pn.op in ("ScalarFromTensor", "TensorFromScalar"):
                return rows_of(pn.inputs[0])
            if pn is not None and pn.op == "DimShuffle":
                r = rows_of(pn.inputs[0])
                if r is None:
                    return None
                order = [0] + [(o if o == "x" else o + 1) for o in pn.params["new_order"]]
                o_ = plan.new_var(inner.vars[v].dtype, [None] + list(inner.vars[v].shape))
                post.append(Node("DimShuffle", [r], [o_], {"new_order": order}))
                return o_
            if v in inner.inputs:
                k = inner.inputs.index(v)
                if k >= n_seqs:
                    return None        # a tap or an invariant: not a stack of per-step rows
                o_ = plan.new_var(inner.vars[v].dtype, [None] + list(inner.vars[v].shape))
                post.append(Node("Subtensor", [node.inputs[1 + k], n_steps_v], [o_],
                                 {"idx_list": [{"slice": [None, "in", None]}]}))
                return o_
            if pn is None:
                return None            # a constant
            nits_ = inner.outputs[nit0:nit0 + n_nit]
            if v in nits_ and need_equal is not None:
                j_ = nits_.index(v)
                len_v = node.inputs[1 + n_seqs + n_mm + n_ms + n_ss + n_sh + j_]
                if len_v != n_steps_v:
                    need_equal.append((len_v, n_steps_v))
                return node.outputs[n_mm + n_ms + n_ss + j_]
            if need_equal is not None:
                # ... or handed out with broadcast axes added (a vector as a [1, n] row): the same rows
                for j_, o in enumerate(nits_):
                    on = iprod.get(o)
                    if on is not None and on.op == "DimShuffle" and on.inputs[0] == v and \
                            [d for d in on.params["new_order"] if d != "x"] == list(range(inner.vars[v].ndim)):
                        len_v = node.inputs[1 + n_seqs + n_mm + n_ms + n_ss + n_sh + j_]
                        if len_v != n_steps_v:
                            need_equal.append((len_v, n_steps_v))
                        o_ = plan.new_var(inner.vars[v].dtype, [None] + list(inner.vars[v].shape))
                        keep = [0] + [1 + k for k, d in enumerate(on.params["new_order"]) if d != "x"]
                        post.append(Node("DimShuffle", [node.outputs[n_mm + n_ms + n_ss + j_]], [o_],
                                         {"new_order": keep}))
                        return o_
            if v not in new_nit_outs:
                new_nits.append(v)
                new_nit_outs[v] = plan.new_var(inner.vars[v].dtype, [None] + list(inner.vars[v].shape))
            return new_nit_outs[v]

        done = []
        for q, alpha, kind, a, b in found:
            mark = (len(post), len(new_nits), dict(new_nit_outs))
            ra, rb = rows_of(a), rows_of(b)
            if ra is None or rb is None:
                del post[mark[0]:]
                del new_nits[mark[1]:]
                new_nit_outs.clear()
                new_nit_outs.update(mark[2])
                continue
            old = node.outputs[base_out + q]
            init_buf = node.inputs[base_in + q]
            dt = V[old].dtype
            if kind in ("scatter", "scatter1"):
                # ra: the added rows [T, ...] ([T, B, ...]), rb: their indices [T] ([T, B])
                neg = plan.add_const(-1, "int64")
                base = plan.new_var(dt, list(V[init_buf].shape[1:]))
                post.append(Node("Subtensor", [init_buf], [base], {"idx_list": [{"index": 0}]}))

                def scatter_sum(rv, iv):
                    """acc_0 with the rows ``rv`` added at ``iv`` (in step order per destination row)"""
                    if kind == "scatter1":
                        dims = [neg] + [plan.new_var("int64", []) for _ in range(V[init_buf].ndim - 2)]
                        for d_, dv in enumerate(dims[1:]):
                            post.append(Node("Shape_i", [init_buf], [dv], {"i": d_ + 2}))
                        sv, si = plan.new_var("int64", [len(dims)]), plan.new_var("int64", [1])
                        post.append(Node("MakeVector", dims, [sv], {"dtype": "int64"}))
                        post.append(Node("MakeVector", [neg], [si], {"dtype": "int64"}))
                        rv2, iv2 = plan.new_var(dt, [None] * len(dims)), plan.new_var(inner.vars[b].dtype, [None])
                        post.append(Node("Reshape", [rv, sv], [rv2], {"ndim": len(dims)}))
                        post.append(Node("Reshape", [iv, si], [iv2], {"ndim": 1}))
                        rv, iv = rv2, iv2
                    out_ = plan.new_var(dt, list(V[init_buf].shape[1:]))
                    post.append(Node("AdvancedIncSubtensor1", [base, rv, iv], [out_],
                                     {"set_instead_of_inc": False, "inplace": False}))
                    return out_
                summed = scatter_sum(ra, rb)
                buf = init_buf
                if old not in last_set:
                    # a buffer of TWO rows keeps [acc_{T-1}, acc_T] (chronological, scan/op.py:2105): the row
                    # before the last is the same sum over the first T - 1 steps
                    if need_le2 is None:
                        del post[mark[0]:]
                        continue
                    tm1 = plan.new_var("int64", [])
                    post.append(Node("Elemwise", [n_steps_v, plan.add_const(1, "int64")], [tm1], {"scalar": {
                        "n_in": 2, "nodes": [{"op": "sub", "in": [["i", 0], ["i", 1]], "dtype": "int64"}], "out": [["t", 0]]}}))
                    cut = []
                    for v_ in (ra, rb):
                        c_ = plan.new_var(plan.vars[v_].dtype, list(plan.vars[v_].shape))
                        post.append(Node("Subtensor", [v_, tm1], [c_], {"idx_list": [{"slice": [None, "in", None]}]}))
                        cut.append(c_)
                    prev = scatter_sum(*cut)
                    # [acc_{T-1}; acc_T] written once, cut to the caller's one or two rows (a view): the
                    # sums of a large table are not copied through the buffer twice
                    both = []
                    for v_ in (prev, summed):
                        r3 = plan.new_var(dt, [1] + list(V[init_buf].shape[1:]))
                        post.append(Node("DimShuffle", [v_], [r3], {"new_order": ["x"] + list(range(V[init_buf].ndim - 1))}))
                        both.append(r3)
                    two = plan.new_var(dt, [2] + list(V[init_buf].shape[1:]))
                    post.append(Node("Join", [plan.add_const(0, "int64")] + both, [two], {}))
                    s_, ms_ = plan.new_var("int64", []), plan.new_var("int64", [])
                    post.append(Node("Shape_i", [init_buf], [s_], {"i": 0}))
                    post.append(Node("Elemwise", [s_], [ms_], {"scalar": {
                        "n_in": 1, "nodes": [{"op": "neg", "in": [["i", 0]], "dtype": "int64"}], "out": [["t", 0]]}}))
                    new = plan.new_var(dt, list(V[old].shape))
                    post.append(Node("Subtensor", [two, ms_], [new], {"idx_list": [{"slice": ["in", None, None]}]}))
                    need_le2.append(init_buf)
                    replaced[old] = new
                    done.append(q)
                    continue
                # the buffer the caller allocated, its last row = the final sum
                new = plan.new_var(dt, list(V[old].shape))
                post.append(Node("IncSubtensor", [buf, summed], [new],
                                 {"idx_list": [{"index": -1}], "set_instead_of_inc": True, "inplace": False}))
                replaced[old] = new
                done.append(q)
                continue
            if kind == "gemm":
                # sum_t A_t[m, k] @ B_t[k, n] = [m, T*k] @ [T*k, n]
                at_ = plan.new_var(dt, [None, None, None])
                post.append(Node("DimShuffle", [ra], [at_], {"new_order": [1, 0, 2]}))
                m_ = plan.new_var("int64", [])
                post.append(Node("Shape_i", [init_buf], [m_], {"i": 1}))
                n_ = plan.new_var("int64", [])
                post.append(Node("Shape_i", [init_buf], [n_], {"i": 2}))
                neg = plan.add_const(-1, "int64")
                sa, sb = plan.new_var("int64", [2]), plan.new_var("int64", [2])
                post.append(Node("MakeVector", [m_, neg], [sa], {"dtype": "int64"}))
                post.append(Node("MakeVector", [neg, n_], [sb], {"dtype": "int64"}))
                a2, b2 = plan.new_var(dt, [None, None]), plan.new_var(dt, [None, None])
                post.append(Node("Reshape", [at_, sa], [a2], {"ndim": 2}))
                post.append(Node("Reshape", [rb, sb], [b2], {"ndim": 2}))
            else:
                # sum_t outer(x_t, y_t) = X[T, m]^T @ Y[T, n]
                a2 = plan.new_var(dt, [None, None])
                post.append(Node("DimShuffle", [ra], [a2], {"new_order": [1, 0]}))
                b2 = rb
            prod = plan.new_var(dt, [None, None])
            post.append(Node("Dot22", [a2, b2], [prod], {}))
            prod3 = plan.new_var(dt, [1, None, None])
            post.append(Node("DimShuffle", [prod], [prod3], {"new_order": ["x", 0, 1]}))
            new = plan.new_var(dt, list(V[old].shape))
            sc = {"n_in": 2, "nodes": [{"op": "mul", "in": [["c", alpha, dt], ["i", 1]], "dtype": dt},
                                       {"op": "add", "in": [["i", 0], ["t", 0]], "dtype": dt}],
                  "out": [["t", 1]]}
            post.append(Node("Elemwise", [init_buf, prod3], [new], {"scalar": sc}))
            replaced[old] = new
            done.append(q)
            if need_one_row is not None:
                need_one_row.append(init_buf)
        if not done:
            out_nodes.append(node)
            continue
        changed = True
        dq = set(done)
        new_inputs = [v for k, v in enumerate(node.inputs) if not (base_in <= k < base_in + n_ss and k - base_in in dq)]
        new_outputs = [v for k, v in enumerate(node.outputs) if not (base_out <= k < base_out + n_ss and k - base_out in dq)]
        in_keep = [v for k, v in enumerate(inner.inputs) if not (tap0 <= k < tap0 + n_ss and k - tap0 in dq)]
        out_keep = [v for k, v in enumerate(inner.outputs) if not (out0 <= k < out0 + n_ss and k - out0 in dq)]
        # the additional nit-sot outputs: behind the existing ones (inner outputs, Scan outputs) and
        # their lengths behind the existing lengths (Scan inputs: ..., shared, nit-sot lengths, non-seqs)
        n_ss2 = n_ss - len(dq)
        o_nit_end = sum(len(t) for t in mm_out) + n_ms + n_ss2 + n_nit
        out_keep = out_keep[:o_nit_end] + new_nits + out_keep[o_nit_end:]
        so_nit_end = n_mm + n_ms + n_ss2 + n_nit
        new_outputs = new_outputs[:so_nit_end] + [new_nit_outs[v] for v in new_nits] + new_outputs[so_nit_end:]
        si_nit_end = 1 + n_seqs + n_mm + n_ms + n_ss2 + n_sh + n_nit
        new_inputs = new_inputs[:si_nit_end] + [n_steps_v] * len(new_nits) + new_inputs[si_nit_end:]
        gone = {id(iprod[inner.outputs[out0 + q]]) for q in dq}        # the accumulating products themselves
        new_inner = Plan(inner.name + "_noprod", inner.vars, in_keep, out_keep,
                         [n_ for n_ in inner.nodes if id(n_) not in gone])
        new_inner.nodes = _prune_dead(new_inner, list(new_inner.nodes))
        p["inner"] = new_inner
        p["sit_sot_in_slices"] = [t for q, t in enumerate(ss) if q not in dq]
        p["n_nit_sot"] = n_nit + len(new_nits)
        if new_outputs:          # (every output was an accumulator over sequence rows: no loop is left)
            out_nodes.append(Node("Scan", new_inputs, new_outputs, p))
        out_nodes.extend(post)
    if not changed:
        return orig
    plan.nodes = [Node(n_.op, [replaced.get(i, i) for i in n_.inputs], list(n_.outputs), n_.params) for n_ in out_nodes]
    plan.outputs = [replaced.get(o, o) for o in plan.outputs]
    return plan


def merge_shared_left_dots(plan: Plan) -> Plan:
    """Several ``Dot22(x, W_k)`` with the SAME left operand and plan-input right operands (the
    ``x_t @ W_gate`` products of every gate, lifted over a whole sequence: ``[T*B, K] @ [K, H]``
    three times) become ONE product against the weights side by side —
    ``P = Dot22(x, Join(1, W_1, .., W_G))``, each result a column view ``P[:, c_{k-1}:c_k]`` (no
    copies of P; the 12 MiB of weights are joined once per call).  A wider N keeps more of the chip
    busy per tile wave and saves G - 1 launch boundaries.  Returns the plan unchanged when no group
    of >= 2 such products exists."""
    groups: Dict[int, list] = {}
    ins = set(plan.inputs)
    for ni, n in enumerate(plan.nodes):
        if n.op == "Dot22" and n.inputs[1] in ins and plan.vars[n.inputs[1]].ndim == 2:
            groups.setdefault(n.inputs[0], []).append(ni)
    groups = {x: g for x, g in groups.items() if len(g) >= 2 and
              len({plan.vars[plan.nodes[ni].inputs[1]].dtype for ni in g}) == 1}
    if not groups:
        return plan
    work = Plan(plan.name + "_wide", dict(plan.vars), list(plan.inputs), list(plan.outputs), [])
    V = work.vars
    drop, emitted_at = set(), {}
    for x, g in groups.items():
        first = min(g)
        dt = V[plan.nodes[first].outputs[0]].dtype
        Ws = [plan.nodes[ni].inputs[1] for ni in g]
        new = []
        wcat = work.new_var(dt, [None, None])
        new.append(Node("Join", [work.add_const(1, "int8")] + Ws, [wcat], {}))
        P = work.new_var(dt, [None, None])
        new.append(Node("Dot22", [x, wcat], [P], {}))
        lo = None
        for k, ni in enumerate(g):
            nk = work.new_var("int64", [])
            new.append(Node("Shape_i", [Ws[k]], [nk], {"i": 1}))
            if lo is None:
                hi = nk
            else:
                hi = work.new_var("int64", [])
                new.append(Node("Elemwise", [lo, nk], [hi], {"scalar": {
                    "n_in": 2, "nodes": [{"op": "add", "in": [["i", 0], ["i", 1]], "dtype": "int64"}],
                    "out": [["t", 0]]}}))
            out = plan.nodes[ni].outputs[0]
            idx = [{"slice": [None, None, None]}, {"slice": [None if lo is None else "in", "in", None]}]
            new.append(Node("Subtensor", [P] + ([] if lo is None else [lo]) + [hi], [out], {"idx_list": idx}))
            lo = hi
            drop.add(ni)
        emitted_at[first] = new
    for ni, n in enumerate(plan.nodes):
        if ni in emitted_at:
            work.nodes.extend(emitted_at[ni])
        if ni not in drop:
            work.nodes.append(n)
    return work


def split_assembled_columns(plan: Plan) -> Plan:
    """The gradient step of a fused-gate recurrence assembles the gate gradients into ONE wide
    value — ``X = zeros(B, 4H); X[:, 0:H] = p_i; X[:, H:2H] = p_f; ...`` (a vector of length 4H for
    a vector state) — only to multiply it with the (transposed) fused weight matrix and to hand it
    out as a nit-sot output.  The product is a sum over the blocks, ``z + X @ W = z + sum_k p_k @
    W[a_k:b_k, :]`` (``z + A x = z + sum_k A[:, a_k:b_k] p_k`` for a Gemv), and the output is the
    blocks side by side, so the assembly leaves the step: the Scan returns the blocks as separate
    nit-sot outputs (H-wide, what the persistent kernels write) and ONE ``Join`` after the loop
    rebuilds the wide array for the weight-gradient GEMMs."""
    def cint(pl, vid):
        v = pl.vars[vid]
        if v.const is not None and len(v.const.get("data", ())) == 1 and v.dtype.startswith("int"):
            return int(v.const["data"][0])
        return None

    orig, out_nodes, replaced, changed = plan, [], {}, False
    plan = Plan(plan.name, dict(plan.vars), list(plan.inputs), list(plan.outputs), list(plan.nodes))
    for node in plan.nodes:
        if node.op != "Scan" or node.params.get("as_while") or not node.params.get("n_nit_sot"):
            out_nodes.append(node)
            continue
        p = dict(node.params)
        inner = p["inner"]
        n_seqs, n_nit, n_sh = p["n_seqs"], p["n_nit_sot"], p.get("n_shared_outs", 0)
        mm_out = [list(t) for t in p.get("mit_mot_out_slices", [])]
        n_mm, n_ms, n_ss = len(mm_out), len(p["mit_sot_in_slices"]), len(p["sit_sot_in_slices"])
        nit0 = sum(len(t) for t in mm_out) + n_ms + n_ss
        n_var = len(inner.inputs) - p["n_non_seqs"]
        inv = set(inner.inputs[n_var:])
        for n_ in inner.nodes:
            if n_.op in ("DimShuffle", "ViewOp") and all(i in inv for i in n_.inputs):
                inv.update(n_.outputs)
        iprod = {o: n_ for n_ in inner.nodes for o in n_.outputs}
        icl = inner.clients()
        found = None
        for j in range(n_nit):
            o = inner.outputs[nit0 + j]
            X, view = o, None
            if o in iprod and iprod[o].op == "DimShuffle" and \
                    [d for d in iprod[o].params["new_order"] if d != "x"] == list(range(inner.vars[iprod[o].inputs[0]].ndim)):
                X, view = iprod[o].inputs[0], iprod[o]       # handed out with broadcast axes added
            nd = inner.vars[X].ndim
            if nd not in (1, 2) or inner.outputs.count(o) != 1 or (view is not None and len(icl[o]) != 1):
                continue
            parts, v = [], X
            while v in iprod and iprod[v].op == "IncSubtensor":
                n_ = iprod[v]
                idx = n_.params["idx_list"]
                if len(idx) != nd or any("slice" not in e for e in idx) or \
                        (nd == 2 and idx[0]["slice"] != [None, None, None]) or idx[-1]["slice"][2] not in (None, 1):
                    parts = None
                    break
                extra, ab = list(n_.inputs[2:]), []
                for e in idx[-1]["slice"][:2]:
                    ab.append(cint(inner, extra.pop(0)) if e == "in" and extra else (e if isinstance(e, int) else None))
                if None in ab or extra:
                    parts = None
                    break
                parts.append((ab[0], ab[1], n_.inputs[1]))
                v = n_.inputs[0]
                if len(icl[v]) != 1:
                    parts = None
                    break
            if not parts or v not in iprod or iprod[v].op != "Alloc":
                continue
            fill = inner.vars[iprod[v].inputs[0]]
            if fill.const is None or any(float(d) != 0.0 for d in fill.const.get("data", [1])):
                continue
            parts.sort()
            if parts[0][0] != 0 or any(parts[k][1] != parts[k + 1][0] for k in range(len(parts) - 1)):
                continue
            # X is read by exactly two things: the product and the output (or the view handed out)
            uses = [c for c in icl[X] if c[0] != "out" and inner.nodes[c[0]] is not view]
            if len(uses) != 1 or len(icl[X]) != 2:
                continue
            un = inner.nodes[uses[0][0]]
            if not ((nd == 2 and un.op == "Gemm" and uses[0][1] == 2 and un.inputs[3] in inv) or
                    (nd == 1 and un.op == "Gemv" and uses[0][1] == 3 and un.inputs[2] in inv)):
                continue
            found = (j, X, view, parts, un, nd)
            break
        if found is None:
            out_nodes.append(node)
            continue
        changed = True
        j, X, view, parts, un, nd = found
        work = Plan(inner.name + "_blocks", dict(inner.vars), list(inner.inputs), list(inner.outputs), [])
        if nd == 2:
            z, al, _x, W, be = un.inputs
        else:
            z, al, W, _x, be = un.inputs
        one = work.add_const(1.0, inner.vars[al].dtype)
        nodes = []
        for n_ in inner.nodes:
            if n_ is not un:
                nodes.append(n_)
                continue
            acc = z
            for k, (a, b, pk) in enumerate(parts):
                wv = inner.vars[W]
                if nd == 2:      # rows a:b of W
                    wk = work.new_var(wv.dtype, [None, wv.shape[1]])
                    nodes.append(Node("Subtensor", [W], [wk], {"idx_list": [{"slice": [a, b, None]}]}))
                else:            # columns a:b of A
                    wk = work.new_var(wv.dtype, [wv.shape[0], None])
                    nodes.append(Node("Subtensor", [W], [wk], {"idx_list": [{"slice": [None, None, None]},
                                                                            {"slice": [a, b, None]}]}))
                out = un.outputs[0] if k == len(parts) - 1 else \
                    work.new_var(inner.vars[un.outputs[0]].dtype, list(inner.vars[un.outputs[0]].shape))
                ins_ = [acc, al, pk, wk, be if k == 0 else one] if nd == 2 else \
                    [acc, al, wk, pk, be if k == 0 else one]
                nodes.append(Node(un.op, ins_, [out], dict(un.params)))
                acc = out
        work.outputs = inner.outputs[:nit0 + j] + [pk for _a, _b, pk in parts] + inner.outputs[nit0 + j + 1:]
        work.nodes = _prune_dead(work, nodes)
        p["inner"] = work
        p["n_nit_sot"] = n_nit + len(parts) - 1
        len0 = 1 + n_seqs + n_mm + n_ms + n_ss + n_sh
        new_inputs = list(node.inputs[:len0 + j]) + [node.inputs[len0 + j]] * len(parts) + list(node.inputs[len0 + j + 1:])
        o0 = n_mm + n_ms + n_ss
        old = node.outputs[o0 + j]
        ov = plan.vars[old]
        xv = inner.vars[X]
        pshape = [None] + list(xv.shape[:-1]) + [None]               # [T, (B,) width]
        pouts = [plan.new_var(ov.dtype, pshape) for _ in parts]
        new_outputs = list(node.outputs[:o0 + j]) + pouts + list(node.outputs[o0 + j + 1:])
        out_nodes.append(Node("Scan", new_inputs, new_outputs, p))
        ax = plan.add_const(nd, "int8")
        if view is None:
            whole = plan.new_var(ov.dtype, list(ov.shape))
            out_nodes.append(Node("Join", [ax] + pouts, [whole], {}))
        else:
            wide = plan.new_var(ov.dtype, pshape)
            out_nodes.append(Node("Join", [ax] + pouts, [wide], {}))
            whole = plan.new_var(ov.dtype, list(ov.shape))
            order = [0] + [d if d == "x" else d + 1 for d in view.params["new_order"]]
            out_nodes.append(Node("DimShuffle", [wide], [whole], {"new_order": order}))
        replaced[old] = whole
    if not changed:
        return orig
    plan.nodes = [Node(n_.op, [replaced.get(i, i) for i in n_.inputs], list(n_.outputs), n_.params) for n_ in out_nodes]
    plan.outputs = [replaced.get(o, o) for o in plan.outputs]
    return plan


def _prune_dead(plan: Plan, keep: List[Node]) -> List[Node]:
    pure = {"AllocEmpty", "Shape_i", "Shape", "DimShuffle", "Elemwise", "ScalarFromTensor",
            "TensorFromScalar", "MakeVector", "Alloc", "ViewOp", "CAReduce", "Subtensor", "IncSubtensor",
            "Assert"}      # (an Assert nobody reads guards nothing that is still computed)
    while True:
        read = {i for n in keep for i in n.inputs} | set(plan.outputs)
        dead = [n for n in keep if n.op in pure and not any(o in read for o in n.outputs)]
        if not dead:
            return keep
        dead_ids = {id(n) for n in dead}
        keep = [n for n in keep if id(n) not in dead_ids]


def _hoist_sequence_only_stacked(plan: Plan, seq_inputs: List[int], invariant: set):
    """The matrix-state class of :func:`hoist_sequence_only` (a batch of recurrences: per-step
    values are ``[B, n]`` matrices).  The steps are stacked along the rows — a sequence ``[T, B, n]``
    is read as ``[T * B, n]`` — so the hoisted nodes run unchanged on the stacked operands:
    ``Elemwise`` (invariant operands must broadcast along the rows: static leading extent 1),
    ``Dot22`` / ``Dot22Scalar`` / ``Gemm`` with an invariant right-hand matrix.  ``lifted["stacked"]``
    tells the executor to fold / unfold the two leading axes."""
    V = plan.vars
    floats = ("float32", "float64")

    def is_const_scalar(vid):
        v = V[vid]
        return v.const is not None and len(v.const.get("data", ())) == 1

    Q = {v for v in seq_inputs if V[v].ndim == 2 and V[v].dtype in floats}
    seq_set = set(seq_inputs)
    hoisted, keep = [], []
    for n in plan.nodes:
        ok = False
        if n.op == "Elemwise" and n.inputs and any(i in Q for i in n.inputs):
            ok = all(i in Q or ((i in invariant or V[i].const is not None) and V[i].ndim == 2
                                and V[i].shape[0] == 1) for i in n.inputs) and \
                all(V[o].ndim == 2 and V[o].dtype in floats for o in n.outputs)
        elif n.op in ("Dot22", "Dot") and len(n.inputs) == 2:
            a, b = n.inputs
            ok = a in Q and b in invariant and V[b].ndim == 2 and V[n.outputs[0]].dtype in floats
        elif n.op == "Dot22Scalar":
            a, b, sc = n.inputs
            ok = a in Q and b in invariant and V[b].ndim == 2 and is_const_scalar(sc)
        elif n.op == "Gemm":
            z, al, a, b, be = n.inputs
            ok = z in Q and a in Q and b in invariant and V[b].ndim == 2 and \
                is_const_scalar(al) and is_const_scalar(be)
        if ok:
            hoisted.append(n)
            Q.update(n.outputs)
        else:
            keep.append(n)
    used_later = {i for n in keep for i in n.inputs} | set(plan.outputs)
    outs = [o for n in hoisted for o in n.outputs if o in used_later]
    if not outs:
        return plan, None
    need, live = set(outs), []
    for n in reversed(hoisted):
        if any(o in need for o in n.outputs):
            live.append(n)
            need.update(n.inputs)
    live.reverse()
    live_ids = {id(n) for n in live}
    keep = _prune_dead(plan, [n for n in plan.nodes if id(n) not in live_ids])
    lp = Plan(plan.name + "_allsteps", {}, [], [], [])
    m, seq_in, inv_in = {}, [], []

    def var(v):
        if v not in m:
            src = V[v]
            m[v] = lp.new_var(src.dtype, list(src.shape), src.name, src.const)
            if src.const is None:
                if v in seq_set:
                    seq_in.append(v)
                elif v in invariant:
                    inv_in.append(v)
        return m[v]

    for n in live:
        lp.nodes.append(Node("Dot22" if n.op == "Dot" else n.op, [var(i) for i in n.inputs],
                             [var(o) for o in n.outputs], dict(n.params)))
    lp.inputs = [m[v] for v in seq_in] + [m[v] for v in inv_in]
    lp.outputs = [m[o] for o in outs]
    loop = Plan(plan.name + "_seqonly", plan.vars, list(plan.inputs) + outs, list(plan.outputs), keep)
    return loop, {"plan": lp, "seq_in": seq_in, "inv_in": inv_in, "outs": outs, "stacked": True}


def hoist_sequence_only(plan: Plan, seq_inputs: List[int], invariant: set):
    """Everything in a Scan step that depends only on *sequence* rows and loop invariants does not
    take part in the recurrence: it is computed for ALL steps before the loop and the step reads
    row t of the result.  Covers the vector class: ``Elemwise`` on per-step vectors and ``Gemv``
    chains with invariant matrices (``W1 x_t + W2 h_t`` of a gate — a gradient Scan recomputes the
    forward gates from stored states this way, scan/op.py:2379 ``Scan.L_op``); the lifted plan runs
    them as ``[T, n]`` Elemwise kernels and GEMMs (T x K @ K x M fills the chip; T separate GEMVs
    cannot).  The reference's ``scan_pushout_seqs_ops`` (scan/rewriting.py) has the same aim but
    leaves the dots, and whatever depends on them, inside.

    Returns ``(loop_plan, lifted)``; ``lifted`` = None or ``{"plan": Plan over whole sequences,
    "seq_in": step-plan ids of the sequences it reads, "inv_in": ids of the invariants it reads,
    "outs": step-plan ids of its results (appended to the loop plan's inputs, in this order)}``."""
    V = plan.vars
    floats = ("float32", "float64")

    def cval(vid):
        v = V[vid]
        if v.const is not None and len(v.const.get("data", ())) == 1:
            return float(v.const["data"][0])
        return None

    Q = {v for v in seq_inputs if V[v].ndim == 1 and V[v].dtype in floats}
    seq_set = set(seq_inputs)
    if not Q and any(V[v].ndim == 2 and V[v].dtype in floats for v in seq_inputs):
        return _hoist_sequence_only_stacked(plan, seq_inputs, invariant)
    hoisted, keep = [], []
    for n in plan.nodes:
        ok = False
        if n.op == "Elemwise" and n.inputs and any(i in Q for i in n.inputs):
            ok = all(i in Q or ((i in invariant or V[i].const is not None) and V[i].ndim == 1)
                     for i in n.inputs) and \
                all(V[o].ndim == 1 and V[o].dtype in floats for o in n.outputs)
        elif n.op == "Gemv":
            y, alpha, A, x, beta = n.inputs
            ok = A in invariant and V[A].ndim == 2 and x in Q and cval(alpha) is not None and \
                cval(beta) is not None and (cval(beta) == 0.0 or y in Q) and \
                V[n.outputs[0]].dtype in floats
        if ok:
            hoisted.append(n)
            Q.update(n.outputs)
        else:
            keep.append(n)
    used_later = {i for n in keep for i in n.inputs} | set(plan.outputs)
    outs = [o for n in hoisted for o in n.outputs if o in used_later]
    if not outs:
        return plan, None
    # keep only the hoisted nodes the results depend on
    need, live = set(outs), []
    for n in reversed(hoisted):
        if any(o in need for o in n.outputs):
            live.append(n)
            need.update(n.inputs)
    live.reverse()
    dropped = [n for n in hoisted if n not in live]
    if dropped:        # (they fed nothing that is kept: leave them where they were)
        live_ids = {id(n) for n in live}
        keep = [n for n in plan.nodes if id(n) not in live_ids]

    # what only fed the hoisted nodes (the AllocEmpty "y" of a beta = 0 Gemv and its shape
    # arithmetic) is dead in the loop now
    keep = _prune_dead(plan, keep)

    lp = Plan(plan.name + "_allsteps", {}, [], [], [])
    m, seq_in, inv_in = {}, [], []

    def same(v):                      # invariant / constant operand: the same variable
        if ("s", v) not in m:
            src = V[v]
            m[("s", v)] = lp.new_var(src.dtype, list(src.shape), src.name, src.const)
            if src.const is None:
                inv_in.append(v)
        return m[("s", v)]

    def rows(v):                      # per-step vector -> [T, n]
        if ("r", v) not in m:
            m[("r", v)] = lp.new_var(V[v].dtype, [None, V[v].shape[0]])
            if v in seq_set:
                seq_in.append(v)
        return m[("r", v)]

    def as_row(v):                    # invariant vector next to [T, n] operands: [1, n]
        if ("b", v) not in m:
            src = V[v]
            if src.const is not None:
                c = dict(src.const)
                c["shape"] = [1] + list(c["shape"])
                m[("b", v)] = lp.new_var(src.dtype, [1] + list(src.shape), None, c)
            else:
                o = lp.new_var(src.dtype, [1] + list(src.shape))
                lp.nodes.append(Node("DimShuffle", [same(v)], [o], {"new_order": ["x", 0]}))
                m[("b", v)] = o
        return m[("b", v)]

    for n in live:
        if n.op == "Elemwise":
            ins = [rows(i) if i in Q else as_row(i) for i in n.inputs]
            lp.nodes.append(Node("Elemwise", ins, [rows(o) for o in n.outputs], dict(n.params)))
        else:
            y, alpha, A, x, beta = n.inputs
            at_ = lp.new_var(V[A].dtype, [V[A].shape[1], V[A].shape[0]])
            lp.nodes.append(Node("DimShuffle", [same(A)], [at_], {"new_order": [1, 0]}))
            if cval(beta) == 0.0:
                lp.nodes.append(Node("Dot22Scalar", [rows(x), at_, same(alpha)], [rows(n.outputs[0])], {}))
            else:
                lp.nodes.append(Node("Gemm", [rows(y), same(alpha), rows(x), at_, same(beta)],
                                     [rows(n.outputs[0])], {"inplace": False}))
    lp.inputs = [m[("r", v)] for v in seq_in] + [m[("s", v)] for v in inv_in]
    lp.outputs = [m[("r", o)] for o in outs]
    loop = Plan(plan.name + "_seqonly", plan.vars, list(plan.inputs) + outs, list(plan.outputs), keep)
    return loop, {"plan": lp, "seq_in": seq_in, "inv_in": inv_in, "outs": outs}


_BATCH_ALIAS = ("ScalarFromTensor", "TensorFromScalar", "ViewOp", "DeepCopyOp")


def batch_map_step(plan: Plan, n_seqs: int):
    """A Scan WITHOUT recurrence (nit-sot outputs only: ``aesara.map``, the row loops of
    ``gradient.jacobian`` / ``hessian`` and of ``Rop`` / ``Lop`` helpers — scan/basic.py:71 with no
    ``outputs_info``) evaluates the same function on every sequence row; no step reads what another
    step wrote (scan_perform.pyx:309-541 with n_mit_mot = n_mit_sot = n_sit_sot = 0), so the loop
    is not a loop: this returns the step plan restated over WHOLE sequences — every per-step
    value ``v[S]`` becomes ``V[T, S]``, invariants stay as they are — or None when a node has no
    such restatement here.  Rules (``step`` = depends on a sequence row):

    * 0-d glue (``ScalarFromTensor`` ...) : the same variable, now ``[T]``;
    * ``Elemwise``: step operands as they are, invariant operands with a leading broadcastable dim;
    * ``CAReduce`` / ``DimShuffle``: axes shifted by one; ``Shape_i`` of a step value: invariant;
    * ``Subtensor(M, i_t, ...)`` of an invariant by the step's index: ``AdvancedSubtensor1(M, i)``
      (the same bounds check and negative wrap, subtensor.py:756 / :2080), then the rest of the index;
      of a step value by invariant indices: one more leading slice;
    * ``AdvancedSubtensor1(M, idx_t)`` with the step's index vector: one gather over all T * B indices;
    * ``IncSubtensor(base, val, i_t)`` (the unit vectors of a Jacobian loop: ``set_subtensor(zeros[i], 1)``):
      ``base`` repeated T times, then ``B[arange(T), i] = val`` (``AdvancedIncSubtensor``; the (t, i_t)
      pairs are distinct); on a step value with invariant indices: one more leading slice;
    * ``Gemv(y, alpha, A, x_t, beta)`` with an invariant matrix: ``X @ A.T`` (``Dot22Scalar`` / ``Gemm``);
    * ``Alloc(v_t, *shape)`` / ``Reshape(v_t, shape)`` / ``SpecifyShape(v_t, *shape)`` with an invariant
      shape: the same node with T in front of the shape.

    Returns ``{"plan": Plan(inputs = sequences + the other step-plan inputs), "outs": n}``."""
    V = plan.vars
    seqs = list(plan.inputs[:n_seqs])
    if not seqs:
        return None
    bp = Plan(plan.name + "_allrows", {}, [], [], [])
    step, m = set(seqs), {}

    def nv(dtype, shape, const=None):
        return bp.new_var(dtype, list(shape), None, const)

    def emit(op, ins, dtype, shape, params=None):
        o = nv(dtype, shape)
        bp.nodes.append(Node(op, list(ins), [o], dict(params or {})))
        return o

    def get(v):
        if v not in m:
            src = V[v]
            m[v] = nv(src.dtype, ([None] + list(src.shape)) if v in step else src.shape, src.const)
        return m[v]

    def lead(v):
        """invariant ``v`` next to ``[T, ...]`` operands"""
        src = V[v]
        if src.const is not None and "data" in src.const:
            c = dict(src.const)
            c["shape"] = [1] + list(c["shape"])
            return nv(src.dtype, [1] + list(src.shape), c)
        return emit("DimShuffle", [get(v)], src.dtype, [1] + list(src.shape),
                    {"new_order": ["x"] + list(range(src.ndim))})

    def i64(x):
        return bp.add_const(np.asarray(x, dtype="int64"), "int64")

    for v in plan.inputs:
        get(v)
    bp.inputs = [m[v] for v in plan.inputs]
    tlen = [None]

    def T_():
        if tlen[0] is None:
            tlen[0] = emit("Shape_i", [m[seqs[0]]], "int64", [], {"i": 0})
        return tlen[0]

    def entries(idx_list, extra):
        """[(entry, its dynamic inputs)]"""
        extra, out = list(extra), []
        for e in idx_list:
            n_dyn = sum(1 for t in (e["slice"] if "slice" in e else [e["index"]]) if t == "in")
            out.append((e, extra[:n_dyn]))
            extra = extra[n_dyn:]
        return out

    full = {"slice": [None, None, None]}
    for n in plan.nodes:
        s_in = [i for i in n.inputs if i in step]
        if not s_in:
            if n.op == "Scan":
                return None
            bp.nodes.append(Node(n.op, [get(i) for i in n.inputs], [get(o) for o in n.outputs], dict(n.params)))
            continue
        o0 = n.outputs[0]
        if n.op in _BATCH_ALIAS and len(n.inputs) == 1:
            step.add(o0)
            m[o0] = get(n.inputs[0])
        elif n.op == "Elemwise":
            step.update(n.outputs)
            ins = [get(i) if i in step else lead(i) for i in n.inputs]
            bp.nodes.append(Node("Elemwise", ins, [get(o) for o in n.outputs], dict(n.params)))
        elif n.op == "CAReduce":
            nd = V[n.inputs[0]].ndim
            ax = n.params.get("axis")
            ax = list(range(nd)) if ax is None else [a % nd for a in ax]
            step.add(o0)
            bp.nodes.append(Node("CAReduce", [get(n.inputs[0])], [get(o0)],
                                 dict(n.params, axis=[a + 1 for a in ax])))
        elif n.op == "DimShuffle":
            step.add(o0)
            bp.nodes.append(Node("DimShuffle", [get(n.inputs[0])], [get(o0)],
                                 dict(n.params, new_order=[0] + [e if e == "x" else e + 1
                                                                 for e in n.params["new_order"]])))
        elif n.op == "Shape_i":
            bp.nodes.append(Node("Shape_i", [get(n.inputs[0])], [get(o0)], {"i": int(n.params["i"]) + 1}))
        elif n.op == "Subtensor":
            x, ents = n.inputs[0], entries(n.params["idx_list"], n.inputs[1:])
            if x in step:
                if any(d in step for _e, dyn in ents for d in dyn):
                    return None
                step.add(o0)
                bp.nodes.append(Node("Subtensor", [get(x)] + [get(d) for _e, dyn in ents for d in dyn],
                                     [get(o0)], {"idx_list": [full] + [e for e, _d in ents]}))
            else:
                e0, d0 = ents[0]
                if "index" not in e0 or len(d0) != 1 or d0[0] not in step or V[d0[0]].ndim != 0 or \
                        any(d in step for _e, dyn in ents[1:] for d in dyn):
                    return None
                step.add(o0)
                rows = emit("AdvancedSubtensor1", [get(x), get(d0[0])], V[x].dtype, [None] + list(V[x].shape[1:]))
                if len(ents) == 1:
                    m[o0] = rows
                else:
                    bp.nodes.append(Node("Subtensor", [rows] + [get(d) for _e, dyn in ents[1:] for d in dyn],
                                         [get(o0)], {"idx_list": [full] + [e for e, _d in ents[1:]]}))
        elif n.op == "AdvancedSubtensor1":
            x, iv = n.inputs
            if x in step or V[iv].ndim != 1:
                return None
            step.add(o0)
            # rows of an invariant table by the step's index VECTOR (an embedding lookup per batch
            # item): one gather over the T * B indices, unfolded to [T, B, ...]
            flat = emit("Reshape", [get(iv), emit("MakeVector", [i64(-1)], "int64", [1], {"dtype": "int64"})],
                        V[iv].dtype, [None], {"ndim": 1})
            rows = emit("AdvancedSubtensor1", [get(x), flat], V[x].dtype, [None] + list(V[x].shape[1:]))
            dims = [T_(), emit("Shape_i", [get(iv)], "int64", [], {"i": 1})] + \
                [emit("Shape_i", [get(x)], "int64", [], {"i": d}) for d in range(1, V[x].ndim)]
            shp = emit("MakeVector", dims, "int64", [len(dims)], {"dtype": "int64"})
            bp.nodes.append(Node("Reshape", [rows, shp], [get(o0)], {"ndim": len(dims)}))
        elif n.op == "IncSubtensor":
            x, y = n.inputs[:2]
            ents = entries(n.params["idx_list"], n.inputs[2:])
            dyn_step = [d for _e, dyn in ents for d in dyn if d in step]
            n_int = sum(1 for e, _d in ents if "index" in e)
            if y in step and V[y].ndim != V[x].ndim - n_int:
                return None          # (a value that broadcasts from the left would misalign under [T, ...])
            step.add(o0)
            if not dyn_step:
                if x not in step:
                    return None
                bp.nodes.append(Node("IncSubtensor", [get(x), get(y)] + [get(d) for _e, dyn in ents for d in dyn],
                                     [get(o0)], dict(n.params, idx_list=[full] + [e for e, _d in ents], inplace=False)))
            else:
                e0, d0 = ents[0]
                if len(ents) != 1 or "index" not in e0 or len(d0) != 1 or V[d0[0]].ndim != 0:
                    return None
                if x in step:
                    B = get(x)
                else:
                    dims = [T_()] + [emit("Shape_i", [get(x)], "int64", [], {"i": d}) for d in range(V[x].ndim)]
                    B = emit("Alloc", [lead(x)] + dims, V[x].dtype, [None] + list(V[x].shape))
                ar = emit("ARange", [i64(0), T_(), i64(1)], "int64", [None], {"dtype": "int64"})
                bp.nodes.append(Node("AdvancedIncSubtensor", [B, get(y), ar, get(d0[0])], [get(o0)],
                                     {"set_instead_of_inc": bool(n.params["set_instead_of_inc"]), "inplace": False}))
        elif n.op == "Gemv":
            y, alpha, A, x, beta = n.inputs
            bv = V[beta]
            beta0 = bv.const is not None and "data" in bv.const and len(bv.const["data"]) == 1 and \
                float(bv.const["data"][0]) == 0.0
            if A in step or alpha in step or beta in step or x not in step or V[A].ndim != 2 or \
                    not (beta0 or y in step):
                return None
            step.add(o0)
            at_ = emit("DimShuffle", [get(A)], V[A].dtype, [V[A].shape[1], V[A].shape[0]], {"new_order": [1, 0]})
            if beta0:
                bp.nodes.append(Node("Dot22Scalar", [get(x), at_, get(alpha)], [get(o0)], {}))
            else:
                bp.nodes.append(Node("Gemm", [get(y), get(alpha), get(x), at_, get(beta)], [get(o0)],
                                     {"inplace": False}))
        elif n.op == "SpecifyShape":
            if any(i in step for i in n.inputs[1:]):
                return None
            step.add(o0)
            nd = V[n.inputs[0]].ndim
            bp.nodes.append(Node("SpecifyShape", [get(i) for i in n.inputs], [get(o0)],
                                 {"dims": [d + 1 for d in n.params.get("dims", range(len(n.inputs) - 1))],
                                  "ndim": int(n.params.get("ndim", nd)) + 1}))
        elif n.op == "Reshape":
            x, sv = n.inputs
            if sv in step or V[sv].dtype != "int64" or V[sv].ndim != 1:
                return None
            step.add(o0)
            t1 = emit("MakeVector", [T_()], "int64", [1], {"dtype": "int64"})
            k = V[sv].shape[0]
            shp = emit("Join", [i64(0), t1, get(sv)], "int64", [None if k is None else k + 1])
            bp.nodes.append(Node("Reshape", [get(x), shp], [get(o0)], {"ndim": int(n.params["ndim"]) + 1}))
        elif n.op == "Alloc":
            if any(i in step for i in n.inputs[1:]) or V[n.inputs[0]].ndim != V[o0].ndim:
                return None
            step.add(o0)
            bp.nodes.append(Node("Alloc", [get(n.inputs[0]), T_()] + [get(i) for i in n.inputs[1:]], [get(o0)],
                                 dict(n.params)))
        else:
            return None
    outs = []
    for o in plan.outputs:
        if o in step:
            outs.append(get(o))
        else:                              # the same value every step
            dims = [T_()] + [emit("Shape_i", [get(o)], "int64", [], {"i": d}) for d in range(V[o].ndim)]
            outs.append(emit("Alloc", [lead(o)] + dims, V[o].dtype, [None] + list(V[o].shape)))
    bp.outputs = outs
    return {"plan": bp, "outs": len(outs)}


_GLUE_TARGETS = ("Subtensor", "AdvancedSubtensor1", "IncSubtensor")
_GLUE_LEAVE = ("Elemwise", "CAReduce", "Gemv", "Dot", "Dot22", "Dot22Scalar", "Gemm")


def push_out_sequence_glue(plan: Plan) -> Plan:
    """Index glue on SEQUENCE rows inside a recurrent step — ``E[idx_t]`` of an embedding table by the
    step's index (a scalar or a vector of a batch), ``set_subtensor(zeros[i_t], 1)``, with the index
    arithmetic that feeds them — does not take part in the recurrence.  Inside the loop it costs a
    host read of the index per step (``Subtensor`` is host-side view arithmetic here) and keeps the
    loop off the persistent kernels; it is taken out as the reference's ``scan_pushout_seqs_ops``
    (scan/rewriting.py) takes Elemwise on sequences out: the glue nodes and their sequence-only
    ancestors are restated over whole sequences (:func:`batch_map_step`: one gather for all steps)
    IN FRONT of the Scan, on the first ``n_steps`` rows of the sequences, and their results enter
    the Scan as additional sequences.  What is left of the step (dots, Elemwise) is what the
    sequence hoists and the persistent kernels already handle."""
    out_nodes, changed = [], False
    new_plan = Plan(plan.name, dict(plan.vars), list(plan.inputs), list(plan.outputs), [])
    for node in plan.nodes:
        p = node.params
        if node.op != "Scan" or p.get("as_while") or not p["n_seqs"] or not (
                p.get("mit_mot_in_slices") or p["mit_sot_in_slices"] or p["sit_sot_in_slices"]
                or p.get("n_shared_outs", 0)):
            # (a Scan without recurrence is restated as a whole, in blocks of rows when its
            # intermediates are large: ScanMixin._scan_all_rows)
            out_nodes.append(node)
            continue
        inner, n_seqs, n_non = p["inner"], p["n_seqs"], p["n_non_seqs"]
        V = inner.vars
        n_var = len(inner.inputs) - n_non
        seq_vars, inv_vars = list(inner.inputs[:n_seqs]), list(inner.inputs[n_var:])
        rec_inputs = set(inner.inputs[n_seqs:n_var])
        prod = {o: n_ for n_ in inner.nodes for o in n_.outputs}
        # 0: invariant / constant, 1: sequence-only, 2: touches the recurrence
        cls = {v: 0 for v in inv_vars}
        cls.update({v: 1 for v in seq_vars})
        cls.update({v: 2 for v in rec_inputs})
        for n_ in inner.nodes:
            k = max([cls.get(i, 0) for i in n_.inputs] or [0])
            if n_.op == "Scan":
                k = 2
            for o in n_.outputs:
                cls[o] = k
        targets = [n_ for n_ in inner.nodes if n_.op in _GLUE_TARGETS and cls[n_.outputs[0]] == 1]
        if not targets:
            out_nodes.append(node)
            continue
        take, stack = set(), list(targets)
        while stack:                      # the targets and everything (not recurrent) they are computed from
            n_ = stack.pop()
            if id(n_) in take:
                continue
            take.add(id(n_))
            stack.extend(prod[i] for i in n_.inputs if i in prod)
        sub_nodes = [n_ for n_ in inner.nodes if id(n_) in take]
        rest = [n_ for n_ in inner.nodes if id(n_) not in take]
        if any(n_.op in _GLUE_TARGETS + ("ScalarFromTensor", "AdvancedIncSubtensor1", "Scan") for n_ in rest):
            # index nodes that DO depend on the recurrence stay: the loop keeps its host reads and
            # its place on the launch list whatever leaves it — nothing to gain
            out_nodes.append(node)
            continue
        used = {i for n_ in rest for i in n_.inputs} | set(inner.outputs)
        outs = [o for n_ in sub_nodes for o in n_.outputs if o in used and cls[o] == 1]
        if not outs or any(o in used and cls[o] == 0 for n_ in sub_nodes for o in n_.outputs):
            out_nodes.append(node)        # (an invariant intermediate is shared with the step: leave it)
            continue
        sub = Plan(inner.name + "_glue", dict(V), seq_vars + inv_vars, list(outs), sub_nodes)
        r = batch_map_step(sub, n_seqs)
        if r is None:
            out_nodes.append(node)
            continue
        bp = r["plan"]
        # splice the whole-sequence plan in front of the Scan
        n_steps_v = node.inputs[0]
        o_seqs = list(node.inputs[1:1 + n_seqs])
        o_invs = list(node.inputs[len(node.inputs) - n_non:]) if n_non else []
        m = {}
        for bv, ov in zip(bp.inputs[:n_seqs], o_seqs):
            src = new_plan.vars[ov]
            cut = new_plan.new_var(src.dtype, [None] + list(src.shape[1:]))
            out_nodes.append(Node("Subtensor", [ov, n_steps_v], [cut], {"idx_list": [{"slice": [None, "in", None]}]}))
            m[bv] = cut
        for bv, ov in zip(bp.inputs[n_seqs:], o_invs):
            m[bv] = ov

        def mv(v):
            if v not in m:
                src = bp.vars[v]
                m[v] = new_plan.new_var(src.dtype, list(src.shape), src.name, src.const)
            return m[v]
        for n_ in bp.nodes:
            out_nodes.append(Node(n_.op, [mv(i) for i in n_.inputs], [mv(o) for o in n_.outputs], dict(n_.params)))
        new_inner = Plan(inner.name + "_noglue", V, seq_vars + outs + list(inner.inputs[n_seqs:]),
                         list(inner.outputs), rest)
        out_nodes.append(Node("Scan", [n_steps_v] + o_seqs + [mv(o) for o in bp.outputs] + list(node.inputs[1 + n_seqs:]),
                              list(node.outputs), dict(p, n_seqs=n_seqs + len(outs), inner=new_inner)))
        changed = True
    if not changed:
        return plan
    new_plan.nodes = out_nodes
    return new_plan


def hoist_sequence_dots(plan: Plan, seq_inputs: List[int], invariant: set):
    """Matrix products of a Scan step that involve only a *sequence* row and loop-invariant
    matrices (``x_t @ W`` of an RNN gate) do not depend on the recurrence: they are removed from
    the step and computed for ALL steps by one GEMM before the loop (T x K @ K x M fills the
    chip; T separate K x M GEMVs cannot), the step then reads row t of the result.  The
    reference's ``scan_pushout`` rewrites (scan/rewriting.py) move Elemwise work on sequences out
    of the loop the same way but leave the dots inside.

    Returns ``(loop_plan, hoists)``; each hoist = {"kind": "gemv" | "dot", "seq": inner input id,
    "mat": invariant id, "alpha": float, "out": var id (appended to the plan inputs)}."""
    def const_value(vid):
        v = plan.vars[vid]
        if v.const is not None and len(v.const.get("data", ())) == 1:
            return float(v.const["data"][0])
        return None

    seqs = set(seq_inputs)
    keep, hoists = [], []
    for n in plan.nodes:
        h = None
        if n.op == "Gemv":
            y, alpha, A, x, beta = n.inputs
            if A in invariant and x in seqs and const_value(alpha) is not None:
                if const_value(beta) == 0.0:
                    h = {"kind": "gemv", "seq": x, "mat": A, "alpha": const_value(alpha)}
                elif const_value(beta) is not None and y in seqs:
                    # a chain  W1 x_t + W2 h_t  (the gate pre-activations a gradient Scan
                    # recomputes from stored states): accumulates onto an earlier hoisted product
                    h = {"kind": "gemv", "seq": x, "mat": A, "alpha": const_value(alpha),
                         "beta": const_value(beta), "acc": y}
        elif n.op in ("Dot22", "Dot"):
            a, b = n.inputs
            if a in seqs and b in invariant and plan.vars[b].ndim == 2:
                h = {"kind": "dot", "seq": a, "mat": b, "alpha": 1.0}
            elif (n.op == "Dot" and a in invariant and b in seqs and plan.vars[a].ndim == 2
                  and plan.vars[b].ndim == 1):
                h = {"kind": "gemv", "seq": b, "mat": a, "alpha": 1.0}
        if h is not None and plan.vars[n.outputs[0]].dtype in ("float32", "float64"):
            h["out"] = n.outputs[0]
            hoists.append(h)
            seqs.add(n.outputs[0])       # row t of the result is a sequence operand from here on
        else:
            keep.append(n)
    if not hoists:
        return plan, []
    loop = Plan(plan.name + "_seqdots", plan.vars, list(plan.inputs) + [h["out"] for h in hoists],
                list(plan.outputs), keep)
    return loop, hoists


def _fuse_rowpass(plan: Plan, steps: List[Step]) -> List[Step]:
    """GLM pattern  z = X.w (+ epilogue) -> row-wise Elemwise / full Sum -> X.T.r  ==>  one
    single-pass "rowpass" step (X read once) + deterministic folds of its per-workgroup partials.

    The reference graph of BASELINE config 5 reads X twice (Gemv on X, Gemv on
    X.T); whenever every step between the two is row-wise the linker can legally keep each row
    in registers for both contractions."""
    from .codegen import RP_MAXOPS, RP_MAXRED

    producer_step = {}
    for s in steps:
        for o in list(s.outputs) + ([s.reduce["out"]] if s.reduce else []):
            producer_step[o] = s
    for i1, g1 in enumerate(steps):
        if g1.kind != "gemv_epi" or len(g1.dots) != 1:
            continue
        xv, wv = g1.dots[0]
        if plan.vars[xv].ndim != 2:
            continue
        # G2: a single dot over DimShuffle{1,0}(X)
        for i2 in range(i1 + 1, len(steps)):
            g2 = steps[i2]
            if g2.kind not in ("gemv_epi", "rowdot"):
                continue
            d2 = g2.dots if g2.kind == "gemv_epi" else [list(g2.inputs)]
            if len(d2) != 1:
                continue
            xt, rv = d2[0]
            ds = producer_step.get(xt)
            if not (ds is not None and ds.kind == "node" and ds.node.op == "DimShuffle"
                    and ds.node.params["new_order"] == [1, 0] and ds.inputs == [xv]):
                continue
            fused = _build_rowpass(plan, steps, i1, i2, xv, wv, rv, RP_MAXOPS, RP_MAXRED)
            if fused is not None:
                return fused
    return steps


def _build_rowpass(plan, steps, i1, i2, xv, wv, rv, max_ops, max_red):
    g1, g2 = steps[i1], steps[i2]
    dt = plan.vars[xv].dtype
    group, produced = [g1], set(g1.outputs)
    outside = []
    for s in steps[i1 + 1:i2]:
        ins = set(s.inputs) | {v for d in s.dots for v in d}
        if ins & produced:
            ok = s.kind in ("elemwise", "reduce") and not s.dots and \
                all(plan.vars[v].ndim <= 1 for v in s.inputs) and \
                all(plan.vars[o].ndim == 1 for o in s.outputs)
            if s.kind == "reduce":
                r = s.reduce
                ok = ok and r["scalar_op"] == "add" and (r["axis"] is None or r["axis"] == [0]) \
                    and r["acc_dtype"] in ("float32", "float64")
            if not ok:
                return None
            group.append(s)
            produced |= set(s.outputs) | ({s.reduce["out"]} if s.reduce else set())
        else:
            outside.append(s)
    if rv not in produced or plan.vars[rv].dtype != dt:
        return None
    # ---- merge the group's scalar programs: input 0 = the dot, then external row operands ----
    ext, nodes, ref_of = [], [], {}

    def slot(v):
        if v not in ext:
            ext.append(v)
        return ["i", 1 + ext.index(v)]

    def absorb(st, in_refs):
        base = len(nodes)

        def rr(r):
            if r[0] == "i":
                return list(in_refs[r[1]])
            if r[0] == "t":
                return ["t", r[1] + base]
            return list(r)
        for n in st.scalar["nodes"]:
            nodes.append({"op": n["op"], "dtype": n["dtype"], "in": [rr(r) for r in n["in"]]})
        return [rr(r) for r in st.scalar["out"]]

    reds = []
    for st in group:
        in_refs = ([["i", 0]] if st is g1 else []) + \
            [ref_of[v] if v in ref_of else slot(v) for v in st.inputs]
        outs = absorb(st, in_refs)
        for o, k in zip(st.outputs, st.out_refs):
            ref_of[o] = outs[k]
        if st.reduce:
            reds.append((outs[st.reduce["ref"]], st.reduce["acc_dtype"], st.reduce["out"]))
    if len(reds) > max_red:
        return None
    # which group-produced vectors must be materialised (consumers outside the fused group)
    users = {}
    for s in steps:
        if s in group or s is g2:
            continue
        for v in list(s.inputs) + [v for d in s.dots for v in d]:
            users.setdefault(v, []).append(s)
    g2_other = set(g2.inputs) if g2.kind == "gemv_epi" else set()
    mat = [v for st in group for v in st.outputs
           if v in users or v in plan.outputs or v in g2_other]
    if len(ext) + len(mat) > max_ops:
        return None
    scalar_out = [ref_of[v] for v in mat] + [r[0] for r in reds] + [ref_of[rv]]
    scalar = {"n_in": 1 + len(ext), "nodes": nodes, "out": scalar_out}
    n_mat = len(mat)
    col_ws = plan.new_var(dt, [None, None], name="rowpass_col_ws")
    d_var = plan.new_var(dt, [None], name="rowpass_xt_r")
    red_ws = [plan.new_var("float64", [None], name="rowpass_red_ws") for _ in reds]
    ident = {"n_in": 1, "nodes": [], "out": [["i", 0]]}
    post = [Step("reduce", [col_ws], [], ident, out_refs=[],
                 reduce={"scalar_op": "add", "axis": [0], "acc_dtype": "float64", "out": d_var,
                         "ref": 0})]
    if g2.kind == "gemv_epi":
        post.append(Step("elemwise", [d_var] + list(g2.inputs), list(g2.outputs), g2.scalar,
                         out_refs=list(g2.out_refs)))
    else:
        post.append(Step("elemwise", [d_var], list(g2.outputs), ident, out_refs=[0]))
    for (ref, acc, out), wsv in zip(reds, red_ws):
        post.append(Step("reduce", [wsv], [], ident, out_refs=[],
                         reduce={"scalar_op": "add", "axis": None, "acc_dtype": "float64",
                                 "out": out, "ref": 0}))
    rp = Step("rowpass", list(ext), list(mat), scalar, out_refs=list(range(n_mat)),
              dots=[[xv, wv]],
              extra={"reds": [[n_mat + j, acc] for j, (_, acc, _) in enumerate(reds)],
                     "col_ref": n_mat + len(reds), "col_ws": col_ws, "red_ws": red_ws,
                     "dtype": dt},
              post=post, fallback=group + [g2])
    return steps[:i1] + outside + [rp] + steps[i2 + 1:]


# ----------------------------------------------------------------------------------------
# row chains: last-axis reductions + the Elemwise steps between them -> one kernel
# ----------------------------------------------------------------------------------------
def _step_reads(st: Step):
    used = list(st.inputs) + [v for d in st.dots for v in d]
    for xp in st.extra.get("xprog", {}).values():
        used += list(xp["step"].inputs)
    for q in st.fallback + st.post:
        used += _step_reads(q)
    return used


def _fuse_rowchain(plan: Plan, steps: List[Step], max_ops: int = 16) -> List[Step]:
    """A last-axis CAReduce whose (keepdims) result feeds Elemwise / further last-axis CAReduce
    steps over the same [..., K] space (softmax = max -> exp-sum -> scale; log-softmax; softmax
    gradient; mean/variance normalisation) becomes ONE "rowchain" step: every operand is read
    once, every intermediate between the reductions lives in registers (codegen.RowChainSpec).
    The reference runs such chains as separate passes (Softmax.c_code tensor/special.py:372-415)
    or separate nodes.  Run-time layout checks are the executor's; the original steps are kept
    as the fallback."""
    while True:
        fused = _fuse_one_rowchain(plan, steps, max_ops)
        if fused is None:
            return steps
        steps = fused


def _fuse_one_rowchain(plan: Plan, steps: List[Step], max_ops: int):
    out_set = set(plan.outputs)
    for i, seed in enumerate(steps):
        if seed.kind != "reduce" or not seed.inputs:
            continue
        D = plan.vars[seed.inputs[0]].ndim
        if D < 2 or seed.reduce["axis"] != [D - 1]:
            continue
        if any(plan.vars[v].ndim != D for v in seed.inputs):
            continue
        keep_order = list(range(D - 1)) + ["x"]
        members, glue = [i], []
        red_of = {seed.reduce["out"]: i}           # raw reduce result -> step index
        row_vars: Dict[int, int] = {}              # keepdims var -> step index of its reduce
        full_vars = {o: i for o in seed.outputs}   # [..., K] intermediates -> producing step
        for j in range(i + 1, len(steps)):
            t = steps[j]
            if (t.kind == "node" and t.node.op == "DimShuffle" and t.inputs[0] in red_of
                    and t.node.params["new_order"] == keep_order):
                row_vars[t.outputs[0]] = red_of[t.inputs[0]]
                glue.append(j)
                continue
            reads = _step_reads(t)
            if not any(v in row_vars or v in full_vars or v in red_of for v in reads):
                continue
            ok = (t.kind in ("elemwise", "reduce") and t.inputs and not t.dots
                  and all(plan.vars[v].ndim == D for v in t.inputs)
                  and not any(v in red_of for v in t.inputs)
                  and (t.kind != "reduce" or t.reduce["axis"] == [D - 1]))
            if not ok:
                break
            members.append(j)
            for o in t.outputs:
                full_vars[o] = j
            if t.kind == "reduce":
                red_of[t.reduce["out"]] = j
        if len(members) < 2:
            continue
        # absorb Elemwise producers whose results only the chain reads (softmax(x / T + mask):
        # the logits never go to memory)
        readers: Dict[int, set] = {}
        made_by: Dict[int, int] = {}
        for j, t in enumerate(steps):
            for v in _step_reads(t):
                readers.setdefault(v, set()).add(j)
            if t.kind == "elemwise" and not t.dots:
                for o in t.outputs:
                    made_by[o] = j
        grown = True
        while grown:
            grown = False
            for j in list(members):
                for v in steps[j].inputs:
                    pj = made_by.get(v)
                    if pj is None or pj in members or pj > members[-1]:
                        continue
                    P = steps[pj]
                    if not P.inputs or any(plan.vars[u].ndim != D for u in P.inputs):
                        continue
                    if any(o in out_set or not readers.get(o, set()) <= set(members)
                           for o in P.outputs):
                        continue
                    members = sorted(members + [pj])
                    for o in P.outputs:
                        full_vars[o] = pj
                    grown = True
                    break
                if grown:
                    break
        last = members[-1]
        glue = [g for g in glue if g < last]
        inside = set(members) | set(glue)
        used_outside = set(out_set)
        for j, t in enumerate(steps):
            if j not in inside:
                used_outside.update(_step_reads(t))
        pos = {j: k for k, j in enumerate(members)}
        ext: List[int] = []
        mlist, stored, keep = [], [], {}
        for j in members:
            t = steps[j]
            ins = []
            for v in t.inputs:
                if v in full_vars and full_vars[v] != j:
                    p = steps[full_vars[v]]
                    ins.append(["f", pos[full_vars[v]], p.out_refs[p.outputs.index(v)]])
                elif v in row_vars:
                    ins.append(["r", pos[row_vars[v]]])
                else:
                    if v not in ext:
                        ext.append(v)
                    ins.append(["e", ext.index(v)])
            m = {"scalar": copy.deepcopy(t.scalar), "ins": ins, "reduce": None, "stores": []}
            for o, ref in zip(t.outputs, t.out_refs):
                if o in used_outside:
                    m["stores"].append([ref, o])
                    stored.append(o)
            if t.kind == "reduce":
                r = t.reduce
                m["reduce"] = {"op": r["scalar_op"], "acc": r["acc_dtype"], "ref": r["ref"],
                               "out": r["out"], "store": False}
                need = r["out"] in used_outside
                for g in glue:
                    gk = steps[g]
                    if gk.inputs[0] == r["out"] and gk.outputs[0] in used_outside:
                        keep[gk.outputs[0]] = r["out"]
                        need = True
                if need:
                    m["reduce"]["store"] = True
                    stored.append(r["out"])
            mlist.append(m)
        if not stored or len(ext) + len(stored) > max_ops:
            continue
        fb = [steps[j] for j in sorted(inside)]
        rc = Step("rowchain", ext, stored, extra={"members": mlist, "D": D, "keep": keep},
                  fallback=fb)
        return [rc if j == last else s for j, s in enumerate(steps) if j == last or j not in inside]
    return None


# ----------------------------------------------------------------------------------------
# small-M GEMM chain + Elemwise consumer -> one kernel (kind "gemm_epi")
# ----------------------------------------------------------------------------------------
def _fuse_gemm_epi(plan: Plan, steps: List[Step], max_dots: int = 3, max_ops: int = 12) -> List[Step]:
    """``Gemm`` / ``Dot22`` nodes whose only reader is one Elemwise step (a recurrent gate
    ``sigmoid(h @ U + V_t) * h``, an MLP layer ``tanh(x @ W + b)``) become one "gemm_epi" step: the
    products are accumulated by the 16-row split-K MFMA schedule and the Elemwise runs on the
    accumulators.  Only worth it (and only generated) for outputs too small to fill the chip with
    128x128 tiles — the executor decides from the run-time shapes and otherwise runs the original
    steps, which are kept as the fallback (the big GEMM has its own alpha/beta epilogue)."""
    out_set = set(plan.outputs)
    readers: Dict[int, List[int]] = {}
    for j, t in enumerate(steps):
        for v in set(_step_reads(t)):
            readers.setdefault(v, []).append(j)

    def const1(vid):
        v = plan.vars[vid]
        return v.const is not None and len(v.const.get("data", ())) == 1

    producer = {}
    for j, t in enumerate(steps):
        if t.kind == "node" and t.node.op in ("Gemm", "Dot22", "Dot22Scalar"):
            producer[t.outputs[0]] = j
    if not producer:
        return steps
    removed, replaced = set(), {}
    for j, e in enumerate(steps):
        if e.kind != "elemwise" or e.dots or any(plan.vars[v].ndim != 2 for v in e.inputs):
            continue
        prods = []
        for v in e.inputs:
            pj = producer.get(v)
            if pj is None or pj in removed or v in out_set or readers.get(v) != [j]:
                continue
            n = steps[pj].node
            if plan.vars[v].dtype not in ("float32", "float64"):
                continue
            if n.op == "Gemm" and not (const1(n.inputs[1]) and const1(n.inputs[4])):
                continue
            if n.op == "Dot22Scalar" and not const1(n.inputs[2]):
                continue
            prods.append((v, pj))
        if not prods or len(prods) > max_dots:
            continue
        merged = Step("elemwise", list(e.inputs), list(e.outputs), copy.deepcopy(e.scalar),
                      out_refs=list(e.out_refs))
        dvars, dots = [], []
        for v, pj in prods:
            n = steps[pj].node
            dt = plan.vars[v].dtype
            d_var = plan.new_var(dt, [None, None], name="matdot")
            if n.op == "Dot22":
                dots.append([n.inputs[0], n.inputs[1]])
                pseudo = Step("elemwise", [d_var], [v], {"n_in": 1, "nodes": [], "out": [["i", 0]]},
                              out_refs=[0])
            elif n.op == "Dot22Scalar":   # dot(x, y) * a
                dots.append([n.inputs[0], n.inputs[1]])
                sc = {"n_in": 2, "out": [["t", 0]], "nodes": [
                    {"op": "mul", "dtype": dt, "in": [["i", 0], ["i", 1]]}]}
                pseudo = Step("elemwise", [d_var, n.inputs[2]], [v], sc, out_refs=[0])
                _inline_constants(pseudo, plan)
            else:   # Gemm(z, alpha, x, y, beta) = beta*z + alpha*dot(x, y)
                z, al, x, y, be = n.inputs
                dots.append([x, y])
                sc = {"n_in": 4, "out": [["t", 2]], "nodes": [
                    {"op": "mul", "dtype": dt, "in": [["i", 1], ["i", 0]]},
                    {"op": "mul", "dtype": dt, "in": [["i", 3], ["i", 2]]},
                    {"op": "add", "dtype": dt, "in": [["t", 1], ["t", 0]]}]}
                pseudo = Step("elemwise", [d_var, al, z, be], [v], sc, out_refs=[0])
                _inline_constants(pseudo, plan)
            _merge_producer(merged, pseudo, v)
            dvars.append(d_var)
        others = [v for v in merged.inputs if v not in dvars]
        if len(others) + len(merged.outputs) > max_ops or \
                any(plan.vars[v].ndim != 2 for v in others):
            continue
        order = dvars + others
        remap = {pos: ["i", order.index(v)] for pos, v in enumerate(merged.inputs)}
        merged.scalar = _remap_inputs(merged.scalar, remap, len(order))
        merged.inputs = others
        merged.dots = dots
        merged.kind = "gemm_epi"
        merged.fallback = [steps[pj] for _, pj in prods] + [e]
        replaced[j] = merged
        removed.update(pj for _, pj in prods)
    if not replaced:
        return steps
    return [replaced.get(j, s) for j, s in enumerate(steps) if j not in removed]


def _fuse_xprologue(plan: Plan, steps: List[Step]) -> List[Step]:
    """The vector of a fused GEMV chain that is itself a small Elemwise of vectors (the
    ``delta * (1 - h**2)`` feeding ``W . (...)`` in every step of a backward RNN Scan) is evaluated
    by the GEMV kernel while it loads the vector (and stored by its first wavefront when anything
    else reads it): one launch less per step.  The Elemwise step is kept inside the fused step and
    run as before whenever the run-time layout does not qualify."""
    out_set = set(plan.outputs)
    readers: Dict[int, List[int]] = {}
    made_by: Dict[int, int] = {}
    for j, t in enumerate(steps):
        for v in set(_step_reads(t)):
            readers.setdefault(v, []).append(j)
        if t.kind == "elemwise" and not t.dots and len(t.outputs) == 1:
            made_by[t.outputs[0]] = j
    removed = set()
    for j, g in enumerate(steps):
        if g.kind != "gemv_epi":
            continue
        for d, (_a, xv) in enumerate(g.dots):
            pj = made_by.get(xv)
            if pj is None or pj in removed or pj > j:
                continue
            P = steps[pj]
            if not (1 <= len(P.inputs) <= 4 and len(P.scalar["nodes"]) <= 12
                    and all(plan.vars[u].ndim <= 1 for u in P.inputs)
                    and all(plan.vars[u].dtype == plan.vars[xv].dtype for u in P.inputs)):
                continue
            rd = [r for r in readers.get(xv, []) if r != j]
            if any(pj < r < j for r in rd):
                continue          # something between needs the vector before the GEMV runs
            g.extra.setdefault("xprog", {})[d] = {"step": P, "store": bool(rd) or xv in out_set}
            removed.add(pj)
    if not removed:
        return steps
    return [s_ for j, s_ in enumerate(steps) if j not in removed]
