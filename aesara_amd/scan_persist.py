"""K10p — the Scan step loop as ONE persistent kernel (SURVEY §8a H11 "v2", BASELINE north_star).

Replaces, for the recurrent-vector class of inner graphs, what ``Scan.perform`` drives from the
host (scan/op.py:1673) through ``scan_perform.pyx:309-541``: per step, slice the sequences and
taps, run the inner function, write the outputs into the (circular) output buffers.  The
launch-list path (executor._scan_loop) issues the fused step kernels T times and re-streams every
weight matrix from L2 / HBM on every step; here one launch runs all T steps:

* the workgroups partition the output rows; each keeps ITS rows of every loop-invariant matrix in
  LDS for the whole loop (BASELINE config 4: 3 x 1024 x 1024 fp32 = 12 MiB over 256 CUs = 48 KiB
  per CU), so HBM sees the weights once per evaluation, not once per step;
* a vector that a later dot product needs in full (the new state h_t, an intermediate such as
  r*h) is exchanged through L2 as 8-byte ``{tag, value}`` granules written with ONE agent-scope
  store each (single-copy atomic, write-through: MI355X guide "handoff-1to1" / "allgather" rows);
  a consumer polls the granules it needs with relaxed agent-scope loads until they carry the tag
  of the producing step, stages the values in LDS and goes on — no grid barrier, no fences, no
  separate flags.  Tags are ``base + step + 1`` with ``base`` read from a device counter that
  workgroup 0 advances by T at the end, so replays (hipGraph) never see a stale tag; four slots
  per vector (step & 3) keep a fast producer from overwriting what a slow consumer still reads;
* everything a row owner needs again (its previous state rows, the rows of sequence operands of
  the next step) stays in registers: lane i of a wavefront owns one output row.

**Full reductions inside the step** (round 5; the reference runs them in the same loop,
scan_perform.pyx:309-541): ``(h ** 2).sum()``, ``x_t.max()`` ... of a vector of the state's length.
The row owners evaluate the reduced expression for their rows and publish it like any exchanged
vector; every workgroup gathers the whole vector and folds it in ONE fixed order (lanes stride over
it, butterfly across the wavefront: the same tree everywhere), so all threads of all workgroups hold
the identical 0-d result with no second protocol.  0-d values computed from such results run on every
lane; a 0-d nit-sot output is stored by one lane of workgroup 0.  That also gives the **do-while** of
a vector state (condition :424-426): every workgroup sees the same condition value, leaves the loop
after the same step, workgroup 0 stores the trip count for the one host read after the launch.

Eligibility (anything else runs the launch-list path, and ``PlanExecutor.scan_modes`` says which
was taken): no shared outputs, do-while only on a condition behind a reduction, sit- / mit-sot outputs with taps down to -8 (tap -1 is
the exchanged state; older taps are values the row owner produced itself and stay in its
registers, usable element-wise), mit-mot groups whose largest in-tap is their largest out-tap (the
gradient forms [0, 1] -> [1], [0 .. m] -> [1 .. m], [0, 1, 3] -> [1, 3] ...: see ``analyze``), the fused
inner steps are Gemv
chains + Elemwise on float32 / float64 vectors of one length M, matrices are loop invariant (rows
that are not whole 16-byte vectors are zero-padded by the executor), at least one exchanged
vector, T >= 2, and the matrix rows of a workgroup fit on chip (LDS + VGPRs).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os

from . import knobs
from . import codegen as cg

SP_MAXMAT = 12
SP_MAXSEQ = 28
SP_MAXNSQ = 12
SP_MAXOUT = 16
SPIN_LIMIT = 1 << int(knobs.get("SPIN_LOG2"))
TRACE_T0, TRACE_NT, TRACE_MARKS = 200, 32, 16      # steps traced, stamps per step (u64 each): tools/sp_trace.py
LDS_BUDGET = 156 * 1024


class SpArgs(C.Structure):
    _fields_ = [
        ("T", C.c_int64),
        ("mat", C.c_void_p * SP_MAXMAT), ("mat_rs", C.c_int64 * SP_MAXMAT),
        ("seq", C.c_void_p * SP_MAXSEQ), ("seq_ts", C.c_int64 * SP_MAXSEQ),
        ("seq_es", C.c_int64 * SP_MAXSEQ),
        ("nsq", C.c_void_p * SP_MAXNSQ), ("nsq_es", C.c_int64 * SP_MAXNSQ),
        ("out", C.c_void_p * SP_MAXOUT), ("out_rs", C.c_int64 * SP_MAXOUT),
        ("out_store", C.c_int64 * SP_MAXOUT), ("out_pos0", C.c_int64 * SP_MAXOUT),
        ("xch", C.c_void_p), ("ctl", C.c_void_p),
    ]


SP_STRUCT = r"""
#define SP_MAXMAT %d
#define SP_MAXSEQ %d
#define SP_MAXNSQ %d
#define SP_MAXOUT %d
struct SpArgs {
  i64 T;
  const void* mat[SP_MAXMAT]; i64 mat_rs[SP_MAXMAT];
  const void* seq[SP_MAXSEQ]; i64 seq_ts[SP_MAXSEQ]; i64 seq_es[SP_MAXSEQ];
  const void* nsq[SP_MAXNSQ]; i64 nsq_es[SP_MAXNSQ];
  void* out[SP_MAXOUT]; i64 out_rs[SP_MAXOUT]; i64 out_store[SP_MAXOUT]; i64 out_pos0[SP_MAXOUT];
  unsigned long long* xch; unsigned* ctl;
};
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
""" % (SP_MAXMAT, SP_MAXSEQ, SP_MAXNSQ, SP_MAXOUT)


# ---------------------------------------------------------------------------------------------
# static analysis of the fused inner steps (once per inner executor)
# ---------------------------------------------------------------------------------------------
class Program:
    """What the loop does, in terms of vector ids (= inner plan variable ids):

    ``seq``    : var -> sequence slot (rows of a [T, ...] array, incl. hoisted sequence dots)
    ``state``  : var -> recurrent output index k (value of the previous step)
    ``nsq``    : var -> invariant slot (vector / scalar operand) ; ``mats``: var -> matrix slot
    ``phases`` : [{"dots": [(matrix var, vector var)], "ins": [...], "outs": [...], "scalar",
                   "out_refs"}]
    ``outs``   : per inner-plan output: (produced var, "rec" | "nit", index)
    """

    def __init__(self):
        self.seq, self.state, self.nsq, self.mats = {}, {}, {}, {}
        self.phases, self.outs = [], []
        self.new_of_state = {}      # state var -> var holding its new value
        self.tap_seq = {}           # mit-mot tap-1 var -> recurrent output whose buffer holds its rows
        self.passthru = {}          # nit-sot output index -> sequence var it hands out unchanged
        self.passthru_t = set()     # inner output vars that are such a sequence TRANSPOSED per step
        self.passthru_tj = set()    # ... as nit-sot output indices
        self.exchanged = []         # produced vars that some dot needs in full
        self.mode = "vec"
        self.dtype = "float32"
        self.older = {}             # inner input var of a tap < -1 -> (recurrent output, depth)
        self.depth = {}             # recurrent output -> deepest tap (1 = the usual sit-sot)
        self.n_rec_inputs = 0       # inner inputs taken by the recurrent outputs' taps
        self.tap_top = {}           # mit-mot sequence var -> the tap it is (the group's largest)
        self.mm_extra = []          # [(slot, group, out tap)]: out-taps 2.. of a mit-mot group write the
        #                             group's buffer too — further output slots on the same buffer
        self.mm_first = {}          # mit-mot group -> its smallest out-tap (the row its slot starts at)
        self.zero_d = set()         # 0-d values (reduction results and what is computed from them)
        self.cond = None            # do-while: the 0-d condition variable (as_while)


class _DotPhase:
    """Stand-in step for a plain ``Dot22(operand, invariant matrix)`` node of a matrix-state loop."""
    kind, reduce, post, fallback, extra = "gemm_epi", None, (), (), {}

    def __init__(self, operand, weight, out, add=None, dtype="float32"):
        self.dots = [[operand, weight]]
        self.inputs, self.outputs = ([] if add is None else [add]), [out]
        self.scalar = {"n_in": 1, "nodes": [], "out": [["i", 0]]} if add is None else \
            {"n_in": 2, "nodes": [{"op": "add", "in": [["i", 0], ["i", 1]], "dtype": dtype}],
             "out": [["t", 0]]}
        self.out_refs = [0]
        self.node = None


class _RowDot:
    """Stand-in step for a bare ``rowdot(A, x)`` of a vector-state loop (a Gemv whose epilogue went
    into another fused step, e.g. a reduction with Elemwise outputs): a product phase that hands
    the dot through."""
    kind, reduce, post, fallback, extra, node = "gemv_epi", None, (), (), {}, None

    def __init__(self, A, x, out):
        self.dots = [[A, x]]
        self.inputs, self.outputs = [], [out]
        self.scalar = {"n_in": 1, "nodes": [], "out": [["i", 0]]}
        self.out_refs = [0]


def _const1(plan, vid):
    v = plan.vars[vid]
    return v.const is not None and len(v.const.get("data", ())) == 1 and float(v.const["data"][0]) == 1.0


RED_BASE = 1 << 20          # id of the vector a reduction folds: RED_BASE + its 0-d result's id
FAKE_BASE = 1 << 21         # id of the unread one-step-ago value of a mit-mot out-tap: FAKE_BASE + its slot


def analyze(inner, p, n_seqdots):
    """Returns (Program, None) or (None, reason)."""
    plan = inner.plan
    n_seqs = p["n_seqs"]
    as_while = bool(p.get("as_while", False))
    if p.get("n_shared_outs", 0):
        return None, "shared outputs / do-while"
    # mit-mot groups of the form a gradient Scan uses for the state it propagates (Scan.L_op,
    # scan/op.py:2379): input taps [0, 1], output tap [1] — step i reads rows i and i + 1 of the
    # buffer and overwrites row i + 1.  Row i is what step i - 1 wrote (the running value: a
    # state with its initial value in row 0), row i + 1 still holds what the caller put there
    # (the incoming gradient of that step): a sequence that happens to live in the output buffer.
    # The gradient of a recurrence with taps [-1 .. -m] gives taps [0 .. m] -> [1 .. m] (in any
    # order): out-tap j of step i is what step i + 1 reads as tap j - 1 — m states, each with its
    # own output slot on the SAME buffer (the slot of out-tap j starts one row after the row tap
    # j - 1 reads first), and the largest tap is the buffer-resident sequence.
    # General rule (round 5; taps [-1, -3] give [0, 1, 3] -> [1, 3]): when step i reads row i + a the
    # last step that wrote it is the one whose out-tap b is the SMALLEST b > a (step i - (b - a): an
    # out-tap <= a of an earlier step lands below, a larger one was overwritten since); in-tap a is
    # that out-tap's value of d = b - a steps ago.  d = 1: a state as above; d >= 2: a value the ROW
    # OWNER wrote itself d steps ago — it stays in its registers (usable element-wise only), like
    # the older taps of a mit-sot output.  The largest in-tap must be the largest out-tap (the
    # sequence entry that the step overwrites after reading it).  Vector class only.
    mm_in = [list(t) for t in p.get("mit_mot_in_slices", [])]
    mm_out = [list(t) for t in p.get("mit_mot_out_slices", [])]
    if len(mm_in) != len(mm_out) or any(
            len(ti) < 2 or len(set(ti)) != len(ti) or len(set(to)) != len(to) or not to or
            min(ti) < 0 or min(to) < 1 or max(ti) != max(to) or max(ti) > 8
            for ti, to in zip(mm_in, mm_out)):
        return None, "mit-mot taps other than [.. top] -> [.. top]"
    mm_general = any(sorted(ti) != list(range(len(ti))) or sorted(to) != list(range(1, len(ti)))
                     for ti, to in zip(mm_in, mm_out))
    n_mm = len(mm_in)
    # sit-sot / mit-sot outputs: the tap -1 value is the recurrent state proper (may feed dots, is
    # exchanged); older taps (-2, -3, ...: scan_perform.pyx:321-340 hands the step one row per
    # tap) are values the ROW OWNER produced itself 2, 3, ... steps ago — they stay in its
    # registers (a shift per step), so they may only be used element-wise
    taps = [list(t) for t in p["mit_sot_in_slices"]] + [list(t) for t in p["sit_sot_in_slices"]]
    if any((-1 not in t) or any(x >= 0 for x in t) or len(set(t)) != len(t) or min(t) < -8 for t in taps):
        return None, "taps without -1 / non-negative / deeper than 8"
    n_rec, n_nit = n_mm + len(taps), p["n_nit_sot"]
    if n_rec == 0:
        return None, "no recurrent state"
    pr = Program()
    ins = list(plan.inputs)
    n_fixed = len(ins) - n_seqdots
    for s, v in enumerate(ins[:n_seqs]):
        pr.seq[v] = s
    idx = n_seqs
    n_outer = n_rec + n_nit
    slot_of_mm = {}                                       # (group, out tap) -> output slot
    for g, ti in enumerate(mm_in):
        m, outs_g = max(ti), sorted(mm_out[g])
        pr.mm_first[g] = outs_g[0]
        for j in outs_g:
            if j == outs_g[0]:
                slot_of_mm[(g, j)] = g
            else:
                slot_of_mm[(g, j)] = n_outer + len(pr.mm_extra)
                pr.mm_extra.append((n_outer + len(pr.mm_extra), g, j))
        read1 = set()
        for tap in ti:
            if tap < m:
                b = min(bb for bb in outs_g if bb > tap)
                sl = slot_of_mm[(g, b)]
                if b - tap == 1:
                    pr.state[ins[idx]] = sl
                    read1.add(sl)
                else:
                    pr.older[ins[idx]] = (sl, b - tap)
                    pr.depth[sl] = max(pr.depth.get(sl, 1), b - tap)
            else:
                pr.seq[ins[idx]] = n_seqs + n_seqdots + g     # slot of the buffer-resident sequence
                pr.tap_seq[ins[idx]] = g
                pr.tap_top[ins[idx]] = m
            idx += 1
        for j in outs_g:
            sl = slot_of_mm[(g, j)]
            if pr.depth.get(sl, 1) > 1 and sl not in read1:
                # an out-tap that is read again only two or more steps later: its value of ONE
                # step ago still has to pass through the owner's registers (a state nobody reads)
                pr.state[FAKE_BASE + sl] = sl
    for k, tk in enumerate(taps):
        for tap in tk:
            if tap == -1:
                pr.state[ins[idx]] = n_mm + k
            else:
                pr.older[ins[idx]] = (n_mm + k, -tap)
            idx += 1
        pr.depth[n_mm + k] = -min(tk)
    pr.n_rec_inputs = idx - n_seqs
    inv = ins[idx:n_fixed]
    for j, v in enumerate(ins[n_fixed:]):
        pr.seq[v] = n_seqs + j
    if n_seqs + n_seqdots + n_mm > SP_MAXSEQ:
        return None, "too many sequence operands"
    inv_set = set(inv)
    produced = {}
    # "vec": Gemv chains on vectors (state h[M]); "mat": small-M GEMM chains on a matrix state
    # (h[B, N], batch of independent recurrences sharing the weights)
    pr.mode = "mat" if plan.vars[plan.outputs[0]].ndim == 2 or any(
        st.kind == "gemm_epi" or (st.kind == "node" and st.node.op in ("Dot22", "Dot")
                                  and plan.vars[st.outputs[0]].ndim == 2) for st in inner.steps) else "vec"
    # one floating dtype throughout (the state's): float32 always, float64 for the vector class
    pr.dtype = plan.vars[plan.outputs[0]].dtype
    if pr.dtype not in ("float32", "float64"):
        return None, "state dtype %s" % pr.dtype
    nd = 2 if pr.mode == "mat" else 1
    ok_kinds = ("gemm_epi", "elemwise") if pr.mode == "mat" else ("gemv_epi", "elemwise")
    alias = {}

    def res(v):
        while v in alias:
            v = alias[v]
        return v
    # a GEMV step that evaluates its vector on the fly (fusion._fuse_xprologue) is taken apart
    # again: here the vector is a phase of its own (computed by the row owners, then exchanged)
    steps = []
    for st in inner.steps:
        for _d, xp in sorted(st.extra.get("xprog", {}).items()):
            steps.append(xp["step"])
        steps.append(st)
    readers = {}
    for st in steps:
        for v in list(st.inputs) + [x for d in st.dots for x in d]:
            readers[v] = readers.get(v, 0) + 1
    if as_while and pr.mode != "vec":
        return None, "shared outputs / do-while"
    if mm_general and pr.mode != "vec":
        return None, "mit-mot taps other than [0 .. m] -> [1 .. m] on a matrix state"
    for st in steps:
        if st.kind == "node" and st.node.op in ("SpecifyShape", "ViewOp"):
            alias[st.outputs[0]] = st.inputs[0]      # value-preserving views: same vector / matrix
            continue
        if st.kind == "node" and st.node.op == "DimShuffle" and pr.mode == "vec" and \
                res(st.inputs[0]) in pr.zero_d and all(d == "x" for d in st.node.params["new_order"]):
            alias[st.outputs[0]] = st.inputs[0]      # a 0-d value made broadcastable against the vectors
            continue
        if st.kind == "node" and st.node.op == "DimShuffle" and pr.mode == "mat" and \
                st.node.params["new_order"] == [1, 0] and st.inputs[0] in pr.seq and \
                st.inputs[0] not in pr.tap_seq and not readers.get(st.outputs[0]) and \
                st.outputs[0] in plan.outputs:
            pr.passthru_t.add(st.outputs[0])         # a sequence-only matrix handed out transposed
            alias[st.outputs[0]] = st.inputs[0]
            continue
        if st.kind == "node" and st.node.op in ("Dot22", "Dot") and pr.mode == "mat" and \
                len(st.inputs) == 2 and st.inputs[1] in inv_set:
            # a product that feeds several Elemwise steps (no single epilogue to fuse with): a
            # phase whose "epilogue" hands the dot through
            st = _DotPhase(st.inputs[0], st.inputs[1], st.outputs[0], dtype=pr.dtype)
        if st.kind == "node" and st.node.op == "Gemm" and pr.mode == "mat" and \
                st.inputs[3] in inv_set and _const1(plan, st.inputs[1]) and _const1(plan, st.inputs[4]):
            # z + x @ W with nothing fused behind it (the state update of a plain RNN's gradient
            # step): a product phase whose epilogue adds z
            st = _DotPhase(st.inputs[2], st.inputs[3], st.outputs[0], add=st.inputs[0], dtype=pr.dtype)
        if st.kind == "node" and st.node.op == "DimShuffle" and pr.mode == "vec" and \
                [d for d in st.node.params["new_order"] if d != "x"] == [0] and \
                plan.vars[st.inputs[0]].ndim == 1 and not readers.get(st.outputs[0]) and \
                st.outputs[0] in plan.outputs:
            alias[st.outputs[0]] = st.inputs[0]      # a step output handed out as a row / column
            continue
        if st.kind == "rowdot" and pr.mode == "vec" and len(st.inputs) == 2:
            st = _RowDot(st.inputs[0], st.inputs[1], st.outputs[0])
        red = None
        if st.kind == "reduce" and pr.mode == "vec" and st.reduce is not None and not st.dots and \
                not st.post and not st.fallback and not st.extra.get("xprog"):
            r_ = st.reduce
            nd_in = max([plan.vars[res(v)].ndim for v in st.inputs] or [0])
            if r_["scalar_op"] not in ("add", "mul", "maximum", "minimum"):
                return None, "reduction with %s" % r_["scalar_op"]
            if nd_in != 1 or (r_["axis"] is not None and list(r_["axis"]) != [0]) or \
                    plan.vars[r_["out"]].ndim != 0:
                return None, "partial reduction inside the step"
            # (float32 values fold in float64 when the reference's accumulator is — CAReduce's
            # acc_dtype rule, tensor/elemwise.py:1495 — and are rounded to the state's dtype once)
            if r_["acc_dtype"] not in (pr.dtype, "float64") or plan.vars[r_["out"]].dtype != pr.dtype:
                return None, "reduction accumulates in another dtype"
            red = {"op": r_["scalar_op"], "ref": r_["ref"], "out": r_["out"], "u": RED_BASE + r_["out"],
                   "acc": r_["acc_dtype"]}
        elif st.kind not in ok_kinds or st.reduce is not None or st.post or \
                (st.fallback and st.kind != "gemm_epi"):
            return None, f"step kind {st.kind} ({st.node.op if st.node else ''})"
        dots = []
        # (invariant matrix, loop operand): gemv_epi stores (A, x), gemm_epi (operand, weight)
        pairs = [(b, a) for a, b in st.dots] if st.kind == "gemm_epi" else list(st.dots)
        pairs = [(a, res(x)) for a, x in pairs]
        st_inputs = [res(v) for v in st.inputs]
        for a, x in pairs:
            if a not in inv_set or plan.vars[a].ndim != 2:
                return None, "dot with a loop-varying matrix"
            if a not in pr.mats:
                pr.mats[a] = len(pr.mats)
            if x in pr.older:
                return None, "dot with a tap older than -1"
            if not (x in pr.seq or x in pr.state or x in produced or x in inv_set):
                return None, "dot vector of unknown origin"
            if x in pr.tap_seq:
                return None, "dot with a mit-mot tap"
            if x in inv_set and x not in pr.nsq:
                pr.nsq[x] = len(pr.nsq)
            dots.append((a, x))
        for v in st_inputs:
            if v in inv_set:
                if plan.vars[v].ndim > nd:
                    return None, "matrix used element-wise"
                if v not in pr.nsq:
                    pr.nsq[v] = len(pr.nsq)
            elif not (v in pr.seq or v in pr.state or v in produced or v in pr.older):
                return None, "operand of unknown origin"
            if plan.vars[v].dtype != pr.dtype:
                return None, "operand dtype differs from the state's"
        # a phase whose operands are all 0-d (results of reductions, scalars): its outputs are 0-d,
        # every lane evaluates it (the do-while condition must be uniform)
        scalar_phase = pr.mode == "vec" and not dots and red is None and bool(st_inputs) and \
            all(v in pr.zero_d or (v in inv_set and plan.vars[v].ndim == 0) for v in st_inputs)
        for o in st.outputs:
            if scalar_phase and plan.vars[o].ndim == 0 and \
                    (plan.vars[o].dtype == pr.dtype or (as_while and plan.vars[o].dtype == "bool")):
                pr.zero_d.add(o)            # (bool: only ever the condition, kept as 0 / 1 in a register)
            elif plan.vars[o].ndim != nd or plan.vars[o].dtype != pr.dtype:
                return None, "step output is not a %s %s" % (pr.dtype, "matrix" if nd == 2 else "vector")
            produced[o] = len(pr.phases)
        ph = {"dots": dots, "ins": list(st_inputs), "outs": list(st.outputs),
              "scalar": st.scalar, "out_refs": list(st.out_refs)}
        if red is not None:
            if any(v in pr.zero_d for v in st_inputs) and all(
                    v in pr.zero_d or v in inv_set and plan.vars[v].ndim == 0 for v in st_inputs):
                return None, "reduction of a 0-d value"
            ph["reduce"] = red
            pr.zero_d.add(red["out"])
            produced[red["out"]] = len(pr.phases)
        if scalar_phase:
            ph["scalar_phase"] = True
        pr.phases.append(ph)
    if len(pr.mats) > SP_MAXMAT or len(pr.nsq) > SP_MAXNSQ or not pr.mats:
        return None, "no / too many matrices"
    # inner outputs: every out-tap of every mit-mot group, the mit-sot / sit-sot outputs, the nit-sots
    slot_of = [slot_of_mm[(g, j)] for g, to in enumerate(mm_out) for j in to] + \
        list(range(n_mm, n_rec)) + list(range(n_rec, n_rec + n_nit))
    if len(plan.outputs) != len(slot_of) + (1 if as_while else 0) or n_outer + len(pr.mm_extra) > SP_MAXOUT:
        return None, "output count"
    if as_while:
        # the condition must come behind a reduction of the same step: every workgroup has then
        # gathered a vector all the others published after reading the launch's tag base, so
        # none can leave the loop (and workgroup 0 advance the base) before all have started
        c = res(plan.outputs[-1])
        first_red = [pi for pi, ph in enumerate(pr.phases) if ph.get("reduce")]
        if c not in pr.zero_d or c not in produced or not first_red or produced[c] < first_red[0]:
            return None, "do-while condition not behind a reduction of the step"
        pr.cond = c
    inner_of_slot = {}
    for ji, o in enumerate(plan.outputs[:len(slot_of)]):
        j = slot_of[ji]
        inner_of_slot[j] = ji
        if res(o) not in produced:
            if n_rec <= j < n_outer and res(o) in pr.seq and res(o) not in pr.tap_seq:
                pr.passthru[j] = res(o)      # a nit-sot output that is a row of a sequence
                if o in pr.passthru_t:
                    pr.passthru_tj.add(j)
                continue
            return None, "a step output is not computed by a fused step"
        if res(o) in pr.zero_d and not (n_rec <= j < n_outer):
            return None, "0-d recurrent output"
        pr.outs.append((res(o), "nit" if n_rec <= j < n_outer else "rec", j))
    for v, k in pr.state.items():
        pr.new_of_state[v] = res(plan.outputs[inner_of_slot[k]])
    # which produced vectors have to be exchanged (some dot reads them in full)
    need = []
    for ph in pr.phases:
        for _a, x in ph["dots"]:
            src = pr.new_of_state.get(x, x)
            if (x in pr.state or x in produced) and src not in need:
                need.append(src)
    if not need:
        return None, "no recurrent dot product"
    if any(v in pr.zero_d for v in need):
        return None, "dot with a 0-d value"
    pr.exchanged = need + [ph["reduce"]["u"] for ph in pr.phases if ph.get("reduce")]
    return pr, None


# ---------------------------------------------------------------------------------------------
# kernel generator
# ---------------------------------------------------------------------------------------------
class Spec:
    """Shape-specialised persistent kernel: M rows, K per matrix, R rows per workgroup."""

    def __init__(self, prog: Program, plan, M, Ks, lens, R, nw, place=None, dtype="float32"):
        self.dtype = dtype
        self.prog, self.plan, self.M, self.Ks, self.lens, self.R, self.nw = \
            prog, plan, M, dict(Ks), dict(lens), R, nw
        assert R % nw == 0
        self.place = dict(place or {})      # matrix var -> "reg" (rows in VGPRs) | "lds"
        # polling shape (measured defaults; environment overrides for sweeps)
        self.var = {"pollw": min(nw, int(knobs.get("SP_POLLW"))),
                    "sleep": int(knobs.get("SP_SLEEP")),
                    "delay": int(knobs.get("SP_DELAY")),
                    "repoll": 0}      # (re-polling only the missing granules: measured null r03, removed)
        # per-phase timeline (tools/sp_trace.py): thread 0 of workgroups 0 and G/2 stamps s_memtime
        # at the marks of TRACE_NT steps into the control buffer's tail
        self.trace = bool(int(knobs.get("SP_TRACE")))
        self.marks = []

    def key(self):
        pr = self.prog
        blob = json.dumps(["sp9" + ("t" if self.trace else "") + ("" if int(knobs.get("SP_EARLYDOTS")) else "n"), sorted(self.prog.older.items()), self.dtype, sorted(self.var.items()), sorted(self.place.items()), self.M, sorted(self.Ks.items()), sorted(self.lens.items()),
                           self.R, self.nw, sorted(pr.seq.items()), sorted(pr.state.items()),
                           sorted(pr.nsq.items()), sorted(pr.mats.items()),
                           [[ph["dots"], ph["ins"], ph["outs"], ph["scalar"], ph["out_refs"]]
                            for ph in pr.phases], pr.outs, pr.exchanged, sorted(pr.tap_seq.items())] +
                          ([["red1", [ph.get("reduce") for ph in pr.phases], sorted(pr.zero_d), pr.cond]]
                           if pr.zero_d else []),
                          sort_keys=True)
        return hashlib.sha256(blob.encode()).hexdigest()[:24]


def xch_layout(prog, lens, gpv=1):
    """granule offsets of the exchange buffer: per exchanged vector 4 slots of padded length
    (``gpv`` 8-byte granules per value: 2 for float64 — high and low word, each tagged)."""
    off, total = {}, 0
    for v in prog.exchanged:
        lp = (gpv * lens[v] + 15) // 16 * 16
        off[v] = (total, lp)
        total += 4 * lp
    return off, total


def generate(spec: Spec):
    pr, M, R, NW = spec.prog, spec.M, spec.R, spec.nw
    BLOCK = 64 * NW
    RPW = R // NW                         # rows per wavefront = lanes that run the epilogue
    name = "sp_" + spec.key()
    F64 = spec.dtype == "float64"
    T = "double" if F64 else "float"
    VEC = 2 if F64 else 4                 # elements per 16-byte vector
    GPV = 2 if F64 else 1                 # 8-byte granules per exchanged value
    comps = "xyzw"[:VEC]
    L = [cg.PRELUDE, SP_STRUCT, "typedef %s TV __attribute__((ext_vector_type(%d)));" % (T, VEC)]
    AG = "__ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT"

    def vdot(w, x):
        return " + ".join("%s.%s * %s.%s" % (w, c, x, c) for c in comps)

    # ---- LDS: matrix rows of this workgroup + staged dot vectors (two step parities) ----------
    woff, wtot = {}, 0
    in_reg = {a for a in pr.mats if spec.place.get(a) == "reg"}
    for a, slot in sorted(pr.mats.items(), key=lambda t: t[1]):
        if a in in_reg:
            continue
        woff[a] = wtot
        wtot += R * spec.Ks[a]
    # staged vectors: key (var, "prev" | "cur" | "glob"); invariant dot vectors are staged once
    stage, stot = {}, 0
    inv_stage, itot = {}, 0
    for ph in pr.phases:
        for a, x in ph["dots"]:
            K = spec.Ks[a]
            if x in pr.nsq:
                if x not in inv_stage:
                    inv_stage[x] = itot
                    itot += K
            else:
                kind = "prev" if x in pr.state else ("cur" if x not in pr.seq else "glob")
                if (x, kind) not in stage:
                    stage[(x, kind)] = stot
                    stot += K
    for ph in pr.phases:
        if ph.get("reduce"):         # the vector a reduction folds is gathered like a dot operand
            stage[(ph["reduce"]["u"], "cur")] = stot
            stot += -(-M // VEC) * VEC
    xoff, _xtot = xch_layout(pr, spec.lens, GPV)
    L.append('extern "C" __global__ __launch_bounds__(%d) void %s(SpArgs a) {' % (BLOCK, name))
    L.append("  __shared__ __attribute__((aligned(16))) %s Wl[%d];" % (T, max(wtot, 4)))
    L.append("  __shared__ __attribute__((aligned(16))) %s Vl[2][%d];" % (T, max(stot, 4)))
    if itot:
        L.append("  __shared__ __attribute__((aligned(16))) %s Il[%d];" % (T, itot))
    L.append("  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;")
    L.append("  const i64 row0 = (i64)blockIdx.x * %d;" % R)
    L.append("  const i64 myrow = row0 + wave * %d + lane;     // row owned by this lane (lane < %d)" % (RPW, RPW))
    L.append("  const bool owner = lane < %d && myrow < %d;" % (RPW, M))
    L.append("  const unsigned base = __hip_atomic_load(a.ctl, %s);" % AG)
    L.append("  unsigned* errp = a.ctl + 1;")
    marks = []
    if spec.trace:
        L.append("  const bool tr_on = threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2);")
        L.append("  unsigned long long* tr = (unsigned long long*)(a.ctl + 16) + (blockIdx.x == 0 ? 0 : %d);"
                 % (TRACE_NT * TRACE_MARKS + 4))
        L.append("  if (tr_on) { tr[%d] = __builtin_amdgcn_s_memtime(); tr[%d] = __builtin_amdgcn_s_memrealtime(); }"
                 % (TRACE_NT * TRACE_MARKS, TRACE_NT * TRACE_MARKS + 1))

    def stamp(label):
        """record s_memtime under `label` (trace builds only; marks are numbered in code order)"""
        if not spec.trace:
            return
        k = len(marks)
        marks.append(label)
        assert k < TRACE_MARKS
        L.append("    if (tr_on && t >= %d && t < %d) tr[(t - %d) * %d + %d] = __builtin_amdgcn_s_memtime();"
                 % (TRACE_T0, TRACE_T0 + TRACE_NT, TRACE_T0, TRACE_MARKS, k))
    # ---- matrix rows -> LDS ------------------------------------------------------------------
    for av, slot in sorted(pr.mats.items(), key=lambda t: t[1]):
        K = spec.Ks[av]
        K4 = K // VEC
        if av in in_reg:
            # rows of this wavefront stay in VGPRs for the whole loop: lane l holds the 16-byte
            # vectors l, l + 64, ... of each of its RPW rows (fully unrolled, static indices)
            for i in range(RPW):
                L.append("  const i64 wrow%d_%d = row0 + wave * %d + %d;" % (slot, i, RPW, i))
                for q in range(K4 // 64):
                    L.append("  const TV wr%d_%d_%d = wrow%d_%d < %d ? *(const TV*)((const %s*)a.mat[%d] + "
                             "wrow%d_%d * a.mat_rs[%d] + %d * (lane + %d)) : TV(0);"
                             % (slot, i, q, slot, i, M, T, slot, slot, i, slot, VEC, 64 * q))
            continue
        L.append("  for (int idx = threadIdx.x; idx < %d; idx += %d) {" % (R * K4, BLOCK))
        L.append("    const int j = idx / %d, k4 = idx %% %d;" % (K4, K4))
        L.append("    const i64 r = row0 + j;")
        L.append("    TV v = TV(0);")
        L.append("    if (r < %d) v = *(const TV*)((const %s*)a.mat[%d] + r * a.mat_rs[%d] + %d * k4);"
                 % (M, T, slot, slot, VEC))
        L.append("    *(TV*)(Wl + %d + j * %d + %d * k4) = v;" % (woff[av], K, VEC))
        L.append("  }")
    if any(spec.lens.get(x, spec.Ks[a]) != spec.Ks[a] for ph in pr.phases for a, x in ph["dots"]):
        # some contraction length is padded: the tails of the staged images must read as zeros
        L.append("  for (int k = threadIdx.x; k < %d; k += %d) { Vl[0][k] = 0; Vl[1][k] = 0; }"
                 % (max(stot, 4), BLOCK))
        if itot:
            L.append("  for (int k = threadIdx.x; k < %d; k += %d) Il[k] = 0;" % (itot, BLOCK))
        L.append("  __syncthreads();")
    for x, off in inv_stage.items():
        K = spec.lens[x]
        L.append("  for (int k = threadIdx.x; k < %d; k += %d) Il[%d + k] = "
                 "((const %s*)a.nsq[%d])[k * a.nsq_es[%d]];" % (K, BLOCK, off, T, pr.nsq[x], pr.nsq[x]))
    # ---- registers of the row owner --------------------------------------------------------------
    out_of = {}
    for o, kind, j in pr.outs:
        out_of.setdefault(o, []).append((kind, j))
    for v, k in pr.state.items():
        # value of the previous step: row (pos0 - 1) of the output buffer holds the initial state
        L.append("  %s own_%d = 0;" % (T, v))
        L.append("  if (owner) own_%d = ((const %s*)a.out[%d])[((a.out_pos0[%d] + a.out_store[%d] - 1) %% "
                 "a.out_store[%d]) * a.out_rs[%d] + myrow];" % (v, T, k, k, k, k, k))
    # older taps: the owner's own values of 2 .. depth steps ago (buffer row pos0 - d holds the
    # initial one); a depth no tap names still needs its register (taps [-1, -3] pass through -2)
    hist = {}
    by_kd = {kd: v for v, kd in pr.older.items()}
    for k, D in sorted(pr.depth.items()):
        for d in range(2, D + 1):
            nm = ("own_%d" % by_kd[(k, d)]) if (k, d) in by_kd else "hist_%d_%d" % (k, d)
            hist[(k, d)] = nm
            L.append("  %s %s = 0;" % (T, nm))
            L.append("  if (owner) %s = ((const %s*)a.out[%d])[((a.out_pos0[%d] + a.out_store[%d] * 8 - %d) %% "
                     "a.out_store[%d]) * a.out_rs[%d] + myrow];" % (nm, T, k, k, k, d, k, k))
    pw_nsq = sorted({v for ph in pr.phases for v in ph["ins"] if v in pr.nsq})
    for v in pw_nsq:
        es_one = spec.lens[v] == 1
        L.append("  %s own_%d = 0;" % (T, v))
        L.append("  if (%s) own_%d = ((const %s*)a.nsq[%d])[%s];"
                 % ("true" if es_one and pr.zero_d else "owner", v, T, pr.nsq[v],
                    "0" if es_one else "myrow * a.nsq_es[%d]" % pr.nsq[v]))
    pw_seq = sorted({v for ph in pr.phases for v in ph["ins"] if v in pr.seq})
    for v in pw_seq:
        s = pr.seq[v]
        idx = "0" if spec.lens[v] == 1 else "myrow * a.seq_es[%d]" % s
        L.append("  %s nxt_%d = 0, own_%d = 0;" % (T, v, v))
        L.append("  if (owner && a.T > 0) nxt_%d = ((const %s*)a.seq[%d])[%s];" % (v, T, s, idx))
    for ph in pr.phases:
        for o in ph["outs"]:
            L.append("  %s own_%d = 0;" % (T, o))
        if ph.get("reduce"):
            L.append("  %s own_%d = 0, own_%d = 0;" % (T, ph["reduce"]["u"], ph["reduce"]["out"]))
    if pr.cond is not None:
        L.append("  i64 steps_ = a.T;           // do-while: steps run (every workgroup stops alike)")
    L.append("  __syncthreads();")
    L.append("  for (i64 t = 0; t < a.T; ++t) {")
    L.append("    const int par = (int)(t & 1);")
    for v in pw_seq:
        s = pr.seq[v]
        idx = "0" if spec.lens[v] == 1 else "myrow * a.seq_es[%d]" % s
        L.append("    own_%d = nxt_%d;" % (v, v))
        L.append("    if (owner && t + 1 < a.T) nxt_%d = ((const %s*)a.seq[%d])[(t + 1) * a.seq_ts[%d] + %s];"
                 % (v, T, s, s, idx))
    def gather(xo, lp, so, K, step_expr, ind, delayed):
        """Poll the granules of an exchanged vector until they carry the producing step's tag and
        stage the values in Vl[par][so ..]; returns the new ``delayed`` (the first poll of a phase
        is held back)."""
        PT = 64 * spec.var["pollw"]
        KG = GPV * K                   # granules of this vector (float64: hi / lo halves)
        NGp = (KG + PT - 1) // PT
        pg = "" if PT == BLOCK else "if (threadIdx.x < %d) " % PT
        L.append(ind + "const u64* src = a.xch + %d + (%s & 3) * %d;" % (xo, step_expr, lp))
        L.append(ind + "const unsigned want = base + (unsigned)%s + 1u;" % step_expr)
        L.append(ind + pg + "{")
        L.append(ind + "u64 g[%d];" % NGp)
        # a poll that comes back without the tags costs a whole further round trip (~0.7 us)
        # and its traffic slows the very stores it waits for: hold the first one back until it
        # can succeed (r04: config 4 B = 1 4.59 -> 4.05 us per step with 12 x 64 cycles)
        if not delayed:
            for _ in range(spec.var["delay"] // 15):
                L.append(ind + "__builtin_amdgcn_s_sleep(15);")
            if spec.var["delay"] % 15:
                L.append(ind + "__builtin_amdgcn_s_sleep(%d);" % (spec.var["delay"] % 15))
            delayed = True
        L.append(ind + "for (int spin = 0;; ++spin) {")
        L.append(ind + "  bool ok = true;")
        for q in range(NGp):
            guard = "" if (q + 1) * PT <= KG else "if (threadIdx.x + %d < %d) " % (q * PT, KG)
            L.append(ind + "  %s{ g[%d] = __hip_atomic_load(src + threadIdx.x + %d, %s); "
                     "ok = ok && ((unsigned)(g[%d] >> 32) == want); }" % (guard, q, q * PT, AG, q))
        L.append(ind + "  if (ok) break;")
        # bounded spin; once any workgroup has given up every later wait ends within 256 polls
        L.append(ind + "  if (spin > %d || ((spin & 255) == 255 && __hip_atomic_load(errp, %s) != 0u)) "
                 "{ __hip_atomic_store(errp, 1u, %s); break; }" % (SPIN_LIMIT, AG, AG))
        if spec.var["sleep"]:
            L.append(ind + "  __builtin_amdgcn_s_sleep(%d);" % spec.var["sleep"])
        L.append(ind + "}")
        for q in range(NGp):
            guard = "" if (q + 1) * PT <= KG else "if (threadIdx.x + %d < %d) " % (q * PT, KG)
            if F64:
                # granule 2k = {tag, high word}, 2k + 1 = {tag, low word}: two 4-byte LDS stores
                # rebuild the double in place (little endian: low word first)
                L.append(ind + "%s((unsigned*)(Vl[par] + %d))[(threadIdx.x + %d) ^ 1] = (unsigned)g[%d];"
                         % (guard, so, q * PT, q))
            else:
                L.append(ind + "%sVl[par][%d + threadIdx.x + %d] = __uint_as_float((unsigned)g[%d]);"
                         % (guard, so, q * PT, q))
        return delayed

    def emit_dot(pi, d, a_, x):
        """accumulate dot d of phase pi: rows of matrix a_ (LDS or VGPRs) times the staged vector x"""
        K = spec.Ks[a_]
        K4 = K // VEC
        if x in pr.nsq:
            vsrc = "Il + %d" % inv_stage[x]
        else:
            kind = "prev" if x in pr.state else ("cur" if x not in pr.seq else "glob")
            vsrc = "Vl[par] + %d" % stage[(x, kind)]
        for i in range(RPW):
            L.append("    %s acc%d_%d_%d = 0;" % (T, pi, d, i))
        if a_ in in_reg:
            slot = pr.mats[a_]
            L.append("    {")
            L.append("      const %s* vx = %s;" % (T, vsrc))
            for q in range(K4 // 64):
                L.append("      { const TV xv = *(const TV*)(vx + %d * (lane + %d));" % (VEC, 64 * q))
                for i in range(RPW):
                    L.append("        acc%d_%d_%d += %s;" % (pi, d, i, vdot("wr%d_%d_%d" % (slot, i, q), "xv")))
                L.append("      }")
            L.append("    }")
            return
        L.append("    {")
        L.append("      const %s* vx = %s;" % (T, vsrc))
        L.append("      const %s* wr = Wl + %d + (wave * %d) * %d;" % (T, woff[a_], RPW, K))
        L.append("#pragma unroll")
        L.append("      for (int k4 = lane; k4 < %d; k4 += 64) {" % K4)
        L.append("        const TV xv = *(const TV*)(vx + %d * k4);" % VEC)
        for i in range(RPW):
            L.append("        { const TV wv = *(const TV*)(wr + %d + %d * k4); acc%d_%d_%d += %s; }"
                     % (i * K, VEC, pi, d, i, vdot("wv", "xv")))
        L.append("      }")
        L.append("    }")

    staged_this_step = set()
    for pi, ph in enumerate(pr.phases):
        L.append("    // ---- phase %d" % pi)
        stamp("p%d start" % pi)
        # -- dots whose vector operand is ALREADY staged (the state gathered by an earlier phase of
        #    this step, an invariant vector) go in front of this phase's hand-off: their LDS reads and
        #    FMAs run while the granules this phase waits for are still on their way (GRU: the z-gate
        #    product U_z h sits in the phase that waits for r * h)
        early_dots = set()
        if int(knobs.get("SP_EARLYDOTS")):
            for d, (a_, x) in enumerate(ph["dots"]):
                if x in pr.nsq or (x, "prev" if x in pr.state else ("cur" if x not in pr.seq else "glob")) \
                        in staged_this_step:
                    early_dots.add(d)
            if len(early_dots) == len(ph["dots"]):
                early_dots = set()            # nothing to wait for in this phase: the plain order
            for d, (a_, x) in enumerate(ph["dots"]):
                if d in early_dots:
                    emit_dot(pi, d, a_, x)
            if early_dots:
                stamp("p%d early dots" % pi)
        # -- gather the dot vectors that are not staged yet
        need_sync = False
        delayed = False     # the first exchanged operand of a phase waits before its first poll
        for a_, x in ph["dots"]:
            if x in pr.nsq:
                continue
            kind = "prev" if x in pr.state else ("cur" if x not in pr.seq else "glob")
            if (x, kind) in staged_this_step:
                continue
            staged_this_step.add((x, kind))
            need_sync = True
            # the vector's TRUE length (the LDS image is Ks[a_] long: zero-padded to whole 16-byte
            # vectors when the contraction length is not a multiple of them, see _scan_persist)
            K = spec.lens[x]
            so = stage[(x, kind)]
            NG = (K + BLOCK - 1) // BLOCK
            if kind == "glob":
                s = pr.seq[x]
                L.append("    for (int k = threadIdx.x; k < %d; k += %d) Vl[par][%d + k] = "
                         "((const %s*)a.seq[%d])[t * a.seq_ts[%d] + k * a.seq_es[%d]];"
                         % (K, BLOCK, so, T, s, s, s))
                continue
            src = pr.new_of_state.get(x, x)
            xo, lp = xoff[src]
            if kind == "prev":
                k_out = pr.state[x]
                L.append("    if (t == 0) {")
                L.append("      const %s* ini = (const %s*)a.out[%d] + ((a.out_pos0[%d] + a.out_store[%d] - 1) %% "
                         "a.out_store[%d]) * a.out_rs[%d];" % (T, T, k_out, k_out, k_out, k_out, k_out))
                L.append("      for (int k = threadIdx.x; k < %d; k += %d) Vl[par][%d + k] = ini[k];" % (K, BLOCK, so))
                L.append("    } else {")
                step_expr, ind = "(t - 1)", "      "
            else:
                L.append("    {")
                step_expr, ind = "t", "      "
            delayed = gather(xo, lp, so, K, step_expr, ind, delayed)
            L.append(ind + "}")
            L.append("    }")
        if need_sync:
            stamp("p%d tags seen, staged" % pi)
            L.append("    __syncthreads();")
            stamp("p%d barrier" % pi)
        # -- row dots: wavefront `wave` owns rows wave*RPW .. +RPW
        D = len(ph["dots"])
        for d, (a_, x) in enumerate(ph["dots"]):
            if d not in early_dots:
                emit_dot(pi, d, a_, x)
        if D:
            L.append("    for (int s = 32; s > 0; s >>= 1) {")
            for d in range(D):
                for i in range(RPW):
                    L.append("      acc%d_%d_%d += shfl_xor_<%s>(acc%d_%d_%d, s);" % (pi, d, i, T, pi, d, i))
            L.append("    }")
        if D:
            stamp("p%d dots + fold" % pi)
        for d in range(D):
            sel = "acc%d_%d_%d" % (pi, d, RPW - 1)
            for i in range(RPW - 2, -1, -1):
                sel = "(lane == %d ? acc%d_%d_%d : %s)" % (i, pi, d, i, sel)
            L.append("    const %s dot_%d_%d = %s;" % (T, pi, d, sel))
        # -- epilogue on the row owners (a phase of 0-d values: on every lane — what follows from a
        #    reduction, the do-while condition included, must be uniform)
        red = ph.get("reduce")
        L.append("    {" if ph.get("scalar_phase") else "    if (owner) {")
        ins = ["dot_%d_%d" % (pi, d) for d in range(D)] + ["own_%d" % v for v in ph["ins"]]
        lines, outs, odts = cg.emit_scalar_body(ph["scalar"], ins, [spec.dtype] * len(ins),
                                                indent="      ", suffix="_p%d" % pi)
        L.extend(lines)
        pub = list(zip(ph["outs"], ph["out_refs"])) + ([(red["u"], red["ref"])] if red else [])
        for k, (o, ri) in enumerate(pub):
            L.append("      own_%d = %s;" % (o, cg._cast(outs[ri], odts[ri], spec.dtype)))
            if o in pr.zero_d:
                # a 0-d result (every lane holds it): one lane of workgroup 0 stores it
                for kind, j in out_of.get(o, []):
                    L.append("      if (blockIdx.x == 0 && threadIdx.x == 0) ((%s*)a.out[%d])[((a.out_pos0[%d] + t) "
                             "%% a.out_store[%d]) * a.out_rs[%d]] = own_%d;" % (T, j, j, j, j, o))
                continue
            if o in xoff:
                xo, lp = xoff[o]
                if F64:
                    L.append("      { const u64 bits = (u64)__double_as_longlong(own_%d), tg = (u64)(base + (unsigned)t + 1u) << 32;" % o)
                    L.append("        __hip_atomic_store(a.xch + %d + (t & 3) * %d + 2 * myrow, tg | (bits >> 32), %s);" % (xo, lp, AG))
                    L.append("        __hip_atomic_store(a.xch + %d + (t & 3) * %d + 2 * myrow + 1, tg | (bits & 0xFFFFFFFFull), %s); }" % (xo, lp, AG))
                else:
                    L.append("      __hip_atomic_store(a.xch + %d + (t & 3) * %d + myrow, "
                             "((u64)(base + (unsigned)t + 1u) << 32) | (u64)__float_as_uint(own_%d), %s);"
                             % (xo, lp, o, AG))
            for kind, j in out_of.get(o, []):
                L.append("      ((%s*)a.out[%d])[((a.out_pos0[%d] + t) %% a.out_store[%d]) * a.out_rs[%d] + myrow] = own_%d;"
                         % (T, j, j, j, j, o))
        L.append("    }")
        stamp("p%d epilogue, published" % pi)
        if red:
            # -- the reduction: gather the whole vector the owners just published, fold it in one
            #    fixed order (lane l folds elements l, l + 64, ...; butterfly over the wavefront:
            #    every lane of every wavefront of every workgroup ends with the same value)
            xo, lp = xoff[red["u"]]
            so = stage[(red["u"], "cur")]
            AT = cg.RTYPE[red["acc"]]
            comb = lambda a_, b_, _r=red: cg.red_combine(_r["op"], _r["acc"], a_, b_)   # noqa: E731
            L.append("    {")
            gather(xo, lp, so, M, "t", "      ", False)
            L.append("      }")
            L.append("    }")
            L.append("    __syncthreads();")
            L.append("    {")
            L.append("      %s rr = (%s)%s;" % (AT, AT, cg.red_identity(red["op"], red["acc"])))
            L.append("      for (int k = lane; k < %d; k += 64) rr = %s;" % (M, comb("rr", "(%s)Vl[par][%d + k]" % (AT, so))))
            L.append("      for (int s = 32; s > 0; s >>= 1) rr = %s;" % comb("rr", "shfl_xor_<%s>(rr, s)" % AT))
            L.append("      own_%d = (%s)rr;" % (red["out"], T))
            L.append("    }")
            for kind, j in out_of.get(red["out"], []):
                L.append("    if (blockIdx.x == 0 && threadIdx.x == 0) ((%s*)a.out[%d])[((a.out_pos0[%d] + t) "
                         "%% a.out_store[%d]) * a.out_rs[%d]] = own_%d;" % (T, j, j, j, j, red["out"]))
    for v, nv in pr.new_of_state.items():
        k = pr.state[v]
        for d in range(pr.depth.get(k, 1), 1, -1):      # shift the owner's history, oldest first
            L.append("    %s = %s;" % (hist[(k, d)], hist[(k, d - 1)] if d > 2 else "own_%d" % v))
        L.append("    own_%d = own_%d;" % (v, nv))
    if pr.cond is not None:
        L.append("    if (own_%d != 0) { steps_ = t + 1; break; }" % pr.cond)
    L.append("  }")
    if spec.trace:
        L.append("  if (tr_on) { tr[%d] = __builtin_amdgcn_s_memtime(); tr[%d] = __builtin_amdgcn_s_memrealtime(); }"
                 % (TRACE_NT * TRACE_MARKS + 2, TRACE_NT * TRACE_MARKS + 3))
        spec.marks = list(marks)
    # every workgroup read `base` before workgroup 0 can get here (it gathered, in its last
    # step, a vector that every workgroup published after reading `base`; T >= 2)
    L.append("  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.ctl, base + (unsigned)a.T, %s);" % AG)
    if pr.cond is not None:
        L.append("  if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl[4] = (unsigned)steps_;   // the one host read")
    L.append("}")
    return "\n".join(L) + "\n", (name,)


REG_BUDGET = 160      # VGPRs per lane that may hold matrix rows


def choose_rows(M, Ks, stage_floats, cu_count=256, itemsize=4):
    """Geometry of the persistent kernel: rows of every matrix per workgroup (R), wavefronts per
    workgroup, grid size and where each matrix's rows live ("lds" / "reg").

    Measured on MI355X (BASELINE config 4, T = 512, H = 1024; ``tools/sp_sweep.sh``): 256
    workgroups x 4 rows 6.1 us per step, **128 x 8 rows 4.6 us**, 64 x 16 rows (all rows in
    VGPRs) 4.9 us, 32 x 32 rows 5.9 us — the exchange latency falls with the number of pollers
    down to ~128 workgroups, below that the per-workgroup dot time grows faster.  So: half the CUs
    when the rows fit, else one workgroup per CU, else fewer; rows go to LDS first and spill into
    VGPRs (K % 256 == 0, REG_BUDGET per lane) — that is what lets H = 2048 (48 MiB of weights)
    stay on chip.  ``AESARA_HIP_SCAN_ROWS`` / ``AESARA_HIP_SCAN_WAVES`` override."""
    env = knobs.get("SCAN_ROWS")
    nw = int(knobs.get("SCAN_WAVES"))
    half = max(cu_count // 2, 1)
    # (round 4, first poll of a hand-off held back + 4 polling waves: 256 x 4 rows 3.42 us per step,
    # 128 x 8 rows 3.56 — one workgroup per CU first now, then half the CUs, then fewer)
    cands = [int(env)] if env else [-(-M // cu_count), -(-M // half), -(-M // (half // 2 or 1)),
                                    -(-M // (half // 4 or 1))]
    for R in cands:
        R = -(-max(R, nw) // nw) * nw
        G = -(-M // R)
        if G > cu_count:
            continue
        rpw = R // nw
        place, regs, lds = {}, 0, stage_floats * itemsize
        for av, K in sorted(Ks.items(), key=lambda t: -t[1]):
            if lds + R * K * itemsize <= LDS_BUDGET:
                place[av], lds = "lds", lds + R * K * itemsize
            elif K % 256 == 0 and regs + rpw * (K // 64) * (itemsize // 4) <= REG_BUDGET:
                place[av], regs = "reg", regs + rpw * (K // 64) * (itemsize // 4)
            else:
                place = None
                break
        if place is not None:
            return R, nw, G, place
    return None
