"""K10e — a Scan whose step is purely ELEMENT-WISE as ONE kernel launch (round 4).

Replaces, for this class, what ``Scan.perform`` drives from the host (scan/op.py:1673) through
``scan_perform.pyx:309-541``: per step, slice the sequences and the taps (:321-340), run the inner
function, store the outputs into the (circular) output buffers (:430-520), evaluate the ``until``
condition of a do-while Scan (:424-426).  When every fused step of the inner graph is an Elemwise
over values of ONE shape (cumulative sums / products, filters with taps, running statistics,
counters, a scalar recurrence with a stop condition ...) element e of every value depends only on
element e of the others, so no value ever has to cross between threads: thread e runs the whole
recurrence for its element — ``T`` steps in one launch, the taps in registers (a shift per step),
the sequences read eight steps ahead of their use, every step's outputs stored as they are made.
No exchange, no polling, no co-residency requirement (any grid size runs).

Covered: sequences (incl. hoisted sequence-only rows), sit-sot / mit-sot outputs with taps down to
-8, nit-sot outputs, shared outputs (``n_shared_outs``: a recurrent value without a history),
non-sequences (per element or one broadcast scalar), any dtype the Elemwise generator covers
(float32/64, (u)int8-64, bool), and **do-while** (``as_while``) for recurrences of ONE element (the
condition is a 0-d value: with more elements it would be a reduction): the thread stops at the
first step whose condition is true and stores the number of steps it ran in a device word the
executor reads once after the launch (the trip count sizes the outputs, scan/op.py:2139-2159).
**mit-mot** outputs with any non-negative taps (what ``Scan.L_op`` builds to propagate gradients,
scan/op.py:2379 — [0, 1] -> [1] for a one-tap recurrence, [0, 1, 2] -> [1, 2] for taps [-1, -2], ...):
step t reads rows t + tap of the buffer and overwrites rows t + out-tap (scan_perform.pyx:343-352,
:437-452); the thread keeps the window of rows t .. t + max tap of its element in registers, the row
that enters the window is read ahead like a sequence row (no earlier step writes it), every out-tap
is stored as it is made.
**Full reductions inside the step** (round 5: ``x / abs(x).sum()``, ``until(at_all(x > u))``,
``until(max(abs(x_new - x)) < tol)``, a per-step energy handed out as a nit-sot output — the
reference runs them in the same loop, scan_perform.pyx:309-541, condition :424-426): the loop then has
values of TWO kinds, per-element ones (shape S) and ones every element sees alike (0-d results of the
reductions and whatever is computed from them, ``ProgramEw.cls``).  Such a loop runs as ONE workgroup
of up to 1024 threads, thread e still owning element e: a reduction folds inside the wavefronts by
DPP (fixed tree), the wavefronts' partials cross through LDS (two parities: one barrier per
reduction) and every thread folds them in wave order — all threads hold the identical value, so a
do-while condition computed from it is uniform (the vector-state do-while).  0-d recurrent states /
outputs next to the vectors are kept by every thread and stored by thread 0.
Not covered (launch-list path, ``PlanExecutor.scan_modes`` says why): steps with dots / partial
reductions / indexing, reductions over more than 1024 elements or in gradient (mit-mot) loops, values
of different non-scalar shapes inside one step.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json

from . import codegen as cg

EW_MAXSEQ, EW_MAXNSQ, EW_MAXOUT, EW_MAXSH = 16, 16, 16, 8
AHEAD = 8                 # steps whose sequence rows are loaded before the first of them is used


class EwScanArgs(C.Structure):
    _fields_ = [
        ("T", C.c_int64), ("n", C.c_int64),
        ("seq", C.c_void_p * EW_MAXSEQ), ("seq_ts", C.c_int64 * EW_MAXSEQ),
        ("nsq", C.c_void_p * EW_MAXNSQ), ("nsq_es", C.c_int64 * EW_MAXNSQ),
        ("out", C.c_void_p * EW_MAXOUT), ("out_rs", C.c_int64 * EW_MAXOUT),
        ("out_store", C.c_int64 * EW_MAXOUT), ("out_pos0", C.c_int64 * EW_MAXOUT),
        ("sh_in", C.c_void_p * EW_MAXSH), ("sh_out", C.c_void_p * EW_MAXSH),
        ("ctl", C.c_void_p),
    ]


EW_STRUCT = r"""
#define EW_MAXSEQ %d
#define EW_MAXNSQ %d
#define EW_MAXOUT %d
#define EW_MAXSH %d
struct EwScanArgs {
  i64 T; i64 n;
  const void* seq[EW_MAXSEQ]; i64 seq_ts[EW_MAXSEQ];
  const void* nsq[EW_MAXNSQ]; i64 nsq_es[EW_MAXNSQ];
  void* out[EW_MAXOUT]; i64 out_rs[EW_MAXOUT]; i64 out_store[EW_MAXOUT]; i64 out_pos0[EW_MAXOUT];
  const void* sh_in[EW_MAXSH]; void* sh_out[EW_MAXSH];
  unsigned* ctl;
};
""" % (EW_MAXSEQ, EW_MAXNSQ, EW_MAXOUT, EW_MAXSH)

_ALIAS_OPS = ("SpecifyShape", "ViewOp", "ScalarFromTensor", "TensorFromScalar", "DeepCopyOp")


class ProgramEw:
    """The loop in terms of inner-plan variable ids.

    ``seq``: var -> sequence slot; ``tap``: var -> (recurrent output k, depth d >= 1);
    ``shared``: var -> shared slot; ``nsq``: var -> invariant slot; ``steps``: the Elemwise steps
    in order; ``rec_new`` / ``nit_new`` / ``sh_new``: the variables holding each output's new
    value; ``cond``: the do-while condition variable or None; ``depth``: k -> deepest tap."""

    def __init__(self):
        self.seq, self.tap, self.shared, self.nsq = {}, {}, {}, {}
        # mit-mot groups: ``mm`` = [(in taps, out taps)], ``mm_in``: var -> (group, tap),
        # ``mm_new``: per group [(out tap, var)] in the order of the inner outputs
        self.mm, self.mm_in, self.mm_new = [], {}, []
        self.steps, self.rec_new, self.nit_new, self.sh_new = [], [], [], []
        self.cond, self.depth, self.dtype_of = None, {}, {}
        self.as_while = False
        # ``cls``: var -> 1 for a value every element sees alike (0-d / all dims broadcastable:
        # reduction results and what follows from them), 0 (or absent) for a per-element value of
        # shape S; only filled when the loop mixes the two (``wg``: it then runs as ONE workgroup)
        self.cls, self.nred, self.wg = {}, 0, False


def analyze(inner, p, n_pre):
    """(ProgramEw, None) or (None, reason)."""
    plan = inner.plan
    mm_in = [list(t) for t in p.get("mit_mot_in_slices", [])]
    mm_out = [list(t) for t in p.get("mit_mot_out_slices", [])]
    if len(mm_in) != len(mm_out) or any(
            any(x < 0 or x > 8 for x in ti + to) or len(set(ti)) != len(ti) or len(set(to)) != len(to)
            for ti, to in zip(mm_in, mm_out)):
        return None, "mit-mot taps negative / deeper than 8 / repeated"
    n_seqs, n_sh, n_nit = p["n_seqs"], p.get("n_shared_outs", 0), p["n_nit_sot"]
    taps = [list(t) for t in p["mit_sot_in_slices"]] + [list(t) for t in p["sit_sot_in_slices"]]
    if any(any(x >= 0 for x in t) or len(set(t)) != len(t) or min(t) < -8 for t in taps):
        return None, "taps non-negative / deeper than 8"
    pr = ProgramEw()
    pr.as_while = bool(p.get("as_while", False))
    ins = list(plan.inputs)
    n_fixed = len(ins) - n_pre
    if n_seqs + n_pre > EW_MAXSEQ or n_sh > EW_MAXSH:
        return None, "too many sequences / shared outputs"
    for s, v in enumerate(ins[:n_seqs]):
        pr.seq[v] = s
    idx = n_seqs
    for g_, (ti, to) in enumerate(zip(mm_in, mm_out)):
        pr.mm.append((ti, to))
        for tap in ti:
            pr.mm_in[ins[idx]] = (g_, tap)
            idx += 1
    for k, tk in enumerate(taps):
        for tap in tk:
            pr.tap[ins[idx]] = (k, -tap)
            idx += 1
        pr.depth[k] = -min(tk)
    pr.n_rec_inputs = idx - n_seqs
    for m in range(n_sh):
        pr.shared[ins[idx]] = m
        idx += 1
    inv = ins[idx:n_fixed]
    for j, v in enumerate(ins[n_fixed:]):
        pr.seq[v] = n_seqs + j
    inv_set = set(inv)
    alias, produced = {}, set()

    def res(v):
        while v in alias:
            v = alias[v]
        return v
    def scalar_like(v):
        sh = plan.vars[v].shape
        return plan.vars[v].ndim == 0 or all(x == 1 for x in sh)
    # values of two kinds (see the module docstring) only when the loop holds a reduction or mixes
    # 0-d values with arrays; a loop of one rank keeps the plain one-thread-per-element form
    ranks = {plan.vars[v].ndim for v in ins[:n_fixed - len(inv)] + ins[n_fixed:]}
    # an ``IfElse`` on a 0-d condition between values the step computes element-wise anyway is a
    # ``switch`` per element (ifelse.py:61 is lazy to save the untaken branch's WORK; both branches
    # are pure Elemwise here and one thread evaluates both in registers)
    from .fusion import Step
    steps_ = []
    for st in inner.steps:
        if st.kind == "node" and st.node.op == "IfElse":
            n_o = len(st.outputs)
            if len(st.inputs) != 1 + 2 * n_o or plan.vars[st.inputs[0]].ndim != 0:
                return None, "step kind node (IfElse)"
            for k, o in enumerate(st.outputs):
                a_, b_ = st.inputs[1 + k], st.inputs[1 + n_o + k]
                if not (plan.vars[a_].dtype == plan.vars[b_].dtype == plan.vars[o].dtype):
                    return None, "IfElse branches of another dtype"
                steps_.append(Step("elemwise", [st.inputs[0], a_, b_], [o], {
                    "n_in": 3, "nodes": [{"op": "switch", "in": [["i", 0], ["i", 1], ["i", 2]],
                                          "dtype": plan.vars[o].dtype}], "out": [["t", 0]]}, out_refs=[0]))
        else:
            steps_.append(st)
    for st in steps_:
        if st.kind == "reduce":
            pr.nred += 1
        ranks.update(plan.vars[o].ndim for o in st.outputs)
    pr.wg = pr.nred > 0 or (len(ranks) > 1 and 0 in ranks)
    if pr.wg and pr.mm:
        return None, "reductions / 0-d values in a gradient (mit-mot) loop"
    if pr.wg:
        for v in ins[:n_fixed - len(inv)] + ins[n_fixed:]:
            pr.cls[v] = 1 if scalar_like(v) else 0
        for v in inv:
            pr.cls[v] = 1 if plan.vars[v].ndim == 0 else 0       # (one element at run time: ``bc``)
    for st in steps_:
        if st.kind == "node" and st.node.op in _ALIAS_OPS:
            alias[st.outputs[0]] = st.inputs[0]
            continue
        if st.kind == "node" and st.node.op == "DimShuffle" and \
                plan.vars[st.outputs[0]].ndim == plan.vars[st.inputs[0]].ndim == 0:
            alias[st.outputs[0]] = st.inputs[0]
            continue
        if st.kind == "node" and st.node.op == "DimShuffle" and pr.wg and \
                pr.cls.get(res(st.inputs[0])) == 1 and scalar_like(st.outputs[0]):
            alias[st.outputs[0]] = st.inputs[0]      # a 0-d value made broadcastable against S
            continue
        red = None
        if st.kind == "reduce" and st.reduce is not None and not (st.post or st.fallback or st.dots
                                                                  or st.extra.get("xprog")):
            nd_in = max([plan.vars[res(v)].ndim for v in st.inputs] or [0])
            ax = st.reduce["axis"]
            if st.reduce["scalar_op"] not in ("add", "mul", "maximum", "minimum", "and", "or", "xor"):
                return None, "reduction with %s" % st.reduce["scalar_op"]
            if nd_in == 0 or (ax is not None and sorted(ax) != list(range(nd_in))) or \
                    plan.vars[st.reduce["out"]].ndim != 0:
                return None, "partial reduction inside the step"
            red = {"op": st.reduce["scalar_op"], "acc": st.reduce["acc_dtype"], "ref": st.reduce["ref"],
                   "out": st.reduce["out"]}
            if red["acc"] not in cg.CTYPE:
                return None, "a dtype without kernels"
        elif st.kind != "elemwise" or st.reduce is not None or st.post or st.fallback or st.extra.get("xprog"):
            return None, f"step kind {st.kind} ({st.node.op if st.node else ''})"
        st_in = [res(v) for v in st.inputs]
        for v in st_in:
            if plan.vars[v].const is not None:
                return None, "array constant inside the step"
            if v in inv_set:
                if v not in pr.nsq:
                    if len(pr.nsq) >= EW_MAXNSQ:
                        return None, "too many invariant operands"
                    pr.nsq[v] = len(pr.nsq)
            elif not (v in pr.seq or v in pr.tap or v in pr.mm_in or v in pr.shared or v in produced):
                return None, "operand of unknown origin"
        step = {"ins": st_in, "outs": list(st.outputs), "scalar": st.scalar,
                "out_refs": list(st.out_refs)}
        if pr.wg:
            kinds = [pr.cls.get(v, 0) for v in st_in]
            oc = 1 if (kinds and all(kinds)) else 0
            if not kinds:
                oc = 1 if all(scalar_like(o) for o in st.outputs) else 0
            for o in st.outputs:
                pr.cls[o] = oc
            if red is not None:
                if oc == 1:
                    return None, "reduction of a 0-d value"
                step["reduce"] = red
                pr.cls[red["out"]] = 1
                produced.add(red["out"])
        pr.steps.append(step)
        produced.update(st.outputs)
    n_rec = len(taps)
    n_mmo = sum(len(to) for _ti, to in pr.mm)
    want = n_mmo + n_rec + n_nit + n_sh + (1 if pr.as_while else 0)
    if len(plan.outputs) != want or len(pr.mm) + n_rec + n_nit > EW_MAXOUT or n_mmo + n_rec + n_nit + n_sh == 0:
        return None, "output count"
    outs = [res(o) for o in plan.outputs]

    def legal_src(v):
        return v in produced or v in pr.seq or v in pr.tap or v in pr.mm_in or v in pr.shared or \
            v in pr.nsq or v in inv_set
    for v in outs:
        if not legal_src(v):
            return None, "a step output is not computed by a fused step"
        if v in inv_set and v not in pr.nsq:
            pr.nsq[v] = len(pr.nsq)
    o_ = 0
    for _ti, to in pr.mm:
        pr.mm_new.append([(tap, outs[o_ + j]) for j, tap in enumerate(to)])
        o_ += len(to)
    outs = outs[n_mmo:]
    pr.rec_new = outs[:n_rec]
    pr.nit_new = outs[n_rec:n_rec + n_nit]
    pr.sh_new = outs[n_rec + n_nit:n_rec + n_nit + n_sh]
    pr.cond = outs[-1] if pr.as_while else None
    pr.dtype_of = {v: plan.vars[v].dtype for v in plan.vars}
    if any(dt not in cg.CTYPE for dt in pr.dtype_of.values() if dt):
        return None, "a dtype without kernels"
    return pr, None


class SpecEw:
    def __init__(self, prog: ProgramEw, plan, out_dtypes, sh_dtypes, broadcast_nsq):
        self.prog, self.plan = prog, plan
        self.out_dtypes, self.sh_dtypes = list(out_dtypes), list(sh_dtypes)
        self.bc = tuple(sorted(broadcast_nsq))       # invariant slots read as one scalar

    def key(self):
        pr = self.prog
        blob = ["se3", pr.mm, sorted((k, list(v)) for k, v in pr.mm_in.items()), pr.mm_new,
                [[s["ins"], s["outs"], s["scalar"], s["out_refs"]] for s in pr.steps],
                sorted(pr.seq.items()), sorted((k, list(v)) for k, v in pr.tap.items()),
                sorted(pr.shared.items()), sorted(pr.nsq.items()), pr.rec_new, pr.nit_new,
                pr.sh_new, pr.cond, sorted(pr.depth.items()), self.out_dtypes, self.sh_dtypes,
                self.bc, sorted((k, v) for k, v in pr.dtype_of.items())]
        if pr.wg:       # (the one-workgroup form: value kinds and reductions are part of the kernel)
            blob += ["wg1", sorted(pr.cls.items()), [s.get("reduce") for s in pr.steps]]
        return hashlib.sha256(json.dumps(blob, sort_keys=True).encode()).hexdigest()[:24]


def slot_kinds(pr: ProgramEw):
    """(kinds of the recurrent outputs, of the nit-sot outputs, of the shared values): 1 = one
    value for all elements (a 0-d output: stored by thread 0), 0 = per element."""
    if not pr.wg:
        return [0] * len(pr.rec_new), [0] * len(pr.nit_new), [0] * len(pr.sh_new)
    rec = [0] * len(pr.rec_new)
    for v, (k, _d) in pr.tap.items():
        rec[k] = pr.cls.get(v, 0)
    sh = [0] * len(pr.sh_new)
    for v, m in pr.shared.items():
        sh[m] = pr.cls.get(v, 0)
    return rec, [pr.cls.get(v, 0) for v in pr.nit_new], sh


def generate(spec: SpecEw):
    """Return (source, (kernel name,))."""
    pr = spec.prog
    name = "se_" + spec.key()
    dt = pr.dtype_of
    CT, RT = cg.CTYPE, cg.RTYPE
    wg = pr.wg
    rec_kind, nit_kind, sh_kind = slot_kinds(pr)
    E = "el" if wg else "e"                # element a thread LOADS (clamped in the one-workgroup form)

    def kind(v):
        return pr.cls.get(v, 0) if wg else 0

    def at(c):                             # "+ element" of an address
        return "" if c else " + " + E

    def guard(c):                          # who stores a value of kind c
        return ("if (e == 0) " if c else "if (live) ") if wg else ""
    L = [cg.PRELUDE, EW_STRUCT]
    L.append('extern "C" __global__ __launch_bounds__(%d) void %s(EwScanArgs a) {' % (1024 if wg else 256, name))
    if wg:
        # ONE workgroup, thread e = element e; threads past the state stay (barriers), load the
        # last element and never store
        L.append("  const i64 e = threadIdx.x;")
        L.append("  const bool live = e < a.n;")
        L.append("  const i64 el = live ? e : a.n - 1;")
        L.append("  const unsigned nw_ = blockDim.x >> 6, wv_ = threadIdx.x >> 6, ln_ = threadIdx.x & 63u;")
        ri = 0
        for st in pr.steps:
            red = st.get("reduce")
            if red is not None:
                acc_t = RT[red["acc"]]
                L.append("  __shared__ %s smr%d[2][16];" % (acc_t if acc_t != "bool" else "unsigned char", ri))
                ri += 1
    else:
        L.append("  const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;")
        L.append("  if (e >= a.n) return;")
    mm_vals = [v for grp in pr.mm_new for _tap, v in grp]
    used_seq = sorted({v for s in pr.steps for v in s["ins"] if v in pr.seq} |
                      {v for v in mm_vals + pr.rec_new + pr.nit_new + pr.sh_new +
                       ([pr.cond] if pr.cond is not None else []) if v in pr.seq}, key=lambda v: pr.seq[v])
    # output slots (the order of the Scan's outer outputs): mit-mot groups, recurrent outputs, nit-sot
    n_mm = len(pr.mm)
    mm_top = [max(ti + to) for ti, to in pr.mm]         # the window of group g: rows t .. t + mm_top[g]

    def rd(expr, d):                       # a stored element -> register value
        return "(%s != 0)" % expr if d == "bool" else expr
    # invariants
    for v, j in sorted(pr.nsq.items(), key=lambda kv: kv[1]):
        off = "0" if (j in spec.bc or kind(v)) else "%s * a.nsq_es[%d]" % (E, j)
        L.append("  const %s nv%d = %s;" % (RT[dt[v]], j, rd("((const %s*)a.nsq[%d])[%s]" % (CT[dt[v]], j, off), dt[v])))
    # recurrent state registers (tap -1 .. -depth) from the output buffers, shared values
    n_rec, n_nit = len(pr.rec_new), len(pr.nit_new)
    for g_ in range(n_mm):
        # a mit-mot buffer is not circular: row of step t, tap j = pos0 + t + j; the rows below the
        # top of the window start in registers, the top row enters per step (read ahead below)
        odt = spec.out_dtypes[g_]
        L.append("  %s* const mb%d = (%s*)a.out[%d] + a.out_pos0[%d] * a.out_rs[%d] + e;" % (CT[odt], g_, CT[odt], g_, g_, g_))
        for j in range(mm_top[g_]):
            L.append("  %s w%d_%d = %s;" % (RT[odt], g_, j, rd("mb%d[%d * a.out_rs[%d]]" % (g_, j, g_), odt)))
        L.append("  %s w%d_%d;" % (RT[odt], g_, mm_top[g_]))
    for k in range(n_rec):
        sl = n_mm + k
        odt = spec.out_dtypes[sl]
        L.append("  %s* const ob%d = (%s*)a.out[%d];" % (CT[odt], sl, CT[odt], sl))
        L.append("  i64 op%d = a.out_pos0[%d];" % (sl, sl))
        for d in range(1, pr.depth[k] + 1):
            L.append("  %s r%d_%d = %s;" % (RT[odt], k, d, rd(
                "ob%d[((op%d - %d + a.out_store[%d]) %% a.out_store[%d]) * a.out_rs[%d]%s]" % (
                    sl, sl, d, sl, sl, sl, at(rec_kind[k])), odt)))
    for j in range(n_nit):
        sl = n_mm + n_rec + j
        odt = spec.out_dtypes[sl]
        L.append("  %s* const ob%d = (%s*)a.out[%d];" % (CT[odt], sl, CT[odt], sl))
        L.append("  i64 op%d = a.out_pos0[%d];" % (sl, sl))
    for v, m in sorted(pr.shared.items(), key=lambda kv: kv[1]):
        sdt = spec.sh_dtypes[m]
        L.append("  %s s%d = %s;" % (RT[sdt], m, rd("((const %s*)a.sh_in[%d])[%s]" % (
            CT[sdt], m, "0" if sh_kind[m] else E), sdt)))
    for v in used_seq:
        L.append("  const %s* const sq%d = (const %s*)a.seq[%d];" % (CT[dt[v]], pr.seq[v], CT[dt[v]], pr.seq[v]))
    if wg and (any(rec_kind) or any(sh_kind)):
        # every thread has read the 0-d initial values before thread 0 may overwrite their rows
        # (a circular buffer of two rows: step 1 lands on the row of the initial value)
        L.append("  __syncthreads();")
    L.append("  i64 t = 0;")
    L.append("  for (i64 t0 = 0; t0 < a.T; t0 += %d) {" % AHEAD)
    # the sequence rows of the next AHEAD steps, loaded before any of them is used (clamped rows:
    # no branch between the loads)
    for u in range(AHEAD):
        L.append("    const i64 tt%d = t0 + %d < a.T ? t0 + %d : a.T - 1;" % (u, u, u))
        for v in used_seq:
            s = pr.seq[v]
            L.append("    const %s x%d_%d = %s;" % (RT[dt[v]], s, u, rd("sq%d[tt%d * a.seq_ts[%d]%s]" % (
                s, u, s, at(kind(v))), dt[v])))
        for g_ in range(n_mm):
            # the row that enters the window at that step: written by no earlier step (an out-tap j
            # of step t' lands on it when t' + j = t + top, i.e. t' >= t)
            odt = spec.out_dtypes[g_]
            L.append("    const %s mx%d_%d = %s;" % (RT[odt], g_, u, rd(
                "mb%d[(tt%d + %d) * a.out_rs[%d]]" % (g_, u, mm_top[g_], g_), odt)))
    for u in range(AHEAD):
        L.append("    if (t0 + %d < a.T) {" % u)
        env = {}
        for g_ in range(n_mm):
            L.append("      w%d_%d = mx%d_%d;" % (g_, mm_top[g_], g_, u))
        for v, (g_, tap) in pr.mm_in.items():
            env[v] = "w%d_%d" % (g_, tap)
        for v in used_seq:
            env[v] = "x%d_%d" % (pr.seq[v], u)
        for v, (k, d) in pr.tap.items():
            env[v] = "r%d_%d" % (k, d)
        for v, m in pr.shared.items():
            env[v] = "s%d" % m
        for v, j in pr.nsq.items():
            env[v] = "nv%d" % j
        ri = 0
        for si, st in enumerate(pr.steps):
            in_exprs = [env[v] for v in st["ins"]]
            in_dts = [dt[v] for v in st["ins"]]
            lines, outs, odts = cg.emit_scalar_body(st["scalar"], in_exprs, in_dts, indent="      ",
                                                    suffix="_s%d_u%d" % (si, u))
            L.extend(lines)
            for o, ri_ in zip(st["outs"], st["out_refs"]):
                nm = "v%d_u%d" % (o, u)
                L.append("      const %s %s = %s;" % (RT[dt[o]], nm, cg._cast(outs[ri_], odts[ri_], dt[o])))
                env[o] = nm
            red = st.get("reduce")
            if red is not None:
                # the step's full reduction: dead threads contribute the identity; wavefront fold
                # (DPP, fixed tree: every lane ends with it), the wavefronts' folds through LDS
                # (parity of t: one barrier), every thread folds them in wave order
                acc_t = RT[red["acc"]]
                comb = lambda a_, b_, _r=red: cg.red_combine(_r["op"], _r["acc"], a_, b_)  # noqa: E731
                rv = "rd%d_u%d" % (si, u)
                L.append("      %s %s = live ? %s : %s;" % (
                    acc_t, rv, cg._cast(outs[red["ref"]], odts[red["ref"]], red["acc"]),
                    "(%s)%s" % (acc_t, cg.red_identity(red["op"], red["acc"]))))
                L.extend(cg.wave_fold_lines(acc_t, comb, var=rv, indent="      "))
                L.append("      if (nw_ > 1u) {")
                L.append("        if (ln_ == 0) smr%d[t & 1][wv_] = %s;" % (ri, rv))
                L.append("        __syncthreads();")
                L.append("        %s = (%s)smr%d[t & 1][0];" % (rv, acc_t, ri))
                L.append("        for (unsigned w_ = 1; w_ < nw_; ++w_) %s = %s;" % (
                    rv, comb(rv, "(%s)smr%d[t & 1][w_]" % (acc_t, ri))))
                L.append("      }")
                nm = "v%d_u%d" % (red["out"], u)
                L.append("      const %s %s = %s;" % (RT[dt[red["out"]]], nm, cg._cast(rv, red["acc"], dt[red["out"]])))
                env[red["out"]] = nm
                ri += 1

        def st_val(v, odt):
            e_ = cg._cast(env[v], dt[v], odt)
            return "(unsigned char)(%s)" % e_ if odt == "bool" else e_
        # every new value is taken before any register of the step's inputs is overwritten
        for g_ in range(n_mm):
            odt = spec.out_dtypes[g_]
            for tap, v in pr.mm_new[g_]:
                L.append("      const %s nm%d_%d = %s;" % (RT[odt], g_, tap, cg._cast(env[v], dt[v], odt)))
        for k in range(n_rec):
            odt = spec.out_dtypes[n_mm + k]
            L.append("      const %s n%d = %s;" % (RT[odt], k, cg._cast(env[pr.rec_new[k]], dt[pr.rec_new[k]], odt)))
        for g_ in range(n_mm):
            odt = spec.out_dtypes[g_]
            for tap, v in pr.mm_new[g_]:
                L.append("      w%d_%d = nm%d_%d;" % (g_, tap, g_, tap))
                L.append("      mb%d[(t + %d) * a.out_rs[%d]] = %s;" % (
                    g_, tap, g_, "(unsigned char)nm%d_%d" % (g_, tap) if odt == "bool" else "nm%d_%d" % (g_, tap)))
        for k in range(n_rec):
            sl = n_mm + k
            odt = spec.out_dtypes[sl]
            L.append("      %sob%d[op%d * a.out_rs[%d]%s] = %s;" % (
                guard(rec_kind[k]), sl, sl, sl, " + e" if not rec_kind[k] else "",
                "(unsigned char)n%d" % k if odt == "bool" else "n%d" % k))
        for j in range(n_nit):
            sl = n_mm + n_rec + j
            L.append("      %sob%d[op%d * a.out_rs[%d]%s] = %s;" % (
                guard(nit_kind[j]), sl, sl, sl, " + e" if not nit_kind[j] else "",
                st_val(pr.nit_new[j], spec.out_dtypes[sl])))
        sh_tmp = []
        for m, v in enumerate(pr.sh_new):
            L.append("      const %s ns%d = %s;" % (RT[spec.sh_dtypes[m]], m, cg._cast(env[v], dt[v], spec.sh_dtypes[m])))
            sh_tmp.append(m)
        if pr.cond is not None:
            L.append("      const bool stop_ = (bool)(%s);" % cg._cast(env[pr.cond], dt[pr.cond], "bool"))
        # shift the taps, advance the circular positions
        for k in range(n_rec):
            for d in range(pr.depth[k], 1, -1):
                L.append("      r%d_%d = r%d_%d;" % (k, d, k, d - 1))
            L.append("      r%d_1 = n%d;" % (k, k))
        for m in sh_tmp:
            L.append("      s%d = ns%d;" % (m, m))
        for g_ in range(n_mm):          # the window moves on one row
            for j in range(mm_top[g_]):
                L.append("      w%d_%d = w%d_%d;" % (g_, j, g_, j + 1))
        for sl in range(n_mm, n_mm + n_rec + n_nit):
            L.append("      if (++op%d == a.out_store[%d]) op%d = 0;" % (sl, sl, sl))
        L.append("      ++t;")
        if pr.cond is not None:
            L.append("      if (stop_) goto done_;")
        L.append("    }")
    L.append("  }")
    if pr.cond is not None:
        L.append("done_:")
    for m in range(len(pr.sh_new)):
        sdt = spec.sh_dtypes[m]
        L.append("  %s((%s*)a.sh_out[%d])[%s] = %s;" % (
            guard(sh_kind[m]), CT[sdt], m, "0" if sh_kind[m] else "e",
            "(unsigned char)s%d" % m if sdt == "bool" else "s%d" % m))
    if pr.as_while:
        L.append("  if (e == 0) a.ctl[0] = (unsigned)t;      // steps this do-while ran")
    L.append("}")
    return "\n".join(L) + "\n", (name,)
