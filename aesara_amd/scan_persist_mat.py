"""K10p (matrix state) — the persistent one-kernel Scan loop for a BATCH of recurrences.

Same boundary and exchange protocol as ``scan_persist`` (which see), for inner graphs whose fused
steps are small-M GEMM chains + Elemwise (``gemm_epi``) on a float32 matrix state ``h[B, N]`` —
BASELINE config 4 with B = 64: ``sigmoid(h @ Ur + xr) * h``, ``tanh((r*h) @ Uh + xh)`` … —
instead of 2 launches per step that re-stream 12 MiB of weights from the memory-side cache:

* workgroup (bi, nj) owns the 16 x 16 output tile of batch block ``bi`` (16 rows) and column
  slice ``nj`` (16 columns); its 4 wavefronts split K, and each lane keeps ITS share of the
  weight columns of every matrix in VGPRs in MFMA B-operand layout for the whole loop
  (K / 16 registers per matrix: 192 for config 4) — weights are read from HBM once per eval;
* the A operand (the 16 x K block of ``h`` or ``r*h`` that belongs to batch block ``bi``) is what
  the 64 workgroups of that block exchange: every workgroup publishes its 256 tile values as
  tagged 8-byte granules and polls the 16 x K block (chunks of 16 granules per thread) into a
  padded LDS image (pitch K + 4: conflict-free ``ds_read_b128`` in MFMA A layout);
* ``v_mfma_f32_16x16x4_f32`` over the wave's K quarter, partial tiles summed in wave order through
  LDS, one thread per tile element runs the epilogue, keeps its state / sequence operands in
  registers, publishes and stores;
* products whose operand block is already staged (``h @ Uz`` in the second GRU phase) are issued
  BEFORE the wait for the phase's new operand, so their MFMA time hides behind the hand-off.

Batch blocks are independent recurrences: block -> workgroup mapping ``bi = blockIdx % NB`` puts
the workgroups of one block on one or two XCDs (``blockIdx % 8``), so a block's granules are
polled through two L2s only.

The same kernel runs gradient Scans (mit-mot groups [0, 1] -> [1]: a state plus a sequence that
lives in the output buffer, ``scan_persist.analyze``), phases without a product (Elemwise only:
no barrier, publishes deferred to the next hand-off), plain ``Dot22`` / ``Gemm`` nodes as product
phases, several operands published together (one polling pass, loads in flight together),
operand blocks that lend each other their LDS slot, and state widths that are not a multiple of
64 (``SpecMat.Nt``: the executor zero-pads the weights, unowned columns are never published).

Round 3 added the FRAGMENT form of the exchange (``_generate_frag``: payload in MFMA A-fragment
order, pulled straight into the registers the MFMAs read — the default whenever the weight columns
and the operand fragments alive at the same time fit in 320 registers, which now includes the
gradient kernels) and, on it, the sequence products ``x_t @ W`` computed INSIDE the loop
(``xfold_pairs`` / ``xfold_windows`` / ``emit_window``: weight columns in LDS, ``x`` in fragment
order a step ahead, the work placed in the idle windows in front of each operand fetch).
DESIGN.md §3.3b has the timelines each of these decisions came from.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os

from . import knobs
from . import codegen as cg
from .scan_persist import SPIN_LIMIT

SM_MAXMAT = 8
SM_MAXSEQ = 28
SM_MAXNSQ = 8
SM_MAXOUT = 16
TRACE_T0, TRACE_NT, TRACE_MARKS = 200, 32, 32      # steps traced, stamps per step (u64 each)


class SmArgs(C.Structure):
    _fields_ = [
        ("T", C.c_int64),
        ("mat", C.c_void_p * SM_MAXMAT), ("mat_rs", C.c_int64 * SM_MAXMAT),
        ("seq", C.c_void_p * SM_MAXSEQ), ("seq_ts", C.c_int64 * SM_MAXSEQ),
        ("seq_rs", C.c_int64 * SM_MAXSEQ), ("seq_cs", C.c_int64 * SM_MAXSEQ),
        ("nsq", C.c_void_p * SM_MAXNSQ), ("nsq_rs", C.c_int64 * SM_MAXNSQ),
        ("nsq_cs", C.c_int64 * SM_MAXNSQ),
        ("out", C.c_void_p * SM_MAXOUT), ("out_ts", C.c_int64 * SM_MAXOUT),
        ("out_rs", C.c_int64 * SM_MAXOUT), ("out_store", C.c_int64 * SM_MAXOUT),
        ("out_pos0", C.c_int64 * SM_MAXOUT),
        ("xch", C.c_void_p), ("ctl", C.c_void_p),
    ]


SM_STRUCT = r"""
#define SM_MAXMAT %d
#define SM_MAXSEQ %d
#define SM_MAXNSQ %d
#define SM_MAXOUT %d
struct SmArgs {
  i64 T;
  const void* mat[SM_MAXMAT]; i64 mat_rs[SM_MAXMAT];
  const void* seq[SM_MAXSEQ]; i64 seq_ts[SM_MAXSEQ]; i64 seq_rs[SM_MAXSEQ]; i64 seq_cs[SM_MAXSEQ];
  const void* nsq[SM_MAXNSQ]; i64 nsq_rs[SM_MAXNSQ]; i64 nsq_cs[SM_MAXNSQ];
  void* out[SM_MAXOUT]; i64 out_ts[SM_MAXOUT]; i64 out_rs[SM_MAXOUT]; i64 out_store[SM_MAXOUT];
  i64 out_pos0[SM_MAXOUT];
  unsigned long long* xch; unsigned* ctl;
};
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
""" % (SM_MAXMAT, SM_MAXSEQ, SM_MAXNSQ, SM_MAXOUT)


class SpecMat:
    """Shape-specialised kernel: B batch rows, N state columns, K per weight matrix."""

    def __init__(self, prog, B, N, Ks, Nt=None, dtype="float32", xfold=None, nblk=1):
        self.dtype = dtype
        # nblk: batch blocks (of 16 rows) per workgroup.  A workgroup owns ONE 16 x 16 tile per
        # block; when B * N / 256 exceeds the CU count it takes two blocks (independent
        # recurrences: the whole step body runs once per block, sharing the weight registers)
        self.nblk = nblk
        # xfold: None, or {"sx": sequence slot of the fragment-ordered x, "items": [(v, phase,
        # dot, mat slot)]} — sequence products x_t @ W computed inside the loop (columns in LDS)
        self.xfold = xfold
        # N: state width the tiles cover (a multiple of 64); Nt: the true width (<= N) when the
        # executor zero-padded the weights — columns Nt .. N-1 are never owned, never published
        # (they read as the zeros the exchange buffer starts with) and hit zero weight rows
        self.prog, self.B, self.N, self.Ks = prog, B, N, dict(Ks)
        self.Nt = N if Nt is None else Nt
        self.NB = -(-B // 16)
        self.NJ = N // 16
        # granule loads a thread keeps in flight per polling pass (2 VGPRs each)
        self.chunk = int(knobs.get("SM_CHUNK"))
        # exchange form: "flag" = untagged 8-byte float pairs + ONE tag word per producing
        # workgroup (half the bytes, one polling pass); "granule" = {tag, value} per element
        #   "frag" (default when the registers allow it) = the flag form with the payload laid out
        #   in MFMA A-FRAGMENT order: a consumer wavefront polls only the tags of the producers of
        #   ITS K quarter and pulls that quarter with coalesced 16-byte sc1 loads straight into the
        #   registers the MFMAs read — no LDS image, no staging pass, no barrier between hand-off
        #   and product (round 3: the r03 timeline showed 2.5 of 8.6 us per step in the LDS staging
        #   of the payload and 3.6 us in LDS-latency-serialised dependent MFMA chains)
        # operand fragments alive at the same time: an operand is fetched in the phase of its first
        # product and dead after its last
        first, last = {}, {}
        for pi, ph in enumerate(prog.phases):
            for _a, x in ph["dots"]:
                k_ = (x, "prev" if x in prog.state else "cur")
                first.setdefault(k_, pi)
                last[k_] = pi
        nstaged = max([sum(1 for k_ in first if first[k_] <= pi <= last[k_]) for pi in range(len(prog.phases))],
                      default=0)
        self.nstaged = nstaged
        K0 = max(Ks.values()) if Ks else 64
        wpr = 2 if dtype == "float64" else 1          # 32-bit registers per value
        regs = wpr * (sum(K // 16 for K in Ks.values()) + (K0 // 16) * nstaged)
        # measured (r03, T = 512, H = 1024, B = 64): the forward GRU kernel (192 registers of
        # weights + two fragments) 6.09 -> 5.7 ms in the fragment form; the backward kernel (three
        # operands, two alive at a time: 320 as well) 6.4 -> 5.8 ms.  In-loop sequence products
        # exist in the fragment form only (their weights live in LDS, one more fragment for x)
        if xfold:
            self.xmode = "frag"
        else:
            self.xmode = knobs.get("SM_XMODE") or ("frag" if regs <= 320 else "flag")
        if dtype == "float64" and not xfold:
            self.xmode = "frag" if regs <= 384 else "none"     # float64 exists in the fragment form only
        # products on operands that are already in registers run BEFORE the wait for the new
        # operand's tags (r03 timeline, config 4 B = 64: 6.7 us per step; issuing them after the
        # loads — to cover the load latency instead of the tag latency — measured 8.5)
        self.early_first = knobs.get("SM_EARLY") == "first"
        # fragment form: the weight columns pinned to ARCHITECTURAL registers (the fragments the
        # loads deliver then live in accumulation registers).  Left to the allocator, products
        # end up with A and B both in AGPRs and get a v_accvgpr_read through ONE temporary in
        # front of every MFMA (r03 timeline: 64 such MFMAs 1.2 us, 64 plain ones 0.93)
        #   2 = pinned to ACCUMULATION registers instead (MFMA reads B from either file): the
        #   fragments and x then stay in the architectural file the loads deliver into
        pin_k = knobs.get("SM_PIN")
        pin_k = (2 if xfold else 1) if pin_k is None else int(pin_k)
        self.pin = pin_k if (self.xmode == "frag" and wpr * sum(K // 16 for K in Ks.values()) <= 192) else 0
        # per-phase timeline (tools/sm_trace.py): thread 0 of workgroups 0 and NB*NJ/2 stamps
        # s_memtime at every mark of steps TRACE_T0 .. TRACE_T0+TRACE_NT-1 into ctl[16..]
        self.trace = bool(int(knobs.get("SM_TRACE")))
        # fragment form schedule:
        #   init     "publish": every workgroup publishes its tile of the initial state through the
        #            exchange before the loop — the loop body has no step-0 path (the merge of the two
        #            paths cost 84 register copies per step and conservative s_waitcnt vmcnt(0))
        #   xreload  where the next x is requested (16 loads per lane): behind the MFMAs of a product
        #            that waits for payload ("late": each load finds the queue full of the 16 payload
        #            loads and stalls the instruction stream) or of one that waits for nothing ("early")
        #   ackfill / xpre   MFMAs that need nothing from the hand-off, issued between the payload
        #            stores and the wait for their acknowledgement (0.35 us per hand-off otherwise idle)
        self.init = str(knobs.get("SM_INIT")) if self.xmode == "frag" else "branch"
        sched = self.xmode == "frag" and bool(xfold) and nblk == 1 and self.init == "publish"
        self.xreload = str(knobs.get("SM_XRELOAD")) if sched else "late"
        self.ackfill = int(knobs.get("SM_ACKFILL")) if sched else 0
        self.xpre = int(knobs.get("SM_XPRE")) if sched else 0
        self.look = int(knobs.get("SM_LOOK")) if sched else 0
        self.epre = int(knobs.get("SM_EPRE")) if (sched and self.ackfill) else 0
        self.elook = int(knobs.get("SM_ELOOK")) if (sched and self.ackfill) else 0
        self.xsplit = int(knobs.get("SM_XSPLIT")) if sched else 0
        self.nxt = str(knobs.get("SM_NXT")) if self.xmode == "frag" else "top"
        self.polls = int(knobs.get("SM_POLLS")) if self.xmode == "frag" else 1
        self.spin_delay = int(knobs.get("SM_SPIN_DELAY")) if self.xmode == "frag" else 0

    def key(self):
        pr = self.prog
        blob = json.dumps(["sm9" + ("t" if self.trace else "") + ("/f5p%d%s" % (self.pin, str(knobs.get("SM_FENCE"))) if self.xmode == "frag" else ""), self.xfold and [self.xfold, "w15", str(knobs.get("SM_XTAIL"))], [self.init, self.xreload, self.ackfill, self.xpre, self.look, self.nxt, self.epre, self.elook, self.xsplit, self.polls, self.spin_delay], self.dtype, self.chunk, self.xmode, self.early_first, self.B, self.N, self.Nt, sorted(self.Ks.items()), sorted(pr.seq.items()),
                           sorted(pr.state.items()), sorted(pr.older.items()), sorted(pr.depth.items()), sorted(pr.nsq.items()), sorted(pr.mats.items()),
                           [[ph["dots"], ph["ins"], ph["outs"], ph["scalar"], ph["out_refs"]]
                            for ph in pr.phases], pr.outs, pr.exchanged, sorted(pr.tap_seq.items())] +
                          ([["nblk", self.nblk, str(knobs.get("SM_INTERLEAVE"))]] if self.nblk != 1 else []),
                          sort_keys=True)
        return hashlib.sha256(blob.encode()).hexdigest()[:24]


def xfold_pairs(prog, lifted_outs):
    """In-kernel sequence products (round 3).  ``lifted_outs``: loop-plan variables that are rows of
    whole-sequence products ``x_t @ W_k`` (what ``fusion.hoist_sequence_only`` lifted out).  The
    product can move INTO the persistent loop — accumulated on the matrix cores by the same
    wavefronts, in the shadow of the hand-off latency, feeding the accumulator of the recurrent
    product it is added to — when the step uses it in exactly one place: ``dot_d + v`` (possibly
    through ``mul(1.0, .)`` wrappers of a Gemm's alpha).  Returns ``[(v, phase, dot index)]`` in
    ``lifted_outs`` order, or None when some variable does not have that form."""
    def const1(r):
        return r[0] == "c" and float(r[1]) == 1.0

    pairs = []
    for v in lifted_outs:
        where = [pi for pi, ph in enumerate(prog.phases) if v in ph["ins"]]
        if len(where) != 1 or any(v == x for ph in prog.phases for _a, x in ph["dots"]) or \
                any(v == o for o, _k, _j in prog.outs) or v in prog.passthru.values():
            return None
        pi = where[0]
        ph = prog.phases[pi]
        if ph["ins"].count(v) != 1:
            return None
        D = len(ph["dots"])
        nodes = ph["scalar"]["nodes"]
        me = ["i", D + ph["ins"].index(v)]

        def users(ref):
            return [k for k, n in enumerate(nodes) if any(list(r) == list(ref) for r in n["in"])] + \
                   (["out"] if any(list(r) == list(ref) for r in ph["scalar"]["out"]) else [])

        def through_unit_mul(ref):
            """follow ref through mul(1.0, ref) wrappers that have no other user"""
            while True:
                us = users(ref)
                if len(us) != 1 or us[0] == "out":
                    return ref, us
                n = nodes[us[0]]
                if n["op"] == "mul" and len(n["in"]) == 2 and any(const1(r) for r in n["in"]):
                    ref = ["t", us[0]]
                    continue
                return ref, us
        ref, us = through_unit_mul(me)
        if len(us) != 1 or us[0] == "out":
            return None
        add = nodes[us[0]]
        if add["op"] != "add" or len(add["in"]) != 2 or add["dtype"] != "float32":
            return None
        other = [r for r in add["in"] if list(r) != list(ref)]
        if len(other) != 1:
            return None
        # the other operand must be dot_d (possibly behind its own unit mul, used only here)
        o = other[0]
        while o[0] == "t":
            n = nodes[o[1]]
            if n["op"] == "mul" and len(n["in"]) == 2 and any(const1(r) for r in n["in"]) and \
                    len(users(o)) == 1:
                o = [r for r in n["in"] if not const1(r)][0]
            else:
                return None
        if o[0] != "i" or o[1] >= D or len(users(o)) != 1:
            return None
        if any(q[1] == pi and q[2] == o[1] for q in pairs):
            return None
        pairs.append((v, pi, o[1]))
    return pairs


def phase_fetches(prog):
    """Per phase of the step: (early, late, fresh) — the products on operands staged earlier in
    the step, the products on operands this phase fetches, and those operands [(var, kind)]."""
    staged, res = set(), []
    for ph in prog.phases:
        early, late, fresh = [], [], []
        for d, (a_, x) in enumerate(ph["dots"]):
            kind = "prev" if x in prog.state else "cur"
            if (x, kind) in staged:
                early.append((d, a_, x))
            else:
                late.append((d, a_, x))
                if (x, kind) not in fresh:
                    fresh.append((x, kind))
        staged.update(fresh)
        res.append((early, late, fresh))
    return res


def xfold_windows(prog, pairs):
    """Where the in-kernel sequence products run.  The free issue slots of a step are the WINDOWS
    in front of each operand fetch: a wavefront that has raised its tag waits there for its
    producers' tags and then for the payload.  The sequence product feeding recurrent product
    (phase p, d) runs in the window of the last fetch strictly before p — for the products of the
    first fetching phase that is the last window of the PREVIOUS step, computed from x_{t+1}
    ("wrapped").  -> {"win": {phase: [pair index]}, "wrapped": set, "reload": phase after whose
    window x moves on, "ahead": how many steps ahead that load reaches} or None when one window
    would need two different x."""
    fetch = [pi for pi, (_e, _l, fresh) in enumerate(phase_fetches(prog)) if fresh]
    if not fetch or not pairs:
        return None
    win, wrapped = {f: [] for f in fetch}, set()
    for gi, (_v, pi, _d) in enumerate(pairs):
        prev = [f for f in fetch if f < pi]
        if prev:
            win[prev[-1]].append(gi)
        else:
            win[fetch[-1]].append(gi)
            wrapped.add(gi)
    plain = [f for f in fetch if any(gi not in wrapped for gi in win[f])]
    if wrapped and plain and plain[-1] >= fetch[-1]:
        return None
    if plain:
        return {"win": win, "wrapped": wrapped, "reload": plain[-1], "ahead": 1}
    return {"win": win, "wrapped": wrapped, "reload": fetch[-1], "ahead": 2}


def xch_layout(prog, NB, N, xmode="flag", itemsize=4):
    """u64 offsets of the exchange buffer.  granule form: per exchanged matrix 4 slots of NB
    blocks of 16 x N granules -> {var: (offset, slot length)}.  flag form: 4 slots of
    NB x 16 x N/2 float pairs followed by 4 slots of NB x NJ tag words ->
    {var: (payload offset, payload slot length, flag offset, flag slot length)}."""
    off, total = {}, 0
    if xmode == "granule":
        lp = NB * 16 * N
        for v in prog.exchanged:
            off[v] = (total, lp)
            total += 4 * lp
        return off, total
    # ("flag" and "frag" share this layout; "frag" orders the payload of a block by MFMA fragment)
    lpp, lpf = NB * 16 * N * itemsize // 8, NB * (N // 16)
    for v in prog.exchanged:
        off[v] = (total, lpp, total + 4 * lpp, lpf)
        total += 4 * (lpp + lpf)
    return off, total


def stage_slots(prog, Ks):
    """LDS images of the staged operand blocks (16 x (K + 4) floats each): {(var, kind): (float
    offset, K)} and the total.  A block whose last product lies before another block's first one
    lends it its space (a gradient step stages three operands, two at a time)."""
    first, last, order = {}, {}, []
    for pi, ph in enumerate(prog.phases):
        for a_, x in ph["dots"]:
            key = (x, "prev" if x in prog.state else "cur")
            if key not in first:
                first[key] = pi
                order.append((key, Ks[a_]))
            last[key] = pi
    stage, free, stot = {}, [], 0
    live = []                                   # (last phase, offset, K)
    for key, K in order:
        for ent in [e for e in live if e[0] < first[key]]:
            live.remove(ent)
            free.append((ent[1], ent[2]))
        slot = next((f for f in free if f[1] == K), None)
        if slot is not None and key[1] == "cur":
            free.remove(slot)
            off = slot[0]
        else:
            off = stot
            stot += 16 * (K + 4)
        stage[key] = (off, K)
        live.append((last[key], off, K))
    return stage, stot


def generate(spec: SpecMat):
    pr, B, N, NB, NJ = spec.prog, spec.B, spec.N, spec.NB, spec.NJ
    name = "sm_" + spec.key()
    AG = "__ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT"
    L = [cg.PRELUDE, SM_STRUCT, "typedef unsigned u4v __attribute__((ext_vector_type(4)));"]
    xoff, _tot = xch_layout(pr, NB, N, spec.xmode, 8 if spec.dtype == "float64" else 4)
    FRAG = spec.xmode == "frag"
    FLAG = spec.xmode in ("flag", "frag")
    if FRAG:
        return _generate_frag(spec, name, xoff, _tot)

    # staged operand blocks (LDS images, pitch K + 4 floats)
    stage, stot = stage_slots(pr, spec.Ks)
    ndots_max = max(len(ph["dots"]) for ph in pr.phases)
    L.append('extern "C" __global__ __launch_bounds__(256) void %s(SmArgs a) {' % name)
    L.append("  __shared__ __attribute__((aligned(16))) float Hl[%d];" % max(stot, 4))
    L.append("  __shared__ float part[%d][4][256];" % max(ndots_max, 1))
    L.append("  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;")
    L.append("  const int r16 = lane & 15, grp = lane >> 4;")
    L.append("  const int bi = blockIdx.x %% %d, nj = blockIdx.x / %d;" % (NB, NB))
    L.append("  const int erow = tid >> 4, ecol = tid & 15;        // tile element owned by this thread")
    L.append("  const i64 eb = (i64)bi * 16 + erow, en = (i64)nj * 16 + ecol;")
    L.append("  const bool owner = eb < %d && en < %d;" % (B, spec.Nt))
    L.append("  const int vrows = (%d - bi * 16) < 16 ? (%d - bi * 16) : 16;   // valid rows of this batch block" % (B, B))
    L.append("  const unsigned base = __hip_atomic_load(a.ctl, %s);" % AG)
    L.append("  unsigned* errp = a.ctl + 1;")
    if FLAG:
        L.append("  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, %du, 0x00020000);"
                 % (_tot * 8))
    if spec.trace:
        L.append("  const bool tr_on = tid == 0 && (blockIdx.x == 0 || blockIdx.x == %d);" % (NB * NJ // 2))
        L.append("  unsigned long long* tr = (unsigned long long*)(a.ctl + 16) + (blockIdx.x == 0 ? 0 : %d);"
                 % (TRACE_NT * TRACE_MARKS + 4))
        L.append("  if (tr_on) { tr[%d] = __builtin_amdgcn_s_memtime(); tr[%d] = __builtin_amdgcn_s_memrealtime(); }"
                 % (TRACE_NT * TRACE_MARKS, TRACE_NT * TRACE_MARKS + 1))
    # ---- weights -> registers, MFMA B layout: lane (r16, grp) of wave w holds
    #      W[w*K/4 + grp*K/16 + s][nj*16 + r16], s = 0 .. K/16 - 1
    for av, slot in sorted(pr.mats.items(), key=lambda t: t[1]):
        K = spec.Ks[av]
        L.append("  const i64 wk%d = (i64)wave * %d + grp * %d;" % (slot, K // 4, K // 16))
        for s_ in range(K // 16):
            L.append("  const float w%d_%d = ((const float*)a.mat[%d])[(wk%d + %d) * a.mat_rs[%d] + nj * 16 + r16];"
                     % (slot, s_, slot, slot, s_, slot))
    # ---- registers of the element owner
    out_of = {}
    for o, kind, j in pr.outs:
        out_of.setdefault(o, []).append((kind, j))
    for v, k in pr.state.items():
        L.append("  float own_%d = 0.f;" % v)
        L.append("  if (owner) own_%d = ((const float*)a.out[%d])[((a.out_pos0[%d] + a.out_store[%d] - 1) %% "
                 "a.out_store[%d]) * a.out_ts[%d] + eb * a.out_rs[%d] + en];" % (v, k, k, k, k, k, k))
    pw_nsq = sorted({v for ph in pr.phases for v in ph["ins"] if v in pr.nsq})
    for v in pw_nsq:
        s_ = pr.nsq[v]
        L.append("  float own_%d = 0.f;" % v)
        L.append("  if (owner) own_%d = ((const float*)a.nsq[%d])[eb * a.nsq_rs[%d] + en * a.nsq_cs[%d]];"
                 % (v, s_, s_, s_))
    pw_seq = sorted({v for ph in pr.phases for v in ph["ins"] if v in pr.seq})
    for v in pw_seq:
        s_ = pr.seq[v]
        L.append("  float nxt_%d = 0.f, own_%d = 0.f;" % (v, v))
        L.append("  if (owner && a.T > 0) nxt_%d = ((const float*)a.seq[%d])[eb * a.seq_rs[%d] + en * a.seq_cs[%d]];"
                 % (v, s_, s_, s_))
    for ph in pr.phases:
        for o in ph["outs"]:
            L.append("  float own_%d = 0.f;" % o)
    L.append("  for (i64 t = 0; t < a.T; ++t) {")
    for v in pw_seq:
        s_ = pr.seq[v]
        L.append("    own_%d = nxt_%d;" % (v, v))
        L.append("    if (owner && t + 1 < a.T) nxt_%d = ((const float*)a.seq[%d])[(t + 1) * a.seq_ts[%d] + "
                 "eb * a.seq_rs[%d] + en * a.seq_cs[%d]];" % (v, s_, s_, s_, s_))
    staged_this_step = set()

    def emit_mfma(pi, d, a_, x):
        slot = pr.mats[a_]
        K = spec.Ks[a_]
        kind = "prev" if x in pr.state else "cur"
        so, _k = stage[(x, kind)]
        P = K + 4
        # r03: the fragment fetches of a product are issued together (their LDS latency used to
        # be exposed once per 4 MFMAs) and the k-steps alternate between two accumulator chains
        # (a chain of dependent 16x16x4 MFMAs issues every 40 cycles, two interleaved every 32)
        Qn = K // 64
        L.append("    {")
        L.append("      f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};")
        L.append("      const float* hp = Hl + %d + r16 * %d + wave * %d + grp * %d;" % (so, P, K // 4, K // 16))
        L.append("      f4 av[%d];" % Qn)
        for q in range(Qn):
            L.append("      av[%d] = *(const f4*)(hp + %d);" % (q, 4 * q))
        for q in range(Qn):
            for e, c in enumerate("xyzw"):
                acc = "acc%d" % ((4 * q + e) & 1)
                L.append("      %s = __builtin_amdgcn_mfma_f32_16x16x4f32(av[%d].%s, w%d_%d, %s, 0, 0, 0);"
                         % (acc, q, c, slot, 4 * q + e, acc))
        L.append("      for (int i = 0; i < 4; ++i) part[%d][wave][(4 * grp + i) * 16 + r16] = acc0[i] + acc1[i];" % d)
        L.append("    }")

    marks = []

    def stamp(label):
        """record s_memtime under `label` (trace builds only; marks are numbered in code order)"""
        if not spec.trace:
            return
        k = len(marks)
        marks.append(label)
        assert k < TRACE_MARKS
        L.append("    if (tr_on && t >= %d && t < %d) tr[(t - %d) * %d + %d] = __builtin_amdgcn_s_memtime();"
                 % (TRACE_T0, TRACE_T0 + TRACE_NT, TRACE_T0, TRACE_MARKS, k))

    pending_pub = []
    for pi, ph in enumerate(pr.phases):
        L.append("    // ---- phase %d" % pi)
        stamp("p%d start" % pi)
        early, late = [], []
        for d, (a_, x) in enumerate(ph["dots"]):
            kind = "prev" if x in pr.state else "cur"
            (early if (x, kind) in staged_this_step else late).append((d, a_, x))
        # products on blocks that are already staged run before the wait for the new operand
        # (after a barrier: the previous phase's epilogue threads may still be reading `part`)
        if early:
            L.append("    __syncthreads();")
        for d, a_, x in early:
            emit_mfma(pi, d, a_, x)
        newly = []
        fresh = []
        for d, a_, x in late:
            kind = "prev" if x in pr.state else "cur"
            if (x, kind) not in staged_this_step and (x, kind) not in fresh:
                fresh.append((x, kind))
        if FLAG and len(fresh) >= 2 and all(k_ == "cur" for _x, k_ in fresh):
            # several operands published by the same epilogue (the two last-phase products of a
            # gradient step): ONE polling pass over all their tags, all bulk loads in flight
            # together, then per operand  LDS image -> barrier -> its products, so the MFMAs of
            # the first hide the arrival of the second
            for x, kind in fresh:
                so = stage[(x, kind)][0]
                staged_this_step -= {q for q in staged_this_step if stage[q][0] == so}
                staged_this_step.add((x, kind))
            ind = "    "
            L.append(ind + "const unsigned long long want64_p%d = (unsigned long long)(base + (unsigned)t + 1u);" % pi)
            L.append(ind + "if (wave == 0) {")
            L.append(ind + "  for (int spin = 0;; ++spin) {")
            L.append(ind + "    bool ok = true;")
            for x, kind in fresh:
                po_, lpp, fo_, lpf = xoff[pr.new_of_state.get(x, x)]
                L.append(ind + "    for (int j = lane; j < %d; j += 64) ok = ok && (__hip_atomic_load("
                         "a.xch + %d + (t & 3) * %d + (i64)bi * %d + j, %s) == want64_p%d);"
                         % (NJ, fo_, lpf, NJ, AG, pi))
            L.append(ind + "    if (__all(ok)) break;")
            L.append(ind + "    if (spin > %d || ((spin & 255) == 255 && __hip_atomic_load(errp, %s) != 0u)) "
                     "{ __hip_atomic_store(errp, 1u, %s); break; }" % (SPIN_LIMIT, AG, AG))
            L.append(ind + "    __builtin_amdgcn_s_sleep(1);")
            L.append(ind + "  }")
            L.append(ind + "}")
            L.append(ind + "__syncthreads();")
            # payload as 16-byte sc1 loads (r03; guide: 8-byte accesses run at 0.54-0.70x the
            # 16-byte rate): thread tid takes float4 number u * 256 + tid of the 16 x K block
            for q, (x, kind) in enumerate(fresh):
                so, K = stage[(x, kind)]
                po_, lpp, fo_, lpf = xoff[pr.new_of_state.get(x, x)]
                PT = 16 * K // 4 // 256
                L.append(ind + "const unsigned so%d_%d = (unsigned)((%d + (t & 3) * %d + (i64)bi * %d) * 8);"
                         % (pi, q, po_, lpp, 16 * K // 2))
                L.append(ind + "u4v g%d_%d[%d];" % (pi, q, PT))
                for u in range(PT):
                    L.append(ind + "{ const int idx = %d * 256 + tid; const u4v z_ = {0u, 0u, 0u, 0u}; g%d_%d[%d] = (idx / %d < vrows) ? "
                             "__builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)idx * 16u, so%d_%d, 16) : z_; }"
                             % (u, pi, q, u, K // 4, pi, q))
            done = set()
            for q, (x, kind) in enumerate(fresh):
                so, K = stage[(x, kind)]
                P = K + 4
                PT = 16 * K // 4 // 256
                for u in range(PT):
                    L.append(ind + "{ const int idx = %d * 256 + tid; "
                             "*(u4v*)(Hl + %d + (idx / %d) * %d + 4 * (idx %% %d)) = g%d_%d[%d]; }"
                             % (u, so, K // 4, P, K // 4, pi, q, u))
                L.append(ind + "__syncthreads();")
                for d, a_, x2 in late:
                    if x2 == x and d not in done:
                        done.add(d)
                        emit_mfma(pi, d, a_, x2)
            for d, a_, x2 in late:
                if d not in done:
                    emit_mfma(pi, d, a_, x2)
            late = []
            newly = list(fresh)
        for d, a_, x in late:
            kind = "prev" if x in pr.state else "cur"
            if (x, kind) in staged_this_step:
                continue
            so, K = stage[(x, kind)]
            # a block staged into a lent slot replaces whatever lived there
            staged_this_step -= {q for q in staged_this_step if stage[q][0] == so}
            staged_this_step.add((x, kind))
            newly.append((x, kind))
            P = K + 4
            per_thread = 16 * K // 256          # granules per thread (K % 64 == 0)
            nchunk = max(1, per_thread // spec.chunk)
            cl = per_thread // nchunk           # loads per thread per chunk
            src = pr.new_of_state.get(x, x)
            if kind == "prev":
                k_out = pr.state[x]
                L.append("    if (t == 0) {")
                L.append("      const float* ini = (const float*)a.out[%d] + ((a.out_pos0[%d] + a.out_store[%d] - 1) %% "
                         "a.out_store[%d]) * a.out_ts[%d];" % (k_out, k_out, k_out, k_out, k_out))
                L.append("      for (int idx = tid; idx < %d; idx += 256) {" % (16 * K))
                L.append("        const int rr = idx / %d, cc = idx %% %d;" % (K, K))
                L.append("        Hl[%d + rr * %d + cc] = (rr < vrows && cc < %d) ? ini[((i64)bi * 16 + rr) * a.out_rs[%d] + cc] : 0.f;"
                         % (so, P, spec.Nt, k_out))
                L.append("      }")
                L.append("    } else {")
                step_expr = "(t - 1)"
            else:
                L.append("    {")
                step_expr = "t"
            ind = "      "
            if FLAG:
                po_, lpp, fo_, lpf = xoff[src]
                PT = 16 * K // 2 // 256          # float pairs per thread
                L.append(ind + "const unsigned long long want64 = (unsigned long long)(base + (unsigned)%s + 1u);" % step_expr)
                L.append(ind + "if (wave == 0) {")
                L.append(ind + "  const u64* fl = a.xch + %d + (%s & 3) * %d + (i64)bi * %d;" % (fo_, step_expr, lpf, NJ))
                L.append(ind + "  for (int spin = 0;; ++spin) {")
                L.append(ind + "    bool ok = true;")
                L.append(ind + "    for (int j = lane; j < %d; j += 64) ok = ok && (__hip_atomic_load(fl + j, %s) == want64);" % (NJ, AG))
                L.append(ind + "    if (__all(ok)) break;")
                L.append(ind + "    if (spin > %d || ((spin & 255) == 255 && __hip_atomic_load(errp, %s) != 0u)) "
                         "{ __hip_atomic_store(errp, 1u, %s); break; }" % (SPIN_LIMIT, AG, AG))
                L.append(ind + "    __builtin_amdgcn_s_sleep(1);")
                L.append(ind + "  }")
                L.append(ind + "}")
                L.append(ind + "__syncthreads();")
                stamp("p%d flags seen" % pi)
                PT = 16 * K // 4 // 256          # float4 per thread
                L.append(ind + "const unsigned so_ = (unsigned)((%d + (%s & 3) * %d + (i64)bi * %d) * 8);"
                         % (po_, step_expr, lpp, 16 * K // 2))
                L.append(ind + "u4v g[%d];" % PT)
                for u in range(PT):
                    L.append(ind + "{ const int idx = %d * 256 + tid; const u4v z_ = {0u, 0u, 0u, 0u}; g[%d] = (idx / %d < vrows) ? "
                             "__builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)idx * 16u, so_, 16) : z_; }"
                             % (u, u, K // 4))
                for u in range(PT):
                    L.append(ind + "{ const int idx = %d * 256 + tid; "
                             "*(u4v*)(Hl + %d + (idx / %d) * %d + 4 * (idx %% %d)) = g[%d]; }"
                             % (u, so, K // 4, P, K // 4, u))
                L.append("    }")
                continue
            xo, lp = xoff[src]
            L.append(ind + "const u64* src = a.xch + %d + (%s & 3) * %d + (i64)bi * %d;" % (xo, step_expr, lp, 16 * K))
            L.append(ind + "const unsigned want = base + (unsigned)%s + 1u;" % step_expr)
            L.append(ind + "for (int ch = 0; ch < %d; ++ch) {" % nchunk)
            L.append(ind + "  u64 g[%d];" % cl)
            L.append(ind + "  for (int spin = 0;; ++spin) {")
            L.append(ind + "    bool ok = true;")
            for u in range(cl):
                L.append(ind + "    { const int idx = (ch * %d + %d) * 256 + tid; "
                         "if (idx / %d < vrows) { g[%d] = __hip_atomic_load(src + idx, %s); "
                         "ok = ok && ((unsigned)(g[%d] >> 32) == want); } else g[%d] = 0; }"
                         % (cl, u, K, u, AG, u, u))
            L.append(ind + "    if (ok) break;")
            L.append(ind + "    if (spin > %d || ((spin & 255) == 255 && __hip_atomic_load(errp, %s) != 0u)) "
                     "{ __hip_atomic_store(errp, 1u, %s); break; }" % (SPIN_LIMIT, AG, AG))
            L.append(ind + "    __builtin_amdgcn_s_sleep(1);")
            L.append(ind + "  }")
            for u in range(cl):
                L.append(ind + "  { const int idx = (ch * %d + %d) * 256 + tid; "
                         "Hl[%d + (idx / %d) * %d + idx %% %d] = __uint_as_float((unsigned)g[%d]); }"
                         % (cl, u, so, K, P, K, u))
            L.append(ind + "}")
            L.append("    }")
        D = len(ph["dots"])
        if D:
            if newly or not early:
                L.append("    __syncthreads();")
            stamp("p%d operand in LDS" % pi)
            for d, a_, x in late:
                emit_mfma(pi, d, a_, x)
            L.append("    __syncthreads();")
            stamp("p%d products done" % pi)
        for d in range(D):
            L.append("    const float dot_%d_%d = part[%d][0][tid] + part[%d][1][tid] + part[%d][2][tid] + part[%d][3][tid];"
                     % (pi, d, d, d, d, d))
        L.append("    if (owner) {")
        ins = ["dot_%d_%d" % (pi, d) for d in range(D)] + ["own_%d" % v for v in ph["ins"]]
        lines, outs, odts = cg.emit_scalar_body(ph["scalar"], ins, ["float32"] * len(ins),
                                                indent="      ", suffix="_p%d" % pi)
        L.extend(lines)
        for o, ri in zip(ph["outs"], ph["out_refs"]):
            L.append("      own_%d = %s;" % (o, cg._cast(outs[ri], odts[ri], "float32")))
            if o in xoff and not FLAG:
                xo, lp = xoff[o]
                L.append("      __hip_atomic_store(a.xch + %d + (t & 3) * %d + (i64)bi * %d + erow * %d + en, "
                         "((u64)(base + (unsigned)t + 1u) << 32) | (u64)__float_as_uint(own_%d), %s);"
                         % (xo, lp, 16 * N, N, o, AG))
            for _kind, j in out_of.get(o, []):
                L.append("      ((float*)a.out[%d])[((a.out_pos0[%d] + t) %% a.out_store[%d]) * a.out_ts[%d] + "
                         "eb * a.out_rs[%d] + en] = own_%d;" % (j, j, j, j, j, o))
        L.append("    }")
        if FLAG:
            # Elemwise-only phases run back to back in the element owner's registers: what they
            # have to publish goes out with ONE store-acknowledge wait and barrier, right before
            # the next phase that stages an operand (or at the end of the step)
            pending_pub.extend(o for o in ph["outs"] if o in xoff)
            nxt_has_dots = pi + 1 < len(pr.phases) and bool(pr.phases[pi + 1]["dots"])
            pub = pending_pub if (nxt_has_dots or pi + 1 == len(pr.phases)) else []
            if pub:
                pending_pub = []
            for o in pub:
                po_, lpp, fo_, lpf = xoff[o]
                L.append("    { const float nb_ = __shfl_down(own_%d, 1, 64);" % o)
                L.append("      if (owner && (ecol & 1) == 0) {")
                L.append("        union { float f[2]; u64 u; } pk; pk.f[0] = own_%d; pk.f[1] = nb_;" % o)
                L.append("        __hip_atomic_store(a.xch + %d + (t & 3) * %d + (i64)bi * %d + erow * %d + (en >> 1), pk.u, %s);"
                         % (po_, lpp, 16 * N // 2, N // 2, AG))
                L.append("      } }")
            if pub:
                stamp("p%d epilogue + payload stores issued" % pi)
                # payload complete (write-through stores acknowledged) before the tag is raised
                L.append('    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");')
                stamp("p%d own stores acknowledged" % pi)
                L.append("    __syncthreads();")
                stamp("p%d all stores acknowledged" % pi)
                for o in pub:
                    po_, lpp, fo_, lpf = xoff[o]
                    L.append("    if (tid == 0) __hip_atomic_store(a.xch + %d + (t & 3) * %d + (i64)bi * %d + nj, "
                             "(unsigned long long)(base + (unsigned)t + 1u), %s);" % (fo_, lpf, NJ, AG))
    stamp("step end (tag raised)")
    for v, nv in pr.new_of_state.items():
        L.append("    own_%d = own_%d;" % (v, nv))
    L.append("  }")
    if spec.trace:
        L.append("  if (tr_on) { tr[%d] = __builtin_amdgcn_s_memtime(); tr[%d] = __builtin_amdgcn_s_memrealtime(); }"
                 % (TRACE_NT * TRACE_MARKS + 2, TRACE_NT * TRACE_MARKS + 3))
        spec.marks = list(marks)
    L.append("  if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(a.ctl, base + (unsigned)a.T, %s);" % AG)
    L.append("}")
    return "\n".join(L) + "\n", (name,)


def _generate_frag(spec: SpecMat, name, xoff, xtot):
    """The "frag" exchange form (see ``SpecMat``).  Exchange buffer of one operand block (16 rows
    x K floats, K = N): float4 number ``(w * K/64 + q) * 64 + lane`` holds, for lane
    (r16 = lane & 15, grp = lane >> 4) of consumer wavefront w, the A-operand values
    ``h[r16][w*K/4 + grp*K/16 + 4q .. +3]`` — exactly MFMA k-steps 4q .. 4q+3 of that lane, so the
    q-th load of a wavefront is one coalesced 1 KiB line set.  A producer (column slice nj) owns
    16 consecutive k: four float4 per row, written with 16-byte sc1 (write-through) stores."""
    pr, B, N, NB, NJ = spec.prog, spec.B, spec.N, spec.NB, spec.NJ
    AG = "__ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT"
    F64 = spec.dtype == "float64"
    T = "double" if F64 else "float"          # element type
    PV = 2 if F64 else 4                      # values per 16-byte vector
    VT, AT = ("d2v", "d4v") if F64 else ("f4", "f4")       # fragment vector / MFMA accumulator
    ISZ = 8 if F64 else 4
    ZERO = "0.0" if F64 else "0.f"
    MFMA = "__builtin_amdgcn_mfma_f64_16x16x4f64" if F64 else "__builtin_amdgcn_mfma_f32_16x16x4f32"
    L = [cg.PRELUDE, SM_STRUCT, "typedef unsigned u4v __attribute__((ext_vector_type(4)));",
         "typedef double d2v __attribute__((ext_vector_type(2)));",
         "typedef double d4v __attribute__((ext_vector_type(4)));",
         "template <typename T> __device__ __forceinline__ T shfl_down_(T v, int d) {\n"
         "  if constexpr (sizeof(T) == 8) { union { T t; int i[2]; } u; u.t = v;\n"
         "    u.i[0] = __shfl_down(u.i[0], d, 64); u.i[1] = __shfl_down(u.i[1], d, 64); return u.t; }\n"
         "  else { union { T t; int i; } u; u.t = v; u.i = __shfl_down(u.i, d, 64); return u.t; } }"]
    keys = []
    for ph in pr.phases:
        for a_, x in ph["dots"]:
            key = (x, "prev" if x in pr.state else "cur")
            if key not in keys:
                keys.append(key)
    K = N
    Q = K // (16 * PV)                # 16-byte fragments per lane per operand block
    PW = NJ // 4                      # producers (column slices) feeding one wavefront's K quarter
    ndots_max = max(len(ph["dots"]) for ph in pr.phases)
    XF = spec.xfold["items"] if spec.xfold else []
    xf_of = {(pi, d): (gi, slot) for gi, (_v, pi, d, slot) in enumerate(XF)}
    folded = {v for v, _pi, _d, _s in XF}
    L.append('extern "C" __global__ __launch_bounds__(256) void %s(SmArgs a) {' % name)
    L.append("  __shared__ %s part[2][%d][4][256];" % (T, max(ndots_max, 1)))
    XREG = set(spec.xfold.get("reg", ())) if spec.xfold else set()     # weight columns in registers
    lds_of = {gi: n for n, gi in enumerate(gi for gi in range(len(XF)) if gi not in XREG)}
    if lds_of:
        L.append("  __shared__ __attribute__((aligned(16))) float Wl[%d];" % (len(lds_of) * K * 16))
    L.append("  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;")
    L.append("  const int r16 = lane & 15, grp = lane >> 4;")
    NBLK = spec.nblk
    assert NBLK == 1 or (not XF and not spec.trace and NB % NBLK == 0)

    def per_block(lines):
        """instantiate template lines per batch block of the workgroup: '§' = the block's name
        suffix ('' with one block per workgroup), '¤' = its number"""
        out = []
        for blk in range(NBLK):
            sfx = "" if NBLK == 1 else "_b%d" % blk
            out.extend(l.replace("§", sfx).replace("¤", str(blk)) for l in lines)
        return out
    if NBLK == 1:
        L.append("  const int bi = blockIdx.x %% %d, nj = blockIdx.x / %d;" % (NB, NB))
        L.append("  const int erow = tid >> 4, ecol = tid & 15;        // tile element owned by this thread")
        L.append("  const i64 eb = (i64)bi * 16 + erow, en = (i64)nj * 16 + ecol;")
        L.append("  const bool owner = eb < %d && en < %d;" % (B, spec.Nt))
        L.append("  const int vrows = (%d - bi * 16) < 16 ? (%d - bi * 16) : 16;   // valid rows of this batch block" % (B, B))
    else:
        L.append("  const int nj = blockIdx.x / %d;" % (NB // NBLK))
        L.append("  const int erow = tid >> 4, ecol = tid & 15;        // tile element owned by this thread")
        L.append("  const i64 en = (i64)nj * 16 + ecol;")
        L.extend(per_block([
            "  const int bi§ = (blockIdx.x %% %d) * %d + ¤;" % (NB // NBLK, NBLK),
            "  const i64 eb§ = (i64)bi§ * 16 + erow;",
            "  const bool owner§ = eb§ < %d && en < %d;" % (B, spec.Nt),
            "  const int vrows§ = (%d - bi§ * 16) < 16 ? (%d - bi§ * 16) : 16;" % (B, B)]))
    L.append("  const unsigned base = __hip_atomic_load(a.ctl, %s);" % AG)
    L.append("  unsigned* errp = a.ctl + 1;")
    L.append("  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, %du, 0x00020000);"
             % (xtot * 8))
    # where this workgroup's 16 columns live in a consumer's fragment order
    L.append("  const int pk = nj * 16 + (ecol & ~%d);                      // first of %d consecutive k" % (PV - 1, PV))
    L.append("  const int pw = pk / %d, pkk = pk %% %d;" % (K // 4, K // 4))
    L.append("  const unsigned pub_off = (unsigned)((((pw * %d + (pkk %% %d) / %d) * 64) + (pkk / %d) * 16 + erow) * 16);"
             % (Q, K // 16, PV, K // 16))
    L.append("  const unsigned ld_off = (unsigned)(((wave * %d) * 64 + lane) * 16);" % Q)
    marks = []
    FENCE = int(knobs.get("SM_FENCE")) != 0

    ASM_MARKS = int(knobs.get("SM_ASM_MARKS")) != 0

    def stamp(label):
        if not spec.trace:
            if FENCE:
                L.append("    __builtin_amdgcn_sched_barrier(0);")
                if ASM_MARKS:
                    L.append('    asm volatile("; @@ %s");' % label)
                    L.append("    __builtin_amdgcn_sched_barrier(0);")
            return
        k = len(marks)
        marks.append(label)
        assert k < TRACE_MARKS
        L.append("    if (tr_on && t >= %d && t < %d) tr[(t - %d) * %d + %d] = __builtin_amdgcn_s_memtime();"
                 % (TRACE_T0, TRACE_T0 + TRACE_NT, TRACE_T0, TRACE_MARKS, k))
    if spec.trace:
        L.append("  const bool tr_on = tid == 0 && (blockIdx.x == 0 || blockIdx.x == %d);" % (NB * NJ // 2))
        L.append("  unsigned long long* tr = (unsigned long long*)(a.ctl + 16) + (blockIdx.x == 0 ? 0 : %d);"
                 % (TRACE_NT * TRACE_MARKS + 4))
        L.append("  if (tr_on) { tr[%d] = __builtin_amdgcn_s_memtime(); tr[%d] = __builtin_amdgcn_s_memrealtime(); }"
                 % (TRACE_NT * TRACE_MARKS, TRACE_NT * TRACE_MARKS + 1))
    # ---- weights -> registers, MFMA B layout (as in the LDS form)
    for av, slot in sorted(pr.mats.items(), key=lambda t: t[1]):
        Km = spec.Ks[av]
        assert Km == K
        L.append("  const i64 wk%d = (i64)wave * %d + grp * %d;" % (slot, Km // 4, Km // 16))
        for s_ in range(Km // 16):
            L.append("  %s%s w%d_%d = ((const %s*)a.mat[%d])[(wk%d + %d) * a.mat_rs[%d] + nj * 16 + r16];"
                     % ("" if spec.pin else "const ", T, slot, s_, T, slot, slot, s_, slot))
    # ---- in-kernel sequence products (xfold): the weight columns of x_t @ W_g in LDS, in MFMA
    #      B-fragment order: float4 (w*Q + q)*64 + lane = W_g[w*K/4 + grp*K/16 + 4q .. +3][nj*16 + r16]
    for gi, (_v, _pi, _d, slot) in enumerate(XF):
        if gi in XREG:          # MFMA B layout in registers, like the recurrent weights
            L.append("  const i64 wxk%d = (i64)wave * %d + grp * %d;" % (gi, K // 4, K // 16))
            for s_ in range(K // 16):
                L.append("  float wx%d_%d = ((const float*)a.mat[%d])[(wxk%d + %d) * a.mat_rs[%d] + nj * 16 + r16];"
                         % (gi, s_, slot, gi, s_, slot))
            continue
        L.append("  for (int f = tid; f < %d; f += 256) {" % (K * 16 // 4))
        L.append("    const int ln = f & 63, qq = (f >> 6) %% %d, ww = (f >> 6) / %d;" % (Q, Q))
        L.append("    const i64 k0 = (i64)ww * %d + (ln >> 4) * %d + 4 * qq;" % (K // 4, K // 16))
        L.append("    const float* wp = (const float*)a.mat[%d] + k0 * a.mat_rs[%d] + nj * 16 + (ln & 15);" % (slot, slot))
        L.append("    const f4 wv = {wp[0], wp[a.mat_rs[%d]], wp[2 * a.mat_rs[%d]], wp[3 * a.mat_rs[%d]]};" % (slot, slot, slot))
        L.append("    *(f4*)(Wl + %d + 4 * f) = wv;" % (lds_of[gi] * K * 16))
        L.append("  }")
    XW = None
    XTAIL = max(1, int(knobs.get("SM_XTAIL")))  # fragments (4 MFMAs each) behind the payload loads

    def emit_xload(step_expr, ind):
        """x of step ``step_expr`` -> xfr: one buffer over that step's x (uniform base, the
        fragment index in the scalar offset: no per-load address registers)"""
        L.append(ind + "{ const __amdgpu_buffer_rsrc_t xs_ = __builtin_amdgcn_make_buffer_rsrc("
                 "(void*)((const float*)a.seq[%d] + (i64)(%s) * a.seq_ts[%d]), 0, %du, 0x00020000);"
                 % (spec.xfold["sx"], step_expr, spec.xfold["sx"], NB * 16 * K * 4))
        for q in range(Q):
            L.append(ind + "  xfr[%d] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xs_, xl_off, %du, 0));"
                     % (q, q * 64 * 16))
        L.append(ind + "}")

    def emit_xunits(units, ind, between=None):
        """The MFMA k-steps 4q .. 4q+3 of x @ W_<gi> on this wavefront's K quarter into accx<gi>,
        for every (gi, q) of ``units``.  Weight columns in LDS are read one unit AHEAD into the
        other of two registers quads (an LDS read in front of each group of four MFMAs, waited for
        at once, costs as much as the MFMAs).  ``between(i)``: code emitted in front of unit i."""
        ldsu = [i for i, (gi, _q) in enumerate(units) if gi not in XREG]    # units that read LDS
        nxt_lds = {ldsu[n]: ldsu[n + 1] for n in range(len(ldsu) - 1)}
        slot_of = {i: n & 1 for n, i in enumerate(ldsu)}
        if ldsu:
            L.append(ind + "bw0 = wl%d[%d];" % (units[ldsu[0]][0], units[ldsu[0]][1] * 64))
        for i, (gi, q) in enumerate(units):
            if between is not None:
                between(i)
            if q == 0:
                L.append(ind + "accx%d_0 = zero4; accx%d_1 = zero4;" % (gi, gi))
            if gi in XREG:
                for e, c in enumerate("xyzw"):
                    L.append(ind + "accx%d_%d = __builtin_amdgcn_mfma_f32_16x16x4f32(xfr[%d].%s, wx%d_%d, accx%d_%d, 0, 0, 0);"
                             % (gi, e & 1, q, c, gi, 4 * q + e, gi, e & 1))
                continue
            if i in nxt_lds:
                gj, qj = units[nxt_lds[i]]
                L.append(ind + "bw%d = wl%d[%d];" % (slot_of[nxt_lds[i]], gj, qj * 64))
                L.append(ind + "__builtin_amdgcn_sched_barrier(0);     // (the scheduler sinks the read to its use)")
            bs = ["bw%d.%s" % (slot_of[i], c) for c in "xyzw"]
            for e, c in enumerate("xyzw"):
                L.append(ind + "accx%d_%d = __builtin_amdgcn_mfma_f32_16x16x4f32(xfr[%d].%s, %s, accx%d_%d, 0, 0, 0);"
                         % (gi, e & 1, q, c, bs[e], gi, e & 1))
    if XF:
        XW = xfold_windows(pr, [(v, pi, d) for v, pi, d, _s in XF])
        assert XW is not None
        L.append("  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};")
        L.append("  f4 xfr[%d], bw0 = zero4, bw1 = zero4;" % Q)
        L.append("  const unsigned xl_off = (unsigned)((((i64)bi * %d) + (wave * %d) * 64 + lane) * 16);" % (K * 16 // 4, Q))
        emit_xload("0", "  ")
        for gi in range(len(XF)):
            L.append("  f4 accx%d_0 = zero4, accx%d_1 = zero4;" % (gi, gi))
            if gi in lds_of:
                L.append("  const f4* wl%d = (const f4*)(Wl + %d) + (wave * %d) * 64 + lane;" % (gi, lds_of[gi] * K * 16, Q))
        L.append("  __syncthreads();")
        # the products the first fetching phase of step 0 needs (later steps get theirs from the
        # last window of the step before)
        i_wrap0 = len(L)
        emit_xunits([(gi, q) for gi in sorted(XW["wrapped"]) for q in range(Q)], "  ")
        i_wrap1 = len(L)
        if XW["ahead"] == 2:
            L.append("  if (a.T > 1) {")
            emit_xload("1", "    ")
            L.append("  }")
    out_of = {}
    for o, kind, j in pr.outs:
        out_of.setdefault(o, []).append((kind, j))
    i_tpl = len(L)
    for v, k in pr.state.items():
        L.append("  %s own_%d§ = %s;" % (T, v, ZERO))
        L.append("  if (owner§) own_%d§ = ((const %s*)a.out[%d])[((a.out_pos0[%d] + a.out_store[%d] - 1) %% "
                 "a.out_store[%d]) * a.out_ts[%d] + eb§ * a.out_rs[%d] + en];" % (v, T, k, k, k, k, k, k))
    # older taps (-2, -3, ...: scan_perform.pyx:321-340): the tile element's own values of 2 ..
    # depth steps ago — registers of its owner, a shift per step (buffer row pos0 - d holds the
    # initial one; a depth no tap names still needs its register: taps [-1, -3] pass through -2)
    hist = {}
    by_kd = {kd: v for v, kd in pr.older.items()}
    for k, D_ in sorted(pr.depth.items()):
        for d in range(2, D_ + 1):
            nm = ("own_%d§" % by_kd[(k, d)]) if (k, d) in by_kd else "hist_%d_%d§" % (k, d)
            hist[(k, d)] = nm
            L.append("  %s %s = %s;" % (T, nm, ZERO))
            L.append("  if (owner§) %s = ((const %s*)a.out[%d])[((a.out_pos0[%d] + a.out_store[%d] * 8 - %d) %% "
                     "a.out_store[%d]) * a.out_ts[%d] + eb§ * a.out_rs[%d] + en];" % (nm, T, k, k, k, d, k, k, k))
    pw_nsq = sorted({v for ph in pr.phases for v in ph["ins"] if v in pr.nsq})
    for v in pw_nsq:
        s_ = pr.nsq[v]
        L.append("  %s own_%d§ = %s;" % (T, v, ZERO))
        L.append("  if (owner§) own_%d§ = ((const %s*)a.nsq[%d])[eb§ * a.nsq_rs[%d] + en * a.nsq_cs[%d]];"
                 % (v, T, s_, s_, s_))
    pw_seq = sorted({v for ph in pr.phases for v in ph["ins"] if v in pr.seq and v not in folded})
    for v in pw_seq:
        s_ = pr.seq[v]
        L.append("  %s nxt_%d§ = %s, own_%d§ = %s;" % (T, v, ZERO, v, ZERO))
        L.append("  if (owner§ && a.T > 0) nxt_%d§ = ((const %s*)a.seq[%d])[eb§ * a.seq_rs[%d] + en * a.seq_cs[%d]];"
                 % (v, T, s_, s_, s_))
    for ph in pr.phases:
        for o in ph["outs"]:
            L.append("  %s own_%d§ = %s;" % (T, o, ZERO))
    PUBLISH = spec.init == "publish"
    # the wait for the acknowledgement of the payload stores.  "publish" form: as an s_waitcnt the
    # compiler's wait-count pass sees (after an inline-asm wait it still believes the x loads of the
    # step before are pending and puts a vmcnt(0) in front of the first MFMA that reads x — where
    # only the tag store of wavefront 0 is pending: 0.3 us of skew between the wavefronts)
    WAIT_ACK = ('    __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory");' if PUBLISH else
                '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");')
    INIT_TAG = "(1ull << 32)"          # tags of the initial state: bit 32 set (step tags are 32-bit counts)
    prev_ops = [x for (x, kind) in keys if kind == "prev"]
    if PUBLISH and prev_ops:
        # the initial state goes through the exchange like every later one: slot (0 - 1) & 3 = 3
        L.append("  if (a.T > 0) {")
        for x in prev_ops:
            po_, lpp, fo_, lpf = xoff[pr.new_of_state.get(x, x)]
            L.append("    { const %s v0_ = owner§ ? own_%d§ : %s;" % (T, x, ZERO))
            L.append("      " + " ".join("const %s v%d_ = shfl_down_<%s>(v0_, %d);" % (T, e, T, e) for e in range(1, PV)))
            L.append("      if ((ecol & %d) == 0 && eb§ < %d) {" % (PV - 1, B))
            L.append("        const %s pv = {%s};" % (VT, ", ".join("v%d_" % e for e in range(PV))))
            L.append("        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, pv), xr, pub_off, "
                     "(unsigned)((%d + 3 * %d + (i64)bi§ * %d) * 8), 16);" % (po_, lpp, 16 * K * ISZ // 8))
            L.append("      } }")
        L.append(WAIT_ACK)
        L.append("    __syncthreads();")
        for x in prev_ops:
            po_, lpp, fo_, lpf = xoff[pr.new_of_state.get(x, x)]
            L.append("    if (tid == 0) __hip_atomic_store(a.xch + %d + 3 * %d + (i64)bi§ * %d + nj, "
                     "%s | (unsigned long long)base, %s);" % (fo_, lpf, NJ, INIT_TAG, AG))
        L.append("  }")
    L[i_tpl:] = per_block(L[i_tpl:])
    L.append("  for (i64 t = 0; t < a.T; ++t) {")
    # the operand fragments are per-step values: declared here, one whose last product lies
    # before another's fetch lends it its registers (a gradient step fetches three, two at a time)
    # (several blocks per workgroup, phases interleaved: an operand used by a later phase too must
    # survive the other blocks' fetches in between -> its own registers per block)
    ph_of = {}
    for pi_, ph_ in enumerate(pr.phases):
        for _a, x_ in ph_["dots"]:
            ph_of.setdefault((x_, "prev" if x_ in pr.state else "cur"), set()).add(pi_)
    # measured (B = 128, H = 1024): the backward GRU kernel (every operand used by one phase)
    # gains 11 % on the training step interleaved; the forward kernel (h feeds two phases: 64 more
    # registers per block, and its early product already covered the wait) loses 5 % -> interleave
    # only when no operand needs registers per block ("2" forces it, "0" turns it off)
    ilv_env = str(knobs.get("SM_INTERLEAVE"))
    ILV = NBLK > 1 and (ilv_env == "2" or (ilv_env == "1" and all(len(v_) == 1 for v_ in ph_of.values())))
    frs = ["fr%d%s" % (ki, "§" if (ILV and len(ph_of[keys[ki]]) > 1) else "") for ki in range(len(keys))]
    for ki in range(len(keys)):
        if frs[ki].endswith("§"):
            L.extend(per_block(["    %s %s[%d];" % (VT, frs[ki], Q)]))
        else:
            L.append("    %s fr%d[%d];" % (VT, ki, Q))
    if spec.pin:
        for av, slot in sorted(pr.mats.items(), key=lambda t: t[1]):
            for s0 in range(0, spec.Ks[av] // 16, 16):
                ops = ", ".join('"+%s"(w%d_%d)' % ("a" if spec.pin == 2 else "v", slot, s_)
                                for s_ in range(s0, min(s0 + 16, spec.Ks[av] // 16)))
                L.append('    asm volatile("" : %s);' % ops)
    if spec.pin:
        for gi in sorted(XREG):
            for s0 in range(0, K // 16, 16):
                ops = ", ".join('"+%s"(wx%d_%d)' % ("a" if spec.pin == 2 else "v", gi, s_)
                                for s_ in range(s0, min(s0 + 16, K // 16)))
                L.append('    asm volatile("" : %s);' % ops)
    i_body = len(L)              # the step body: a template, instantiated per batch block below
    nxt_loads = []
    for v in pw_seq:
        s_ = pr.seq[v]
        L.append("    own_%d§ = nxt_%d§;" % (v, v))
        nxt_loads.append("    if (owner§ && t + 1 < a.T) nxt_%d§ = ((const %s*)a.seq[%d])[(t + 1) * a.seq_ts[%d] + "
                         "eb§ * a.seq_rs[%d] + en * a.seq_cs[%d]];" % (v, T, s_, s_, s_, s_))
    i_top = len(L)
    if spec.nxt == "top":
        L.extend(nxt_loads)
        nxt_loads = []
    staged_this_step = set()

    def want_expr(step_expr, kind):
        """the tag a consumer of step ``step_expr``'s value waits for (initial state: INIT_TAG | base)"""
        e = "(unsigned long long)(base + (unsigned)%s + 1u)" % step_expr
        if kind == "prev" and PUBLISH:
            e = "(%s | ((unsigned long long)(t == 0) << 32))" % e
        return e

    def acc_init(pi, d):
        """what the two accumulator chains of product (pi, d) start from"""
        if (pi, d) in xf_of:        # continue the accumulation the sequence product started
            gi_ = xf_of[(pi, d)][0]
            nm = "accs" if gi_ in snap else "accx"
            return "%s%d_0" % (nm, gi_), "%s%d_1" % (nm, gi_)
        z = "{%s}" % ", ".join([ZERO] * 4)
        return z, z

    def emit_mfma(pi, d, a_, x, after_q=None, q0=0, q1=None, acc=None):
        """one product on this wavefront's K quarter: two independent accumulator chains (a chain
        of dependent 16x16x4 MFMAs issues every 40 cycles, two interleaved ones every 32).
        ``q0 .. q1``: a part of the quarter — ``acc``: the accumulators' name prefix, declared by
        the caller (a product issued in two parts, around a hand-off)"""
        slot = pr.mats[a_]
        ki = keys.index((x, "prev" if x in pr.state else "cur"))
        q1 = Q if q1 is None else q1
        if acc is None:
            L.append("    {")
            L.append("      %s acc0 = %s, acc1 = %s;" % ((AT,) + acc_init(pi, d)))
            n0, n1 = "acc0", "acc1"
        else:
            n0, n1 = acc + "_0", acc + "_1"
        for q in range(q0, q1):
            for e, c in enumerate("xyzw"[:PV]):
                nm = (n0, n1)[(PV * q + e) & 1]
                L.append("      %s = %s(%s[%d].%s, w%d_%d, %s, 0, 0, 0);"
                         % (nm, MFMA, frs[ki], q, c, slot, PV * q + e, nm))
            if after_q is not None:
                after_q(q)
        if q1 == Q:
            # C/D rows of a lane: f32 16x16x4 -> 4 * grp + i ; f64 16x16x4 -> grp + 4 * i
            L.append("      for (int i = 0; i < 4; ++i) part[pp%d][%d][wave][(%s) * 16 + r16] = %s[i] + %s[i];"
                     % (pi, d, "grp + 4 * i" if F64 else "4 * grp + i", n0, n1))
        if acc is None:
            L.append("    }")

    POLLS = max(1, int(spec.polls)) if PW <= 64 else 1

    def emit_spin(ind, fls, want="want64", delay=False):
        """wait until every tag of this wavefront's producers (``fls``: one tag array per operand)
        reads ``want``.  POLLS looks in flight: the oldest is examined while the others travel, a
        new one replaces it — the tags are found a quarter of a round trip after they arrive, not
        up to a whole one (r04 timeline of the gradient kernel: 2.0-2.7 us from a workgroup's own
        tag to its loads, of which ~1.1 are the tag's way there and one look's way back)."""
        if delay and spec.spin_delay > 0:
            for _ in range(spec.spin_delay // 15):
                L.append(ind + "__builtin_amdgcn_s_sleep(15);")
            if spec.spin_delay % 15:
                L.append(ind + "__builtin_amdgcn_s_sleep(%d);" % (spec.spin_delay % 15))
        if POLLS == 1:
            L.append(ind + "for (int spin = 0;; ++spin) {")
            L.append(ind + "  bool ok = true;")
            for f in fls:
                L.append(ind + "  for (int j = lane; j < %d; j += 64) ok = ok && (__hip_atomic_load(%s + j, %s) == %s);"
                         % (PW, f, AG, want))
            L.append(ind + "  if (__all(ok)) break;")
            L.append(ind + "  if (spin > %d || ((spin & 255) == 255 && __hip_atomic_load(errp, %s) != 0u)) "
                     "{ __hip_atomic_store(errp, 1u, %s); break; }" % (SPIN_LIMIT, AG, AG))
            L.append(ind + "  __builtin_amdgcn_s_sleep(1);")
            L.append(ind + "}")
            return
        L.append(ind + "{")
        for n, f in enumerate(fls):
            L.append(ind + "  const u64* fp%d_ = %s + (lane < %d ? lane : 0);" % (n, f, PW))
        for d_ in range(POLLS):
            for n in range(len(fls)):
                L.append(ind + "  unsigned long long pl%d_%d = __hip_atomic_load(fp%d_, %s);" % (n, d_, n, AG))
            if d_ + 1 < POLLS:
                L.append(ind + "  __builtin_amdgcn_s_sleep(3);")
        L.append(ind + "  for (int spin = 0;; ++spin) {")
        L.append(ind + "    if (__all(%s)) break;" % " && ".join("pl%d_0 == %s" % (n, want) for n in range(len(fls))))
        L.append(ind + "    if (spin > %d || ((spin & 255) == 255 && __hip_atomic_load(errp, %s) != 0u)) "
                 "{ __hip_atomic_store(errp, 1u, %s); break; }" % (SPIN_LIMIT, AG, AG))
        for n in range(len(fls)):
            for d_ in range(POLLS - 1):
                L.append(ind + "    pl%d_%d = pl%d_%d;" % (n, d_, n, d_ + 1))
            L.append(ind + "    pl%d_%d = __hip_atomic_load(fp%d_, %s);" % (n, POLLS - 1, n, AG))
        L.append(ind + "  }")
        L.append(ind + "}")

    def emit_fetch(pi, x, kind, bare=False):
        """wait for the tags of this wavefront's producers, then pull its K quarter into fr.
        ``bare``: nothing was issued between this workgroup's own tag and this wait — the first
        poll is held back (SM_SPIN_DELAY) until it can find the tags"""
        ki = keys.index((x, kind))
        src = pr.new_of_state.get(x, x)
        po_, lpp, fo_, lpf = xoff[src]
        if kind == "prev" and PUBLISH:
            L.append("    {")
            step_expr = "(t - 1)"
        elif kind == "prev":
            k_out = pr.state[x]
            L.append("    if (t == 0) {")
            L.append("      const %s* ini = (const %s*)a.out[%d] + ((a.out_pos0[%d] + a.out_store[%d] - 1) %% "
                     "a.out_store[%d]) * a.out_ts[%d] + ((i64)bi§ * 16 + r16) * a.out_rs[%d] + wave * %d + grp * %d;"
                     % (T, T, k_out, k_out, k_out, k_out, k_out, k_out, K // 4, K // 16))
            L.append("      const int c0 = wave * %d + grp * %d;" % (K // 4, K // 16))
            for q in range(Q):
                L.append("      { %s v = {%s};" % (VT, ", ".join([ZERO] * PV)))
                L.append("        if (r16 < vrows§) { for (int e = 0; e < %d; ++e) if (c0 + %d + e < %d) v[e] = ini[%d + e]; }"
                         % (PV, PV * q, spec.Nt, PV * q))
                L.append("        %s[%d] = v; }" % (frs[ki], q))
            # nothing of this path pending where the two meet: the compiler merges the two load
            # orders into an s_waitcnt vmcnt(0) in front of the SECOND MFMA of every step
            L.append("      __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)")
            L.append("    } else {")
            step_expr = "(t - 1)"
        else:
            L.append("    {")
            step_expr = "t"
        ind = "      "
        L.append(ind + "const unsigned long long want64 = %s;" % want_expr(step_expr, kind))
        L.append(ind + "const u64* fl = a.xch + %d + (%s & 3) * %d + (i64)bi§ * %d + wave * %d;"
                 % (fo_, step_expr, lpf, NJ, PW))
        emit_spin(ind, ["fl"], delay=bare)
        L.append(ind + "const unsigned so_ = (unsigned)((%d + (%s & 3) * %d + (i64)bi§ * %d) * 8);"
                 % (po_, step_expr, lpp, 16 * K * ISZ // 8))
        for q in range(Q):
            L.append(ind + "{ const u4v g = __builtin_amdgcn_raw_buffer_load_b128(xr, ld_off + %du, so_, 16);"
                     " %s[%d] = __builtin_bit_cast(%s, g); }" % (q * 64 * 16, frs[ki], q, VT))
        L.append("    }")

    def emit_fetch_joint(pi, ops, bare=False):
        """Several operands of the current step published by the same epilogue (the two
        last-phase products of a gradient step): ONE polling pass over all their tags — fetched
        one after the other, the second wait costs a tag round trip (0.6 us) although its tags
        are there — then all the loads in flight together."""
        ind = "      "
        L.append("    {")
        L.append(ind + "const unsigned long long want64 = (unsigned long long)(base + (unsigned)t + 1u);")
        for n, (x, kind) in enumerate(ops):
            po_, lpp, fo_, lpf = xoff[pr.new_of_state.get(x, x)]
            L.append(ind + "const u64* fl%d = a.xch + %d + (t & 3) * %d + (i64)bi§ * %d + wave * %d;"
                     % (n, fo_, lpf, NJ, PW))
        emit_spin(ind, ["fl%d" % n for n in range(len(ops))], delay=bare)
        for n, (x, kind) in enumerate(ops):
            ki = keys.index((x, kind))
            po_, lpp, fo_, lpf = xoff[pr.new_of_state.get(x, x)]
            L.append(ind + "const unsigned so%d_ = (unsigned)((%d + (t & 3) * %d + (i64)bi§ * %d) * 8);"
                     % (n, po_, lpp, 16 * K * ISZ // 8))
            for q in range(Q):
                L.append(ind + "{ const u4v g = __builtin_amdgcn_raw_buffer_load_b128(xr, ld_off + %du, so%d_, 16);"
                         " %s[%d] = __builtin_bit_cast(%s, g); }" % (q * 64 * 16, n, frs[ki], q, VT))
        L.append("    }")

    def emit_look(pi, ind, assign=False):
        """the first look at the tags of phase pi's operand (examined after the head of the window);
        every lane loads (a clamped address): no branch, no copy the compiler would wait for"""
        decl = "" if assign else ("const unsigned long long " if PUBLISH else "const unsigned ")
        if PUBLISH:     # 64-bit: the low word of an initial-state tag is a step count of the launch before
            L.append(ind + "%stg0_w%d = __hip_atomic_load(fl_w%d + (lane < %d ? lane : 0), %s);"
                     % (decl, pi, pi, PW, AG))
        else:
            L.append(ind + "%stg0_w%d = __hip_atomic_load((const unsigned*)(fl_w%d + (lane < %d ? lane : 0)), %s);"
                     % (decl, pi, pi, PW, AG))

    def window_begin(pi, x, kind):
        """(sequence products in the loop) the first look at the tags of this phase's operand:
        started here — in front of the products on operands already in registers — and examined
        when the head of the window has been issued"""
        src = pr.new_of_state.get(x, x)
        po_, lpp, fo_, lpf = xoff[src]
        step_expr = "(t - 1)" if kind == "prev" else "t"
        ind = "    "
        if PUBLISH:
            L.append(ind + "const unsigned long long want64_w%d = %s;" % (pi, want_expr(step_expr, kind)))
        else:
            L.append(ind + "const unsigned want32_w%d = base + (unsigned)%s + 1u;" % (pi, step_expr))
        L.append(ind + "const u64* fl_w%d = a.xch + %d + (%s & 3) * %d + (i64)bi * %d + wave * %d;"
                 % (pi, fo_, step_expr, lpf, NJ, PW))
        L.append(ind + "const unsigned so_w%d = (unsigned)((%d + (%s & 3) * %d + (i64)bi * %d) * 8);"
                 % (pi, po_, step_expr, lpp, 16 * K * ISZ // 8))
        if pi in look_in_product:       # issued among the MFMAs of the product in front of the window
            L.append(ind + "unsigned long long tg0_w%d = 0ull;" % pi)
        elif not spec.look:
            emit_look(pi, ind)

    def emit_window(pi, x, kind, xunits, done=0):
        """The fetch of a phase with sequence products to run (r03 timelines of this kernel): a
        wavefront has two idle stretches here — until its producers' tags are visible (~1 us
        after its own tag) and while the payload is in flight (16 KiB per wavefront: ~0.4 us of
        load issue, which stalls the instruction stream when issued back to back, + ~0.7 us).
        HEAD: the fragments beyond the last XTAIL, issued while the first look at the tags is in
        flight; then the usual spin if that look did not find every tag.  TAIL: one payload load
        in front of each of the remaining fragments' four MFMAs.  ``done``: fragments of this
        window already issued in front of the previous hand-off's acknowledgement wait."""
        assert PW <= 64
        ki = keys.index((x, kind))
        xunits = list(xunits)[done:]
        n_head = max(0, len(xunits) - XTAIL)
        tail = xunits[n_head:]
        per = -(-Q // max(len(tail), 1))       # payload loads in front of each tail fragment

        def load(q, ind2):
            L.append(ind2 + "{ const u4v g = __builtin_amdgcn_raw_buffer_load_b128(xr, ld_off, so_w%d + %du, 16);"
                     " fr%d[%d] = __builtin_bit_cast(%s, g); }" % (pi, q * 64 * 16, ki, q, VT))
        if kind == "prev" and not PUBLISH:
            k_out = pr.state[x]
            L.append("    if (t == 0) {")
            L.append("      const %s* ini = (const %s*)a.out[%d] + ((a.out_pos0[%d] + a.out_store[%d] - 1) %% "
                     "a.out_store[%d]) * a.out_ts[%d] + ((i64)bi * 16 + r16) * a.out_rs[%d] + wave * %d + grp * %d;"
                     % (T, T, k_out, k_out, k_out, k_out, k_out, k_out, K // 4, K // 16))
            L.append("      const int c0 = wave * %d + grp * %d;" % (K // 4, K // 16))
            for q in range(Q):
                L.append("      { %s v = {%s};" % (VT, ", ".join([ZERO] * PV)))
                L.append("        if (r16 < vrows) { for (int e = 0; e < %d; ++e) if (c0 + %d + e < %d) v[e] = ini[%d + e]; }"
                         % (PV, PV * q, spec.Nt, PV * q))
                L.append("        fr%d[%d] = v; }" % (ki, q))
            emit_xunits(xunits, "      ")
            # nothing of this path pending where the two meet: the compiler merges the two
            # load orders into an s_waitcnt vmcnt(0) in front of the SECOND MFMA of every step
            L.append("      __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)")
            L.append("    } else {")
        else:
            L.append("    {")
        ind = "      "
        if pi in look_in_product:
            if pi not in look_emitted:      # the product that was to carry it was not split after all
                emit_look(pi, ind, assign=True)
            emit_xunits(xunits[:n_head], ind)
        elif spec.look:
            n_before = min(spec.look, n_head)
            emit_xunits(xunits[:n_before], ind)
            L.append(ind + "__builtin_amdgcn_sched_barrier(0);")
            emit_look(pi, ind)
            L.append(ind + "__builtin_amdgcn_sched_barrier(0);")
            emit_xunits(xunits[n_before:n_head], ind)
        else:
            emit_xunits(xunits[:n_head], ind)
        L.append(ind + "__builtin_amdgcn_sched_barrier(0);")
        if n_head or done:
            stamp("p%d window head issued" % pi)
        if PUBLISH:
            L.append(ind + "if (!__all(tg0_w%d == want64_w%d)) {" % (pi, pi))
            L.append(ind + "  const unsigned long long want64 = want64_w%d;" % pi)
        else:
            L.append(ind + "if (!__all(tg0_w%d == want32_w%d)) {" % (pi, pi))
            L.append(ind + "  const unsigned long long want64 = (unsigned long long)want32_w%d;" % pi)
        emit_spin(ind + "  ", ["fl_w%d" % pi])
        L.append(ind + "}")
        L.append(ind + "__builtin_amdgcn_sched_barrier(0);")
        stamp("p%d tags seen" % pi)
        def between(i):
            qs = range(i * per, min(Q, (i + 1) * per))
            if qs:
                L.append(ind + "__builtin_amdgcn_sched_barrier(0);")
                for q in qs:
                    load(q, ind)
                L.append(ind + "__builtin_amdgcn_sched_barrier(0);")
        emit_xunits(tail, ind, between)
        for q in range(len(tail) * per, Q):
            load(q, ind)
        L.append(ind + "__builtin_amdgcn_sched_barrier(0);")
        L.append("    }")

    pending_pub = []
    fetches = phase_fetches(pr)
    snap = set()
    if XW:
        last_f = max(XW["win"])
        snap = {gi for gi in XW["wrapped"] if XF[gi][1] == last_f}
    # `part` is double-buffered by the running count of product phases (across steps): a phase's
    # epilogue reads never meet the next product phase's writes, and the barrier of the phase in
    # between orders everything two phases apart — no barrier between hand-off and products
    dot_phases = [pi for pi, ph in enumerate(pr.phases) if ph["dots"]]
    nph = len(pr.phases)
    SCHED = bool(XW) and NBLK == 1 and PUBLISH
    done_early = set()          # (phase, dot) issued ahead, in front of the previous acknowledgement wait
    reload_early = None         # the phase whose first staged-operand product carries the x loads
    if SCHED and spec.xreload == "early" and spec.early_first:
        nx_ = XW["reload"] + 1
        if nx_ < nph and fetches[nx_][0] and fetches[XW["reload"]][1]:
            reload_early = nx_
    f0 = min((f for f in XW["win"] if XW["win"][f] and fetches[f][2]), default=None) if XW else None
    # a wrapped product (computed at the end of step t for step t + 1) may leave its last fragments
    # to the first window of step t + 1: the same x, more MFMAs in front of THAT hand-off
    moved = set()
    if SCHED and spec.xsplit and f0 is not None and f0 != last_f:
        moved = {gi_ for gi_ in XW["win"][last_f] if gi_ in XW["wrapped"] and gi_ not in snap}
    KEEP = min(max(spec.xsplit, 1), Q) if moved else Q

    def units_of(pi):
        u = []
        if pi == f0:
            u += [(gi_, q) for gi_ in sorted(moved) for q in range(KEEP, Q)]
        for gi_ in XW["win"][pi]:
            u += [(gi_, q) for q in range(KEEP if (gi_ in moved and pi == last_f) else Q)]
        return u
    # the staged-operand product of a phase issued in two parts around the hand-off in front of it
    split_early = {}            # (phase, dot) -> fragments already issued
    look_in_product = set()     # phases whose first look is issued among that product's MFMAs
    look_emitted = set()
    if SCHED and spec.ackfill and spec.early_first and 0 < spec.epre < Q:
        for pi_ in range(1, nph):
            if fetches[pi_][0] and XW["win"].get(pi_) and fetches[pi_][2] and 0 < spec.elook <= Q - spec.epre:
                look_in_product.add(pi_)
    if moved:                   # step 0's share of the moved products: the fragments that stay
        i_ins = len(L)
        emit_xunits([(gi_, q) for gi_ in sorted(XW["wrapped"]) for q in range(KEEP if gi_ in moved else Q)], "  ")
        repl = L[i_ins:]
        del L[i_ins:]
        shift = len(repl) - (i_wrap1 - i_wrap0)
        L[i_wrap0:i_wrap1] = repl
        i_body += shift
        i_top += shift
    pre_n = 0                   # fragments of phase f0's window issued at the end of the step before
    if SCHED and spec.xpre and f0 is not None:
        pre_n = min(spec.xpre, len(units_of(f0)))
        pre_units = units_of(f0)[:pre_n]
        if any(gi_ in snap for gi_, _q in pre_units):
            pre_n = 0           # that window's products start from what the fragments would overwrite
    if pre_n:
        i_ins = len(L)
        emit_xunits(pre_units, "    ")           # step 0's: in front of the loop (moved there below)
        pre0_lines = [l[2:] for l in L[i_ins:]]
        del L[i_ins:]
    if NBLK == 1:
        for pi in dot_phases:
            L.append("    const int pp%d = (int)((t * %d + %d) & 1);" % (pi, len(dot_phases), dot_phases.index(pi)))
    XLPG = 2                    # x loads behind each fragment's MFMAs of the product that carries them

    xs_declared = []

    def declare_xs():
        if xs_declared:
            return
        xs_declared.append(1)
        L.append("    const i64 xt_ = (t + %d < a.T) ? t + %d : a.T - 1;" % (XW["ahead"], XW["ahead"]))
        L.append("    const __amdgpu_buffer_rsrc_t xs_ = __builtin_amdgcn_make_buffer_rsrc("
                 "(void*)((const float*)a.seq[%d] + xt_ * a.seq_ts[%d]), 0, %du, 0x00020000);"
                 % (spec.xfold["sx"], spec.xfold["sx"], NB * 16 * K * 4))

    def xl_early(q):
        # nothing else is in flight here: the loads issue without waiting for queue slots; all of
        # them in the first half of the product (the rest of it covers their latency)
        qs = range(q * XLPG, min(Q, (q + 1) * XLPG))
        if qs:
            L.append("      __builtin_amdgcn_sched_barrier(0);")
            for q2 in qs:
                L.append("      xfr[%d] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xs_, xl_off, %du, 0));"
                         % (q2, q2 * 64 * 16))
            L.append("      __builtin_amdgcn_sched_barrier(0);")
    i_phase = []
    for pi, ph in enumerate(pr.phases):
        i_phase.append(len(L))
        L.append("    // ---- phase %d" % pi)
        if ph["dots"]:
            if NBLK == 1:
                pass
            elif ILV:   # product events in program order: phase by phase, block by block
                L.append("    const int pp%d = (int)((t * %d + %d + ¤) & 1);"
                         % (pi, NBLK * len(dot_phases), NBLK * dot_phases.index(pi)))
            else:       # product phases counted across the blocks of the workgroup
                L.append("    const int pp%d = (int)((t * %d + ¤ * %d + %d) & 1);"
                         % (pi, NBLK * len(dot_phases), len(dot_phases), dot_phases.index(pi)))
        stamp("p%d start" % pi)
        early, late, fresh = fetches[pi]
        if XW and pi == max(XW["win"]):
            for gi_ in sorted(snap):     # this window overwrites what this phase's products start from
                L.append("    const f4 accs%d_0 = accx%d_0, accs%d_1 = accx%d_1;" % (gi_, gi_, gi_, gi_))
        win_units = units_of(pi) if (XW and fresh) else []
        if win_units:
            window_begin(pi, *fresh[0])
        early_all = early
        early = [e_ for e_ in early if (pi, e_[0]) not in done_early]
        if pi == reload_early and len(early) == len(early_all):
            declare_xs()
        if spec.early_first:
            for li, (d, a_, x) in enumerate(early):
                if (pi, d) in split_early:          # the rest of a product begun before the hand-off
                    def look_hook(q, pi=pi, li=li):
                        if pi == reload_early and li == 0:      # the x loads the first part did not get to
                            xl_early(q)
                        if pi in look_in_product and q == split_early_q0 + spec.elook - 1:
                            look_emitted.add(pi)
                            L.append("      __builtin_amdgcn_sched_barrier(0);")
                            emit_look(pi, "      ", assign=True)
                            L.append("      __builtin_amdgcn_sched_barrier(0);")
                    split_early_q0 = split_early[(pi, d)]
                    emit_mfma(pi, d, a_, x, look_hook, q0=split_early_q0, acc="eacc%d_%d" % (pi, d))
                    continue
                emit_mfma(pi, d, a_, x, xl_early if (pi == reload_early and li == 0 and
                                                     len(early) == len(early_all)) else None)
        # a fetch with nothing in front of it: one block per workgroup (other blocks' phases are
        # work in between), no product on staged operands issued first, no window
        bare = NBLK == 1 and not (spec.early_first and early_all)
        if not win_units and len(fresh) >= 2 and all(kind == "cur" for _x, kind in fresh):
            emit_fetch_joint(pi, fresh, bare)
            staged_this_step.update(fresh)
        else:
            for fi, (x, kind) in enumerate(fresh):
                if win_units and fi == 0:
                    emit_window(pi, x, kind, win_units, pre_n if pi == f0 else 0)
                else:
                    emit_fetch(pi, x, kind, bare and fi == 0)
                staged_this_step.add((x, kind))
        if fresh:
            stamp("p%d tags seen, loads issued" % pi)
            if spec.nxt == "fetch":      # behind the payload loads in the in-order queue
                L.extend(nxt_loads)
                nxt_loads = []
        reload_here = bool(XW) and pi == XW["reload"] and bool(late) and reload_early is None
        if reload_here:
            # the next x: same registers; the step index clamped (no branch); one load behind
            # each fragment's MFMAs of the first product below (issued back to back, 16 loads of
            # 1 KiB stall the wavefront's instruction stream for ~0.4 us)
            L.append("    const i64 xt_ = (t + %d < a.T) ? t + %d : a.T - 1;" % (XW["ahead"], XW["ahead"]))
            L.append("    const __amdgpu_buffer_rsrc_t xs_ = __builtin_amdgcn_make_buffer_rsrc("
                     "(void*)((const float*)a.seq[%d] + xt_ * a.seq_ts[%d]), 0, %du, 0x00020000);"
                     % (spec.xfold["sx"], spec.xfold["sx"], NB * 16 * K * 4))
        if not spec.early_first:
            # the products on operands already in registers run while the new operand is in flight
            for d, a_, x in early:
                emit_mfma(pi, d, a_, x)
        if early:
            stamp("p%d early products done" % pi)
        for li, (d, a_, x) in enumerate(late):
            if reload_here and li == 0:
                def xl(q):
                    if spec.trace and q in (3, 7, 11):
                        stamp("p%d product, %d MFMAs issued" % (pi, 4 * q + 4))
                    L.append("      __builtin_amdgcn_sched_barrier(0);")
                    L.append("      xfr[%d] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xs_, xl_off, %du, 0));"
                             % (q, q * 64 * 16))
                    L.append("      __builtin_amdgcn_sched_barrier(0);")
                emit_mfma(pi, d, a_, x, xl)
            else:
                emit_mfma(pi, d, a_, x)
        if reload_here:
            stamp("p%d next x requested" % pi)
        D = len(ph["dots"])
        if D:
            if XW:
                stamp("p%d own products written" % pi)
            L.append("    __syncthreads();")
            stamp("p%d products done" % pi)
        for d in range(D):
            L.append("    const %s dot_%d_%d = part[pp%d][%d][0][tid] + part[pp%d][%d][1][tid] + part[pp%d][%d][2][tid] + part[pp%d][%d][3][tid];"
                     % (T, pi, d, pi, d, pi, d, pi, d, pi, d))
        L.append("    if (owner§) {")
        ins = ["dot_%d_%d" % (pi, d) for d in range(D)] + \
            [("0.f" if v in folded else "own_%d§" % v) for v in ph["ins"]]     # folded: already in the dot
        lines, outs, odts = cg.emit_scalar_body(ph["scalar"], ins, [spec.dtype] * len(ins),
                                                indent="      ", suffix="_p%d" % pi)
        L.extend(lines)
        for o, ri in zip(ph["outs"], ph["out_refs"]):
            L.append("      own_%d§ = %s;" % (o, cg._cast(outs[ri], odts[ri], spec.dtype)))
        L.append("    }")
        # publish first (the hand-off is the critical path), the stores into the output buffers after
        pending_pub.extend(o for o in ph["outs"] if o in xoff)
        nxt_has_dots = pi + 1 < len(pr.phases) and bool(pr.phases[pi + 1]["dots"])
        pub = pending_pub if (nxt_has_dots or pi + 1 == len(pr.phases)) else []
        if pub:
            pending_pub = []
        for o in pub:
            po_, lpp, fo_, lpf = xoff[o]
            L.append("    { const %s v0_ = owner§ ? own_%d§ : %s;" % (T, o, ZERO))
            L.append("      " + " ".join("const %s v%d_ = shfl_down_<%s>(v0_, %d);" % (T, e, T, e) for e in range(1, PV)))
            L.append("      if ((ecol & %d) == 0 && eb§ < %d) {" % (PV - 1, B))
            L.append("        const %s pv = {%s};" % (VT, ", ".join("v%d_" % e for e in range(PV))))
            L.append("        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, pv), xr, pub_off, "
                     "(unsigned)((%d + (t & 3) * %d + (i64)bi§ * %d) * 8), 16);" % (po_, lpp, 16 * K * ISZ // 8))
            L.append("      } }")
        if pub:
            stamp("p%d epilogue + payload stores issued" % pi)
        L.append("    if (owner§) {")
        for o in ph["outs"]:
            for _kind, j in out_of.get(o, []):
                L.append("      ((%s*)a.out[%d])[((a.out_pos0[%d] + t) %% a.out_store[%d]) * a.out_ts[%d] + "
                         "eb§ * a.out_rs[%d] + en] = own_%d§;" % (T, j, j, j, j, j, o))
        L.append("    }")
        if SCHED and pub and spec.ackfill and pi + 1 < nph and spec.early_first and fetches[pi + 1][0]:
            # products of the next phase that need nothing from this hand-off: issued while the
            # payload stores travel (the wait below costs 0.35 us with nothing in front of it) —
            # and in THEIR shadow the loads nobody waits for yet: the next step's epilogue operands
            # and the next x (in front of any other wait of the step an HBM-cold load holds up the
            # in-order queue; here it has the whole product to arrive)
            if spec.nxt == "ack":
                L.extend(nxt_loads)
                nxt_loads = []
            if pi + 1 == reload_early:
                declare_xs()
            for li, (d, a_, x) in enumerate(fetches[pi + 1][0]):
                if 0 < spec.epre < Q:
                    if li == 0:     # only its first fragments here: the tag goes out sooner
                        L.append("    %s eacc%d_%d_0 = %s, eacc%d_%d_1 = %s;"
                                 % ((AT, pi + 1, d, acc_init(pi + 1, d)[0], pi + 1, d, acc_init(pi + 1, d)[1])))
                        emit_mfma(pi + 1, d, a_, x, xl_early if pi + 1 == reload_early else None,
                                  q0=0, q1=spec.epre, acc="eacc%d_%d" % (pi + 1, d))
                        split_early[(pi + 1, d)] = spec.epre
                    continue
                emit_mfma(pi + 1, d, a_, x, xl_early if (pi + 1 == reload_early and li == 0) else None)
                done_early.add((pi + 1, d))
            stamp("p%d products of p%d issued" % (pi, pi + 1))
        if pre_n and pi == nph - 1:
            emit_xunits(pre_units, "    ")
            stamp("p%d fragments of the next window issued" % pi)
        if pub:
            L.append(WAIT_ACK)
            stamp("p%d own stores acknowledged" % pi)
            L.append("    __syncthreads();")
            for o in pub:
                po_, lpp, fo_, lpf = xoff[o]
                L.append("    if (tid == 0) __hip_atomic_store(a.xch + %d + (t & 3) * %d + (i64)bi§ * %d + nj, "
                         "(unsigned long long)(base + (unsigned)t + 1u), %s);" % (fo_, lpf, NJ, AG))
            stamp("p%d tag raised" % pi)
        elif D:
            # no publish between this phase's `part` reads and the next writes of the same parity
            # two phases on: the products-done barrier of the next phase orders them
            pass
    i_phase.append(len(L))
    if nxt_loads:               # no place found further down: at the top of the step after all
        L[i_top:i_top] = nxt_loads
        i_phase = [i + len(nxt_loads) if i >= i_top else i for i in i_phase]
        nxt_loads = []
    stamp("step end")
    for v, nv in pr.new_of_state.items():
        k = pr.state[v]
        for d in range(pr.depth.get(k, 1), 1, -1):      # shift the owner's history, oldest first
            L.append("    %s = %s;" % (hist[(k, d)], hist[(k, d - 1)] if d > 2 else "own_%d§" % v))
        L.append("    own_%d§ = own_%d§;" % (v, nv))
    body = L[i_body:]
    del L[i_body:]
    i_for = max(i for i, l in enumerate(L) if l.startswith("  for (i64 t = 0;"))
    if PUBLISH:     # nothing pending where the prologue meets the back edge (see WAIT_ACK)
        L[i_for:i_for] = ["  __builtin_amdgcn_s_waitcnt(0x0F70);"]
    if pre_n:
        L[i_for:i_for] = pre0_lines
    if ILV:
        # phase by phase over the blocks: while block b's hand-off is in flight the workgroup works
        # on block b + 1 — by the time it comes back to b the tags are there (what the round-2
        # review asked for; possible because batch blocks are independent recurrences)
        cuts = [0] + [i - i_body for i in i_phase] + [len(body)]
        for k in range(len(cuts) - 1):
            seg = body[cuts[k]:cuts[k + 1]]
            for blk_ in range(NBLK):
                sfx = "_b%d" % blk_
                L.append("    {   // ---- batch block %d" % blk_)
                L.extend(l.replace("§", sfx).replace("¤", str(blk_)) for l in seg)
                L.append("    }")
    for blk_ in range(0 if ILV else NBLK):
        sfx = "" if NBLK == 1 else "_b%d" % blk_
        if NBLK > 1:
            L.append("    {   // ---- batch block %d of this workgroup" % blk_)
        L.extend(l.replace("§", sfx).replace("¤", str(blk_)) for l in body)
        if NBLK > 1:
            L.append("    }")
    L.append("  }")
    if spec.trace:
        L.append("  if (tr_on) { tr[%d] = __builtin_amdgcn_s_memtime(); tr[%d] = __builtin_amdgcn_s_memrealtime(); }"
                 % (TRACE_NT * TRACE_MARKS + 2, TRACE_NT * TRACE_MARKS + 3))
        spec.marks = list(marks)
    L.append("  if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(a.ctl, base + (unsigned)a.T, %s);" % AG)
    L.append("}")
    return "\n".join(L) + "\n", (name,)
