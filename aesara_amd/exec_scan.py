"""Scan of a PlanExecutor: the step loop as a launch list / hipGraph, and as ONE persistent kernel
(scan_persist.py, scan_persist_mat.py) where the class allows (reference: scan/op.py:1673
Scan.perform, scan/scan_perform.pyx:309-541).

Part of :class:`aesara_amd.executor.PlanExecutor` (a mixin: the methods run on the executor's
state; split out of executor.py in round 4, no behaviour change)."""
from __future__ import annotations

from .exec_common import *  # noqa: F401,F403
from .exec_common import (_I64, _VP, _i64arr, _Kernels, _FakeBuf, _CAST_SCALARS, _prod, _Arena, _os, _time)  # noqa: F401


class ScanMixin:

    # ------------------------------------------------------------------ Scan (K10) ----
    def _op_Scan(self, node, args):
        """reference: scan/op.py:1673 Scan.perform / scan_perform.pyx:71 — the step loop is
        driven from the host once and (with use_graph) captured into a single hipGraph."""
        p = node.params
        inner_plan = p["inner"]
        key = id(inner_plan)
        if self.fuse and p.get("sit_sot_in_slices") and not p.get("as_while"):
            # a weight gradient summed inside the loop (sit-sot Gemm / Ger accumulator) whose
            # buffer holds ONE row: the loop without it + one product over the stacked operands
            # behind it (fusion.push_out_product_accumulators, applied to this node alone; the
            # buffer length is known only now)
            alt = self._inner.get(("sunk", key))
            if alt is None:
                need, same, le2 = [], [], []
                wrap = Plan("scan_%d_sunk" % node.outputs[0], dict(self.plan.vars), list(node.inputs),
                            list(node.outputs), [node])
                if not hasattr(self, "_zero_vars"):
                    self._zero_vars, self._last_only = zero_filled_vars(self.plan), read_last_row_only(self.plan)
                wrap2 = push_out_product_accumulators(wrap, need, same, zeros=self._zero_vars,
                                                      last_only=self._last_only, need_le2=le2)
                alt = False
                if wrap2 is not wrap:
                    sub = PlanExecutor(wrap2, use_graph=False, dry_run=self.dry_run, fuse=self.fuse,
                                       device=None if self.dry_run else self.device.index)
                    alt = (sub, [node.inputs.index(v) for v in need],
                           [(node.inputs.index(a_), node.inputs.index(b_)) for a_, b_ in same],
                           [node.inputs.index(v) for v in le2])
                self._inner[("sunk", key)] = alt
            if alt and all(getattr(args[i], "shape", (0,))[0] == 1 for i in alt[1]) and \
                    all(getattr(args[i], "shape", (0,))[0] in (1, 2) for i in alt[3]) and \
                    all(self.host_int(args[i]) == self.host_int(args[j]) for i, j in alt[2]):
                sub = alt[0]
                sub._arena, sub._capturing = self._arena, self._capturing
                sub._root = self._root or self
                try:
                    res = sub.run(list(args))
                finally:
                    sub._arena = None
                    sub._capturing = False
                self.scan_modes.update(sub.scan_modes)
                self.scan_notes.update(sub.scan_notes)
                return res
        if self.fuse and TUNE["scan_persist"] and p["n_nit_sot"] and not (
                p.get("as_while") or p.get("n_shared_outs", 0) or p.get("mit_mot_in_slices")
                or p["mit_sot_in_slices"] or p["sit_sot_in_slices"]):
            # no recurrence at all (aesara.map, the row loops of jacobian / hessian / Rop): the step
            # restated over whole sequences, evaluated ONCE (fusion.batch_map_step)
            rows = self._inner.get(("rows", key))
            if rows is None:
                r = batch_map_step(inner_plan, p["n_seqs"])
                rows = False
                if r is not None:
                    rows = PlanExecutor(r["plan"], use_graph=False, dry_run=self.dry_run, fuse=self.fuse,
                                        device=None if self.dry_run else self.device.index)
                self._inner[("rows", key)] = rows
            if rows:
                res = self._scan_all_rows(node, p, args, rows)
                if res is not None:
                    return res
        ent = self._inner.get(key)
        if ent is None:
            # loop-invariant view/shape nodes of the inner graph (the W.T DimShuffles of every
            # Gemv in a GRU step) are hoisted and evaluated once per Scan call
            n_var = len(inner_plan.inputs) - p["n_non_seqs"]
            if self.fuse:
                # fused-gate steps (one product, sliced by columns per gate) -> one product per gate
                inner_plan = split_column_slices(inner_plan, set(inner_plan.inputs[n_var:]))
            pre_plan, loop_plan, hoisted = split_invariant(inner_plan, inner_plan.inputs[n_var:])
            # sequence-only work leaves the loop: Elemwise / Gemv chains on per-step vectors as
            # whole-sequence kernels and GEMMs (hoist_sequence_only), x_t @ W of a matrix state
            # as one GEMM over all steps (hoist_sequence_dots)
            seqdots, lifted = [], None
            kw = dict(use_graph=False, dry_run=self.dry_run, fuse=self.fuse,
                      device=None if self.dry_run else self.device.index)
            if self.fuse:
                seq_ids, inv_set = list(loop_plan.inputs[:p["n_seqs"]]), set(loop_plan.inputs[n_var:])
                loop_plan, lifted = hoist_sequence_only(loop_plan, seq_ids, inv_set)
                loop_plan, seqdots = hoist_sequence_dots(loop_plan, seq_ids, inv_set)
                if lifted is not None:
                    lp_ = merge_shared_left_dots(lifted["plan"]) if lifted.get("stacked") else lifted["plan"]
                    lifted = dict(lifted, plan0=lifted["plan"], plan=lp_, exec=PlanExecutor(lp_, **kw))
            ent = (PlanExecutor(loop_plan, **kw),
                   PlanExecutor(pre_plan, **kw) if pre_plan is not None else None, seqdots, lifted)
            self._inner[key] = ent
        inner, pre, seqdots, lifted = ent
        inner._pre = pre
        inner._seqdots = seqdots
        inner._lifted = lifted
        inner._n_pre = len(seqdots) + (len(lifted["outs"]) if lifted else 0)
        # share allocation/capture state with the parent
        inner._arena, inner._capturing = self._arena, self._capturing    # one arena per call tree
        inner._root = self._root or self                                 # error words, step slots
        for sub in (pre, lifted["exec"] if lifted else None):
            if sub is not None:
                sub._root = inner._root
        try:
            return self._scan_loop(node, p, args, inner)
        finally:
            inner._arena = None
            inner._capturing = False

    def _scan_all_rows(self, node, p, args, rows):
        """A Scan without recurrence as ONE evaluation of the step over whole sequences.  None:
        take the step loop (no steps, a sequence shorter than the trip count — the loop raises the
        reference's error —, an output buffer longer than the trip count)."""
        n_seqs, n_nit = p["n_seqs"], p["n_nit_sot"]
        n_steps = self.host_int(args[0])
        if n_steps < 1:
            return None
        seqs = [self.to_device(a) for a in args[1:1 + n_seqs]]
        keep = [self.host_int(a) for a in args[1 + n_seqs:1 + n_seqs + n_nit]]
        if any(sq.shape[0] < n_steps for sq in seqs) or any(L < 1 or L > n_steps for L in keep):
            return None
        inv = list(args[1 + n_seqs + n_nit:])

        def evaluate(t0, t1):
            largs = [sq.view((t1 - t0,) + tuple(sq.shape[1:]), sq.strides, sq.offset + t0 * sq.strides[0])
                     for sq in seqs] + inv
            rows._arena, rows._capturing = self._arena, self._capturing
            rows._root = self._root or self
            try:
                return largs, [self.to_device(r) for r in rows.run(largs)]
            finally:
                rows._arena = None
                rows._capturing = False

        # the whole-sequence intermediates are T times a step's (the unit vectors of a Jacobian loop
        # alone are [T, n]): when their sum would not fit comfortably, the rows go in blocks
        block = self._rows_block(rows, seqs, inv, n_steps)
        if block >= n_steps:
            largs, res = evaluate(0, n_steps)
            # every output is a buffer of its own (never a view of an operand or of another output)
            taken = [a.buf for a in largs if isinstance(a, DevArray)]
            for k, r in enumerate(res):
                if any(r.buf is b for b in taken):
                    res[k] = r = self.materialize(r)
                taken.append(r.buf)
            note = "no recurrence: the step evaluated once over whole sequences"
        else:
            res = None
            for t0 in range(0, n_steps, block):
                t1 = min(n_steps, t0 + block)
                _l, part = evaluate(t0, t1)
                if res is None:
                    res = [self.alloc((n_steps,) + tuple(r.shape[1:]), r.dtype) for r in part]
                for full, r in zip(res, part):
                    self.copy_into(full.view((t1 - t0,) + tuple(full.shape[1:]), full.strides,
                                             full.offset + t0 * full.strides[0]), r)
                del part
            note = "no recurrence: the step evaluated over blocks of %d rows" % block
        self.scan_modes.update(rows.scan_modes)
        self.scan_notes.update(rows.scan_notes)
        outs = []
        for r, L in zip(res, keep):
            if L < n_steps:          # scan_save_mem: only the last L rows are kept (scan/op.py:2105-2134)
                r = r.view((L,) + tuple(r.shape[1:]), r.strides, r.offset + (n_steps - L) * r.strides[0])
            outs.append(r)
        self.scan_modes[node.outputs[0]] = "all-rows"
        self.scan_notes[node.outputs[0]] = note
        return outs

    def _rows_block(self, rows, seqs, inv, n_steps):
        """Rows per evaluation of a whole-sequence step plan: all of them unless the bytes it would
        allocate (counted by a DRY run of the same plan on the same shapes: host only, exact shape
        propagation, memoised per shape signature) exceed ``ROWS_BYTES_CAP`` (default: a quarter of
        the free device memory)."""
        if self.dry_run:
            return n_steps
        cap = int(float(knobs.get("ROWS_BYTES_CAP_GB")) * (1 << 30))
        if cap <= 0:
            cap = torch.cuda.mem_get_info(self.device)[0] // 4

        def sig(a, lead):
            if isinstance(a, DevArray):
                return (tuple(a.shape[1:]) if lead else tuple(a.shape), a.dtype)
            return ("host", np.shape(a))
        key = tuple(sig(a, True) for a in seqs) + tuple(sig(a, False) for a in inv)
        memo = rows.__dict__.setdefault("_rows_bytes", {})
        est = memo.get(key)
        if est is None:
            est = memo[key] = self._rows_bytes_per_row(rows, seqs, inv)
        fixed, per_row = est
        if per_row is None or fixed + per_row * n_steps <= cap:
            return n_steps
        return int(max(1, min(n_steps, (cap - fixed) // max(per_row, 1))))

    def _rows_bytes_per_row(self, rows, seqs, inv):
        """(bytes allocated whatever T is, bytes per row) of the whole-sequence plan, from dry runs
        with 1 and 2 rows; (0, None) when a dry run cannot follow the plan (value-dependent shapes)."""
        twin = rows.__dict__.get("_twin")
        if twin is None:
            twin = rows._twin = PlanExecutor(rows.plan, use_graph=False, dry_run=True, fuse=self.fuse)
        count = [0]

        def alloc(shape, dtype, _base=type(twin).alloc):
            r = _base(twin, shape, dtype)
            count[0] += r.size * ITEMSIZE[r.dtype]
            return r
        twin.alloc = alloc

        def fake(a, t=None):
            if not isinstance(a, DevArray):
                return a
            shape = tuple(a.shape) if t is None else (t,) + tuple(a.shape[1:])
            return DevArray(_FakeBuf(max(_prod(shape), 1), a.dtype), 0, shape, contiguous_strides(shape), a.dtype)
        got = []
        try:
            for t in (1, 2):
                count[0] = 0
                twin.run([fake(a, t) for a in seqs] + [fake(a) for a in inv])
                got.append(count[0])
        except Exception:                        # noqa: BLE001
            return 0, None
        finally:
            del twin.alloc
        per_row = max(got[1] - got[0], 0)
        return max(got[0] - per_row, 0), per_row

    @staticmethod
    def _mitmot_inplace(inner, n_seqs, mm_in, mm_out):
        """Per mit-mot inner output: may the producing kernel write straight over the input tap
        of the same slice?  Yes when that tap variable is read by exactly one step, that step is
        a fused Elemwise-like kernel producing the output, and the tap is one of its pointwise
        operands (never the vector of a dot)."""
        plan, ok, ro = inner.plan, [], 0
        base = n_seqs
        for g, in_taps in enumerate(mm_in):
            for sl in mm_out[g]:
                good = False
                if sl in in_taps:
                    tap_var = plan.inputs[base + in_taps.index(sl)]
                    out_var = plan.outputs[ro]
                    def in_prologue(st_):
                        return any(tap_var in xp["step"].inputs
                                   for xp in st_.extra.get("xprog", {}).values())
                    readers = [st for st in inner.steps
                               if tap_var in st.inputs or any(tap_var in d for d in st.dots)
                               or any(tap_var in q.inputs for q in st.fallback + st.post)
                               or in_prologue(st)]
                    if len(readers) == 1:
                        st = readers[0]
                        good = (st.kind in ("elemwise", "gemv_epi", "gemm_epi")
                                and not in_prologue(st)
                                and out_var in st.outputs and tap_var in st.inputs
                                and not any(tap_var in d for d in st.dots)
                                and out_var not in plan.outputs[ro + 1:]
                                and plan.vars[tap_var].dtype == plan.vars[out_var].dtype)
                ok.append(good)
                ro += 1
            base += len(in_taps)
        return ok

    def _xfold_try(self, inner, p, lifted, seqs, n_steps, inv_vals):
        """Can the whole-sequence products of this Scan (``lifted``: every output a
        ``Dot22(x stacked over time, W_k)`` of ONE sequence) be computed inside the persistent
        matrix kernel instead of up front?  -> {"pairs", "x", "W"} or None.  Static part cached on
        the inner executor; the layout checks are per call."""
        if not int(knobs.get("SM_XFOLD")) or not lifted.get("stacked"):
            return None
        from . import scan_persist as sp
        from . import scan_persist_mat as sm
        st = getattr(inner, "_xfold_static", None)
        if st is None:
            st = False
            ent = getattr(inner, "_sp_prog", None)
            if ent is None:
                ent = inner._sp_prog = sp.analyze(inner, p, getattr(inner, "_n_pre", 0))
            prog = ent[0]
            plan0 = lifted.get("plan0")
            if prog is not None and prog.mode == "mat" and prog.dtype == "float32" and plan0 is not None \
                    and len(lifted["seq_in"]) == 1 and len(plan0.outputs) == len(lifted["outs"]) <= 4:
                xin, wins = plan0.inputs[0], list(plan0.inputs[1:])
                prod = {n.outputs[0]: n for n in plan0.nodes}
                wvars = []
                for o in plan0.outputs:
                    n = prod.get(o)
                    if n is None or n.op != "Dot22" or n.inputs[0] != xin or n.inputs[1] not in wins:
                        wvars = None
                        break
                    wvars.append(lifted["inv_in"][wins.index(n.inputs[1])])
                if wvars is not None and len(plan0.nodes) == len(plan0.outputs):
                    pairs = sm.xfold_pairs(prog, lifted["outs"])
                    if pairs:
                        st = (pairs, wvars)
            inner._xfold_static = st
        if not st:
            return None
        pairs, wvars = st
        lp = inner.plan
        x = seqs[lp.inputs.index(lifted["seq_in"][0])]
        if x.ndim != 3 or x.dtype != "float32" or x.shape[0] < n_steps or x.shape[1] % 16 or x.shape[2] % 64:
            return None
        Ws = []
        for wv in wvars:
            W = inv_vals.get(wv)
            if W is None or isinstance(W, np.ndarray):
                return None
            W = inner.to_device(W)
            if W.ndim != 2 or W.dtype != "float32" or W.shape != (x.shape[2], x.shape[2]):
                return None
            Ws.append(W)
        return {"pairs": pairs, "x": x, "W": Ws}

    def _scan_persist(self, node, p, inner, n_steps, seqs, outs, store, pos, non_seqs, pre_rows,
                      n_rec, n_nit, xfold=None):
        """Run the whole step loop as ONE persistent kernel (aesara_amd/scan_persist.py) when the
        inner graph belongs to the recurrent-vector class.  Returns None when it ran (the output
        buffers are filled: rows pos .. pos+n_steps-1 modulo store), else the reason it did not."""
        from . import scan_persist as sp
        ent = getattr(inner, "_sp_prog", None)
        if ent is None:
            ent = inner._sp_prog = sp.analyze(inner, p, getattr(inner, "_n_pre", 0))
        prog, why = ent
        if prog is None:
            return why
        if prog.mm_extra and len(outs) == n_rec + n_nit:
            # out-taps 2 .. m of a mit-mot group write the group's buffer too: further output slots
            # on the same buffer, each starting one row after the row its state is first read from
            n_outer = len(outs)
            outs.extend(outs[g_] for _sl, g_, _j in prog.mm_extra)
            try:
                return self._scan_persist(node, p, inner, n_steps, seqs, outs,
                                          list(store) + [store[g_] for _sl, g_, _j in prog.mm_extra],
                                          list(pos) + [pos[g_] + j_ for _sl, g_, j_ in prog.mm_extra],
                                          non_seqs, pre_rows, n_rec, n_nit, xfold=xfold)
            finally:
                del outs[n_outer:]
        if prog.mode == "mat":
            return self._scan_persist_mat(prog, p, inner, n_steps, seqs, outs, store, pos,
                                          non_seqs, pre_rows, n_rec, n_nit, xfold=xfold)
        if xfold is not None:
            return "sequence products fold into the matrix-state kernel only"
        lp = inner.plan
        n_seqs = p["n_seqs"]
        n_fixed = len(lp.inputs) - len(pre_rows)
        n_mm = len(prog.tap_seq)                      # mit-mot groups ([0, 1] -> [1]): two taps each
        inv_vars = lp.inputs[n_seqs + (prog.n_rec_inputs or n_rec + n_mm):n_fixed]
        if len(inv_vars) != len(non_seqs):
            return "invariant operand count"
        inv_val = dict(zip(inv_vars, non_seqs))
        used = {v for ph in prog.phases for v in ph["ins"]} | \
               {x for ph in prog.phases for _a, x in ph["dots"]}
        # a 2-row circular buffer holds the initial state in the row step 1 writes: safe only
        # while no workgroup can be in step 1 before every workgroup staged that row in step 0
        # — i.e. unless the new state is produced in an EARLIER phase than the first dot on its
        # previous value (then a fast workgroup's step-1 store races a slow one's step-0 read)
        for v, k in prog.state.items():
            if store[k] == 2:
                dots_at = [pi for pi, ph in enumerate(prog.phases) if any(x == v for _a, x in ph["dots"])]
                made_at = [pi for pi, ph in enumerate(prog.phases) if prog.new_of_state.get(v) in ph["outs"]]
                if dots_at and made_at and made_at[0] < dots_at[0]:
                    return "2-row circular buffer with the state produced before its first dot"
            if store[k] < prog.depth.get(k, 1):
                return "output buffer shorter than its deepest tap"
        seq_arr = {}
        for v, s in prog.seq.items():
            if v not in used:
                continue                              # a sequence the step never reads
            if v in prog.tap_seq:
                # rows pos + 1 .. pos + n_steps of the mit-mot buffer (what each step's tap 1 reads
                # before it overwrites that very row)
                g_, top = prog.tap_seq[v], prog.tap_top.get(v, 1)
                b = outs[g_]
                if pos[g_] + top + n_steps > store[g_]:
                    return "mit-mot buffer shorter than the loop"
                seq_arr[v] = b.view((n_steps,) + tuple(b.shape[1:]), b.strides,
                                    b.offset + (pos[g_] + top) * b.strides[0])
            else:
                seq_arr[v] = seqs[s] if s < n_seqs else pre_rows[s - n_seqs]
        f32 = prog.dtype                      # the one floating dtype of the loop (float32 / float64)
        isz = ITEMSIZE[f32]
        vecw = 16 // isz
        rec = outs[:n_rec]
        M = rec[0].shape[1] if rec[0].ndim == 2 else -1
        for k, b in enumerate(rec):
            if b.ndim != 2 or b.shape[1] != M or b.dtype != f32 or store[k] < 2 or \
                    (b.strides[1] != 1 and M != 1):
                return "recurrent output layout"
        if M < 1:
            return "empty state"
        Ks, lens, mats = {}, {}, {}
        for av in prog.mats:
            A = inner.to_device(inv_val[av])
            if A.ndim == 2 and A.strides[1] != 1 and A.shape[1] > 1:
                A = inner.contiguous(A)     # a transposed view: one row-major copy per call
            if A.ndim != 2 or A.dtype != f32 or A.shape[0] != M or A.shape[1] == 0 or \
                    (A.strides[1] != 1 and A.shape[1] > 1):
                return "matrix layout"
            if A.shape[1] % vecw or A.ptr % 16 or A.strides[0] % vecw:
                # rows are read as 16-byte vectors: a contraction length (or row pitch) that is
                # not a whole number of them gets a zero-padded row-major copy, once per call
                Kp = -(-A.shape[1] // vecw) * vecw
                Ap = inner.alloc((M, Kp), f32)
                if Kp != A.shape[1]:
                    inner.fill_zero(Ap)
                inner.copy_into(Ap.view((M, A.shape[1]), Ap.strides), A)
                A = Ap
            mats[av], Ks[av] = A, A.shape[1]
        nsq = {}
        for v in prog.nsq:
            x = inner.to_device(inv_val[v])
            if x.dtype != f32 or x.ndim > 1:
                return "invariant operand layout"
            x = x if x.ndim else x.view((1,), (0,))
            nsq[v], lens[v] = x, x.shape[0]
        for v, a_ in seq_arr.items():
            if a_.dtype != f32 or a_.ndim > 2 or a_.shape[0] < n_steps:
                return "sequence operand layout"
            lens[v] = a_.shape[1] if a_.ndim == 2 else 1
        for v in list(prog.state) + list(prog.older):
            lens[v] = M
        for ph in prog.phases:
            for o in ph["outs"]:
                lens[o] = M
            if ph.get("reduce"):
                lens[ph["reduce"]["u"]] = M          # the vector the reduction folds (exchanged)
        for v in prog.zero_d:
            lens[v] = 1
        for ph in prog.phases:
            for av, x in ph["dots"]:
                if lens.get(x) is None or -(-lens[x] // vecw) * vecw != Ks[av]:
                    return "dot vector length"
            for v in ph["ins"]:
                if lens.get(v) not in (1, M):
                    return "operand length"
        staged = {}
        for ph in prog.phases:
            for av, x in ph["dots"]:
                staged[x] = Ks[av]
            if ph.get("reduce"):
                staged[ph["reduce"]["u"]] = -(-M // vecw) * vecw
        stage = 2 * sum(staged.values()) + 64      # LDS floats for the staged dot vectors
        cus = 256 if self.dry_run else torch.cuda.get_device_properties(self.device).multi_processor_count
        geo = sp.choose_rows(M, Ks, stage, cus, isz)
        if geo is None:
            return "matrix rows of a workgroup do not fit on chip"
        R, nw, G, place = geo
        spec = sp.Spec(prog, lp, M, Ks, lens, R, nw, place, dtype=f32)
        key = spec.key()
        ent = _Kernels.cache.get(key) if not self.dry_run else \
            ([None] if key in _Kernels.compiled else None)
        if ent is None:
            src, names = sp.generate(spec)
            if self.dry_run:
                from .device import compile_cached
                compile_cached(src)
                _Kernels.compiled[key] = 1
                ent = [None]
            else:
                ent = load_kernels(src, names)
                _Kernels.cache[key] = ent
        # nit-sot buffers (the host loop allocates them after its first step)
        for k, v in prog.passthru.items():
            src = seqs[prog.seq[v]] if prog.seq[v] < n_seqs else pre_rows[prog.seq[v] - n_seqs]
            if store[k] != n_steps or src.ndim != 2 or src.shape[1] != M or src.dtype != f32:
                return "pass-through output layout"
        for j in range(n_nit):
            k = n_rec + j
            if k in prog.passthru:
                v = prog.passthru[k]
                s_ = prog.seq[v]
                # rows of a sequence-only value: the whole-sequence array IS the output (a caller's
                # own sequence is copied, a hoisted result is handed out as it is)
                src = seqs[s_] if s_ < n_seqs else pre_rows[s_ - n_seqs]
                src = src.view((n_steps, M), src.strides)
                outs[k] = inner.materialize(src) if s_ < n_seqs else inner.contiguous(src)
            elif any(j_ == k and o_ in prog.zero_d for o_, _kd, j_ in prog.outs):
                outs[k] = inner.alloc((store[k],), f32)       # a 0-d value per step
            else:
                outs[k] = inner.alloc((store[k], M), f32)
        g = sp.SpArgs()
        n_mmo = n_mm + len(prog.mm_extra)      # inner outputs of the mit-mot groups (one per out-tap)
        nit_shapes = [tuple(1 if d == 1 else M for d in lp.vars[lp.outputs[n_mmo + (n_rec - n_mm) + j]].shape)
                      for j in range(n_nit)]
        g.T = n_steps
        for av, slot in prog.mats.items():
            g.mat[slot], g.mat_rs[slot] = mats[av].ptr, mats[av].strides[0]
        for v, s in prog.seq.items():
            if v not in seq_arr:
                continue
            a_ = seq_arr[v]
            g.seq[s], g.seq_ts[s] = a_.ptr, a_.strides[0]
            g.seq_es[s] = a_.strides[1] if a_.ndim == 2 and a_.shape[1] != 1 else 0
        for v, s in prog.nsq.items():
            g.nsq[s], g.nsq_es[s] = nsq[v].ptr, (nsq[v].strides[0] if nsq[v].shape[0] != 1 else 0)
        for o, _kind, j in prog.outs:
            b = outs[j]
            # a mit-mot group writes its tap 1: one row after the one it reads as the running value
            g.out[j], g.out_rs[j], g.out_store[j], g.out_pos0[j] = \
                b.ptr, b.strides[0], store[j], pos[j] + (prog.mm_first.get(j, 1) if j < n_mm else 0)
        _off, total = sp.xch_layout(prog, lens, 2 if f32 == "float64" else 1)
        wkey = (id(inner), key)
        ws = self._sp_ws.get(wkey)
        if ws is None and not self.dry_run:
            trace = spec.trace
            nctl = 16 + (4 * (sp.TRACE_NT * sp.TRACE_MARKS + 4) if trace else 0)
            ws = (torch.zeros(max(total, 1), dtype=torch.int64, device=self.device),
                  torch.zeros(nctl, dtype=torch.int32, device=self.device))
            self._sp_ws[wkey] = ws
            if trace:
                sp.generate(spec)                 # (fills spec.marks when the kernel came from the cache)
                self.sp_trace = (ws[1], list(spec.marks))
        if not self.dry_run:
            g.xch, g.ctl = ws[0].data_ptr(), ws[1].data_ptr()
        why = self._launch_persistent(ent[0], G, 64 * nw, g)
        if why:
            return why
        self._sp_done = None
        if prog.cond is not None:
            # do-while: every workgroup left the loop after the same step; ONE host read per Scan
            # (the launch list: one per step) — the trip count sizes the outputs (scan/op.py:2139-2159)
            if self._capturing:
                raise HostReadInReplay("do-while Scan: the trip count is read on the host")
            self._sp_done = n_steps if self.dry_run else int(ws[1][4].item())
        for j, shp in enumerate(nit_shapes):
            if len(shp) != 1:      # the step hands the vector out as a row / column (DimShuffle 'x')
                b = outs[n_rec + j]
                outs[n_rec + j] = b.view((b.shape[0],) + shp,
                                         (b.strides[0],) + tuple(0 if d == 1 else 1 for d in shp))
        return None

    def _scan_persist_mat_chunks(self, rows, prog, p, inner, n_steps, seqs, outs, store, pos, non_seqs,
                                 pre_rows, n_rec, n_nit, xfold, inv_vars):
        """A batch too large for one 16 x 16 tile per workgroup, run as consecutive launches of the
        one-block-per-workgroup kernel over slices of ``rows`` batch rows: batch rows are
        independent recurrences, every operand with a batch axis is sliced (views: no copies), the
        weights are shared.  Each launch is the kernel the B = ``rows`` case runs (in-loop sequence
        products and the schedule around the hand-offs included: config 4 at B = 128 two launches of
        the B = 64 kernel, 0.66 of the MFMA peak, instead of one launch with two blocks per workgroup
        at 0.53)."""
        Bn = outs[0].shape[1]
        lo_hi = [(b0, min(b0 + rows, Bn)) for b0 in range(0, Bn, rows)]

        def cut(a, axis, b0, b1):
            shape = list(a.shape)
            shape[axis] = b1 - b0
            return a.view(tuple(shape), a.strides, a.offset + b0 * a.strides[axis])
        nsq_vars = set(prog.nsq)
        full_nit = {}
        full_front = {}
        for b0, b1 in lo_hi:
            seqs_c = [cut(a, 1, b0, b1) if a.ndim == 3 and a.shape[1] == Bn else a for a in seqs]
            pre_c = [None if a is None else (cut(a, 1, b0, b1) if a.ndim == 3 and a.shape[1] == Bn else a)
                     for a in pre_rows]
            outs_c = [None if a is None else cut(a, 1, b0, b1) for a in outs]
            non_c = []
            for v, a in zip(inv_vars, non_seqs):
                if v in nsq_vars and isinstance(a, DevArray) and a.ndim == 2 and a.shape[0] == Bn and Bn != 1:
                    a = cut(a, 0, b0, b1)
                non_c.append(a)
            xf_c = None
            if xfold is not None:
                def front(k, b0=b0, b1=b1):
                    if k not in full_front:
                        full_front[k] = xfold["up_front"](k)        # the whole batch once
                    return cut(full_front[k], 1, b0, b1)
                xf_c = dict(xfold, x=cut(xfold["x"], 1, b0, b1), up_front=front)
            why = self._scan_persist_mat(prog, p, inner, n_steps, seqs_c, outs_c, store, pos, non_c, pre_c,
                                         n_rec, n_nit, xfold=xf_c, _chunk=True)
            if why is not None and xf_c is not None:
                # (a ragged last slice: its sequence products up front after all)
                pre_c = [front(k) if a is None else a for k, a in enumerate(pre_c)]
                why = self._scan_persist_mat(prog, p, inner, n_steps, seqs_c, outs_c, store, pos, non_c, pre_c,
                                             n_rec, n_nit, xfold=None, _chunk=True)
            if why is not None:
                if b0 == 0:
                    return why
                # earlier slices already ran INTO ``outs``: the launch-list fallback restarts from
                # the initial-state rows, which a circular (memory-saved) buffer no longer holds
                # for the finished slices -> only a buffer that keeps every step may fall back
                if any(store[k] < n_steps + (pos[k] if k < n_rec else 0) for k in range(n_rec)):
                    raise RuntimeError("persistent Scan: batch slice %d could not be launched (%s) after "
                                       "earlier slices had overwritten the circular state buffers"
                                       % (b0 // rows, why))
                return "batch slice %d: %s" % (b0 // rows, why)
            for j in range(n_nit):               # nit-sot outputs of the slice -> their rows of the whole
                k = n_rec + j
                o = outs_c[k]
                if k not in full_nit:
                    full_nit[k] = inner.alloc((o.shape[0], Bn) + tuple(o.shape[2:]), o.dtype)
                inner.copy_into(cut(full_nit[k], 1, b0, b1), o)
        for k, a in full_nit.items():
            outs[k] = a
        return None

    def _scan_persist_mat(self, prog, p, inner, n_steps, seqs, outs, store, pos, non_seqs,
                          pre_rows, n_rec, n_nit, xfold=None, _chunk=False):
        """Matrix-state class (a batch of recurrences, small-M GEMM chains): one persistent
        kernel, weights in VGPRs in MFMA layout (aesara_amd/scan_persist_mat.py)."""
        from . import scan_persist_mat as sm
        lp = inner.plan
        n_seqs = p["n_seqs"]
        n_fixed = len(lp.inputs) - len(pre_rows)
        n_mm = len(prog.tap_seq)                      # mit-mot groups ([0, 1] -> [1]): two taps each
        inv_vars = lp.inputs[n_seqs + (prog.n_rec_inputs or n_rec + n_mm):n_fixed]
        if len(inv_vars) != len(non_seqs):
            return "invariant operand count"
        inv_val = dict(zip(inv_vars, non_seqs))
        f32 = prog.dtype                      # float32, or float64 (fragment form only)
        isz = ITEMSIZE[f32]
        rec = outs[:n_rec]
        if rec[0].ndim != 3:
            return "recurrent output layout"
        Bn, N = rec[0].shape[1], rec[0].shape[2]
        for k, b in enumerate(rec):
            if b.ndim != 3 or b.shape[1:] != (Bn, N) or b.dtype != f32 or store[k] < 2 or \
                    b.strides[2] != 1:
                return "recurrent output layout"
        if Bn < 1 or N < 1:
            return "empty state"
        Nt, N = N, -(-N // 64) * 64           # tiles / MFMA K loops cover whole multiples of 64
        cus = 256 if self.dry_run else torch.cuda.get_device_properties(self.device).multi_processor_count
        NB, NJ = -(-Bn // 16), N // 16
        nblk = 1
        mode_ = int(knobs.get("SM_BATCH_CHUNKS"))
        if NB * NJ > cus and not _chunk and cus // NJ >= 1 and not prog.passthru and \
                (mode_ == 2 or (mode_ == 1 and not prog.tap_seq)):
            # consecutive launches over slices of the batch, one block per workgroup each
            return self._scan_persist_mat_chunks((cus // NJ) * 16, prog, p, inner, n_steps, seqs, outs, store,
                                                 pos, non_seqs, pre_rows, n_rec, n_nit, xfold, inv_vars)
        if NB * NJ > cus:
            # 2 / 4 / 8 batch blocks per workgroup (independent recurrences sharing the weight registers:
            # the step body runs once per block) — fragment form, products up front only
            nblk = next((n for n in (2, 4, 8) if NB % n == 0 and (NB // n) * NJ <= cus), None)
            if nblk is None:
                return "more 16x16 tiles than CUs"
            if xfold is not None:
                return "sequence products: one batch block per workgroup only"
        Ks, mats = {}, {}
        for av in prog.mats:
            A = inner.to_device(inv_val[av])
            if A.ndim != 2 or A.dtype != f32 or tuple(A.shape) != (Nt, Nt):
                return "weight layout (square matrices of the state's width)"
            if N != Nt:
                # a state width that is not a multiple of 64: zero-padded copy (rows = K, columns)
                Ap = inner.alloc((N, N), f32)
                inner.fill_zero(Ap)
                inner.copy_into(Ap.view((Nt, Nt), Ap.strides), A)
                A = Ap
            elif A.strides[1] != 1:
                A = inner.contiguous(A)     # a transposed view (W.T of a gradient loop): one copy per call
            mats[av], Ks[av] = A, N
        if (isz // 4) * sum(K // 16 for K in Ks.values()) > 320:
            return "weight columns do not fit in registers"
        xf_spec, xf_w, folded = None, [], set()
        if xfold is not None:
            # sequence products inside the loop: x in fragment order (one strided copy per call),
            # their weight columns in LDS (as many as fit next to `part`) or registers
            ndm = max(len(ph["dots"]) for ph in prog.phases)
            x = xfold["x"]
            if N != Nt or Bn % 16 or tuple(x.shape[1:]) != (Bn, N) or len(prog.mats) + len(xfold["W"]) > sm.SM_MAXMAT:
                return "sequence products: layout"
            # as many of them as have room for their weight columns in LDS next to `part` (the
            # registers hold the recurrent weights and the operand fragments: a product with its
            # columns in registers spilled, r03) — spread over the windows of the step; the others
            # are computed up front after all
            n_lds = min(len(xfold["W"]), max(0, (158 * 1024 - 2 * ndm * 4096) // (N * 64)))
            # ... and, with the recurrent weights in the accumulation file (SpecMat.pin == 2), further
            # products with their columns in what is left of its 256 registers
            n_reg = 0
            if int(knobs.get("SM_XREG")) and knobs.get("SM_PIN") in (None, 2) and n_lds and \
                    str(knobs.get("SM_INIT")) == "publish":
                n_reg = min(len(xfold["W"]) - n_lds, max(0, (256 - sum(K // 16 for K in Ks.values())) // (N // 16)))
            wins = sm.xfold_windows(prog, xfold["pairs"])
            if not n_lds or wins is None:
                return "sequence products: no room for their weight columns in LDS"
            where = {gi: f for f, gis in wins["win"].items() for gi in gis}
            fold, load = [], {}
            while len(fold) < n_lds + n_reg:
                gi = min((g for g in range(len(xfold["pairs"])) if g not in fold),
                         key=lambda g: (load.get(where[g], 0), g))
                fold.append(gi)
                load[where[gi]] = load.get(where[gi], 0) + 1
            fold.sort()
            if sm.xfold_windows(prog, [xfold["pairs"][gi] for gi in fold]) is None:
                return "sequence products: one window would need two different x"
            items = []
            for gi, ((v, pi, d), W) in enumerate(zip(xfold["pairs"], xfold["W"])):
                if gi not in fold:
                    pre_rows[gi] = xfold["up_front"](gi)
                    continue
                if W.strides[1] != 1:
                    W = inner.contiguous(W)
                items.append((v, pi, d, len(prog.mats) + len(xf_w)))
                xf_w.append(W)
                folded.add(v)
            xf_spec = {"sx": prog.seq[items[0][0]], "items": items}
            if n_reg:           # the last ones: weight columns in registers
                xf_spec["reg"] = list(range(len(items) - n_reg, len(items)))
        spec = sm.SpecMat(prog, Bn, N, Ks, Nt, dtype=f32, xfold=xf_spec, nblk=nblk)
        if nblk > 1 and (spec.xmode != "frag" or spec.trace):
            return "more 16x16 tiles than CUs"
        if prog.older:
            # taps older than -1: values the tile element's owner made itself (registers, a shift
            # per step) — fragment form only; the rows they start from must be in the buffer
            if spec.xmode != "frag":
                return "taps other than [-1] on a matrix state (LDS exchange form)"
            if any(store[k] < D for k, D in prog.depth.items()):
                return "output buffer shorter than the deepest tap"
        if xf_spec is not None and sum(K // 16 for K in Ks.values()) + \
                (spec.nstaged + 1 + len(xf_spec.get("reg", ()))) * (N // 16) > (448 if xf_spec.get("reg") else 384):
            # recurrent weight columns + the operand fragments alive together + one more fragment
            # for x: beyond this the allocator spills (r03: 442 -> 512 registers + scratch reloads
            # in front of the MFMAs made the folded form slower than the products up front)
            return "sequence products: weights + fragments exceed the register budget"
        if spec.xmode == "none":
            return "float64 operand fragments do not fit in registers"
        _slots, stage_floats = sm.stage_slots(prog, Ks)
        ndmax = max(len(ph["dots"]) for ph in prog.phases)
        if spec.xmode != "frag" and stage_floats * 4 + ndmax * 4096 > 156 * 1024:
            return "operand blocks do not fit in LDS"
        if max(prog.seq.values(), default=-1) >= sm.SM_MAXSEQ or len(prog.nsq) > sm.SM_MAXNSQ:
            return "too many operands"

        def strides2(a_, rows, cols):
            """(row stride, column stride) of a pointwise operand broadcast against [B, N]"""
            if a_.ndim == 0:
                return 0, 0
            if a_.ndim == 1:
                return (0, a_.strides[0] if a_.shape[0] != 1 else 0) if a_.shape[0] in (1, cols) else None
            if a_.ndim == 2 and a_.shape[0] in (1, rows) and a_.shape[1] in (1, cols):
                return (a_.strides[0] if a_.shape[0] != 1 else 0, a_.strides[1] if a_.shape[1] != 1 else 0)
            return None

        used = {v for ph in prog.phases for v in ph["ins"]} | \
               {x for ph in prog.phases for _a, x in ph["dots"]}
        seq_arr = {}
        for v, s_ in prog.seq.items():
            if v not in used or v in folded:
                continue                              # a sequence the step never reads / folded in
            if v in prog.tap_seq:
                # rows pos + 1 .. pos + n_steps of the mit-mot buffer (tap 1: read by the element
                # owner before it overwrites that very element)
                g_, top = prog.tap_seq[v], prog.tap_top.get(v, 1)
                b = outs[g_]
                if pos[g_] + top + n_steps > store[g_]:
                    return "mit-mot buffer shorter than the loop"
                a_ = b.view((n_steps,) + tuple(b.shape[1:]), b.strides,
                            b.offset + (pos[g_] + top) * b.strides[0])
            else:
                a_ = seqs[s_] if s_ < n_seqs else pre_rows[s_ - n_seqs]
            if a_.dtype != f32 or a_.shape[0] < n_steps:
                return "sequence operand layout"
            st2 = strides2(a_.view(a_.shape[1:], a_.strides[1:]), Bn, Nt)
            if st2 is None:
                return "sequence operand layout"
            seq_arr[v] = (a_, st2)
        nsq = {}
        for v in prog.nsq:
            x = inner.to_device(inv_val[v])
            st2 = strides2(x, Bn, Nt) if x.dtype == f32 else None
            if st2 is None:
                return "invariant operand layout"
            nsq[v] = (x, st2)
        for ph in prog.phases:
            for v in ph["ins"]:
                if v in prog.state or v in prog.older or any(v in q["outs"] for q in prog.phases):
                    continue
                if v not in seq_arr and v not in nsq and v not in folded:
                    return "operand of unknown layout"
        key = spec.key()
        ent = _Kernels.cache.get(key) if not self.dry_run else \
            ([None] if key in _Kernels.compiled else None)
        if ent is None:
            src, names = sm.generate(spec)
            if self.dry_run:
                from .device import compile_cached
                compile_cached(src)
                _Kernels.compiled[key] = 1
                ent = [None]
            else:
                ent = load_kernels(src, names)
                _Kernels.cache[key] = ent
        for k, v in prog.passthru.items():
            src = seqs[prog.seq[v]] if prog.seq[v] < n_seqs else pre_rows[prog.seq[v] - n_seqs]
            if store[k] != n_steps or src.ndim != 3 or tuple(src.shape[1:]) != (Bn, Nt) or src.dtype != f32:
                return "pass-through output layout"
        for j in range(n_nit):
            k = n_rec + j
            if k in prog.passthru:
                # a sequence-only value handed out per step (as it is, or transposed): the
                # whole-sequence array is the output
                s_ = prog.seq[prog.passthru[k]]
                src = seqs[s_] if s_ < n_seqs else pre_rows[s_ - n_seqs]
                src = src.view((n_steps, Bn, Nt), src.strides)
                if k in prog.passthru_tj:
                    outs[k] = inner.materialize(src.view((n_steps, Nt, Bn),
                                                         (src.strides[0], src.strides[2], src.strides[1])))
                else:
                    outs[k] = inner.materialize(src) if s_ < n_seqs else inner.contiguous(src)
            else:
                outs[k] = inner.alloc((store[k], Bn, Nt), f32)
        g = sm.SmArgs()
        g.T = n_steps
        for av, slot in prog.mats.items():
            g.mat[slot], g.mat_rs[slot] = mats[av].ptr, mats[av].strides[0]
        for v, s_ in prog.seq.items():
            if v not in seq_arr:
                continue
            a_, (rs, cs) = seq_arr[v]
            g.seq[s_], g.seq_ts[s_], g.seq_rs[s_], g.seq_cs[s_] = a_.ptr, a_.strides[0], rs, cs
        for v, s_ in prog.nsq.items():
            x, (rs, cs) = nsq[v]
            g.nsq[s_], g.nsq_rs[s_], g.nsq_cs[s_] = x.ptr, rs, cs
        if xf_spec is not None:
            # x[t, bi*16 + r16, w*K/4 + grp*K/16 + 4q + e]  ->  xp[t, bi, w, q, grp, r16, e]
            x, K_ = xfold["x"], N
            Q_ = K_ // 64
            xs = x.view((n_steps,) + tuple(x.shape[1:]), x.strides)
            src = xs.view((n_steps, NB, 4, Q_, 4, 16, 4),
                          (xs.strides[0], 16 * xs.strides[1], (K_ // 4) * xs.strides[2], 4 * xs.strides[2],
                           (K_ // 16) * xs.strides[2], xs.strides[1], xs.strides[2]))
            xp = inner.alloc((n_steps, NB, 4, Q_, 4, 16, 4), f32)
            if xs.strides[0] == xs.shape[1] * xs.strides[1] or n_steps == 1 or NB == 1:
                inner.copy_into(xp, src)        # (steps and batch blocks collapse: six dimensions)
            else:
                # a slice of a larger batch: seven dimensions that do not collapse — one copy per
                # K quarter (the kernels take AHIP_MAXD = 6)
                for w_ in range(4):
                    inner.copy_into(
                        xp.view((n_steps, NB, Q_, 4, 16, 4), xp.strides[:2] + xp.strides[3:],
                                xp.offset + w_ * xp.strides[2]),
                        src.view((n_steps, NB, Q_, 4, 16, 4), src.strides[:2] + src.strides[3:],
                                 src.offset + w_ * src.strides[2]))
            sx = xf_spec["sx"]
            g.seq[sx], g.seq_ts[sx], g.seq_rs[sx], g.seq_cs[sx] = xp.ptr, NB * 16 * K_, 0, 0
            for (_v, _pi, _d, slot), W in zip(xf_spec["items"], xf_w):
                g.mat[slot], g.mat_rs[slot] = W.ptr, W.strides[0]
        for o, _kind, j in prog.outs:
            b = outs[j]
            g.out[j], g.out_ts[j], g.out_rs[j] = b.ptr, b.strides[0], b.strides[1]
            g.out_store[j], g.out_pos0[j] = store[j], pos[j] + (1 if j < n_mm else 0)
        _off, total = sm.xch_layout(prog, NB, N, spec.xmode, isz)
        wkey = (id(inner), key)
        ws = self._sp_ws.get(wkey)
        if ws is None and not self.dry_run:
            nctl = 16 + (4 * (sm.TRACE_NT * sm.TRACE_MARKS + 4) if spec.trace else 0)
            ws = (torch.zeros(max(total, 1), dtype=torch.int64, device=self.device),
                  torch.zeros(nctl, dtype=torch.int32, device=self.device))
            self._sp_ws[wkey] = ws
            if spec.trace:
                if not hasattr(spec, "marks"):
                    sm.generate(spec)
                self.sm_trace = (ws[1], spec.marks)
        if not self.dry_run:
            g.xch, g.ctl = ws[0].data_ptr(), ws[1].data_ptr()
        why = self._launch_persistent(ent[0], (NB // nblk) * NJ, 256, g)
        if why:
            return why
        return None

    def _scan_persist_ew(self, node, p, inner, n_steps, seqs, outs, store, pos, shared, non_seqs,
                         pre_rows, n_rec, n_nit):
        """Run a Scan whose step is purely element-wise as ONE kernel launch
        (aesara_amd/scan_persist_ew.py: thread e runs the whole recurrence of element e).  Returns
        (None, shared outputs, steps run) when it ran — the output buffers hold rows
        pos .. pos + steps - 1 modulo store — else (reason, None, 0)."""
        from . import scan_persist_ew as se
        ent = getattr(inner, "_se_prog", None)
        if ent is None:
            ent = inner._se_prog = se.analyze(inner, p, getattr(inner, "_n_pre", 0))
        prog, why = ent
        if prog is None:
            return why, None, 0
        lp = inner.plan
        n_seqs, n_sh = p["n_seqs"], p.get("n_shared_outs", 0)
        n_fixed = len(lp.inputs) - len(pre_rows)
        n_mm = len(prog.mm)
        n_rec -= n_mm                         # here: the mit-sot / sit-sot outputs (slots n_mm ..)
        inv_vars = lp.inputs[n_seqs + prog.n_rec_inputs + n_sh:n_fixed]
        if len(inv_vars) != len(non_seqs):
            return "invariant operand count", None, 0
        inv_val = dict(zip(inv_vars, non_seqs))
        wg = prog.wg
        rec_kind, nit_kind, sh_kind = se.slot_kinds(prog)

        def kind(v):
            return prog.cls.get(v, 0) if wg else 0
        # the ONE shape every per-element value has (``wg``: 0-d values ride along, ProgramEw.cls)
        S = None
        for k in range(n_mm + n_rec):
            if k < n_mm or not rec_kind[k - n_mm]:
                S = tuple(outs[k].shape[1:])
                break
        if S is None:
            for m in range(n_sh):
                if not sh_kind[m]:
                    S = tuple(inner.to_device(shared[m]).shape)
                    break
        if S is None:
            for v, s_ in sorted(prog.seq.items(), key=lambda kv: kv[1]):
                a_ = seqs[s_] if s_ < n_seqs else pre_rows[s_ - n_seqs]
                if a_ is not None and not kind(v):
                    S = tuple(a_.shape[1:])
                    break
        if S is None and wg:
            for v in prog.nsq:
                if not kind(v) and _prod(inner.to_device(inv_val[v]).shape) != 1:
                    S = tuple(inner.to_device(inv_val[v]).shape)
                    break
        if S is None:
            if not wg:
                return "no loop-varying operand to take the shape from", None, 0
            S = ()
        n = _prod(S) if S else 1
        if n < 1:
            return "empty state", None, 0
        if prog.as_while and n != 1 and not (wg and kind(prog.cond)):
            return "do-while over more than one element (the condition would be a reduction)", None, 0
        if wg and n > 1024:
            return "a reduction / 0-d value in the step and more than 1024 elements (one workgroup)", None, 0
        cs = contiguous_strides(S)

        def rows_ok(a):
            return tuple(a.shape[1:]) == S and (n == 1 or tuple(a.strides[1:]) == tuple(cs))

        def one(a):
            return _prod(a.shape[1:]) == 1
        g = se.EwScanArgs()
        g.T, g.n = n_steps, n
        for v, s_ in prog.seq.items():
            a_ = seqs[s_] if s_ < n_seqs else pre_rows[s_ - n_seqs]
            if a_ is None or a_.dtype != lp.vars[v].dtype or a_.shape[0] < n_steps or \
                    (not one(a_) if kind(v) else tuple(a_.shape[1:]) != S):
                return "sequence operand layout", None, 0
            if not kind(v) and not rows_ok(a_):
                a_ = inner.contiguous(a_)
            g.seq[s_], g.seq_ts[s_] = a_.ptr, a_.strides[0]
        bc = set()
        for v, j in prog.nsq.items():
            x = inner.to_device(inv_val[v])
            if x.dtype != lp.vars[v].dtype:
                return "invariant operand dtype", None, 0
            if _prod(x.shape) == 1:
                bc.add(j)
                g.nsq[j], g.nsq_es[j] = x.ptr, 0
            else:
                if tuple(x.shape) != S or kind(v):
                    return "invariant operand shape", None, 0
                x = inner.contiguous(x)
                g.nsq[j], g.nsq_es[j] = x.ptr, 1
        n_mmo = sum(len(to) for _ti, to in prog.mm)   # inner outputs of the mit-mot groups come first
        for g_, (ti, to) in enumerate(prog.mm):
            b = outs[g_]
            # (not circular) rows pos .. pos + n_steps - 1 + the deepest tap must exist; one dtype
            # for the buffer and everything the step reads from / writes to it
            if tuple(b.shape[1:]) != S or (n != 1 and tuple(b.strides[1:]) != tuple(cs)) or \
                    pos[g_] + n_steps + max(ti + to) > store[g_] or \
                    any(lp.vars[v].dtype != b.dtype for v, (gg, _t) in prog.mm_in.items() if gg == g_):
                return "mit-mot buffer layout", None, 0
        for k in range(n_rec):
            b = outs[n_mm + k]
            if (not one(b) if rec_kind[k] else
                    (tuple(b.shape[1:]) != S or (n != 1 and tuple(b.strides[1:]) != tuple(cs)))) or \
                    store[n_mm + k] < prog.depth[k] or b.dtype != lp.vars[lp.outputs[n_mmo + k]].dtype:
                return "recurrent output layout", None, 0
            if wg and not rec_kind[k] and kind(prog.rec_new[k]) and n != 1:
                pass        # (a 0-d value written into every element of a vector state: fine)
            if wg and rec_kind[k] and not kind(prog.rec_new[k]) and n != 1:
                return "per-element value stored into a 0-d output", None, 0
        out_dt = [outs[k].dtype for k in range(n_mm + n_rec)]
        nit_shape = []
        for j in range(n_nit):
            ov = lp.vars[lp.outputs[n_mmo + n_rec + j]]
            if nit_kind[j]:
                if not kind(prog.nit_new[j]) and n != 1:
                    return "per-element value stored into a 0-d output", None, 0
                nit_shape.append((1,) * ov.ndim)
            elif ov.ndim != len(S):
                return "nit-sot output of another rank", None, 0
            else:
                nit_shape.append(S)
            out_dt.append(ov.dtype)
        sh_vals, sh_dt = [], []
        for m in range(n_sh):
            x = inner.to_device(shared[m])
            if (_prod(x.shape) != 1) if sh_kind[m] else (tuple(x.shape) != S):
                return "shared value shape", None, 0
            sh_vals.append(inner.contiguous(x))
            sh_dt.append(x.dtype)
        spec = se.SpecEw(prog, lp, out_dt, sh_dt, bc)
        key = spec.key()
        ent = _Kernels.cache.get(key) if not self.dry_run else \
            ([None] if key in _Kernels.compiled else None)
        if ent is None:
            src, names = se.generate(spec)
            if self.dry_run:
                from .device import compile_cached
                compile_cached(src)
                _Kernels.compiled[key] = 1
                ent = [None]
            else:
                ent = load_kernels(src, names)
                _Kernels.cache[key] = ent
        for j in range(n_nit):
            sl = n_mm + n_rec + j
            outs[sl] = inner.alloc((store[sl],) + tuple(nit_shape[j]), out_dt[sl])
        for k in range(n_mm + n_rec + n_nit):
            b = outs[k]
            g.out[k], g.out_rs[k], g.out_store[k], g.out_pos0[k] = b.ptr, b.strides[0], store[k], pos[k]
        sh_out = []
        for m in range(n_sh):
            o = inner.alloc(tuple(sh_vals[m].shape) if sh_kind[m] else S, sh_dt[m])
            sh_out.append(o)
            g.sh_in[m], g.sh_out[m] = sh_vals[m].ptr, o.ptr
        ctl = None
        if prog.as_while and not self.dry_run:
            ctl = self._sp_ws.get(("se_ctl", id(inner)))
            if ctl is None:
                ctl = (None, torch.zeros(16, dtype=torch.int32, device=self.device))
                self._sp_ws[("se_ctl", id(inner))] = ctl
            g.ctl = ctl[1].data_ptr() + 16          # (word 1 of a persistent-Scan workspace is its error word)
        offs, nptr = ptr_offsets(type(g))
        # (``wg``: ONE workgroup of whole wavefronts covering the n elements, see scan_persist_ew)
        grid_, block_ = (1, max(64, (n + 63) // 64 * 64)) if wg else ((n + 255) // 256, 256)
        self._launch("ahip_launch_p", (ent[0], grid_, 1, 1, block_, 1, 1, 0, C.byref(g), C.sizeof(g),
                                       offs, nptr, 0, self._stream()))
        done = n_steps
        if prog.as_while:
            if self._capturing:
                raise HostReadInReplay("do-while Scan: the trip count is read on the host")
            if not self.dry_run:
                done = int(ctl[1][4].item())        # ONE host read per Scan (the launch list: one per step)
        return None, sh_out, done

    def _launch_persistent(self, fn, grid, block, g):
        """Launch a persistent Scan kernel whose workgroups WAIT FOR EACH OTHER: only when the
        whole grid can be co-resident (occupancy query of THIS kernel x CU count; cached per
        kernel) — else the caller takes the launch-list path.  ``AESARA_HIP_COOP=1`` makes it a
        cooperative launch (the runtime then refuses a grid that cannot be resident; +15-19 us per
        launch, MI355X guide "coop-launch").  The argument block goes with its pointer map, so a
        recorded launch is rebindable without guessing (ahip_list_bind_bases)."""
        if not self.dry_run:
            key = ("occ", fn.value if hasattr(fn, "value") else id(fn), block)
            cap = _Kernels.cache.get(key)
            if cap is None:
                per_cu, cus = C.c_int(), C.c_int()
                check(lib.ahip_occupancy(fn, block, 0, C.byref(per_cu), C.byref(cus)))
                cap = _Kernels.cache[key] = per_cu.value * cus.value
            if grid > cap:
                return "grid of %d workgroups cannot be co-resident (device holds %d)" % (grid, cap)
        offs, n = ptr_offsets(type(g))
        coop = 1 if int(knobs.get("COOP")) == 1 else 0
        self._launch("ahip_launch_p", (fn, grid, 1, 1, block, 1, 1, 0, C.byref(g), C.sizeof(g), offs, n,
                                       coop, self._stream()))
        return None

    def _scan_loop(self, node, p, args, inner):
        n_seqs = p["n_seqs"]
        mm_in = [list(t) for t in p.get("mit_mot_in_slices", [])]
        mm_out = [list(t) for t in p.get("mit_mot_out_slices", [])]
        n_mm = len(mm_in)
        n_mm_outs = sum(len(t) for t in mm_out)
        taps = mm_in + [list(t) for t in p["mit_sot_in_slices"]] + \
            [list(t) for t in p["sit_sot_in_slices"]]
        n_rec, n_nit = len(taps), p["n_nit_sot"]
        n_sh, as_while = p.get("n_shared_outs", 0), p.get("as_while", False)
        n_steps = self.host_int(args[0])
        if n_steps < 0:
            raise IndexError(f"Scan was asked to run for negative number of step {n_steps}")
        seqs = [inner.to_device(a) for a in args[1:1 + n_seqs]]
        for k, sq in enumerate(seqs):
            if sq.shape[0] < n_steps:
                raise ValueError(f"Sequence {k} has shape {sq.shape} but the Scan's required "
                                 f"number of steps is {n_steps}")
        o = 1 + n_seqs
        inits = [inner.to_device(a) for a in args[o:o + n_rec]]
        shared = list(args[o + n_rec:o + n_rec + n_sh])
        o += n_rec + n_sh
        nit_len = [self.host_int(a) for a in args[o:o + n_nit]]
        non_seqs = list(args[o + n_nit:])
        pre = getattr(inner, "_pre", None)
        if pre is not None:
            pre._arena, pre._capturing = inner._arena, inner._capturing
            try:
                hoisted = pre.run([inner.to_device(a) if not isinstance(a, np.ndarray) or a.ndim
                                   else a for a in non_seqs])
            finally:
                pre._arena = None
                pre._capturing = False
            # loop-invariant matrices are laid out row-major ONCE so that every step's
            # Gemv takes the coalesced one-wave-per-row path
            hoisted = [inner.contiguous(h) if isinstance(h, DevArray) and h.ndim == 2 else h
                       for h in hoisted]
            non_seqs = non_seqs + hoisted
        # sequence-only work, computed for all steps at once
        pre_rows = []
        lp = inner.plan
        n_fixed = len(lp.inputs) - getattr(inner, "_n_pre", 0)
        inv_vals = dict(zip(lp.inputs[n_fixed - len(non_seqs):n_fixed], non_seqs))
        lifted = getattr(inner, "_lifted", None)
        xfold = None
        if lifted is not None and n_steps >= 2 and TUNE["scan_persist"] and self.fuse:
            # sequence products x_t @ W that the persistent matrix kernel can compute itself, in
            # the shadow of its hand-off latency (scan_persist_mat.xfold_pairs): not run up front
            xfold = self._xfold_try(inner, p, lifted, seqs, n_steps, inv_vals)

        def run_lifted():
            lx = lifted["exec"]
            largs = []
            stacked = lifted.get("stacked", False)
            nb = None
            for v in lifted["seq_in"]:
                S = seqs[lp.inputs.index(v)]
                S = S.view((n_steps,) + tuple(S.shape[1:]), S.strides)
                if stacked:      # [T, B, n] read as [T * B, n]: the steps stacked along the rows
                    S = inner.contiguous(S)
                    nb = S.shape[1]
                    S = S.view((n_steps * nb, S.shape[2]), (S.shape[2], 1))
                largs.append(S)
            largs += [inv_vals[v] if isinstance(inv_vals[v], np.ndarray) and inv_vals[v].ndim == 0
                      else inner.to_device(inv_vals[v]) for v in lifted["inv_in"]]
            lx._arena, lx._capturing = inner._arena, inner._capturing
            try:
                rows = [lx.to_device(r) for r in lx.run(largs)]
                # (column views of one wide product keep their row pitch: merge_shared_left_dots)
                rows = [r if r.ndim == 2 and r.strides[1] == 1 and stacked else inner.contiguous(r)
                        for r in rows]
                if stacked:
                    # (sequences of one Scan may differ in their per-step row count — x_t [3, 4]
                    # next to y_t [5, 6]: every result unfolds with ITS rows per step)
                    rows = [r.view((n_steps, r.shape[0] // n_steps, r.shape[1]),
                                   ((r.shape[0] // n_steps) * r.strides[0], r.strides[0], 1))
                            for r in rows]
            finally:
                lx._arena = None
                lx._capturing = False
            return rows

        if xfold is not None:
            pre_rows = [None] * len(lifted["outs"])

            def up_front(k):
                """x stacked over time @ W_k -> rows [T, B, n] (a product the kernel leaves outside)"""
                X = xfold["x"]
                X = inner.contiguous(X.view((n_steps,) + tuple(X.shape[1:]), X.strides))
                nb, n_in = X.shape[1], X.shape[2]
                V = inner._gemm(1.0, X.view((n_steps * nb, n_in), (n_in, 1)), xfold["W"][k], 0.0, None)
                return V.view((n_steps, nb, V.shape[1]), (nb * V.strides[0], V.strides[0], 1))
            xfold["up_front"] = up_front
        elif lifted is not None and n_steps > 0:
            pre_rows = run_lifted()
        elif lifted is not None:
            pre_rows = [None] * len(lifted["outs"])
        if getattr(inner, "_seqdots", None) and n_steps > 0:
            hoisted_rows = {}
            for h in inner._seqdots:
                S = hoisted_rows.get(h["seq"])
                if S is None:
                    S = seqs[lp.inputs.index(h["seq"])]
                S = S.view((n_steps,) + tuple(S.shape[1:]), S.strides)
                M_ = inner.to_device(inv_vals[h["mat"]])
                if h["kind"] == "gemv":      # rows of M_ . x_t  ->  S[T,K] @ M_.T (+ beta * acc rows)
                    acc = h.get("acc")
                    Cin = None
                    if acc is not None:
                        Cin = hoisted_rows.get(acc)
                        if Cin is None:
                            Cin = seqs[lp.inputs.index(acc)]
                        Cin = Cin.view((n_steps,) + tuple(Cin.shape[1:]), Cin.strides)
                    V = inner._gemm(h["alpha"], S, M_.view([M_.shape[1], M_.shape[0]],
                                                           [M_.strides[1], M_.strides[0]]),
                                    h.get("beta", 0.0) if Cin is not None else 0.0, Cin)
                elif S.ndim == 2:            # x_t[K] @ M_[K,M]
                    V = inner._gemm(h["alpha"], S, M_, 0.0, None)
                else:                        # x_t[B,K] @ M_[K,M]
                    Sc = inner.contiguous(S)
                    T_, B_, K_ = Sc.shape
                    V = inner._gemm(h["alpha"], Sc.view([T_ * B_, K_], [K_, 1]), M_, 0.0, None)
                    V = V.view([T_, B_, V.shape[1]], [B_ * V.shape[1], V.shape[1], 1])
                pre_rows.append(V)
                hoisted_rows[h["out"]] = V
        mintaps = [min(t) for t in taps] + [0] * n_nit
        store = [b.shape[0] for b in inits] + nit_len
        outs, pos0 = [], {}
        for k, b in enumerate(inits):
            ob = inner.alloc(b.shape, b.dtype)
            ntaps, S = -mintaps[k], store[k]
            if k >= n_mm and not as_while and n_steps >= 1 and 0 < ntaps < S:
                # sit-sot / mit-sot with a known trip count: only the initial taps carry over
                # (every other row is written by a step, or zero-filled below when never reached),
                # so the (n_steps + 1)-row buffer is not copied.  A circular buffer (store <
                # n_steps + taps, scan_save_mem) starts at the row that makes the LAST step land
                # on row S - 1: the result is chronological without the rotation copies
                # (scan/op.py:2105-2134 rotates afterwards; same final layout)
                p0 = (-n_steps) % S if S < n_steps + ntaps else ntaps
                for j in range(ntaps):
                    dst = (p0 + j - ntaps) % S
                    inner.copy_into(ob.view((1,) + tuple(ob.shape[1:]), ob.strides,
                                            ob.offset + dst * ob.strides[0]),
                                    b.view((1,) + tuple(b.shape[1:]), b.strides,
                                           b.offset + j * b.strides[0]))
                pos0[k] = p0
            else:
                inner.copy_into(ob, b)
            outs.append(ob)
        outs += [None] * n_nit
        if n_steps == 0:
            # scan/op.py:1753-1762: empty nit-sot outputs, shared outputs left unset
            res = outs[:n_rec]
            for j in range(n_nit):
                ov = self.plan.vars[node.outputs[n_rec + j]]
                res.append(inner.alloc((0,) * len(ov.shape), ov.dtype))
            return res + [None] * n_sh
        pos = [pos0.get(k, (-mintaps[k]) % store[k]) for k in range(n_rec + n_nit)]

        def row(buf, i):
            return buf.view(buf.shape[1:], buf.strides[1:], buf.offset + i * buf.strides[0])

        def same_place(a, b):
            return (a.buf is b.buf and a.offset == b.offset and a.shape == b.shape and
                    all(sa == sb or n == 1 for sa, sb, n in zip(a.strides, b.strides, a.shape)))

        i, go = 0, True
        mm_inplace = None
        ew_ran = False
        if TUNE["scan_persist"] and self.fuse and n_steps >= 1:
            # a purely element-wise step: the whole recurrence as one launch, no exchange at all
            why_ew, sh_out, done = self._scan_persist_ew(node, p, inner, n_steps, seqs, outs, store, pos,
                                                         shared, non_seqs, pre_rows, n_rec, n_nit)
            if why_ew is None:
                ew_ran = True
                self.scan_modes[node.outputs[0]] = "persistent"
                self.scan_notes[node.outputs[0]] = "element-wise recurrence: one thread per element"
                i, go = done, False
                pos = [(pp + done) % st for pp, st in zip(pos, store)]
                shared = sh_out
        if ew_ran:
            pass
        elif TUNE["scan_persist"] and self.fuse and n_steps >= 2:
            self._sp_done = None
            why = self._scan_persist(node, p, inner, n_steps, seqs, outs, store, pos, non_seqs,
                                     pre_rows, n_rec, n_nit, xfold=xfold)
            if why is None and xfold is not None:
                self.scan_notes[node.outputs[0]] = "sequence products in the loop"
            if why is not None and xfold is not None:
                # the kernel declined with the sequence products folded in: compute them up front
                # after all and try the plain form (then the launch list)
                self.scan_notes[node.outputs[0]] = "sequence products up front: " + why
                pre_rows[:len(lifted["outs"])] = run_lifted()
                xfold = None
                why = self._scan_persist(node, p, inner, n_steps, seqs, outs, store, pos, non_seqs,
                                         pre_rows, n_rec, n_nit)
            self.scan_modes[node.outputs[0]] = "persistent" if why is None else "launch-list: " + why
            if why is None:
                done = getattr(self, "_sp_done", None)
                i = n_steps if done is None else done
                if done is not None:
                    go = False                # (a do-while that stopped on the device)
                pos = [(pp + i) % st for pp, st in zip(pos, store)]
        elif n_steps >= 1:
            self.scan_modes[node.outputs[0]] = "launch-list: disabled or fewer than 2 steps"
        while i < n_steps and go:
            step_args = [row(sq, i) for sq in seqs]
            for k in range(n_rec):
                for t in taps[k]:
                    step_args.append(row(outs[k], (pos[k] + t) % store[k]))
            step_args.extend(shared)
            step_args.extend(non_seqs)
            step_args.extend(row(V, i) for V in pre_rows)
            # mit-sot / sit-sot results are written straight into the output buffers when the
            # producing kernel can take the row view as its output (no aliasing: the row being
            # written is never one of the tap rows read in the same step).  mit-mot outputs
            # usually overwrite one of their own input taps -> copied after the step.
            targets = {}
            for k in range(n_mm, n_rec):
                if all((pos[k] + t) % store[k] != pos[k] for t in taps[k]):
                    targets[inner.plan.outputs[n_mm_outs + k - n_mm]] = row(outs[k], pos[k])
            # a mit-mot output that overwrites the tap it was computed from (the gradient
            # accumulators of a backward Scan) is written in place when the tap is read only
            # element-for-element by the one kernel that produces the output
            if n_mm:
                if mm_inplace is None:
                    mm_inplace = self._mitmot_inplace(inner, n_seqs, mm_in, mm_out)
                ro_ = 0
                for g in range(n_mm):
                    for sl in mm_out[g]:
                        if mm_inplace[ro_]:
                            targets[inner.plan.outputs[ro_]] = row(outs[g], sl + pos[g])
                        ro_ += 1
            # nit-sot results too, once their buffer exists (allocated after the first step)
            for j in range(n_nit):
                k = n_rec + j
                if outs[k] is not None:
                    targets[inner.plan.outputs[n_mm_outs + (n_rec - n_mm) + j]] = \
                        row(outs[k], pos[k])
            res = inner.run(step_args, out_targets=targets)
            if as_while:
                # scan/op.py:1947-1949: the loop goes on while the condition output is 0.
                # One host read per step: data-dependent trip count (never replayed).
                c = res[n_mm_outs + (n_rec - n_mm) + n_nit + n_sh]
                go = self.dry_run or not bool(np.asarray(self.host_array(c) if isinstance(c, DevArray)
                                         else c).reshape(()))
            writes, ro = [], 0
            for g in range(n_mm):
                for sl in mm_out[g]:
                    writes.append((row(outs[g], sl + pos[g]), inner.to_device(res[ro])))
                    ro += 1
            for k in range(n_mm, n_rec):
                writes.append((row(outs[k], pos[k]), inner.to_device(res[ro])))
                ro += 1
            # a result that is itself a view of an output buffer (pass-through of a tap) is
            # materialised before any row is overwritten: all reads of a step precede its writes
            bufs = [b.buf for b in outs if b is not None]
            writes = [(tgt, r if same_place(tgt, r) or not any(r.buf is b for b in bufs)
                       else inner.materialize(r)) for tgt, r in writes]
            for tgt, r in writes:
                if not same_place(tgt, r):
                    inner.copy_into(tgt, r)
            for j in range(n_nit):
                k = n_rec + j
                r = inner.to_device(res[ro])
                ro += 1
                if i == 0:
                    outs[k] = inner.alloc((store[k],) + r.shape, r.dtype)
                if not same_place(row(outs[k], pos[k]), r):
                    inner.copy_into(row(outs[k], pos[k]), r)
            shared = list(res[ro:ro + n_sh])
            pos = [(pp + 1) % st for pp, st in zip(pos, store)]
            i += 1
        # rotate circular buffers into chronological order (scan/op.py:2105-2134); zero and, for
        # a do-while that stopped early, truncate what was never written (:2139-2159)
        for k in range(n_mm, n_rec + n_nit):
            if store[k] < i - mintaps[k] and pos[k] < store[k]:
                if pos[k] == 0:
                    continue
                b = outs[k]
                rot = inner.alloc(b.shape, b.dtype)
                head = store[k] - pos[k]
                inner.copy_into(rot.view((head,) + b.shape[1:], rot.strides),
                                b.view((head,) + b.shape[1:], b.strides,
                                       b.offset + pos[k] * b.strides[0]))
                inner.copy_into(rot.view((pos[k],) + b.shape[1:], rot.strides,
                                         rot.offset + head * rot.strides[0]),
                                b.view((pos[k],) + b.shape[1:], b.strides))
                outs[k] = rot
            elif store[k] > i - mintaps[k]:
                b = outs[k]
                tail = store[k] - (i - mintaps[k])
                inner.fill_zero(b.view((tail,) + b.shape[1:], b.strides,
                                       b.offset + (i - mintaps[k]) * b.strides[0]))
                if i < n_steps:
                    outs[k] = b.view((store[k] - (n_steps - i),) + b.shape[1:], b.strides)
        return outs + shared
