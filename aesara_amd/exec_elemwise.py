"""Elemwise / CAReduce / row-program steps of a PlanExecutor: layout analysis on the host, kernel
selection (codegen specs), launches; horizontal groups (reference: tensor/elemwise.py:725/835
Elemwise.perform/_c_all, :1495/:1522 CAReduce).

Part of :class:`aesara_amd.executor.PlanExecutor` (a mixin: the methods run on the executor's
state; split out of executor.py in round 4, no behaviour change)."""
from __future__ import annotations

from .exec_common import *  # noqa: F401,F403
from .exec_common import (_I64, _VP, _i64arr, _Kernels, _FakeBuf, _CAST_SCALARS, _prod, _Arena, _os, _time)  # noqa: F401


class ElemwiseMixin:

    def _reduce_ws(self):
        if self._ws is None:
            if self.dry_run:
                self._ws = _FakeBuf(lib.ahip_reduce_ws_bytes(), "uint8")
            else:
                # zero-initialised once: the epoch-tagged partials, the epoch, the error word
                self._ws = torch.zeros(lib.ahip_reduce_ws_bytes(), dtype=torch.uint8,
                                       device=self.device)
                # a finalize that times out also raises a flag in pinned (device-visible) host
                # memory: the host looks at it after every call, no device round trip
                self._ws_flag = torch.zeros(2, dtype=torch.int32).pin_memory()
                self._ws_flag_np = self._ws_flag.numpy()
                off = lib.ahip_reduce_partials_bytes() + 2048 + cg.REDUCE_HOSTFLAG_OFF
                self._ws[off:off + 8].view(torch.int64).fill_(self._ws_flag.data_ptr())
        return self._ws

    def _check_reduce_flag(self):
        """Raise if a full-reduction launch reported (through the pinned flag) that a partial
        never arrived.  That launch's float result is NaN; an integer result is undefined."""
        f = self._ws_flag_np
        if f is not None and f[0]:
            f[0] = 0
            raise RuntimeError("full-reduction kernel: a workgroup's partial never arrived (the "
                               "device was not making progress on this grid); the result of that "
                               "evaluation (this call or a recent replayed one) is NaN / undefined")

    # ------------------------------------------------------------------ horizontal fusion ---
    def _find_hgroups(self):
        """Runs of consecutive, mutually independent full-reduction steps with ONE scalar program
        (structurally equal: towers of one model, per-parameter norms, several losses):
        {first step index: [step indices]}.  At run time the members whose layouts agree go out as
        ONE launch (``_run_hgroup``): the ~5 us a launch costs beyond its bytes (ramp, finalize
        hop, kernel boundary) is paid once per group, not once per member."""
        import json
        from . import knobs
        groups = {}
        if not knobs.get("HFUSE") or len(self.steps) < 2:
            return groups
        cur, cur_sig, written = [], None, set()

        def close():
            if len(cur) >= 2:
                groups[cur[0]] = list(cur)
        for si, st in enumerate(self.steps):
            sig = None
            ax = st.reduce.get("axis") if st.kind == "reduce" and st.reduce is not None else ()
            if ax is not None and ax != () and st.inputs and \
                    sorted(ax) == list(range(max(self.plan.vars[v].ndim for v in st.inputs))):
                ax = None                                  # every axis: a full reduction as well
            if st.kind == "reduce" and st.reduce is not None and ax is None \
                    and si not in self._owner and not st.post and not st.fallback:
                sig = json.dumps([st.scalar, st.out_refs, st.reduce["scalar_op"], st.reduce["acc_dtype"],
                                  st.reduce["ref"], self.plan.vars[st.reduce["out"]].dtype,
                                  [self.plan.vars[v].dtype for v in st.inputs]], sort_keys=True)
            if sig is not None and sig == cur_sig and not (set(st.inputs) & written):
                cur.append(si)
                written |= self._step_writes(st)
                continue
            close()
            cur, cur_sig, written = ([si], sig, set(self._step_writes(st))) if sig is not None \
                else ([], None, set())
        close()
        return groups

    def _run_hgroup(self, idxs, env, out_targets):
        """Run the steps of one horizontal group: every member goes through its normal host
        logic, but a flat full reduction hands its launch over (``self._defer``) instead of
        issuing it; jobs of one kernel specialisation then leave as ONE
        ``ahip_elemwise_reduce_all_multi`` launch (<= AHIP_HJOBS jobs each), the rest singly.
        Values the members read stay alive until the launches are out (drops are postponed)."""
        self._defer = []
        try:
            for si in idxs:
                self._run_step(si, env, out_targets)
            jobs, self._defer = self._defer, None
            buckets = {}
            for j in jobs:
                buckets.setdefault(j["sig"], []).append(j)
            for sig, js in buckets.items():
                while js:
                    chunk, js = js[:16], js[16:]
                    if len(chunk) == 1:
                        chunk[0]["single"]()
                    else:
                        self._launch_hjobs(chunk)
        finally:
            self._defer = None
        for si in idxs:
            for v in self._drop[si]:
                env.pop(v, None)

    def _launch_hjobs(self, jobs):
        j0 = jobs[0]
        spec = j0["mkspec"](True)
        (k,) = _Kernels.get(spec, load=not self.dry_run)
        nops = len(j0["ptrs"])
        ws = self._reduce_ws()
        ptrs = [p for j in jobs for p in j["ptrs"]]
        self._launch("ahip_elemwise_reduce_all_multi", (
            k, len(jobs), nops, (_VP * len(ptrs))(*ptrs), _i64arr([j["n"] for j in jobs]), j0["vec"],
            spec.block, (_VP * len(jobs))(*[j["result"].ptr for j in jobs]), _VP(ws.data_ptr()),
            ws.numel(), self._stream()))

    def _run_rowchain(self, st: Step, env, out_targets=None) -> bool:
        """Chain of last-axis reductions + Elemwise in one pass (ahip_rowchain).  Returns False
        when the run-time layout does not qualify; the caller then runs the original steps."""
        from ._lib import AHIP_RC_MAXLEAD, AHIP_RC_MAXOPS, RcArgs
        ex = st.extra
        D, members = ex["D"], ex["members"]
        arrs = [self.to_device(env[v]) for v in st.inputs]
        if any(a.ndim != D for a in arrs):
            return False
        shape = []
        for d in range(D):
            sizes = {a.shape[d] for a in arrs} - {1}
            if len(sizes) > 1:
                return False       # the unfused steps raise the reference's shape error
            shape.append(sizes.pop() if sizes else 1)
        K, lead = shape[-1], shape[:-1]
        N = _prod((lead))
        if K < 2 or N < 1:
            return False
        # operand classes: last dim K (unit stride) or broadcast; leading dims strided / broadcast
        ext, lstr = [], []
        for a in arrs:
            ls = [0 if a.shape[d] == 1 else a.strides[d] for d in range(D - 1)]
            bcast_lead = all(x == 0 for x in ls)
            if a.shape[-1] == K:
                if a.strides[-1] != 1:
                    return False
                ext.append((a.dtype, "c" if bcast_lead else "f"))
            else:
                ext.append((a.dtype, "s" if bcast_lead else "r"))
            lstr.append(ls)
        # members whose inputs are all per-row / scalar produce [..., 1]-shaped values
        rowlike = []
        for m in members:
            rl = all(r[0] == "r" or (r[0] == "e" and ext[r[1]][1] in "rs")
                     or (r[0] == "f" and rowlike[r[1]]) for r in m["ins"])
            if rl and m["reduce"]:
                return False
            rowlike.append(rl)
        # outputs: full [*shape], row-like [*lead, 1], reduce results [*lead] — all contiguous
        slots, spec_members, out_arrs = len(arrs), [], []
        cl = contiguous_strides(lead)
        for m, rl in zip(members, rowlike):
            sm = {"scalar": m["scalar"], "ins": m["ins"], "reduce": None, "stores": [],
                  "rowlike": rl}
            for ref, o in m["stores"]:
                ov = self.plan.vars[o]
                buf = self._out_buffer(o, lead + [1] if rl else shape, ov.dtype, out_targets)
                if not buf.is_contiguous():
                    return False
                env[o] = buf
                sm["stores"].append([ref, ov.dtype, slots])
                out_arrs.append(buf)
                lstr.append(list(cl) if rl else [x * K for x in cl])
                slots += 1
            if m["reduce"]:
                r = m["reduce"]
                ov = self.plan.vars[r["out"]]
                sm["reduce"] = {"op": r["op"], "acc": r["acc"], "ref": r["ref"], "out": ov.dtype,
                                "slot": None}
                if r["store"]:
                    buf = self._out_buffer(r["out"], lead, ov.dtype, out_targets)
                    if not buf.is_contiguous():
                        return False
                    env[r["out"]] = buf
                    sm["reduce"]["slot"] = slots
                    out_arrs.append(buf)
                    lstr.append(list(cl))
                    slots += 1
            spec_members.append(sm)
        if slots > AHIP_RC_MAXOPS:
            return False
        for rk, r in ex["keep"].items():
            b = env[r]
            env[rk] = b.view(list(b.shape) + [1], list(b.strides) + [1])
        lsh, lst = collapse_dims(list(lead), lstr)
        if len(lsh) > AHIP_RC_MAXLEAD:
            return False
        # vector width: 16-byte packs when K, every row start and every pointer allow it
        vec_ops = [(a, e, t) for a, e, t in zip(arrs, ext, lst) if e[1] in "fc"] + \
            [(b, (b.dtype, "f"), t) for b, t in zip(out_arrs, lst[len(arrs):])
             if b.shape[-1] == K and b.ndim == D]
        width = max(ITEMSIZE[a.dtype] for a, _, _ in vec_ops)
        V = max(1, 16 // width)
        while V > 1 and (K % V or any(a.ptr % (V * ITEMSIZE[a.dtype]) or any(x % V for x in t)
                                       for a, _, t in vec_ops)):
            V //= 2
        per = -(-K // V)
        L = 1
        while L < 64 and L < per:
            L *= 2
        nch = -(-per // L)
        # register budget: full operands + the intermediates that cross a reduction stay live
        n_live = sum(1 for e in ext if e[1] == "f") + \
            len({(r[1], r[2]) for m in members for r in m["ins"] if r[0] == "f"})
        long_rows = False
        if nch * V * max(1, width // 4) * n_live > 320 or nch > 16:
            # the row does not fit a wavefront's registers: one workgroup per row, every
            # reduction a sweep (re-reads hit the L2 / memory-side cache)
            if K < 1024:
                return False
            long_rows, L, nch = True, 64, 1
        # kernel lookup memoised on the step and the run-time layout class (building and hashing
        # the spec costs more host time than everything else in this function)
        memo = ex.setdefault("_kernels", {})
        blk = 256 if not long_rows or K < 16384 else (512 if K < 65536 else 1024)
        # streaming policy (BIG_STREAM): full operands of 96 MiB or more are read once, with
        # non-temporal loads (AESARA_HIP_RC_NT forces it on / off for the A/B)
        big = N * K * width >= self.BIG_STREAM and not long_rows
        if knobs.is_set("RC_NT"):
            big = bool(int(knobs.get("RC_NT"))) and not long_rows
        mkey = (tuple(ext), L, V, nch, len(lsh), self.dry_run, long_rows, blk, big)
        hit = memo.get(mkey)
        if hit is None:
            spec = cg.RowChainSpec(ext, spec_members, L, V, nch, lnd=len(lsh), block=blk, nt=big)
            key = ("long-" if long_rows else "") + spec.key()
            ent = _Kernels.cache.get(key) if not self.dry_run else \
                ([None] if key in _Kernels.compiled else None)
            if ent is None:
                src, names = (cg.generate_rowchain_long if long_rows else cg.generate_rowchain)(spec)
                if self.dry_run:
                    from .device import compile_cached
                    compile_cached(src)
                    _Kernels.compiled[key] = 1
                    ent = [None]
                else:
                    ent = load_kernels(src, names)
                    _Kernels.cache[key] = ent
            hit = memo[mkey] = (ent, spec.block)
        ent, block = hit
        g = RcArgs()
        g.N, g.K = N, K
        for d, x in enumerate(lsh):
            g.lshape[d] = x
        for k, (a, t) in enumerate(zip(arrs + out_arrs, lst)):
            g.ptr[k] = a.ptr
            for d, x in enumerate(t):
                g.ls[k][d] = x
        self._launch("ahip_rowchain", (ent[0], C.byref(g), block, 0 if long_rows else 64 // L,
                                       self._stream()))
        return True

    def _run_rowpass(self, st: Step, env) -> bool:
        """Single-pass GLM row program (ahip_rowpass).  Returns False when the run-time layout
        does not qualify (the caller then runs the original, unfused steps)."""
        from ._lib import RpArgs
        ex = st.extra
        dt = ex["dtype"]
        X, w = env[st.dots[0][0]], env[st.dots[0][1]]
        if not isinstance(X, DevArray) or not isinstance(w, DevArray):
            return False
        N, K = X.shape
        vecw = 4 if dt == "float32" else 2
        kv = K // (64 * vecw) if K % (64 * vecw) == 0 else 0
        ops = [self.to_device(env[v]) for v in st.inputs]
        ok = (X.dtype == dt and w.dtype == dt and 1 <= kv <= 8 and N >= 1024
              and X.strides[1] == 1 and X.strides[0] % vecw == 0 and X.ptr % 16 == 0
              and w.ndim == 1 and w.shape[0] == K and (w.strides[0] == 1) and w.ptr % 16 == 0
              and all(o.ndim == 1 and o.shape[0] in (1, N) for o in ops))
        if not ok:
            return False
        out_vars = [self.plan.vars[o] for o in st.outputs]
        outs = [self.alloc((N,), ov.dtype) for ov in out_vars]
        for o, arr in zip(st.outputs, outs):
            env[o] = arr
        rpw = {1: 16, 2: 8}.get(kv, 4 if kv <= 4 else 2)
        big = not (knobs.is_set("VECBYTES") or knobs.is_set("NT")) and N * K * ITEMSIZE[dt] >= self.BIG_STREAM
        spec = cg.RowPassSpec(dt, kv, st.scalar, [o.dtype for o in ops], [o.dtype for o in outs],
                              st.out_refs, ex["reds"], ex["col_ref"], rpw=rpw,
                              nt=big or bool(TUNE["nt"] & 1))
        grid = int(lib.ahip_rowpass_grid(N, spec.block, rpw))
        col_ws = self.alloc((grid, K), dt)
        nred = len(ex["reds"])
        red_ws = self.alloc((grid, max(nred, 1)), "float64")
        env[ex["col_ws"]] = col_ws
        for j, wsv in enumerate(ex["red_ws"]):
            env[wsv] = red_ws.view((grid,), (nred,), red_ws.offset + j)
        g = RpArgs()
        g.N, g.K, g.X, g.x_rs, g.w = N, K, X.ptr, X.strides[0], w.ptr
        allops = ops + outs
        for k, o in enumerate(allops):
            g.ptr[k] = o.ptr
            g.stride[k] = 0 if (o.shape[0] == 1 and N != 1) else o.strides[0]
        g.col_ws, g.red_ws = col_ws.ptr, red_ws.ptr
        g.nops, g.nred = len(allops), nred
        key = spec.key()
        ent = _Kernels.cache.get(key) if not self.dry_run else \
            ([None] if key in _Kernels.compiled else None)
        if ent is None:
            src, names = cg.generate_rowpass(spec)
            if self.dry_run:
                from .device import compile_cached
                compile_cached(src)
                _Kernels.compiled[key] = 1
                ent = [None]
            else:
                ent = load_kernels(src, names)
                _Kernels.cache[key] = ent
        waves = spec.block // 64
        shmem = waves * K * ITEMSIZE[dt] + waves * 8 * 8
        self._launch("ahip_rowpass", (ent[0], C.byref(g), spec.block, rpw, shmem, self._stream()))
        return True

    # ------------------------------------------------------------------ Elemwise ------
    def _operands(self, st: Step, env):
        ins = [env[i] for i in st.inputs]
        return ins

    def _broadcast(self, arrs):
        nd = max((a.ndim for a in arrs), default=0)
        for a in arrs:
            if a.ndim != nd:
                raise ValueError("Elemwise operands must have the same number of dimensions")
        shape = []
        for d in range(nd):
            sizes = [a.shape[d] for a in arrs]
            nz = set(sizes) - {1}
            if len(nz) > 1:
                raise ValueError(f"Shapes on dimension {d} do not match: {tuple(sizes)}")
            shape.append(nz.pop() if nz else 1)
        strides = [[0 if a.shape[d] == 1 else a.strides[d] for d in range(nd)] for a in arrs]
        return shape, strides

    # A flat stream with an operand of this many bytes or more is read ONCE per evaluation as far
    # as the caches are concerned (half of the 256 MiB memory-side cache: it cannot still be there
    # when it is read again): 16 bytes per lane and load, non-temporal loads and stores.  Measured
    # MALL-cold on the MI355X (profiles/r05_cfg2_cold_sweep8.txt / 9.txt, r05_stream_probe.txt): config 2
    # 26.5 -> 25.1 us per eval (0.63 -> 0.67 of the HBM peak), config 1b add 73.6 -> 61.8 us (0.68 ->
    # 0.81), the read-only ceiling 0.68 -> 0.75; either switch alone does nothing (16 bytes) or loses
    # (non-temporal 32-byte packs: 28.2 us).  Smaller operands keep 32 bytes per lane and the default
    # cache policy: eight 32 MiB towers re-read every evaluation sit in the memory-side cache and lose
    # with streaming loads (45 -> 51 us).  AESARA_HIP_VECBYTES / AESARA_HIP_NT override both.
    BIG_STREAM = 96 << 20

    def _big_stream(self, n, dtypes, classes):
        if knobs.is_set("VECBYTES") or knobs.is_set("NT"):
            return False
        return any(c == "c" and n * ITEMSIZE[dt] >= self.BIG_STREAM for dt, c in zip(dtypes, classes))

    def _pick_vec(self, shape, strides_list, ptrs, dtypes, classes, vecbytes=None):
        if any(c == "s" for c in classes):
            return 1
        csz = [ITEMSIZE[dt] for dt, c in zip(dtypes, classes) if c == "c"]
        if not csz:
            return 1
        target = max(1, min(8, (vecbytes or TUNE["vecbytes"]) // max(csz)))
        v = target
        while v > 1:
            ok = shape[-1] % v == 0
            if ok:
                for st, p, dt, c in zip(strides_list, ptrs, dtypes, classes):
                    if c != "c":
                        continue
                    al = min(16, v * ITEMSIZE[dt])
                    if p % al or any((s * ITEMSIZE[dt]) % al for s in st[:-1]):
                        ok = False
                        break
            if ok:
                return v
            v //= 2
        return 1

    @staticmethod
    def _tile_plan(cshape, cstrides, nin, dtypes):
        """Layout test for the LDS-tiled Elemwise kernel (codegen.generate_tiled): some input has
        its unit stride along a dim other than the last one (a transposing DimShuffle view) while
        every output is contiguous along the last dim.  Returns (tile_dim, tile, classes) or
        None."""
        nd = len(cshape)
        if nd < 2 or not TUNE.get("tiled", 1) or cshape[-1] < 16:
            return None
        if any(s[-1] != 1 for s in cstrides[nin:]):
            return None
        votes = {}
        for s in cstrides[:nin]:
            if s[-1] in (0, 1):
                continue
            for d in range(nd - 1):
                if s[d] == 1 and cshape[d] >= 16:
                    votes[d] = votes.get(d, 0) + 1
        if not votes:
            return None
        td = max(votes, key=lambda d: (votes[d], d))
        classes = []
        for k, s in enumerate(cstrides):
            if s[-1] == 1:
                classes.append("c")
            elif s[-1] == 0:
                classes.append("b")
            elif k < nin and s[td] == 1:
                classes.append("t")
            else:
                classes.append("s")
        tsz = [ITEMSIZE[dtypes[k]] for k in range(nin) if classes[k] == "t"]
        # registers one lane holds while the loads of a tile are in flight: tile^2 / 256 values
        # per non-scalar input
        dwords = sum((ITEMSIZE[dtypes[k]] + 3) // 4 for k in range(nin) if any(cstrides[k]))
        for tile in (64, 32):
            if tile == 64 and (min(cshape[td], cshape[-1]) < 48 or dwords * 16 > 128):
                continue
            if sum(tile * (tile + 1) * z for z in tsz) <= 48 * 1024:
                return td, tile, classes
        return None

    def _out_buffer(self, vid, shape, dtype, out_targets):
        """Output buffer for plan variable ``vid``: the caller-provided target view (Scan writes
        step results straight into its output buffers) when it fits, else a fresh allocation."""
        if out_targets:
            t = out_targets.get(vid)
            if t is not None and t.shape == tuple(shape) and t.dtype == dtype:
                return t
        return self.alloc(shape, dtype)

    def _gemv_epi_without_prologue(self, st, env, out_targets, fused_x):
        """Not the length-specialised row kernel after all: materialise the prologue vectors with
        their own Elemwise steps and run the chain again without prologue fusion."""
        for _d, (xp, _o, _c, _k) in fused_x.items():
            self._exec_step(xp["step"], env, None)
        self._xprog_off = True
        try:
            return self._run_gemv_epi(st, env, out_targets)
        finally:
            self._xprog_off = False

    def _run_gemv_epi(self, st: Step, env, out_targets=None):
        """Chain of row dots + elementwise epilogue in one kernel (ahip_gemv_epilogue)."""
        from ._lib import GvArgs
        # vector prologues (fusion._fuse_xprologue): fuse when every operand of the Elemwise is a
        # unit-stride 16-byte-aligned vector of the dot's length or a scalar and the kernel is
        # length-specialised; otherwise run the Elemwise step as the reference does
        xprog = st.extra.get("xprog", {})
        fused_x = {}
        if xprog:
            vecw_ = 4 if self.plan.vars[st.dots[0][0]].dtype == "float32" else 2
            n_dots = len(st.dots)
            for d, xp in xprog.items():
                A_ = self.to_device(env[st.dots[d][0]])
                K_ = A_.shape[1] if A_.ndim == 2 else -1
                ops_ = [self.to_device(env[u]) for u in xp["step"].inputs]
                cls_ = []
                for o in ops_:
                    if o.size == 1:
                        cls_.append("s")
                    elif (o.ndim == 1 and o.shape[0] == K_ and o.strides[0] == 1
                          and o.ptr % 16 == 0 and o.dtype == A_.dtype):
                        cls_.append("v")
                    else:
                        cls_ = None
                        break
                kv_ok = (K_ > 0 and K_ % (64 * vecw_) == 0 and A_.dtype in ("float32", "float64")
                         and (A_.strides[1] == 1 or K_ == 1))
                if cls_ is not None and "v" in cls_ and kv_ok and not self._xprog_off:
                    fused_x[d] = (xp, ops_, cls_, K_)
                else:
                    self._exec_step(xp["step"], env, None)
            del n_dots
        dots = []
        for d, (a, x) in enumerate(st.dots):
            A_ = self.to_device(env[a])
            if d in fused_x:
                xp, ops_, cls_, K_ = fused_x[d]
                if xp["store"]:
                    xbuf = self._out_buffer(x, (K_,), A_.dtype, out_targets)
                    if xbuf.strides[0] != 1 or xbuf.ptr % 16:
                        xbuf = self.alloc((K_,), A_.dtype)
                    env[x] = xbuf
                else:
                    xbuf = ops_[cls_.index("v")]
                dots.append((A_, xbuf))      # a unit-stride stand-in of the right length
            else:
                dots.append((A_, self.to_device(env[x])))
        others = [self.to_device(env[v]) for v in st.inputs]
        # 0-d operands (a symbolic alpha / beta of Gemv, blas.py:231) broadcast like length-1 ones
        others = [o if o.ndim else o.view((1,), (0,)) for o in others]
        out_vars = [self.plan.vars[o] for o in st.outputs]
        dt = dots[0][0].dtype
        M = dots[0][0].shape[0]
        for A, x in dots:
            if A.ndim != 2 or x.ndim != 1 or A.shape[1] != x.shape[0]:
                raise ValueError(f"Incompatible shapes for gemv: A {A.shape}, x {x.shape}")
            if A.shape[0] != M or A.dtype != dt or x.dtype != dt:
                raise ValueError("Incompatible operands in a fused gemv chain")
        row_ok = all((A.strides[1] == 1 or A.shape[1] == 1) and A.shape[1] > 0 for A, _ in dots)
        ew_ok = all(o.ndim == 1 and o.shape[0] in (1, M) for o in others)
        if fused_x and not (row_ok and ew_ok):
            return self._gemv_epi_without_prologue(st, env, out_targets, fused_x)
        if not (row_ok and ew_ok):
            # column-major (transposed view) or irregular operands: dots through K5 (ROW/COL
            # gemv picks the coalesced layout), then the ordinary fused Elemwise kernel
            darrs = [self._gemv(1.0, A, x, 0.0, None) for A, x in dots]
            arrs = darrs + others
            shape, in_strides = self._broadcast(arrs)
            outs = [self._out_buffer(o, shape, ov.dtype, out_targets)
                    for o, ov in zip(st.outputs, out_vars)]
            for o, arr in zip(st.outputs, outs):
                env[o] = arr
            if _prod((shape)):
                self._launch_elemwise(st.scalar, arrs, in_strides, outs, st.out_refs, shape)
            return
        outs = [self._out_buffer(o, (M,), ov.dtype, out_targets)
                for o, ov in zip(st.outputs, out_vars)]
        for o, arr in zip(st.outputs, outs):
            env[o] = arr
        if M == 0:
            return
        vecw = 4 if dt == "float32" else 2
        dot_vec = []
        g = GvArgs()
        g.M = M
        for d, (A, x) in enumerate(dots):
            K = A.shape[1]
            rs = A.strides[0] if M != 1 else 0
            incx = x.strides[0] if K != 1 else 1
            ok = (K % vecw == 0 and incx == 1 and rs % vecw == 0 and A.ptr % 16 == 0
                  and x.ptr % 16 == 0)
            dot_vec.append(bool(ok))
            g.A[d], g.a_rs[d], g.a_cs[d], g.K[d] = A.ptr, rs, 1, K
            g.x[d], g.incx[d] = x.ptr, incx
        ops = others + outs
        for k, o in enumerate(ops):
            g.ptr[k] = o.ptr
            g.stride[k] = 0 if o.shape[0] == 1 and M != 1 else o.strides[0]
        g.ndots, g.nops = len(dots), len(ops)
        kmax = max(A.shape[1] for A, _ in dots)
        rpw = 4 if (kmax * ITEMSIZE[dt] <= 4096 and M >= 4096 and len(dots) <= 2) else 1
        # rows of 64*VEC*kv elements (kv small): length-specialised kernel, all loads up front
        per = 64 * vecw
        kvs = [A.shape[1] // per for A, _ in dots]
        if not (all(dot_vec) and all(A.shape[1] % per == 0 and A.shape[1] > 0 for A, _ in dots)
                and sum(kvs) * rpw <= 24):
            kvs = None
        xprogs = None
        if fused_x:
            if kvs is None:
                return self._gemv_epi_without_prologue(st, env, out_targets, fused_x)
            xprogs = [None] * len(dots)
            for d, (xp, ops_, cls_, _k) in fused_x.items():
                P = xp["step"]
                xprogs[d] = {"scalar": P.scalar, "cls": cls_, "out_ref": P.out_refs[0],
                             "store": bool(xp["store"])}
                for q, o in enumerate(ops_):
                    g.xin[d][q] = o.ptr
                g.xout[d] = dots[d][1].ptr if xp["store"] else None
        big = not (knobs.is_set("VECBYTES") or knobs.is_set("NT")) and all(dot_vec) and any(
            A.shape[0] * A.shape[1] * ITEMSIZE[dt] >= self.BIG_STREAM for A, _ in dots)
        spec = cg.GemvEpiSpec(dt, dot_vec, st.scalar, [o.dtype for o in others],
                              [o.dtype for o in outs], st.out_refs, rpw=rpw, kvs=kvs,
                              xprogs=xprogs, nt=big or bool(TUNE["nt"] & 1))
        key = spec.key()
        ent = _Kernels.cache.get(key) if not self.dry_run else \
            ([None] if key in _Kernels.compiled else None)
        if ent is None:
            src, names = cg.generate_gemv_epilogue(spec)
            if self.dry_run:
                from .device import compile_cached
                compile_cached(src)
                _Kernels.compiled[key] = 1
                ent = [None]
            else:
                ent = load_kernels(src, names)
                _Kernels.cache[key] = ent
        g.M = M
        self._launch("ahip_gemv_epilogue", (ent[0], C.byref(g), spec.block * 1, self._stream()))

    def _run_elemwise(self, st: Step, env, out_targets=None):
        ins = self._operands(st, env)
        out_vars = [self.plan.vars[o] for o in st.outputs]
        if all(not isinstance(x, DevArray) for x in ins) and hostops.host_evaluable(st.scalar):
            # host glue (integer shape arithmetic)
            res = hostops.eval_scalar_host(st.scalar, [np.asarray(x) for x in ins])
            shape = np.broadcast_shapes(*[np.shape(x) for x in ins]) if ins else ()
            for o, ov, k in zip(st.outputs, out_vars, st.out_refs):
                env[o] = np.broadcast_to(res[k], shape).astype(ov.dtype)
            return
        arrs = [self.to_device(x) for x in ins]
        shape, in_strides = self._broadcast(arrs)
        outs = [self._out_buffer(o, shape, ov.dtype, out_targets)
                for o, ov in zip(st.outputs, out_vars)]
        for o, arr in zip(st.outputs, outs):
            env[o] = arr
        n = _prod((shape)) if shape else 1
        if n == 0:
            return
        self._launch_elemwise(st.scalar, arrs, in_strides, outs, st.out_refs, shape)

    def _launch_elemwise(self, scalar, arrs, in_strides, outs, out_refs, shape):
        strides = [list(s) for s in in_strides] + [list(o.strides) for o in outs]
        cshape, cstrides = collapse_dims(shape, strides)
        if len(cshape) > AHIP_MAXD:
            # more non-mergeable dims than a kernel takes (7-d+ arrays whose operands alternate
            # broadcast / full dims): one launch per index of the outermost dim, recursively —
            # the reference's loop nest has no such limit (elemwise_cgen.py:228 make_loop)
            d0 = next(d for d in range(len(shape)) if shape[d] != 1)
            for i in range(shape[d0]):
                def sub(a, st):
                    return a.view(tuple(a.shape[:d0]) + (1,) + tuple(a.shape[d0 + 1:]), a.strides,
                                  a.offset + i * st[d0])
                sub_in = [sub(a.view(shape, st_), st_) if tuple(a.shape) != tuple(shape) or
                          tuple(a.strides) != tuple(st_) else sub(a, st_) for a, st_ in zip(arrs, in_strides)]
                sub_out = [sub(o, o.strides) for o in outs]
                sshape = tuple(shape[:d0]) + (1,) + tuple(shape[d0 + 1:])
                self._launch_elemwise(scalar, sub_in, [a.strides for a in sub_in], sub_out, out_refs, sshape)
            return
        ops = arrs + outs
        dtypes = [a.dtype for a in ops]
        ptrs = [a.ptr for a in ops]
        tp = self._tile_plan(cshape, cstrides, len(arrs), dtypes)
        if tp is not None:
            td, tile, classes = tp
            spec = cg.KernelSpec(scalar, dtypes[:len(arrs)], dtypes[len(arrs):], out_refs, classes,
                                 len(cshape), 1, tile_dim=[td, tile],
                                 invariant=[all(x == 0 for x in st_) for st_ in cstrides[:len(arrs)]])
            (fn,) = _Kernels.get(spec, load=not self.dry_run)
            flat = [s for st in cstrides for s in st]
            self._launch("ahip_elemwise_tiled", (fn, len(cshape), _i64arr(cshape), len(ops),
                                                 (_VP * len(ops))(*ptrs), _i64arr(flat), td, tile,
                                                 None, None, 0, self._stream()))
            return
        classes = ["c" if s[-1] == 1 else ("b" if s[-1] == 0 else "s") for s in cstrides]
        if cshape[-1] == 1:
            classes = ["b"] * len(ops)
        n = _prod((cshape))
        big = len(cshape) == 1 and self._big_stream(n, dtypes, classes)
        vec = self._pick_vec(cshape, cstrides, ptrs, dtypes, classes, vecbytes=16 if big else None)
        flat = len(cshape) == 1 and vec > 1
        idx64 = n >= (1 << 31) - (1 << 24)
        invariant = tuple(not any(st_) for st_ in cstrides[:len(arrs)])
        nt = (3 if big else TUNE["nt"]) if flat else 0
        # launch memo: (program identity, layout class) -> kernel; skips spec construction and
        # key hashing on the eager path (the program object is kept alive so its id stays unique)
        mk = (id(scalar), tuple(out_refs), tuple(dtypes), tuple(classes), len(cshape), vec, idx64,
              flat, invariant, nt)
        hit = self._ew_memo.get(mk)
        if hit is None:
            spec = cg.KernelSpec(scalar, dtypes[:len(arrs)], dtypes[len(arrs):], out_refs, classes,
                                 len(cshape), vec, idx64=idx64,
                                 unroll=TUNE["unroll"] if flat else 1, nt=nt,
                                 invariant=list(invariant))
            (fn,) = _Kernels.get(spec, load=not self.dry_run)
            if len(self._ew_memo) > 4096:
                self._ew_memo.clear()
            hit = self._ew_memo[mk] = (fn, spec.block, scalar)
        fn, block = hit[0], hit[1]
        is_flat, flat = flat, [s for st in cstrides for s in st]
        if big and is_flat and not knobs.is_set("STREAM_BPC"):
            # read-once streams run fastest with FEW wavefronts in flight: 2 x 256 threads per CU
            # (config 1b 64.3-65.4 -> 62.0-62.4 us, profiles/r05_cfg1b_stream_sweep3.txt)
            self._launch("ahip_elemwise_wg", (fn, len(cshape), _i64arr(cshape), len(ops),
                                              (_VP * len(ops))(*ptrs), _i64arr(flat), vec, block, 2,
                                              self._stream()))
            return
        self._launch("ahip_elemwise", (fn, len(cshape), _i64arr(cshape), len(ops),
                                (_VP * len(ops))(*ptrs), _i64arr(flat), vec, block,
                                self._stream()))

    # ------------------------------------------------------------------ CAReduce ------
    def _run_reduce(self, st: Step, env, out_targets=None):
        red = st.reduce
        ins = self._operands(st, env)
        arrs = [self.to_device(x) for x in ins]
        shape, in_strides = self._broadcast(arrs)
        nd = len(shape)
        out_var = self.plan.vars[red["out"]]
        axis = list(range(nd)) if red["axis"] is None else list(red["axis"])
        kept = [d for d in range(nd) if d not in axis]
        if nd > AHIP_MAXD and axis and not st.outputs:
            # more non-mergeable dims than a kernel takes (a Sum over the 8-d DimShuffle view that
            # ``tile``'s gradient builds): operands re-laid out contiguously in [kept | reduced]
            # order first — each group then folds into one dim.  The reference's loop nest has no
            # such limit (elemwise_cgen.py:305 make_reordered_loop)
            ksh, _ = collapse_dims([shape[d] for d in kept] or [1], [[s[d] for d in kept] or [0] for s in in_strides])
            rsh, _ = collapse_dims([shape[d] for d in axis], [[s[d] for d in axis] for s in in_strides])
            if len(ksh) + len(rsh) > AHIP_MAXD or (not kept and len(collapse_dims(shape, in_strides)[0]) > AHIP_MAXD):
                perm = kept + axis
                relaid = []
                for a, st_ in zip(arrs, in_strides):
                    buf = self.alloc([shape[d] for d in perm], a.dtype)
                    view = buf.view(tuple(shape), tuple(buf.strides[perm.index(d)] for d in range(nd)))
                    self.copy_into(view, a.view(tuple(shape), tuple(st_)))
                    relaid.append(view)
                arrs, in_strides = relaid, [list(v.strides) for v in relaid]
        out_shape = [shape[d] for d in kept]
        mat_vars = [self.plan.vars[o] for o in st.outputs]
        mats = [self.alloc(shape, ov.dtype) for ov in mat_vars]
        for o, arr in zip(st.outputs, mats):
            env[o] = arr
        result = self._out_buffer(red["out"], out_shape, out_var.dtype, out_targets)
        env[red["out"]] = result
        nred = _prod(([shape[d] for d in axis])) if axis else 1
        nkept = _prod((out_shape)) if out_shape else 1
        if not axis:
            # CAReduce over no axes == cast through the accumulator dtype (elemwise.py:1495)
            if nkept:
                sc = {"n_in": st.scalar["n_in"], "out": list(st.scalar["out"]),
                      "nodes": list(st.scalar["nodes"])}
                src = st.scalar["out"][red["ref"]]
                sc["nodes"] = sc["nodes"] + [
                    {"op": "cast", "in": [src], "dtype": red["acc_dtype"]},
                    {"op": "cast", "in": [["t", len(sc["nodes"])]], "dtype": out_var.dtype}]
                sc["out"] = sc["out"] + [["t", len(sc["nodes"]) - 1]]
                self._launch_elemwise(sc, arrs, in_strides, mats + [result],
                                      list(st.out_refs) + [len(sc["out"]) - 1], shape)
            return
        if nred == 0 and red["scalar_op"] in ("maximum", "minimum") and nkept > 0:
            raise ValueError("zero-size array to reduction operation which has no identity")
        if nkept == 0:
            return
        rspec = {"op": red["scalar_op"], "acc": red["acc_dtype"], "out": out_var.dtype,
                 "ref": red["ref"]}
        ops = arrs + mats
        dtypes = [a.dtype for a in ops]
        ptrs = [a.ptr for a in ops]
        strides = [list(s) for s in in_strides] + [list(m.strides) for m in mats]
        if nkept == 1:
            # full reduction: K2 two-stage, one partial per workgroup
            cshape, cstrides = collapse_dims(shape, strides)
            if len(cshape) > AHIP_MAXD:
                raise NotImplementedError("more than %d non-mergeable dims" % AHIP_MAXD)
            tp = self._tile_plan(cshape, cstrides, len(arrs), dtypes)
            if tp is not None:
                td, tile, classes = tp
                spec = cg.KernelSpec(st.scalar, dtypes[:len(arrs)], dtypes[len(arrs):],
                                     st.out_refs, classes, len(cshape), 1, tile_dim=[td, tile],
                                     reduce=dict(rspec, kind="all"),
                                     invariant=[all(x == 0 for x in st_)
                                                for st_ in cstrides[:len(arrs)]])
                (k_main,) = _Kernels.get(spec, load=not self.dry_run)
                ws = self._reduce_ws()
                flat = [s for stt in cstrides for s in stt]
                self._launch("ahip_elemwise_tiled", (
                    k_main, len(cshape), _i64arr(cshape), len(ops), (_VP * len(ops))(*ptrs),
                    _i64arr(flat), td, tile, _VP(result.ptr), _VP(ws.data_ptr()), ws.numel(),
                    self._stream()))
                return
            classes = ["c" if s[-1] == 1 else ("b" if s[-1] == 0 else "s") for s in cstrides]
            if cshape[-1] == 1:
                classes = ["b"] * len(ops)
            n = _prod((cshape))
            big = len(cshape) == 1 and self._big_stream(n, dtypes, classes)
            vec = self._pick_vec(cshape, cstrides, ptrs, dtypes, classes, vecbytes=16 if big else None)
            flat1 = len(cshape) == 1 and vec > 1
            idx64 = n >= (1 << 31) - (1 << 24)
            invariant = tuple(not any(st_) for st_ in cstrides[:len(arrs)])
            nt = (3 if big else TUNE["nt"]) if flat1 else 0
            # short streams (a few sweeps of the grid per thread: BASELINE config 2 is 8) keep two
            # vectors in flight per lane: r03 MALL-cold sweep (tools/tune_cold*.sh) 128 MiB fp64
            # 33.6 -> 29.1 / 31.7 -> 29.3 us on two boxes; long streams (512 MiB: 96.9 vs 101.9 us)
            # are better off with one
            # (16-byte streaming loads, see BIG_STREAM: two in flight at every size — the same bytes
            # in flight per lane as one 32-byte pack)
            unroll_all = TUNE["unroll"] if knobs.is_set("UNROLL") else \
                (2 if big or n * max(ITEMSIZE[d] for d in dtypes[:len(arrs)]) <= (1 << 28) else 1)
            mk = ("all", id(st.scalar), tuple(st.out_refs), tuple(dtypes), tuple(classes),
                  len(cshape), vec, idx64, flat1, invariant, rspec["op"], rspec["acc"],
                  rspec["out"], rspec["ref"], unroll_all, nt)

            def mkspec(hjobs=False, st=st, dtypes=dtypes, classes=classes, nd_=len(cshape), vec=vec,
                       idx64=idx64, flat1=flat1, invariant=invariant, rspec=rspec, unroll_all=unroll_all,
                       nin=len(arrs), nt=nt):
                return cg.KernelSpec(st.scalar, dtypes[:nin], dtypes[nin:], st.out_refs,
                                     classes, nd_, vec, idx64=idx64,
                                     reduce=dict(rspec, kind="all"), block=TUNE["red_block"],
                                     unroll=unroll_all if flat1 else 1,
                                     nt=nt, invariant=list(invariant), hjobs=hjobs)

            def single(mk=mk, mkspec=mkspec, st=st, cshape=cshape, cstrides=cstrides, ops=ops,
                       ptrs=ptrs, vec=vec, result=result):
                hit = self._ew_memo.get(mk)
                if hit is None:
                    spec = mkspec()
                    (k_main,) = _Kernels.get(spec, load=not self.dry_run)
                    if len(self._ew_memo) > 4096:
                        self._ew_memo.clear()
                    hit = self._ew_memo[mk] = (k_main, spec.block, st.scalar)
                k_main, block = hit[0], hit[1]
                ws = self._reduce_ws()
                flat = [s for stt in cstrides for s in stt]
                self._launch("ahip_elemwise_reduce_all", (
                    k_main, len(cshape), _i64arr(cshape), len(ops), (_VP * len(ops))(*ptrs),
                    _i64arr(flat), vec, block, _VP(result.ptr), _VP(ws.data_ptr()),
                    ws.numel(), self._stream()))

            if self._defer is not None and flat1 and not idx64 and len(ops) <= 6 \
                    and all(c in ("c", "b") for c in classes):
                # member of a horizontal group: hand the launch over (``_run_hgroup``)
                self._defer.append({
                    # (the members of a group share their scalar program: the layout decides)
                    "sig": (tuple(dtypes), tuple(classes), vec, invariant, unroll_all, nt),
                    "mkspec": mkspec, "single": single, "ptrs": ptrs, "n": n, "vec": vec,
                    "result": result})
                return
            single()
            return
        self._launch_reduce_axis(st.scalar, st.out_refs, rspec, shape, kept, axis, ops,
                                 len(arrs), strides, result, nkept, nred)

    def _launch_reduce_axis(self, scalar, out_refs, rspec, shape, kept, axis, ops, nin, strides,
                            result, nkept, nred, slicing=True):
        dtypes = [a.dtype for a in ops]
        ptrs = [a.ptr for a in ops]
        # permute to [kept | reduced], collapse each group on its own
        ksh, kst = collapse_dims([shape[d] for d in kept], [[s[d] for d in kept] for s in strides])
        rsh, rst = collapse_dims([shape[d] for d in axis], [[s[d] for d in axis] for s in strides])
        nk, nr = len(ksh), len(rsh)
        if nk + nr > AHIP_MAXD:
            raise NotImplementedError("more than %d non-mergeable dims" % AHIP_MAXD)
        # layout choice from the widest input operand: which group owns its smallest stride
        main = max(range(nin), key=lambda k: sum(1 for s in kst[k] + rst[k] if s != 0))
        kmin = min([abs(s) for s in kst[main] if s != 0] or [1 << 62])
        rmin = min([abs(s) for s in rst[main] if s != 0] or [1 << 62])
        mode = 0 if (rmin < kmin and nred >= 2) else 1
        cshape = ksh + rsh
        cstrides = [a + b for a, b in zip(kst, rst)]
        flat = [s for stt in cstrides for s in stt]
        resident = 256 * 2048        # lanes the device keeps resident

        def pick_vec(vdim):
            """Vector width along collapsed dim `vdim` (every operand unit-stride or broadcast
            there, 16-byte vectors, aligned) and the per-operand classes."""
            cls = ["c" if st[vdim] == 1 else ("b" if st[vdim] == 0 else "s") for st in cstrides]
            csz = [ITEMSIZE[dt] for dt, c in zip(dtypes, cls) if c == "c"]
            if "s" in cls or not csz:
                return 1, ["s"] * len(ops)
            v = max(1, 16 // max(csz))
            while v > 1:
                ok = cshape[vdim] % v == 0
                for st, p, dt, c in zip(cstrides, ptrs, dtypes, cls):
                    if not ok:
                        break
                    if c == "c":
                        al = min(16, v * ITEMSIZE[dt])
                        ok = p % al == 0 and all((s * ITEMSIZE[dt]) % al == 0
                                                 for d_, s in enumerate(st) if d_ != vdim)
                if ok:
                    return v, cls
                v //= 2
            return 1, ["s"] * len(ops)

        nslices = 1
        if mode == 0:
            # row: `lanes` adjacent lanes per output, enough slices of long runs to fill the device
            vec, classes = pick_vec(nk + nr - 1)
            nredv = nred // vec
            lanes = 1
            while lanes < 64 and lanes * 2 <= nredv:
                lanes *= 2
            if slicing and nkept * lanes < resident and nredv >= lanes * 8:
                nslices = int(min(1024, -(-resident // (nkept * lanes)), nredv // (lanes * 4)))
        else:
            # col: TX x TY threads per workgroup, TX * vec adjacent outputs, TY rows in flight
            vec, classes = pick_vec(nk - 1)
            nkeptv = nkept // vec
            lanes = 1
            # a strip of `lanes` x vec adjacent outputs (2 KiB rows stream best: measured on
            # the COL gemv and here), not wider than the contiguous kept run
            run = max(64, ksh[-1] // vec) if nk > 1 else nkeptv
            while lanes < TUNE["col_lanes"] and lanes < min(nkeptv, run):
                lanes *= 2
            ty = 256 // lanes
            bx = -(-nkeptv // lanes)
            if slicing and bx * 256 < resident and nred >= ty * 8:
                nslices = int(min(1024, -(-2048 // bx), nred // (ty * 4)))
        nslices = max(nslices, 1)
        idx64 = max(nkept, nred) >= (1 << 31) - 1
        out_dt = rspec["out"] if nslices == 1 else rspec["acc"]
        # (an operand of 96 MiB or more: streaming loads, BIG_STREAM — rotating (MALL-cold) inputs: 3-15 %
        # faster on eight of nine layouts, a tie on the ninth, profiles/r05_axisred_cold_ab.txt; the SAME
        # 128-256 MiB input re-read every call is partly served by the memory-side cache and mixed:
        # up to 6 % faster on five layouts, up to 24 % slower on four, profiles/r05_axisred_nt_ab.txt)
        nt = bool(TUNE["nt"] & 1) or (vec > 1 and self._big_stream(nkept * nred, dtypes[:nin], classes[:nin]))
        # (row mode, every run at most one vector per lane: the kernel drops its inner loop)
        short = bool(mode == 0 and nslices == 1 and nred // vec <= lanes)
        mk = ("axis", id(scalar), tuple(out_refs), tuple(dtypes), tuple(classes), nk, nr, vec,
              idx64, mode, lanes, rspec["op"], rspec["acc"], out_dt, rspec["ref"], nt, short)
        hit = self._ew_memo.get(mk)
        if hit is None:
            spec = cg.KernelSpec(scalar, dtypes[:nin], dtypes[nin:], out_refs, classes,
                                 nk + nr, vec, idx64=idx64, unroll=TUNE["red_unroll"], nt=nt,
                                 reduce=dict(rspec, kind="row" if mode == 0 else "col", nk=nk,
                                             nr=nr, lanes=lanes, out=out_dt, **({"short": True} if short else {})))
            (fn,) = _Kernels.get(spec, load=not self.dry_run)
            if len(self._ew_memo) > 4096:
                self._ew_memo.clear()
            hit = self._ew_memo[mk] = (fn, spec.block, scalar)
        fn, block = hit[0], hit[1]
        if nslices == 1:
            target = result
        else:
            target = self.alloc((nslices, nkept), rspec["acc"])
        self._launch("ahip_elemwise_reduce_axis", (
            fn, mode, nk, nr, _i64arr(cshape), len(ops), (_VP * len(ops))(*ptrs), _i64arr(flat),
            nslices, _VP(target.ptr), block, vec, lanes, self._stream()))
        if nslices > 1:
            # second pass: fold the [nslices, nkept] partials (accumulator dtype) in slice order
            self._launch_reduce_axis(cg.IDENTITY_SCALAR, [], dict(rspec, ref=0), [nslices, nkept],
                                     [1], [0], [target], 1, [[nkept, 1]], result, nkept, nslices,
                                     slicing=False)

    # ------------------------------------------------------------------ copies --------
    def copy_into(self, dst: DevArray, src: DevArray, accumulate=False):
        """dst[...] (+)= broadcast(src) through K8."""
        if dst.size == 0:
            return
        nd = dst.ndim
        sshape = (1,) * (nd - src.ndim) + tuple(src.shape)
        sstr = (0,) * (nd - src.ndim) + tuple(src.strides)
        ss = []
        for d in range(nd):
            if sshape[d] == dst.shape[d]:
                ss.append(sstr[d])
            elif sshape[d] == 1:
                ss.append(0)
            else:
                raise ValueError(f"cannot broadcast shape {src.shape} into {dst.shape}")
        if src.dtype != dst.dtype:
            raise TypeError("copy_into needs equal dtypes")
        if nd > AHIP_MAXD and len(collapse_dims(list(dst.shape), [list(ss), list(dst.strides)])[0]) > AHIP_MAXD:
            # more non-mergeable dims than the copy kernel takes (``tile``: an 8-d DimShuffle view
            # made contiguous for a Reshape): one copy per index of the SMALLEST non-unit dim (the
            # fewest launches: a large leading extent would otherwise put tens of thousands of
            # copies into the launch list)
            d0 = min((d for d in range(nd) if dst.shape[d] != 1), key=lambda d: dst.shape[d])
            rest = tuple(dst.shape[:d0]) + (1,) + tuple(dst.shape[d0 + 1:])
            sv = src.view(tuple(dst.shape), tuple(ss))
            for i in range(dst.shape[d0]):
                self.copy_into(dst.view(rest, dst.strides, dst.offset + i * dst.strides[d0]),
                               sv.view(rest, sv.strides, sv.offset + i * ss[d0]), accumulate)
            return
        if not accumulate and nd >= 2 and dst.size >= 4096:
            # transposing copy: identity program through the LDS-tiled Elemwise kernel
            cshape, cstr = collapse_dims(list(dst.shape), [list(ss), list(dst.strides)])
            if self._tile_plan(cshape, cstr, 1, [src.dtype, dst.dtype]) is not None:
                self._launch_elemwise(cg.IDENTITY_SCALAR, [src], [ss], [dst], [0], list(dst.shape))
                return
        self._launch("ahip_copy_strided", (dtype_code(dst.dtype), max(nd, 1),
                                    _i64arr(dst.shape or (1,)), _VP(src.ptr),
                                    _i64arr(ss or (0,)), _VP(dst.ptr),
                                    _i64arr(dst.strides or (0,)), int(accumulate), self._stream()))

    def fill_zero(self, dst: DevArray):
        """Zero a contiguous view (K7)."""
        if dst.size == 0:
            return
        if not dst.is_contiguous():
            raise ValueError("fill_zero needs a contiguous view")
        zero = C.c_uint64(0)
        self._launch("ahip_fill", (dtype_code(dst.dtype), C.byref(zero), _VP(dst.ptr), dst.size,
                            self._stream()))

    def cast(self, a: DevArray, dtype) -> DevArray:
        """Elementwise dtype conversion through a generated kernel (NumPy assignment casting)."""
        if a.dtype == dtype:
            return a
        out = self.alloc(a.shape, dtype)
        if a.size:
            sc = _CAST_SCALARS.get(dtype)
            if sc is None:      # one persistent program per target dtype (spec-key memo)
                sc = _CAST_SCALARS[dtype] = {
                    "n_in": 1, "nodes": [{"op": "cast", "in": [["i", 0]], "dtype": dtype}],
                    "out": [["t", 0]]}
            self._launch_elemwise(sc, [a], [[0 if n == 1 else s for s, n in
                                             zip(a.strides, a.shape)]], [out], [0], list(a.shape))
        return out

    def materialize(self, a: DevArray) -> DevArray:
        out = self.alloc(a.shape, a.dtype)
        self.copy_into(out, a)
        return out

    def contiguous(self, a: DevArray) -> DevArray:
        return a if a.is_contiguous() else self.materialize(a)
