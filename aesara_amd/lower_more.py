"""Lowerings of the remaining tensor Ops the reference's own test files ask for,
each DECOMPOSED into plan primitives that already have HIP kernels (Elemwise / CAReduce /
Subtensor / AdvancedSubtensor / CumOp / Sort / Nonzero / Join ...) plus two new primitives:
``Searchsorted`` (csrc/index.hip, one binary search per element) and ``HostCall`` (an Op whose
DEFINITION is a Python callable: ``Print``'s print function, ``as_op`` functions).

Registered by ``lower._register_handlers``; every handler cites the reference Op it restates.
"""
from __future__ import annotations

import numpy as np

from .plan import Node


class _B:
    """Small builder over a lowering context: plan variables in, plan variables out."""

    def __init__(self, ctx, static_shape, unsupported):
        self.ctx, self.plan = ctx, ctx.plan
        self._static_shape, self.Unsupported = static_shape, unsupported

    # -- metadata ------------------------------------------------------------------------
    def dtype(self, v):
        return str(self.plan.vars[v].dtype)

    def shape(self, v):
        return list(self.plan.vars[v].shape)

    def ndim(self, v):
        return len(self.plan.vars[v].shape)

    # -- primitives ----------------------------------------------------------------------
    def const(self, value, dtype):
        return self.plan.add_const(np.asarray(value, dtype=dtype), dtype)

    def raw(self, op, ins, dtype, shape, params=None):
        return self.ctx.raw(op, ins, dtype, shape, params)

    def multi(self, op, ins, outs, params=None):
        """A node with several outputs: ``outs`` = [(dtype, shape)]."""
        vids = [self.plan.new_var(dt, list(sh), None) for dt, sh in outs]
        self.plan.nodes.append(Node(op, list(ins), vids, params or {}))
        return vids

    def ew(self, ins, dtype, shape, nodes, out=None):
        """One Elemwise: nodes = [(op, dtype, [refs])]; the last node is the output by default."""
        sc = {"n_in": len(ins), "nodes": [{"op": o, "dtype": d, "in": r} for o, d, r in nodes],
              "out": [out if out is not None else ["t", len(nodes) - 1]]}
        # Elemwise operands share one rank (tensor/elemwise.py:304: the front end pads with
        # DimShuffle): lower-rank operands get leading broadcastable dims here
        R = max([self.ndim(v) for v in ins] + [len(shape)])
        ins = [v if self.ndim(v) == R else
               self.dimshuffle(v, ["x"] * (R - self.ndim(v)) + list(range(self.ndim(v)))) for v in ins]
        return self.raw("Elemwise", ins, dtype, shape, {"scalar": sc})

    def cast(self, v, dtype):
        if self.dtype(v) == dtype:
            return v
        return self.ew([v], dtype, self.shape(v), [("cast", dtype, [["i", 0]])])

    def shape_i(self, v, i):
        s = self.shape(v)[i]
        if s is not None and s != 1:
            return self.const(s, "int64")
        return self.raw("Shape_i", [v], "int64", [], {"i": int(i)})

    def iop(self, op, a, b):
        """0-d int64 arithmetic (host-evaluated shape glue)."""
        return self.ew([a, b], "int64", [], [(op, "int64", [["i", 0], ["i", 1]])])

    def make_vector(self, scalars, dtype="int64"):
        return self.raw("MakeVector", list(scalars), dtype, [len(scalars)], {"dtype": dtype})

    def reshape(self, v, dims, static=None):
        """``dims``: 0-d int64 PLAN VARIABLES (``lit(n)`` for a literal extent)."""
        shp = self.make_vector(dims)
        return self.raw("Reshape", [v, shp], self.dtype(v), static or [None] * len(dims),
                        {"ndim": len(dims)})

    def lit(self, n):
        return self.const(int(n), "int64")

    def flatten(self, v):
        if self.ndim(v) == 1:
            return v
        return self.reshape(v, [self.lit(-1)])

    def dimshuffle(self, v, order):
        src = self.shape(v)
        shp = [1 if o == "x" else src[o] for o in order]
        return self.raw("DimShuffle", [v], self.dtype(v), shp, {"new_order": list(order)})

    def subtensor(self, v, entries, extra=(), static=None):
        """entries: python slice / int / ("in",) placeholders consuming ``extra`` in order."""
        idx = []
        for e in entries:
            if isinstance(e, slice):
                idx.append({"slice": [e.start, e.stop, e.step]})
            else:
                idx.append({"index": e})
        nd_out = self.ndim(v) - sum(1 for e in entries if not isinstance(e, slice))
        return self.raw("Subtensor", [v] + list(extra), self.dtype(v),
                        static if static is not None else [None] * nd_out, {"idx_list": idx})

    def arange(self, start, stop, step, dtype="int64"):
        return self.raw("ARange", [start, stop, step], dtype, [None], {"dtype": dtype})

    def iota(self, n):
        return self.arange(self.const(0, "int64"), n, self.const(1, "int64"))

    def alloc(self, value, dims, static=None):
        return self.raw("Alloc", [value] + list(dims), self.dtype(value),
                        static or [None] * len(dims))

    def careduce(self, v, op, axes, dtype=None, acc=None):
        dtype = dtype or self.dtype(v)
        shp = [s for d, s in enumerate(self.shape(v)) if d not in axes]
        return self.raw("CAReduce", [v], dtype, shp,
                        {"scalar_op": op, "axis": sorted(axes), "acc_dtype": acc or dtype})

    def all_true(self, v):
        """0-d bool: every element of the bool array ``v``."""
        if self.ndim(v) == 0:
            return v
        return self.careduce(v, "and", list(range(self.ndim(v))), "bool", "bool")

    def check(self, x, cond, msg, exc="ValueError"):
        """``x`` as a view that raises ``exc(msg)`` unless the 0-d ``cond`` holds."""
        return self.raw("Assert", [x, cond], self.dtype(x), self.shape(x),
                        {"msg": msg, "exc_type": exc})

    def join(self, axis, parts, static=None):
        ax = self.const(axis, "int64")
        return self.raw("Join", [ax] + list(parts), self.dtype(parts[0]),
                        static or [None] * self.ndim(parts[0]))

    def take_rows(self, v, idx):
        shp = [None] + self.shape(v)[1:]
        return self.raw("AdvancedSubtensor1", [v, idx], self.dtype(v), shp)

    def gather_axis(self, v, idx, axis):
        """``v`` indexed by the int64 VECTOR ``idx`` along ``axis``."""
        if axis == 0:
            return self.take_rows(v, idx)
        entries = [{"slice": [None, None, None]}] * axis + [{"array": 0}]
        shp = self.shape(v)
        shp[axis] = None
        return self.raw("AdvancedSubtensor", [v, idx], self.dtype(v), shp, {"index": entries})

    def adv_get(self, v, idx_arrays, out_shape):
        """``v[idx_arrays...]`` (integer arrays only, one per leading dim of ``v``)."""
        return self.raw("AdvancedSubtensor", [v] + list(idx_arrays), self.dtype(v), out_shape)

    def adv_set(self, dst, val, idx_arrays, inc=False):
        return self.raw("AdvancedIncSubtensor", [dst, val] + list(idx_arrays), self.dtype(dst),
                        self.shape(dst), {"set_instead_of_inc": not inc, "inplace": False})

    def searchsorted(self, x, v, side, sorter=None):
        ins = [x, v] + ([sorter] if sorter is not None else [])
        return self.raw("Searchsorted", ins, "int64", self.shape(v), {"side": side})

    def grid(self, n, pos, nd):
        """``arange(n)`` placed at dim ``pos`` of an ``nd``-dim index array (1 elsewhere)."""
        ar = self.iota(n)
        if nd == 1:
            return ar
        return self.dimshuffle(ar, ["x"] * pos + [0] + ["x"] * (nd - pos - 1))


def register(hip_lower, static_shape, UnsupportedOp):                       # noqa: C901
    from aesara.graph.basic import Constant

    def B(ctx):
        return _B(ctx, static_shape, UnsupportedOp)

    def _const_value(v):
        return v.data if isinstance(v, Constant) else None

    # ---------------------------------------------------------------------------------------
    from aesara.tensor.extra_ops import (Bartlett, CpuContiguous, FillDiagonalOffset,
                                         RavelMultiIndex, Repeat, SearchsortedOp, Unique,
                                         UnravelIndex)

    @hip_lower.register(CpuContiguous)
    def _(op, node, ctx):
        # reference: tensor/extra_ops.py:40 CpuContiguous (perform :55: the input, or a C-contiguous
        # copy of it): a strided copy (K8) into a fresh contiguous buffer
        ctx.emit("DeepCopyOp", node)

    @hip_lower.register(SearchsortedOp)
    def _(op, node, ctx):
        # reference: tensor/extra_ops.py:102 SearchsortedOp (perform :144 np.searchsorted(x, v,
        # side, sorter)): operands are brought to their common dtype first, like NumPy
        b = B(ctx)
        x, v = ctx.vid(node.inputs[0]), ctx.vid(node.inputs[1])
        ct = np.promote_types(b.dtype(x), b.dtype(v)).name
        if ct not in ("float32", "float64", "int8", "int16", "int32", "int64", "uint8", "uint16",
                      "uint32", "uint64", "bool"):
            raise UnsupportedOp(f"SearchsortedOp over dtype {ct}")
        sorter = ctx.vid(node.inputs[2]) if len(node.inputs) == 3 else None
        if sorter is not None:
            sorter = b.cast(sorter, "int64")
        out = b.searchsorted(b.cast(x, ct), b.cast(v, ct), op.side, sorter)
        ctx.vmap[node.outputs[0]] = out

    @hip_lower.register(Repeat)
    def _(op, node, ctx):
        # reference: tensor/extra_ops.py:637 Repeat (perform :689 np.repeat(x, repeats, axis)).
        # scalar repeats: a broadcast copy  x[..., :, None, ...] -> (..., n, r, ...) -> (..., n*r, ...);
        # vector repeats: source row of output j = searchsorted(cumsum(repeats), j, "right")
        b = B(ctx)
        x, rep = node.inputs
        xv, rv = ctx.vid(x), b.cast(ctx.vid(rep), "int64")
        nd, axis = x.type.ndim, op.axis
        if axis is None:
            xv = b.flatten(xv)
            nd, axis = 1, 0
        else:
            axis = int(axis) % nd
        dims = [b.shape_i(xv, d) for d in range(nd)]
        odt = str(node.outputs[0].type.dtype)
        if rep.type.ndim == 0:
            ok = b.ew([rv], "bool", [], [("ge", "bool", [["i", 0], ["c", 0, "int64"]])])
            xv = b.check(xv, ok, "repeats may not contain negative values.")
            ds = b.dimshuffle(xv, list(range(axis + 1)) + ["x"] + list(range(axis + 1, nd)))
            bt = b.raw("BroadcastTo", [ds] + dims[:axis + 1] + [rv] + dims[axis + 1:], odt,
                       [None] * (nd + 1))
            nr = b.iop("mul", dims[axis], rv)
            out = b.reshape(bt, dims[:axis] + [nr] + dims[axis + 1:])
        elif rep.type.ndim == 1:
            rb = b.alloc(rv, [dims[axis]])                   # length 1 or n (NumPy broadcasts it)
            ok = b.all_true(b.ew([rb], "bool", [None], [("ge", "bool", [["i", 0], ["c", 0, "int64"]])]))
            rb = b.check(rb, ok, "repeats may not contain negative values.")
            cs = b.raw("CumOp", [rb], "int64", [None], {"axis": 0, "mode": "add"})
            total = b.careduce(rb, "add", [0], "int64", "int64")
            idx = b.searchsorted(cs, b.iota(total), "right")
            out = b.gather_axis(xv, idx, axis)
        else:
            raise UnsupportedOp("Repeat with repeats of more than one dimension")
        ctx.vmap[node.outputs[0]] = out

    @hip_lower.register(Unique)
    def _(op, node, ctx):
        # reference: tensor/extra_ops.py:1152 Unique (perform :1216 np.unique): sort -> first of
        # every run (NaNs form ONE run, NumPy's equal_nan) -> Nonzero -> gathers; return_index
        # through the stable ArgSort (NumPy sorts with mergesort then), return_inverse =
        # (running count of the flags - 1) scattered through the permutation, return_counts = the
        # distances between run starts
        b = B(ctx)
        x = node.inputs[0]
        dt = str(x.type.dtype)
        rows_as_items = op.axis is not None and x.type.ndim != 1
        ax = b.const(0, "int64")
        need_perm = op.return_index or op.return_inverse or rows_as_items
        if rows_as_items:
            # np.unique(x, axis=k): the slices along axis k are the items, ordered lexicographically
            # (NumPy sorts them as records): axis k to the front, items flattened to rows, a
            # LEXICOGRAPHIC stable argsort of the rows (LexArgSortRows: one stable column argsort
            # per column, last column first), then the vector recipe with "row differs from the
            # previous row" as the run-start flag
            nd = x.type.ndim
            k = int(op.axis) % nd
            order = [k] + [d for d in range(nd) if d != k]
            xt = b.dimshuffle(ctx.vid(x), order) if k else ctx.vid(x)
            other = [b.shape_i(ctx.vid(x), d) for d in range(nd) if d != k]
            n = b.shape_i(ctx.vid(x), k)
            rows = b.reshape(xt, [n, b.lit(-1)])
            perm = b.raw("LexArgSortRows", [rows], "int64", [None])
            xs = b.take_rows(rows, perm)
            hi2, lo2 = b.subtensor(xs, [slice(1, None, None)]), b.subtensor(xs, [slice(None, -1, None)])
            ne2 = b.ew([hi2, lo2], "bool", [None, None], [("neq", "bool", [["i", 0], ["i", 1]])])
            ne = b.careduce(ne2, "or", [1], "bool", "bool")
            xf = None
        else:
            xf = b.flatten(ctx.vid(x))
            n = b.shape_i(xf, 0)
            if need_perm:
                perm = b.raw("ArgSort", [xf, ax], "int64", [None], {"kind": "stable"})
                xs = b.take_rows(xf, perm)
            else:
                xs = b.raw("Sort", [xf, ax], dt, [None], {"kind": "quicksort"})
            hi, lo = b.subtensor(xs, [slice(1, None, None)]), b.subtensor(xs, [slice(None, -1, None)])
            if dt.startswith("float"):
                ne = b.ew([hi, lo], "bool", [None], [
                    ("neq", "bool", [["i", 0], ["i", 1]]), ("isnan", "bool", [["i", 0]]),
                    ("isnan", "bool", [["i", 1]]), ("and", "bool", [["t", 1], ["t", 2]]),
                    ("invert", "bool", [["t", 3]]), ("and", "bool", [["t", 0], ["t", 4]])])
            else:
                ne = b.ew([hi, lo], "bool", [None], [("neq", "bool", [["i", 0], ["i", 1]])])
        first = b.plan.add_const(np.ones((1,), "bool"), "bool")
        flag = b.raw("Subtensor", [b.join(0, [first, ne]), n], "bool", [None],
                     {"idx_list": [{"slice": [None, "in", None]}]})          # (n == 0: no run)
        pos = b.multi("Nonzero", [flag], [("int64", [None])])[0]
        uniq = b.take_rows(xs, pos)
        if rows_as_items:
            nu = b.shape_i(uniq, 0)
            u = b.reshape(uniq, [nu] + other)
            if k:
                back = list(range(1, k + 1)) + [0] + list(range(k + 1, nd))
                u = b.dimshuffle(u, back)
            uniq = u
        outs = [uniq]
        if op.return_index:
            outs.append(b.take_rows(perm, pos))
        if op.return_inverse:
            cnt = b.raw("CumOp", [b.cast(flag, "int64")], "int64", [None], {"axis": 0, "mode": "add"})
            imask = b.ew([cnt], "int64", [None], [("sub", "int64", [["i", 0], ["c", 1, "int64"]])])
            empty = b.raw("AllocEmpty", [n], "int64", [None], {"dtype": "int64"})
            outs.append(b.raw("AdvancedIncSubtensor1", [empty, imask, perm], "int64", [None],
                              {"set_instead_of_inc": True, "inplace": False}))
        if op.return_counts:
            nxt = b.join(0, [b.subtensor(pos, [slice(1, None, None)]), b.make_vector([n])])
            outs.append(b.ew([nxt, pos], "int64", [None], [("sub", "int64", [["i", 0], ["i", 1]])]))
        for o, v in zip(node.outputs, outs):
            ctx.vmap[o] = v

    @hip_lower.register(Bartlett)
    def _(op, node, ctx):
        # reference: tensor/extra_ops.py:822 Bartlett (perform :836 np.bartlett(M)):
        # n = arange(1 - M, M, 2); w = where(n <= 0, 1 + n / (M - 1), 1 - n / (M - 1)); M == 1 -> [1.]
        b = B(ctx)
        Mf = b.cast(ctx.vid(node.inputs[0]), "float64")
        start = b.ew([Mf], "float64", [], [("sub", "float64", [["c", 1.0, "float64"], ["i", 0]])])
        n = b.arange(start, Mf, b.const(2.0, "float64"), "float64")
        out = b.ew([n, Mf], "float64", [None], [
            ("sub", "float64", [["i", 1], ["c", 1.0, "float64"]]),                # 0: M - 1
            ("true_div", "float64", [["i", 0], ["t", 0]]),                        # 1: n / (M - 1)
            ("add", "float64", [["c", 1.0, "float64"], ["t", 1]]),                # 2
            ("sub", "float64", [["c", 1.0, "float64"], ["t", 1]]),                # 3
            ("le", "bool", [["i", 0], ["c", 0.0, "float64"]]),                    # 4
            ("switch", "float64", [["t", 4], ["t", 2], ["t", 3]]),                # 5
            ("eq", "bool", [["i", 1], ["c", 1.0, "float64"]]),                    # 6
            ("switch", "float64", [["t", 6], ["c", 1.0, "float64"], ["t", 5]])])  # 7
        ctx.vmap[node.outputs[0]] = out

    @hip_lower.register(FillDiagonalOffset)
    def _(op, node, ctx):
        # reference: tensor/extra_ops.py:980 FillDiagonalOffset (perform :1018): a copy of the
        # matrix with `val` on the diagonal `offset` (not wrapped for tall matrices):
        # a[r0 + i, c0 + i] = val for i < num_of_step — a scatter-set through two index vectors
        b = B(ctx)
        a, val, off = (ctx.vid(v) for v in node.inputs)
        off = b.cast(off, "int64")
        h, w = b.shape_i(a, 0), b.shape_i(a, 1)
        i64 = "int64"
        zero = ["c", 0, i64]
        # inputs: off, h, w
        r0 = b.ew([off], i64, [], [("neg", i64, [["i", 0]]), ("maximum", i64, [["t", 0], zero])])
        c0 = b.ew([off], i64, [], [("maximum", i64, [["i", 0], zero])])
        num = b.ew([off, h, w], i64, [], [
            ("minimum", i64, [["i", 1], ["i", 2]]),                      # 0 min(w, h)
            ("sub", i64, [["i", 2], ["i", 0]]),                          # 1 w - off
            ("add", i64, [["i", 1], ["i", 0]]),                          # 2 h + off
            ("ge", "bool", [["i", 0], zero]),                            # 3
            ("switch", i64, [["t", 3], ["t", 1], ["t", 2]]),             # 4
            ("minimum", i64, [["t", 0], ["t", 4]]),                      # 5
            ("maximum", i64, [["t", 5], zero])])                         # 6
        ar = b.iota(num)
        rows = b.ew([ar, r0], i64, [None], [("add", i64, [["i", 0], ["i", 1]])])
        cols = b.ew([ar, c0], i64, [None], [("add", i64, [["i", 0], ["i", 1]])])
        ctx.vmap[node.outputs[0]] = b.adv_set(a, b.cast(val, b.dtype(a)), [rows, cols])

    def _dims_scalars(b, dims_var, ctx):
        from aesara.tensor.basic import get_vector_length
        nd = int(get_vector_length(dims_var))
        dv = b.cast(ctx.vid(dims_var), "int64")
        return [b.subtensor(dv, [k], static=[]) for k in range(nd)]

    def _strides_of(b, ds, order):
        one = b.const(1, "int64")
        st = [None] * len(ds)
        acc = one
        rng = range(len(ds) - 1, -1, -1) if order == "C" else range(len(ds))
        for k in rng:
            st[k] = acc
            acc = b.iop("mul", acc, ds[k])
        return st, acc                       # strides, total size

    @hip_lower.register(RavelMultiIndex)
    def _(op, node, ctx):
        # reference: tensor/extra_ops.py:1362 RavelMultiIndex (perform :1392 np.ravel_multi_index):
        # sum_k idx_k * stride_k on int64, after the boundary mode (raise / wrap / clip)
        b = B(ctx)
        idx = [b.cast(ctx.vid(v), "int64") for v in node.inputs[:-1]]
        ds = _dims_scalars(b, node.inputs[-1], ctx)
        if len(ds) != len(idx):
            raise UnsupportedOp("RavelMultiIndex: len(multi_index) != len(dims)")
        st, _ = _strides_of(b, ds, op.order)
        shp = static_shape(node.outputs[0].type)
        i64 = "int64"
        fixed = []
        for k, (iv, d) in enumerate(zip(idx, ds)):
            if op.mode == "wrap":
                iv = b.ew([iv, d], i64, b.shape(iv), [("mod", i64, [["i", 0], ["i", 1]])])
            elif op.mode == "clip":
                iv = b.ew([iv, d], i64, b.shape(iv), [
                    ("sub", i64, [["i", 1], ["c", 1, i64]]),
                    ("clip", i64, [["i", 0], ["c", 0, i64], ["t", 0]])])
            else:
                ok = b.all_true(b.ew([iv, d], "bool", b.shape(iv), [
                    ("ge", "bool", [["i", 0], ["c", 0, i64]]), ("lt", "bool", [["i", 0], ["i", 1]]),
                    ("and", "bool", [["t", 0], ["t", 1]])]))
                iv = b.check(iv, ok, "invalid entry in coordinates array")
            fixed.append(iv)
        nodes, terms = [], []
        K = len(fixed)
        for k in range(K):
            nodes.append(("mul", i64, [["i", k], ["i", K + k]]))
            terms.append(["t", k])
        nodes.append(("add", i64, terms) if K > 1 else ("identity", i64, terms))
        ctx.vmap[node.outputs[0]] = b.ew(fixed + st, i64, shp, nodes)

    @hip_lower.register(UnravelIndex)
    def _(op, node, ctx):
        # reference: tensor/extra_ops.py:1283 UnravelIndex (perform :1316 np.unravel_index):
        # coordinate k = (idx // stride_k) % dims_k; an index outside [0, prod(dims)) raises
        b = B(ctx)
        iv = b.cast(ctx.vid(node.inputs[0]), "int64")
        ds = _dims_scalars(b, node.inputs[1], ctx)
        st, total = _strides_of(b, ds, op.order)
        i64 = "int64"
        ok = b.all_true(b.ew([iv, total], "bool", b.shape(iv), [
            ("ge", "bool", [["i", 0], ["c", 0, i64]]), ("lt", "bool", [["i", 0], ["i", 1]]),
            ("and", "bool", [["t", 0], ["t", 1]])]))
        iv = b.check(iv, ok, "index is out of bounds for array with the given dims")
        for k, o in enumerate(node.outputs):
            ctx.vmap[o] = b.ew([iv, st[k], ds[k]], i64, b.shape(iv), [
                ("int_div", i64, [["i", 0], ["i", 1]]), ("mod", i64, [["t", 0], ["i", 2]])])

    # ---------------------------------------------------------------------------------------
    from aesara.tensor.basic import Choose, Default, PermuteRowElements

    @hip_lower.register(Default)
    def _(op, node, ctx):
        # reference: tensor/basic.py:1800 Default (perform :1819): x, or a copy of `default` when the
        # caller passed None for x
        ctx.emit("Default", node)

    @hip_lower.register(Choose)
    def _(op, node, ctx):
        # reference: tensor/basic.py:3773 Choose (perform :3823 np.choose(a, choices, mode)):
        # out[...] = choices[a[...], ...] with a and choices[0] broadcast together — a gather through
        # index arrays (a after its boundary mode, and one arange per remaining dim of choices)
        b = B(ctx)
        a, ch = node.inputs
        if not hasattr(ch.type, "ndim") or not hasattr(ch.type, "dtype"):
            raise UnsupportedOp("non-tensor variable type TypedList (Choose over a list of choices)")
        av, cv = b.cast(ctx.vid(a), "int64"), ctx.vid(ch)
        cnd = ch.type.ndim - 1
        R = max(a.type.ndim, cnd)
        n = b.shape_i(cv, 0)
        i64 = "int64"
        if op.mode == "wrap":
            av = b.ew([av, n], i64, b.shape(av), [("mod", i64, [["i", 0], ["i", 1]])])
        elif op.mode == "clip":
            av = b.ew([av, n], i64, b.shape(av), [("sub", i64, [["i", 1], ["c", 1, i64]]),
                                                  ("clip", i64, [["i", 0], ["c", 0, i64], ["t", 0]])])
        else:
            ok = b.all_true(b.ew([av, n], "bool", b.shape(av), [
                ("ge", "bool", [["i", 0], ["c", 0, i64]]), ("lt", "bool", [["i", 0], ["i", 1]]),
                ("and", "bool", [["t", 0], ["t", 1]])]))
            av = b.check(av, ok, "invalid entry in choice array")
        if a.type.ndim < R:
            av = b.dimshuffle(av, ["x"] * (R - a.type.ndim) + list(range(a.type.ndim)))
        grids = [b.grid(b.shape_i(cv, 1 + j), R - cnd + j, R) for j in range(cnd)]
        out = b.adv_get(cv, [av] + grids, static_shape(node.outputs[0].type))
        ctx.vmap[node.outputs[0]] = out

    @hip_lower.register(PermuteRowElements)
    def _(op, node, ctx):
        # reference: tensor/basic.py:3111 PermuteRowElements (_rec_perform :3167): per row,
        # out = x[y] (forward) or out[y] = x (inverse); leading dims of x and y broadcast by their
        # STATIC patterns — a gather / scatter-set through (aranges of the leading dims..., y)
        b = B(ctx)
        x, y, inv = node.inputs
        invc = _const_value(inv)
        if invc is None:
            raise UnsupportedOp("PermuteRowElements with a run-time `inverse` flag")
        inverse = bool(np.asarray(invc))
        xv, yv = ctx.vid(x), b.cast(ctx.vid(y), "int64")
        nd = x.type.ndim
        xs, ys = static_shape(x.type), static_shape(y.type)
        oshape = static_shape(node.outputs[0].type)
        lead = []                 # per leading dim: 0-d int64 extent of the result
        for d in range(nd - 1):
            if xs[d] == 1 and ys[d] != 1:
                lead.append(b.shape_i(yv, d))
            else:
                lead.append(b.shape_i(xv, d))
        if not inverse:
            grids = []
            for d in range(nd - 1):
                if xs[d] == 1:        # x is broadcast along d: every row reads x[0]
                    z = b.plan.add_const(np.zeros((1,) * nd, "int64"), "int64")
                    b.plan.vars[z].shape = [1] * nd
                    grids.append(z)
                else:
                    grids.append(b.grid(lead[d], d, nd))
            out = b.adv_get(xv, grids + [yv], oshape)
        else:
            last = b.shape_i(xv, nd - 1)
            empty = b.raw("AllocEmpty", lead + [last], str(x.type.dtype), oshape,
                          {"dtype": str(x.type.dtype)})
            grids = [b.grid(lead[d], d, nd) for d in range(nd - 1)]
            out = b.adv_set(empty, xv, grids + [yv])
        ctx.vmap[node.outputs[0]] = out

    # ---------------------------------------------------------------------------------------
    from aesara.tensor.sort import TopKOp

    @hip_lower.register(TopKOp)
    def _(op, node, ctx):
        # reference: tensor/sort.py:309 TopKOp (perform :417, _topk_py_impl :236): the k largest
        # (k > 0) / |k| smallest (k < 0) along `axis`, in NO guaranteed order (np.partition) —
        # here: the tail / head of the row sort (K13), so the result comes out ascending
        b = B(ctx)
        x, kth = node.inputs
        nd = x.type.ndim
        axis = int(op.axis) % nd
        xv, kv = ctx.vid(x), b.cast(ctx.vid(kth), "int64")
        n = b.shape_i(xv, axis)
        i64 = "int64"
        ok = b.ew([kv, n], "bool", [], [
            ("neq", "bool", [["i", 0], ["c", 0, i64]]), ("abs", i64, [["i", 0]]),
            ("le", "bool", [["t", 1], ["i", 1]]), ("and", "bool", [["t", 0], ["t", 2]])])
        xv = b.check(xv, ok, "topk: kth must not be zero and must satisfy abs(kth) <= the size of the axis")
        start = b.ew([kv, n], i64, [], [("gt", "bool", [["i", 0], ["c", 0, i64]]),
                                        ("sub", i64, [["i", 1], ["i", 0]]),
                                        ("switch", i64, [["t", 0], ["t", 1], ["c", 0, i64]])])
        stop = b.ew([kv, n], i64, [], [("gt", "bool", [["i", 0], ["c", 0, i64]]),
                                       ("neg", i64, [["i", 0]]),
                                       ("switch", i64, [["t", 0], ["i", 1], ["t", 1]])])
        ax = b.const(axis, "int64")
        idx = [{"slice": [None, None, None]}] * axis + [{"slice": ["in", "in", None]}]
        shp = [None] * nd
        outs = []
        if op.return_values:
            sv = b.raw("Sort", [xv, ax], str(x.type.dtype), shp, {"kind": "quicksort"})
            outs.append(b.raw("Subtensor", [sv, start, stop], str(x.type.dtype), shp, {"idx_list": idx}))
        if op.return_indices:
            si = b.raw("ArgSort", [xv, ax], "int64", shp, {"kind": "stable"})
            si = b.raw("Subtensor", [si, start, stop], "int64", shp, {"idx_list": idx})
            outs.append(b.cast(si, str(op.idx_dtype)))
        for o, v in zip(node.outputs, outs):
            ctx.vmap[o] = v

    # ---------------------------------------------------------------------------------------
    from aesara.tensor.nnet.basic import (CrossentropySoftmax1HotWithBiasDx,
                                          CrossentropySoftmaxArgmax1HotWithBias,
                                          Prepend_scalar_constant_to_each_row,
                                          Prepend_scalar_to_each_row, SoftmaxWithBias)

    def _softmax_rows(b, row, dt):
        """(max, exp(row - max), sum of that in float64) of a matrix, row-wise."""
        m = b.dimshuffle(b.careduce(row, "maximum", [1]), [0, "x"])
        e = b.ew([row, m], dt, b.shape(row), [("sub", dt, [["i", 0], ["i", 1]]), ("exp", dt, [["t", 0]])])
        s = b.dimshuffle(b.careduce(e, "add", [1], "float64", "float64"), [0, "x"])
        return m, e, s

    def _normalise(b, e, s, dt):
        return b.ew([e, s], dt, b.shape(e), [
            ("reciprocal", "float64", [["i", 1]]), ("mul", "float64", [["i", 0], ["t", 0]]),
            ("cast", dt, [["t", 1]])])

    def _plus_bias(b, ctx, x, bias, dt):
        xv, bv = b.cast(ctx.vid(x), dt), b.cast(ctx.vid(bias), dt)
        ncol, nb = b.shape_i(xv, 1), b.shape_i(bv, 0)
        ok = b.ew([ncol, nb], "bool", [], [("eq", "bool", [["i", 0], ["i", 1]])])
        xv = b.check(xv, ok, "b must have same number of columns as x")
        return b.ew([xv, b.dimshuffle(bv, ["x", 0])], dt, b.shape(xv), [("add", dt, [["i", 0], ["i", 1]])])

    @hip_lower.register(SoftmaxWithBias)
    def _(op, node, ctx):
        # reference: tensor/nnet/basic.py:57 SoftmaxWithBias (perform :84): softmax(x + b) per row
        b = B(ctx)
        dt = str(node.outputs[0].type.dtype)
        row = _plus_bias(b, ctx, node.inputs[0], node.inputs[1], dt)
        _, e, s = _softmax_rows(b, row, dt)
        ctx.vmap[node.outputs[0]] = _normalise(b, e, s, dt)

    @hip_lower.register(CrossentropySoftmaxArgmax1HotWithBias)
    def _(op, node, ctx):
        # reference: tensor/nnet/basic.py:458 CrossentropySoftmaxArgmax1HotWithBias (perform :512):
        # row = x[i] + b; am = argmax(row); sm = softmax(row); nll = -row[y_i] + max + log(sum exp)
        b = B(ctx)
        x, bias, y = node.inputs
        dt = str(node.outputs[1].type.dtype)
        row = _plus_bias(b, ctx, x, bias, dt)
        yv = b.cast(ctx.vid(y), "int64")
        okn = b.ew([b.shape_i(row, 0), b.shape_i(yv, 0)], "bool", [], [("eq", "bool", [["i", 0], ["i", 1]])])
        row = b.check(row, okn, "y_idx must have same number of rows as x")
        okp = b.all_true(b.ew([yv], "bool", b.shape(yv), [("ge", "bool", [["i", 0], ["c", 0, "int64"]])]))
        yv = b.check(yv, okp, "y_i value out of bounds")
        m, e, s = _softmax_rows(b, row, dt)
        sm = _normalise(b, e, s, dt)
        am = b.cast(b.raw("Argmax", [row], "int64", [None], {"axis": [1]}), str(node.outputs[2].type.dtype))
        picked = b.adv_get(row, [b.iota(b.shape_i(row, 0)), yv], [None])
        m1 = b.careduce(row, "maximum", [1])
        s1 = b.careduce(e, "add", [1], "float64", "float64")
        ndt = str(node.outputs[0].type.dtype)
        nll = b.ew([picked, m1, s1], ndt, [None], [
            ("log", "float64", [["i", 2]]), ("cast", dt, [["t", 0]]), ("neg", dt, [["i", 0]]),
            ("add", dt, [["t", 2], ["i", 1], ["t", 1]]), ("cast", ndt, [["t", 3]])])
        for o, v in zip(node.outputs, (nll, sm, am)):
            ctx.vmap[o] = v

    @hip_lower.register(CrossentropySoftmax1HotWithBiasDx)
    def _(op, node, ctx):
        # reference: tensor/nnet/basic.py:716 CrossentropySoftmax1HotWithBiasDx (perform :737):
        # dx[i] = dy_i * sm[i]; dx[i, y_i] -= dy_i
        b = B(ctx)
        dy, sm, y = node.inputs
        dt = str(node.outputs[0].type.dtype)
        dv, sv, yv = b.cast(ctx.vid(dy), dt), b.cast(ctx.vid(sm), dt), b.cast(ctx.vid(y), "int64")
        okp = b.all_true(b.ew([yv], "bool", b.shape(yv), [("ge", "bool", [["i", 0], ["c", 0, "int64"]])]))
        yv = b.check(yv, okp, "y_i value out of bounds")
        dcol = b.dimshuffle(dv, ["x", "x"] if dy.type.ndim == 0 else [0, "x"])
        p = b.ew([dcol, sv], dt, b.shape(sv), [("mul", dt, [["i", 0], ["i", 1]])])
        nrow = b.shape_i(sv, 0)
        neg = b.ew([dv], dt, b.shape(dv), [("neg", dt, [["i", 0]])])
        if dy.type.ndim == 0 or static_shape(dy.type) == [1]:
            neg = b.alloc(b.subtensor(neg, [0], static=[]) if dy.type.ndim else neg, [nrow])
        ctx.vmap[node.outputs[0]] = b.adv_set(p, neg, [b.iota(nrow), yv], inc=True)

    def _prepend(b, val, mat):
        dt = b.dtype(mat)
        col = b.alloc(b.cast(val, dt), [b.shape_i(mat, 0), b.const(1, "int64")], [None, 1])
        return b.join(1, [col, mat], [None, None])

    @hip_lower.register(Prepend_scalar_constant_to_each_row)
    def _(op, node, ctx):
        # reference: tensor/nnet/basic.py:1655 Prepend_scalar_constant_to_each_row (perform :1677):
        # out[:, 0] = val; out[:, 1:] = mat
        b = B(ctx)
        mat = ctx.vid(node.inputs[0])
        val = b.const(np.asarray(op.val.data), b.dtype(mat))
        ctx.vmap[node.outputs[0]] = _prepend(b, val, mat)

    @hip_lower.register(Prepend_scalar_to_each_row)
    def _(op, node, ctx):
        # reference: tensor/nnet/basic.py:1707 Prepend_scalar_to_each_row (perform :1723)
        b = B(ctx)
        ctx.vmap[node.outputs[0]] = _prepend(b, ctx.vid(node.inputs[0]), ctx.vid(node.inputs[1]))

    # ---------------------------------------------------------------------------------------
    from aesara.compile.builders import OpFromGraph
    from aesara.compile.ops import FromFunctionOp
    from aesara.printing import Print

    @hip_lower.register(OpFromGraph)
    def _(op, node, ctx):
        # reference: compile/builders.py:188 OpFromGraph (perform :1040 runs the compiled inner
        # function; inline=True expands it in the rewriter :1047): here ALWAYS expanded — the inner
        # graph is rewritten like the reference rewrites it when it compiles ``op.fn`` and lowered in
        # place, so its Elemwise / reductions fuse with their surroundings
        inner = op.fgraph.clone()
        if ctx.inner_rewriter is not None:
            ctx.inner_rewriter.rewrite(inner)
        for iv, ov in zip(inner.inputs, node.inputs):
            ctx.vmap[iv] = ctx.vid(ov)
        for n in inner.toposort():
            hip_lower(n.op, n, ctx)
        for iv, ov in zip(inner.outputs, node.outputs):
            ctx.vmap[ov] = ctx.vid(iv)

    @hip_lower.register(Print)
    def _(op, node, ctx):
        # reference: printing.py:825 Print (perform :863): identity; as a side effect the host sees
        # the VALUE (``global_fn(op, value)``, default prints ``message attr = value``) — a device ->
        # host copy behind the producing kernel, then the callback; the output aliases the input
        def call(x, _op=op):
            _op.global_fn(_op, x)
        ctx.emit("HostCall", node, {"fn": call, "view": True, "what": "Print{%s}" % op.message})

    @hip_lower.register(FromFunctionOp)
    def _(op, node, ctx):
        # reference: compile/ops.py:226 FromFunctionOp (``as_op``; perform :258): the Op IS a Python
        # function of ndarrays — by definition host code: operands are copied to the host, the
        # function runs, its results are uploaded (checked against the declared output types)
        otypes = [(str(o.type.dtype), o.type.ndim) for o in node.outputs]

        def call(*xs, _op=op, _node=node):
            cells = [[None] for _ in _node.outputs]
            _op.perform(_node, list(xs), cells)
            return [c[0] for c in cells]
        ctx.emit("HostCall", node, {"fn": call, "view": False, "otypes": otypes,
                                    "what": str(op)})
