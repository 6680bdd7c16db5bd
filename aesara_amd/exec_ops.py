"""Shape, allocation, indexing, sorting and cumulative Ops of a PlanExecutor (views on the host,
csrc/copy.hip / index.hip / sort.hip / cumulative.hip kernels; reference: tensor/basic.py,
tensor/subtensor.py, tensor/shape.py, tensor/sort.py, tensor/extra_ops.py).

Part of :class:`aesara_amd.executor.PlanExecutor` (a mixin: the methods run on the executor's
state; split out of executor.py in round 4, no behaviour change)."""
from __future__ import annotations

from .exec_common import *  # noqa: F401,F403
from .exec_common import (_I64, _VP, _i64arr, _Kernels, _FakeBuf, _CAST_SCALARS, _prod, _Arena, _os, _time)  # noqa: F401


class OpsMixin:

    # ------------------------------------------------------------------ view ops ------
    def _op_DimShuffle(self, node, args):
        x = args[0]
        order = node.params["new_order"]
        if not isinstance(x, DevArray):
            x = np.asarray(x)
            keep = [d for d in order if d != "x"]
            v = x.transpose(keep + [d for d in range(x.ndim) if d not in keep])
            return [v.reshape([v.shape[keep.index(d)] if d != "x" else 1 for d in order])]
        for d in range(x.ndim):
            if d not in order and x.shape[d] != 1:
                raise ValueError("Cannot drop a non-broadcastable dimension")
        shape = [1 if d == "x" else x.shape[d] for d in order]
        strides = [0 if d == "x" else x.strides[d] for d in order]
        return [x.view(shape, strides)]

    def _op_ViewOp(self, node, args):
        return [args[0]]

    def _op_SpecifyShape(self, node, args):
        """reference: tensor/shape.py:439 SpecifyShape.perform (shape check, then a view)."""
        p = node.params
        given = [self.host_int(v) for v in args[1:]]
        dims = p.get("dims", list(range(len(given))))
        ndim = p.get("ndim", len(np.shape(args[0])) if not isinstance(args[0], DevArray)
                     else args[0].ndim)
        xshape = tuple(args[0].shape) if isinstance(args[0], DevArray) else np.shape(args[0])
        want = [None] * ndim
        for d, sv in zip(dims, given):
            want[d] = sv
        if len(xshape) != ndim:
            raise AssertionError(f"SpecifyShape: Got {len(xshape)} dimensions (shape {xshape}), "
                                 f"expected {ndim} dimensions with shape {tuple(want)}.")
        if not all(xs == sv for xs, sv in zip(xshape, want) if sv is not None):
            raise AssertionError(f"SpecifyShape: Got shape {xshape}, expected {tuple(want)}.")
        return [args[0]]

    def _op_Assert(self, node, args):
        """reference: raise_op.py:94 CheckAndRaise.perform — conditions are 0-d; device-valued
        ones cost a host read (and keep the plan out of replay mode)."""
        for c in args[1:]:
            if self.dry_run and isinstance(c, DevArray) and getattr(c.buf, "data", None) is None:
                continue            # a dry run has no values for computed conditions
            ok = bool(np.all(self.host_array(c))) if isinstance(c, DevArray) else bool(np.all(c))
            if not ok:
                import builtins
                exc = getattr(builtins, node.params.get("exc_type", "AssertionError"),
                              AssertionError)
                if not (isinstance(exc, type) and issubclass(exc, Exception)):
                    exc = AssertionError
                raise exc(node.params.get("msg", ""))
        return [args[0]]
    _op_ScalarFromTensor = _op_ViewOp
    _op_TensorFromScalar = _op_ViewOp

    def _op_DeepCopyOp(self, node, args):
        x = args[0]
        if isinstance(x, DevArray):
            return [self.materialize(x)]
        return [np.array(x, copy=True)]

    def _op_Shape_i(self, node, args):
        return [np.asarray(np.shape(args[0])[node.params["i"]] if not isinstance(args[0], DevArray)
                           else args[0].shape[node.params["i"]], dtype="int64")]

    def _op_Shape(self, node, args):
        shp = args[0].shape if isinstance(args[0], DevArray) else np.shape(args[0])
        return [np.asarray(shp, dtype="int64")]

    def _op_MakeVector(self, node, args):
        if all(not isinstance(a, DevArray) for a in args):
            return [np.array([np.asarray(a).reshape(()) for a in args], dtype=node.params["dtype"])]
        vals = [self.host_scalar(a) for a in args]
        return [np.array(vals, dtype=node.params["dtype"])]

    def _op_Reshape(self, node, args):
        x = args[0]
        shp = [int(s) for s in self.host_array(args[1]).reshape(-1)]
        nd = node.params.get("ndim")
        if nd is not None and len(shp) != nd:          # tensor/shape.py:649 Reshape.perform
            raise ValueError(f"Shape argument to Reshape has incorrect length: {len(shp)}, should be {nd}")
        if not isinstance(x, DevArray):
            return [np.reshape(x, shp)]
        want = tuple(shp)
        if shp.count(-1) > 1:                          # np.reshape's own errors from here on
            raise ValueError("can only specify one unknown dimension")
        if any(s < -1 for s in shp):
            raise ValueError("negative dimensions not allowed")
        if -1 in shp:
            known = _prod(([s for s in shp if s != -1]))
            if known == 0 or x.size % known:
                raise ValueError(f"cannot reshape array of size {x.size} into shape {want}")
            shp[shp.index(-1)] = x.size // known
        if _prod((shp)) != x.size:
            raise ValueError(f"cannot reshape array of size {x.size} into shape {want}")
        x = self.contiguous(x)
        return [x.view(shp, contiguous_strides(shp))]

    def _op_Subtensor(self, node, args):
        x = args[0]
        index = hostops.resolve_index(node.params["idx_list"], [self.host_int(a) for a in args[1:]])
        if not isinstance(x, DevArray):
            return [np.asarray(x)[index]]
        shape, strides, off = hostops.view_from_index(x.shape, x.strides, x.offset, index)
        return [x.view(shape, strides, off)]

    # ------------------------------------------------------------------ alloc / set ---
    def _op_AllocEmpty(self, node, args):
        return [self.alloc([self.host_int(a) for a in args], node.params["dtype"])]

    def _op_Alloc(self, node, args):
        shape = [self.host_int(a) for a in args[1:]]
        ov = self.plan.vars[node.outputs[0]]
        val = self.to_device(args[0])
        if val.dtype != ov.dtype:
            raise TypeError("Alloc value dtype mismatch")
        out = self.alloc(shape, ov.dtype)
        self.copy_into(out, val)
        return [out]

    def _op_BroadcastTo(self, node, args):
        x = self.to_device(args[0])
        shape = [self.host_int(a) for a in args[1:]]
        lead = len(shape) - x.ndim
        if lead < 0:
            raise ValueError("BroadcastTo: target has fewer dimensions than the input")
        strides = [0] * lead
        for d in range(x.ndim):
            if x.shape[d] == shape[lead + d]:
                strides.append(x.strides[d])
            elif x.shape[d] == 1:
                strides.append(0)
            else:
                raise ValueError(f"operands could not be broadcast together with remapped shapes "
                                 f"{x.shape} -> {tuple(shape)}")
        return [x.view(shape, strides)]

    def _own_or_copy(self, vid, x: DevArray) -> DevArray:
        """The buffer an update op may overwrite: ``x`` itself when it is a private fresh
        allocation read for the last time by the current step, else a copy (the reference's
        non-inplace semantics: perform() starts with ``x.copy()``, subtensor.py:1556 / :2688)."""
        if vid in self._private and self._last_use.get(vid) == self._si and x.is_contiguous() \
                and self._alias_last.get(vid, self._si) <= self._si and vid not in self._alias_out:
            return x
        return self.materialize(x)

    def _op_IncSubtensor(self, node, args):
        x, y = self.to_device(args[0]), self.to_device(args[1])
        index = hostops.resolve_index(node.params["idx_list"], [self.host_int(a) for a in args[2:]])
        out = self._own_or_copy(node.inputs[0], x)
        shape, strides, off = hostops.view_from_index(out.shape, out.strides, out.offset, index)
        sub = out.view(shape, strides, off)
        if y.ndim > sub.ndim:
            raise ValueError("IncSubtensor: value has more dimensions than the indexed view")
        y = self.cast(y, out.dtype)
        self.copy_into(sub, y, accumulate=not node.params["set_instead_of_inc"])
        return [out]

    def _op_Join(self, node, args):
        axis = self.host_int(args[0])
        parts = [self.to_device(a) for a in args[1:]]
        nd = parts[0].ndim
        if axis < -nd or axis >= nd:
            raise IndexError(f"Join axis {axis} out of bounds [0, {nd})")
        axis %= nd
        shape = list(parts[0].shape)
        shape[axis] = sum(p.shape[axis] for p in parts)
        for p in parts:
            if any(p.shape[d] != shape[d] for d in range(nd) if d != axis):
                raise ValueError("all the input array dimensions except for the concatenation "
                                 "axis must match exactly")
        out = self.alloc(shape, self.plan.vars[node.outputs[0]].dtype)
        pos = 0
        for p in parts:
            idx = tuple(slice(pos, pos + p.shape[axis]) if d == axis else slice(None)
                        for d in range(nd))
            sh, st, off = hostops.view_from_index(out.shape, out.strides, out.offset, idx)
            # (Join.make_node upcasts to the common dtype; the arrays arrive in their own:
            # tensor/basic.py:2214 — np.concatenate's own promotion in perform :2342)
            self.copy_into(out.view(sh, st, off), p if p.dtype == out.dtype else self.cast(p, out.dtype))
            pos += p.shape[axis]
        return [out]

    # ------------------------------------------------------------------ index ops -----
    def _bad(self):
        """Error words of the index kernels: ONE int64 slot per step of the outermost plan, so
        that a bad index is reported with the step (-> Apply node) that met it.  Executors of
        Scan step plans write into the slot of the Scan step of their root."""
        root = self._root or self
        if root._bad_index is None:
            n = len(root.steps) + 1
            root._bad_index = (_FakeBuf(n, "int64") if self.dry_run else
                               torch.zeros(n, dtype=torch.int64, device=root.device))
        return root._bad_index

    def _bad_ptr(self):
        root = self._root or self
        return self._bad().data_ptr() + 8 * max(root._si, 0)

    def _rows(self, a: DevArray):
        """(row stride, elements per row) of `a` viewed as [rows, contiguous inner]."""
        inner = a.shape[1:]
        if a.ndim > 1 and not a.view(inner, a.strides[1:]).is_contiguous():
            return None
        n = 1
        for s in inner:
            n *= s
        return (a.strides[0] if a.shape[0] != 1 else n), n

    def _op_Argmax(self, node, args):
        """reference: tensor/math.py:388 Argmax.perform.  Kept axes in front, reduced axes last
        (a stride permutation); when the reduced axes collapse to one stride the kernel reads
        the view in place, otherwise one contiguous copy is made first."""
        x = self.to_device(args[0])
        axes = list(node.params["axis"])
        keep = [d for d in range(x.ndim) if d not in axes]
        kshape = [x.shape[d] for d in keep]
        rshape = [x.shape[d] for d in axes]
        k = _prod((rshape)) if rshape else 1
        n = _prod((kshape)) if kshape else 1
        if k == 0:
            raise ValueError("attempt to get argmax of an empty sequence")
        out = self.alloc(kshape, "int64")
        if n == 0:
            return [out]
        t = x.view(kshape + rshape, [x.strides[d] for d in keep] + [x.strides[d] for d in axes])

        def collapse(shape, strides):
            """single (size, stride) covering the dims in C order, or None"""
            size, stride = 1, None
            for sh, st in reversed(list(zip(shape, strides))):
                if sh == 1:
                    continue
                if stride is None:
                    stride = st
                elif st != stride * size:
                    return None
                size *= sh
            return size, (stride if stride is not None else 0)

        rows = collapse(t.shape[:len(keep)], t.strides[:len(keep)])
        cols = collapse(t.shape[len(keep):], t.strides[len(keep):])
        if rows is None or cols is None:
            t = self.materialize(t)
            rs, cs = k, 1
        else:
            rs, cs = rows[1], cols[1]
        need = int(lib.ahip_argmax_ws_bytes(dtype_code(x.dtype), n, k, rs, cs))
        ws = self.alloc((need,), "uint8") if need else None
        self._launch("ahip_argmax_rows", (dtype_code(x.dtype), _VP(t.ptr), n, k, rs, cs,
                                          _VP(out.ptr), _VP(ws.ptr) if ws is not None else None,
                                          need, self._stream()))
        return [out]

    # ------------------------------------------------------------------ ARange / N-d gather ---
    def _host_scalar(self, v):
        a = self.host_array(v) if isinstance(v, DevArray) else np.asarray(v)
        return a.reshape(()).item()

    def _op_ARange(self, node, args):
        """reference: tensor/basic.py:2937 ARange.perform — np.arange(start, stop, step, dtype);
        length and fill rule (first + i * delta in the output dtype) are NumPy's."""
        start, stop, step = (self._host_scalar(a) for a in args)
        dt = node.params["dtype"]
        if step == 0:
            raise ValueError("Maximum allowed size exceeded")   # np.arange(a, b, 0)
        n = max(0, int(np.ceil((stop - start) / step)))
        out = self.alloc((n,), dt)
        if n:
            npdt = np.dtype(dt)
            first = np.array([start, start + step]).astype(npdt)          # [first, next]
            delta = (first[1:] - first[:1]).astype(npdt)
            self._launch("ahip_arange", (dtype_code(dt), first.ctypes.data_as(_VP),
                                         delta.ctypes.data_as(_VP), n, _VP(out.ptr),
                                         self._stream()))
        return [out]

    # ------------------------------------------------------------------ Eye / Tri / diagonals ---
    def _one(self, dtype):
        key = ("one", dtype)
        c = self._const_cache.get(key)
        if c is None:
            c = self._const_cache[key] = self._from_numpy(np.ones((), dtype))
        return c

    @staticmethod
    def _diag_span(n, m, k):
        """(first row, first col, length) of the k-th diagonal of an n x m matrix"""
        r0, c0 = max(0, -k), max(0, k)
        return r0, c0, max(0, min(n - r0, m - c0))

    def _op_Eye(self, node, args):
        """reference: tensor/basic.py:1278 Eye.perform (np.eye(n, m, k, dtype)): zero fill (K7) +
        a strided copy of ones onto the diagonal view (K8)."""
        n, m, k = (self.host_int(a) for a in args)
        if n < 0 or m < 0:
            raise ValueError("negative dimensions are not allowed")
        out = self.alloc((n, m), node.params["dtype"])
        self.fill_zero(out)
        r0, c0, L = self._diag_span(n, m, k)
        if L:
            self.copy_into(out.view((L,), (m + 1,), out.offset + r0 * m + c0), self._one(out.dtype))
        return [out]

    def _op_Tri(self, node, args):
        """reference: tensor/basic.py:1000 Tri.perform (np.tri): out[i, j] = (j <= i + k), one
        generated Elemwise over a row-index column and a column-index row."""
        n, m, k = (self.host_int(a) for a in args)
        if n < 0 or m < 0:
            raise ValueError("negative dimensions are not allowed")
        dt = node.params["dtype"]
        out = self.alloc((n, m), dt)
        if out.size:
            ar = self.alloc((max(n, m),), "int64")
            first, delta = np.array([0], "int64"), np.array([1], "int64")
            self._launch("ahip_arange", (dtype_code("int64"), first.ctypes.data_as(_VP),
                                         delta.ctypes.data_as(_VP), max(n, m), _VP(ar.ptr),
                                         self._stream()))
            sc = {"n_in": 2, "nodes": [
                {"op": "add", "in": [["i", 0], ["c", int(k), "int64"]], "dtype": "int64"},
                {"op": "le", "in": [["i", 1], ["t", 0]], "dtype": "bool"},
                {"op": "cast", "in": [["t", 1]], "dtype": dt}], "out": [["t", 2]]}
            rows, cols = ar.view((n, 1), (1, 0)), ar.view((1, m), (0, 1))
            self._launch_elemwise(sc, [rows, cols], [[1, 0], [0, 1]], [out], [0], [n, m])
        return [out]

    def _op_FillDiagonal(self, node, args):
        """reference: tensor/extra_ops.py:906 FillDiagonal.perform: copy, then `val` along the
        main diagonal — a strided view of the copy (K8); 2-d may be rectangular."""
        x = self.to_device(args[0])
        val = self.cast(self.to_device(args[1]), x.dtype)
        if x.ndim < 2:
            raise ValueError("FillDiagonal needs at least 2 dimensions")
        out = self.materialize(x)
        if x.ndim == 2:
            n, m = x.shape
            L = min(m, -(-(m * m) // (m + 1))) if m else 0     # len(range(0, m*m, m+1))
            L = min(L, n)
        else:
            if len(set(x.shape)) != 1:
                raise ValueError("All dimensions of input must be of equal length")
            L = x.shape[0]
        if L:
            self.copy_into(out.view((L,), (sum(out.strides),), out.offset), val)
        return [out]

    def _op_ExtractDiag(self, node, args):
        """reference: tensor/basic.py:3402 ExtractDiag.perform: x.diagonal(offset, axis1, axis2) —
        a stride-(s1+s2) view with the diagonal as the last dim, copied unless view=True."""
        x = self.to_device(args[0])
        p = node.params
        a1, a2 = p["axis1"] % x.ndim, p["axis2"] % x.ndim
        if a1 == a2:
            raise ValueError("axis1 and axis2 cannot be the same")
        r0, c0, L = self._diag_span(x.shape[a1], x.shape[a2], p["offset"])
        keep = [d for d in range(x.ndim) if d not in (a1, a2)]
        v = x.view([x.shape[d] for d in keep] + [L],
                   [x.strides[d] for d in keep] + [x.strides[a1] + x.strides[a2]],
                   x.offset + r0 * x.strides[a1] + c0 * x.strides[a2])
        return [v if p["view"] else self.materialize(v)]

    def _op_AllocDiag(self, node, args):
        """reference: tensor/basic.py:3523 AllocDiag.perform: zeros, the input copied onto the
        offset diagonal of the two new trailing axes, then those axes moved to axis1 / axis2."""
        x = self.to_device(args[0])
        p = node.params
        off = p["offset"]
        ax1, ax2 = min(p["axis1"], p["axis2"]), max(p["axis1"], p["axis2"])
        L = x.shape[-1]
        n = L + abs(off)
        res = self.alloc(tuple(x.shape[:-1]) + (n, n), x.dtype)
        self.fill_zero(res)
        if L:
            r0, c0 = max(0, -off), max(0, off)
            dv = res.view(tuple(x.shape[:-1]) + (L,), tuple(res.strides[:-2]) + (n + 1,),
                          res.offset + r0 * n + c0)
            self.copy_into(dv, x)
        if x.ndim > 1:
            axes = list(range(x.ndim - 1))
            last = axes[-1]
            axes = axes[:ax1] + [last + 1] + axes[ax1:]
            axes = axes[:ax2] + [last + 2] + axes[ax2:]
            res = res.view([res.shape[q] for q in axes], [res.strides[q] for q in axes])
        return [res]

    def _flat_rows(self, x, idx):
        """Flat row indices (device int64 [n]) of x[idx0, idx1, ...] with integer index arrays
        (NumPy advanced indexing on the leading dims), the broadcast index shape, and x as
        contiguous rows.  reference: tensor/subtensor.py:2607 / :2688."""
        x = self.contiguous(self.to_device(x))
        idx = [self.to_device(i) for i in idx]
        k = len(idx)
        if k > x.ndim:
            raise IndexError("too many indices for array")
        try:
            bshape = list(np.broadcast_shapes(*[tuple(i.shape) for i in idx]))
        except ValueError:
            raise IndexError("shape mismatch: indexing arrays could not be broadcast together")
        n = _prod((bshape))
        ptrs, dts, strides = [], [], []
        keep = []
        for i in idx:
            pad = [1] * (len(bshape) - i.ndim)
            v = i.view(pad + list(i.shape), [0] * len(pad) + list(i.strides))
            st = [0 if v.shape[d] == 1 and bshape[d] != 1 else v.strides[d]
                  for d in range(len(bshape))]
            sh, stt = collapse_dims(list(bshape), [st])
            if len(sh) != 1:
                full = self.alloc(bshape, i.dtype)
                self.copy_into(full, v.view(bshape, st))
                keep.append(full)
                v, st1 = full, 1
            else:
                st1 = stt[0][0]
            ptrs.append(v.ptr)
            dts.append(dtype_code(i.dtype))
            strides.append(st1)
        dims = list(x.shape[:k])
        mults = [_prod((x.shape[d + 1:k])) for d in range(k)]
        lin = self.alloc((n,), "int64")
        if n:
            self._launch("ahip_linearize_indices", (
                k, (_VP * k)(*ptrs), (C.c_int * k)(*dts), _i64arr(strides), _i64arr(dims),
                _i64arr(mults), n, _VP(lin.ptr), _VP(self._bad_ptr()), self._stream()))
        row = _prod((x.shape[k:]))
        nrows = _prod((dims))
        return x, lin, bshape, nrows, row, keep

    def _sort(self, node, args, want_idx):
        """reference: tensor/sort.py:48 SortOp.perform / :184 ArgSortOp.perform — np.sort /
        np.argsort along `axis` (None: the flattened array).  The axis is moved last (a view),
        rows are sorted by the LDS bitonic kernel (K13), the result is viewed back."""
        x = self.to_device(args[0])
        axis = args[1]
        axis = None if axis is None else self.host_int(axis)
        if axis is None or x.ndim == 0:
            v = self.contiguous(x).view((x.size,), (1,))
            perm = None
        else:
            if axis < -x.ndim or axis >= x.ndim:
                raise ValueError(f"axis(={axis}) out of bounds")
            axis %= x.ndim
            perm = [d for d in range(x.ndim) if d != axis] + [axis]
            v = x.view([x.shape[d] for d in perm], [x.strides[d] for d in perm])
        n = v.shape[-1]
        lead = list(v.shape[:-1])
        rows = _prod(lead) if lead else 1
        osh, ost = collapse_dims(lead or [1], [list(v.strides[:-1]) or [0]])
        if len(osh) != 1:
            v = self.materialize(v)
            rs = n
        else:
            rs = ost[0][0]
        out = self.alloc(lead + [n], "int64" if want_idx else x.dtype)
        if out.size and n > int(lib.ahip_sort_max_row(dtype_code(x.dtype))):
            # rows longer than LDS: chunks sorted in LDS, then rank-based merge passes
            need = int(lib.ahip_sort_large_ws_bytes(dtype_code(x.dtype), rows, n))
            ws = self.alloc((need,), "uint8")
            self._launch("ahip_sort_rows_large", (dtype_code(x.dtype), _VP(v.ptr), rows, n, rs,
                                                  v.strides[-1] if n > 1 else 1,
                                                  None if want_idx else _VP(out.ptr),
                                                  _VP(out.ptr) if want_idx else None, _VP(ws.ptr),
                                                  need, self._stream()))
        elif out.size:
            self._launch("ahip_sort_rows", (dtype_code(x.dtype), _VP(v.ptr), rows, n, rs,
                                            v.strides[-1] if n > 1 else 1,
                                            None if want_idx else _VP(out.ptr),
                                            _VP(out.ptr) if want_idx else None, self._stream()))
        if perm is None:
            return [out]
        inv = [perm.index(d) for d in range(x.ndim)]
        return [out.view([out.shape[q] for q in inv], [out.strides[q] for q in inv])]

    def _op_LexArgSortRows(self, node, args):
        """Stable LEXICOGRAPHIC argsort of the rows of a matrix (first column most significant):
        what ``np.unique(x, axis=k)`` orders its items by (tensor/extra_ops.py:1216).  One stable
        argsort per column, last column first (LSD), each on the column gathered through the
        permutation so far — the sort / gather kernels, no new one."""
        x = self.to_device(args[0])
        if x.ndim != 2:
            raise ValueError("LexArgSortRows needs a matrix")
        n, m = x.shape

        class _P:
            params = {"dtype": "int64"}
        perm = self._op_ARange(_P, [np.int64(0), np.int64(n), np.int64(1)])[0]
        for c in range(m - 1, -1, -1):
            col = x.view((n,), (x.strides[0],), x.offset + c * x.strides[1])
            keys = self._op_AdvancedSubtensor1(None, [col, perm])[0]
            order = self._sort(None, [keys, np.int64(0)], True)[0]
            perm = self._op_AdvancedSubtensor1(None, [perm, order])[0]
        return [perm]

    def _op_Sort(self, node, args):
        return self._sort(node, args, False)

    def _op_ArgSort(self, node, args):
        return self._sort(node, args, True)

    def _nonzero(self, x):
        """Index arrays (int64 device vectors, one per dim) of the non-zero entries of ``x`` in C
        order: flag -> inclusive running count (K12) -> ONE host read of the total -> coordinate
        write (ahip_nonzero_write).  reference: tensor/basic.py:870 Nonzero.perform."""
        x = self.to_device(x)
        if x.ndim == 0 or x.ndim > 8:
            raise ValueError("Nonzero needs 1..8 dims")
        n = x.size
        if n == 0:
            return [self.alloc((0,), "int64") for _ in range(x.ndim)]
        sc = {"n_in": 1, "nodes": [{"op": "neq", "in": [["i", 0], ["c", 0, x.dtype]], "dtype": "bool"},
                                   {"op": "cast", "in": [["t", 0]], "dtype": "int64"}],
              "out": [["t", 1]]}
        sc = self._nz_prog.setdefault(x.dtype, sc)
        flag = self.alloc(x.shape, "int64")
        self._launch_elemwise(sc, [x], [[0 if m == 1 else st for st, m in zip(x.strides, x.shape)]],
                              [flag], [0], list(x.shape))
        cnt = self.alloc((n,), "int64")
        need = int(lib.ahip_cumulative_ws_bytes(dtype_code("int64"), 1, n, 1))
        ws = self.alloc((need,), "uint8") if need else None
        self._launch("ahip_cumulative", (dtype_code("int64"), 0, _VP(flag.ptr), 1, n, 1, 0, 1, 0,
                                         _VP(cnt.ptr), _VP(ws.ptr) if ws is not None else None,
                                         need, self._stream()))
        if self.dry_run:
            total = n       # value unknown in a dry run: the largest possible count
        else:
            total = int(self.host_array(cnt.view((1,), (1,), cnt.offset + n - 1))[0])
        outs = [self.alloc((total,), "int64") for _ in range(x.ndim)]
        if total:
            self._launch("ahip_nonzero_write", (_VP(cnt.ptr), n, x.ndim, _i64arr(x.shape),
                                                (_VP * x.ndim)(*[o.ptr for o in outs]),
                                                self._stream()))
        return outs

    def _op_Nonzero(self, node, args):
        return self._nonzero(args[0])

    def _op_Default(self, node, args):
        """reference: tensor/basic.py:1819 Default.perform — x, or a copy of `default` for x = None."""
        if args[0] is None:
            return [self.materialize(self.to_device(args[1]))]
        return [args[0]]

    def _op_Searchsorted(self, node, args):
        """reference: tensor/extra_ops.py:144 SearchsortedOp.perform — np.searchsorted(x, v, side,
        sorter): one binary search per element of ``v`` (csrc/index.hip ahip_searchsorted)."""
        x, v = self.to_device(args[0]), self.contiguous(self.to_device(args[1]))
        if x.ndim != 1:
            raise ValueError("object too deep for desired array")
        if x.dtype != v.dtype:
            raise TypeError("Searchsorted operands must share one dtype")
        sorter = None
        if len(args) > 2:
            sorter = self.contiguous(self.cast(self.to_device(args[2]), "int64"))
            if sorter.ndim != 1 or sorter.shape[0] != x.shape[0]:
                raise ValueError("sorter.size must equal a.size")
        out = self.alloc(v.shape, "int64")
        if out.size:
            self._launch("ahip_searchsorted", (
                dtype_code(x.dtype), _VP(x.ptr), x.shape[0], x.strides[0] if x.shape[0] > 1 else 1,
                _VP(v.ptr), v.size, 1 if node.params["side"] == "right" else 0,
                _VP(sorter.ptr) if sorter is not None else None, _VP(out.ptr), self._stream()))
        return [out]

    def _op_HostCall(self, node, args):
        """An Op that IS a Python callable (``Print``'s print function, an ``as_op`` function:
        printing.py:863, compile/ops.py:258): operands are copied to the host behind the kernels
        that produce them, the callable runs, results (if any) are uploaded.  Never replayed: the
        host read makes the call take the eager path."""
        p = node.params
        host = [a if not isinstance(a, DevArray) else self.host_array(a) for a in args]
        res = p["fn"](*host)
        if p.get("view"):
            return [args[0]]
        outs = []
        for r, (dt, nd) in zip(res, p["otypes"]):
            r = np.asarray(r)
            if str(r.dtype) != dt or r.ndim != nd:
                raise TypeError(f"{p.get('what', 'HostCall')}: the function returned {r.dtype}[{r.ndim}-d], "
                                f"declared {dt}[{nd}-d]")
            outs.append(self._from_numpy(np.ascontiguousarray(r)))
        return outs

    def _adv_view(self, x, entries, extra):
        """Mixed advanced index (integer arrays + slices + newaxis): apply the basic entries as a
        view, move the dims addressed by arrays to the front (NumPy: tensor/subtensor.py:2607
        perform = ``x.__getitem__``).  Returns (front-permuted view, index arrays, number of
        leading result dims that precede the broadcast index block in NumPy's layout — 0 when
        the arrays are not adjacent and the block goes first)."""
        x = self.to_device(x)
        shape, strides, off = [], [], x.offset
        arrays, adv = [], []
        d = 0
        for e in entries:
            if "newaxis" in e:
                shape.append(1)
                strides.append(0)
                continue
            if d >= x.ndim:
                raise IndexError("too many indices for array")
            if "array" in e:
                arrays.append(extra[e["array"]])
                adv.append(len(shape))
                shape.append(x.shape[d])
                strides.append(x.strides[d])
            elif "mask" in e:
                m = self.to_device(extra[e["mask"]])
                if d + m.ndim > x.ndim or tuple(m.shape) != tuple(x.shape[d:d + m.ndim]):
                    raise IndexError("boolean index did not match indexed array along its dimensions")
                for q, ia in enumerate(self._nonzero(m)):
                    arrays.append(ia)
                    adv.append(len(shape))
                    shape.append(x.shape[d + q])
                    strides.append(x.strides[d + q])
                d += m.ndim
                continue
            else:
                comps = [c if (c is None or isinstance(c, int)) else self.host_int(extra[c["in"]])
                         for c in e["slice"]]
                start, stop, step = slice(*comps).indices(x.shape[d])
                n = len(range(start, stop, step))
                off += start * x.strides[d] if n > 0 else 0
                shape.append(n)
                strides.append(x.strides[d] * step)
            d += 1
        shape += list(x.shape[d:])
        strides += list(x.strides[d:])
        adjacent = adv == list(range(adv[0], adv[0] + len(adv)))
        rest = [q for q in range(len(shape)) if q not in adv]
        perm = adv + rest
        v = x.view([shape[q] for q in perm], [strides[q] for q in perm], off)
        return v, arrays, (adv[0] if adjacent else 0)

    @staticmethod
    def _block_after(nd, nb, lead):
        """dim order that moves the leading nb-dim index block behind the next `lead` dims"""
        return list(range(nb, nb + lead)) + list(range(nb)) + list(range(nb + lead, nd))

    def _op_AdvancedSubtensor(self, node, args):
        if "index" in node.params:
            v, arrays, lead = self._adv_view(args[0], node.params["index"], args[1:])
            x, lin, bshape, nrows, row, _keep = self._flat_rows(v, arrays)
            out = self.alloc(bshape + list(x.shape[len(arrays):]), x.dtype)
            if out.size:
                self._launch("ahip_take_rows", (dtype_code(x.dtype), _VP(x.ptr), nrows, row, row,
                                                _VP(lin.ptr), dtype_code("int64"), lin.shape[0], 1,
                                                _VP(out.ptr), row, _VP(self._bad_ptr()),
                                                self._stream()))
            if lead:
                order = self._block_after(out.ndim, len(bshape), lead)
                out = out.view([out.shape[q] for q in order], [out.strides[q] for q in order])
            return [out]
        x, lin, bshape, nrows, row, _keep = self._flat_rows(args[0], args[1:])
        out = self.alloc(bshape + list(x.shape[len(args) - 1:]), x.dtype)
        if out.size:
            self._launch("ahip_take_rows", (dtype_code(x.dtype), _VP(x.ptr), nrows, row, row,
                                            _VP(lin.ptr), dtype_code("int64"), lin.shape[0], 1,
                                            _VP(out.ptr), row, _VP(self._bad_ptr()),
                                            self._stream()))
        return [out]

    def _scatter(self, dst, nrows, row, idx_ptr, idx_dtype, nidx, idx_stride, src_ptr, src_rs,
                 accumulate):
        """x[idx] = y / x[idx] += y over rows.  Floating-point accumulation takes the ordered form
        (K9: contributions are added in index-list order like np.add.at; rows fed by more than
        64 entries use atomics), everything else — integers are exact under atomics — the plain one."""
        dt = dst.dtype
        if accumulate and dt in ("float32", "float64") and nidx > 1:
            need = int(lib.ahip_scatter_add_ws_bytes(nrows, nidx))
            if need:
                ws = self.alloc((need,), "uint8")
                self._launch("ahip_scatter_add_rows_ordered",
                             (dtype_code(dt), _VP(dst.ptr), nrows, row, row, _VP(idx_ptr),
                              dtype_code(idx_dtype), nidx, idx_stride, _VP(src_ptr), src_rs,
                              _VP(ws.ptr), need, _VP(self._bad_ptr()), self._stream()))
                return
        self._launch("ahip_scatter_rows", (dtype_code(dt), _VP(dst.ptr), nrows, row, row,
                                           _VP(idx_ptr), dtype_code(idx_dtype), nidx, idx_stride,
                                           _VP(src_ptr), src_rs, 1 if accumulate else 0,
                                           _VP(self._bad_ptr()), self._stream()))

    def _scatter_nodup(self, dst, nrows, row, lin, src):
        """``x[idx] += y`` with NumPy's buffered semantics (AdvancedIncSubtensor
        ``ignore_duplicates=True``, tensor/subtensor.py:2693): the indexed rows are READ first,
        ``y`` is added to the copies, the sums are SET in index order (the last duplicate wins, its
        sum started from the ORIGINAL row).  ``src``: contiguous [nidx, row] array, clobbered."""
        nidx = lin.shape[0]
        cur = self.alloc((nidx, row), dst.dtype)
        self._launch("ahip_take_rows", (dtype_code(dst.dtype), _VP(dst.ptr), nrows, row, row,
                                        _VP(lin.ptr), dtype_code("int64"), nidx, 1, _VP(cur.ptr), row,
                                        _VP(self._bad_ptr()), self._stream()))
        self.copy_into(cur, src.view((nidx, row), (row, 1)), accumulate=True)
        self._scatter(dst, nrows, row, lin.ptr, "int64", nidx, 1, cur.ptr, row, False)

    def _op_AdvancedIncSubtensor(self, node, args):
        x0 = self.to_device(args[0])
        out = self._own_or_copy(node.inputs[0], x0)
        if "index" in node.params:
            v, arrays, lead = self._adv_view(out, node.params["index"], args[2:])
            # the scatter kernel writes contiguous rows: work on a contiguous image of the
            # (front-permuted) view and copy it back when the view itself is not contiguous
            t = v if v.is_contiguous() else self.materialize(v)
            _x, lin, bshape, nrows, row, _keep = self._flat_rows(t, arrays)
            y = self.cast(self.to_device(args[1]), out.dtype)
            front = bshape + list(t.shape[len(arrays):])            # [index block | rest]
            nd, nb = len(front), len(bshape)
            order = self._block_after(nd, nb, lead) if lead else list(range(nd))
            numpy_shape = [front[q] for q in order]                 # layout y broadcasts against
            if y.ndim > nd:
                raise ValueError("shape mismatch: value array could not be broadcast to indexing result")
            full = self.alloc(numpy_shape, y.dtype)
            self.copy_into(full, y)
            if lead:
                inv = [order.index(q) for q in range(nd)]
                full = self.materialize(full.view([full.shape[q] for q in inv],
                                                  [full.strides[q] for q in inv]))
            if lin.shape[0] and row and node.params.get("ignore_duplicates"):
                self._scatter_nodup(t, nrows, row, lin, full)
            elif lin.shape[0] and row:
                self._scatter(t, nrows, row, lin.ptr, "int64", lin.shape[0], 1, full.ptr, row,
                              not node.params["set_instead_of_inc"])
            if t is not v:
                self.copy_into(v, t)
            return [out]
        _x, lin, bshape, nrows, row, _keep = self._flat_rows(out, args[2:])
        y = self.cast(self.to_device(args[1]), out.dtype)
        tgt = bshape + list(out.shape[len(args) - 2:])
        if y.ndim > len(tgt):
            raise ValueError("shape mismatch: value array could not be broadcast to indexing result")
        if list(y.shape) != tgt or not y.is_contiguous():
            full = self.alloc(tgt, y.dtype)
            self.copy_into(full, y)
            y = full
        if lin.shape[0] and row and node.params.get("ignore_duplicates"):
            self._scatter_nodup(out, nrows, row, lin, y)
        elif lin.shape[0] and row:
            self._scatter(out, nrows, row, lin.ptr, "int64", lin.shape[0], 1, y.ptr, row,
                          not node.params["set_instead_of_inc"])
        return [out]

    def _op_Split(self, node, args):
        """reference: tensor/basic.py:1929 Split.perform — contiguous copies of the slices."""
        x = self.to_device(args[0])
        axis = self.host_int(args[1])
        sp = args[2]
        splits = [int(v) for v in (self.host_array(sp) if isinstance(sp, DevArray)
                                   else np.asarray(sp)).reshape(-1)]
        if axis < -x.ndim or axis >= max(x.ndim, 1):
            raise IndexError(f"Split axis {axis} out of bounds")
        axis %= x.ndim
        if len(splits) != node.params["len_splits"]:
            raise ValueError("Length of `splits` is not equal to `len_splits`")
        if sum(splits) != x.shape[axis]:
            raise ValueError(f"The splits sum to {sum(splits)}; expected {x.shape[axis]}")
        if any(nb < 0 for nb in splits):
            raise ValueError("Attempted to make an array with a negative number of elements")
        outs, lo = [], 0
        for nb in splits:
            shape = list(x.shape)
            shape[axis] = nb
            v = x.view(shape, x.strides, x.offset + lo * x.strides[axis])
            outs.append(self.materialize(v))
            lo += nb
        return outs

    def _op_CumOp(self, node, args):
        """reference: tensor/extra_ops.py:311 CumOp.perform (np.cumsum / np.cumprod; axis=None
        scans the flattened array)."""
        x = self.to_device(args[0])
        axis = node.params["axis"]
        dt = self.plan.vars[node.outputs[0]].dtype
        if x.dtype != dt:
            x = self.cast(x, dt)
        if axis is None:
            x = self.contiguous(x)
            outer, n, inner, so, sn, si = 1, x.size, 1, 0, 1, 0
            oshape = [x.size]
        else:
            if axis < -x.ndim or axis >= x.ndim:
                raise ValueError(f"axis(={axis}) out of bounds")
            axis %= x.ndim
            osh, ost = collapse_dims(list(x.shape[:axis]) or [1], [list(x.strides[:axis]) or [0]])
            ish, ist = collapse_dims(list(x.shape[axis + 1:]) or [1],
                                     [list(x.strides[axis + 1:]) or [0]])
            if len(osh) != 1 or len(ish) != 1:
                x = self.materialize(x)
                osh, ost = [_prod((x.shape[:axis]))], [[x.strides[axis - 1] if axis else 0]]
                ish, ist = [_prod((x.shape[axis + 1:]))], [[1]]
                if axis:
                    ost = [[x.shape[axis] * ish[0]]]
            outer, inner, so, si = osh[0], ish[0], ost[0][0], ist[0][0]
            n, sn = x.shape[axis], x.strides[axis]
            oshape = list(x.shape)
        out = self.alloc(oshape, dt)
        if out.size:
            need = int(lib.ahip_cumulative_ws_bytes(dtype_code(dt), outer, n, inner))
            ws = self.alloc((need,), "uint8") if need else None
            self._launch("ahip_cumulative", (dtype_code(dt), 1 if node.params["mode"] == "mul" else 0,
                                             _VP(x.ptr), outer, n, inner, so, sn, si, _VP(out.ptr),
                                             _VP(ws.ptr) if ws is not None else None, need,
                                             self._stream()))
        return [out]

    def _op_AdvancedSubtensor1(self, node, args):
        x, idx = self.to_device(args[0]), self.to_device(args[1])
        if self._rows(x) is None:
            x = self.materialize(x)
        rs, row = self._rows(x)
        out = self.alloc((idx.shape[0],) + x.shape[1:], x.dtype)
        self._launch("ahip_take_rows", (dtype_code(x.dtype), _VP(x.ptr), x.shape[0], rs, row,
                                 _VP(idx.ptr), dtype_code(idx.dtype), idx.shape[0],
                                 idx.strides[0] if idx.shape[0] != 1 else 1, _VP(out.ptr), row,
                                 _VP(self._bad_ptr()), self._stream()))
        return [out]

    def _op_AdvancedIncSubtensor1(self, node, args):
        x, y, idx = (self.to_device(a) for a in args)
        out = self._own_or_copy(node.inputs[0], x)
        y = self.cast(y, out.dtype)
        row = 1
        for s in out.shape[1:]:
            row *= s
        # y broadcasts against x[idx] == (len(idx),) + x.shape[1:]
        tgt = (idx.shape[0],) + out.shape[1:]
        if y.shape != tgt or self._rows(y) is None:
            full = self.alloc(tgt, y.dtype)
            self.copy_into(full, y)
            y = full
        rs_y, _ = self._rows(y)
        self._scatter(out, out.shape[0], row, idx.ptr, idx.dtype, idx.shape[0],
                      idx.strides[0] if idx.shape[0] != 1 else 1, y.ptr, rs_y,
                      not node.params["set_instead_of_inc"])
        return [out]
