"""BLAS-class steps of a PlanExecutor: Gemm / Dot22 / Dot22Scalar / BatchedDot / MatMul on the MFMA
GEMM, Gemv / Ger, integer products, the small-M GEMM chain with epilogue (reference:
tensor/blas.py:231/330/872/1659/1954/2179, tensor/math.py:1879/2871).

Part of :class:`aesara_amd.executor.PlanExecutor` (a mixin: the methods run on the executor's
state; split out of executor.py in round 4, no behaviour change)."""
from __future__ import annotations

from .exec_common import *  # noqa: F401,F403
from .exec_common import (_I64, _VP, _i64arr, _Kernels, _FakeBuf, _CAST_SCALARS, _prod, _Arena, _os, _time)  # noqa: F401


class BlasMixin:

    def _run_gemm_epi(self, st: Step, env, out_targets=None) -> bool:
        """Small-M GEMM chain + Elemwise epilogue in one kernel (ahip_gemm_epilogue).  Returns
        False when the shapes are large enough for the 128x128 GEMM (whose alpha/beta epilogue the
        original nodes use) or the layouts do not qualify; the caller runs the original steps."""
        from ._lib import GeArgs
        dots = [(self.to_device(env[a]), self.to_device(env[b])) for a, b in st.dots]
        if any(A.ndim != 2 or B.ndim != 2 for A, B in dots):
            return False
        M, N = dots[0][0].shape[0], dots[0][1].shape[1]
        dt = dots[0][0].dtype
        if dt not in ("float32", "float64") or M < 1 or N < 1:
            return False
        # regime of the small-output kernels (csrc/gemm.hip gemm_dispatch)
        if -(-M // 128) * -(-N // 128) >= 256:
            return False
        vec = 4 if dt == "float32" else 2
        isz = ITEMSIZE[dt]
        bkc = []
        for A, B in dots:
            K = A.shape[1]
            if (A.dtype != dt or B.dtype != dt or A.shape[0] != M or B.shape[1] != N
                    or B.shape[0] != K or K < 1):
                return False
            a_rs = A.strides[0] if M != 1 else K
            if not ((A.strides[1] == 1 or K == 1) and a_rs % vec == 0 and K % vec == 0
                    and A.ptr % 16 == 0):
                return False
            if B.strides[0] == 1 and B.strides[1] % vec == 0 and B.ptr % 16 == 0 and K > 1:
                bkc.append(True)
            elif B.strides[1] == 1 or N == 1:
                bkc.append(False)
            else:
                return False
        others = [self.to_device(env[v]) for v in st.inputs]
        ostr = []
        for o in others:
            if o.ndim != 2 or o.shape[0] not in (1, M) or o.shape[1] not in (1, N):
                return False
            ostr.append((0 if o.shape[0] == 1 and M != 1 else o.strides[0],
                         0 if o.shape[1] == 1 and N != 1 else o.strides[1]))
        # column fragments per workgroup: widest that still gives every CU a tile; n-contiguous B
        # operands need N, their row stride and base aligned to it; mixed B layouts force 1
        nf = vec
        while nf > 1 and -(-M // 16) * -(-N // (16 * nf)) < 256:
            nf //= 2
        if any(bkc) and not all(bkc):
            nf = 1
        for (A, B), k in zip(dots, bkc):
            if not k:
                while nf > 1 and (N % nf or B.strides[0] % nf or B.ptr % (nf * isz)):
                    nf //= 2
        out_vars = [self.plan.vars[o] for o in st.outputs]
        outs = [self._out_buffer(o, (M, N), ov.dtype, out_targets)
                for o, ov in zip(st.outputs, out_vars)]
        for o, arr in zip(st.outputs, outs):
            env[o] = arr
        # K slices per workgroup: these kernels are bound by memory round trips per wavefront
        kmax = max(A.shape[1] for A, _ in dots)
        waves = int(knobs.get("GE_WAVES")) or \
            (16 if kmax >= 1024 else (8 if kmax >= 512 else 4))
        if waves * len(dots) * 256 * nf * isz > 60 * 1024:     # LDS partials
            waves = 4
        spec = cg.GemmEpiSpec(dt, nf, bkc, st.scalar, [o.dtype for o in others],
                              [o.dtype for o in outs], st.out_refs, waves=waves)
        key = spec.key()
        ent = _Kernels.cache.get(key) if not self.dry_run else \
            ([None] if key in _Kernels.compiled else None)
        if ent is None:
            src, names = cg.generate_gemm_epilogue(spec)
            if self.dry_run:
                from .device import compile_cached
                compile_cached(src)
                _Kernels.compiled[key] = 1
                ent = [None]
            else:
                ent = load_kernels(src, names)
                _Kernels.cache[key] = ent
        g = GeArgs()
        g.M, g.N = M, N
        for d, (A, B) in enumerate(dots):
            g.K[d] = A.shape[1]
            g.A[d], g.a_rs[d] = A.ptr, (A.strides[0] if M != 1 else A.shape[1])
            g.B[d], g.b_rs[d], g.b_cs[d] = B.ptr, B.strides[0], B.strides[1]
        for k, (o, (rs, cs)) in enumerate(zip(others, ostr)):
            g.ptr[k], g.rs[k], g.cs[k] = o.ptr, rs, cs
        for k, o in enumerate(outs):
            j = len(others) + k
            g.ptr[j], g.rs[j], g.cs[j] = o.ptr, o.strides[0], o.strides[1]
        self._launch("ahip_gemm_epilogue", (ent[0], C.byref(g), nf, waves, self._stream()))
        return True

    # ------------------------------------------------------------------ BLAS ----------
    def _scalar_arg(self, v, dtype):
        val = self.host_scalar(v)
        return C.c_float(val) if dtype == "float32" else C.c_double(val)

    _INT_DOT = ("bool", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64")

    def _check_blas_dtype(self, *arrs, ints=False):
        dt = arrs[0].dtype
        ok = ("float32", "float64") + (self._INT_DOT if ints else ())
        if dt not in ok or any(a.dtype != dt for a in arrs):
            # (the linker casts the operands of a Dot / BatchedDot to the node's output dtype,
            # lower.py; a hand-built plan that gets here has a dtype error)
            raise TypeError("BLAS operands must share one float32 / float64 dtype (Dot, BatchedDot, "
                            "MatMul: or one integer / bool dtype), got " + ", ".join(a.dtype for a in arrs))
        return dt

    def _igemm(self, A, B, batch):
        """Integer / bool product on the vector ALU (csrc/igemm.hip): NumPy's wrap-around
        arithmetic bit for bit (Dot.perform = np.dot, tensor/math.py:1929)."""
        dt = A.dtype
        if batch:
            (nb, M, K), (nb2, K2, N) = A.shape, B.shape
            if nb != nb2:
                raise ValueError("batch sizes do not match")
        else:
            (M, K), (K2, N) = A.shape, B.shape
            nb = 1
        if K != K2:
            raise ValueError(f"Shape mismatch: A.shape[1] != B.shape[0] ({K} vs {K2})")
        out = self.alloc((nb, M, N) if batch else (M, N), dt)

        def st(a):
            s_ = [s if n != 1 else 0 for s, n in zip(a.strides, a.shape)]
            return s_ if batch else [0] + s_
        sa, sb, so = st(A), st(B), (list(out.strides) if batch else [0] + list(out.strides))
        self._launch("ahip_igemm_batched", (dtype_code(dt), nb, M, N, K, _VP(A.ptr), sa[0], sa[1], sa[2],
                                            _VP(B.ptr), sb[0], sb[1], sb[2], _VP(out.ptr), so[0], so[1],
                                            so[2], self._stream()))
        return out

    def _gemm(self, alpha, A, B, beta, Cin, batch=False):
        if A.dtype in self._INT_DOT:
            self._check_blas_dtype(A, B, ints=True)
            if alpha != 1 or (Cin is not None and beta != 0):
                raise TypeError("integer products take no alpha / beta (Gemm is float-only in the reference)")
            return self._igemm(A, B, batch)
        dt = self._check_blas_dtype(A, B)
        if batch:
            nb, M, K = A.shape
            nb2, K2, N = B.shape
            if nb != nb2:
                raise ValueError("batch sizes do not match")
        else:
            (M, K), (K2, N) = A.shape, B.shape
            nb = 1
        if K != K2:
            raise ValueError(f"Shape mismatch: A.shape[1] != B.shape[0] ({K} vs {K2})")
        out = self.alloc((nb, M, N) if batch else (M, N), dt)
        a_ = C.c_float(alpha) if dt == "float32" else C.c_double(alpha)
        b_ = C.c_float(beta) if dt == "float32" else C.c_double(beta)
        if Cin is not None and beta != 0:
            if tuple(Cin.shape[-2:]) != (M, N):
                # z may broadcast in the reference only via explicit DimShuffle; be strict
                raise ValueError("Gemm: z has the wrong shape")
            cin_ptr, ci = _VP(Cin.ptr), Cin.strides
        else:
            cin_ptr, ci = _VP(out.ptr), out.strides

        def st(a):  # strides with size-1 dims neutralised
            return [s if n != 1 else 0 for s, n in zip(a.strides, a.shape)]

        sa, sb, so = st(A), st(B), out.strides
        if batch:
            sci = list(ci)
            self._launch("ahip_gemm_batched", (dtype_code(dt), nb, M, N, K, C.byref(a_), _VP(A.ptr),
                                        sa[0], sa[1], sa[2], _VP(B.ptr), sb[0], sb[1], sb[2],
                                        C.byref(b_), cin_ptr, sci[0], sci[1], sci[2],
                                        _VP(out.ptr), so[0], so[1], so[2], self._stream()))
        else:
            wsb = int(lib.ahip_gemm_ws_bytes(dtype_code(dt), 1, M, N, K))
            if wsb:
                # few output tiles, very long K: K slices into a workspace, summed in order
                ws = self.alloc((wsb // ITEMSIZE[dt],), dt)
                self._launch("ahip_gemm_splitk", (
                    dtype_code(dt), M, N, K, C.byref(a_), _VP(A.ptr), sa[0], sa[1], _VP(B.ptr),
                    sb[0], sb[1], C.byref(b_), cin_ptr, ci[0], ci[1], _VP(out.ptr), so[0], so[1],
                    _VP(ws.ptr), wsb, self._stream()))
                return out
            self._launch("ahip_gemm", (dtype_code(dt), M, N, K, C.byref(a_), _VP(A.ptr), sa[0], sa[1],
                                _VP(B.ptr), sb[0], sb[1], C.byref(b_), cin_ptr, ci[0], ci[1],
                                _VP(out.ptr), so[0], so[1], self._stream()))
        return out

    def _op_MatMul(self, node, args):
        """reference: tensor/math.py:2941 MatMul.perform (np.matmul): 1-d operands are promoted
        to a row / a column, batch dims broadcast; one strided-batched GEMM (K6) — a broadcast
        batch dim is a zero batch stride, no copy."""
        a, b = self.to_device(args[0]), self.to_device(args[1])
        self._check_blas_dtype(a, b, ints=True)
        va, vb = a.ndim == 1, b.ndim == 1
        if va:
            a = a.view((1,) + tuple(a.shape), (0,) + tuple(a.strides))
        if vb:
            b = b.view(tuple(b.shape) + (1,), tuple(b.strides) + (0,))
        (M, K), (K2, N) = a.shape[-2:], b.shape[-2:]
        if K != K2:
            raise ValueError("matmul: Input operand 1 has a mismatch in its core dimension 0 "
                             f"(size {K2} is different from {K})")
        try:
            bshape = list(np.broadcast_shapes(tuple(a.shape[:-2]), tuple(b.shape[:-2])))
        except ValueError:
            raise ValueError("matmul: operands could not be broadcast together")
        nbd = len(bshape)

        def bstrides(x):
            pad = nbd - (x.ndim - 2)
            return [0] * pad + [0 if n == 1 and bn != 1 else s
                                for n, bn, s in zip(x.shape[:-2], bshape[pad:], x.strides[:-2])]
        sa, sb = bstrides(a), bstrides(b)
        nb = _prod(bshape) if bshape else 1
        csh, cst = collapse_dims(bshape or [1], [sa or [0], sb or [0]])
        if len(csh) != 1:
            # batch dims that do not fold into one stride: materialise the broadcast operands
            a = self.materialize(a.view(bshape + [M, K], sa + list(a.strides[-2:])))
            b = self.materialize(b.view(bshape + [K, N], sb + list(b.strides[-2:])))
            cst = [[M * K], [K * N]]
        a3 = a.view((nb, M, K), (cst[0][0],) + tuple(a.strides[-2:]))
        b3 = b.view((nb, K, N), (cst[1][0],) + tuple(b.strides[-2:]))
        out = self._gemm(1.0, a3, b3, 0.0, None, batch=True)
        shape = bshape + ([] if va else [M]) + ([] if vb else [N])
        return [out.view(shape, contiguous_strides(shape))]

    def _op_Gemm(self, node, args):
        z, a, x, y, b = args
        z, x, y = self.to_device(z), self.to_device(x), self.to_device(y)
        self._check_blas_dtype(z, x, y)
        if x.ndim != 2 or y.ndim != 2 or z.ndim != 2:
            raise ValueError("Gemm operands must be matrices")
        if z.shape != (x.shape[0], y.shape[1]):
            # Gemm.perform (tensor/blas.py:995-1016): z is broadcast UP to dot(x, y)'s shape, and
            # ``z += a * dot(x, y)`` broadcasts the product up to z's (infer_shape :1018: the
            # maximum per dim) — extent-1 dims become zero strides of the operands, no copy
            M, N, K = max(x.shape[0], z.shape[0]), max(y.shape[1], z.shape[1]), x.shape[1]

            def up(t, shape, what):
                if any(n != m and n != 1 for n, m in zip(t.shape, shape)):
                    raise ValueError(f"Gemm: {what} of shape {tuple(t.shape)} does not broadcast to {shape}")
                return t.view(shape, tuple(0 if n != m else s_ for n, m, s_ in zip(t.shape, shape, t.strides)))
            z, x, y = up(z, (M, N), "z"), up(x, (M, K), "x"), up(y, (y.shape[0], N), "y")
        return [self._gemm(self.host_scalar(a), x, y, self.host_scalar(b), z)]

    def _op_Dot22(self, node, args):
        x, y = self.to_device(args[0]), self.to_device(args[1])
        return [self._gemm(1.0, x, y, 0.0, None)]

    def _op_Dot22Scalar(self, node, args):
        x, y = self.to_device(args[0]), self.to_device(args[1])
        return [self._gemm(self.host_scalar(args[2]), x, y, 0.0, None)]

    def _op_BatchedDot(self, node, args):
        x, y = self.to_device(args[0]), self.to_device(args[1])
        # 2-d operands are a batch of vectors (tensor/blas.py:2196-2207, perform :2224:
        # z[i] = np.dot(x[i], y[i])): promoted to [b, 1, k] / [b, k, 1] views, the unit dims
        # dropped from the result again
        if x.ndim not in (2, 3) or y.ndim not in (2, 3):
            raise TypeError("BatchedDot operands must have 2 or 3 dimensions")
        if x.shape[0] != y.shape[0]:
            raise TypeError(f"Shape mismatch: x has {x.shape[0]} rows but y has {y.shape[0]} rows")
        x2, y2 = x.ndim == 2, y.ndim == 2
        if x2:
            x = x.view((x.shape[0], 1, x.shape[1]), (x.strides[0], 0, x.strides[1]))
        if y2:
            y = y.view((y.shape[0], y.shape[1], 1), (y.strides[0], y.strides[1], 0))
        z = self._gemm(1.0, x, y, 0.0, None, batch=True)
        if x2 or y2:
            keep = [0] + ([] if x2 else [1]) + ([] if y2 else [2])
            z = z.view(tuple(z.shape[d] for d in keep), tuple(z.strides[d] for d in keep))
        return [z]

    def _gemv(self, alpha, A, x, beta, y):
        if A.dtype in self._INT_DOT:
            # integer matrix-vector product: the integer GEMM with a one-column right operand
            self._check_blas_dtype(A, x, ints=True)
            if alpha != 1 or (y is not None and beta != 0):
                raise TypeError("integer products take no alpha / beta (Gemv is float-only in the reference)")
            if x.shape[0] != A.shape[1]:
                raise ValueError(f"Incompatible shapes for gemv: A {A.shape}, x {x.shape}")
            r = self._igemm(A, x.view((x.shape[0], 1), (x.strides[0], 0)), False)
            return r.view((A.shape[0],), (r.strides[0],))
        dt = self._check_blas_dtype(A, x)
        M, N = A.shape
        if x.shape[0] != N:
            raise ValueError(f"Incompatible shapes for gemv: A {A.shape}, x {x.shape}")
        out = self.alloc((M,), dt)
        a_ = C.c_float(alpha) if dt == "float32" else C.c_double(alpha)
        b_ = C.c_float(beta) if dt == "float32" else C.c_double(beta)
        if y is not None and beta != 0:
            if y.shape != (M,):
                raise ValueError("Incompatible shapes for gemv (y)")
            yin, incy = _VP(y.ptr), (y.strides[0] if M != 1 else 1)
        else:
            yin, incy = _VP(out.ptr), 1
        ars = A.strides[0] if M != 1 else 0
        acs = A.strides[1] if N != 1 else 1
        ws_bytes = int(lib.ahip_gemv_ws_bytes(dtype_code(dt), M, N))
        ws = self.alloc((max(ws_bytes // ITEMSIZE[dt], 1),), dt)
        self._launch("ahip_gemv", (dtype_code(dt), M, N, C.byref(a_), _VP(A.ptr), ars, acs, _VP(x.ptr),
                            x.strides[0] if N != 1 else 1, C.byref(b_), yin, incy, _VP(out.ptr), 1,
                            _VP(ws.ptr), ws_bytes, self._stream()))
        return out

    def _op_Gemv(self, node, args):
        y, alpha, A, x, beta = args
        A, x = self.to_device(A), self.to_device(x)
        b = self.host_scalar(beta)
        return [self._gemv(self.host_scalar(alpha), A, x, b,
                           self.to_device(y) if b != 0 else None)]

    def _op_Ger(self, node, args):
        A, alpha, x, y = args
        A, x, y = self.to_device(A), self.to_device(x), self.to_device(y)
        dt = self._check_blas_dtype(A, x, y)
        M, N = A.shape
        if x.shape[0] != M or y.shape[0] != N:
            raise ValueError("Ger: shape mismatch")
        out = self.alloc((M, N), dt)
        a_ = self._scalar_arg(alpha, dt)
        self._launch("ahip_ger", (dtype_code(dt), M, N, C.byref(a_), _VP(x.ptr), x.strides[0],
                           _VP(y.ptr), y.strides[0], _VP(A.ptr), A.strides[0], A.strides[1],
                           _VP(out.ptr), out.strides[0], out.strides[1], self._stream()))
        return [out]

    def _op_Dot(self, node, args):
        # reference: tensor/math.py:1879 Dot (vector/matrix combinations, no BLAS rewrite)
        x, y = self.to_device(args[0]), self.to_device(args[1])
        if x.ndim == 2 and y.ndim == 2:
            return [self._gemm(1.0, x, y, 0.0, None)]
        if x.ndim == 2 and y.ndim == 1:
            return [self._gemv(1.0, x, y, 0.0, None)]
        if x.ndim == 1 and y.ndim == 2:
            yt = y.view((y.shape[1], y.shape[0]), (y.strides[1], y.strides[0]))
            return [self._gemv(1.0, yt, x, 0.0, None)]
        if x.ndim == 1 and y.ndim == 1:
            xm = x.view((1, x.shape[0]), (0, x.strides[0]))
            r = self._gemv(1.0, xm, y, 0.0, None)
            return [r.view((), ())]
        raise NotImplementedError("Dot: unsupported operand ranks")
