"""Lower a rewritten Aesara ``FunctionGraph`` to a :class:`aesara_amd.plan.Plan`.

This is the per-Op "dispatch" layer of the HIP linker — the analogue of ``jax_funcify`` /
``numba_funcify`` (reference link/jax/dispatch/basic.py:38-58), except that an Op is turned
into a *plan node* (plain data) rather than a Python callable, so the device executor can run
it through the C-ABI without any Aesara object.  Aesara is imported lazily: this module is
only used where the reference front end is installed.

Every handler cites the reference Op it restates.
"""
from __future__ import annotations

from functools import singledispatch

import numpy as np

from .plan import Node, Plan

# scalar Op class name (reference aesara/scalar/basic.py, scalar/math.py) -> plan scalar op name
SCALAR_OP_NAMES = {
    "Add": "add", "Mul": "mul", "Sub": "sub", "TrueDivide": "true_div",
    "FloorDivide": "int_div", "Mod": "mod", "Pow": "pow", "Neg": "neg", "Abs": "abs",
    "Sgn": "sgn", "Sqr": "sqr", "Sqrt": "sqrt", "Exp": "exp", "Exp2": "exp2",
    "Expm1": "expm1", "Log": "log", "Log2": "log2", "Log10": "log10", "Log1p": "log1p",
    "Sin": "sin", "Cos": "cos", "Tan": "tan", "ArcSin": "arcsin", "ArcCos": "arccos",
    "ArcTan": "arctan", "ArcTan2": "arctan2", "Sinh": "sinh", "Cosh": "cosh", "Tanh": "tanh",
    "ArcSinh": "arcsinh", "ArcCosh": "arccosh", "ArcTanh": "arctanh",
    "Ceil": "ceil", "Floor": "floor", "Trunc": "trunc",
    "RoundHalfToEven": "round_half_to_even",
    "RoundHalfAwayFromZero": "round_half_away_from_zero",
    "Reciprocal": "reciprocal", "Identity": "identity", "Cast": "cast", "Second": "second",
    "Switch": "switch", "Clip": "clip", "ScalarMaximum": "maximum",
    "ScalarMinimum": "minimum", "LT": "lt", "GT": "gt", "LE": "le", "GE": "ge", "EQ": "eq",
    "NEQ": "neq", "AND": "and", "OR": "or", "XOR": "xor", "Invert": "invert",
    "IsNan": "isnan", "IsInf": "isinf", "Sigmoid": "sigmoid", "Softplus": "softplus",
    "Erf": "erf", "Erfc": "erfc", "Log1mexp": "log1mexp", "Deg2Rad": "deg2rad",
    "Rad2Deg": "rad2deg",
    # scalar/math.py special functions (unary, floating point)
    "Erfcx": "erfcx", "Erfinv": "erfinv", "Erfcinv": "erfcinv", "Gamma": "gamma",
    "GammaLn": "gammaln", "Psi": "psi", "TriGamma": "tri_gamma", "J0": "j0", "J1": "j1",
    "I0": "i0", "I1": "i1",
    # tensor/math.py:2713 MulWithoutZeros (the CAReduce inside ProdWithoutZeros: grad of prod)
    "MulWithoutZeros": "mul_without_zeros",
    "ScalarSoftsign": "softsign",      # tensor/nnet/basic.py:2040: x / (1 + |x|)
    # tensor/xlogx.py:7 XlogX (x == 0 ? 0 : x * log(x)), :36 XlogY0 (x == 0 ? 0 : x * log(y))
    "XlogX": "xlogx", "XlogY0": "xlogy0",
    # tensor/nnet/sigm.py:20 UltraFastScalarSigmoid (the rewrite `local_ultra_fast_sigmoid` puts it in place of Sigmoid)
    "UltraFastScalarSigmoid": "ultra_fast_sigmoid",
    # scalar/math.py: the incomplete-gamma family that has C code in the reference (c_code/gamma.c)
    "GammaInc": "gammainc", "GammaIncC": "gammaincc", "Chi2SF": "chi2sf", "GammaU": "gammau",
    "GammaL": "gammal",
}


_PLAN_DTYPES = {"float32", "float64", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32",
                "uint64", "bool"}


class UnsupportedOp(NotImplementedError):
    """Raised for Ops outside the hot path: the linker reports them loudly."""


def lower_scalar_op(scalar_op, n_in):
    """``ScalarOp`` / ``Composite`` -> scalar expression dict (see plan.py docstring).

    Reference: ``Composite.fgraph`` scalar/basic.py:4128, ``c_code_template`` :4250 (walks the
    inner scalar graph in toposort order assigning one temporary per scalar Apply).
    """
    from aesara.scalar.basic import Composite, ScalarConstant

    cls = type(scalar_op).__name__
    if isinstance(scalar_op, Composite):
        fg = scalar_op.fgraph
        refs = {v: ["i", k] for k, v in enumerate(fg.inputs)}
        nodes = []
        for sn in fg.toposort():
            ins = []
            for v in sn.inputs:
                if v in refs:
                    ins.append(refs[v])
                elif isinstance(v, ScalarConstant):
                    ins.append(["c", _pyscalar(v.data), str(v.type.dtype)])
                else:  # pragma: no cover
                    raise UnsupportedOp(f"dangling scalar variable {v}")
            sub = lower_scalar_op(sn.op, len(sn.inputs))
            if len(sub["nodes"]) != 1:
                # nested Composite: splice it in
                base = len(nodes)
                for n2 in sub["nodes"]:
                    nodes.append({"op": n2["op"], "dtype": n2["dtype"],
                                  "in": [_rebase(r, ins, base) for r in n2["in"]]})
                for v, r in zip(sn.outputs, sub["out"]):
                    refs[v] = _rebase(r, ins, base)
                continue
            for k, v in enumerate(sn.outputs):
                if k > 0:
                    raise UnsupportedOp("multi-output scalar op inside Composite")
                nodes.append({"op": sub["nodes"][0]["op"], "in": ins,
                              "dtype": str(v.type.dtype)})
                refs[v] = ["t", len(nodes) - 1]
        outs = []
        for v in fg.outputs:
            if v in refs:
                outs.append(refs[v])
            elif isinstance(v, ScalarConstant):
                outs.append(["c", _pyscalar(v.data), str(v.type.dtype)])
            else:  # pragma: no cover
                raise UnsupportedOp("composite output not computed")
        return {"n_in": n_in, "nodes": nodes, "out": outs}
    if cls not in SCALAR_OP_NAMES:
        raise UnsupportedOp(f"scalar op {cls} is outside the HIP hot path")
    name = SCALAR_OP_NAMES[cls]
    # dtype filled in by the caller (needs the Apply's output type)
    return {"n_in": n_in,
            "nodes": [{"op": name, "in": [["i", k] for k in range(n_in)], "dtype": None}],
            "out": [["t", 0]]}


def _rebase(r, ins, base):
    if r[0] == "i":
        return ins[r[1]]
    if r[0] == "t":
        return ["t", r[1] + base]
    return r


def _pyscalar(x):
    x = np.asarray(x)
    if x.dtype.kind == "b":
        return bool(x)
    if x.dtype.kind in "iu":
        return int(x)
    v = float(x)
    return v


# --------------------------------------------------------------------------------------
# per-Op lowering (singledispatch on the Op class, like jax_funcify)
# --------------------------------------------------------------------------------------
@singledispatch
def hip_lower(op, node, ctx):
    raise UnsupportedOp(
        f"{type(op).__name__} has no HIP lowering (outside the hot path of SURVEY §8a)")


class _Ctx:
    def __init__(self, plan, inner_rewriter=None):
        self.plan = plan
        self.vmap = {}
        self.slices = {}     # SliceType variable -> [start, stop, step] variables (MakeSlice)
        self.inner_rewriter = inner_rewriter

    def vid(self, v):
        from aesara.graph.basic import Constant

        if v in self.vmap:
            return self.vmap[v]
        if isinstance(v, Constant):
            if hasattr(v.type, "format") and type(v.type).__name__.startswith("Sparse"):
                raise UnsupportedOp(f"non-tensor variable type Sparse ({v.type})")
            data = np.asarray(v.data)
            vid = self.plan.add_const(data, dtype=v.type.dtype, name=None)
            # keep static broadcast pattern of the constant's type
            self.plan.vars[vid].shape = _static_shape(v.type)
            self.vmap[v] = vid
            return vid
        if v in self.slices or type(v.type).__name__ == "SliceType":
            # a MakeSlice result used as anything but the index of a Subtensor-family node (a graph output)
            raise UnsupportedOp(f"non-tensor variable type SliceType ({v}): graph boundaries of a plan are tensors")
        raise KeyError(f"variable {v} not produced by any lowered node")

    def new(self, v):
        t = v.type
        if not hasattr(t, "dtype"):
            raise UnsupportedOp(f"non-tensor variable type {t}")
        if hasattr(t, "format") and type(t).__name__.startswith("Sparse"):
            # (SparseTensorType subclasses TensorType: a scipy.sparse matrix is not a dense buffer)
            raise UnsupportedOp(f"non-tensor variable type Sparse ({t})")
        if str(t.dtype) not in _PLAN_DTYPES:
            raise UnsupportedOp(f"dtype {t.dtype} of {v} has no HIP kernels (SURVEY §8a H2: "
                                "float32/64, int8-64, uint8-64, bool)")
        shape = _static_shape(t)
        vid = self.plan.new_var(t.dtype, shape, getattr(v, "name", None))
        self.vmap[v] = vid
        return vid

    def raw(self, opname, in_vids, dtype, shape, params=None):
        """Append a node on plan variables directly (decompositions); returns the output id."""
        out = self.plan.new_var(dtype, list(shape), None)
        self.plan.nodes.append(Node(opname, list(in_vids), [out], params or {}))
        return out

    def emit(self, opname, node, params=None, inputs=None):
        ins = [self.vid(i) for i in (node.inputs if inputs is None else inputs)]
        outs = [self.new(o) for o in node.outputs]
        self.plan.nodes.append(Node(opname, ins, outs, params or {}))


def _static_shape(t):
    if hasattr(t, "shape") and t.shape is not None and hasattr(t, "ndim"):
        return [None if s is None else int(s) for s in t.shape]
    return []  # aesara.scalar ScalarType (0-d)


def lower_fgraph(fgraph, order=None, name="fgraph", inner_rewriter=None, extra_outputs=()) -> Plan:
    """FunctionGraph -> Plan.  ``order`` is the linker's schedule (``Linker.schedule``,
    link/basic.py:222); defaults to ``fgraph.toposort()``.  ``inner_rewriter`` is applied to
    a clone of every Scan inner graph (the reference rewrites it lazily when ``Scan.fn`` is
    compiled with ``mode_instance``, scan/op.py:1431-1459)."""
    _register_handlers()
    plan = Plan(name, {}, [], [], [])
    ctx = _Ctx(plan, inner_rewriter)
    for v in fgraph.inputs:
        plan.inputs.append(ctx.new(v))
    # plan variable -> position (in ``order``) of the Apply node it was lowered from: how the
    # linker maps a failing / timed executor step back to the graph (raise_with_op, fn.profile).
    # A plain attribute, not part of the serialised plan.
    origin = {}
    for k, node in enumerate(order if order is not None else fgraph.toposort()):
        n0 = len(plan.nodes)
        hip_lower(node.op, node, ctx)
        for pn in plan.nodes[n0:]:
            for o in pn.outputs:
                origin[o] = k
        for o in node.outputs:
            if o in ctx.vmap:
                origin.setdefault(ctx.vmap[o], k)
    # ``extra_outputs``: variables the linker needs next to the graph's outputs (the final content
    # of an input buffer a destructive Op overwrites: linker.HipLinker write-back)
    plan.outputs = [ctx.vid(o) for o in fgraph.outputs] + [ctx.vid(o) for o in extra_outputs]
    plan.var_origin = origin
    return plan


_registered = False


def _register_handlers():
    """Register lowering handlers (deferred: needs Aesara importable)."""
    global _registered
    if _registered:
        return
    _registered = True

    from aesara.compile.ops import DeepCopyOp, ViewOp
    from aesara.scan.op import Scan
    from aesara.tensor.basic import (Alloc, AllocEmpty, Join, MakeVector, ScalarFromTensor,
                                     TensorFromScalar)
    from aesara.tensor.blas import BatchedDot, Dot22, Dot22Scalar, Gemm, Gemv, Ger
    from aesara.tensor.elemwise import CAReduce, DimShuffle, Elemwise
    from aesara.tensor.math import Dot
    from aesara.tensor.shape import Reshape, Shape, Shape_i, SpecifyShape, Unbroadcast
    from aesara.tensor.subtensor import (AdvancedIncSubtensor1, AdvancedSubtensor1,
                                         IncSubtensor, Subtensor)

    def _natural_dtype(scalar_op, node):
        """Result dtype of a ``*_inplace`` scalar op's OWN arithmetic: its output type is
        transferred from an input (``transfer_type(0)``, scalar/basic.py:4435 ff.), but the C body
        ``z = x + y`` computes in the promoted type of the operands and converts on the store —
        ``add_inplace(int64, float64)`` truncates the float sum.  None: nothing to separate."""
        import aesara.scalar as aes
        pref = getattr(scalar_op, "output_types_preference", None)
        if not isinstance(pref, aes.basic.transfer_type):
            return None
        base = None
        for mod in (aes.basic, aes.math):
            for cand in vars(mod).values():
                if type(cand) is type(scalar_op) and not isinstance(
                        getattr(cand, "output_types_preference", None), aes.basic.transfer_type):
                    base = cand
                    break
            if base is not None:
                break
        if base is None:
            return None
        try:
            nat = base.output_types([aes.get_scalar_type(str(i.type.dtype)) for i in node.inputs])[0].dtype
        except Exception:                                   # noqa: BLE001
            return None
        return str(nat) if str(nat) in _PLAN_DTYPES else None

    @hip_lower.register(Elemwise)
    def _(op, node, ctx):
        # reference: tensor/elemwise.py:304 Elemwise (perform :725, _c_all :835)
        s = lower_scalar_op(op.scalar_op, len(node.inputs))
        if len(s["nodes"]) == 1 and s["nodes"][0]["dtype"] is None:
            odt = str(node.outputs[0].type.dtype)
            nat = _natural_dtype(op.scalar_op, node)
            if nat is not None and nat != odt:
                s["nodes"][0]["dtype"] = nat
                s["nodes"].append({"op": "cast", "in": [["t", 0]], "dtype": odt})
                s["out"] = [["t", 1]]
            else:
                s["nodes"][0]["dtype"] = odt
        if len(s["out"]) != len(node.outputs):
            raise UnsupportedOp("Elemwise output arity mismatch")
        ctx.emit("Elemwise", node, {"scalar": s})

    from aesara.scalar.basic import ScalarOp

    @hip_lower.register(ScalarOp)
    def _(op, node, ctx):
        # a ScalarOp applied directly to 0-d ScalarType variables (index arithmetic around
        # Subtensor/Scan, scalar/basic.py:1082): same expression, 0-d operands, host-evaluated
        s = lower_scalar_op(op, len(node.inputs))
        if len(s["nodes"]) == 1 and s["nodes"][0]["dtype"] is None:
            s["nodes"][0]["dtype"] = str(node.outputs[0].type.dtype)
        ctx.emit("Elemwise", node, {"scalar": s})

    @hip_lower.register(CAReduce)
    def _(op, node, ctx):
        # reference: tensor/elemwise.py:1221 CAReduce; acc rule _acc_dtype :1371; perform :1495
        sname = SCALAR_OP_NAMES.get(type(op.scalar_op).__name__)
        if sname not in ("add", "mul", "maximum", "minimum", "and", "or", "xor", "mul_without_zeros"):
            raise UnsupportedOp(f"CAReduce over scalar op {op.scalar_op}")
        idtype = node.inputs[0].type.dtype
        odtype = node.outputs[0].type.dtype
        if hasattr(op, "_acc_dtype"):
            acc = op._acc_dtype(idtype)
        else:
            acc = getattr(op, "acc_dtype", None) or odtype
        axis = op.axis
        if axis is not None:
            nd = node.inputs[0].type.ndim
            axis = sorted(int(a) % nd if nd else int(a) for a in axis)
        ctx.emit("CAReduce", node, {"scalar_op": sname, "axis": axis, "acc_dtype": str(acc)})

    @hip_lower.register(DimShuffle)
    def _(op, node, ctx):
        # reference: tensor/elemwise.py:39 DimShuffle (view; perform :222)
        ctx.emit("DimShuffle", node, {"new_order": [x if x == "x" else int(x)
                                                    for x in op.new_order]})

    @hip_lower.register(Dot)
    def _(op, node, ctx):
        # reference: tensor/math.py:1879 Dot (1-d/2-d only; make_node :1903 upcasts the output dtype,
        # perform :1929 = np.dot, which converts mixed operands to the common type first).  The
        # kernels take ONE dtype: operands are cast to the node's output dtype (exact for every
        # pair NumPy promotes: the common type holds both), then float32 / float64 products run on
        # the MFMA kernels and bool / integer products on the vector-ALU kernel (csrc/igemm.hip,
        # NumPy's wrap-around arithmetic bit for bit).  float16 / complex: refused at compile time.
        dts = [str(i.type.dtype) for i in node.inputs]
        odt = str(node.outputs[0].type.dtype)
        _ok = ("float32", "float64", "bool", "int8", "int16", "int32", "int64", "uint8", "uint16",
               "uint32", "uint64")
        if any(d not in _ok for d in dts + [odt]):
            raise UnsupportedOp(f"Dot over {dts} -> {odt}: float16 / complex products are not on the HIP "
                                "path (SURVEY §2)")
        ins = []
        for i, d in zip(node.inputs, dts):
            vid = ctx.vid(i)
            if d != odt:
                vid = ctx.raw("Elemwise", [vid], odt, _static_shape(i.type), {"scalar": {
                    "n_in": 1, "nodes": [{"op": "cast", "in": [["i", 0]], "dtype": odt}], "out": [["t", 0]]}})
            ins.append(vid)
        outs = [ctx.new(o) for o in node.outputs]
        ctx.plan.nodes.append(Node("Dot", ins, outs, {}))

    @hip_lower.register(Dot22)
    def _(op, node, ctx):
        # reference: tensor/blas.py:1659 Dot22
        ctx.emit("Dot22", node)

    @hip_lower.register(Dot22Scalar)
    def _(op, node, ctx):
        # reference: tensor/blas.py:1954 Dot22Scalar
        ctx.emit("Dot22Scalar", node)

    @hip_lower.register(Gemm)
    def _(op, node, ctx):
        # reference: tensor/blas.py:872 Gemm: inputs (z, a, x, y, b) -> b*z + a*dot(x, y)
        ctx.emit("Gemm", node, {"inplace": bool(op.inplace)})

    @hip_lower.register(Gemv)
    def _(op, node, ctx):
        # reference: tensor/blas.py:231 Gemv: inputs (y, alpha, A, x, beta) -> beta*y + alpha*A.x
        ctx.emit("Gemv", node, {"inplace": bool(op.inplace)})

    @hip_lower.register(Ger)
    def _(op, node, ctx):
        # reference: tensor/blas.py:330 Ger: inputs (A, alpha, x, y) -> A + alpha*outer(x, y)
        ctx.emit("Ger", node, {"destructive": bool(op.destructive)})

    @hip_lower.register(BatchedDot)
    def _(op, node, ctx):
        # reference: tensor/blas.py:2179 BatchedDot (make_node :2196 upcasts the output dtype;
        # perform :2224 z[i] = np.dot(x[i], y[i])): operands cast to the output dtype like Dot
        dts = [str(i.type.dtype) for i in node.inputs]
        odt = str(node.outputs[0].type.dtype)
        if all(d == odt for d in dts):
            ctx.emit("BatchedDot", node)
            return
        _ok = ("float32", "float64", "bool", "int8", "int16", "int32", "int64", "uint8", "uint16",
               "uint32", "uint64")
        if any(d not in _ok for d in dts + [odt]):
            raise UnsupportedOp(f"BatchedDot over {dts} -> {odt}")
        ins = []
        for i, d in zip(node.inputs, dts):
            vid = ctx.vid(i)
            if d != odt:
                vid = ctx.raw("Elemwise", [vid], odt, _static_shape(i.type), {"scalar": {
                    "n_in": 1, "nodes": [{"op": "cast", "in": [["i", 0]], "dtype": odt}], "out": [["t", 0]]}})
            ins.append(vid)
        outs = [ctx.new(o) for o in node.outputs]
        ctx.plan.nodes.append(Node("BatchedDot", ins, outs, {}))

    @hip_lower.register(Alloc)
    def _(op, node, ctx):
        # reference: tensor/basic.py:1389 Alloc(value, *shape)
        ctx.emit("Alloc", node)

    from aesara.tensor.extra_ops import BroadcastTo

    @hip_lower.register(BroadcastTo)
    def _(op, node, ctx):
        # reference: tensor/extra_ops.py BroadcastTo(x, *shape) — a broadcast (stride-0) view
        ctx.emit("BroadcastTo", node)

    @hip_lower.register(AllocEmpty)
    def _(op, node, ctx):
        # reference: tensor/basic.py:3833 AllocEmpty(*shape)
        ctx.emit("AllocEmpty", node, {"dtype": str(op.dtype)})

    @hip_lower.register(MakeVector)
    def _(op, node, ctx):
        # reference: tensor/basic.py:1629 MakeVector
        ctx.emit("MakeVector", node, {"dtype": str(op.dtype)})

    @hip_lower.register(Join)
    def _(op, node, ctx):
        # reference: tensor/basic.py:2142 Join(axis, *tensors)
        ctx.emit("Join", node)

    @hip_lower.register(ScalarFromTensor)
    def _(op, node, ctx):
        ctx.emit("ScalarFromTensor", node)

    @hip_lower.register(TensorFromScalar)
    def _(op, node, ctx):
        ctx.emit("TensorFromScalar", node)

    @hip_lower.register(Shape_i)
    def _(op, node, ctx):
        # reference: tensor/shape.py:189 Shape_i
        ctx.emit("Shape_i", node, {"i": int(op.i)})

    @hip_lower.register(Shape)
    def _(op, node, ctx):
        ctx.emit("Shape", node)

    @hip_lower.register(Reshape)
    def _(op, node, ctx):
        # reference: tensor/shape.py:589 Reshape(x, shape)
        ctx.emit("Reshape", node, {"ndim": int(op.ndim)})

    @hip_lower.register(Unbroadcast)
    def _(op, node, ctx):
        # reference: tensor/shape.py:939 Unbroadcast (type-level no-op view)
        ctx.emit("ViewOp", node)

    @hip_lower.register(SpecifyShape)
    def _(op, node, ctx):
        # reference: tensor/shape.py:376 SpecifyShape(x, *shape) (perform :439); unknown dims
        # are NoneConst inputs and are dropped here ("dims" lists the checked positions)
        from aesara.graph.basic import Constant
        dims, ins = [], [node.inputs[0]]
        for d, sv in enumerate(node.inputs[1:]):
            if isinstance(sv, Constant) and sv.data is None:
                continue
            dims.append(d)
            ins.append(sv)
        ctx.emit("SpecifyShape", node, {"dims": dims, "ndim": len(node.inputs) - 1}, inputs=ins)

    @hip_lower.register(ViewOp)
    def _(op, node, ctx):
        ctx.emit("ViewOp", node)

    from aesara.raise_op import CheckAndRaise

    @hip_lower.register(CheckAndRaise)
    def _(op, node, ctx):
        # reference: raise_op.py:28 CheckAndRaise / Assert (perform :94): a view of its first
        # input that raises ``exc_type(msg)`` unless every 0-d condition is true
        ctx.emit("Assert", node, {"msg": str(op.msg), "exc_type": op.exc_type.__name__})

    @hip_lower.register(DeepCopyOp)
    def _(op, node, ctx):
        # reference: compile/ops.py:149 DeepCopyOp (inserted by insert_deepcopy types.py:1172)
        ctx.emit("DeepCopyOp", node)

    def _idx_list(idx_list):
        # reference: tensor/subtensor.py:682 Subtensor.idx_list — slices whose entries are
        # either None, python ints, or scalar *Types* standing for a dynamic input.
        out = []
        for e in idx_list:
            if isinstance(e, slice):
                out.append({"slice": [_idx_entry(e.start), _idx_entry(e.stop),
                                      _idx_entry(e.step)]})
            else:
                out.append({"index": _idx_entry(e)})
        return out

    def _idx_entry(e):
        if e is None:
            return None
        if isinstance(e, (int, np.integer)):
            return int(e)
        return "in"  # dynamic: consumes the next extra input

    @hip_lower.register(Subtensor)
    def _(op, node, ctx):
        ctx.emit("Subtensor", node, {"idx_list": _idx_list(op.idx_list)})

    @hip_lower.register(IncSubtensor)
    def _(op, node, ctx):
        # reference: tensor/subtensor.py:1454 IncSubtensor(x, y, *idx)
        ctx.emit("IncSubtensor", node, {
            "idx_list": _idx_list(op.idx_list),
            "set_instead_of_inc": bool(op.set_instead_of_inc),
            "inplace": bool(op.inplace)})

    @hip_lower.register(AdvancedSubtensor1)
    def _(op, node, ctx):
        # reference: tensor/subtensor.py:1925 AdvancedSubtensor1(x, ilist)
        ctx.emit("AdvancedSubtensor1", node)

    @hip_lower.register(AdvancedIncSubtensor1)
    def _(op, node, ctx):
        # reference: tensor/subtensor.py:2128 AdvancedIncSubtensor1(x, y, ilist)
        ctx.emit("AdvancedIncSubtensor1", node, {
            "set_instead_of_inc": bool(op.set_instead_of_inc), "inplace": bool(op.inplace)})

    # ---- Softmax family: decomposed into the plan's own primitives (max / exp-sum / normalise);
    # the fusion pass merges them into row kernels.  Accumulation in fp64 for fp32 inputs like
    # the reference's C loops (tensor/special.py:393 ``double sum_exp_dev``).
    from aesara.tensor.special import LogSoftmax, Softmax, SoftmaxGrad

    def _sx(n_in, nodes, out):
        return {"n_in": n_in, "out": [out],
                "nodes": [{"op": o, "dtype": d, "in": i} for o, d, i in nodes]}

    def _row_axes(op, x):
        nd = x.type.ndim
        axes = list(range(nd)) if op.axis is None else [int(op.axis) % nd]
        shape = _static_shape(x.type)
        red = [sh for d, sh in enumerate(shape) if d not in axes]
        keep_order, j = [], 0
        for d in range(nd):
            if d in axes:
                keep_order.append("x")
            else:
                keep_order.append(j)
                j += 1
        kshape = [1 if d in axes else sh for d, sh in enumerate(shape)]
        return axes, shape, red, kshape, keep_order

    def _row_reduce(ctx, vid, sop, axes, red, kshape, order, dtype, acc):
        r = ctx.raw("CAReduce", [vid], dtype, red,
                    {"scalar_op": sop, "axis": axes, "acc_dtype": acc})
        return ctx.raw("DimShuffle", [r], dtype, kshape, {"new_order": order})

    def _float_only(op, x):
        if x.type.dtype not in ("float32", "float64"):
            raise UnsupportedOp(f"{type(op).__name__} on dtype {x.type.dtype}")
        return x.type.dtype

    @hip_lower.register(Softmax)
    def _(op, node, ctx):
        # reference: tensor/special.py:239 Softmax (perform :268 scipy.special.softmax; C :372-415)
        x = node.inputs[0]
        dt = _float_only(op, x)
        axes, shape, red, kshape, order = _row_axes(op, x)
        xv = ctx.vid(x)
        m = _row_reduce(ctx, xv, "maximum", axes, red, kshape, order, dt, dt)
        e = ctx.raw("Elemwise", [xv, m], dt, shape, {"scalar": _sx(
            2, [("sub", dt, [["i", 0], ["i", 1]]), ("exp", dt, [["t", 0]])], ["t", 1])})
        ssum = _row_reduce(ctx, e, "add", axes, red, kshape, order, "float64", "float64")
        out = ctx.raw("Elemwise", [e, ssum], dt, shape, {"scalar": _sx(
            2, [("reciprocal", "float64", [["i", 1]]), ("mul", "float64", [["i", 0], ["t", 0]]),
                ("cast", dt, [["t", 1]])], ["t", 2])})
        ctx.vmap[node.outputs[0]] = out

    @hip_lower.register(LogSoftmax)
    def _(op, node, ctx):
        # reference: tensor/special.py:508 LogSoftmax (perform :537; C :640-740):
        # xdev = x - max(x); out = xdev - log(sum(exp(xdev)))
        x = node.inputs[0]
        dt = _float_only(op, x)
        axes, shape, red, kshape, order = _row_axes(op, x)
        xv = ctx.vid(x)
        m = _row_reduce(ctx, xv, "maximum", axes, red, kshape, order, dt, dt)
        e = ctx.raw("Elemwise", [xv, m], dt, shape, {"scalar": _sx(
            2, [("sub", dt, [["i", 0], ["i", 1]]), ("exp", dt, [["t", 0]])], ["t", 1])})
        ssum = _row_reduce(ctx, e, "add", axes, red, kshape, order, "float64", "float64")
        out = ctx.raw("Elemwise", [xv, m, ssum], dt, shape, {"scalar": _sx(
            3, [("sub", dt, [["i", 0], ["i", 1]]), ("log", "float64", [["i", 2]]),
                ("cast", dt, [["t", 1]]), ("sub", dt, [["t", 0], ["t", 2]])], ["t", 3])})
        ctx.vmap[node.outputs[0]] = out

    @hip_lower.register(SoftmaxGrad)
    def _(op, node, ctx):
        # reference: tensor/special.py:26 SoftmaxGrad (perform :39):
        # dx = dy*sm - sum(dy*sm, axis, keepdims) * sm
        dy, sm = node.inputs
        dt = _float_only(op, node.outputs[0])
        if dy.type.dtype != dt or sm.type.dtype != dt:
            raise UnsupportedOp("SoftmaxGrad with mixed dtypes")
        axes, shape, red, kshape, order = _row_axes(op, node.outputs[0])
        dv, sv = ctx.vid(dy), ctx.vid(sm)
        p = ctx.raw("Elemwise", [dv, sv], dt, shape, {"scalar": _sx(
            2, [("mul", dt, [["i", 0], ["i", 1]])], ["t", 0])})
        ssum = _row_reduce(ctx, p, "add", axes, red, kshape, order, "float64", "float64")
        out = ctx.raw("Elemwise", [p, ssum, sv], dt, shape, {"scalar": _sx(
            3, [("cast", dt, [["i", 1]]), ("mul", dt, [["t", 0], ["i", 2]]),
                ("sub", dt, [["i", 0], ["t", 1]])], ["t", 2])})
        ctx.vmap[node.outputs[0]] = out

    from aesara.tensor.math import Argmax, MaxAndArgmax

    def _argmax_axes(op, x):
        nd = x.type.ndim
        if op.axis is None:
            return list(range(nd))
        return sorted(int(a) % nd for a in op.axis)

    @hip_lower.register(Argmax)
    def _(op, node, ctx):
        # reference: tensor/math.py:330 Argmax (perform :388)
        ctx.emit("Argmax", node, {"axis": _argmax_axes(op, node.inputs[0])})

    @hip_lower.register(MaxAndArgmax)
    def _(op, node, ctx):
        # reference: tensor/math.py:126 MaxAndArgmax (perform :163): both outputs of one pass
        x = node.inputs[0]
        axes = _argmax_axes(op, x)
        xv = ctx.vid(x)
        red = [sh for d, sh in enumerate(_static_shape(x.type)) if d not in axes]
        mx = ctx.raw("CAReduce", [xv], x.type.dtype, red,
                     {"scalar_op": "maximum", "axis": axes, "acc_dtype": x.type.dtype})
        am = ctx.raw("Argmax", [xv], "int64", red, {"axis": axes})
        ctx.vmap[node.outputs[0]] = mx
        ctx.vmap[node.outputs[1]] = am

    from aesara.tensor.basic import ARange
    from aesara.tensor.subtensor import AdvancedIncSubtensor, AdvancedSubtensor

    @hip_lower.register(ARange)
    def _(op, node, ctx):
        # reference: tensor/basic.py:2867 ARange(start, stop, step) (perform :2937)
        ctx.emit("ARange", node, {"dtype": str(op.dtype)})

    def _int_array_indices(op, idx_vars):
        from aesara.tensor.type import TensorType
        for v in idx_vars:
            if not (isinstance(v.type, TensorType) and v.type.dtype.startswith(("int", "uint"))):
                raise UnsupportedOp(f"{type(op).__name__} with a non-integer-array index "
                                    f"({v.type}): slices / masks / newaxis are outside the path")
        if not 1 <= len(idx_vars) <= 8:
            raise UnsupportedOp("advanced indexing with more than 8 index arrays")

    from aesara.graph.basic import Constant as _Constant
    from aesara.tensor.type_other import MakeSlice, NoneTypeT, SliceType

    @hip_lower.register(MakeSlice)
    def _(op, node, ctx):
        # reference: tensor/type_other.py:27 MakeSlice(start, stop, step): the bounds of a slice
        # inside an advanced index.  No plan node: the consumer reads the recorded components.
        ctx.slices[node.outputs[0]] = [None if isinstance(v.type, NoneTypeT) else v
                                       for v in node.inputs]

    def _adv_index(op, idx_vars, ctx):
        """Entries of an advanced index mixing integer arrays with slices / newaxis ->
        (entries, extra input variables).  {"array": k} | {"slice": [c, c, c]} with c = None |
        int | {"in": k} (k = position among the extra inputs) | {"newaxis": true}."""
        from aesara.tensor.type import TensorType
        entries, ins = [], []
        for v in idx_vars:
            if isinstance(v.type, TensorType) and v.type.dtype.startswith(("int", "uint")):
                entries.append({"array": len(ins)})
                ins.append(v)
            elif isinstance(v.type, SliceType):
                if v in ctx.slices:
                    comps = ctx.slices[v]
                elif isinstance(v, _Constant):
                    comps = [v.data.start, v.data.stop, v.data.step]
                else:
                    raise UnsupportedOp(f"{type(op).__name__}: slice of unknown origin")
                enc = []
                for c in comps:
                    if c is None:
                        enc.append(None)
                    elif not hasattr(c, "owner"):          # python / NumPy integer of a constant slice
                        enc.append(int(np.asarray(c).reshape(())))
                    else:
                        enc.append({"in": len(ins)})
                        ins.append(c)
                entries.append({"slice": enc})
            elif isinstance(v.type, NoneTypeT):
                entries.append({"newaxis": True})
            elif isinstance(v.type, TensorType) and v.type.dtype == "bool" and v.type.ndim >= 1:
                # a boolean mask stands for the index arrays of its non-zero entries (NumPy;
                # subtensor.py:2543): expanded at run time through the Nonzero kernels
                entries.append({"mask": len(ins), "ndim": int(v.type.ndim)})
                ins.append(v)
            else:
                raise UnsupportedOp(f"{type(op).__name__} with a {v.type} index")
        n_arr = sum(e.get("ndim", 1) for e in entries if "array" in e or "mask" in e)
        if not 1 <= n_arr <= 8:
            raise UnsupportedOp("advanced indexing with more than 8 index arrays")
        return entries, ins

    def _only_arrays(idx_vars):
        from aesara.tensor.type import TensorType
        return all(isinstance(v.type, TensorType) and v.type.dtype != "bool" for v in idx_vars)

    @hip_lower.register(AdvancedSubtensor)
    def _(op, node, ctx):
        # reference: tensor/subtensor.py:2543 AdvancedSubtensor (perform :2607): integer arrays,
        # optionally mixed with slices / newaxis (NumPy placement rules)
        if _only_arrays(node.inputs[1:]):
            _int_array_indices(op, node.inputs[1:])
            ctx.emit("AdvancedSubtensor", node)
            return
        entries, ins = _adv_index(op, node.inputs[1:], ctx)
        ctx.emit("AdvancedSubtensor", node, {"index": entries}, inputs=[node.inputs[0]] + ins)

    @hip_lower.register(AdvancedIncSubtensor)
    def _(op, node, ctx):
        # reference: tensor/subtensor.py:2647 AdvancedIncSubtensor (perform :2688: np.add.at / set)
        # ignore_duplicates (inc only; :2693): ``out[idx] += y`` = read, add, sequential set
        params = {"set_instead_of_inc": bool(op.set_instead_of_inc), "inplace": bool(op.inplace)}
        if getattr(op, "ignore_duplicates", False) and not op.set_instead_of_inc:
            params["ignore_duplicates"] = True
        if _only_arrays(node.inputs[2:]):
            _int_array_indices(op, node.inputs[2:])
            ctx.emit("AdvancedIncSubtensor", node, params)
            return
        entries, ins = _adv_index(op, node.inputs[2:], ctx)
        ctx.emit("AdvancedIncSubtensor", node, dict(params, index=entries),
                 inputs=list(node.inputs[:2]) + ins)

    from aesara.tensor.basic import AllocDiag, ExtractDiag, Eye, Tri

    @hip_lower.register(Eye)
    def _(op, node, ctx):
        # reference: tensor/basic.py:1257 Eye(n, m, k) (perform :1278 np.eye)
        ctx.emit("Eye", node, {"dtype": np.dtype(op.dtype).name})

    @hip_lower.register(Tri)
    def _(op, node, ctx):
        # reference: tensor/basic.py:982 Tri(N, M, k) (perform :1000 np.tri)
        ctx.emit("Tri", node, {"dtype": np.dtype(op.dtype).name})    # (``dtype=bool`` the class)

    @hip_lower.register(ExtractDiag)
    def _(op, node, ctx):
        # reference: tensor/basic.py:3327 ExtractDiag (perform :3402 x.diagonal(offset, axis1, axis2))
        ctx.emit("ExtractDiag", node, {"offset": int(op.offset), "axis1": int(op.axis1),
                                       "axis2": int(op.axis2), "view": bool(op.view)})

    @hip_lower.register(AllocDiag)
    def _(op, node, ctx):
        # reference: tensor/basic.py:3487 AllocDiag (perform :3523)
        ctx.emit("AllocDiag", node, {"offset": int(op.offset), "axis1": int(op.axis1),
                                     "axis2": int(op.axis2)})

    from aesara.tensor.nnet.basic import (CrossentropyCategorical1Hot,
                                          CrossentropyCategorical1HotGrad)

    def _row_pick(ctx, coding, idx):
        """coding[arange(N), idx] on plan variables -> (gathered vid, arange vid, N vid, K vid)"""
        n = ctx.raw("Shape_i", [coding], "int64", [], {"i": 0})
        k = ctx.raw("Shape_i", [coding], "int64", [], {"i": 1})
        zero, one = ctx.plan.add_const(0, "int64"), ctx.plan.add_const(1, "int64")
        ar = ctx.raw("ARange", [zero, n, one], "int64", [None], {"dtype": "int64"})
        dt = ctx.plan.vars[coding].dtype
        picked = ctx.raw("AdvancedSubtensor", [coding, ar, idx], dt, [None])
        return picked, ar, n, k

    @hip_lower.register(CrossentropyCategorical1Hot)
    def _(op, node, ctx):
        # reference: tensor/nnet/basic.py:940 CrossentropyCategorical1Hot (perform :990):
        # y[i] = -log(coding_dist[i, true_one_of_n[i]]) — a row gather + one Elemwise
        coding, idx = (ctx.vid(v) for v in node.inputs)
        dt = node.outputs[0].type.dtype
        picked, _, _, _ = _row_pick(ctx, coding, idx)
        sc = {"n_in": 1, "nodes": [{"op": "log", "in": [["i", 0]], "dtype": dt},
                                   {"op": "neg", "in": [["t", 0]], "dtype": dt}], "out": [["t", 1]]}
        ctx.vmap[node.outputs[0]] = ctx.raw("Elemwise", [picked], dt, [None], {"scalar": sc})

    @hip_lower.register(CrossentropyCategorical1HotGrad)
    def _(op, node, ctx):
        # reference: tensor/nnet/basic.py:905 CrossentropyCategorical1HotGrad (perform :928):
        # zeros_like(coding) with g[i, idx[i]] = -g_y[i] / coding[i, idx[i]]
        g_y, coding, idx = (ctx.vid(v) for v in node.inputs)
        dt = node.outputs[0].type.dtype
        picked, ar, n, k = _row_pick(ctx, coding, idx)
        sc = {"n_in": 2, "nodes": [{"op": "neg", "in": [["i", 0]], "dtype": dt},
                                   {"op": "true_div", "in": [["t", 0], ["i", 1]], "dtype": dt}],
              "out": [["t", 1]]}
        val = ctx.raw("Elemwise", [g_y, picked], dt, [None], {"scalar": sc})
        zeros = ctx.raw("Alloc", [ctx.plan.add_const(0, dt), n, k], dt, [None, None])
        ctx.vmap[node.outputs[0]] = ctx.raw(
            "AdvancedIncSubtensor", [zeros, val, ar, idx], dt, [None, None],
            {"set_instead_of_inc": True, "inplace": False})

    from aesara.tensor.sort import ArgSortOp, SortOp

    def _sort_common(op, node, ctx, name):
        # reference: tensor/sort.py:29 SortOp / :150 ArgSortOp (x, axis); structured-array
        # `order` is meaningless for plain tensors
        if getattr(op, "order", None) is not None:
            raise UnsupportedOp(f"{name} with a field order")
        ctx.emit(name, node, {"kind": str(op.kind)})

    @hip_lower.register(SortOp)
    def _(op, node, ctx):
        _sort_common(op, node, ctx, "Sort")

    @hip_lower.register(ArgSortOp)
    def _(op, node, ctx):
        _sort_common(op, node, ctx, "ArgSort")

    from aesara.tensor.basic import Nonzero

    @hip_lower.register(Nonzero)
    def _(op, node, ctx):
        # reference: tensor/basic.py:845 Nonzero (perform :870 np.nonzero): ndim int64 vectors
        # whose length is known only at run time (one device->host read of the count)
        if not 1 <= node.inputs[0].type.ndim <= 8:
            raise UnsupportedOp("Nonzero of a 0-d or > 8-d array")
        ctx.emit("Nonzero", node)

    from aesara.tensor.extra_ops import FillDiagonal

    @hip_lower.register(FillDiagonal)
    def _(op, node, ctx):
        # reference: tensor/extra_ops.py:870 FillDiagonal(a, val) (perform :906): a copy of `a` with
        # `val` on the main diagonal (rectangular 2-d allowed; n-d needs equal extents)
        ctx.emit("FillDiagonal", node)

    from aesara.tensor.math import MatMul

    @hip_lower.register(MatMul)
    def _(op, node, ctx):
        # reference: tensor/math.py:2871 MatMul (perform :2941 np.matmul): stacks of matrices in
        # the last two dims, batch dims broadcast, 1-d operands promoted
        dts = {v.type.dtype for v in node.inputs} | {node.outputs[0].type.dtype}
        if len(dts) != 1 or dts.pop() not in ("float32", "float64", "bool", "int8", "int16", "int32",
                                              "int64", "uint8", "uint16", "uint32", "uint64"):
            raise UnsupportedOp("MatMul is lowered for operands of ONE float32 / float64 / integer dtype")
        ctx.emit("MatMul", node)

    from aesara.ifelse import IfElse

    @hip_lower.register(IfElse)
    def _(op, node, ctx):
        # reference: ifelse.py:61 IfElse(cond, *then, *else) — a lazy Op: the VM computes only the
        # branch the condition selects (make_thunk ifelse.py:240, lazy scheduling link/vm.py:430
        # Stack.__call__ / lazylinker_c.c:544 lazy_rec_eval).  The executor keeps that: steps
        # that only feed one branch run after the condition is known.
        ctx.emit("IfElse", node, {"n_outs": int(op.n_outs)})

    from aesara.tensor.basic import Split
    from aesara.tensor.extra_ops import CumOp

    @hip_lower.register(Split)
    def _(op, node, ctx):
        # reference: tensor/basic.py:1882 Split(x, axis, splits) (perform :1929)
        ctx.emit("Split", node, {"len_splits": int(op.len_splits)})

    @hip_lower.register(CumOp)
    def _(op, node, ctx):
        # reference: tensor/extra_ops.py:283 CumOp (perform :311)
        ctx.emit("CumOp", node, {"axis": None if op.axis is None else int(op.axis),
                                 "mode": str(op.mode)})

    from . import lower_more
    lower_more.register(hip_lower, _static_shape, UnsupportedOp)

    @hip_lower.register(Scan)
    def _(op, node, ctx):
        # reference: scan/op.py:637 Scan; info layout scan/op.py:206 ScanInfo.  The inner
        # fgraph is lowered recursively; the executor owns the step loop (K10).  All output
        # kinds are carried: mit-mot (gradient accumulators of Scan.L_op, op.py:2379), mit-sot,
        # sit-sot, nit-sot, shared (scan updates) and the do-while condition (scan/utils.py until).
        info = op.info
        inner_fg = op.fgraph.clone()
        if ctx.inner_rewriter is not None:
            ctx.inner_rewriter.rewrite(inner_fg)
        inner = lower_fgraph(inner_fg, name="scan_inner", inner_rewriter=ctx.inner_rewriter)
        ctx.emit("Scan", node, {
            "n_seqs": info.n_seqs,
            "mit_mot_in_slices": [list(map(int, t)) for t in info.mit_mot_in_slices],
            "mit_mot_out_slices": [list(map(int, t)) for t in info.mit_mot_out_slices],
            "mit_sot_in_slices": [list(map(int, t)) for t in info.mit_sot_in_slices],
            "sit_sot_in_slices": [list(map(int, t)) for t in info.sit_sot_in_slices],
            "n_nit_sot": info.n_nit_sot,
            "n_shared_outs": info.n_shared_outs,
            "n_non_seqs": info.n_non_seqs,
            "as_while": bool(info.as_while),
            "inner": inner,
        })
