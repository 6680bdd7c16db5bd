"""HIP source generator for the fused broadcast Elemwise / CAReduce kernels (K1, K2, K3).

Plays the role of the reference's per-Op C generators — ``Elemwise._c_all``
(tensor/elemwise.py:835), ``elemwise_cgen.make_loop`` / ``make_reordered_loop`` (:228/:305),
``CAReduce._c_all`` (:1522) and ``Composite.c_code_template`` (scalar/basic.py:4250) — but emits
one gfx950 kernel per fused group instead of a CPU loop nest:

* one lane handles ``VEC`` consecutive elements of the innermost (collapsed) dimension with
  16-byte global loads; outer dimensions are index arithmetic over at most ``AHIP_MAXD``
  collapsed dims; a grid-stride loop covers the rest (HBM-bound streaming shape);
* broadcast operands (stride 0 on the inner dim) are loaded once per lane, not per element;
* a CAReduce consumer is fused into the same kernel: per-lane accumulation in the reference's
  accumulator dtype (``CAReduce._acc_dtype`` :1371), wavefront reduction with cross-lane
  shuffles, one partial per workgroup, and a deterministic fixed-order finalize.

The scalar bodies restate the ``c_code`` of the reference ScalarOps (scalar/basic.py,
scalar/math.py) for the supported dtypes (bool, (u)int8-64, float32/64).
"""
from __future__ import annotations

import hashlib

import numpy as np

from ._lib import AHIP_MAXD, AHIP_MAXOPS

CTYPE = {
    "bool": "unsigned char", "int8": "signed char", "int16": "short", "int32": "int",
    "int64": "long long", "uint8": "unsigned char", "uint16": "unsigned short",
    "uint32": "unsigned int", "uint64": "unsigned long long", "float32": "float",
    "float64": "double",
}
# type used for values held in registers (bool is a real C++ bool there)
RTYPE = dict(CTYPE, bool="bool")

PRELUDE = r"""
typedef long long i64;
#ifndef NAN
#define NAN __builtin_nanf("")
#endif
#ifndef INFINITY
#define INFINITY __builtin_huge_valf()
#endif
#define AHIP_MAXD %d
#define AHIP_MAXOPS %d
struct Args {
  i64 n; i64 shape[AHIP_MAXD]; i64 stride[AHIP_MAXOPS][AHIP_MAXD]; void* ptr[AHIP_MAXOPS];
  void* ws; void* out; i64 aux0; i64 aux1; int nd; int nops;
};
template <typename T, int N> struct alignas((sizeof(T) * N) >= 16 ? 16 : (sizeof(T) * N)) Pack { T v[N]; };

// ---- scalar helpers (reference: aesara/scalar/basic.py, scalar/math.py c_code) ----
template <typename T> __device__ __forceinline__ T idiv_floor(T x, T y) {  // FloorDivide :2039
  if (y == 0) return 0;
  T q = x / y;
  if ((x %% y != 0) && ((x < 0) != (y < 0))) q -= 1;
  return q;
}
template <typename T> __device__ __forceinline__ T udiv_floor(T x, T y) { return y == 0 ? 0 : x / y; }
template <typename T> __device__ __forceinline__ T imod_py(T x, T y) {      // Mod :2144
  if (y == 0) return 0;
  T r = x %% y;
  if (r != 0 && ((r < 0) != (y < 0))) r += y;
  return r;
}
template <typename T> __device__ __forceinline__ T umod(T x, T y) { return y == 0 ? 0 : x %% y; }
__device__ __forceinline__ float fmod_py(float x, float y) {
  if (y == 0.0f) return fmodf(x, y);
  float r = fmodf(x, y);
  if (r != 0.0f && ((r < 0.0f) != (y < 0.0f))) r += y;
  return r;
}
__device__ __forceinline__ double fmod_py(double x, double y) {
  if (y == 0.0) return fmod(x, y);
  double r = fmod(x, y);
  if (r != 0.0 && ((r < 0.0) != (y < 0.0))) r += y;
  return r;
}
template <typename T> __device__ __forceinline__ T ipow(T b, T e) {
  T r = 1;
  if (e < 0) return (b == 1) ? 1 : ((b == (T)-1) ? ((e & 1) ? (T)-1 : 1) : 0);
  while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
  return r;
}
template <typename T> __device__ __forceinline__ T upow(T b, T e) {
  T r = 1;
  while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
  return r;
}
// ScalarMaximum/ScalarMinimum c_code (:1745): NaN propagates
template <typename T> __device__ __forceinline__ T fmax_nan(T x, T y) { return (y > x) ? y : ((x >= y) ? x : (T)NAN); }
template <typename T> __device__ __forceinline__ T fmin_nan(T x, T y) { return (y < x) ? y : ((x <= y) ? x : (T)NAN); }
template <typename T> __device__ __forceinline__ T imax(T x, T y) { return x > y ? x : y; }
template <typename T> __device__ __forceinline__ T imin(T x, T y) { return x < y ? x : y; }
__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }   // Sigmoid :1110
__device__ __forceinline__ double sigmoid_(double x) { return 1.0 / (1.0 + exp(-x)); }
__device__ __forceinline__ float softplus_(float x) {                                      // Softplus :1173
  return x < -37.0f ? expf(x) : (x < 18.0f ? log1pf(expf(x)) : (x < 33.3f ? x + expf(-x) : x));
}
__device__ __forceinline__ double softplus_(double x) {
  return x < -37.0 ? exp(x) : (x < 18.0 ? log1p(exp(x)) : (x < 33.3 ? x + exp(-x) : x));
}
__device__ __forceinline__ float log1mexp_(float x) { return x < -0.6931471805599453f ? log1pf(-expf(x)) : logf(-expm1f(x)); }
__device__ __forceinline__ double log1mexp_(double x) { return x < -0.6931471805599453 ? log1p(-exp(x)) : log(-expm1(x)); }
__device__ __forceinline__ float round_away(float x) { return x < 0 ? ceilf(x - 0.5f) : floorf(x + 0.5f); }
__device__ __forceinline__ double round_away(double x) { return x < 0 ? ceil(x - 0.5) : floor(x + 0.5); }

// ---- cross-lane reduction plumbing (64-wide wavefronts) ----
template <typename T> __device__ __forceinline__ T shfl_xor_(T v, int m) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int i[2]; } u; u.t = v;
    u.i[0] = __shfl_xor(u.i[0], m, 64); u.i[1] = __shfl_xor(u.i[1], m, 64);
    return u.t;
  } else if constexpr (sizeof(T) == 4) {
    union { T t; int i; } u; u.t = v; u.i = __shfl_xor(u.i, m, 64); return u.t;
  } else {
    int i = (int)v; i = __shfl_xor(i, m, 64); return (T)i;
  }
}
""" % (AHIP_MAXD, AHIP_MAXOPS)

_FLOAT_FN = {
    "sqrt": "sqrt", "exp": "exp", "exp2": "exp2", "expm1": "expm1", "log": "log",
    "log2": "log2", "log10": "log10", "log1p": "log1p", "sin": "sin", "cos": "cos",
    "tan": "tan", "arcsin": "asin", "arccos": "acos", "arctan": "atan", "sinh": "sinh",
    "cosh": "cosh", "tanh": "tanh", "arcsinh": "asinh", "arccosh": "acosh", "arctanh": "atanh",
    "ceil": "ceil", "floor": "floor", "trunc": "trunc", "round_half_to_even": "rint",
    "erf": "erf", "erfc": "erfc",
}

_IDENT = {  # reduction identities
    "add": lambda dt: "0", "mul": lambda dt: "1", "or": lambda dt: "0", "xor": lambda dt: "0",
    "and": lambda dt: "true" if dt == "bool" else "(%s)~(%s)0" % (RTYPE[dt], RTYPE[dt]),
}


def _is_float(dt):
    return dt in ("float32", "float64")


def _is_uint(dt):
    return dt.startswith("uint")


def _lit(value, dt):
    """C literal for a scalar constant of dtype ``dt``."""
    if dt == "bool":
        return "true" if value else "false"
    if _is_float(dt):
        v = float(value)
        if np.isnan(v):
            return "(%s)NAN" % RTYPE[dt]
        if np.isinf(v):
            return "(%s)(%sINFINITY)" % (RTYPE[dt], "-" if v < 0 else "")
        r = repr(float(np.float32(v))) if dt == "float32" else repr(v)
        if "e" not in r and "." not in r:
            r += ".0"
        return r + ("f" if dt == "float32" else "")
    v = int(value)
    if dt == "int64":
        return "(%dLL)" % v if v > -(2 ** 63) else "(-9223372036854775807LL - 1)"
    if dt == "uint64":
        return "(%dULL)" % v
    return "((%s)%d)" % (RTYPE[dt], v)


def _cast(expr, src_dt, dst_dt):
    if src_dt == dst_dt:
        return expr
    if dst_dt == "bool":
        return "((%s) != 0)" % expr
    return "((%s)(%s))" % (RTYPE[dst_dt], expr)


def _fname(base, dt):
    return base + ("f" if dt == "float32" else "")


def scalar_node_expr(op, ins, in_dts, dt):
    """C++ expression for one scalar node.  ``ins``: C expressions of the inputs (already in
    their own dtypes ``in_dts``); result must have register type ``RTYPE[dt]``."""
    T = RTYPE[dt]
    c = [_cast(e, d, dt) for e, d in zip(ins, in_dts)]  # inputs cast to the output dtype
    if op in ("add", "mul"):
        if dt == "bool":
            return "(" + (" || " if op == "add" else " && ").join(c) + ")"
        return "(" + (" + " if op == "add" else " * ").join(c) + ")"
    if op == "sub":
        return "(%s)(%s - %s)" % (T, c[0], c[1]) if dt != "bool" else "(%s != %s)" % (c[0], c[1])
    if op == "neg":
        return "(%s)(-%s)" % (T, c[0])
    if op == "true_div":
        return "(%s / %s)" % (c[0], c[1])
    if op == "int_div":
        if _is_float(dt):
            return "%s(%s / %s)" % (_fname("floor", dt), c[0], c[1])
        return "%s<%s>(%s, %s)" % ("udiv_floor" if _is_uint(dt) or dt == "bool" else "idiv_floor",
                                   T, c[0], c[1])
    if op == "mod":
        if _is_float(dt):
            return "fmod_py(%s, %s)" % (c[0], c[1])
        return "%s<%s>(%s, %s)" % ("umod" if _is_uint(dt) or dt == "bool" else "imod_py", T,
                                   c[0], c[1])
    if op == "pow":
        if _is_float(dt):
            return "%s(%s, %s)" % (_fname("pow", dt), c[0], c[1])
        return "%s<%s>(%s, %s)" % ("upow" if _is_uint(dt) or dt == "bool" else "ipow", T, c[0], c[1])
    if op in ("maximum", "minimum"):
        if _is_float(dt):
            f = "fmax_nan" if op == "maximum" else "fmin_nan"
        else:
            f = "imax" if op == "maximum" else "imin"
        e = c[0]
        for x in c[1:]:
            e = "%s<%s>(%s, %s)" % (f, T, e, x)
        return e
    if op in ("lt", "gt", "le", "ge", "eq", "neq"):
        sym = {"lt": "<", "gt": ">", "le": "<=", "ge": ">=", "eq": "==", "neq": "!="}[op]
        ct = np.result_type(*[np.dtype(d) for d in in_dts]).name
        if ct not in RTYPE:
            ct = "float64"
        a, b = (_cast(e, d, ct) for e, d in zip(ins, in_dts))
        return _cast("(%s %s %s)" % (a, sym, b), "bool", dt)
    if op in ("and", "or", "xor"):
        if dt == "bool":
            sym = {"and": "&&", "or": "||", "xor": "!="}[op]
        else:
            sym = {"and": "&", "or": "|", "xor": "^"}[op]
        return "(%s)(" % T + (" %s " % sym).join(c) + ")"
    if op == "invert":
        return "(!%s)" % c[0] if dt == "bool" else "(%s)(~%s)" % (T, c[0])
    if op == "abs":
        if _is_float(dt):
            return "%s(%s)" % (_fname("fabs", dt), c[0])
        if _is_uint(dt) or dt == "bool":
            return c[0]
        return "(%s)(%s < 0 ? -%s : %s)" % (T, c[0], c[0], c[0])
    if op == "sgn":
        if _is_uint(dt) or dt == "bool":
            return "(%s)(%s != 0)" % (T, c[0])
        return "(%s)((%s > 0) - (%s < 0))" % (T, c[0], c[0])
    if op == "sqr":
        return "(%s)(%s * %s)" % (T, c[0], c[0]) if dt != "bool" else c[0]
    if op in _FLOAT_FN and _is_float(dt):
        return "%s(%s)" % (_fname(_FLOAT_FN[op], dt), c[0])
    if op in ("ceil", "floor", "trunc", "round_half_to_even", "round_half_away_from_zero") \
            and not _is_float(dt):
        return c[0]
    if op == "round_half_away_from_zero":
        return "round_away(%s)" % c[0]
    if op == "reciprocal":
        return "((%s)1 / %s)" % (T, c[0])
    if op == "sigmoid":
        return "sigmoid_(%s)" % c[0]
    if op == "softplus":
        return "softplus_(%s)" % c[0]
    if op == "log1mexp":
        return "log1mexp_(%s)" % c[0]
    if op == "deg2rad":
        return "(%s * (%s)0.017453292519943295)" % (c[0], T)
    if op == "rad2deg":
        return "(%s * (%s)57.29577951308232)" % (c[0], T)
    if op == "arctan2":
        return "%s(%s, %s)" % (_fname("atan2", dt), c[0], c[1])
    if op in ("identity", "cast"):
        return c[0]
    if op == "second":
        return c[1]
    if op == "switch":
        return "((%s) ? %s : %s)" % (_cast(ins[0], in_dts[0], "bool"), c[1], c[2])
    if op == "clip":
        return "(%s < %s ? %s : (%s > %s ? %s : %s))" % (c[0], c[1], c[1], c[0], c[2], c[2], c[0])
    if op == "isnan":
        return _cast("(%s != %s)" % (ins[0], ins[0]) if _is_float(in_dts[0]) else "false", "bool", dt)
    if op == "isinf":
        if _is_float(in_dts[0]):
            return _cast("(%s(%s) == INFINITY)" % (_fname("fabs", in_dts[0]), ins[0]), "bool", dt)
        return _cast("false", "bool", dt)
    raise NotImplementedError(f"HIP codegen: scalar op {op!r} for dtype {dt}")


def emit_scalar_body(scalar, in_exprs, in_dts, indent="      ", suffix=""):
    """Lines computing all temporaries of a plan scalar expression; returns (lines, out_exprs,
    out_dtypes)."""
    lines = []
    tdt = []

    def ref(r):
        if r[0] == "i":
            return in_exprs[r[1]], in_dts[r[1]]
        if r[0] == "t":
            return "t%d%s" % (r[1], suffix), tdt[r[1]]
        return _lit(r[1], r[2]), r[2]

    for k, n in enumerate(scalar["nodes"]):
        refs = [ref(r) for r in n["in"]]
        e = scalar_node_expr(n["op"], [x[0] for x in refs], [x[1] for x in refs], n["dtype"])
        lines.append("%sconst %s t%d%s = %s;" % (indent, RTYPE[n["dtype"]], k, suffix, e))
        tdt.append(n["dtype"])
    outs = [ref(r) for r in scalar["out"]]
    return lines, [o[0] for o in outs], [o[1] for o in outs]


def red_combine(op, acc_dt, a, b):
    T = RTYPE[acc_dt]
    if op == "add":
        return "(%s || %s)" % (a, b) if acc_dt == "bool" else "(%s)(%s + %s)" % (T, a, b)
    if op == "mul":
        return "(%s && %s)" % (a, b) if acc_dt == "bool" else "(%s)(%s * %s)" % (T, a, b)
    if op == "maximum":
        return "%s<%s>(%s, %s)" % ("fmax_nan" if _is_float(acc_dt) else "imax", T, a, b)
    if op == "minimum":
        return "%s<%s>(%s, %s)" % ("fmin_nan" if _is_float(acc_dt) else "imin", T, a, b)
    if acc_dt == "bool":
        sym = {"and": "&&", "or": "||", "xor": "!="}[op]
    else:
        sym = {"and": "&", "or": "|", "xor": "^"}[op]
    return "(%s)(%s %s %s)" % (T, a, sym, b)


def red_identity(op, acc_dt):
    if op in _IDENT:
        return _IDENT[op](acc_dt)
    info_max = op == "minimum"
    if _is_float(acc_dt):
        return "(%s)(%sINFINITY)" % (RTYPE[acc_dt], "" if info_max else "-")
    if acc_dt == "bool":
        return "true" if info_max else "false"
    ii = np.iinfo(acc_dt)
    return _lit(ii.max if info_max else ii.min, acc_dt)


# ----------------------------------------------------------------------------------------
# kernel specs
# ----------------------------------------------------------------------------------------
class KernelSpec:
    """Everything that determines the generated source of one fused kernel.

    scalar      : plan scalar expression (dict)
    in_dtypes   : dtypes of the Elemwise inputs
    out_dtypes  : dtypes of the materialised Elemwise outputs (may be empty for pure reduce)
    out_refs    : indices into scalar["out"] that are stored
    inner       : per-operand (inputs then stored outputs) inner-dim class: 'c' unit stride,
                  'b' broadcast (stride 0), 's' arbitrary stride (forces vec == 1)
    nd          : number of collapsed dims (elemwise / reduce_all) or nk + nr (reduce_axis)
    vec, block  : elements per lane along the inner dim; threads per workgroup
    idx64       : use 64-bit index arithmetic
    reduce      : None | dict(kind='all'|'row'|'col', op=, acc=, out=, ref=<scalar out index>,
                  nk=, nr=)
    """

    def __init__(self, scalar, in_dtypes, out_dtypes, out_refs, inner, nd, vec, block=256,
                 idx64=False, reduce=None):
        self.scalar = scalar
        self.in_dtypes = list(in_dtypes)
        self.out_dtypes = list(out_dtypes)
        self.out_refs = list(out_refs)
        self.inner = list(inner)
        self.nd = nd
        self.vec = vec
        self.block = block
        self.idx64 = idx64
        self.reduce = reduce
        assert len(self.inner) == len(self.in_dtypes) + len(self.out_dtypes)
        assert 1 <= nd <= AHIP_MAXD and len(self.inner) <= AHIP_MAXOPS

    def key(self):
        import json
        blob = json.dumps([self.scalar, self.in_dtypes, self.out_dtypes, self.out_refs,
                           self.inner, self.nd, self.vec, self.block, self.idx64, self.reduce],
                          sort_keys=True)
        return hashlib.sha256(blob.encode()).hexdigest()[:24]


def _offset_code(spec, nops, nd_lo, nd_hi, var, idx_t, inner_vecs=None):
    """Index decomposition of `var` over dims [nd_lo, nd_hi) (outermost first), accumulating
    per-operand offsets into off<k>.  If inner_vecs, the innermost dim is counted in vectors
    and its index is left in `inner` (not multiplied into the offsets)."""
    L = []
    dims = list(range(nd_lo, nd_hi))
    L.append("      %s rem = %s;" % (idx_t, var))
    for pos, d in enumerate(reversed(dims)):
        last = pos == len(dims) - 1
        is_inner = (d == nd_hi - 1) and inner_vecs
        ext = inner_vecs if is_inner else "(%s)a.shape[%d]" % (idx_t, d)
        if last:
            L.append("      { const %s r = rem;" % idx_t)
        else:
            L.append("      { const %s q = rem / %s; const %s r = rem - q * %s; rem = q;" %
                     (idx_t, ext, idx_t, ext))
        if is_inner:
            L.append("        inner = r;")
        else:
            for k in range(nops):
                L.append("        off%d += (i64)r * a.stride[%d][%d];" % (k, k, d))
        L.append("      }")
    return L


def generate(spec: KernelSpec):
    """Return (source, kernel_names) for a spec.  kernel_names: ('k',) or ('k', 'k_fin')."""
    nin = len(spec.in_dtypes)
    nout = len(spec.out_dtypes)
    nops = nin + nout
    V = spec.vec
    idx_t = "i64" if spec.idx64 else "int"
    red = spec.reduce
    name = "ew_" + spec.key()
    L = [PRELUDE]
    L.append('extern "C" __global__ __launch_bounds__(%d) void %s(Args a) {' % (spec.block, name))
    for k in range(nin):
        L.append("  const %s* __restrict__ p%d = (const %s*)a.ptr[%d];" %
                 (CTYPE[spec.in_dtypes[k]], k, CTYPE[spec.in_dtypes[k]], k))
    for k in range(nout):
        L.append("  %s* __restrict__ p%d = (%s*)a.ptr[%d];" %
                 (CTYPE[spec.out_dtypes[k]], nin + k, CTYPE[spec.out_dtypes[k]], nin + k))
    if any(c == "s" for c in spec.inner):
        assert V == 1
        for k in range(nops):
            if spec.inner[k] == "s":
                L.append("  const i64 is%d = a.stride[%d][%d];" % (k, k, spec.nd - 1))

    if red is not None:
        acc_t = RTYPE[red["acc"]]
        L.append("  %s acc = %s;" % (acc_t, red_identity(red["op"], red["acc"])))

    def body(elem_off_exprs):
        """Load VEC elements per operand, evaluate, store/accumulate.  elem_off_exprs[k] is
        the element offset expression of operand k at the start of the vector."""
        B = []
        # loads
        for k in range(nin):
            ct = CTYPE[spec.in_dtypes[k]]
            cls = spec.inner[k]
            if cls == "c" and V > 1:
                B.append("      const Pack<%s, %d> x%d = *(const Pack<%s, %d>*)(p%d + %s);" %
                         (ct, V, k, ct, V, k, elem_off_exprs[k]))
            elif cls == "b" or V == 1:
                B.append("      const %s x%d = p%d[%s];" % (ct, k, k, elem_off_exprs[k]))
        for k in range(nout):
            if V > 1:
                B.append("      Pack<%s, %d> y%d;" % (CTYPE[spec.out_dtypes[k]], V, k))
        for v in range(V):
            ins = []
            for k in range(nin):
                if spec.inner[k] == "c" and V > 1:
                    e = "x%d.v[%d]" % (k, v)
                else:
                    e = "x%d" % k
                if spec.in_dtypes[k] == "bool":
                    e = "(%s != 0)" % e
                ins.append(e)
            lines, outs, odts = emit_scalar_body(spec.scalar, ins, spec.in_dtypes,
                                                 suffix="_%d" % v)
            B.extend(lines)
            for k, ri in enumerate(spec.out_refs):
                val = _cast(outs[ri], odts[ri], spec.out_dtypes[k])
                if spec.out_dtypes[k] == "bool":
                    val = "(unsigned char)(%s)" % val
                if V > 1:
                    B.append("      y%d.v[%d] = %s;" % (k, v, val))
                else:
                    B.append("      p%d[%s] = %s;" % (nin + k, elem_off_exprs[nin + k], val))
            if red is not None:
                val = _cast(outs[red["ref"]], odts[red["ref"]], red["acc"])
                B.append("      acc = %s;" % red_combine(red["op"], red["acc"], "acc", val))
        if V > 1:
            for k in range(nout):
                B.append("      *(Pack<%s, %d>*)(p%d + %s) = y%d;" %
                         (CTYPE[spec.out_dtypes[k]], V, nin + k, elem_off_exprs[nin + k], k))
        return B

    def elem_offsets():
        out = []
        for k in range(nops):
            cls = spec.inner[k]
            if cls == "c":
                out.append("off%d + (i64)inner * %d" % (k, V))
            elif cls == "b":
                out.append("off%d" % k)
            else:
                out.append("off%d + (i64)inner * is%d" % (k, k))
        return out

    if red is None or red["kind"] == "all":
        nd = spec.nd
        L.append("  const %s inner_vecs = (%s)(a.shape[%d] / %d);" % (idx_t, idx_t, nd - 1, V))
        L.append("  const %s items = (%s)(a.n / %d);" % (idx_t, idx_t, V))
        L.append("  const %s step = (%s)gridDim.x * %d;" % (idx_t, idx_t, spec.block))
        L.append("  for (%s item = (%s)blockIdx.x * %d + threadIdx.x; item < items; item += step) {"
                 % (idx_t, idx_t, spec.block))
        L.append("      i64 " + ", ".join("off%d = 0" % k for k in range(nops)) + ";")
        L.append("      %s inner = 0;" % idx_t)
        L.extend(_offset_code(spec, nops, 0, nd, "item", idx_t, inner_vecs="inner_vecs"))
        L.extend(body(elem_offsets()))
        L.append("  }")
    else:
        nk, nr = red["nk"], red["nr"]
        assert V == 1
        if red["kind"] == "row":
            L.append("  const int lane = threadIdx.x & 63;")
            L.append("  const i64 o = (i64)blockIdx.x * %d + (threadIdx.x >> 6);" % (spec.block // 64))
            L.append("  if (o >= a.n) return;")
        else:
            L.append("  const i64 o = (i64)blockIdx.x * %d + threadIdx.x;" % spec.block)
            L.append("  if (o >= a.n) return;")
        L.append("  i64 " + ", ".join("base%d = 0" % k for k in range(nops)) + ";")
        L.append("  {")
        L.append("      i64 " + ", ".join("off%d = 0" % k for k in range(nops)) + ";")
        L.extend(_offset_code(spec, nops, 0, nk, "o", "i64"))
        L.append("      " + " ".join("base%d = off%d;" % (k, k) for k in range(nops)))
        L.append("  }")
        L.append("  const i64 nred = a.aux0;")
        if red["kind"] == "row":
            L.append("  for (i64 r0 = lane; r0 < nred; r0 += 64) {")
        else:
            L.append("  const i64 per = (nred + a.aux1 - 1) / a.aux1;")
            L.append("  const i64 rbeg = (i64)blockIdx.y * per;")
            L.append("  const i64 rend = (rbeg + per < nred) ? rbeg + per : nred;")
            L.append("  for (i64 r0 = rbeg; r0 < rend; ++r0) {")
        L.append("      i64 " + ", ".join("off%d = base%d" % (k, k) for k in range(nops)) + ";")
        L.extend(_offset_code(spec, nops, nk, nk + nr, "r0", "i64"))
        L.extend(body(["off%d" % k for k in range(nops)]))
        L.append("  }")

    if red is not None:
        acc_t = RTYPE[red["acc"]]
        comb = lambda a_, b_: red_combine(red["op"], red["acc"], a_, b_)  # noqa: E731
        wave_red = ["  for (int m = 32; m > 0; m >>= 1) acc = %s;" %
                    comb("acc", "shfl_xor_<%s>(acc, m)" % acc_t)]
        if red["kind"] == "all":
            nw = spec.block // 64
            L.extend(wave_red)
            L.append("  __shared__ %s sm[%d];" % (acc_t if acc_t != "bool" else "unsigned char", nw))
            L.append("  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;")
            L.append("  __syncthreads();")
            L.append("  if (threadIdx.x == 0) {")
            L.append("    %s r = sm[0];" % acc_t)
            L.append("    for (int w = 1; w < %d; ++w) r = %s;" % (nw, comb("r", "(%s)sm[w]" % acc_t)))
            L.append("    ((%s*)a.ws)[blockIdx.x] = r;" % CTYPE[red["acc"]])
            L.append("  }")
        elif red["kind"] == "row":
            L.extend(wave_red)
            L.append("  if (lane == 0) ((%s*)a.out)[o] = %s;" %
                     (CTYPE[red["out"]], _store_val("acc", red["acc"], red["out"])))
        else:
            L.append("  if (a.aux1 == 1) ((%s*)a.out)[o] = %s;" %
                     (CTYPE[red["out"]], _store_val("acc", red["acc"], red["out"])))
            L.append("  else ((%s*)a.out)[(i64)blockIdx.y * a.n + o] = acc;" % CTYPE[red["acc"]])
    L.append("}")

    names = [name]
    if red is not None and red["kind"] == "all":
        acc_t = RTYPE[red["acc"]]
        comb = lambda a_, b_: red_combine(red["op"], red["acc"], a_, b_)  # noqa: E731
        fin = name + "_fin"
        names.append(fin)
        L.append('extern "C" __global__ __launch_bounds__(256) void %s(Args a) {' % fin)
        L.append("  const %s* ws = (const %s*)a.ws;" % (CTYPE[red["acc"]], CTYPE[red["acc"]]))
        L.append("  const int np = (int)a.aux0;")
        L.append("  %s acc = %s;" % (acc_t, red_identity(red["op"], red["acc"])))
        L.append("  for (int i = threadIdx.x; i < np; i += 256) acc = %s;" %
                 comb("acc", "(%s)ws[i]" % acc_t))
        L.append("  for (int m = 32; m > 0; m >>= 1) acc = %s;" %
                 comb("acc", "shfl_xor_<%s>(acc, m)" % acc_t))
        L.append("  __shared__ %s sm[4];" % (acc_t if acc_t != "bool" else "unsigned char"))
        L.append("  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;")
        L.append("  __syncthreads();")
        L.append("  if (threadIdx.x == 0) {")
        L.append("    %s r = sm[0];" % acc_t)
        L.append("    for (int w = 1; w < 4; ++w) r = %s;" % comb("r", "(%s)sm[w]" % acc_t))
        L.append("    *(%s*)a.out = %s;" % (CTYPE[red["out"]], _store_val("r", red["acc"], red["out"])))
        L.append("  }")
        L.append("}")
    return "\n".join(L) + "\n", tuple(names)


def _store_val(expr, src_dt, dst_dt):
    v = _cast(expr, src_dt, dst_dt)
    if dst_dt == "bool":
        v = "(unsigned char)(%s)" % v
    return v


IDENTITY_SCALAR = {"n_in": 1, "nodes": [], "out": [["i", 0]]}
