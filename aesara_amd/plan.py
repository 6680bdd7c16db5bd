"""Backend-neutral launch-plan IR for the HIP linker.

A :class:`Plan` is what ``HipLinker.fgraph_convert`` (aesara_amd/linker.py) produces from a
rewritten Aesara ``FunctionGraph`` and what the device executor (aesara_amd/executor.py)
consumes.  It is plain data (JSON-serialisable) so that graphs lowered from the real
reference front end in the authoring container can be executed on a GPU box that has no
Aesara installation, and so that the C-ABI run loop never touches Python graph objects
(the role `link/vm.py:1212 VMLinker.make_all` + `lazylinker_c.c` play in the reference).

Node ``op`` names mirror the reference Op class names (``Elemwise``, ``CAReduce``,
``DimShuffle``, ``Gemm``, ``Gemv``, ``Dot22``, ``BatchedDot``, ``Alloc``, ``Subtensor`` …);
``params`` holds the Op's ``__props__`` restated as JSON values.

Scalar expressions (the body of an ``Elemwise``; reference: ``ScalarOp.c_code`` /
``Composite.c_code_template`` scalar/basic.py:4250) are encoded as a topologically ordered
list of scalar nodes::

    {"n_in": 4, "nodes": [{"op": "sub", "in": [["i", 1], ["i", 2]], "dtype": "float64"}, ...],
     "out": [["t", 3]]}

where a reference is ``["i", k]`` (k-th Elemwise input), ``["t", k]`` (result of scalar node
k) or ``["c", value, dtype]`` (a scalar constant).
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import numpy as np

PLAN_VERSION = 1

# dtypes the device path supports (reference supports complex too; out of scope per SURVEY §8a H2)
DTYPES = (
    "bool", "int8", "int16", "int32", "int64",
    "uint8", "uint16", "uint32", "uint64",
    "float32", "float64",
)


@dataclass
class Var:
    id: int
    dtype: str
    shape: List[Optional[int]]  # static shape; 1 == broadcastable dim (TensorType.shape)
    name: Optional[str] = None
    # set for constants: flat python list + concrete shape
    const: Optional[Dict[str, Any]] = None

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def bcast(self):
        return [s == 1 for s in self.shape]

    def to_json(self):
        d = {"id": self.id, "dtype": self.dtype, "shape": self.shape}
        if self.name:
            d["name"] = self.name
        if self.const is not None:
            c = self.const
            if "array" in c:    # large constants are held as arrays; a list only when serialised
                c = {"shape": c["shape"], "data": np.asarray(c["array"]).ravel().tolist()}
            d["const"] = c
        return d

    @staticmethod
    def from_json(d):
        return Var(d["id"], d["dtype"], list(d["shape"]), d.get("name"), d.get("const"))

    def const_value(self) -> np.ndarray:
        assert self.const is not None
        if "array" in self.const:
            return np.asarray(self.const["array"], dtype=self.dtype).reshape(self.const["shape"])
        arr = np.array(self.const["data"], dtype=self.dtype)
        return arr.reshape(self.const["shape"])


@dataclass
class Node:
    op: str
    inputs: List[int]
    outputs: List[int]
    params: Dict[str, Any] = field(default_factory=dict)

    def to_json(self):
        return {"op": self.op, "inputs": self.inputs, "outputs": self.outputs,
                "params": _params_to_json(self.params)}

    @staticmethod
    def from_json(d):
        return Node(d["op"], list(d["inputs"]), list(d["outputs"]),
                    _params_from_json(d.get("params", {})))


def _params_to_json(p):
    out = {}
    for k, v in p.items():
        if isinstance(v, Plan):
            out[k] = {"__plan__": v.to_json()}
        elif callable(v):
            # HostCall: the Op is a Python callable (Print's print function, an as_op function)
            raise TypeError("a plan with a host callback (%r) cannot be serialised" % (v,))
        else:
            out[k] = v
    return out


def _params_from_json(p):
    out = {}
    for k, v in p.items():
        if isinstance(v, dict) and "__plan__" in v:
            out[k] = Plan.from_json(v["__plan__"])
        else:
            out[k] = v
    return out


@dataclass
class Plan:
    name: str
    vars: Dict[int, Var]
    inputs: List[int]
    outputs: List[int]
    nodes: List[Node]

    # ---- construction helpers -------------------------------------------------
    def new_var(self, dtype, shape, name=None, const=None) -> int:
        vid = (max(self.vars) + 1) if self.vars else 0
        self.vars[vid] = Var(vid, str(dtype), list(shape), name, const)
        return vid

    def add_const(self, value, dtype=None, name=None) -> int:
        arr = np.asarray(value, dtype=dtype)
        if arr.size > 4096:
            # data embedded in the graph (a design matrix, a lookup table): kept by reference
            const = {"shape": list(arr.shape), "array": np.ascontiguousarray(arr)}
        else:
            const = {"shape": list(arr.shape), "data": arr.ravel().tolist()}
        shape = [1 if s == 1 else int(s) for s in arr.shape]
        return self.new_var(arr.dtype.name, shape, name, const)

    # ---- (de)serialisation ----------------------------------------------------
    def to_json(self):
        return {
            "version": PLAN_VERSION,
            "name": self.name,
            "vars": [v.to_json() for v in self.vars.values()],
            "inputs": self.inputs,
            "outputs": self.outputs,
            "nodes": [n.to_json() for n in self.nodes],
        }

    @staticmethod
    def from_json(d):
        if d.get("version", PLAN_VERSION) != PLAN_VERSION:
            raise ValueError(f"unsupported plan version {d.get('version')}")
        vs = {v["id"]: Var.from_json(v) for v in d["vars"]}
        return Plan(d["name"], vs, list(d["inputs"]), list(d["outputs"]),
                    [Node.from_json(n) for n in d["nodes"]])

    def dumps(self, **kw):
        return json.dumps(self.to_json(), **kw)

    @staticmethod
    def loads(s):
        return Plan.from_json(json.loads(s))

    # ---- queries ----------------------------------------------------------------
    def clients(self):
        """var id -> list of (node index, input position); outputs counted as client 'out'."""
        c: Dict[int, list] = {vid: [] for vid in self.vars}
        for ni, n in enumerate(self.nodes):
            for pos, vid in enumerate(n.inputs):
                c[vid].append((ni, pos))
        for pos, vid in enumerate(self.outputs):
            c[vid].append(("out", pos))
        return c

    def pretty(self):
        lines = [f"Plan {self.name}: inputs={self.inputs} outputs={self.outputs}"]
        for i, n in enumerate(self.nodes):
            p = {k: ("<plan>" if isinstance(v, Plan) else v) for k, v in n.params.items()}
            if "scalar" in p:
                p["scalar"] = scalar_to_str(p["scalar"])
            outs = ", ".join(f"v{o}:{self.vars[o].dtype}{self.vars[o].shape}" for o in n.outputs)
            lines.append(f"  {i:3d} {outs} = {n.op}({', '.join('v%d' % x for x in n.inputs)}) {p}")
        return "\n".join(lines)


def scalar_to_str(s):
    """Human-readable form of a scalar expression (for debugging / plan dumps)."""
    def ref(r):
        if r[0] == "i":
            return f"i{r[1]}"
        if r[0] == "t":
            return f"t{r[1]}"
        return repr(r[1])
    parts = [f"t{k}={n['op']}({', '.join(ref(r) for r in n['in'])})"
             for k, n in enumerate(s["nodes"])]
    return "; ".join(parts) + " -> " + ", ".join(ref(r) for r in s["out"])
