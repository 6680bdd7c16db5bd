"""``HipLinker`` — the drop-in ``aesara.link`` Linker for MI355X (gfx950).

Boundary: subclasses ``aesara.link.basic.JITLinker`` (reference
link/basic.py:580) and implements exactly its three abstract methods + ``output_filter``:

* ``fgraph_convert`` (:589)  -> lowers the rewritten FunctionGraph to a launch ``Plan``
* ``jit_compile``   (:609)  -> binds the plan to the device executor (C-ABI ``libaesara_hip.so``)
* ``create_thunk_inputs`` (:595) -> the storage cells of ``fgraph.inputs``
* ``output_filter`` (:612) -> device arrays are returned as is (JAX precedent,
  tests/link/jax/test_basic.py:75-79) or converted to ``numpy`` when ``return_numpy=True``

Registration mirrors the JAX/Numba modes (compile/mode.py:448-455): ``register()`` adds linker
``"hip"`` (mode.py:54 ``register_linker``) and mode ``"HIP"`` (mode.py:525 ``register_mode``).
The rewrite query excludes ``inplace`` (destroy maps are handled by the executor's buffer
planner instead), ``cxx_only`` and ``c_blas`` exactly like the non-C backends.

This module needs the reference front end (``import aesara``); everything below it
(plan, executor, C-ABI) does not.
"""
from __future__ import annotations

from aesara.compile.mode import Mode, register_linker, register_mode
from aesara.graph.rewriting.db import RewriteDatabaseQuery
from aesara.link.basic import JITLinker

from .lower import lower_fgraph

# the query below names rewrites of tensor/nnet/basic.py: they must be registered before a query
# mentions them (a RewriteDatabase refuses to register a name that a query has already created)
import aesara.tensor.nnet.basic  # noqa: E402,F401  isort:skip

HIP_QUERY = RewriteDatabaseQuery(
    include=["fast_run"],
    exclude=["cxx_only", "inplace", "c_blas", "BlasOpt_inplace",
             # tensor/nnet/basic.py rewrites that replace Softmax / log / indexing sub-graphs by the
             # CPU-specific fused Ops SoftmaxWithBias (:321), CrossentropySoftmaxArgmax1HotWithBias
             # (:1016/:1043/:1214), CrossentropySoftmax1HotWithBiasDx (:1108/:1262): the primitives
             # they start from are what the row-chain kernels fuse here
             "local_softmax_with_bias", "crossentropy_to_crossentropy_with_softmax_with_bias",
             "crossentropy_to_crossentropy_with_softmax", "xent",
             "local_softmax_grad_to_crossentropy_with_softmax_grad",
             "local_crossentropy_to_crossentropy_with_softmax_grad", "local_argmax_pushdown",
             "local_advanced_indexing_crossentropy_onehot",
             "local_advanced_indexing_crossentropy_onehot_grad"],
)


# (the reference names its queries — ``OPT_FAST_RUN.name = "OPT_FAST_RUN"``, compile/mode.py:265 — and
# code keys off the name: ``"FAST_RUN" in get_default_mode().optimizer.name``)
HIP_QUERY.name = "OPT_FAST_RUN_HIP"


class HipLinker(JITLinker):
    """A ``Linker`` that runs a whole ``FunctionGraph`` as HIP kernels on an MI355X."""

    def __init__(self, *args, return_numpy=False, use_graph=True, executor_factory=None,
                 fast_call=True, check_indices=True, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_numpy = return_numpy
        # check_indices: True (default) — an out-of-range index / a persistent-Scan time-out is
        # raised by the call that met it, as the reference does (one blocking read of the error
        # words per call of a plan that HAS index kernels or a persistent Scan; plans without
        # them never wait); "deferred" — the words are copied behind the launches and examined
        # when they have landed (next call at the latest, ``executor.check()`` waits): no stall in
        # a training loop, the error surfaces one call late; False — unchecked.
        self.check_indices = check_indices
        # use_graph (default): the first call per input signature runs the host logic and records
        # the launches; later calls of that signature replay them with one host call (new device
        # buffers are rebound, host arrays staged) into a lifetime-packed arena and return fresh
        # outputs.  Plans with data-dependent host control flow fall back to per-node launches.
        self.use_graph = use_graph
        # fast_call: the function's VM is one closure (bind cells -> executor -> store cells)
        # instead of ``streamline`` over a one-thunk list (H0: the per-eval host protocol)
        self.fast_call = fast_call
        # test hook: lets CPU-only tests substitute a checker for the device executor.
        self.executor_factory = executor_factory
        self.plan = None
        self.profile = None

    def accept(self, fgraph, no_recycling=None, profile=None):
        """``PerformLinker.accept`` (link/basic.py:300) drops ``profile`` and, when the instance
        is already bound to another graph, re-creates the linker from ``allow_gc`` alone; this
        one keeps its options (``return_numpy``, ``use_graph`` ...) and remembers the
        ``ProfileStats`` so the executor can time its steps (``VMLinker.accept`` link/vm.py:868)."""
        if self.fgraph is not None and self.fgraph is not fgraph:
            from copy import copy

            new = copy(self)
            new.fgraph = None
            for k in ("executor", "_update_outputs"):
                new.__dict__.pop(k, None)
            new.plan = None
            return new.accept(fgraph, no_recycling, profile)
        super().accept(fgraph, no_recycling=no_recycling, profile=profile)
        self.profile = profile
        return self

    def __getstate__(self):
        # a pickled / deep-copied Function is re-linked when it is loaded (the reference pickles the
        # FunctionMaker, compile/function/types.py:1125 _pickle_Function): device executors,
        # module handles and the lowered plan are rebuilt then, never serialised
        state = dict(self.__dict__)
        for k in ("executor", "plan", "_update_outputs", "_order"):
            state.pop(k, None)
        state["plan"] = None
        return state

    # -- JITLinker API ---------------------------------------------------------------
    def fgraph_convert(self, fgraph, order=None, input_storage=None, output_storage=None,
                       storage_map=None, **kwargs):
        self._order = list(order) if order is not None else list(fgraph.toposort())
        # inputs a destructive Op overwrites (``destroy_map`` on a graph input: only graphs the
        # caller built with in-place Ops and compiled with ``accept_inplace`` / ``In(mutable=True)``
        # — the HIP rewrite query never inserts them): the reference leaves the result in the
        # caller's own array.  The executor never writes into an argument; the final content of
        # such a buffer becomes a hidden plan output that is copied back after the call.
        self._writeback = _destroyed_inputs(fgraph, self._order)
        self.plan = lower_fgraph(fgraph, order=order, name=getattr(fgraph, "name", None)
                                 or "fgraph", inner_rewriter=hip_mode.optimizer,
                                 extra_outputs=[v for _i, v in self._writeback])
        self._n_outputs = len(fgraph.outputs)
        # outputs that are ``updates=`` expressions go back into shared-variable cells
        # (types.py:1060-1069), never to the caller: they stay device tensors
        self._update_outputs = {fgraph.outputs[i]
                                for i in getattr(fgraph, "update_mapping", None) or {}}
        return self.plan

    def jit_compile(self, plan):
        if self.executor_factory is not None:
            return self._with_writeback(self.executor_factory(plan))
        from .executor import PlanExecutor  # imports the C-ABI; fails loudly if missing

        ex = PlanExecutor(plan, use_graph=self.use_graph, check_indices=self.check_indices)
        if self.profile:
            ex.enable_profile()
        self.executor = ex
        return self._with_writeback(ex)

    def _with_writeback(self, ex):
        wb = [i for i, _v in getattr(self, "_writeback", ())]
        if not wb:
            return ex
        n = self._n_outputs

        def call(*args):
            res = ex(*args)
            for k, i in enumerate(wb):
                _write_back(args[i], res[n + k])
            return res[:n]
        return call

    def make_all(self, *args, **kwargs):
        """``JITLinker.make_all`` (link/basic.py:684-747) plus one change: the input cells handed
        to ``Function`` accept device tensors as they are (``sharedvar.DeviceContainer``), so an
        ``updates=`` output is stored back into its shared variable without leaving HBM, and
        their ``type`` filters a device-tensor ARGUMENT on the device (``devcell.DeviceFilterType``:
        ``Function.__call__`` needs no ``trust_input`` for device inputs)."""
        from .sharedvar import DeviceContainer

        if not self.fgraph.apply_nodes:
            # nothing to compute (outputs are inputs / constants, e.g. a statically known shape):
            # ``JITLinker.make_all`` hands ``streamline`` one thunk for zero nodes and raises
            # (link/basic.py:747, link/utils.py:183) — the per-node loop of its parent class is
            # the empty loop, which is all such a graph needs
            from aesara.link.basic import PerformLinker
            fn, ins, outs, thunks, nodes = PerformLinker.make_all(self, *args, **kwargs)
            return fn, [DeviceContainer.adopt_input(c) for c in ins], outs, thunks, nodes
        fn, ins, outs, thunks, nodes = super().make_all(*args, **kwargs)
        if self.fast_call and len(thunks) == 1:
            fn = self._fast_vm(fn, thunks[0])
        return fn, [DeviceContainer.adopt_input(c) for c in ins], outs, thunks, nodes

    def _fast_vm(self, slow_fn, thunk):
        """What ``Function.__call__`` invokes as ``self.vm()`` (compile/function/types.py:967-973).

        ``streamline`` (link/utils.py:150) wraps the single JIT thunk in a generic loop: clear the
        no-recycling cells, iterate (thunk, node, old_storage) triples under a try block, clear
        dead storage; the thunk itself (link/basic.py:663-678) rebuilds its argument list, walks
        ``zip(fgraph.outputs, cells, values)`` and sets a compute-map flag per output.  With one
        thunk that owns the whole graph none of that carries information: the closure below binds
        the input cells to the executor call and stores the results, nothing else.  Attributes
        the reference reads from the VM (``allow_gc``, ``storage_map``, ``jit_fn``, ``thunks``,
        ``nodes``) are forwarded."""
        jit = slow_fn.jit_fn
        in_cells = list(thunk.inputs)
        out_cells = list(thunk.outputs)
        out_vars = list(self.fgraph.outputs)
        # (a ScalarType output — ScalarFromTensor handed out — is a NumPy scalar in the reference,
        # scalar/basic.py:302 ScalarType.filter: it goes through ``output_filter`` as well)
        plain = not self.return_numpy and not any(_is_scalar_type(v) for v in out_vars)
        ofilter = self.output_filter
        n_in = len(in_cells)
        if n_in == 1:
            c0 = in_cells[0]
            call = lambda: jit(c0[0])                                  # noqa: E731
        elif n_in == 2:
            c0, c1 = in_cells
            call = lambda: jit(c0[0], c1[0])                           # noqa: E731
        elif n_in == 3:
            c0, c1, c2 = in_cells
            call = lambda: jit(c0[0], c1[0], c2[0])                    # noqa: E731
        else:
            call = lambda: jit(*[c[0] for c in in_cells])              # noqa: E731

        # cells the reference's ``streamline`` clears before every run (link/utils.py:196-197):
        # outputs that are not borrowed must never be recycled — the executor hands out fresh
        # tensors for them anyway; the cells are dropped so no stale reference outlives a failure
        out_ids = {id(c) for c in out_cells}        # identity, not list equality ([None] == [None])
        clear = [c for c in self._no_recycling_cells(slow_fn) if id(c) in out_ids]

        if plain and len(out_cells) == 1:
            oc = out_cells[0]

            def run():
                oc[0] = call()[0]
        elif plain:
            def run():
                for cell, val in zip(out_cells, call()):
                    cell[0] = val
        else:
            def run():
                for var, cell, val in zip(out_vars, out_cells, call()):
                    cell[0] = ofilter(var, val)

        def vm():
            try:
                run()
            except Exception as e:
                # ``Function.__call__`` (types.py:974-991) re-raises through ``raise_with_op`` with
                # ``vm.nodes[vm.position_of_error]`` when the VM names a position
                for c in clear:
                    c[0] = None
                pos = vm.position_of_error = self._locate_error(e)
                n = order[pos]
                vm.thunks[pos] = _NodeCells([[_describe(smap[i][0])] for i in n.inputs],
                                            [smap[o] for o in n.outputs])
                raise

        for attr in ("jit_fn", "allow_gc", "storage_map"):
            setattr(vm, attr, getattr(slow_fn, attr))
        # one entry per Apply node of the schedule (what ``position_of_error`` indexes); the
        # "thunk" of a node only lends ``raise_with_op`` its input cells (shapes / strides)
        order = list(self._order)
        smap = slow_fn.storage_map
        vm.nodes = order
        vm.thunks = [_NodeCells([smap[i] for i in n.inputs], [smap[o] for o in n.outputs])
                     for n in order]
        vm.jit_thunk = thunk
        vm.slow_vm = slow_fn
        vm.update_profile = self._update_profile
        return vm

    def _no_recycling_cells(self, slow_fn):
        nr = self.no_recycling
        smap = slow_fn.storage_map
        if nr is True:
            return list(smap.values())
        return [smap[r] for r in (nr or ()) if r not in self.fgraph.inputs]

    def _apply_position(self, ex, si):
        """Executor step -> position of its Apply node in the schedule (None if unknown)."""
        origin = getattr(self.plan, "var_origin", None) or {}
        st = ex.steps[si]
        outs = list(st.outputs) + ([st.reduce["out"]] if st.reduce is not None else [])
        for o in outs:
            if o in origin:
                return origin[o]
        return None

    def _locate_error(self, e):
        hs = getattr(e, "hip_step", None)
        pos = None
        if hs is not None and hs[0] is getattr(self, "executor", None):
            pos = self._apply_position(*hs)
        if pos is None:                      # unknown: the node of the first computed output
            owners = [o.owner for o in self.fgraph.outputs if o.owner is not None]
            pos = self._order.index(owners[0]) if owners and owners[0] in self._order else 0
        return pos

    def _update_profile(self, profile):
        """``VM.update_profile`` (link/vm.py:251): per-Apply device time (HIP events around the
        executor's steps; a fused step is booked on the node that produces its result) and call
        counts.  Nodes that became views / host shape arithmetic cost no device time."""
        ex = getattr(self, "executor", None)
        if ex is None or not getattr(ex, "profiling", False):
            return
        fg = self.fgraph
        for si, (t, c) in enumerate(zip(ex.step_time, ex.step_count)):
            pos = self._apply_position(ex, si)
            if pos is None or not c:
                continue
            node = self._order[pos]
            profile.apply_time[(fg, node)] = profile.apply_time.get((fg, node), 0.0) + t
            profile.apply_callcount[(fg, node)] = profile.apply_callcount.get((fg, node), 0) + c
            profile.apply_cimpl[node] = False
            ex.step_time[si], ex.step_count[si] = 0.0, 0

    def create_thunk_inputs(self, storage_map):
        return [storage_map[n] for n in self.fgraph.inputs]

    def output_filter(self, var, out):
        # return_numpy=True: results handed to the caller become ndarrays, ``updates=`` results stay
        # device tensors (they go back into shared-variable cells); "all": those too — for host
        # ``aesara.shared`` state, e.g. the reference's own test-suite (tests/reference_suites.py)
        if _is_scalar_type(var):
            # a host value of the reference's ScalarType: ``np.int8(56)``, not a 0-d array
            if hasattr(out, "detach"):
                out = out.detach().cpu().numpy()
            import numpy as np
            return np.asarray(out, dtype=var.type.dtype)[()]
        if self.return_numpy and hasattr(out, "detach") \
                and (self.return_numpy == "all" or var not in getattr(self, "_update_outputs", ())):
            return out.detach().cpu().numpy()
        return out


def _destroyed_inputs(fgraph, order):
    """[(input position, variable holding the final content of that input's buffer)] for the
    graph inputs some Op of the schedule destroys (``destroy_map``, graph/destroyhandler.py:286).
    Chains of destructive Ops on the same buffer are followed; a destructive Op on a VIEW of an
    input is not (the view's buffer is the executor's own copy)."""
    owner_of = {v: i for i, v in enumerate(fgraph.inputs)}     # variable -> input whose buffer it is
    final = {}
    for node in order:
        for oi, iis in (getattr(node.op, "destroy_map", None) or {}).items():
            src = node.inputs[iis[0]]
            if src in owner_of and getattr(src.type, "ndim", None) == getattr(node.outputs[oi].type, "ndim", -1):
                owner_of[node.outputs[oi]] = owner_of[src]
                final[owner_of[src]] = node.outputs[oi]
    return sorted(final.items())


def _write_back(dst, src):
    """Leave ``src`` (a result of the call) in the caller's argument ``dst``."""
    import numpy as np
    if dst is None:
        return
    if hasattr(dst, "copy_") and hasattr(src, "detach"):       # device argument <- device result
        if tuple(dst.shape) == tuple(src.shape):
            dst.copy_(src)
        return
    if hasattr(src, "detach"):
        src = src.detach().cpu().numpy()
    if isinstance(dst, np.ndarray) and dst.flags.writeable and dst.shape == np.shape(src):
        np.copyto(dst, np.asarray(src), casting="unsafe")


def _is_scalar_type(var):
    from aesara.scalar.basic import ScalarType
    return isinstance(var.type, ScalarType)


class _ValueView:
    """What ``raise_with_op`` prints about a value (shape, strides, size) for a device tensor —
    ``torch.Tensor.size`` is a method, ``raise_with_op`` compares it with an int."""

    def __init__(self, t):
        self.shape = tuple(t.shape)
        self.strides = tuple(s * t.element_size() for s in t.stride())
        self.size = t.numel()
        self.dtype = str(t.dtype).replace("torch.", "")
        self.device = str(t.device)

    def __repr__(self):
        return f"<{self.dtype}{list(self.shape)} on {self.device}>"


def _describe(v):
    return _ValueView(v) if hasattr(v, "element_size") and callable(getattr(v, "size", None)) else v


class _NodeCells:
    """The storage cells of one Apply node, in the shape ``raise_with_op`` reads from a thunk."""
    __slots__ = ("inputs", "outputs")

    def __init__(self, inputs, outputs):
        self.inputs, self.outputs = inputs, outputs


hip_linker = HipLinker()
hip_mode = Mode(hip_linker, HIP_QUERY)

_registered = False


def register():
    """Idempotently register linker ``"hip"`` and mode ``"HIP"`` with the reference."""
    global _registered
    if _registered:
        return hip_mode
    try:
        register_linker("hip", hip_linker)
    except ValueError:
        pass
    try:
        register_mode("HIP", hip_mode)
    except ValueError:
        pass
    _registered = True
    return hip_mode
