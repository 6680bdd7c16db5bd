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


class HipLinker(JITLinker):
    """A ``Linker`` that runs a whole ``FunctionGraph`` as HIP kernels on an MI355X."""

    def __init__(self, *args, return_numpy=False, use_graph=True, executor_factory=None,
                 fast_call=True, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_numpy = return_numpy
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

    def __getstate__(self):
        # a pickled / deep-copied Function is re-linked when it is loaded (the reference pickles the
        # FunctionMaker, compile/function/types.py:1125 _pickle_Function): device executors,
        # module handles and the lowered plan are rebuilt then, never serialised
        state = dict(self.__dict__)
        for k in ("executor", "plan", "_update_outputs"):
            state.pop(k, None)
        state["plan"] = None
        return state

    # -- JITLinker API ---------------------------------------------------------------
    def fgraph_convert(self, fgraph, order=None, input_storage=None, output_storage=None,
                       storage_map=None, **kwargs):
        self.plan = lower_fgraph(fgraph, order=order, name=getattr(fgraph, "name", None)
                                 or "fgraph", inner_rewriter=hip_mode.optimizer)
        # outputs that are ``updates=`` expressions go back into shared-variable cells
        # (types.py:1060-1069), never to the caller: they stay device tensors
        self._update_outputs = {fgraph.outputs[i]
                                for i in getattr(fgraph, "update_mapping", None) or {}}
        return self.plan

    def jit_compile(self, plan):
        if self.executor_factory is not None:
            return self.executor_factory(plan)
        from .executor import PlanExecutor  # imports the C-ABI; fails loudly if missing

        ex = PlanExecutor(plan, use_graph=self.use_graph)
        self.executor = ex
        return ex

    def make_all(self, *args, **kwargs):
        """``JITLinker.make_all`` (link/basic.py:684-747) plus one change: the input cells handed
        to ``Function`` accept device tensors as they are (``sharedvar.DeviceContainer``), so an
        ``updates=`` output is stored back into its shared variable without leaving HBM."""
        from .sharedvar import DeviceContainer

        fn, ins, outs, thunks, nodes = super().make_all(*args, **kwargs)
        if self.fast_call and len(thunks) == 1:
            fn = self._fast_vm(fn, thunks[0])
        return fn, [DeviceContainer.adopt(c) for c in ins], outs, thunks, nodes

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
        plain = not self.return_numpy
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

        if plain and len(out_cells) == 1:
            oc = out_cells[0]

            def vm():
                oc[0] = call()[0]
        elif plain:
            def vm():
                for cell, val in zip(out_cells, call()):
                    cell[0] = val
        else:
            def vm():
                for var, cell, val in zip(out_vars, out_cells, call()):
                    cell[0] = ofilter(var, val)
        for attr in ("jit_fn", "allow_gc", "storage_map"):
            setattr(vm, attr, getattr(slow_fn, attr))
        vm.thunks, vm.nodes = [thunk], getattr(slow_fn, "nodes", None)
        vm.slow_vm = slow_fn
        return vm

    def create_thunk_inputs(self, storage_map):
        return [storage_map[n] for n in self.fgraph.inputs]

    def output_filter(self, var, out):
        if self.return_numpy and hasattr(out, "detach") \
                and var not in getattr(self, "_update_outputs", ()):
            return out.detach().cpu().numpy()
        return out


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
