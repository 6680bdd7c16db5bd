"""aesara_amd — MI355X-native (gfx950) execution backend for the Aesara Linker hot path.

Layers (bottom up):

* ``csrc/``      hand-written HIP kernels + the C-ABI shim ``libaesara_hip.so`` (include/aesara_hip.h)
* ``_lib``       ctypes binding of that C-ABI (fails loudly when the library is missing)
* ``codegen``    HIP source generator for fused broadcast Elemwise(+CAReduce) kernels
* ``executor``   runs a ``Plan`` on device arrays (PyTorch-ROCm tensors are only the container)
* ``plan``       backend-neutral launch-plan IR (plain data, JSON)
* ``lower`` / ``linker``  Aesara FunctionGraph -> Plan and the ``HipLinker(JITLinker)`` plug-in
  (these two import the reference front end lazily; the rest runs without it)
"""
__version__ = "0.1.0"

from .plan import Node, Plan, Var  # noqa: F401


def get_mode():
    """Return the registered Aesara ``Mode`` for the HIP linker (needs ``aesara``)."""
    from .linker import register

    return register()
