"""Device-resident shared state on the GPU: the `check_blas.py:54-57` pattern
``updates=[(C, 0.4*C + 0.8*dot(A, B))]`` driven the way ``Function.__call__`` drives it
(compile/function/types.py:1060-1069: ``storage.data = outputs.pop()``) through the HIP executor
and ``devcell.DeviceCell`` — the state never leaves HBM and equals the oracle after 100 steps."""
import numpy as np
import pytest

from golden_util import CASES, case_plan

pytestmark = pytest.mark.gpu


class _Fn:
    """The trust_input call protocol of ``Function.__call__`` around one executor: bind inputs
    into their cells, run, write the update outputs back into the shared cells."""

    def __init__(self, executor, in_cells, update_cells):
        self.ex, self.in_cells, self.update_cells = executor, in_cells, update_cells

    def __call__(self, *args):
        for c, a in zip(self.in_cells, args):          # types.py:838-843 (trust_input)
            c.storage[0] = a
        outputs = list(self.ex(*[c.storage[0] for c in self.in_cells + self.update_cells]))
        for cell in reversed(self.update_cells):       # types.py:1060-1069
            cell.data = outputs.pop()
        return outputs


@pytest.mark.parametrize("use_graph", [False, True])
def test_gemm_update_state_stays_on_device_100_steps(use_graph):
    import interp
    import torch
    from aesara_amd.devcell import DeviceCell, PlainType
    from aesara_amd.executor import PlanExecutor
    plan = case_plan(next(c for c in CASES if c["name"] == "cfg3b_gemm_update"))   # (C, A, B) -> C'
    rng = np.random.default_rng(0)
    n = 96
    C0 = rng.standard_normal((n, n)).astype(np.float32)
    A = (rng.standard_normal((n, n)) * 0.1).astype(np.float32)
    B = (rng.standard_normal((n, n)) * 0.1).astype(np.float32)
    typ = PlainType("float32", (None, None))
    cell = DeviceCell(typ, name="C", device=torch.device("cuda"))
    cell.value = C0                                    # host value: filtered + uploaded ONCE
    assert isinstance(cell.storage[0], torch.Tensor) and cell.storage[0].is_cuda
    ex = PlanExecutor(plan, use_graph=use_graph)
    # plan inputs are (C, A, B): C is the shared cell, A / B ordinary inputs
    a_cell, b_cell = DeviceCell(typ, name="A"), DeviceCell(typ, name="B")
    f = _Fn(lambda A_, B_, C_: ex(C_, A_, B_), [a_cell, b_cell], [cell])
    Ad, Bd = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    want = C0.copy()
    for step in range(100):
        assert f(Ad, Bd) == []
        v = cell.storage[0]
        assert isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.float32
        (want,) = interp.run_plan(plan, [want, A, B])
        if step in (0, 1, 9, 99):
            np.testing.assert_allclose(v.cpu().numpy(), want, rtol=2e-5, atol=1e-6)
    # the cell refuses a wrong device value exactly like TensorType.filter would
    with pytest.raises(TypeError):
        cell.value = torch.zeros(n, dtype=torch.float32, device="cuda")
    with pytest.raises(TypeError):
        cell.value = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    cell.value = np.zeros((n, n), np.float32)          # host data still goes through the filter
    assert cell.storage[0].is_cuda and float(cell.storage[0].abs().sum()) == 0.0
