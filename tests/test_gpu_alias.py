"""Buffer ownership on the GPU: in-place updates never corrupt live views;
replayed calls hand out fresh outputs unless borrowed and still raise on a bad index (medium)."""
import numpy as np
import pytest

from test_host_logic import _alias_plan

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("view_is_output", [True, False])
@pytest.mark.parametrize("use_graph", [False, True])
def test_view_of_updated_alloc_keeps_its_values(view_is_output, use_graph):
    from aesara_amd.executor import PlanExecutor
    ex = PlanExecutor(_alias_plan(view_is_output), use_graph=use_graph)
    y = np.array([7.0, 8.0, 9.0])
    for _ in range(3):
        a, r = [o.cpu().numpy() for o in ex(np.float64(1.5), np.int64(4), y)]
        z = np.full((4, 3), 1.5)
        want_r = z.copy()
        want_r[0] = y
        np.testing.assert_array_equal(r, want_r)
        np.testing.assert_array_equal(a, z.T if view_is_output else z.T.sum(axis=0))


def test_replay_outputs_are_fresh_unless_borrowed():
    import torch
    from golden_util import CASES, case_plan
    from aesara_amd.executor import PlanExecutor
    plan = case_plan(next(c for c in CASES if c["name"] == "cfg1b_matrix_add"))
    x = torch.ones(64, 64, dtype=torch.float64, device="cuda")
    for borrow in (False, True):
        ex = PlanExecutor(plan, use_graph=True, borrow=borrow)
        outs = []
        for k in range(4):
            y = torch.full((64, 64), float(k), dtype=torch.float64, device="cuda")
            outs.append(ex(x, y)[0])
        torch.cuda.synchronize()
        if borrow:      # function-owned buffer: later calls overwrite what was handed out
            assert len({o.data_ptr() for o in outs[1:]}) < 3
        else:           # every call's result survives the following calls
            for k, o in enumerate(outs):
                assert float(o[0, 0]) == 1.0 + k, (k, float(o[0, 0]))
            # ... and without a device copy: the recorded launch is re-pointed at a newly
            # allocated tensor (the output is a rebindable range of the launch list)
            ents = list(ex._graphs.values())
            assert ents and all(e[8] for e in ents), [e[8] for e in ents]
            arena_ptr = ents[0][2][0].data_ptr()
            assert all(o.data_ptr() != arena_ptr for o in outs[1:])


def test_replayed_call_with_bad_index_raises():
    """Default: the IndexError is raised by the call that met the bad index (the reference's
    behaviour), names the step, and leaves nothing behind for later calls;
    ``check_indices="deferred"``: no stall, reported when the flag has landed."""
    import torch
    from golden_util import CASES, case_inputs, case_plan
    from aesara_amd.executor import PlanExecutor
    c = next(c for c in CASES if c["name"] == "advsub1")
    plan, ins = case_plan(c), case_inputs(c)
    ipos = next(k for k, a in enumerate(ins) if np.asarray(a).dtype.kind == "i" and np.asarray(a).ndim == 1)
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if np.asarray(a).ndim else a for a in ins]
    bad = dev[ipos].clone()
    bad[0] = 10 ** 6
    good = dev[ipos].clone()

    ex = PlanExecutor(plan, use_graph=True)
    for _ in range(3):
        want = [o.clone() for o in ex(*dev)]       # recorded, then replayed
    dev[ipos].copy_(bad)                            # same tensor, new values: the replay fast path
    with pytest.raises(IndexError, match=r"out of bounds\n\[HIP step \d+ of plan"):
        ex(*dev)                                    # raised HERE, not one call later
    dev[ipos].copy_(good)
    got = ex(*dev)                                  # the next (valid) call is not poisoned
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    ex.check()

    ex = PlanExecutor(plan, use_graph=True, check_indices="deferred")
    for _ in range(3):
        ex(*dev)
    dev[ipos].copy_(bad)
    ex(*dev)                                        # the check is deferred (no pipeline stall) ...
    with pytest.raises(IndexError):
        ex.check()                                  # ... and reported once the flag has landed
    dev[ipos].copy_(good)
    ex(*dev)
    ex.check()                                      # flag was reset
    dev[ipos].copy_(bad)
    ex(*dev)
    torch.cuda.synchronize()
    with pytest.raises(IndexError):                 # without check(): the next calls report it
        for _ in range(3):
            ex(*dev)
            torch.cuda.synchronize()


def test_rebound_replay_does_not_pin_generations_of_inputs():
    """fresh outputs make every step's updated state a NEW tensor; the replay cache
    must not keep those generations (or the batches) alive.  200 update steps of the
    check_blas pattern: device memory stays flat."""
    import torch
    from golden_util import CASES, case_plan
    from aesara_amd.executor import PlanExecutor
    plan = case_plan(next(c for c in CASES if c["name"] == "cfg3b_gemm_update"))   # (C, A, B) -> C'
    ex = PlanExecutor(plan, use_graph=True)
    n = 512                                                     # 1 MiB per matrix
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    C = torch.randn(n, n, device="cuda", generator=g)
    A = torch.randn(n, n, device="cuda", generator=g) * 0.03
    B = torch.randn(n, n, device="cuda", generator=g) * 0.03
    want = C.double()
    ab = A.double() @ B.double()
    base = None
    for step in range(200):
        (C,) = ex(C, A.clone(), B)                              # new state AND a new "batch" tensor
        want = 0.4 * want + 0.8 * ab
        if step == 20:
            torch.cuda.synchronize()
            base = torch.cuda.memory_allocated()
    torch.cuda.synchronize()
    grown = torch.cuda.memory_allocated() - base
    assert grown < 8 << 20, f"replay cache pinned {grown / 2**20:.1f} MiB of dead inputs"
    assert torch.allclose(C.double(), want, rtol=2e-5, atol=1e-5)
    assert len(ex._reloc) >= 1, "the loop was meant to take the rebinding path"
