"""The reference's OWN test files, unmodified, with the default compilation mode = HIP.

``tests/reference_suites.py`` subclasses a handful of the reference's backend-parameterised test
classes; this module goes the whole way for the files that cover SURVEY §8's hot path: every test
of ``tests/tensor/test_subtensor.py`` (TestSubtensor / TestIncSubtensor / TestIncSubtensor1 /
TestAdvancedSubtensor ...), ``test_basic.py`` (TestAlloc, TestJoinAndSplit ...), ``test_blas.py``
(TestGemm, Gemv, Ger, BlasStrides, ``test_batched_dot*``), ``test_math.py`` (TestDot,
TestMaxAndArgmax, the elementwise / reduction makers), ``test_special.py`` (Softmax family),
``test_extra_ops.py`` (TestCumOp ...), ``test_elemwise.py``, ``test_shape.py`` and
``tests/scan/test_basic.py`` (TestScan, TestExamples, TestGradUntil) is collected from the overlay
of the reference (``oracle/ref_overlay.py``) by a child ``pytest`` whose plugin
(``tests/hip_suite_plugin.py``) sets ``config.mode`` to the HIP linker before the test modules
are imported.  No test body is edited, wrapped or re-stated.

A test of those files that does not pass must be explained by exactly one of:

* ``environment``   it fails the same way with the reference's OWN default mode (C linker) in this
                    image (NumPy 2, ...): ``tests/golden/reference_files_env.json``, written by
                    ``python tests/reference_files.py --record-environment`` in the authoring
                    container (script = this file);
* ``out_of_scope``  the lowering refused the graph with ``UnsupportedOp``: a complex / float16
                    dtype, an Op without a HIP lowering (outside §8a: Choose, Repeat,
                    Unique, SearchsortedOp, ...), a RandomVariable / non-tensor graph input;
* ``not_applicable`` listed in ``NOT_APPLICABLE`` below with its reason (assertions about the C
                    implementation or about destroy-map nodes the HIP rewrite query excludes by
                    design — in-place is the executor's buffer planner here).

Anything else is a failure of the HIP path and fails the calling test.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

FILES = [
    "tests/tensor/test_subtensor.py",
    "tests/tensor/test_basic.py",
    "tests/tensor/test_blas.py",
    "tests/tensor/test_special.py",
    "tests/tensor/test_extra_ops.py",
    "tests/tensor/test_elemwise.py",
    "tests/tensor/test_shape.py",
    "tests/tensor/test_math.py",
    "tests/scan/test_basic.py",
    "tests/scan/test_rewriting.py",
    "tests/scan/test_views.py",
    # H0 / f2: Function.__call__'s protocol, In / Out / updates / givens, shared variables
    "tests/compile/function/test_types.py",
    "tests/compile/function/test_pfunc.py",
    "tests/compile/function/test_function.py",
    "tests/compile/test_shared.py",
    "tests/tensor/test_sharedvar.py",
    # the TensorVariable sugar (indexing, operators, reductions), casting, keepdims, gradients
    "tests/tensor/test_var.py",
    "tests/tensor/test_casting.py",
    "tests/tensor/test_keepdims.py",
    "tests/tensor/nnet/test_basic.py",
    "tests/test_gradient.py",
    "tests/test_rop.py",
    # round 6: lazy IfElse, Sort / ArgSort, the Elemwise fusion the generated kernels start from,
    # Scan with checkpoints, Assert, OrderedUpdates, xlogx, the erf / gamma family, compile/ops
    "tests/test_ifelse.py",
    "tests/tensor/test_sort.py",
    "tests/tensor/rewriting/test_elemwise.py",
    "tests/scan/test_checkpoints.py",
    "tests/test_raise_op.py",
    "tests/test_updates.py",
    "tests/tensor/test_xlogx.py",
    "tests/tensor/test_math_scipy.py",
    "tests/compile/test_ops.py",
    # the in-place Elemwise variants on mutable HOST arguments (destroyed inputs are written back)
    "tests/tensor/test_inplace.py",
    # round 6 (second batch): the graph rewrites in front of the lowering, evaluated through it —
    # the canonicalize / stabilize / specialize passes the HIP query shares with FAST_RUN
    "tests/tensor/rewriting/test_basic.py",
    "tests/tensor/rewriting/test_math.py",
    "tests/tensor/rewriting/test_subtensor.py",
    "tests/tensor/rewriting/test_shape.py",
    "tests/tensor/rewriting/test_special.py",
    "tests/tensor/rewriting/test_uncanonicalize.py",
    "tests/tensor/rewriting/test_extra_ops.py",
    # OpFromGraph (expanded at lowering), scalar Ops / Composite, sigmoid rewrites
    "tests/compile/test_builders.py",
    "tests/scalar/test_basic.py",
    "tests/scalar/test_math.py",
    "tests/tensor/nnet/test_sigm.py",
    # round 6 (third batch): batch normalisation as Elemwise graphs, the BLAS rewrites as the C / SciPy
    # test files state them (Gemv / Ger here), test values, printing of compiled graphs
    "tests/tensor/nnet/test_batchnorm.py",
    "tests/tensor/test_blas_c.py",
    "tests/tensor/test_blas_scipy.py",
    "tests/graph/test_compute_test_value.py",
    "tests/tensor/test_utils.py",
    "tests/tensor/test_gc.py",
    "tests/tensor/test_misc.py",
    "tests/tensor/test_io.py",
    "tests/tensor/test_type_other.py",
    "tests/tensor/nnet/test_rewriting.py",
    "tests/compile/test_misc.py",
    "tests/test_printing.py",
]
ENV_FILE = os.path.join(HERE, "golden", "reference_files_env.json")

_INPLACE = ("asserts a destroy-map (in-place) node in the rewritten graph: the HIP rewrite query "
            "excludes `inplace` by design — in-place updates are the executor's buffer planner")
_HOSTVIEW = ("asserts that the RESULT aliases / is the HOST argument (view_map on an input): "
             "arguments are uploaded to HBM, results are device buffers")
_ASVIEW = ("asserts an `if{inplace}` node (IfElse(as_view=True), ifelse.py:93: the `cond_make_inplace` rewrite "
           "is tagged `inplace`, which the HIP rewrite query excludes by design); the VALUES of these "
           "graphs are covered by the other tests of the class")
# nodeid -> reason (covers every parametrisation of the test; a trailing * makes it a prefix)
NOT_APPLICABLE = {
    "tests/tensor/test_subtensor.py::TestSubtensor::test_grad_list": _INPLACE,
    "tests/tensor/test_subtensor.py::TestAdvancedSubtensor::test_adv_sub_slice":
        "a SliceType variable as a FUNCTION INPUT (a Python slice passed at call time): graph "
        "inputs of the plan are tensors (SURVEY §8b)",
    "tests/tensor/test_blas.py::TestGer::test_f32_*": _INPLACE,
    "tests/tensor/test_blas.py::TestGer::test_inplace": _INPLACE,
    "tests/tensor/test_blas.py::TestGer::test_outer": _INPLACE,
    "tests/tensor/test_blas.py::TestGemv::test_dot_*": _INPLACE,
    "tests/tensor/test_blas.py::TestGemv::test_gemv*": _INPLACE,
    "tests/tensor/test_blas.py::TestGemm::test_factorised_scalar": _INPLACE,
    "tests/tensor/test_blas.py::TestSgemv::test_default_beta_y": _INPLACE,
    "tests/tensor/test_blas.py::TestDgemv::test_default_beta_y": _INPLACE,
    "tests/tensor/test_blas.py::TestSgemv::test_upcasting_scalar_nogemv": _INPLACE,
    "tests/tensor/test_blas.py::TestDgemv::test_upcasting_scalar_nogemv": _INPLACE,
    "tests/tensor/test_basic.py::test_join_inplace": _HOSTVIEW,
    "tests/tensor/test_basic.py::TestTriangle::test_tri":
        "passes None as the VALUE of an int32 scalar input (the reference forwards None to np.tri)",
    "tests/tensor/test_basic.py::test_eye":
        "passes None as the VALUE of an int32 scalar input (the reference forwards None to np.eye)",
    "tests/scan/test_rewriting.py::TestScanInplaceOptimizer::test_inplace3": _INPLACE,
    "tests/scan/test_rewriting.py::TestScanInplaceOptimizer::test_simple_rnn": _INPLACE,
    "tests/scan/test_rewriting.py::TestScanInplaceOptimizer::test_simple_rnn_2": _INPLACE,
    "tests/tensor/test_sharedvar.py::TestSharedOptions::test_specify_shape_inplace": _INPLACE,
    "tests/tensor/test_blas_scipy.py::TestScipyGer::test_outer": _INPLACE,                        # ScipyGer{destructive}
    "tests/test_printing.py::test_debugprint":
        "asserts the TEXT of a debugprint of the C linker's graph (`CGemv{inplace}`): the HIP query has neither "
        "the C BLAS Ops nor in-place nodes",
    "tests/tensor/rewriting/test_math.py::test_log1p": _INPLACE,                                   # log1p_inplace, neg_inplace
    "tests/tensor/rewriting/test_math.py::TestSigmoidRewrites::test_exp_over_1_plus_exp": _INPLACE,  # neg_inplace
    "tests/tensor/rewriting/test_subtensor.py::TestLocalSubtensorLift::test_basic_5": _INPLACE,    # exp_inplace
    "tests/tensor/rewriting/test_subtensor.py::TestSubtensorIncSubtensor::test_inplace": _INPLACE,  # mode.including("inplace")
    "tests/compile/function/test_pfunc.py::TestAliasingRules::test_no_aliasing_2b":
        "asserts that two updated shared variables end up as VIEWS of each other's HOST buffers "
        "(no copy): shared values updated on the device are device buffers",
    "tests/compile/function/test_types.py::TestFunction::test_constant_output":
        "asserts that a borrowed CONSTANT output is the constant's own host ndarray (a write by the "
        "caller shows in the next call): results are device buffers converted per call",
    "tests/compile/function/test_types.py::TestPicklefunction::test_deepcopy_trust_input":
        "expects whatever the C thunk raises for a wrongly typed argument under trust_input=True "
        "(unchecked by definition); the HIP call converts the value",
    "tests/test_ifelse.py::TestIfelse::test_lazy_if": _ASVIEW,
    "tests/test_ifelse.py::TestIfelse::test_mixed_dtype": _ASVIEW,
    "tests/test_ifelse.py::TestIfelse::test_multiple_out": _ASVIEW,
    "tests/test_ifelse.py::TestIfelse::test_multiple_out_crash": _ASVIEW,
    "tests/test_ifelse.py::TestIfelse::test_grad_lazy_if": _ASVIEW,
    "tests/tensor/test_extra_ops.py::TestUnique::test_basic_vector[x1-inp1-None]":
        "compares the INVERSE of a matrix with NumPy 2's n-d inverse (np.unique(..., axis=None) returns it in "
        "the input's shape since NumPy 2.0); the Op declares a VECTOR (extra_ops.py:1199, reference pins "
        "numpy < 2) and the HIP lowering returns the flat inverse the type promises",
    "tests/tensor/test_math_scipy.py::TestSigmoidInplaceBroadcast::test_good":
        "truncates sigmoid(int8) INTO int8 and compares with SciPy's expit of an int8 array, whose precision is "
        "SciPy's choice (float64 in SciPy 1.15: expit(20) = 0.999... -> 0); the Op's own type rule "
        "(upgrade_to_float, scalar/basic.py:790) says float32 (expit(20) = 1.0f -> 1), which is what the HIP kernel "
        "computes in.  The non-in-place int8 case fails with the reference's own linker here (environment list)",
    "tests/tensor/test_extra_ops.py::TestSearchsortedOp::test_searchsortedOp_on_right_side":
        "searches an UNSORTED sequence (self.a is not sorted) and compares with the positions NumPy's "
        "SEQUENTIAL loop happens to return: npy_binsearch carries the previous key's bounds over to the next "
        "key, so on unsorted input the result depends on the order the keys are visited; the kernel runs one "
        "independent binary search per key (identical on sorted input, which is the Op's contract)",
    "tests/tensor/test_math_scipy.py::TestGammaUInplaceBroadcast::test_good":
        "random small-integer arguments: whenever k = x + 1 the continued fraction of gamma.c (upperGamma, the "
        "C body of GammaU) starts with 1 / (x + 1 - k) = 1 / 0 and ends in NaN, in the reference's C linker "
        "and in the kernel alike, while the test expects SciPy's value — it passes or fails with the draw",
    "tests/scan/test_basic.py::TestScan::test_monitor_mode":
        "MonitorMode hooks the per-node thunks of the C / Python VM; HipLinker runs one thunk",
}

_RULES = [
    (re.compile(r"UnsupportedOp: .*dtype (complex64|complex128|float16)"), "out_of_scope",
     "complex / float16 dtype (SURVEY §2)"),
    (re.compile(r"UnsupportedOp: .*non-tensor variable type Random"), "out_of_scope",
     "RandomVariable / RNG state (outside §8)"),
    (re.compile(r"UnsupportedOp: .*non-tensor variable type (Generic|Sparse)"), "out_of_scope",
     "a Generic (arbitrary Python object) / sparse variable in the graph (SURVEY §2)"),
    (re.compile(r"UnsupportedOp: .*non-tensor variable type SliceType"), "out_of_scope",
     "a SliceType variable at a graph boundary (inputs / outputs of a plan are tensors, SURVEY §8b)"),
    (re.compile(r"UnsupportedOp: (\w+) has no HIP lowering"), "out_of_scope_op",
     "an Op outside SURVEY §8a"),
    (re.compile(r"UnsupportedOp: .*scalar op (Complex\w*|Real|Imag|Angle|Conj) is outside the HIP hot path"),
     "out_of_scope", "complex scalar op (SURVEY §2)"),
    (re.compile(r"UnsupportedOp: .*scalar op (BetaInc|BetaIncDer|GammaIncDer|GammaIncCDer|Hyp2F1|Hyp2F1Der|Iv|Jv|"
                r"Owens_t) is outside the HIP hot path"), "out_of_scope",
     "a scipy.special routine the reference itself evaluates only through SciPy in Python (its c_code raises "
     "NotImplementedError, scalar/math.py:276/746/829/940/1031/1290/1478/1510/1655): no compiled path to replace"),
    (re.compile(r"UnsupportedOp: scalar op (TimesN) is outside the HIP hot path"), "out_of_scope",
     "a toy ScalarOp defined inside tests/tensor/rewriting/test_elemwise.py (c_code only)"),
    (re.compile(r"UnsupportedOp: CAReduce over scalar op (mean)"), "out_of_scope",
     "the legacy Mean(CAReduce) Op, tensor/math.py:1495 (at.mean() builds Sum / true_div, which is lowered)"),
]


# Ops whose ``UnsupportedOp: <Name> has no HIP lowering`` is an explanation: subsystems SURVEY §2
# marks out of scope.  ANY OTHER Op name in that message is UNEXPLAINED — a lowering that is
# dropped by a regression fails the run instead of being filed here.
OUT_OF_SCOPE_OPS = {
    # tensor/nnet conv / pooling / block-sparse (SURVEY §2 row 14, §8(f)4 "conv last")
    "Convolve": "tensor/signal + nnet convolution", "Pool": "tensor/signal/pool.py",
    "SparseBlockGemv": "tensor/nnet/blocksparse.py", "SparseBlockOuter": "tensor/nnet/blocksparse.py",
    "CorrMM": "nnet/corr.py", "CorrMM_gradWeights": "nnet/corr.py", "CorrMM_gradInputs": "nnet/corr.py",
    "BaseAbstractConv": "nnet/abstract_conv.py", "AbstractConv2d": "nnet/abstract_conv.py",
    "AbstractConv2d_gradWeights": "nnet/abstract_conv.py", "AbstractConv2d_gradInputs": "nnet/abstract_conv.py",
    # sparse: any Op of aesara/sparse
    "AddSS": "aesara/sparse", "CSM": "aesara/sparse", "DenseFromSparse": "aesara/sparse",
    "SparseFromDense": "aesara/sparse", "StructuredDot": "aesara/sparse",
    # typed_list
    "MakeList": "aesara/typed_list", "GetItem": "aesara/typed_list", "Append": "aesara/typed_list",
    # test-local toy Ops of the reference's own test files (perform-only Python Ops defined in
    # the test module; no tensor semantics to lower)
    "DotModulo": "toy Op of tests/tensor/test_math.py", "BreakRop": "toy Op of tests/test_rop.py",
    "IdentityNoShape": "toy Op of tests/tensor/rewriting/test_shape.py",
    # linalg
    "MatrixInverse": "tensor/nlinalg.py", "Det": "tensor/nlinalg.py", "SVD": "tensor/nlinalg.py",
    "Cholesky": "tensor/slinalg.py", "Solve": "tensor/slinalg.py", "Eigh": "tensor/nlinalg.py",
    "MatrixPinv": "tensor/nlinalg.py", "Eig": "tensor/nlinalg.py", "QRFull": "tensor/nlinalg.py",
    "SolveTriangular": "tensor/slinalg.py",
}


def overlay_dir():
    import ref_overlay
    if ref_overlay.source() in ("reference", "archive"):
        # builds / unpacks the overlay as a side effect of the first import, in a child so that
        # this process does not import the front end
        subprocess.run([sys.executable, "-c", "import sys; sys.path[:0] = %r; import ref_overlay; "
                        "ref_overlay.import_reference()" % [os.path.join(ROOT, "oracle")]],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return ref_overlay.OVERLAY


def run(executor, files=None, workers=4, timeout=3000, extra=()):
    """One child ``pytest`` over ``files`` of the overlay with the plugin; -> {nodeid: [outcome,
    message]}.  ``executor``: device | oracle | prebuild | reference (hip_suite_plugin.py)."""
    files = list(files or FILES)
    fd, rep = tempfile.mkstemp(prefix="hip_suite_", suffix=".json")
    os.close(fd)
    env = dict(os.environ, AESARA_HIP_SUITE_EXECUTOR=executor, AESARA_HIP_SUITE_REPORT=rep)
    env["PYTHONPATH"] = HERE + os.pathsep + env.get("PYTHONPATH", "")
    env.pop("AESARA_FLAGS", None)
    # (--timeout: pytest-timeout — a test that hangs on the device is reported as a failure of THAT
    # test instead of costing the whole run its time limit)
    cmd = [sys.executable, "-m", "pytest", "-p", "hip_suite_plugin", "-q", "--tb=no", "-p",
           "no:cacheprovider", "-W", "ignore", "--timeout", os.environ.get("AESARA_HIP_SUITE_TEST_TIMEOUT", "600")] + \
        (["-n", str(workers)] if workers > 1 else []) + list(extra) + files
    try:
        p = subprocess.run(cmd, cwd=overlay_dir(), env=env, capture_output=True, text=True,
                           timeout=timeout)
        with open(rep) as f:
            txt = f.read()
        if not txt:
            raise RuntimeError("the child pytest wrote no report:\n" + (p.stdout + p.stderr)[-3000:])
        return json.loads(txt)
    finally:
        try:
            os.unlink(rep)
        except OSError:
            pass


def environment_failures():
    if not os.path.exists(ENV_FILE):
        return {}
    with open(ENV_FILE) as f:
        return json.load(f)["failed"]


# messages that are the ENVIRONMENT's whatever the linker under test (checked before everything else)
_ENV_RULES = [
    (re.compile(r"Key not found in unpickled KeyData file"),
     "race in the REFERENCE'S OWN C-module cache (link/c/cmodule.py: key.pkl written by one pytest worker "
     "while another reads it); the test compiles a comparison function with the C linker"),
]


def classify(report):
    """-> (summary dict, unexplained {nodeid: message}).  A report entry is ``[outcome, message]``
    or ``[outcome, message, [compiled, executed]]`` (functions the test compiled through
    ``HipLinker`` / executor calls it made: ``hip_suite_plugin``)."""
    env = environment_failures()
    counts = {"passed": 0, "skipped": 0, "xfailed": 0, "xpassed": 0, "environment": 0,
              "out_of_scope": 0, "not_applicable": 0, "unexplained": 0,
              # of the PASSED tests: compiled >= 1 function through HipLinker / also ran one
              "through_hip": 0, "executed_hip": 0}
    detail = {"out_of_scope": {}, "not_applicable": {}, "environment": {}}
    bad = {}
    per_file = {}
    for nid, entry in sorted(report.items()):
        outcome, msg = entry[0], entry[1]
        hip = entry[2] if len(entry) > 2 else [0, 0]
        fkey = nid.split("::")[0]
        pf = per_file.setdefault(fkey, {"passed": 0, "not_passed": 0, "through_hip": 0,
                                        "executed_hip": 0})
        if outcome != "failed":
            counts[outcome] = counts.get(outcome, 0) + 1
            if outcome == "passed":
                pf["passed"] += 1
                if hip[0] > 0:
                    counts["through_hip"] += 1
                    pf["through_hip"] += 1
                    if hip[1] > 0:
                        counts["executed_hip"] += 1
                        pf["executed_hip"] += 1
            continue
        pf["not_passed"] += 1
        envr = next((why for rx, why in _ENV_RULES if rx.search(msg)), None)
        if envr is not None:
            counts["environment"] += 1
            detail["environment"][nid] = envr[:120]
            continue
        na = next((why for pre, why in NOT_APPLICABLE.items()
                   if (nid.startswith(pre[:-1]) if pre.endswith("*") else nid == pre or nid.startswith(pre + "["))),
                  None)
        if na is not None:
            counts["not_applicable"] += 1
            detail["not_applicable"][nid] = na
            continue
        rule = next(((cat, why, m) for rx, cat, why in _RULES for m in [rx.search(msg)] if m), None)
        if rule is not None and rule[0] == "out_of_scope_op" and rule[2].group(1) not in OUT_OF_SCOPE_OPS:
            rule = None                     # an Op that is not on the allow-list: unexplained
        if rule is not None:
            counts["out_of_scope"] += 1
            why = rule[1] + (": " + rule[2].group(1) if rule[2].groups() else "")
            if rule[0] == "out_of_scope_op":
                why += " (%s)" % OUT_OF_SCOPE_OPS[rule[2].group(1)]
            detail["out_of_scope"][why] = detail["out_of_scope"].get(why, 0) + 1
            continue
        if nid in env:
            counts["environment"] += 1
            detail["environment"][nid] = env[nid][:120]
            continue
        counts["unexplained"] += 1
        bad[nid] = msg
    return {"counts": counts, "per_file": per_file, "detail": detail}, bad


def format_summary(executor, summary, bad):
    c = summary["counts"]
    lines = ["reference test files under the default mode HIP (executor: %s)" % executor,
             "  passed %d, of which THROUGH HipLinker %d (compiled >= 1 function with it; %d of "
             "those also executed it) | skipped %d | xfailed %d | xpassed %d" % (
                 c["passed"], c["through_hip"], c["executed_hip"], c["skipped"], c["xfailed"],
                 c["xpassed"]),
             "  (a passing test that compiles nothing — symbolic dtype / shape checks — or names its "
             "own linker proves nothing about the HIP path: it is in the first number only)",
             "  not passed, explained: environment %d (fail with the reference's own C linker here "
             "too) | out of scope %d | not applicable %d" % (
                 c["environment"], c["out_of_scope"], c["not_applicable"]),
             "  UNEXPLAINED: %d" % c["unexplained"],
             "  per file (passed / through HipLinker / executed on it / not passed):"]
    for f, v in sorted(summary["per_file"].items()):
        lines.append("    %-42s %5d / %5d / %5d / %d" % (f, v["passed"], v["through_hip"],
                                                        v["executed_hip"], v["not_passed"]))
    lines.append("  out of scope, by reason:")
    for why, n in sorted(summary["detail"]["out_of_scope"].items(), key=lambda kv: -kv[1]):
        lines.append("    %4d  %s" % (n, why))
    lines.append("  not applicable:")
    for nid, why in sorted(summary["detail"]["not_applicable"].items()):
        lines.append("    %s — %s" % (nid, why))
    lines.append("  environment:")
    for nid, why in sorted(summary["detail"]["environment"].items()):
        lines.append("    %s — %s" % (nid, why.replace("\n", " ")))
    for nid, msg in sorted(bad.items()):
        lines.append("  UNEXPLAINED %s: %s" % (nid, msg.replace("\n", " ")[:300]))
    return "\n".join(lines)


def check(executor, files=None, workers=4, log_path=None, timeout=3000):
    rep = run(executor, files, workers, timeout)
    summary, bad = classify(rep)
    text = format_summary(executor, summary, bad)
    if log_path:
        os.makedirs(os.path.dirname(log_path), exist_ok=True)
        with open(log_path, "w") as f:
            f.write(text + "\n")
    return summary, bad, text


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--record-environment", action="store_true",
                    help="run the files with the reference's own default mode and write the tests "
                         "that fail there to tests/golden/reference_files_env.json")
    ap.add_argument("--executor", default="oracle")
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--log", default="")
    ap.add_argument("files", nargs="*")
    a = ap.parse_args()
    if a.record_environment:
        rep = run("reference", a.files or None, a.workers)
        failed = {k: v[1][:200] for k, v in rep.items() if v[0] == "failed"}
        by_file = {}
        for k in rep:
            by_file[k.split("::")[0]] = by_file.get(k.split("::")[0], 0) + 1
        if a.files and os.path.exists(ENV_FILE):       # a partial run extends the record
            with open(ENV_FILE) as f:
                old = json.load(f)
            failed = dict({k: v for k, v in old["failed"].items()
                           if k.split("::")[0] not in by_file}, **failed)
            by_file = dict(old.get("collected_by_file", {"(files recorded before round 6)": old["collected"]}),
                           **by_file)
        n_collected = sum(by_file.values())
        with open(ENV_FILE, "w") as f:
            json.dump({"what": "tests of the reference's own files that fail with the reference's "
                               "own default mode (C linker) in this image — python "
                               "tests/reference_files.py --record-environment",
                       "collected": n_collected, "collected_by_file": by_file, "failed": failed}, f, indent=1, sort_keys=True)
        print("recorded %d environment failures of %d tests" % (len(failed), n_collected))
    else:
        s, bad, text = check(a.executor, a.files or None, a.workers, a.log or None)
        print(text)
        sys.exit(1 if bad else 0)
