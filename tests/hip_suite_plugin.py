"""pytest plugin (``-p hip_suite_plugin``): run the REFERENCE'S OWN test files, unmodified, with
the default compilation mode = the HIP linker.

TEST INFRASTRUCTURE.  Loaded only by ``tests/reference_files.py``'s child ``pytest`` processes
(their rootdir is the reference overlay, ``oracle/ref_overlay.py``); the product never imports it.

What it does, before any test module is imported (module-level ``mode_opt`` / ``get_default_mode()``
values must already see it): imports the reference front end from the overlay, registers the HIP
linker and sets ``config.mode`` to ``Mode(HipLinker(return_numpy="all"), HIP_QUERY)`` — every
``aesara.function(...)`` / ``inplace_func`` / ``verify_grad`` / ``self.mode = get_default_mode()``
of the reference's tests then compiles through ``HipLinker`` and gets ndarrays back for its own
assertions.

``AESARA_HIP_SUITE_EXECUTOR``:
* ``device``   the real ``PlanExecutor`` (``-m gpu`` runs on the MI355X);
* ``oracle``   ``oracle/interp.py`` evaluates the lowered plan (CPU container: checks the lowering
               of every graph the suites build and the harness itself);
* ``prebuild`` as ``oracle``, and the ``PlanExecutor`` replays its host logic in dry-run mode so
               that every generated kernel the suites need is cross-compiled into the in-tree
               kernel cache (no device needed; the GPU box then only loads code objects).

Outcomes are written to ``$AESARA_HIP_SUITE_REPORT`` (JSON: nodeid -> [outcome, message,
[functions compiled through HipLinker, executor calls] during the test's setup + call]).
"""
import json
import os
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import ref_overlay  # noqa: E402

KIND = os.environ.get("AESARA_HIP_SUITE_EXECUTOR", "oracle")
if KIND != "reference":
    ae = ref_overlay.import_reference()
    import aesara_amd  # noqa: E402

    aesara_amd.get_mode()
    from aesara.compile.mode import Mode  # noqa: E402
    from aesara.configdefaults import config  # noqa: E402
    from aesara_amd.linker import HIP_QUERY, HipLinker  # noqa: E402

    if KIND == "device":
        def HipL(**kw):
            return HipLinker(return_numpy="all", **kw)
    else:
        import interp  # noqa: E402

        def _factory(plan):
            dry = None
            if KIND == "prebuild":
                from aesara_amd.executor import PlanExecutor
                try:
                    dry = PlanExecutor(plan, dry_run=True)
                except Exception:                       # noqa: BLE001
                    dry = None

            def run(*a):
                if dry is not None:
                    try:
                        dry(*a)
                    except Exception:                   # noqa: BLE001  (a dry run cannot follow
                        if os.environ.get("AESARA_HIP_SUITE_DRY_STRICT"):   # data-dependent control flow)
                            raise
                    log = os.environ.get("AESARA_HIP_SUITE_SCANLOG")
                    if log and dry.scan_modes:
                        # (which Scans of the reference's tests take a one-launch kernel: one line per run)
                        import json
                        with open(log, "a") as f:
                            f.write(json.dumps([os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0],
                                                sorted(dry.scan_modes.values())]) + "\n")
                return interp.run_plan(plan, a)
            return run

        def HipL(**kw):
            return HipLinker(return_numpy="all", executor_factory=_factory, **kw)

    config.mode = Mode(HipL(), HIP_QUERY)

    # -- what a passing test PROVES: did it compile / run anything through HipLinker? -----------
    # (a test of these files that builds no function, or passes its own Mode(linker="py"), passes
    # without touching the HIP path: the summary counts those separately)
    _HIP = {"compiled": 0, "executed": 0}
    _jit_compile = HipLinker.jit_compile

    def _counting_jit_compile(self, plan):
        fn = _jit_compile(self, plan)
        _HIP["compiled"] += 1

        def counted(*a):
            _HIP["executed"] += 1
            return fn(*a)
        for attr in ("close", "check", "enable_profile"):
            if hasattr(fn, attr):
                setattr(counted, attr, getattr(fn, attr))
        counted.__wrapped__ = fn
        return counted

    if KIND != "device":
        HipLinker.jit_compile = _counting_jit_compile
    else:
        # on the device the executor object itself is what ``linker.executor`` / the fast VM hold:
        # it is marked, and counted in its own ``__call__`` (executors of inner plans are not)
        from aesara_amd.executor import PlanExecutor as _PE
        _pe_call = _PE.__call__

        def _marking_jit_compile(self, plan):
            ex = _jit_compile(self, plan)
            _HIP["compiled"] += 1
            try:
                ex._suite_counted = True
            except AttributeError:
                pass
            return ex

        def _call(self, *a, **k):
            if getattr(self, "_suite_counted", False):
                _HIP["executed"] += 1
            return _pe_call(self, *a, **k)
        HipLinker.jit_compile = _marking_jit_compile
        _PE.__call__ = _call
else:
    _HIP = {"compiled": 0, "executed": 0}
    # the same files with the reference's own default mode (C linker): which tests cannot pass in
    # this environment whatever the linker (NumPy 2, no pytest-benchmark ...)
    ae = ref_overlay.import_reference()

def _KeepdimsMode(*a, **k):
    """Stands in for ``aesara.compile.mode.Mode`` inside tests/tensor/test_keepdims.py only."""
    return config.mode


_IGNORED = [("numpy.core", DeprecationWarning), ("numpy._core", DeprecationWarning)]


def _filters():
    # several reference test modules run under ``filterwarnings("error")``; NumPy 2's own
    # deprecation notice about ``numpy.core`` (hit inside the reference's code) is not their subject
    for msg, cat in _IGNORED:
        warnings.filterwarnings("ignore", message=".*" + msg, category=cat)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_setup(item):
    _filters()
    item._hip_before = (_HIP["compiled"], _HIP["executed"])      # setup_method compiles count too
    mod = getattr(item, "module", None)
    if KIND != "reference" and mod is not None and mod.__name__.endswith("test_keepdims") \
            and getattr(mod, "Mode", None) is not _KeepdimsMode:
        # THE ONE EDIT of a test module: tests/tensor/test_keepdims.py builds every function with
        # ``Mode(optimizer="fast_compile", linker="py")`` — its own Python linker, whatever the
        # default mode.  The module-level name ``Mode`` is re-pointed at the HIP mode so that the
        # file's 200 keepdims / multi-axis reductions run through HipLinker (test bodies untouched).
        mod.Mode = _KeepdimsMode
    yield


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    _filters()
    yield
    c0, e0 = getattr(item, "_hip_before", (0, 0))
    # user_properties travel with the report (also from an xdist worker to the controller)
    item.user_properties.append(("through_hip", [_HIP["compiled"] - c0, _HIP["executed"] - e0]))


@pytest.fixture
def benchmark():
    """Stand-in for pytest-benchmark's fixture (not installed here): one call, its result — the
    reference's benchmark tests assert on the value they get back."""
    def run(fn, *args, **kwargs):
        return fn(*args, **kwargs)
    return run


_OUT = {}


def pytest_runtest_logreport(report):
    if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
        msg = ""
        if report.outcome == "failed":
            lr = report.longrepr
            crash = getattr(getattr(lr, "reprcrash", None), "message", None)
            msg = (crash or str(lr))[:400]
        elif report.outcome == "skipped":
            lr = report.longrepr
            msg = str(lr[2] if isinstance(lr, tuple) else lr)[:200]
        out = report.outcome
        if hasattr(report, "wasxfail"):
            out = "xfailed" if report.outcome == "skipped" else "xpassed"
        hip = dict(report.user_properties).get("through_hip", [0, 0])
        _OUT[report.nodeid] = [out, msg, hip]


def pytest_sessionfinish(session):
    path = os.environ.get("AESARA_HIP_SUITE_REPORT")
    if path and not hasattr(session.config, "workerinput"):      # the xdist controller / a plain run
        with open(path, "w") as f:
            json.dump(_OUT, f, indent=0, sort_keys=True)
