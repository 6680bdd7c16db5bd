"""CPU baseline of the REFERENCE itself (authoring container only): BASELINE
configs 1b / 2 / 3b compiled with the reference's own ``Mode("cvm", "fast_run")`` C linker,
``trust_input``, 2 warm-ups + median of 5 evals; run once with the default flags (Elemwise loops
single-threaded: ``openmp=False``, configdefaults.py:1037) and once with
``AESARA_FLAGS=openmp=True OMP_NUM_THREADS=<nproc>``.  TEST / BASELINE INFRASTRUCTURE.

usage: python oracle/time_reference.py            -> profiles/r02_reference_cpu_timings.json
"""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def child():
    sys.path.insert(0, HERE)
    import numpy as np
    import ref_overlay
    ae = ref_overlay.import_reference()
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    mode = Mode("cvm", "fast_run")
    res = {}

    def timed(f, args, n=5):
        f.trust_input = True
        for _ in range(2):
            f(*args)
        ts = []
        for _ in range(n):
            t = time.perf_counter()
            f(*args)
            ts.append(time.perf_counter() - t)
        return sorted(ts)[len(ts) // 2] * 1e3

    x, y = at.dmatrix("x"), at.dmatrix("y")
    xv = np.random.default_rng(0).random((4096, 4096))
    yv = np.random.default_rng(1).random((4096, 4096))
    res["cfg1b add f64 4096^2 ms"] = timed(ae.function([x, y], x + y, mode=mode), (xv, yv))
    mu, sg = at.dscalar("mu"), at.dscalar("sigma")
    f2 = ae.function([x, mu, sg], at.exp(-((x - mu) ** 2) / (2 * sg ** 2)).sum(), mode=mode)
    xn = np.random.default_rng(1).standard_normal((4096, 4096))
    res["cfg2 exp-sum f64 4096^2 ms"] = timed(f2, (xn, np.asarray(0.1), np.asarray(1.3)))
    A = ae.shared(np.random.default_rng(3).standard_normal((4096, 4096)).astype("float32"), "A")
    B = ae.shared(np.random.default_rng(4).standard_normal((4096, 4096)).astype("float32"), "B")
    Cs = ae.shared(np.zeros((4096, 4096), "float32"), "C")
    f3 = ae.function([], [], updates=[(Cs, 0.4 * Cs + 0.8 * at.dot(A, B))], mode=mode)
    res["cfg3b gemm f32 4096^3 ms"] = timed(f3, (), n=3)
    res["openmp"] = bool(ae.config.openmp)
    res["nodes cfg2"] = [str(n.op) for n in f2.maker.fgraph.toposort()]
    print("RESULT " + json.dumps(res))


def main():
    if os.environ.get("_TIME_REF_CHILD"):
        return child()
    nproc = os.cpu_count()
    out = {"host": {"nproc": nproc}, "mode": "Mode('cvm','fast_run'), trust_input, median of 5",
           "blas": "NumPy C-API alt-BLAS -> bundled OpenBLAS (blas__ldflags empty), threads = OpenBLAS default"}
    for label, flags in (("default (openmp=False)", ""), ("openmp=True", "openmp=True")):
        env = dict(os.environ, _TIME_REF_CHILD="1", OMP_NUM_THREADS=str(nproc))
        env["AESARA_FLAGS"] = flags
        if flags:
            env["AESARA_REF_COMPILEDIR_SUFFIX"] = "_omp"
        p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True,
                           text=True)
        line = next((l for l in p.stdout.splitlines() if l.startswith("RESULT ")), None)
        out[label] = json.loads(line[7:]) if line else {"error": (p.stderr or p.stdout)[-800:]}
    path = os.path.join(ROOT, "profiles", "r02_reference_cpu_timings.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
