"""CPU baseline = the REFERENCE itself: the BASELINE configs compiled with the reference's own
``Mode("cvm", "fast_run")`` C linker (compile/mode.py:443-446) and timed on this host's cores.
TEST / BASELINE INFRASTRUCTURE — used by ``bench.py``'s ``cpu_baseline`` leg (never by the product).

The reference front end comes from ``ref_overlay`` (``/root/reference`` in the authoring container,
the packed overlay ``oracle/_ref/`` on the GPU box).  Thread settings are the reference's defaults:
Elemwise / CAReduce loops single-threaded (``openmp=False``, configdefaults.py:1037), BLAS = the
OpenBLAS bundled with NumPy / SciPy at its default thread count (``blas__ldflags`` is empty here, so
Gemm/Dot22 go through the NumPy C-API alt-BLAS, blas_headers.py:744-777, and Gemv through SciPy
fblas, blas.py:279-310).  ``--openmp`` runs the ``AESARA_FLAGS=openmp=True`` form instead.

usage:
  python oracle/time_reference.py --warm [--configs a,b]     compile everything into the cache, exit
  python oracle/time_reference.py --configs cfg2,cfg3b --budget 6
        -> one JSON line {"host": .., "rows": {cfg: {"ms_per_eval": .., "evals": .., "cores": ..}}}
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ALL = ["cfg2", "cfg1b", "cfg3a", "cfg3b", "cfg4_b1", "cfg4_b64", "cfg5"]
# BatchedDot / Dot22 away from the square case (bench.py sec_gemmshapes): (batch, M, N, K) / (M, N, K)
GEMM_SHAPES = {"bdot_64x512": (64, 512, 512, 512), "bdot_1024x64": (1024, 64, 64, 64),
               "dot22_tall": (16384, 64, 1024), "dot22_wide": (64, 16384, 1024)}


def _normal_f32_parallel(np, seed, rows, cols):
    """(rows, cols) float32 standard normals, filled in 2^20-row chunks by a thread pool (NumPy's
    generators release the GIL): chunk k comes from ``default_rng([seed, k])``, so the array is the
    same whatever the number of threads — 16 GiB in seconds on the GPU box's host cores instead
    of ~30 s single-threaded."""
    from concurrent.futures import ThreadPoolExecutor
    out = np.empty((rows, cols), dtype="float32")
    blk = 1 << 20

    def fill(k):
        lo = k * blk
        np.random.default_rng([seed, k]).standard_normal(out=out[lo:lo + blk], dtype="float32")
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as pool:
        list(pool.map(fill, range((rows + blk - 1) // blk)))
    return out


def make_inputs(name, np, tiny=False, full=False):
    """The seeded input arrays of one config — a dict; no front end needed, so ``bench.py`` and the
    GPU tests regenerate exactly what the reference child evaluated and compare results.
    ``tiny``: small shapes (same graph and dtypes), used by --warm.  ``full``: BASELINE's full
    shapes for the two configs whose TIMED rows are samples — config 4 with all T = 512 steps,
    config 5 with all N = 2^24 rows (16 GiB of X) — evaluated ONCE by the reference for the
    full-shape result check (SURVEY §8d "run it once")."""
    n = 64 if tiny else 4096
    if name == "cfg2":
        return {"x": np.random.default_rng(1).standard_normal((n, n)), "mu": np.asarray(0.1),
                "sigma": np.asarray(1.3)}
    if name == "cfg1b":
        return {"x": np.random.default_rng(0).random((n, n)), "y": np.random.default_rng(1).random((n, n))}
    if name == "cfg3a":
        return {"M": np.random.default_rng(2).standard_normal((n, n)),
                "v": np.random.default_rng(3).standard_normal(n), "a": np.asarray(2.0)}
    if name == "cfg3b":
        return {"A": np.random.default_rng(3).standard_normal((n, n)).astype("float32"),
                "B": np.random.default_rng(4).standard_normal((n, n)).astype("float32"),
                "C": np.zeros((n, n), "float32")}
    if name in ("cfg4_b1", "cfg4_b64"):
        B_ = 1 if name == "cfg4_b1" else 64
        T_, H = (8, 32) if tiny else ((512, 1024) if full else (64, 1024))   # sample: 1/8 of the 512 steps
        if tiny and B_ > 1:
            B_ = 4
        xs = (T_, H) if B_ == 1 else (T_, B_, H)
        d = {"x": (np.random.default_rng(4).standard_normal(xs) * 0.1).astype("float32"),
             "h0": np.zeros(xs[1:], "float32")}
        for k, nm in enumerate(("Wz", "Uz", "Wr", "Ur", "Wh", "Uh")):
            d[nm] = (np.random.default_rng(5 + k).standard_normal((H, H)) / np.sqrt(H)).astype("float32")
        return d
    if name in GEMM_SHAPES:
        shp = GEMM_SHAPES[name]
        if tiny:
            shp = tuple(min(8, v) for v in shp)
        if len(shp) == 4:
            Bn, M, N_, K = shp
            return {"x": np.random.default_rng(11).standard_normal((Bn, M, K), dtype="float32"),
                    "y": np.random.default_rng(12).standard_normal((Bn, K, N_), dtype="float32")}
        M, N_, K = shp
        return {"x": np.random.default_rng(13).standard_normal((M, K), dtype="float32"),
                "y": np.random.default_rng(14).standard_normal((K, N_), dtype="float32")}
    if name == "cfg5":
        N, D = (256, 16) if tiny else (1 << 20, 256)
        if full and not tiny:
            N = 1 << 24
            return {"X": _normal_f32_parallel(np, 6, N, D),
                    "w": (np.random.default_rng(7).standard_normal(D) / 16).astype("float32"),
                    "b": np.asarray(0.1, "float32"),
                    "y": (np.random.default_rng(8).random(N) < 0.5).astype("float32")}
        return {"X": np.random.default_rng(6).standard_normal((N, D), dtype="float32"),
                "w": (np.random.default_rng(7).standard_normal(D) / 16).astype("float32"),
                "b": np.asarray(0.1, "float32"),
                "y": (np.random.default_rng(8).random(N) < 0.5).astype("float32")}
    raise ValueError(name)


def build(ae, name, np, tiny=False, full=False):
    """-> (function, args, sample description, cores used, result getter).  ``tiny``: small shapes
    (same graph and dtypes — the compiled modules are shape-independent), used by --warm."""
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    mode = Mode("cvm", "fast_run")
    nthreads = os.cpu_count()
    n = 64 if tiny else 4096
    d = make_inputs(name, np, tiny, full)
    if name == "cfg2":
        x, mu, sg = at.dmatrix("x"), at.dscalar("mu"), at.dscalar("sigma")
        f = ae.function([x, mu, sg], at.exp(-((x - mu) ** 2) / (2 * sg ** 2)).sum(), mode=mode)
        return f, (d["x"], d["mu"], d["sigma"]), "fp64 %dx%d exp-sum graph" % (n, n), \
            (nthreads if ae.config.openmp else 1)
    if name == "cfg1b":
        x, y = at.dmatrix("x"), at.dmatrix("y")
        f = ae.function([x, y], x + y, mode=mode)
        return f, (d["x"], d["y"]), "fp64 %dx%d add" % (n, n), (nthreads if ae.config.openmp else 1)
    if name == "cfg3a":
        M, v, a = at.dmatrix("M"), at.dvector("v"), at.dscalar("a")
        f = ae.function([M, v, a], at.dot(M, v) + a, mode=mode)
        return f, (d["M"], d["v"], d["a"]), \
            "fp64 %dx%d M.dot(v)+a (Gemv through SciPy fblas)" % (n, n), nthreads
    if name == "cfg3b":
        A = ae.shared(d["A"], "A")
        B = ae.shared(d["B"], "B")
        Cs = ae.shared(d["C"], "C")
        f = ae.function([], [], updates=[(Cs, np.float32(0.4) * Cs + np.float32(0.8) * at.dot(A, B))],
                        mode=mode)
        f.result_of_first_eval = lambda: [Cs.get_value()]
        return f, (), "fp32 %d^3 Gemm update (alt-BLAS -> NumPy's OpenBLAS)" % n, nthreads
    if name in ("cfg4_b1", "cfg4_b64"):
        B_ = 1 if name == "cfg4_b1" else 64
        T_, H = d["x"].shape[0], d["x"].shape[-1]
        if tiny and B_ > 1:
            B_ = 4
        x = at.fmatrix("x") if B_ == 1 else at.ftensor3("x")
        h0 = at.fvector("h0") if B_ == 1 else at.fmatrix("h0")
        Ws = [ae.shared(d[nm], nm) for nm in ("Wz", "Uz", "Wr", "Ur", "Wh", "Uh")]

        def step(x_t, h, Wz, Uz, Wr, Ur, Wh, Uh):
            z = at.sigmoid(at.dot(x_t, Wz) + at.dot(h, Uz))
            r = at.sigmoid(at.dot(x_t, Wr) + at.dot(h, Ur))
            hh = at.tanh(at.dot(x_t, Wh) + at.dot(r * h, Uh))
            return (1 - z) * h + z * hh
        hs, _ = ae.scan(step, sequences=[x], outputs_info=[h0], non_sequences=Ws)
        f = ae.function([x, h0], hs[-1], mode=mode)
        return f, (d["x"], d["h0"]), \
            "fp32 Scan GRU T=%d of 512 steps%s, H=%d B=%d (scan_perform.pyx loop, inner cvm " \
            "function)" % (T_, "" if T_ == 512 else " (x8 for the config)", H, B_), nthreads
    if name in GEMM_SHAPES:
        if d["x"].ndim == 3:
            x, y = at.ftensor3("x"), at.ftensor3("y")
            # BatchedDot's C thunk (blas.py:2243 batch_gemm) links sgemm_ directly: with
            # blas__ldflags == '' (no system BLAS in this image) the module does not load, so the
            # reference itself runs the Op's Python perform (:2224: np.dot per item -> OpenBLAS)
            f = ae.function([x, y], at.batched_dot(x, y), mode=Mode("py", "fast_run"))
            what = "fp32 BatchedDot %s (BatchedDot.perform: np.dot per item; its C thunk needs a system " \
                   "BLAS that this image lacks)" % (d["x"].shape,)
        else:
            x, y = at.fmatrix("x"), at.fmatrix("y")
            f = ae.function([x, y], at.dot(x, y), mode=mode)              # -> Dot22, tensor/blas.py:1659
            what = "fp32 Dot22 %s @ %s" % (d["x"].shape, d["y"].shape)
        return f, (d["x"], d["y"]), what, nthreads
    if name == "cfg5":
        N, D = d["X"].shape
        X, w, b, y = at.fmatrix("X"), at.fvector("w"), at.fscalar("b"), at.fvector("y")
        p = at.sigmoid(at.dot(X, w) + b)
        logp = (y * at.log(p) + (1 - y) * at.log(1 - p)).sum()
        gw, gb = ae.grad(logp, [w, b])
        f = ae.function([X, w, b, y], [logp, gw, gb], mode=mode)
        return f, (d["X"], d["w"], d["b"], d["y"]), \
            "fp32 logistic logp+grad N=2^%d D=%d%s" % (
                N.bit_length() - 1, D, "" if N == 1 << 24 else
                " (1/16 of the config's rows: x16 for the full batch)"), nthreads
    raise ValueError(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default=",".join(ALL))
    ap.add_argument("--budget", type=float, default=5.0, help="seconds of timed evals per config")
    ap.add_argument("--warm", action="store_true")
    ap.add_argument("--openmp", action="store_true")
    ap.add_argument("--full", action="store_true",
                    help="BASELINE's full shapes for cfg4 (T = 512) and cfg5 (N = 2^24); with "
                         "--budget 0 every config is evaluated exactly once (the result check)")
    ap.add_argument("--dump-dir", default="",
                    help="save what the FIRST evaluation of every config returned (npy files "
                         "<cfg>_out<i>.npy): the values bench.py / the GPU tests check the HIP path against")
    args = ap.parse_args()
    if args.openmp:
        os.environ["AESARA_FLAGS"] = "openmp=True"
        os.environ.setdefault("OMP_NUM_THREADS", str(os.cpu_count()))
        os.environ["AESARA_REF_COMPILEDIR_SUFFIX"] = "_omp"
    sys.path.insert(0, HERE)
    import numpy as np
    import ref_overlay
    t_imp = time.perf_counter()
    ae = ref_overlay.import_reference()
    t_imp = time.perf_counter() - t_imp
    rows = {}
    for name in [c for c in args.configs.split(",") if c]:
        t0 = time.perf_counter()
        try:
            f, fargs, sample, cores = build(ae, name, np, tiny=args.warm, full=args.full)
            f.trust_input = True
            row = {"cores": cores, "sample": sample}
            t = time.perf_counter()
            res = f(*fargs)                             # first eval (page faults, lazy imports)
            first = time.perf_counter() - t
            if args.dump_dir and not args.warm:
                getter = getattr(f, "result_of_first_eval", None)
                vals = getter() if getter else (res if isinstance(res, (list, tuple)) else [res])
                os.makedirs(args.dump_dir, exist_ok=True)
                for i, v in enumerate(vals):
                    np.save(os.path.join(args.dump_dir, "%s_out%d.npy" % (name, i)), np.asarray(v))
                row["dumped_outputs"] = len(vals)
            row["build_s"] = time.perf_counter() - t0 - first
            if not args.warm:
                ts = []
                # cfg3a streams ONE 128 MiB matrix per eval: the GPU leg rotates over 8 matrices so
                # that none is cache-resident (MALL) — the host leg rotates over 16 copies (2 GiB,
                # beyond the L3 of any EPYC) for the same reason (it used to
                # re-read one L3-resident matrix)
                alts = [fargs]
                if name == "cfg3a":
                    alts = [(np.array(fargs[0], copy=True),) + tuple(fargs[1:]) for _ in range(16)]
                    row["sample"] = sample + ", rotating over 16 copies of M (cache-cold like the GPU leg)"
                t_end = time.perf_counter() + args.budget
                if first > args.budget / 2:             # slow config: the first eval IS the sample
                    ts = [first]
                while not ts or (time.perf_counter() < t_end and len(ts) < 50):
                    a_ = alts[len(ts) % len(alts)]
                    t = time.perf_counter()
                    f(*a_)
                    ts.append(time.perf_counter() - t)
                ts.sort()
                row.update(ms_per_eval=ts[len(ts) // 2] * 1e3, evals=len(ts),
                           nodes=[str(n.op) for n in f.maker.fgraph.toposort()][:24])
        except Exception as e:                          # noqa: BLE001  (a baseline row must not kill the rest)
            row = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        rows[name] = row
    print("RESULT " + json.dumps({
        "host": {"nproc": os.cpu_count()}, "import_s": t_imp, "source": ref_overlay.source(),
        "mode": "Mode('cvm','fast_run'), trust_input, median", "openmp": bool(ae.config.openmp),
        "blas": "NumPy C-API alt-BLAS / SciPy fblas -> bundled OpenBLAS, default threads", "rows": rows}))


if __name__ == "__main__":
    main()
