"""One registry for every ``AESARA_HIP_*`` environment switch of the package.

The defaults are the measured ones (DESIGN §3.5 lists them with the sweep each came from); the
environment overrides exist for the sweeps under ``tools/`` and for A/B runs on the GPU box.  A
switch is read when it is asked for (``get``), never cached here, so a test may flip one with
``monkeypatch.setenv``.  Nothing outside this module reads ``os.environ["AESARA_HIP_..."]``.
"""
from __future__ import annotations

import os

_REG = {}     # name -> (default, type, doc)


def _def(name, default, typ, doc):
    _REG[name] = (default, typ, doc)


# ---- library / caches --------------------------------------------------------------------------
_def("LIB", None, str, "path of libaesara_hip.so (default: next to the package)")
_def("KCACHE", None, str, "directory of the generated-kernel cache (default: aesara_amd/_kcache)")
_def("RCCL", None, str, "path of the RCCL library the C-ABI communicator dlopens")
_def("ARENA_CAP_GB", 64.0, float, "replay arenas of all signatures together, GiB, before eviction")
# ---- generated Elemwise / CAReduce kernels -----------------------------------------------------
_def("UNROLL", None, int, "vectors in flight per lane of a flat stream (default: 2 up to 256 MiB, else 1)")
_def("NT", 0, int, "bit 0: non-temporal loads, bit 1: non-temporal stores (flat streams); unset: operands of 96 MiB or more "
     "stream (16-byte non-temporal accesses, exec_elemwise.BIG_STREAM), setting it or VECBYTES switches that policy off")
_def("VECBYTES", 32, int, "bytes per lane per iteration of a flat stream (unset: 16 for operands of 96 MiB or more, see NT)")
_def("BLOCK", 256, int, "threads per workgroup of Elemwise kernels")
_def("RED_BLOCK", None, int, "threads per workgroup of full reductions (default 1024)")
_def("TILED", 1, int, "LDS-tiled form for transposed operands")
_def("FASTEXP", 1, int, "float64 exp through the 64-entry table in LDS (0: ocml's exp)")
_def("FASTDIV", 2, int, "x / c for a loop-invariant RUN-TIME c (a broadcast scalar divisor; constant divisors never "
     "get here: the reference's canonicaliser has already turned them into a multiplication).  0 = always the hoisted "
     "reciprocal + Markstein step: the correctly rounded IEEE quotient, what the reference computes; 1 = always "
     "x * (1/c), <= 1.5 ulp from the quotient (exact quotients such as 21/7 are then inexact: floor / eq / int casts "
     "downstream can differ — opt-in only); 2 (default) = x * (1/c) ONLY for a quotient that reaches memory solely as a "
     "term of the kernel's own floating-point sum through continuous functions (codegen.sum_only_nodes: a result that "
     "already depends on the order of summation), IEEE everywhere else — any quotient that is stored, compared, rounded "
     "or cast is exact.  Config 2 (r06, one box): 26.9 us per eval with 0, 24.8 with 1 / 2")
_def("RC_NT", None, int, "row chains: non-temporal loads of the full operands (default: operands of 96 MiB or more; "
     "0 / 1 forces it)")
_def("EARLY", 1, int, "flat full reductions issue their first loads before the invariant prologue")
_def("RED_BLOCKED", 2, int, "flat full reductions: 0 grid-stride walk, 1 one contiguous chunk per workgroup, "
     "2 (default) chunks ordered so that every XCD streams one contiguous eighth (r04 sweeps: 27.1 vs 27.7 us)")
_def("EW_TRACE", 0, int, "full reductions stamp s_memrealtime per workgroup into the workspace (tools/ew_trace.py)")
_def("HFUSE", 1, int, "horizontal fusion of independent same-shape Elemwise/CAReduce steps into one launch")
# ---- launch shapes owned by the C side (ahip_set_param) ----------------------------------------
_def("RED_ROW_BPC", None, int, "workgroups per CU of row-mode axis reductions that walk many outputs (default 16)")
_def("RED_BPC", None, int, "256-thread blocks per CU of full reductions")
_def("STREAM_BPC", None, int, "256-thread blocks per CU of streaming Elemwise kernels")
_def("GEMV_COL_BPC", None, int, "blocks per CU of the COL gemv")
_def("GEMV_COL_LANES", None, int, "strip width (lanes) of the COL gemv")
_def("ARGMAX_SLICES", None, int, "maximum slices of a column argmax")
_def("GEMM_GROUP", None, int, "tile-group width of the GEMM's XCD-aware block order")
_def("GEMM_HALF_MAX", None, int, "largest 128x128-tile count that still takes the 64x64 tile")
_def("GEMM_HALF_MIN", None, int, "smallest 128x128-tile count that takes the 64x64 tile")
_def("COPY_STREAM_BYTES", None, int, "strided copies of at least this many bytes read their source with non-temporal "
     "loads (default 96 MiB, the streaming policy; a huge value switches it off)")
_def("GEMM_HALF_KSPLIT", None, int, "K groups inside a 64x64-tile workgroup (default: 4 up to one workgroup per CU, "
     "2 up to two, else 1; 1 / 2 / 4 forces it)")
_def("GE_WAVES", 0, int, "waves per workgroup of the generated GEMM epilogue (0: automatic)")
# ---- Scan ---------------------------------------------------------------------------------------
_def("SCAN_PERSIST", 1, int, "Scan loops as ONE persistent kernel where the class allows")
_def("ROWS_BYTES_CAP_GB", 0.0, float, "bytes a whole-sequence evaluation of a Scan without recurrence may allocate before "
     "its rows go in blocks (GiB; 0: a quarter of the free device memory)")
_def("COOP", 0, int, "persistent kernels through hipLaunchCooperativeKernel (launch-time size check)")
_def("SCAN_ROWS", None, str, "rows x waves geometry of the vector-state persistent kernel")
_def("SCAN_WAVES", 4, int, "waves per workgroup of the vector-state persistent kernel")
_def("SP_POLLW", 4, int, "polling waves of the vector-state kernel (r04, with the first poll held back: 2 -> 4 "
     "waves 4.05 -> 3.56 us per step at config 4 B = 1)")
_def("SP_SLEEP", 1, int, "s_sleep between polls of the vector-state kernel")
_def("SP_DELAY", 15, int, "vector-state kernel: s_sleep units (64 cycles) before the first poll of a hand-off "
     "(config 4 B = 1, 128 x 8 rows, 2 polling waves: 4.59 us per step with 0, 4.05 with 12, 4.27 with 20; "
     "256 x 4 rows, 4 polling waves: 3.52 / 3.42 / 3.38 / 3.37 / 3.43 with 10 / 12 / 14 / 16 / 18)")
_def("SP_EARLYDOTS", 1, int, "vector-state kernel: products on operands staged by an earlier phase of the step are "
     "issued in front of the phase's hand-off wait (0: behind it, the round-5 order)")
_def("SP_TRACE", 0, int, "vector-state kernel: stamp s_memtime at the phase marks (tools/sp_trace.py)")
_def("SM_CHUNK", 32, int, "k-chunk of the matrix-state kernel's exchange")
_def("SM_XMODE", None, str, "exchange form of the matrix-state kernel: frag | flag (default by registers)")
_def("SM_EARLY", "first", str, "which product of a step starts before the hand-off: first | none")
_def("SM_PIN", None, int, "fragment form: pin the recurrent weights to architectural (1) or accumulation (2) "
     "registers, 0 = leave them to the allocator (default: 2 with in-loop sequence products, else 1)")
_def("SM_TRACE", 0, int, "stamp s_memtime at the phase marks (tools/sm_trace.py)")
_def("SM_FENCE", 1, int, "scheduler fences at the phase marks")
_def("SM_INIT", "publish", str, "fragment form, initial state: branch (step 0 reads the output buffer) | publish "
     "(every workgroup publishes its tile before the loop: no step-0 code in the loop)")
_def("SM_XRELOAD", "late", str, "in-loop sequence products: the next x is requested behind the MFMAs of the first "
     "product on the fetched operand (late) | of the next phase's product on staged operands (early)")
_def("SM_ACKFILL", 1, int, "the next phase's products on staged operands are issued before the wait for the "
     "acknowledgement of this phase's payload stores")
_def("SM_EPRE", 6, int, "with SM_ACKFILL: fragments (of 16) of the staged-operand product issued before the "
     "acknowledgement wait, the rest behind the tag (0 = all of them before)")
_def("SM_ELOOK", 6, int, "fragments of the rest of that product issued before the first look at the next tags "
     "(0 = the look belongs to the window, SM_LOOK)")
_def("SM_XSPLIT", 4, int, "fragments of a wrapped sequence product that stay in the last window of the step; the "
     "others move to the first window of the next step (0 = all stay)")
_def("SM_XPRE", 6, int, "fragments of the next step's first sequence-product window issued before the wait for "
     "the acknowledgement of the last phase's payload stores")
_def("SM_NXT", "fetch", str, "per-step sequence operands of the epilogue: requested at the top of the step (top) | "
     "behind the first fetch of the step (fetch: an HBM-cold load in front of the fetch delays every "
     "in-order wait behind it)")
_def("SM_POLLS", 1, int, "fragment form: tag polls a wavefront keeps in flight while it waits for a hand-off "
     "(1 = one at a time.  More find the tags sooner after they arrive but 64 workgroups poll each tag line: "
     "measured r04, GRU training step 17.7 ms with 1, 18.6 with 4; forward kernel unchanged)")
_def("SM_SPIN_DELAY", 15, int, "fragment form, fetches with nothing in front of them: s_sleep units (64 cycles) "
     "before the first poll of the tags (a poll that comes back empty costs a whole further round trip AND "
     "slows the tag stores it polls for: GRU training step B = 64 17.7 -> 15.6 ms with 15)")
_def("SM_LOOK", 8, int, "fragments of a window's head issued before the first look at the tags")
_def("SM_ASM_MARKS", 0, int, "label the phase marks in the ISA (asm comments; for reading disassembly)")
_def("SM_XTAIL", 8, int, "fragments of the sequence product behind the payload loads")
_def("SM_BATCH_CHUNKS", 1, int, "a batch whose 16 x 16 tiles outnumber the CUs runs as several launches of the "
     "one-block-per-workgroup kernel over slices of the batch (independent recurrences) instead of one launch with "
     "2 / 4 / 8 blocks per workgroup: 1 = forward Scans (their in-loop sequence products need one block per "
     "workgroup), 2 = every Scan, 0 = never")
_def("SM_XFOLD", 1, int, "sequence products x_t @ W inside the loop")
_def("SM_XREG", 0, int, "1 = in-loop sequence products beyond the LDS capacity keep their weight columns in "
     "accumulation registers (as many as fit next to the recurrent weights); 0 = those are computed up front "
     "(config 4, B = 64: the third product in the loop costs 0.6 ms of exposed MFMA time, the GEMM it replaces 0.69 — "
     "4.30 vs 4.12 ms)")
_def("SM_INTERLEAVE", 1, int, "interleave the phases of the batch blocks of one workgroup")
_def("SPIN_LOG2", 21, int, "log2 of the poll limit of the persistent Scan kernels before they raise their error word "
     "(tools/profile_scan_r04.sh raises it: under the counter passes a kernel runs many times slower)")
_def("SM_DEPTH", 8, int, "16-byte payload loads in flight per lane of the fragment fetch")


def names():
    return sorted(_REG)


def doc(name):
    return _REG[name][2]


def raw(name):
    """The environment string of a switch, or None when it is not set."""
    if name not in _REG:
        raise KeyError("unknown switch AESARA_HIP_%s (add it to aesara_amd/knobs.py)" % name)
    return os.environ.get("AESARA_HIP_" + name)


def is_set(name):
    return bool(raw(name))


def get(name, default=None):
    """Value of a switch: the environment override parsed with the registered type, else the
    registered default (or ``default`` when the registry leaves it to the caller)."""
    dflt, typ, _ = _REG[name] if name in _REG else (None, None, None)
    v = raw(name)
    if v is None or v == "":
        return dflt if default is None else default
    return typ(v)


def table():
    """Rows (name, default, current, doc) for DESIGN §3.5 / debugging."""
    return [("AESARA_HIP_" + n, _REG[n][0], raw(n), _REG[n][2]) for n in names()]
