#!/usr/bin/env python
"""The vector hand-off of the persistent Scan kernels, in isolation.

What a step of `aesara_amd/scan_persist.py` pays per exchanged vector, with NOTHING else in the
step: G co-resident workgroups; every step each workgroup writes its K / G values of a K-vector as
tagged 8-byte granules ``{tag, value}`` (one relaxed agent-scope store each, write-through to L2),
then its polling wavefronts read ALL K granules with relaxed agent-scope loads until every one
carries this step's tag (four alternating slots, as the kernel uses), the first poll held back by
``delay`` x 64 cycles (`AESARA_HIP_SP_DELAY`), `s_sleep 1` between polls.  Time per step = the
floor of ONE all-to-all vector hand-off for that geometry; `bench.py`'s config-4 B = 1 row divides
its measured step time by (hand-offs per step x this number).

usage (GPU box): python tools/handoff_floor.py > gpurun_out/r05_handoff_floor.json
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SRC = r"""
#include <hip/hip_runtime.h>
typedef unsigned long long u64;
struct Args { u64* xch; unsigned* ctl; int T; };
#define AG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
extern "C" __global__ __launch_bounds__(%(BLOCK)d) void handoff(Args a) {
  constexpr int K = %(K)d, G = %(G)d, PT = %(PT)d, NG = (K + PT - 1) / PT, PER = K / G;
  const unsigned base = __hip_atomic_load(a.ctl, AG);
  float acc = 0.f;
  for (int t = 0; t < a.T; ++t) {
    u64* slot = a.xch + (t & 3) * K;
    const u64 tg = (u64)(base + (unsigned)t + 1u) << 32;
    if (threadIdx.x < PER)
      __hip_atomic_store(slot + blockIdx.x * PER + threadIdx.x,
                         tg | (u64)__float_as_uint(acc + (float)(blockIdx.x + t)), AG);
    if (threadIdx.x < PT) {
      u64 g[NG];
      %(DELAY)s
      const unsigned want = base + (unsigned)t + 1u;
      for (int spin = 0; spin < (1 << 22); ++spin) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < NG; ++q)
          if (threadIdx.x + q * PT < K) {
            g[q] = __hip_atomic_load(slot + threadIdx.x + q * PT, AG);
            ok = ok && ((unsigned)(g[q] >> 32) == want);
          }
        if (ok) break;
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int q = 0; q < NG; ++q) acc += __uint_as_float((unsigned)g[q]) * 1e-30f;
    }
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    __hip_atomic_store(a.ctl, base + (unsigned)a.T, AG);
    a.xch[4 * K] = (u64)__float_as_uint(acc);
  }
}
"""


def main():
    import torch
    from aesara_amd._lib import check, lib
    from aesara_amd.device import load_kernels

    class Args(C.Structure):
        _fields_ = [("xch", C.c_void_p), ("ctl", C.c_void_p), ("T", C.c_int)]
    K, T = 1024, 4096
    out = {"what": "one all-to-all hand-off of a %d-float vector between G co-resident workgroups "
                   "(tagged 8-byte granules, first poll held back by `delay` x 64 cycles); us per "
                   "step over %d steps, best of 5 launches" % (K, T), "rows": []}
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    check(lib.ahip_event_create(C.byref(ev0)))
    check(lib.ahip_event_create(C.byref(ev1)))
    for G in (64, 128, 256):
        for pollw in (2, 4, 8):
            for delay in (0, 8, 12, 15, 20):
                PT = 64 * pollw
                block = max(PT, K // G, 64)
                dl = "".join("__builtin_amdgcn_s_sleep(%d);" % min(15, delay - k)
                             for k in range(0, delay, 15))
                src = SRC % dict(BLOCK=block, K=K, G=G, PT=PT, DELAY=dl)
                (fn,) = load_kernels(src, ["handoff"])
                xch = torch.zeros(4 * K + 8, dtype=torch.int64, device="cuda")
                ctl = torch.zeros(4, dtype=torch.int32, device="cuda")
                a = Args(xch.data_ptr(), ctl.data_ptr(), T)
                best = None
                for _ in range(6):
                    check(lib.ahip_event_record(ev0, stream))
                    check(lib.ahip_launch(fn, G, 1, 1, block, 1, 1, 0, C.byref(a), C.sizeof(a), stream))
                    check(lib.ahip_event_record(ev1, stream))
                    ms = C.c_float()
                    check(lib.ahip_event_elapsed_ms(ev0, ev1, C.byref(ms)))
                    best = ms.value if best is None else min(best, ms.value)
                assert int(ctl[0].item()) == 6 * T
                out["rows"].append({"workgroups": G, "polling_waves": pollw, "delay": delay,
                                    "us_per_handoff": best * 1e3 / T})
    best = {}
    for r in out["rows"]:
        k = r["workgroups"]
        if k not in best or r["us_per_handoff"] < best[k]["us_per_handoff"]:
            best[k] = r
    out["best_per_grid"] = best
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
