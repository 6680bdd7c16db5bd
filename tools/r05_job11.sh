#!/bin/bash
# axis reductions, MALL-cold (rotating), streaming loads on: rows in flight per lane (RED_UNROLL: 1 = the default 8)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fmt='import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); print("%-14s %-60s %8.2f us  %5.0f GB/s (%.3f)" % (" ".join("%s=%s"%(k[11:],v) for k,v in r["tune"].items()), r["config"][:60], r["dev_ms"]*1e3, r["achieved"], r["frac"]))'
for e in "A=1" "AESARA_HIP_RED_UNROLL=2" "AESARA_HIP_RED_UNROLL=4" "AESARA_HIP_COL_LANES=64" "AESARA_HIP_COL_LANES=256"; do
  env $e PROBE_ROTATE=1 timeout 200 python tools/perf_probe.py --only axisred --out gpurun_out/r05_axisred_cold2.jsonl 2>&1 | python -c "$fmt"
done
