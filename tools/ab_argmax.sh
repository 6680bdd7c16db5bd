fmt='import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l)
        if "argmax" in r["config"] or "cumsum" in r["config"]: print(r["tune"], r["config"][:30], "%.2f us"%(r["dev_ms"]*1e3), "%.0f GB/s"%r["achieved"])'
run() { timeout 120 python tools/perf_probe.py --only misc --out /dev/null 2>&1 | python -c "$fmt"; }
AESARA_HIP_ARGMAX_SLICES=64 run
AESARA_HIP_ARGMAX_SLICES=128 run
AESARA_HIP_ARGMAX_SLICES=256 run
AESARA_HIP_ARGMAX_SLICES=32 run
