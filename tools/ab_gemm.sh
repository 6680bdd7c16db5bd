fmt='import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); print(r["tune"], r["config"][:30], "%.4f ms"%r["dev_ms"], "%.1f"%r["achieved"])'
run() { timeout 120 python tools/perf_probe.py --only nn32 --out /dev/null 2>&1 | python -c "$fmt"; }
for g in 8 4 16 32 2; do AESARA_HIP_GEMM_GROUP=$g run; done
