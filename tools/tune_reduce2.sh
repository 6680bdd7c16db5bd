fmt='import sys, json
for l in sys.stdin:
    if l.startswith("{") and "UNFUSED" not in l:
        r=json.loads(l); print(r["tune"], r["config"][:10], "%.2f us"%(r["dev_ms"]*1e3))'
run() { timeout 120 python tools/perf_probe.py --only cfg2 --out /dev/null 2>&1 | python -c "$fmt"; }
run
AESARA_HIP_BLOCK=1024 AESARA_HIP_RED_BPC=1 run
AESARA_HIP_BLOCK=1024 AESARA_HIP_RED_BPC=2 run
AESARA_HIP_BLOCK=512 AESARA_HIP_RED_BPC=2 run
AESARA_HIP_BLOCK=512 AESARA_HIP_RED_BPC=4 run
AESARA_HIP_BLOCK=256 AESARA_HIP_RED_BPC=4 run
AESARA_HIP_BLOCK=256 AESARA_HIP_RED_BPC=16 run
run
