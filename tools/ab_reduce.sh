fmt='import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l)
        if "UNFUSED" in r["config"]: continue
        print(r["tune"], r["config"][:44], "%.2f us"%(r["dev_ms"]*1e3))'
run() { timeout 120 python tools/perf_probe.py --only cfg2,redsum --out /dev/null 2>&1 | python -c "$fmt"; }
for rep in 1 2 3; do
AESARA_HIP_BLOCK=256 AESARA_HIP_RED_BPC=8 run
AESARA_HIP_BLOCK=1024 AESARA_HIP_RED_BPC=2 run
done
