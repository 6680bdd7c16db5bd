fmt='import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); print(r["tune"], "%.2f us"%(r["dev_ms"]*1e3))'
run() { timeout 120 python tools/perf_probe.py --only cfg2 --out /dev/null 2>&1 | grep -v UNFUSED | python -c "$fmt"; }
for rep in 1 2 3; do
AESARA_HIP_BLOCK=256 AESARA_HIP_RED_BPC=8 run
AESARA_HIP_BLOCK=1024 AESARA_HIP_RED_BPC=2 run
AESARA_HIP_BLOCK=512 AESARA_HIP_RED_BPC=2 run
AESARA_HIP_BLOCK=1024 AESARA_HIP_RED_BPC=3 run
done
