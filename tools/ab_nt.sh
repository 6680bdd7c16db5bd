fmt='import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); print(r["tune"], r["config"][:34], "%.2f us"%(r["dev_ms"]*1e3), "%.0f GB/s"%r["achieved"])'
run() { timeout 120 python tools/perf_probe.py --only cfg1b --out /dev/null 2>&1 | python -c "$fmt"; }
for rep in 1 2; do
AESARA_HIP_NT=0 run
AESARA_HIP_NT=1 run
AESARA_HIP_NT=2 run
AESARA_HIP_NT=3 run
done
