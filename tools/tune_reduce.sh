mkdir -p gpurun_out; rm -f gpurun_out/tune.jsonl
fmt='import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); print(r["tune"], r["config"][:14], "%.2f us"%(r["dev_ms"]*1e3), "%.0f GB/s"%r["achieved"])'
run() { timeout 120 python tools/perf_probe.py --only cfg2 --out gpurun_out/tune.jsonl 2>&1 | grep -v UNFUSED | python -c "$fmt"; }
AESARA_HIP_VECBYTES=16 AESARA_HIP_RED_BPC=8 run
AESARA_HIP_VECBYTES=16 AESARA_HIP_RED_BPC=16 run
AESARA_HIP_VECBYTES=16 AESARA_HIP_UNROLL=2 AESARA_HIP_RED_BPC=8 run
AESARA_HIP_VECBYTES=64 AESARA_HIP_RED_BPC=4 run
AESARA_HIP_VECBYTES=64 AESARA_HIP_RED_BPC=2 run
AESARA_HIP_BLOCK=512 AESARA_HIP_RED_BPC=2 run
AESARA_HIP_BLOCK=512 AESARA_HIP_RED_BPC=4 run
AESARA_HIP_BLOCK=1024 AESARA_HIP_RED_BPC=1 run
AESARA_HIP_BLOCK=1024 AESARA_HIP_RED_BPC=2 run
AESARA_HIP_BLOCK=128 AESARA_HIP_RED_BPC=8 run
AESARA_HIP_BLOCK=128 AESARA_HIP_RED_BPC=16 run
