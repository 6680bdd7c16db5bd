# sweep of the streaming-Elemwise launch knobs on BASELINE config 1b (z = x + y, fp64 4096^2)
mkdir -p gpurun_out; rm -f gpurun_out/tune_s.jsonl
fmt='import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); print(r["tune"], r["config"][:14], "%.2f us"%(r["dev_ms"]*1e3), "%.0f GB/s"%r["achieved"])'
run() { timeout 120 python tools/perf_probe.py --only cfg1b --out gpurun_out/tune_s.jsonl 2>&1 | python -c "$fmt"; }
AESARA_HIP_STREAM_BPC=8 run
AESARA_HIP_STREAM_BPC=4 run
AESARA_HIP_STREAM_BPC=16 run
AESARA_HIP_STREAM_BPC=32 run
AESARA_HIP_STREAM_BPC=8 AESARA_HIP_UNROLL=2 run
AESARA_HIP_STREAM_BPC=4 AESARA_HIP_UNROLL=4 run
AESARA_HIP_STREAM_BPC=8 AESARA_HIP_NT=1 run
AESARA_HIP_STREAM_BPC=16 AESARA_HIP_NT=1 AESARA_HIP_UNROLL=2 run
AESARA_HIP_STREAM_BPC=8 AESARA_HIP_VECBYTES=32 run
AESARA_HIP_STREAM_BPC=8 AESARA_HIP_BLOCK=512 run
AESARA_HIP_STREAM_BPC=16 AESARA_HIP_BLOCK=128 run
