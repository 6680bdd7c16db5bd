#!/bin/bash
# round 5: the stream probe (tools/probes/stream_probe.hip) says FEWER bytes in flight per lane stream
# faster on a MALL-cold 128 MiB read (16 B x 2: 23.4 us, x 4: 24.6, x 8: 25.8 per launch): bytes per
# lane x vectors in flight x non-temporal on the real config-2 kernel (driver flags, 4 extra windows)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("regions"): print("   ", l.strip()[:230])
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]
        print("%-58s window %.2f us (%.4f)  sustained %.4f  exec %.4f  ceiling %.4f" % (os.environ.get("TAG",""), r["roofline"]["kernel_ms"]*1e3, r["roofline"]["frac"], c["sustained"]["frac"], c["executor_level"]["frac"], c["read_only_ceiling"]["frac"]))'
run() { TAG="$*" env "$@" AESARA_BENCH_REGIONS=4 timeout 240 python bench.py --no-cpu-baseline --no-warm --no-secondary --steps 20 --warmup 5 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
run A=default
run AESARA_HIP_VECBYTES=16
run AESARA_HIP_VECBYTES=16 AESARA_HIP_UNROLL=1
run AESARA_HIP_VECBYTES=16 AESARA_HIP_UNROLL=4
run AESARA_HIP_VECBYTES=16 AESARA_HIP_NT=1
run AESARA_HIP_UNROLL=1
run AESARA_HIP_NT=1
run A=default
