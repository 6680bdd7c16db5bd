#!/bin/bash
# round 5 (streaming policy on): wavefronts per CU x 16-byte vectors in flight of the config-2 kernel
# (the probe's best points: 1024 threads x 2-3, 512 x 3-4, 256 x 16 per CU)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]
        print("%-72s window %.2f us (%.4f)  sustained %.4f  exec %.4f  ceiling %.4f" % (os.environ.get("TAG",""), r["roofline"]["kernel_ms"]*1e3, r["roofline"]["frac"], c["sustained"]["frac"], c["executor_level"]["frac"], c["read_only_ceiling"]["frac"]))'
run() { TAG="$*" env "$@" timeout 300 python bench.py --no-cpu-baseline --no-warm --no-secondary --steps 20 --warmup 5 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
run A=default
run AESARA_HIP_UNROLL=3
run AESARA_HIP_RED_BLOCK=512 AESARA_HIP_RED_BPC=2 AESARA_HIP_UNROLL=3
run AESARA_HIP_RED_BLOCK=512 AESARA_HIP_RED_BPC=2 AESARA_HIP_UNROLL=4
run AESARA_HIP_RED_BLOCK=512 AESARA_HIP_RED_BPC=2 AESARA_HIP_UNROLL=6
run AESARA_HIP_RED_BLOCK=256 AESARA_HIP_RED_BPC=1 AESARA_HIP_UNROLL=8
run AESARA_HIP_RED_BLOCK=256 AESARA_HIP_RED_BPC=1 AESARA_HIP_UNROLL=16
run AESARA_HIP_RED_BLOCK=256 AESARA_HIP_RED_BPC=2 AESARA_HIP_UNROLL=4
run A=default
