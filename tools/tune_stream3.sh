#!/bin/bash
# round 5 (streaming policy on): workgroups per CU x vectors in flight of the plain add stream (config 1b)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l)
        for s in r.get("secondary", []):
            print("%-56s add %.2f us (%.3f)" % (os.environ.get("TAG",""), s["roofline"]["kernel_ms"]*1e3, s["roofline"].get("frac") or 0))'
run() { TAG="$*" env "$@" timeout 300 python bench.py --no-cpu-baseline --no-warm --executor-level --only-secondary cfg1b --steps 20 --warmup 5 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
run A=default
run AESARA_HIP_STREAM_BPC=1
run AESARA_HIP_STREAM_BPC=2
run AESARA_HIP_STREAM_BPC=3
run AESARA_HIP_STREAM_BPC=4
run AESARA_HIP_STREAM_BPC=1 AESARA_HIP_UNROLL=2
run AESARA_HIP_STREAM_BPC=2 AESARA_HIP_UNROLL=2
run AESARA_HIP_STREAM_BPC=1 AESARA_HIP_UNROLL=4
run AESARA_HIP_STREAM_BPC=2 AESARA_HIP_UNROLL=4
run AESARA_HIP_STREAM_BPC=2
run A=default
