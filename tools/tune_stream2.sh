#!/bin/bash
# round-4: walks / launch shapes of the plain flat Elemwise stream (BASELINE config 1b: fp64 4096^2 add, 402 MB)
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l)
        for s in r.get("secondary", []):
            print("%-60s %s: %.2f us (%.3f)" % (os.environ.get("TAG",""), s["config"][:28], s["roofline"]["kernel_ms"]*1e3, s["roofline"]["frac"]))'
run() { TAG="$*" env "$@" timeout 240 python bench.py --no-cpu-baseline --no-warm --executor-level --only-secondary cfg1b --steps 20 --warmup 5 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
run A=default
run AESARA_HIP_STREAM_BLOCKED=1
run AESARA_HIP_STREAM_BLOCKED=2
run AESARA_HIP_STREAM_BLOCKED=2 AESARA_HIP_STREAM_BPC=4
run AESARA_HIP_STREAM_BLOCKED=2 AESARA_HIP_BLOCK=1024 AESARA_HIP_STREAM_BPC=16
run AESARA_HIP_STREAM_BLOCKED=2 AESARA_HIP_BLOCK=1024 AESARA_HIP_STREAM_BPC=8
run AESARA_HIP_BLOCK=1024 AESARA_HIP_STREAM_BPC=8
run AESARA_HIP_STREAM_BLOCKED=2 AESARA_HIP_NT=3
run AESARA_HIP_STREAM_BLOCKED=2 AESARA_HIP_UNROLL=2
run A=default
