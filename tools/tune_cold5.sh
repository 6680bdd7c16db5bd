#!/bin/bash
# round-4 sweep 2 (after the head of the kernel issues every load before it waits): walks of the stream
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]; ce=c["read_only_ceiling"]
        print("%-58s cold %.2f us (%.3f) | line %.2f us | sum-only %.2f us (%.3f)" % (os.environ.get("TAG",""), c["sustained"]["kernel_ms"]*1e3, c["sustained"]["frac"], r["roofline"]["kernel_ms"]*1e3, ce["kernel_ms"]*1e3, ce["frac"]))'
run() { TAG="$*" env "$@" timeout 240 python bench.py --no-secondary --no-cpu-baseline --no-warm --steps 20 --warmup 5 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
run A=default
run AESARA_HIP_RED_BLOCKED=1
run AESARA_HIP_RED_BLOCKED=2
run AESARA_HIP_EARLY=0
run AESARA_HIP_FASTEXP=0
run AESARA_HIP_RED_BLOCKED=1 AESARA_HIP_RED_BPC=4
run AESARA_HIP_RED_BLOCKED=2 AESARA_HIP_UNROLL=1
run AESARA_HIP_RED_BLOCKED=2 AESARA_HIP_RED_BLOCK=512
run A=default
run AESARA_HIP_RED_BLOCKED=2
