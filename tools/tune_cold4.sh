#!/bin/bash
# round-4 MALL-cold sweep of config 2: early first loads, table-driven fp64 exp, one-hop finalize,
# blocked walk, launch shapes; every line also shows the read-only ceiling (x.sum()) under the same knobs
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]; ce=c["read_only_ceiling"]
        print("%-58s cold %.2f us (%.3f) | line %.2f us | sum-only %.2f us (%.3f)" % (os.environ.get("TAG",""), c["sustained"]["kernel_ms"]*1e3, c["sustained"]["frac"], r["roofline"]["kernel_ms"]*1e3, ce["kernel_ms"]*1e3, ce["frac"]))'
run() { TAG="$*" env "$@" timeout 240 python bench.py --no-secondary --no-cpu-baseline --no-warm --steps 200 --warmup 20 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
run AESARA_HIP_EARLY=0 AESARA_HIP_FASTEXP=0
run AESARA_HIP_EARLY=1 AESARA_HIP_FASTEXP=0
run AESARA_HIP_EARLY=0 AESARA_HIP_FASTEXP=1
run A=default
run AESARA_HIP_RED_BLOCKED=1
run AESARA_HIP_UNROLL=1
run AESARA_HIP_UNROLL=4
run AESARA_HIP_RED_BLOCK=512
run AESARA_HIP_RED_BLOCK=256
run AESARA_HIP_NT=1
run AESARA_HIP_RED_BPC=4
run AESARA_HIP_RED_BPC=6
run AESARA_HIP_RED_BLOCKED=1 AESARA_HIP_UNROLL=4
run AESARA_HIP_RED_BLOCKED=1 AESARA_HIP_NT=1
run A=default
