#!/bin/bash
# (historical sweep: AESARA_HIP_PIPE / AESARA_HIP_RED_PRIO measured null — profiles/r03_*, r04_cfg2_cold_sweep* — and were removed from the generator in round 5; those rows are no-ops now)
# third MALL-cold sweep of config 2: ping-pong software pipeline (AESARA_HIP_PIPE) x vectors per group
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]
        print("%-60s cold %.2f us (%.3f)  warm %.2f us (%.3f)" % (os.environ.get("TAG",""), c["sustained"]["kernel_ms"]*1e3, c["sustained"]["frac"], c["warm"]["kernel_ms"]*1e3, c["warm"]["frac"]))'
run() { TAG="$*" env "$@" timeout 180 python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
run A=0
run AESARA_HIP_PIPE=1
run AESARA_HIP_PIPE=1 AESARA_HIP_UNROLL=1
run AESARA_HIP_PIPE=1 AESARA_HIP_UNROLL=3
run AESARA_HIP_PIPE=1 AESARA_HIP_UNROLL=4
run AESARA_HIP_PIPE=1 AESARA_HIP_UNROLL=1 AESARA_HIP_VECBYTES=64
run AESARA_HIP_PIPE=1 AESARA_HIP_RED_BPC=1
run AESARA_HIP_PIPE=1 AESARA_HIP_NT=1
run AESARA_HIP_PIPE=1 AESARA_BENCH_ROWS=16384
run A=0
