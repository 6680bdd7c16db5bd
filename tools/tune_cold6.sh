#!/bin/bash
# (historical sweep: AESARA_HIP_PIPE / AESARA_HIP_RED_PRIO measured null — profiles/r03_*, r04_cfg2_cold_sweep* — and were removed from the generator in round 5; those rows are no-ops now)
# round-4 sweep 3: waves per CU x vectors in flight x walk x priority of the younger workgroup
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]; ce=c["read_only_ceiling"]
        print("%-70s cold %.2f us (%.3f) | line %.2f us | sum-only %.2f us (%.3f)" % (os.environ.get("TAG",""), c["sustained"]["kernel_ms"]*1e3, c["sustained"]["frac"], r["roofline"]["kernel_ms"]*1e3, ce["kernel_ms"]*1e3, ce["frac"]))'
run() { TAG="$*" env "$@" timeout 240 python bench.py --no-secondary --no-cpu-baseline --no-warm --steps 20 --warmup 5 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
run A=default
run AESARA_HIP_RED_BLOCKED=2
run AESARA_HIP_RED_BLOCKED=2 AESARA_HIP_RED_PRIO=1
run AESARA_HIP_RED_BLOCKED=2 AESARA_HIP_RED_PRIO=3
run AESARA_HIP_RED_PRIO=2
run AESARA_HIP_RED_BLOCKED=2 AESARA_HIP_UNROLL=4
run AESARA_HIP_RED_BLOCKED=2 AESARA_HIP_RED_BPC=4 AESARA_HIP_UNROLL=4
run AESARA_HIP_RED_BPC=4 AESARA_HIP_UNROLL=4
run AESARA_HIP_RED_BLOCKED=2 AESARA_HIP_RED_BPC=4 AESARA_HIP_UNROLL=8
run AESARA_HIP_RED_BLOCKED=2 AESARA_HIP_RED_BPC=4
run AESARA_HIP_RED_BLOCKED=2 AESARA_HIP_NT=1
run AESARA_HIP_RED_BLOCKED=2
