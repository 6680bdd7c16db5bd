#!/bin/bash
# round 5, fourth pass (streaming policy on): workgroup shapes of the reduction and of the add stream
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]
        print("%-62s window %.2f us (%.4f)  sustained %.4f  exec %.4f  ceiling %.4f" % (os.environ.get("TAG",""), r["roofline"]["kernel_ms"]*1e3, r["roofline"]["frac"], c["sustained"]["frac"], c["executor_level"]["frac"], c["read_only_ceiling"]["frac"]))
        for s in r.get("secondary", []):
            print("      %-64s %.2f us (%.3f)" % (s["config"][:64], s["roofline"]["kernel_ms"]*1e3, s["roofline"].get("frac") or 0))'
run() { TAG="$*" env "$@" timeout 300 python bench.py --no-cpu-baseline --no-warm --only-secondary cfg1b --steps 20 --warmup 5 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
run A=default
run AESARA_HIP_RED_BLOCK=512
run AESARA_HIP_RED_BLOCK=512 AESARA_HIP_RED_BPC=4
run AESARA_HIP_RED_BLOCKED=0
run AESARA_HIP_RED_BLOCKED=1
run AESARA_HIP_STREAM_BPC=4
run AESARA_HIP_STREAM_BPC=16
run AESARA_HIP_BLOCK=512 AESARA_HIP_STREAM_BPC=4
run AESARA_HIP_BLOCK=1024 AESARA_HIP_STREAM_BPC=2
run AESARA_HIP_STREAM_BLOCKED=2
run A=default
