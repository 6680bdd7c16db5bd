#!/bin/bash
# round 5, third pass: the streaming policy (exec_elemwise.BIG_STREAM) as the default against the old
# defaults (AESARA_HIP_VECBYTES=32 AESARA_HIP_NT=0 switches the policy off), all HBM-bound rows
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]
        print("%-46s window %.2f us (%.4f)  sustained %.4f  exec %.4f  ceiling %.4f" % (os.environ.get("TAG",""), r["roofline"]["kernel_ms"]*1e3, r["roofline"]["frac"], c["sustained"]["frac"], c["executor_level"]["frac"], c["read_only_ceiling"]["frac"]))
        for s in r.get("secondary", []):
            print("      %-64s %.2f us (%.3f)" % (s["config"][:64], s["roofline"]["kernel_ms"]*1e3, s["roofline"].get("frac") or 0))'
run() { TAG="$*" env "$@" timeout 300 python bench.py --no-cpu-baseline --no-warm --only-secondary cfg1b,cfg3a,cfg5,placed --steps 20 --warmup 5 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
run A=default
run AESARA_HIP_VECBYTES=32 AESARA_HIP_NT=0
run A=default
run AESARA_HIP_VECBYTES=32 AESARA_HIP_NT=0
