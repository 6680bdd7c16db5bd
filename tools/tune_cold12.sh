#!/bin/bash
# round 5: repeat A/B of the two candidates of tune_cold11 (alternating, same box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]
        print("%-50s window %.2f us (%.4f)  sustained %.4f" % (os.environ.get("TAG",""), r["roofline"]["kernel_ms"]*1e3, r["roofline"]["frac"], c["sustained"]["frac"]), end="")
        for s in r.get("secondary", []):
            print("   add %.2f us (%.3f)" % (s["roofline"]["kernel_ms"]*1e3, s["roofline"].get("frac") or 0), end="")
        print()'
run() { TAG="$*" env "$@" timeout 300 python bench.py --no-cpu-baseline --no-warm --only-secondary cfg1b --steps 20 --warmup 5 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
for i in 1 2 3; do
run A=default
run AESARA_HIP_RED_BLOCK=512 AESARA_HIP_RED_BPC=4 AESARA_HIP_BLOCK=1024 AESARA_HIP_STREAM_BPC=2
done
