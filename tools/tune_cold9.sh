#!/bin/bash
# round 5, second pass: 16-byte non-temporal loads won the first (tune_cold8: 26.5 -> 25.06 us per eval);
# vectors in flight / workgroup size around that point, and the same switches on the other HBM-bound rows
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]
        print("%-66s window %.2f us (%.4f)  sustained %.4f  exec %.4f  ceiling %.4f" % (os.environ.get("TAG",""), r["roofline"]["kernel_ms"]*1e3, r["roofline"]["frac"], c["sustained"]["frac"], c["executor_level"]["frac"], c["read_only_ceiling"]["frac"]))
        for s in r.get("secondary", []):
            print("      %-64s %.2f us (%.3f)" % (s["config"][:64], s["roofline"]["kernel_ms"]*1e3, s["roofline"].get("frac") or 0))'
run() { TAG="$*" env "$@" timeout 300 python bench.py --no-cpu-baseline --no-warm --only-secondary cfg1b,cfg3a,cfg5,placed --steps 20 --warmup 5 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
run A=default
run AESARA_HIP_VECBYTES=16 AESARA_HIP_NT=1
run AESARA_HIP_VECBYTES=16 AESARA_HIP_NT=3
run AESARA_HIP_VECBYTES=16 AESARA_HIP_NT=1 AESARA_HIP_UNROLL=4
run AESARA_HIP_VECBYTES=16 AESARA_HIP_NT=1 AESARA_HIP_UNROLL=1
run AESARA_HIP_VECBYTES=16 AESARA_HIP_NT=1 AESARA_HIP_UNROLL=3
run AESARA_HIP_VECBYTES=16 AESARA_HIP_NT=1 AESARA_HIP_RED_BLOCK=512
run AESARA_HIP_VECBYTES=16 AESARA_HIP_NT=2
