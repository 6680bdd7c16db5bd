#!/bin/bash
# MALL-cold config 2 (bench.py headline, rotating over 8 x 128 MiB inputs): launch-shape /
# load-policy sweep.  Prints cold (rotated, 2000 evals) and warm (same buffer) device us per eval.
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]
        print("%-64s cold %.2f us (%.3f)  warm %.2f us (%.3f)" % (os.environ.get("TAG",""), c["sustained"]["kernel_ms"]*1e3, c["sustained"]["frac"], c["warm"]["kernel_ms"]*1e3, c["warm"]["frac"]))'
run() { TAG="$*" env "$@" timeout 180 python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | TAG="$*" python -c "$fmt"; }
run A=0
run AESARA_HIP_NT=1
run AESARA_HIP_RED_BPC=4
run AESARA_HIP_RED_BPC=1 
run AESARA_HIP_RED_BLOCK=512 AESARA_HIP_RED_BPC=4
run AESARA_HIP_RED_BLOCK=256 AESARA_HIP_RED_BPC=8
run AESARA_HIP_RED_BLOCK=256 AESARA_HIP_RED_BPC=8 AESARA_HIP_NT=1
run AESARA_HIP_UNROLL=2
run AESARA_HIP_UNROLL=2 AESARA_HIP_NT=1
run AESARA_HIP_VECBYTES=16
run AESARA_HIP_VECBYTES=64
run AESARA_HIP_VECBYTES=64 AESARA_HIP_NT=1
run A=0
