#!/bin/bash
# second MALL-cold sweep of config 2: bytes in flight per lane (vector width x unroll), grid shape,
# and a size sweep (fixed cost vs marginal HBM rate)
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]
        print("%-72s cold %.2f us (%.3f)  warm %.2f us (%.3f)" % (os.environ.get("TAG",""), c["sustained"]["kernel_ms"]*1e3, c["sustained"]["frac"], c["warm"]["kernel_ms"]*1e3, c["warm"]["frac"]))'
run() { TAG="$*" env "$@" timeout 180 python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | TAG="$*" python -c "$fmt"; }
run A=0
run AESARA_HIP_UNROLL=2
run AESARA_HIP_UNROLL=4
run AESARA_HIP_UNROLL=8
run AESARA_HIP_VECBYTES=16 AESARA_HIP_UNROLL=2
run AESARA_HIP_VECBYTES=16 AESARA_HIP_UNROLL=4
run AESARA_HIP_VECBYTES=16 AESARA_HIP_UNROLL=8
run AESARA_HIP_VECBYTES=64 AESARA_HIP_UNROLL=2
run AESARA_HIP_UNROLL=2 AESARA_HIP_RED_BPC=1
run AESARA_HIP_UNROLL=4 AESARA_HIP_RED_BPC=1
run AESARA_HIP_UNROLL=2 AESARA_HIP_RED_BLOCK=512 AESARA_HIP_RED_BPC=4
run AESARA_HIP_UNROLL=2 AESARA_HIP_RED_BLOCK=512 AESARA_HIP_RED_BPC=2
run AESARA_HIP_UNROLL=4 AESARA_HIP_RED_BLOCK=256 AESARA_HIP_RED_BPC=4
run AESARA_HIP_UNROLL=2 AESARA_BENCH_ROWS=2048
run AESARA_HIP_UNROLL=2 AESARA_BENCH_ROWS=8192
run AESARA_HIP_UNROLL=2 AESARA_BENCH_ROWS=16384
run AESARA_BENCH_ROWS=16384
run A=0
