#!/bin/bash
# round 6, job 9: software-pipelined body of the flat full reduction (config 2), A/B on one box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]
        print("%-24s window %.2f us (%.4f)  sustained %.4f  exec %.4f  ceiling %.4f" % (os.environ.get("TAG",""), r["roofline"]["kernel_ms"]*1e3, r["roofline"]["frac"], c["sustained"]["frac"], c["executor_level"]["frac"], c["read_only_ceiling"]["frac"]))'
run() { TAG="$*" env "$@" timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
run AESARA_HIP_SWPIPE=0
run AESARA_HIP_SWPIPE=1
run AESARA_HIP_SWPIPE=0
run AESARA_HIP_SWPIPE=1
run AESARA_HIP_SWPIPE=1 AESARA_HIP_UNROLL=1
run AESARA_HIP_SWPIPE=1 AESARA_HIP_UNROLL=4
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "cfg2 or reduce or sum" 2>&1 | tail -2
