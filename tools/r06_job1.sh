#!/bin/bash
# round 6, job 1: current Gemm / BatchedDot numbers away from the square case,
# and the config-2 headline with the correctly rounded division as the default
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/perf_probe.py --only gemmshapes --out gpurun_out/r06_gemm_shapes.jsonl > gpurun_out/r06_gemm_shapes.log 2>&1
for fd in 0 1 0 1; do
  AESARA_HIP_FASTDIV=$fd AESARA_BENCH_REGIONS=4 timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/r06_fastdiv_${fd}_$RANDOM.json 2>> gpurun_out/r06_fastdiv.err
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r06_parity1.log 2>&1
tail -3 gpurun_out/r06_parity1.log
