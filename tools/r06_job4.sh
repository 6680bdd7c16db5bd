#!/bin/bash
# round 6, job 4: the GEMM's persistent tile loop, A/B (AESARA_HIP_GEMM_PERSIST_WG: 99 = one tile per
# workgroup, the form of round 5; 2 = default; 1, 3) on the square headline and the shapes off it
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for pw in 99 2 1 3 99 2; do
  export AESARA_HIP_GEMM_PERSIST_WG=$pw
  timeout 600 python tools/perf_probe.py --only gemmshapes,cfg3b --out gpurun_out/r06_gemm_persist_$pw.jsonl > gpurun_out/r06_gemm_persist_$pw.log 2>&1
  tail -2 gpurun_out/r06_gemm_persist_$pw.log | cut -c1-200
done
unset AESARA_HIP_GEMM_PERSIST_WG
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "gemm or dot or Gemm or Dot or blas" > gpurun_out/r06_parity4.log 2>&1
tail -3 gpurun_out/r06_parity4.log
