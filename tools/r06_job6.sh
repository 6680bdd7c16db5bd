#!/bin/bash
# round 6, job 6: the whole GPU suite (timing: limit 1200 s), then the training-step GEMM shapes
# under the GROUP walk settings
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 2300 python -m pytest tests -q -m gpu) > gpurun_out/r06_gpu_suite.log 2>&1; tail -6 gpurun_out/r06_gpu_suite.log
for g in 8 4 2 16 1; do
  AESARA_HIP_GEMM_GROUP=$g timeout 300 python tools/r06_probe_gemm_tall.py 2>/dev/null | grep "^{" >> gpurun_out/r06_gemm_tall.jsonl
done
cat gpurun_out/r06_gemm_tall.jsonl | cut -c1-220
