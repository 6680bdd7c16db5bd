#!/bin/bash
# round 6, job 2: Gemm shapes after (a) 64x64 tiles for mostly-empty big tiles, (b) K groups inside the
# 64x64-tile workgroup; A/B of the K groups; GPU parity; device checks of the special functions
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for ks in 1 2 4 auto; do
  if [ $ks = auto ]; then unset AESARA_HIP_GEMM_HALF_KSPLIT; else export AESARA_HIP_GEMM_HALF_KSPLIT=$ks; fi
  timeout 600 python tools/perf_probe.py --only gemmshapes --out gpurun_out/r06_gemm_shapes_ks$ks.jsonl > gpurun_out/r06_gemm_shapes_ks$ks.log 2>&1
done
unset AESARA_HIP_GEMM_HALF_KSPLIT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r06_parity2.log 2>&1
tail -3 gpurun_out/r06_parity2.log
