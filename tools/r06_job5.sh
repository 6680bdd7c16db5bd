#!/bin/bash
# round 6, job 5: the reference's test files on the device with a per-test time limit (a hang is
# then ONE failed test), then the s_memtime timeline of the vector-state persistent Scan kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time AESARA_HIP_SUITE_TEST_TIMEOUT=120 timeout 1700 python -m pytest tests/test_gpu_reference_files.py -x -q -m gpu) > gpurun_out/r06_reffiles_pytest.log 2>&1
tail -4 gpurun_out/r06_reffiles_pytest.log
grep -c "UNEXPLAINED " gpurun_out/r06_reference_files.log
timeout 300 python tools/sp_trace.py > gpurun_out/r06_sp_trace.json 2> gpurun_out/r06_sp_trace.err; tail -2 gpurun_out/r06_sp_trace.err
