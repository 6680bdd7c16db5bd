#!/bin/bash
# round 6, job 20: host-scalar fix on the device + the scalar files of the reference
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_errors.py tests/test_gpu_parity.py -x -q -m gpu -k "host_scalars or ultra_fast" 2>&1 | tail -3
AESARA_HIP_SUITE_TEST_TIMEOUT=300 timeout 900 python tests/reference_files.py --executor device --workers 4 tests/scalar/test_basic.py tests/scalar/test_math.py tests/tensor/nnet/test_sigm.py tests/compile/test_builders.py 2>&1 | tail -25
