#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AESARA_HIP_SUITE_TEST_TIMEOUT=300 timeout 900 python tests/reference_files.py --executor device --workers 4 tests/tensor/nnet/test_batchnorm.py tests/tensor/test_blas_c.py tests/tensor/test_blas_scipy.py tests/graph/test_compute_test_value.py tests/tensor/test_utils.py tests/tensor/test_gc.py tests/tensor/test_misc.py tests/tensor/test_io.py tests/tensor/test_type_other.py tests/tensor/nnet/test_rewriting.py tests/compile/test_misc.py tests/test_printing.py 2>&1 | tail -32
