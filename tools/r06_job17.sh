#!/bin/bash
# round 6, job 17: Scans without recurrence as one evaluation: GPU tests, the reference's scan / gradient files on the device
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_scan_elementwise.py tests/test_gpu_parity.py -x -q -m gpu -k "recurrence or jacobian or scan_map or nitsot or elementwise_scans" 2>&1 | tail -5
rm -f /tmp/scanlog.jsonl
AESARA_HIP_SUITE_TEST_TIMEOUT=300 timeout 1500 python tests/reference_files.py --executor device --workers 4 tests/test_rop.py tests/scan/test_rewriting.py tests/scan/test_basic.py tests/test_gradient.py tests/tensor/test_shape.py tests/scan/test_checkpoints.py tests/scan/test_views.py 2>&1 | grep -v "asserts a destroy" | tail -30
