#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_scan_elementwise.py tests/test_gpu_parity.py -x -q -m gpu -k "recurrence or jacobian or scan_map or gemv_runtime" 2>&1 | grep -v Warn | tail -30
timeout 600 python tools/r06_probe_jacobian.py 2>/dev/null | grep "^{" | cut -c1-250
