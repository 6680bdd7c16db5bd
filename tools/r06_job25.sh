#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_scan_persist.py tests/test_gpu_parity.py -x -q -m gpu -k "embedding" 2>&1 | grep -v Warn | tail -30
