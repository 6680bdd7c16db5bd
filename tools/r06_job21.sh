#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_errors.py -x -q -m gpu -k "host_scalars" 2>&1 | grep -v Warning | tail -40
