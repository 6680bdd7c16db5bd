#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_scan_persist.py tests/test_gpu_parity.py tests/test_gpu_function_e2e.py -x -q -m gpu -k "embedding or row_lookup" 2>&1 | grep -v Warn | tail -4
timeout 900 python tools/r06_probe_embedding_rnn.py 2>&1 | grep "^{\|Error\|error" | tee gpurun_out/r06_embedding_rnn.jsonl
bash tools/r06_job29.sh 2>&1 | tail -14
