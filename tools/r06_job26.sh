#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python tools/r06_probe_embedding_rnn.py 2>&1 | grep "^{\|Error\|error" | tee gpurun_out/r06_embedding_rnn.jsonl
