#!/bin/bash
# round 6, job 19: the whole -m gpu suite (time budget 1200 s) with the extended reference-file list
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r06_gpu_suite_tail.log
