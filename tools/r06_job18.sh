#!/bin/bash
# round 6, job 18: Scans without recurrence: one evaluation over whole sequences vs the launch list
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python tools/r06_probe_jacobian.py 2>/dev/null | grep "^{" | tee gpurun_out/r06_scan_all_rows.jsonl
