#!/bin/bash
# round 6, job 16: GEMMs on a second stream beside the persistent matrix-state Scan kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/r06_probe_overlap.py 2>/dev/null | grep "^{" | tee gpurun_out/r06_scan_gemm_overlap.json
