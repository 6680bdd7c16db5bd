#!/bin/bash
# round 6, job 8: row chains, non-temporal stores on top of the streaming loads
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for st in 0 1 0 1; do
  AESARA_HIP_RC_NTST=$st timeout 300 python tools/perf_probe.py --only softmax,layernorm --out gpurun_out/r06_rc_ntst_$st.jsonl 2>/dev/null | grep "^{" | grep -v UNFUSED | cut -c1-220
done
