#!/bin/bash
# round 6, job 14: the GRU training step (B = 64) with the forward Scan's in-loop sequence products off
# (all three x @ W products up front): what would sharing them with the gradient Scan be worth?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for xf in 1 0 1 0; do
  echo "== AESARA_HIP_SM_XFOLD=$xf"
  AESARA_HIP_SM_XFOLD=$xf timeout 300 python tools/bptt_probe.py 512 1024 64 2>/dev/null | grep "^{" | cut -c1-260
done
