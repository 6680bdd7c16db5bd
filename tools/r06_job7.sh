#!/bin/bash
# round 6, job 7: (a) config 4 B = 1 with / without the early products (SP_EARLYDOTS) and the first-poll
# hold-back re-swept around them, + the timeline; (b) row chains with / without streaming loads;
# (c) the persistent-Scan GPU tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for e in 0 1 0 1; do
  AESARA_HIP_SP_EARLYDOTS=$e timeout 300 python tools/perf_probe.py --only gruB1 --out gpurun_out/r06_sp_early_$e.jsonl 2>/dev/null | grep "^{" | cut -c1-260
done
for dly in 8 11 13 15; do
  AESARA_HIP_SP_DELAY=$dly timeout 300 python tools/perf_probe.py --only gruB1 --out gpurun_out/r06_sp_delay_$dly.jsonl 2>/dev/null | grep "^{" | cut -c1-200
done
timeout 300 python tools/sp_trace.py > gpurun_out/r06_sp_trace_early.json 2>/dev/null
for nt in 0 1 0 1; do
  AESARA_HIP_RC_NT=$nt timeout 300 python tools/perf_probe.py --only softmax,layernorm --out gpurun_out/r06_rc_nt_$nt.jsonl 2>/dev/null | grep "^{" | cut -c1-220
done
timeout 900 python -m pytest tests/test_gpu_scan_persist.py tests/test_gpu_scan_elementwise.py -x -q -m gpu 2>&1 | tail -3
