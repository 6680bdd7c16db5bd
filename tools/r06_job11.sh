#!/bin/bash
# round 6, job 11: axis reductions cold after the grid-stride walk over outputs (row mode), A/B by the
# number of resident workgroups per CU; parity of the reductions
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for bpc in 16 8 32 4096; do
  AESARA_HIP_RED_ROW_BPC=$bpc PROBE_ROTATE=1 timeout 600 python tools/perf_probe.py --only axisred --out gpurun_out/r06_axisred_$bpc.jsonl 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('RED_ROW_BPC=%-5s %-62s %8.2f us %6.0f GB/s (%.3f)'%('$bpc', r['config'], r['dev_ms']*1e3, r['achieved'], r['frac']))"
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "red or sum or max or careduce or CAReduce or fuzz" 2>&1 | tail -3
