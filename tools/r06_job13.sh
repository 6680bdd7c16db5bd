#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for w in 0 1024 512 0 1024; do
AESARA_HIP_COL_WIDE=$w PROBE_ROTATE=1 timeout 600 python tools/perf_probe.py --only axisred --out gpurun_out/r06_axisred_wide_$w.jsonl 2>/dev/null | grep "^{" | grep "axis=(0,)" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('COL_WIDE=%-5s %-62s %8.2f us %6.0f GB/s (%.3f)'%('$w', r['config'], r['dev_ms']*1e3, r['achieved'], r['frac']))"
done
AESARA_HIP_COL_WIDE=1024 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "red or sum or max or careduce or CAReduce or fuzz" 2>&1 | tail -3
