#!/bin/bash
# round 6, job 27: 64x64 tiles for short-K products with MANY tiles? (4096 x 4096 x 64 sits at 0.42)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for hm in default 100000 default 100000; do
  if [ $hm = default ]; then unset AESARA_HIP_GEMM_HALF_MAX; else export AESARA_HIP_GEMM_HALF_MAX=$hm; fi
  timeout 600 python tools/perf_probe.py --only gemmshapes 2>/dev/null | grep "^{" | grep "float32" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('HALF_MAX=%-8s %-62s %8.2f us %8.1f (%.3f)'%('$hm', r['config'], r['dev_ms']*1e3, r['achieved'], r['frac']))"
done
