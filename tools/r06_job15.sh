#!/bin/bash
# round 6, job 15: full-line C stores in the GEMM epilogue (v_permlane16_swap), parity then A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu -k "gemm or Gemm or dot or Dot or blas or fuzz or cfg3" 2>&1 | tail -4
for nc in 0 1 0 1; do
  if [ $nc = 1 ]; then export AHIP_GEMM_NARROW_C=1; else unset AHIP_GEMM_NARROW_C; fi
  timeout 600 python tools/perf_probe.py --only gemmshapes,cfg3b --out gpurun_out/r06_gemm_widec_$nc.jsonl 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('NARROW_C=%s %-62s %8.2f us %8.1f (%.3f)'%('$nc', r['config'], r['dev_ms']*1e3, r['achieved'], r['frac']))"
done
