#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fmt='import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); print("%-28s %-52s %8.2f us  %5.0f GB/s (%.3f)" % (" ".join("%s=%s"%(k[11:],v) for k,v in r["tune"].items()), r["config"][:52], r["dev_ms"]*1e3, r["achieved"], r["frac"]))'
for e in "A=1" "AESARA_HIP_NT=0"; do
  env $e timeout 300 python tools/perf_probe.py --only axisred,redsum --out gpurun_out/r05_axisred.jsonl 2>&1 | python -c "$fmt"
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_hfuse.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x --deselect "tests/test_gpu_fullsize.py::test_sampled_configs_at_full_shape_against_the_reference_c_linker" 2>&1 | tail -5
