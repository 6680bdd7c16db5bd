#!/bin/bash
# Round-5 GPU job 6: the driver's bench line (default flags), then the whole -m gpu suite.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
echo "bench rc=$?"; tail -c 600 gpurun_out/r05_bench_line.err
python - <<'PY'
import json
l=json.loads([x for x in open("gpurun_out/r05_bench_line.json") if x.startswith("{")][-1])
print("value",round(l["value"],1),"ms_per_step",round(l["ms_per_step"],5),"frac",round(l["roofline"]["frac"],4),"sustained",round(l["config"]["sustained"]["frac"],4))
for s in l.get("secondary",[]):
    r=s.get("roofline",{})
    print("  ",s["config"][:70], round(r.get("kernel_ms",0),4),"ms", r.get("frac") and round(r["frac"],3), r.get("us_per_step") and round(r["us_per_step"],2))
print("vs_reference", json.dumps(l.get("vs_reference"))[:600])
PY
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 --durations=8 -p no:cacheprovider > gpurun_out/r05_gpu_suite.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_gpu_suite.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" gpurun_out/r05_gpu_suite.log | tail -25
