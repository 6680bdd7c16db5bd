#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_exp64.py tests/test_gpu_reference_files.py -m gpu -q -p no:cacheprovider > gpurun_out/r05_job2_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_job2_tests.log; tail -25 gpurun_out/r05_job2_tests.log
timeout 300 python tools/handoff_floor.py > gpurun_out/r05_handoff_floor.json 2> gpurun_out/r05_handoff_floor.err; echo "handoff rc=$?"
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r05_handoff_floor.json")); print(json.dumps(d["best_per_grid"]))
except Exception as e: print("handoff:", e)
PY
for fd in 0 1 0 1; do
  AESARA_HIP_FASTDIV=$fd AESARA_BENCH_REGIONS=8 timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/r05_fastdiv_$fd.json 2> gpurun_out/r05_fastdiv_$fd.err
  grep -h "^regions" gpurun_out/r05_fastdiv_$fd.json gpurun_out/r05_fastdiv_$fd.err | cut -c1-400
  python - $fd <<'PY'
import json,sys
fd=sys.argv[1]
l=json.loads([x for x in open("gpurun_out/r05_fastdiv_%s.json"%fd) if x.startswith("{")][-1])
print("FASTDIV",fd,"frac",round(l["roofline"]["frac"],4),"kernel_us",round(l["roofline"]["kernel_ms"]*1e3,2),"sustained",round(l["config"]["sustained"]["frac"],4), "regions", l["config"].get("regions") or l.get("regions"))
PY
done
