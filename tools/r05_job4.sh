#!/bin/bash
# Round-5 GPU job 4: the element-wise Scan kernel's reduction form + the memoized host path on the
# device, then the headline window (AESARA_BENCH_REGIONS=8: eight driver-shaped windows) twice.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_scan_elementwise.py "tests/test_gpu_function_e2e.py" "tests/test_gpu_parity.py" -m gpu -q -p no:cacheprovider \
  -k "elementwise or memoized or scan_red or cfg2 or untrusted or training_loop" --maxfail=10 > gpurun_out/r05_job4_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_job4_tests.log; tail -30 gpurun_out/r05_job4_tests.log | cut -c1-300
for k in 1 2; do
  AESARA_BENCH_REGIONS=8 timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/r05_job4_bench_$k.json 2> gpurun_out/r05_job4_bench_$k.err
  grep -h "^regions" gpurun_out/r05_job4_bench_$k.json gpurun_out/r05_job4_bench_$k.err | cut -c1-400
  python - $k <<'PY'
import json,sys
k=sys.argv[1]
l=json.loads([x for x in open("gpurun_out/r05_job4_bench_%s.json"%k) if x.startswith("{")][-1])
s=l["config"]["sustained"]
print("run",k,"value",round(l["value"],1),"ms_per_step",l["ms_per_step"],"frac",round(l["roofline"]["frac"],4),"kernel_us",round(l["roofline"]["kernel_ms"]*1e3,2),"sustained",round(s["frac"],4),"host_us_per_call",round(s.get("host_us_per_call",0),2))
PY
done
