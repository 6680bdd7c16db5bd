#!/bin/bash
# Round-5 GPU job 3: the whole -m gpu suite with the tolerance-mode quotient as the default, the
# rocprofv3 evidence for every driver-line row, the GEMM counter set, and a poll-geometry sweep of
# the config-4 B = 1 kernel.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 --durations=8 -p no:cacheprovider > gpurun_out/r05_gpu_suite.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_gpu_suite.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" gpurun_out/r05_gpu_suite.log | tail -25
bash tools/profile_bench_r05.sh > gpurun_out/r05_profile_bench.log 2>&1; tail -8 gpurun_out/r05_profile_bench.log | cut -c1-600
bash tools/gemm_pmc_r05.sh > gpurun_out/r05_gemm_pmc.log 2>&1; tail -6 gpurun_out/r05_gemm_pmc.log | cut -c1-500
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l)
        for s in r.get("secondary", []):
            if "B=1" in s["config"]:
                print("%-50s %s: %.3f ms  %.3f us/step" % (os.environ.get("TAG",""), s["config"][:60], s["roofline"]["kernel_ms"], s["roofline"].get("us_per_step", 0)))'
run() { TAG="$*" env "$@" timeout 240 python bench.py --no-cpu-baseline --no-warm --executor-level --only-secondary cfg4 --steps 20 --warmup 5 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
{ run A=default; run AESARA_HIP_SP_POLLW=8 AESARA_HIP_SP_DELAY=12; run AESARA_HIP_SP_POLLW=8 AESARA_HIP_SP_DELAY=15; run AESARA_HIP_SP_POLLW=4 AESARA_HIP_SP_DELAY=13; run A=default; } > gpurun_out/r05_scan_cfg4_b1_poll_sweep.txt 2>&1
cat gpurun_out/r05_scan_cfg4_b1_poll_sweep.txt
