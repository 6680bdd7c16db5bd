#!/bin/bash
# Round-5 GPU job 1: the whole -m gpu suite (incl. the reference's own files under mode=HIP, the
# fp64 exp edges, the full-shape reference checks) and one default bench.py run.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
free -g | head -2 > gpurun_out/r05_host.txt; nproc >> gpurun_out/r05_host.txt
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 --durations=15 -p no:cacheprovider > gpurun_out/r05_gpu_suite.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_gpu_suite.log
tail -30 gpurun_out/r05_gpu_suite.log
timeout 900 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench.err
echo "bench rc=$?"; tail -c 1500 gpurun_out/r05_bench.err; head -c 3000 gpurun_out/r05_bench_line.json
