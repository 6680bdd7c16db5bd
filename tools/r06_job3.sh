#!/bin/bash
# round 6, job 3: the reference's own test files on the device (log -> profiles/), then the bench
# line with the BatchedDot / Dot22 rows
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests/test_gpu_reference_files.py -x -q -m gpu) > gpurun_out/r06_reffiles_pytest.log 2>&1
tail -4 gpurun_out/r06_reffiles_pytest.log
(time timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err)
tail -c 600 gpurun_out/r06_bench_a.err
