#!/bin/bash
# round 6, job 10: the bench line with the driver's flags (-> profiles/r06_bench_line.json)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err)
tail -c 400 gpurun_out/r06_bench_line.err
