#!/bin/bash
# Round-5 GPU job 8: quick correctness of the big-stream variants, then the rocprofv3 rows of the
# HBM-bound configs (kernel trace + FETCH / WRITE passes) under the streaming policy.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x --deselect "tests/test_gpu_fullsize.py::test_sampled_configs_at_full_shape_against_the_reference_c_linker" 2>&1 | tail -3
bash tools/profile_bench_r05.sh cfg2 cfg1b cfg3a cfg5 > gpurun_out/r05_profile_bench2.log 2>&1; tail -6 gpurun_out/r05_profile_bench2.log | cut -c1-700
