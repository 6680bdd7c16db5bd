#!/bin/bash
# TEST INFRASTRUCTURE ONLY (authoring container).  Packs the importable overlay of the reference
# front end (build_ref_overlay.sh) into oracle/_ref/aesara_ref_overlay.tar.gz so that it travels
# to the GPU box with the gpurun snapshot, the same way the built .so files do: oracle/_ref/ is
# git-ignored (a build artefact, never committed — no reference source enters the history) but
# not gpurun-ignored.  On the GPU box oracle/ref_overlay.py unpacks it under /tmp; it feeds
#   * tests/test_gpu_function_e2e.py  (aesara.function(mode="HIP") over the real PlanExecutor),
#   * bench.py's  cpu_baseline  (the reference's own Mode("cvm","fast_run") on the box's host)
#     and  through_function  legs.
# The product package never imports it.
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
OVL=$("$HERE/build_ref_overlay.sh")
mkdir -p "$HERE/_ref"
tar -C "$(dirname "$OVL")" --exclude='__pycache__' --exclude='*.pyc' \
    -czf "$HERE/_ref/aesara_ref_overlay.tar.gz" "$(basename "$OVL")"
ls -l "$HERE/_ref/aesara_ref_overlay.tar.gz"
