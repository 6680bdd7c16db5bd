#!/bin/bash
# the reference's tests/scan/test_basic.py under mode HIP on the device (after the general mit-mot rule / IfElse switch)
cd $GRAFT_REPO_ROOT/tests
timeout 58 python - <<'PY'
import sys, collections
sys.path.insert(0, "."); sys.path.insert(0, "../oracle")
import reference_files as rf
rep = rf.run("device", ["tests/scan/test_basic.py"], workers=12, timeout=52)
c = collections.Counter(v[0] for v in rep.values())
print(dict(c))
bad = [(k, v) for k, v in rep.items() if v[0] == "failed" and "UnsupportedOp" not in (v[1] or "") and "monitor_mode" not in k and "test_grad_multiple_outs_taps" not in k]
print("unexpected failures:", [(k, (v[1] or "")[:150]) for k, v in bad])
PY
