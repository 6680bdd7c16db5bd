#!/bin/bash
# round 6, job 29: rocprofv3 kernel stats of the embedding-RNN training step (default path only)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof_emb
cat > /tmp/emb_default.py <<'PY'
import os, sys
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from golden_util import CASES, case_plan
from aesara_amd import executor as E
plan = case_plan(next(c for c in CASES if c["name"] == "scan_embedding_lookup_batch_f32"))
T, B, V, H = 512, 64, 50000, 1024
rng = np.random.default_rng(1)
idx = torch.from_numpy(rng.integers(0, V, (T, B))).cuda()
Em = torch.from_numpy((rng.standard_normal((V, H)) * 0.5).astype("float32")).cuda()
U = torch.from_numpy((rng.standard_normal((H, H)) / np.sqrt(H)).astype("float32")).cuda()
h0 = torch.from_numpy((rng.standard_normal((B, H)) * 0.1).astype("float32")).cuda()
ex = E.PlanExecutor(plan, use_graph=False, borrow=True)
for _ in range(6):
    ex(idx, Em, U, h0)
torch.cuda.synchronize()
print(ex.scan_modes)
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_emb -o p -- python /tmp/emb_default.py 2>&1 | tail -2
f=$(find gpurun_out/prof_emb -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = open("gpurun_out/r06_embedding_rnn_kernel_stats.csv", "w")
w = csv.writer(out); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
for r in rows[:25]:
    w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
    print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY
