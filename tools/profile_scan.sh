#!/bin/bash
# GPU box: rocprofv3 evidence for BASELINE config 4 (Scan GRU T=512 H=1024 fp32 B=1), persistent
# one-kernel loop vs the launch-list path: per-kernel durations and FETCH_SIZE (HBM-side reads)
# per evaluation.  Eager launches (--graph 0) so every kernel is its own dispatch record.
# usage: tools/profile_scan.sh [TAG] [PROBE]     PROBE = gruB1 (default) | cfg4b64 (matrix state, B = 64)
set -u
TAG=${1:-r02}
PROBE=${2:-gruB1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python $R/tools/perf_probe.py --only $PROBE --graph 0 --out /dev/null"
for mode in 1 0; do
  AESARA_HIP_SCAN_PERSIST=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $O/scan_kt$mode -o s -- $P > $O/scan_kt$mode.log 2>&1
  AESARA_HIP_SCAN_PERSIST=$mode rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/scan_f$mode -o s -- $P > $O/scan_f$mode.log 2>&1
  AESARA_HIP_SCAN_PERSIST=$mode rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/scan_w$mode -o s -- $P > $O/scan_w$mode.log 2>&1
done
python - <<PY
import csv, json, glob, os, collections
O="$O"; TAG="$TAG"; PROBE="$PROBE"
def find(d, suffix):
    c = glob.glob(os.path.join(O, d, "**", "*" + suffix), recursive=True)
    return c[0] if c else None
res = {}
for mode, label in ((1, "persistent"), (0, "launch_list")):
    e = {}
    p = find("scan_kt%d" % mode, "kernel_stats.csv")
    rows = list(csv.DictReader(open(p))) if p else []
    e["kernel_stats"] = [{"name": r["Name"][:60], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                          "total_ns": int(r["TotalDurationNs"]), "pct": float(r["Percentage"])}
                         for r in rows if r["Name"].startswith(("sp_", "sm_", "gv_", "ge_", "ew_", "void (anonymous"))][:8]
    for ctr, d in (("FETCH_SIZE", "scan_f%d" % mode), ("WRITE_SIZE", "scan_w%d" % mode)):
        p = find(d, "counter_collection.csv")
        tot = collections.defaultdict(float); n = collections.defaultdict(int)
        if p:
            for r in csv.DictReader(open(p)):
                if r["Counter_Name"] == ctr and r["Kernel_Name"].startswith(("sp_", "sm_", "gv_", "ge_")):
                    tot[r["Kernel_Name"][:40]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:40]] += 1
        e[ctr + "_KB_raw_per_kernel_avg"] = {k: tot[k] / n[k] for k in tot}
        e[ctr + "_dispatches"] = dict(n)
    res[label] = e
# evals in the probe: 1 first call + 1 warm-up + 5 timed (T=512 steps each)
name = "_scan_cfg4_profile.json" if PROBE == "gruB1" else "_scan_cfg4_b64_profile.json"
json.dump(res, open(os.path.join(O, TAG + name), "w"), indent=1)
print(json.dumps(res, indent=1)[:3500])
PY
