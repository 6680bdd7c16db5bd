#!/bin/bash
# Round-6 rocprofv3 evidence for every driver-line row.  Runs on the GPU box; per
# row THREE separate passes of the same bench command — kernel trace + stats, --pmc FETCH_SIZE,
# --pmc WRITE_SIZE (counters in their own runs: MI355X_MICROARCH.md §HBM) — and one JSON summary per
# row under gpurun_out/profiles_r06/ (copied to profiles/ and committed afterwards).
#   usage: tools/profile_bench_r03.sh [rows...]      rows: cfg2 cfg3a cfg1b cfg5 cfg3b cfg4
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline"
ROWS=${@:-cfg2 cfg3a cfg1b cfg5 cfg3b gemmshapes cfg4}
prof() {   # name, bench args, kernel regex, algorithmic bytes per launch (0 = n/a), [nopmc]
  local name=$1 args=$2 rx=$3 algo=$4 nopmc=${5:-}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_kt -o p -- $B $args > $O/${name}_bench.json 2> $O/${name}_kt.err
  if [ -z "$nopmc" ]; then
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${name}_fetch -o p -- $B $args > /dev/null 2> $O/${name}_fetch.err
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${name}_write -o p -- $B $args > /dev/null 2> $O/${name}_write.err
  fi
  python - "$O" "$name" "$rx" "$algo" <<'PY'
import csv, glob, json, os, re, sys, collections
O, name, rx, algo = sys.argv[1], sys.argv[2], re.compile(sys.argv[3]), int(sys.argv[4])
def find(d, suffix):
    c = glob.glob(os.path.join(O, d, "**", "*" + suffix), recursive=True)
    return c[0] if c else None
out = {"row": name, "kernel_regex": rx.pattern, "algorithmic_bytes_per_launch": algo or None}
p = find(name + "_kt", "kernel_stats.csv")
if p:
    rows = list(csv.DictReader(open(p)))
    with open(os.path.join(O, "r06_%s_kernel_stats.csv" % name), "w") as f:
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows: w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    ks = [r for r in rows if rx.search(r["Name"])]
    out["kernels"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                       "min_us": float(r["MinNs"]) / 1e3, "pct_of_gpu_time": float(r["Percentage"])} for r in ks[:6]]
def counter(d, cname):
    p = find(d, "counter_collection.csv")
    acc = collections.defaultdict(list)
    if p:
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == cname and rx.search(r["Kernel_Name"]):
                acc[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
    return {k: {"n": len(v), "avg": sum(v) / len(v)} for k, v in acc.items()}
fe, wr = counter(name + "_fetch", "FETCH_SIZE"), counter(name + "_write", "WRITE_SIZE")
out["FETCH_SIZE_KB_raw"], out["WRITE_SIZE_KB_raw"] = fe, wr
per = {}
for k in out.get("kernels", []):
    f = fe.get(k["name"], {}).get("avg"); w = wr.get(k["name"], {}).get("avg")
    per[k["name"]] = dict(k, FETCH_SIZE_KB_raw=f, WRITE_SIZE_KB_raw=w,
                          hbm_bytes_per_launch_corrected=None if f is None else int(f * 2 * 1024 + (w or 0) * 1024))
out["per_kernel"] = per
# the row's own kernel: the matching kernel with the largest share of GPU time that is not the
# headline kernel every bench process also runs (cfg2's own row excepted)
cands = sorted(out.get("kernels", []), key=lambda k: -k["pct_of_gpu_time"])
if algo:   # (other rows also run the headline kernel: take the match whose counter bytes fit this row)
    fit = [k for k in cands if per.get(k["name"], {}).get("hbm_bytes_per_launch_corrected")]
    if fit:
        # (several kernels may move this row's bytes — the headline row also runs the read-only
        # ceiling kernel: among the matches within 2 % of the algorithmic bytes the most-called one)
        close = [k for k in fit if abs(per[k["name"]]["hbm_bytes_per_launch_corrected"] / algo - 1.0) < 0.02]
        cands = sorted(close, key=lambda k: -k["calls"]) if close else \
            sorted(fit, key=lambda k: abs(per[k["name"]]["hbm_bytes_per_launch_corrected"] / algo - 1.0))
out["row_kernel"] = cands[0]["name"] if cands else None
tot = per.get(out["row_kernel"], {}).get("hbm_bytes_per_launch_corrected") if cands else None
out["hbm_bytes_per_launch_corrected"] = tot
out["correction"] = "FETCH_SIZE x2 (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM) + WRITE_SIZE, of row_kernel"
if algo and tot: out["traffic_over_algorithmic"] = tot / algo
try:
    line = json.loads([l for l in open(os.path.join(O, name + "_bench.json")) if l.startswith("{")][-1])
    out["bench_line_under_profiler"] = {"roofline": line.get("roofline"),
        "secondary": [{k: s.get(k) for k in ("config", "roofline")} for s in line.get("secondary", [])]}
except Exception as e:
    out["bench_line_error"] = str(e)
json.dump(out, open(os.path.join(O, "r06_%s_profile.json" % name), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("row", "kernels", "hbm_bytes_per_launch_corrected", "traffic_over_algorithmic") if k in out})[:1200])
PY
  rm -rf $O/${name}_kt $O/${name}_fetch $O/${name}_write
}
for r in $ROWS; do case $r in
  cfg2)  prof cfg2_rotated "--no-secondary --no-warm --steps 1000 --warmup 50" "^ew_" 134217728 ;;
  cfg3a) prof cfg3a_gemv "--steps 20 --warmup 5 --only-secondary cfg3a" "gemv|gv_|ge_" 134283264 ;;
  cfg1b) prof cfg1b_add "--steps 20 --warmup 5 --only-secondary cfg1b" "^ew_" 402653184 ;;
  # config 5 under the profiler at N = 2^22 (4 GiB of X; the ratio counter bytes : algorithmic bytes
  # is what shows "X is read once", and it does not depend on N); full size: cfg5full
  cfg5)  prof cfg5_rowpass "--steps 20 --warmup 5 --only-secondary cfg5 --cfg5-log2n 22" "^rp_" 4311744512 ;;
  cfg5full) prof cfg5_rowpass_full "--steps 20 --warmup 5 --only-secondary cfg5" "^rp_" 17246978048 ;;
  cfg3b) prof cfg3b_gemm "--steps 20 --warmup 5 --only-secondary cfg3b" "gemm" 0 nopmc ;;
  gemmshapes) prof gemmshapes "--steps 20 --warmup 5 --only-secondary gemmshapes" "gemm" 0 nopmc ;;
  # persistent (spinning) kernels: kernel trace only, no counter passes
  cfg4)  prof cfg4_scan "--steps 20 --warmup 5 --only-secondary cfg4" "^s[mp]_|gemm" 0 nopmc ;;
esac; done
ls $O
