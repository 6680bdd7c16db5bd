# usage: tools/prof_kernels.sh <tag> <perf_probe args...>   -> prints avg duration of ew_* / gemm / gemv kernels and inter-kernel gaps
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/tools/perf_probe.py "$@" --out /dev/null > $out.log 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$out/p_kernel_trace.csv")))
rows = [r for r in rows if r["Kernel_Name"].startswith(("ew_", "gv_", "rp_", "rc_", "rcl_", "ge_", "void (anonymous namespace)::"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
gaps = []
for a, b in zip(rows, rows[1:]):
    gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for r in rows:
    d[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    v2 = sorted(v)
    print("%-62s n=%5d avg=%8.2f us med=%8.2f us min=%8.2f" % (k, len(v), sum(v)/len(v)/1e3, v2[len(v2)//2]/1e3, v2[0]/1e3))
g = sorted(gaps)
if g: print("gaps between consecutive kernels: med=%.2f us p10=%.2f p90=%.2f" % (g[len(g)//2]/1e3, g[len(g)//10]/1e3, g[9*len(g)//10]/1e3))
PY
