# usage: tools/prof_pmc.sh <tag> "<counters>" <perf_probe args...>  -> per-kernel average of each counter (eager launches)
tag=$1; shift; ctrs=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/tools/perf_probe.py "$@" --graph 0 --out /dev/null > $out.log 2>&1
python - <<PY
import csv, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$out/p_counter_collection.csv")):
    k = r["Kernel_Name"][:40]
    if not k.startswith(("ew_", "gv_", "rp_", "rc_", "void (anonymous")): continue
    d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    d[k]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    print(k, " ".join("%s=%.4g" % (c, sum(x)/len(x)) for c, x in sorted(v.items())), "n=%d" % len(v["dur_ns"]))
PY
