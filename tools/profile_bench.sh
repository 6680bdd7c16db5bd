#!/bin/bash
# Runs on the GPU box: rocprofv3 passes over bench.py (kernel trace + stats, then separate PMC
# passes for FETCH_SIZE / WRITE_SIZE as MI355X_MICROARCH.md prescribes) and writes compact
# summaries into gpurun_out/profiles_r01/ (copied to profiles/ and committed afterwards).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_r01; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python $R/bench.py --steps 2000 --warmup 50 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/write.err
python - <<PY
import csv, json, collections
O="$O"
rows=list(csv.DictReader(open(O+"/kt/bench_kernel_stats.csv")))
with open(O+"/r01_bench_kernel_stats.csv","w") as f:
    w=csv.writer(f); w.writerow(["Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"])
    for r in rows: w.writerow([r["Name"][:90],r["Calls"],r["TotalDurationNs"],r["AverageNs"],r["Percentage"],r["MinNs"],r["MaxNs"],r["StdDev"]])
def avg(path, ctr):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"]==ctr and r["Kernel_Name"].startswith("ew_"): d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k:(sum(v)/len(v),len(v)) for k,v in d.items()}
fe=avg(O+"/fetch/bench_counter_collection.csv","FETCH_SIZE"); wr=avg(O+"/write/bench_counter_collection.csv","WRITE_SIZE")
k=max(fe,key=lambda k:fe[k][0])
fetch_kb, n = fe[k]; write_kb = wr.get(k,(0,0))[0]
res={"kernel":k,"dispatches_sampled":n,"FETCH_SIZE_KB_raw":fetch_kb,"WRITE_SIZE_KB_raw":write_kb,
     "correction":"gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md HBM) -> x2",
     "hbm_bytes_per_launch": int(fetch_kb*2*1024+write_kb*1024), "algorithmic_bytes":134217728}
json.dump(res,open(O+"/r01_bench_traffic.json","w"),indent=1)
print(json.dumps(res)); print(open(O+"/r01_bench_kernel_stats.csv").read()[:400]); print(open(O+"/bench_under_rocprof.json").read()[:300])
PY
