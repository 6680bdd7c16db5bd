#!/bin/bash
# Runs on the GPU box: rocprofv3 passes over bench.py and compact summaries into
# gpurun_out/profiles_$TAG/ (copied to profiles/ and committed afterwards).
#   1. kernel trace + stats of the headline (config 2) command
#   2. separate PMC passes FETCH_SIZE / WRITE_SIZE for its dominant kernel (MI355X_MICROARCH.md §HBM)
#   3. kernel trace + stats of the Gemm 4096^3 secondary workload
#   4. PMC passes for the GEMM kernel: MFMA busy, LDS bank conflicts, L2 hit rate
# usage: tools/profile_bench.sh [TAG]      (TAG default r02)
set -u
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- $B --steps 2000 --warmup 50 --no-secondary > $O/bench_under_rocprof.json 2> $O/kt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- $B --steps 200 --warmup 10 --no-secondary > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- $B --steps 200 --warmup 10 --no-secondary > /dev/null 2> $O/write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gkt -o gemm -- $B --steps 20 --warmup 5 --only-secondary cfg3b > $O/gemm_under_rocprof.json 2> $O/gkt.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/gp1 -o gemm -- $B --steps 20 --warmup 5 --only-secondary cfg3b > /dev/null 2> $O/gp1.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS --output-format csv -d $O/gp2 -o gemm -- $B --steps 20 --warmup 5 --only-secondary cfg3b > /dev/null 2> $O/gp2.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/gp3 -o gemm -- $B --steps 20 --warmup 5 --only-secondary cfg3b > /dev/null 2> $O/gp3.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/gp4 -o gemm -- $B --steps 20 --warmup 5 --only-secondary cfg3b > /dev/null 2> $O/gp4.err
python - <<PY
import csv, json, collections, glob, os
O="$O"; TAG="$TAG"
def find(d, suffix):
    c = glob.glob(os.path.join(O, d, "**", "*" + suffix), recursive=True)
    return c[0] if c else None
def stats(d, out):
    p = find(d, "kernel_stats.csv")
    if not p: return
    rows=list(csv.DictReader(open(p)))
    with open(os.path.join(O, out),"w") as f:
        w=csv.writer(f); w.writerow(["Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"])
        for r in rows: w.writerow([r["Name"][:110],r["Calls"],r["TotalDurationNs"],r["AverageNs"],r["Percentage"],r["MinNs"],r["MaxNs"],r["StdDev"]])
    print(open(os.path.join(O,out)).read()[:700])
stats("kt", TAG+"_bench_cfg2_kernel_stats.csv")
stats("gkt", TAG+"_gemm_kernel_stats.csv")
def avg(d, pred):
    p = find(d, "counter_collection.csv")
    out=collections.defaultdict(lambda: collections.defaultdict(list))
    if not p: return out
    for r in csv.DictReader(open(p)):
        if pred(r["Kernel_Name"]):
            out[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            out[r["Kernel_Name"][:100]]["_dur_ns"].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    return out
try:
    fe=avg("fetch", lambda k:k.startswith("ew_")); wr=avg("write", lambda k:k.startswith("ew_"))
    k=max(fe,key=lambda k:sum(fe[k]["FETCH_SIZE"])/len(fe[k]["FETCH_SIZE"]))
    fetch_kb=sum(fe[k]["FETCH_SIZE"])/len(fe[k]["FETCH_SIZE"]); n=len(fe[k]["FETCH_SIZE"])
    write_kb=sum(wr[k]["WRITE_SIZE"])/len(wr[k]["WRITE_SIZE"]) if k in wr else 0.0
    res={"kernel":k,"dispatches_sampled":n,"FETCH_SIZE_KB_raw":fetch_kb,"WRITE_SIZE_KB_raw":write_kb,
         "correction":"gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md HBM) -> x2",
         "hbm_bytes_per_launch": int(fetch_kb*2*1024+write_kb*1024), "algorithmic_bytes":134217728}
    json.dump(res,open(O+"/"+TAG+"_bench_traffic.json","w"),indent=1); print(json.dumps(res))
except Exception as e: print("traffic summary failed:", e)
g={}
for d in ("gp1","gp2","gp3","gp4"):
    for k,v in avg(d, lambda k:"gemm" in k and "big" in k or "gemm_kernel" in k or "gemm_" in k).items():
        e=g.setdefault(k,{})
        for c,x in v.items(): e[c if c!="_dur_ns" else "dur_ns_"+d]=sum(x)/len(x); e["n_"+d]=len(x)
for k,e in g.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (check: 4096^3 fp32 = 2^26
        # v_mfma_f32_16x16x4_f32 x 32 busy cycles = 2^31 exactly); GRBM_GUI_ACTIVE over the 8 XCDs
        e["kernel_cycles"]=e["GRBM_GUI_ACTIVE"]/8.0
        e["mfma_busy_frac"]=e["SQ_VALU_MFMA_BUSY_CYCLES"]/(e["GRBM_GUI_ACTIVE"]/8.0*1024.0)
        if e.get("SQ_WAVE_CYCLES"): e["wave_wait_frac"]=e["SQ_WAIT_ANY"]/e["SQ_WAVE_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"]=e["SQ_LDS_BANK_CONFLICT"]/e["SQ_LDS_IDX_ACTIVE"]
    if "TCC_HIT_sum" in e:
        e["l2_hit_rate"]=e["TCC_HIT_sum"]/(e["TCC_HIT_sum"]+e["TCC_MISS_sum"])
json.dump(g,open(O+"/"+TAG+"_gemm_pmc_summary.json","w"),indent=1)
print(json.dumps(g,indent=1)[:3000])
print(open(O+"/bench_under_rocprof.json").read()[:300])
PY
