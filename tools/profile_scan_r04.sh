#!/bin/bash
# Round-4 counter evidence on the persistent matrix-state Scan kernel (BASELINE config 4, B = 64):
# kernel trace, FETCH_SIZE, WRITE_SIZE and SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE — each counter
# in its own pass (MI355X_MICROARCH.md §rocprofv3), eager launches so every kernel is its own record.
# Under the counter passes a persistent kernel runs several times slower: the spin limit is raised
# (AESARA_HIP_SPIN_LOG2) so that no poll gives up.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export AESARA_HIP_SPIN_LOG2=27
P="python $R/tools/perf_probe.py --only cfg4b64 --graph 0 --out /dev/null"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sm_kt -o s -- $P > $O/sm_kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVES"; do
  n=$(echo $c | tr " " "_")
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/sm_$n -o s -- $P > $O/sm_$n.log 2>&1
done
python - <<PY
import csv, glob, json, os, collections
O = "$O"
def find(d, suffix):
    c = glob.glob(os.path.join(O, d, "**", "*" + suffix), recursive=True)
    return c[0] if c else None
out = {"probe": "tools/perf_probe.py --only cfg4b64 --graph 0 (GRU T=512 H=1024 B=64 fp32, eager launches)",
       "spin_limit_log2": 27}
p = find("sm_kt", "kernel_stats.csv")
if p:
    out["kernel_stats"] = [{"name": r["Name"][:60], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                            "pct": float(r["Percentage"])} for r in csv.DictReader(open(p))][:6]
ctr = {}
for d in glob.glob(os.path.join(O, "sm_*")):
    if not os.path.isdir(d) or d.endswith("sm_kt"):
        continue
    p = find(os.path.basename(d), "counter_collection.csv")
    if not p:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        if r["Kernel_Name"].startswith(("sm_", "gemm")):
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for cn, v in cs.items():
            ctr.setdefault(k, {})[cn] = {"n": len(v), "avg": sum(v) / len(v)}
out["counters_per_dispatch_avg"] = ctr
for k, cs in ctr.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
        # MFMA busy cycles are summed over the SIMDs' matrix pipes (256 CUs x 4): fraction of the
        # kernel's active cycles during which a pipe was busy
        cs["mfma_busy_fraction_under_the_counter_pass"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] / (cs["GRBM_GUI_ACTIVE"]["avg"] * 256 * 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
        # the counter passes slow a persistent kernel several times (GRBM_GUI_ACTIVE above is THEIR
        # duration); the busy cycles are the same work, so price them against the duration of the
        # kernel-trace pass: cycles one SIMD's matrix pipe was busy / cycles the kernel took at the
        # 2.35 GHz the in-kernel clock calibration reads (profiles/r04_cfg4_b64_timeline_after.json)
        per_simd = cs["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] / (256 * 4)
        dur = next((r["avg_us"] for r in out.get("kernel_stats", []) if r["name"].startswith(k[:20])), None)
        cs["mfma_busy_cycles_per_simd"] = per_simd
        if dur:
            cs["kernel_trace_avg_us"] = dur
            cs["mfma_busy_fraction"] = per_simd / (dur * 2350.0)
json.dump(out, open(os.path.join(O, "r04_cfg4_b64_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
rm -rf $O/sm_kt $O/sm_FETCH_SIZE $O/sm_WRITE_SIZE $O/sm_SQ_*
