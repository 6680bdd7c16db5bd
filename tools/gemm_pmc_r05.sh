#!/bin/bash
# Round 5: the counter set VERDICT r4 asks for on the CURRENT 4096^3 fp32 GEMM kernel (bench.py's
# cfg3b row), each group in its own rocprofv3 pass (kernel trace + pmc only) -> r05_gemm_pmc.json
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gemm_pmc_r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --only-secondary cfg3b"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -o p -- $B > $O/g$i.out 2> $O/g$i.err
done
python - "$O" <<'PY'
import csv, glob, json, os, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(os.path.join(O, "g*", "**", "*counter_collection.csv"), recursive=True):
    grp = p.split(os.sep)[-4] if "g" in p else "g"
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "gemm_kernel" not in k:
            continue
        acc[k[:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k[:100]]["dur_ns:" + r["Counter_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, v in acc.items():
    row = {c: sum(x) / len(x) for c, x in v.items()}
    row["n"] = max(len(x) for x in v.values())
    cyc = row.get("GRBM_GUI_ACTIVE")
    if cyc and row.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (check: 4096^3 fp32 = 2^26
        # v_mfma_f32_16x16x4_f32 x 32 busy cycles = 2^31 exactly); GRBM_GUI_ACTIVE over the 8 XCDs
        row["kernel_cycles"] = cyc / 8.0
        row["mfma_busy_frac"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc / 8.0 * 1024.0)
    if row.get("SQ_WAVE_CYCLES"):
        row["wave_wait_frac"] = row.get("SQ_WAIT_ANY", 0) / row["SQ_WAVE_CYCLES"]
        row["wave_wait_lds_frac"] = row.get("SQ_WAIT_INST_LDS", 0) / row["SQ_WAVE_CYCLES"]
    if row.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_bank_conflict_frac"] = row.get("SQ_LDS_BANK_CONFLICT", 0) / row["SQ_LDS_IDX_ACTIVE"]
    if row.get("FETCH_SIZE") is not None:
        row["hbm_read_bytes_corrected"] = row["FETCH_SIZE"] * 2 * 1024
    if row.get("WRITE_SIZE") is not None:
        row["hbm_write_bytes"] = row["WRITE_SIZE"] * 1024
    out[k] = row
json.dump(out, open(os.path.join(os.path.dirname(O), "r05_gemm_pmc.json"), "w"), indent=1)
for k, v in out.items():
    print(k[:70], {c: (round(x, 4) if isinstance(x, float) else x) for c, x in v.items() if "frac" in c or c in ("n",)})
PY
rm -rf $O/g*/
