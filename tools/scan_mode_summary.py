"""Summary of an AESARA_HIP_SUITE_SCANLOG file (tests/hip_suite_plugin.py writes one JSON line per
evaluated function: [test id, sorted scan modes]): which Scans of the reference's own test files run
as one persistent launch, as one evaluation over whole sequences (no recurrence), or on the launch
list — with the reason per fallback.  usage: tools/scan_mode_summary.py LOG > profiles/rNN_scan_modes_reference_tests.txt"""
import collections
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
seen, per_file, why, tests = set(), collections.defaultdict(collections.Counter), collections.Counter(), set()
for test, modes in rows:
    key = (test, tuple(modes))
    if key in seen:          # repeated calls of one compiled function (verify_grad) count once
        continue
    seen.add(key)
    tests.add(test)
    for m in modes:
        kind = "persistent" if m == "persistent" else "all rows" if m == "all-rows" else "launch list"
        per_file[test.split("::")[0]][kind] += 1
        if kind == "launch list":
            why[m.split(": ", 1)[1] if ": " in m else m] += 1
tot = collections.Counter()
for c in per_file.values():
    tot.update(c)
print("# Scans of the reference's OWN test files under mode HIP: persistent one-launch kernel / one evaluation over")
print("# whole sequences (no recurrence: fusion.batch_map_step) / launch list")
print("# (tests/hip_suite_plugin.py, executor = prebuild: the PlanExecutor's dry run records PlanExecutor.scan_modes;")
print("#  AESARA_HIP_SUITE_SCANLOG=... python tests/reference_files.py --executor prebuild; tools/scan_mode_summary.py;")
print("#  one entry per DISTINCT compiled function of a test)")
print("%d tests of the file list run Scans; distinct Scan nodes evaluated: %d persistent, %d all rows at once, %d launch list"
      % (len(tests), tot["persistent"], tot["all rows"], tot["launch list"]))
print("per file:")
for f in sorted(per_file):
    c = per_file[f]
    print("  %-36s persistent %3d   all rows %3d   launch list %3d" % (f, c["persistent"], c["all rows"], c["launch list"]))
print("launch list, by reason (scan_persist*.py eligibility):")
for r, n in why.most_common():
    print("  %4d  %s" % (n, r))
