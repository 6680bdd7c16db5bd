#!/usr/bin/env python
"""Average duration of a kernel and the gap to the next dispatch of the same kernel, from a
rocprofv3 --kernel-trace CSV (what the HIP-event time per eval splits into: in-kernel time +
kernel boundary).  usage: kernel_gaps.py <dir with *kernel_trace.csv> <kernel regex>"""
import csv
import glob
import json
import os
import re
import sys


def main():
    d, rx = sys.argv[1], re.compile(sys.argv[2])
    p = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(p))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    dur, gap = [], []
    for i, (s, e, n) in enumerate(ks):
        if not rx.search(n):
            continue
        dur.append(e - s)
        if i + 1 < len(ks) and rx.search(ks[i + 1][2]):
            g = ks[i + 1][0] - e
            if g < 20000:                       # back-to-back only (not across host pauses)
                gap.append(g)
    import statistics as st
    out = {"kernel_regex": rx.pattern, "dispatches": len(dur),
           "avg_duration_us": sum(dur) / len(dur) / 1e3, "median_duration_us": st.median(dur) / 1e3,
           "min_duration_us": min(dur) / 1e3,
           "back_to_back_pairs": len(gap), "avg_gap_us": sum(gap) / max(len(gap), 1) / 1e3,
           "median_gap_us": (st.median(gap) / 1e3) if gap else None,
           "avg_period_us": (sum(dur) / len(dur) + sum(gap) / max(len(gap), 1)) / 1e3}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
