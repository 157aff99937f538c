"""Where does a run stall?  From a rocprofv3 --kernel-trace CSV: the longest inter-kernel gaps and the kernels that ran far longer than their median.
Usage: python tools/trace_outliers.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[:50]
gaps = []
prev_end = t0
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gaps.append((s - prev_end, i))
    prev_end = max(prev_end, e)
print("largest gaps (us) [time since start ms, before kernel]:")
for g, i in sorted(gaps, reverse=True)[:12]:
    print(f"  {g / 1e3:10.1f} us at {(int(rows[i]['Start_Timestamp']) - t0) / 1e6:9.1f} ms before {name(rows[i])} (after {name(rows[i - 1]) if i else '-'})")
dur = collections.defaultdict(list)
for i, r in enumerate(rows):
    dur[name(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), i))
print("kernels far above their median:")
out = []
for k, v in dur.items():
    med = sorted(d for d, _ in v)[len(v) // 2]
    for d, i in v:
        if d > 3 * med and d - med > 20000:
            out.append((d - med, d, med, k, i))
for ex, d, med, k, i in sorted(out, reverse=True)[:15]:
    print(f"  {k:50s} {d / 1e3:10.1f} us (median {med / 1e3:.1f}) at {(int(rows[i]['Start_Timestamp']) - t0) / 1e6:9.1f} ms")
print(f"total span {(prev_end - t0) / 1e6:.1f} ms, kernels {len(rows)}")
