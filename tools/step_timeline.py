"""One optimisation step as the GPU saw it: kernel sequence with durations and the idle gap in front of each launch, from a rocprofv3
--kernel-trace CSV of `bench.py --steps N` (tools/step_timeline.sh).  Usage: python tools/step_timeline.py <kernel_trace.csv> [step index from the end]"""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def short(name):
    m = re.search(r"(?:\(anonymous namespace\)::|at::native::)?([A-Za-z_0-9]+)(<[^(>]*>)?\(", name)
    if not m:
        return name[:60]
    base, targ = m.group(1), m.group(2) or ""
    if base in ("vectorized_elementwise_kernel", "elementwise_kernel_manual_unroll", "unrolled_elementwise_kernel", "reduce_kernel", "elementwise_kernel"):
        f = re.search(r"at::native::(\w+Functor|\w+_kernel_cuda|\w+Ops|\w+_kernel_impl)", name)
        targ = "<" + (f.group(1) if f else "") + ">"
    return (base + targ)[:60]


# a step starts at the ray set-up launch (camera_rays_kernel, or camera_uniform_kernel since round 5)
starts = [i for i, r in enumerate(rows) if "camera_rays_kernel" in r["Kernel_Name"] or "camera_uniform_kernel" in r["Kernel_Name"]]
# the step shown is the one with the MEDIAN period among the (up to) nine steps around index `back` from the end: a single step can be hit by an
# outlier (one colour backward launch read 1.79 ms instead of 0.71 in a round-5 collection) and would misrepresent the sequence
cand = [k for k in range(back - 4, back + 5) if k >= 2 and k + 1 <= len(starts) - 1]
periods = sorted((int(rows[starts[-k]]["Start_Timestamp"]) - int(rows[starts[-k - 1]]["Start_Timestamp"]), k) for k in cand)
if periods:
    back = periods[len(periods) // 2][1]
    print(f"# step {back} from the end of the trace: median period of {len(periods)} neighbouring steps ({periods[0][0] / 1e3:.1f} .. {periods[-1][0] / 1e3:.1f} us)")
a, b = starts[-back - 1], starts[-back]
seq = rows[a:b]
t0 = int(seq[0]["Start_Timestamp"])
prev_end = t0
busy = gap_total = 0
small = 0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - prev_end)
    busy += e - s
    gap_total += gap
    small += (e - s) if (e - s) < 15000 else 0
    q = r.get("Queue_Id", r.get("Stream_Id", ""))
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  q{q}  {short(r['Kernel_Name'])}" + ("   (starts before the previous kernel ended)" if s < prev_end else ""))
    prev_end = max(prev_end, e)
if back >= 2:
    nxt = int(rows[starts[-back + 1]]["Start_Timestamp"]) if -back + 1 < 0 else None
    if nxt:
        print(f"period (start of this step -> start of the next): {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
print(f"launches {len(seq)}, busy {busy / 1e3:.1f} us, gaps {gap_total / 1e3:.1f} us, span {(prev_end - t0) / 1e3:.1f} us, kernels < 15 us: "
      f"{sum(1 for r in seq if int(r['End_Timestamp']) - int(r['Start_Timestamp']) < 15000)} launches = {small / 1e3:.1f} us")
