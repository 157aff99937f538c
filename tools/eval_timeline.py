"""Timeline of ONE evaluation chunk from a rocprofv3 kernel trace (gpurun_out/prof_<name>_chunk/*kernel_trace.csv): python tools/eval_timeline.py NAME"""
import csv, glob, sys
sys.path.insert(0, "tools")
from prof_summary import short
name = sys.argv[1]
f = glob.glob(f"gpurun_out/prof_{name}_chunk/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "camera_rays_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"]); prev = t0
out = []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:8.1f}  {short(r['Kernel_Name'])}")
    prev = e
out.append(f"span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {b - a} launches, busy {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[a:b]) / 1e3:.1f} us")
open(f"gpurun_out/{name}_eval_chunk_timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
