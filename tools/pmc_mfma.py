"""Matrix-pipe duty and shader clock of the step's MFMA kernels from hardware counters (own rocprofv3 --pmc pass, counter collection +
--kernel-trace only): SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs), GRBM_GUI_ACTIVE (GPU-active clocks, summed over the
8 XCDs) and the kernel durations of the trace.  duty = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024); clock = GUI_ACTIVE / 8 / duration.
Run on the GPU box from the repo root; writes gpurun_out/<round>_pmc_*.json (copy into profiles/); usage: python tools/pmc_mfma.py [round tag, default r03] [eval]."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
MODE = ["--mode", "eval", "--sweep-resolution", "0", "--image", "0", "0"] if "eval" in sys.argv[2:] else []
OUT = TAG + ("_eval" if MODE else "") + "_pmc_mfma.json"

d = "gpurun_out/pmc_mfma"
subprocess.run(["rocprofv3", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                "python3", "bench.py", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--ab-reps", "0", "--sustained", "0", "--settle", "0", "--extras", "off"] + MODE, check=True, stdout=subprocess.DEVNULL,
               stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"))
f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    per[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
    per[(r["Dispatch_Id"], r["Kernel_Name"])]["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
agg = collections.defaultdict(list)
for (_, name), v in per.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0 and v["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
        agg[name[:72]].append(v)
out = {}
for name, vs in agg.items():
    big = [v for v in vs if v["ns"] > 0.3 * max(x["ns"] for x in vs)]           # the large launches of the kernel (main pass)
    gui = sum(v["GRBM_GUI_ACTIVE"] for v in big) / len(big) / 8.0
    out[name] = {"launches": len(big), "avg_us": sum(v["ns"] for v in big) / len(big) / 1e3,
                 "mfma_duty": sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in big) / len(big) / (gui * 1024.0),
                 "clock_ghz": gui / (sum(v["ns"] for v in big) / len(big))}
sys.path.insert(0, ".")
import bench  # noqa: E402  (csrc_digest: identity of the kernel sources this collection was made on; bench.py refuses a stale one)

rec = {"csrc_sha256": bench.csrc_digest(), "note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace on bench.py --steps 4 --warmup 2; per kernel, mean over its large "
               "launches. mfma_duty = MFMA_BUSY / (GUI_ACTIVE / 8 XCDs x 1024 SIMDs); clock_ghz = GUI_ACTIVE / 8 / duration (durations under counter "
               "collection are longer than in a plain run).", "kernels": out}
json.dump(rec, open("gpurun_out/" + OUT, "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_us"])[:10]:
    print(f"{k[:64]:64s} {v['avg_us']:9.1f} us  duty {v['mfma_duty']:.3f}  clock {v['clock_ghz']:.2f} GHz")
