"""HBM traffic per launch of every kernel of the step, the way MI355X_MICROARCH.md prescribes: separate rocprofv3 --pmc passes
for FETCH_SIZE and WRITE_SIZE (counter collection only, plus --kernel-trace), gfx950 correction hbm = (2*FETCH + WRITE) KiB.
Run on the GPU box from the repo root; writes gpurun_out/<round>_pmc_*.json (copy into profiles/); usage: python tools/pmc_traffic.py [round tag, default r03] [eval]."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
MODE = ["--mode", "eval", "--sweep-resolution", "0", "--image", "0", "0"] if "eval" in sys.argv[2:] else []
DENSE = "dense" in sys.argv[2:]          # BASELINE.json configs[4]: 2e5-point cloud (spacing 0.0125), 4096-ray batches
OUT = TAG + ("_eval" if MODE else "") + ("_dense" if DENSE else "") + "_pmc_traffic.json"

POINTS, RAYS = (200000, 4096) if DENSE else (10000, 1024)
MODE += ["--spacing", "0.0125"] if DENSE else []
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = f"gpurun_out/pmc_{ctr}"
    subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", "python3", "bench.py",
                    "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--points", str(POINTS), "--rays", str(RAYS), "--ab-reps", "0", "--sustained", "0", "--settle", "0", "--extras", "off"] + MODE,
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"))
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            acc[r["Kernel_Name"][:72]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})["launches"] = len(v)
        out[k][f"{ctr}_KB_max"] = max(v)
for k, v in out.items():
    v["hbm_bytes_max_corrected"] = (2.0 * v.get("FETCH_SIZE_KB_max", 0.0) + v.get("WRITE_SIZE_KB_max", 0.0)) * 1024.0
sys.path.insert(0, ".")
import bench  # noqa: E402  (csrc_digest: identity of the kernel sources this collection was made on; bench.py refuses a stale one)

rec = {"csrc_sha256": bench.csrc_digest(), "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on bench.py --steps 3 --warmup 1; per-launch values of the "
               "largest launch per kernel (main pass). hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reads half of a wide "
               "coalesced stream (MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are included in both counters.",
       "config": {"points": POINTS, "rays": RAYS, "prior": "fitted", "spacing": 0.0125 if DENSE else 0.025}, "kernels": out}
json.dump(rec, open("gpurun_out/" + OUT, "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_max_corrected"])[:12]:
    print(f"{k[:60]:60s} {v['hbm_bytes_max_corrected'] / 1e6:10.1f} MB")
