#!/bin/bash
# usage (on the GPU box): tools/prof_step.sh NAME [extra bench.py flags, e.g. --mode eval --sweep-resolution 0]
#   -- rocprofv3 kernel stats of the default bench step -> gpurun_out/NAME_kernel_stats.csv (+ NAME_bench.json, NAME_geo_main_pass_launches.csv)
export TMPDIR=/tmp
NAME=${1:-step}
shift
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$NAME -o $NAME -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --sustained 0 --ab-reps 0 "$@" > gpurun_out/prof_$NAME.log 2>&1
f=$(find gpurun_out/prof_$NAME -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${NAME}_kernel_stats.csv
head -32 gpurun_out/${NAME}_kernel_stats.csv | cut -c1-150
tail -1 gpurun_out/prof_$NAME.log | cut -c1-200
# per-launch durations of the dominant kernel's MAIN-PASS launches (the stats file above averages them with the small pseudo-point launches)
python3 - "$NAME" <<'PY'
import csv, glob, sys
import json
name = sys.argv[1]
bench = json.loads([l for l in open(f"gpurun_out/prof_{name}.log") if l.startswith("{")][-1])
roof = bench.get("roofline") or {}
f = glob.glob(f"gpurun_out/prof_{name}/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "geo_pairs_x3_kernel<true>" in r["Kernel_Name"] or "geo_pairs_x3w_kernel<true" in r["Kernel_Name"]]
json.dump(bench, open(f"gpurun_out/{name}_bench.json", "w"))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
big = [x for x in d if x > 0.3 * max(d)]
col = "; ".join("%s: pairs_per_launch %s, avg_ms %s" % (s_["kernel"], s_.get("pairs_per_launch"), s_["avg_ms"]) for s_ in (roof.get("secondary") or []) if "color" in s_["kernel"])
head = ("# " + (roof.get("kernel") or "geo_pairs kernel") + ": %d launches in the trace, %d of them main-pass launches (the others are the pseudo-point pass); main-pass mean %.1f us, "
        "min %.1f, max %.1f; the bench line of the same run (HIP events over its timed region): pairs_per_launch %s, flop_per_pair %s, avg_ms %s, achieved %s TFLOP/s, "
        "frac %s of %s; colour trunk: %s" % (len(d), len(big), sum(big) / len(big) / 1e3, min(big) / 1e3, max(big) / 1e3, roof.get("pairs_per_launch"),
                                               roof.get("flop_per_pair"), roof.get("avg_ms"), roof.get("achieved"), roof.get("frac"), roof.get("peak"), col))
with open(f"gpurun_out/{name}_geo_main_pass_launches.csv", "w") as o:
    o.write('"' + head + '"\n')
    o.write("launch,DurationNs\n")
    for i, x in enumerate(big):
        o.write(f"{i},{x}\n")
print(open(f"gpurun_out/{name}_geo_main_pass_launches.csv").readline())
PY
