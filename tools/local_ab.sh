#!/bin/bash
# The DTU recipe's real step (feature-consistency term on every batch) beside the plain step, eager and graph, 1024 and 128 rays: one box.
# -> gpurun_out/local_ab/*.json  (copy what is to be judged into profiles/)
mkdir -p gpurun_out/local_ab
for rays in 1024 128; do
  for g in "" "--graph"; do
    tag="r${rays}$( [ -n "$g" ] && echo _graph )"
    python bench.py --rays $rays $g --local --no-cpu-baseline --sustained 0 --ab-reps 0 > gpurun_out/local_ab/$tag.json 2> gpurun_out/local_ab/$tag.err
    python - <<PY
import json
r = json.load(open("gpurun_out/local_ab/$tag.json"))
print("$tag", "ms_per_step", r["ms_per_step"], "with_local", r.get("ms_per_step_with_local"), "ratio", round(r.get("ms_per_step_with_local", 0) / r["ms_per_step"], 4))
PY
  done
done
