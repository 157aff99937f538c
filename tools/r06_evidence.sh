#!/bin/bash
# usage (GPU box, repo root): bash tools/r06_evidence.sh [part ...]   parts: bench local eval proxy prof timeline pmc power shapes soak   (default: all but soak)
# Re-collects every measurement DESIGN.md quotes for round 6 on the code as it is (the PMC / power files carry the csrc digest bench.py checks).
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
PARTS=${*:-bench local eval proxy prof timeline pmc power shapes}
for part in $PARTS; do case $part in
bench)
  # the driver's command, twice in a row: the contract line + its two N = 1 extra records (the recipe's step with local_data; one evaluation chunk)
  for n in 1 2; do
    t0=$(date +%s.%N); python3 bench.py > gpurun_out/r06_bench_run$n.json 2> gpurun_out/r06_bench_run$n.err; t1=$(date +%s.%N)
    python3 -c "print($t1 - $t0)" > gpurun_out/r06_bench_run$n.wall
  done
  python3 - <<'PY'
import json
for n in ("run1", "run2"):
    d = json.loads(open(f"gpurun_out/r06_bench_{n}.json").read().strip().splitlines()[-1])
    d["driver_run_s"] = float(open(f"gpurun_out/r06_bench_{n}.wall").read().strip())
    json.dump(d, open(f"gpurun_out/r06_bench_{n}.json", "w"))
    s = d.get("sustained") or {}
    print(n, "ms_per_step %.3f" % d["ms_per_step"], "sustained", s.get("ms_per_step"), "with_local", d.get("ms_per_step_with_local"), "wall s", d["driver_run_s"],
          "roofline frac", (d.get("roofline") or {}).get("frac"), "extras", [(e.get("record", "")[:24], round(e.get("ms_per_step", 0), 3)) for e in d.get("extra", [])])
PY
  ;;
local)
  bash tools/local_ab.sh > gpurun_out/r06_local_ab.log 2>&1
  python3 - <<'PY'
import json
out = {"what": "python bench.py --rays R [--graph] --local: the plain optimisation step and the DTU recipe's real step (feature-consistency term on every batch, "
               "32-channel maps, three views in turn) measured back to back in one process, same box", "rows": {}}
for tag in ("r1024", "r1024_graph", "r128", "r128_graph"):
    r = json.load(open(f"gpurun_out/local_ab/{tag}.json"))
    out["rows"][tag] = {"ms_per_step": r["ms_per_step"], "ms_per_step_with_local": r["ms_per_step_with_local"], "ratio": r["ms_per_step_with_local"] / r["ms_per_step"]}
json.dump(out, open("gpurun_out/r06_local_ab.json", "w"), indent=1)
print(json.dumps(out["rows"]))
PY
  ;;
eval)
  bash tools/eval_prof.sh r06 > gpurun_out/r06_eval_prof.log 2>&1
  python3 tools/eval_timeline.py r06 | tail -2
  python3 bench.py --mode eval --graph --no-cpu-baseline > gpurun_out/r06_eval_bench.json 2> gpurun_out/r06_eval_bench.err
  python3 bench.py --mode eval --graph --eval-outputs reference --no-cpu-baseline --image 0 0 --sweep-resolution 0 > gpurun_out/r06_eval_bench_reference_outputs.json 2> /dev/null
  python3 tools/lite_ab.py > gpurun_out/r06_lite_ab.log 2>&1; cp gpurun_out/lite_ab.json gpurun_out/r06_lite_sampler_ab.json
  python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_eval_bench.json").read().strip().splitlines()[-1])
r = json.loads(open("gpurun_out/r06_eval_bench_reference_outputs.json").read().strip().splitlines()[-1])
print("eval chunk ms (render outputs) %.3f" % d["ms_per_step"], "(reference outputs) %.3f" % r["ms_per_step"], "iters", d["config"]["sampler_iterations_realised"],
      "image s %.3f" % d["image_render"]["seconds"], "sweep s %.3f (+d2h %.3f)" % (d["sdf_eval_sweep"]["seconds"], d["sdf_eval_sweep"]["seconds_incl_d2h"]))
PY
  ;;
proxy)
  timeout 900 python3 tools/strong_proxy.py --out gpurun_out/r06_strong_proxy.json --steps 60 > gpurun_out/r06_strong_proxy.log 2>&1; tail -1 gpurun_out/r06_strong_proxy.log | cut -c1-600
  timeout 900 python3 tools/strong_proxy.py --scenes 11 --out gpurun_out/r06_strong_proxy_configs3.json --steps 20 > gpurun_out/r06_strong_proxy_configs3.log 2>&1; tail -1 gpurun_out/r06_strong_proxy_configs3.log | cut -c1-600
  timeout 900 python3 tools/strong_proxy.py --dense --out gpurun_out/r06_strong_proxy_dense.json --steps 30 > gpurun_out/r06_strong_proxy_dense.log 2>&1; tail -1 gpurun_out/r06_strong_proxy_dense.log | cut -c1-600
  ;;
prof)
  bash tools/prof_step.sh r06_step --extras off > gpurun_out/r06_prof_step.log 2>&1; tail -2 gpurun_out/r06_prof_step.log | cut -c1-400
  bash tools/prof_step.sh r06_step_local --extras off --local > gpurun_out/r06_prof_step_local.log 2>&1; tail -1 gpurun_out/r06_prof_step_local.log | cut -c1-300
  ;;
timeline)
  BACK=30 bash tools/step_timeline.sh r06_1024rays --graph --extras off > /dev/null 2>&1; tail -2 gpurun_out/r06_1024rays_timeline.txt
  BACK=30 bash tools/step_timeline.sh r06_128rays --rays 128 --extras off > /dev/null 2>&1; tail -2 gpurun_out/r06_128rays_timeline.txt
  ;;
pmc)
  python3 tools/pmc_traffic.py r06 > gpurun_out/r06_pmc_traffic.log 2>&1; tail -2 gpurun_out/r06_pmc_traffic.log | cut -c1-300
  python3 tools/pmc_mfma.py r06 > gpurun_out/r06_pmc_mfma.log 2>&1; tail -2 gpurun_out/r06_pmc_mfma.log | cut -c1-300
  ;;
power)
  python3 tools/power_probe.py --out gpurun_out/r06_power.json > gpurun_out/r06_power.log 2>&1; grep -E "^(h2_loop_[0-2]|geo_h2|geo_split_w|color_fwd|color_bwd|wgrad_256|step) " gpurun_out/r06_power.log | cut -c1-400
  ;;
shapes)
  bash tools/shapes_evidence.sh r06 2>&1 | tail -8
  ;;
soak)
  python3 tools/soak.py --local --steps 20000 --out gpurun_out/r06_soak_local.json > gpurun_out/r06_soak_local.log 2>&1; tail -3 gpurun_out/r06_soak_local.log | cut -c1-400
  ;;
esac; done
