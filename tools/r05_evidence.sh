#!/bin/bash
# usage (GPU box, repo root): bash tools/r05_evidence.sh [part ...]   parts: bench proxy prof pmc power   (default: all)
# Re-collects every measurement DESIGN.md quotes for round 5 on the code as it is (the PMC / power files carry the csrc digest bench.py checks).
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
PARTS=${*:-bench proxy prof timeline pmc power}
for part in $PARTS; do case $part in
bench)
  # the driver's command, twice in a row (ms_per_step against the sustained figure), then the recipe with local_data, then every extra record
  python3 bench.py > gpurun_out/r05_bench_run1.json 2> gpurun_out/r05_bench_run1.err
  python3 bench.py > gpurun_out/r05_bench_run2.json 2> gpurun_out/r05_bench_run2.err
  python3 bench.py --local --no-cpu-baseline > gpurun_out/r05_bench_local.json 2> gpurun_out/r05_bench_local.err
  python3 bench.py --extras on --no-cpu-baseline --ab-reps 0 > gpurun_out/r05_bench_extras.json 2> gpurun_out/r05_bench_extras.err
  python3 - <<'PY'
import json
for n in ("run1", "run2", "local", "extras"):
    try:
        d = json.loads(open(f"gpurun_out/r05_bench_{n}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "FAILED", e); continue
    s = d.get("sustained") or {}
    print(n, "ms_per_step %.3f" % d["ms_per_step"], "sustained", s.get("ms_per_step"), "with_local", d.get("ms_per_step_with_local"),
          "roofline frac", (d.get("roofline") or {}).get("frac"), "extras", [(e.get("record"), round(e.get("ms_per_step", 0), 3)) for e in d.get("extra", [])])
PY
  ;;
proxy)
  timeout 900 python3 tools/strong_proxy.py --out gpurun_out/r05_strong_proxy.json --steps 60 > gpurun_out/r05_strong_proxy.log 2>&1; tail -1 gpurun_out/r05_strong_proxy.log | cut -c1-600
  timeout 900 python3 tools/strong_proxy.py --scenes 11 --out gpurun_out/r05_strong_proxy_configs3.json --steps 20 > gpurun_out/r05_strong_proxy_configs3.log 2>&1; tail -1 gpurun_out/r05_strong_proxy_configs3.log | cut -c1-600
  timeout 900 python3 tools/strong_proxy.py --dense --out gpurun_out/r05_strong_proxy_dense.json --steps 30 > gpurun_out/r05_strong_proxy_dense.log 2>&1; tail -1 gpurun_out/r05_strong_proxy_dense.log | cut -c1-600
  ;;
prof)
  bash tools/prof_step.sh r05_step > gpurun_out/r05_prof_step.log 2>&1; tail -2 gpurun_out/r05_prof_step.log | cut -c1-400
  ;;
timeline)
  BACK=30 bash tools/step_timeline.sh r05_1024rays --graph > /dev/null 2>&1; tail -2 gpurun_out/r05_1024rays_timeline.txt
  BACK=30 bash tools/step_timeline.sh r05_128rays --rays 128 > /dev/null 2>&1; tail -2 gpurun_out/r05_128rays_timeline.txt
  ;;
pmc)
  python3 tools/pmc_traffic.py r05 > gpurun_out/r05_pmc_traffic.log 2>&1; tail -2 gpurun_out/r05_pmc_traffic.log | cut -c1-300
  python3 tools/pmc_mfma.py r05 > gpurun_out/r05_pmc_mfma.log 2>&1; tail -2 gpurun_out/r05_pmc_mfma.log | cut -c1-300
  ;;
power)
  python3 tools/power_probe.py --out gpurun_out/r05_power.json > gpurun_out/r05_power.log 2>&1; grep -E "^(geo_split_w|color_fwd|color_bwd|wgrad_256|step) " gpurun_out/r05_power.log | cut -c1-400
  ;;
esac; done
