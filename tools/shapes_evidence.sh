#!/bin/bash
# usage (GPU box): bash tools/shapes_evidence.sh TAG -- driver-shaped bench lines for the other BASELINE.json shapes + kernel stats of the dense one
#   configs[2] (garden-like, 5e4 points, +-2 grid), configs[4] (dense 2e5-point cloud, 4096-ray batches), configs[3] (11 scenes round-robin)
export TMPDIR=/tmp
TAG=${1:-r04}
python3 bench.py --points 50000 --no-cpu-baseline --extras off --sustained 50 --ab-reps 0 > gpurun_out/${TAG}_shapes_configs2.json 2> gpurun_out/${TAG}_shapes_configs2.err
python3 bench.py --points 200000 --spacing 0.0125 --rays 4096 --no-cpu-baseline --extras off --sustained 20 --ab-reps 0 --steps 10 > gpurun_out/${TAG}_shapes_configs4.json 2> gpurun_out/${TAG}_shapes_configs4.err
python3 bench.py --scenes 11 --no-cpu-baseline --sustained 0 --ab-reps 0 --steps 5 --warmup 2 > gpurun_out/${TAG}_shapes_configs3.json 2> gpurun_out/${TAG}_shapes_configs3.err
for f in configs2 configs4 configs3; do python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${TAG}_shapes_$f.json').read().strip().splitlines()[-1])
print('$f', 'ms/step %.3f' % d['ms_per_step'], 'value %.3e' % d['value'], d['config']['workload'][:90])
"; done
# the dense shape under rocprofv3: where the time goes when kNN + 77 MB of latents are stressed
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_c4 -o c4 -- python3 bench.py --points 200000 --spacing 0.0125 --rays 4096 --no-cpu-baseline --extras off --sustained 0 --ab-reps 0 --steps 8 --warmup 4 > gpurun_out/prof_${TAG}_c4.log 2>&1
cp $(find gpurun_out/prof_${TAG}_c4 -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_shapes_configs4_kernel_stats.csv
f=$(find gpurun_out/prof_${TAG}_c4 -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py "$f" 3 > gpurun_out/${TAG}_shapes_configs4_timeline.txt
tail -1 gpurun_out/${TAG}_shapes_configs4_timeline.txt
rm -rf gpurun_out/prof_${TAG}_c4
head -14 gpurun_out/${TAG}_shapes_configs4_kernel_stats.csv | cut -c1-140
