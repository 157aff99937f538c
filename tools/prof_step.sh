#!/bin/bash
# usage (on the GPU box): tools/prof_step.sh NAME  -- rocprofv3 kernel stats of the default bench step -> gpurun_out/NAME_kernel_stats.csv
export TMPDIR=/tmp
NAME=${1:-step}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$NAME -o $NAME -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/prof_$NAME.log 2>&1
f=$(find gpurun_out/prof_$NAME -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${NAME}_kernel_stats.csv
head -32 gpurun_out/${NAME}_kernel_stats.csv | cut -c1-150
tail -1 gpurun_out/prof_$NAME.log | cut -c1-200
