#!/bin/bash
# usage (GPU box): [BACK=30] tools/step_timeline.sh NAME [extra bench.py flags, e.g. --rays 128]   (BACK = which step from the end of the trace; graph-replayed
# runs end with ~20 EAGER profiling passes, so use BACK=30 to land on a replayed step)
# -> gpurun_out/NAME_timeline.txt (one step's kernel sequence), NAME_kernel_stats.csv
export TMPDIR=/tmp
NAME=${1:-tl}
shift
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$NAME -o $NAME -- python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --sustained 0 --ab-reps 0 "$@" > gpurun_out/prof_$NAME.log 2>&1
f=$(find gpurun_out/prof_$NAME -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py "$f" ${BACK:-8} > gpurun_out/${NAME}_timeline.txt
cp $(find gpurun_out/prof_$NAME -name "*kernel_stats.csv" | head -1) gpurun_out/${NAME}_kernel_stats.csv
tail -3 gpurun_out/${NAME}_timeline.txt
rm -rf gpurun_out/prof_$NAME
