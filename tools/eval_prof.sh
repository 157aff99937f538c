#!/bin/bash
# usage (GPU box): tools/eval_prof.sh NAME  -- rocprofv3 kernel stats of (a) evaluation-render chunks (eager + graph), (b) the get_sdf_eval sweep
export TMPDIR=/tmp
NAME=${1:-eval}
for leg in chunk sweep; do
  if [ $leg = chunk ]; then flags="--mode eval --steps 20 --warmup 5 --image 0 0 --sweep-resolution 0"; else flags="--mode eval --steps 2 --warmup 1 --image 0 0 --sweep-resolution 512"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${NAME}_$leg -o ${NAME}_$leg -- python3 bench.py $flags --no-cpu-baseline > gpurun_out/prof_${NAME}_$leg.log 2>&1
  f=$(find gpurun_out/prof_${NAME}_$leg -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/${NAME}_${leg}_kernel_stats.csv
  echo "== $leg"; head -40 gpurun_out/${NAME}_${leg}_kernel_stats.csv | cut -c1-160
  tail -1 gpurun_out/prof_${NAME}_$leg.log | cut -c1-300
done
