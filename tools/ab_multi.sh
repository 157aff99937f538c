#!/bin/bash
# usage (GPU box): bash tools/ab_multi.sh "A.so B.so C.so ..." [rounds]  -- same-box comparison of several PRE-BUILT libraries on the bench step
# (per-kernel rocprofv3 averages); the order is reversed every other round so that clock / thermal drift cancels.
export TMPDIR=/tmp
LIBS=$1; N=${2:-2}
for i in $(seq 1 $N); do
  if [ $((i % 2)) -eq 1 ]; then ORDER="$LIBS"; else ORDER=$(echo $LIBS | tr ' ' '\n' | tac | tr '\n' ' '); fi
  for V in $ORDER; do
    tag=$(basename $V .so)_$i
    SPF_LIB_PATH=$PWD/$V rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$tag -o $tag -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --sustained 0 > gpurun_out/ab_$tag.log 2>&1
    f=$(find gpurun_out/ab_$tag -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_kernel_stats.csv
    echo "== $tag"; python3 tools/prof_summary.py $tag | sed -n 2,12p; python3 tools/prof_summary.py $tag | tail -1
  done
done
