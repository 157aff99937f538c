#!/bin/bash
# usage (GPU box): bash tools/ab_prof.sh A.so B.so [rounds]   -- same-box A/B of two PRE-BUILT libraries (python -m spurfies_amd.build --variant NAME "flags",
# built in the authoring container: they travel with the snapshot) on the bench step: per-kernel rocprofv3 averages, alternating A B A B.
export TMPDIR=/tmp
A=$1; B=$2; N=${3:-2}
for i in $(seq 1 $N); do
  if [ $((i % 2)) -eq 1 ]; then ORDER="$A $B"; else ORDER="$B $A"; fi      # A B B A: clock / thermal drift cancels
  for V in $ORDER; do
    tag=$(basename $V .so)_$i
    SPF_LIB_PATH=$PWD/$V rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$tag -o $tag -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --sustained 0 > gpurun_out/ab_$tag.log 2>&1
    f=$(find gpurun_out/ab_$tag -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_kernel_stats.csv
    echo "== $tag"; python3 tools/prof_summary.py $tag | sed -n 2,9p; python3 tools/prof_summary.py $tag | tail -1
  done
done
