cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_wgrad.py tests/test_gpu_model.py -m gpu -x -q > gpurun_out/r5g_pytest.log 2>&1; tail -3 gpurun_out/r5g_pytest.log
NAME=r5g_128_nofork
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$NAME -o $NAME -- python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --sustained 0 --ab-reps 0 --geo-engine split_w --rays 128 --settle 0 --extras off --no-fork > gpurun_out/prof_$NAME.log 2>&1
f=$(find gpurun_out/prof_$NAME -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py "$f" 30 > gpurun_out/${NAME}_timeline.txt
rm -rf gpurun_out/prof_$NAME
grep -n "wgrad\|launches" gpurun_out/${NAME}_timeline.txt
NAME=r5g_1024
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$NAME -o $NAME -- python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --sustained 0 --ab-reps 0 --geo-engine split_w --settle 0 --extras off > gpurun_out/prof_$NAME.log 2>&1
f=$(find gpurun_out/prof_$NAME -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py "$f" 8 > gpurun_out/${NAME}_timeline.txt
rm -rf gpurun_out/prof_$NAME
grep -n "wgrad\|launches" gpurun_out/${NAME}_timeline.txt
