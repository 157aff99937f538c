cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r5a_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5a_pytest.log
tail -5 gpurun_out/r5a_pytest.log
timeout 900 python3 tools/strong_proxy.py --out gpurun_out/r5a_strong_proxy.json --steps 60 > gpurun_out/r5a_strong_proxy.log 2>&1
tail -2 gpurun_out/r5a_strong_proxy.log | cut -c1-1500
