# usage (GPU box): bash tools/phase.sh -- per-phase shader-clock shares of the MLP kernels (timing build; the product library is rebuilt afterwards)
cd $GRAFT_REPO_ROOT
SPF_EXTRA_HIPCC_FLAGS="-DSPF_TIMING" python -m spurfies_amd.build --force 2>&1 | grep -E "error"
python tools/phase_times.py split rays
python -m spurfies_amd.build --force 2>&1 | grep -E "error"
