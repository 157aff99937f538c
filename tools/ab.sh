# usage (GPU box): bash tools/ab.sh "<flags A>" "<flags B>"  (e.g. "" vs "-DSPF_SOLO")  -- same-box A/B of two library builds on the kernel micro-benchmarks
cd $GRAFT_REPO_ROOT
for F in "$1" "$2" "$1" "$2"; do
  SPF_EXTRA_HIPCC_FLAGS="$F" python -m spurfies_amd.build --force 2>&1 | grep -E "error"
  echo "== flags: [$F]"; python tools/geo_x3_check.py | grep split; python tools/color_bench.py split rays
done
