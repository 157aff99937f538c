# usage (GPU box): bash tools/ab_step.sh "<flags A>" "<flags B>" -- same-box A/B of two library builds on the bench step
cd $GRAFT_REPO_ROOT
for F in "$1" "$2" "$1" "$2"; do
  SPF_EXTRA_HIPCC_FLAGS="$F" python -m spurfies_amd.build --force 2>&1 | grep -E "error"
  echo "== flags: [$F]"; python bench.py --steps 40 --warmup 10 --no-cpu-baseline | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % d['ms_per_step'])"
done
