# A/B of one create-time switch of the learner library on the C5 net's update: bash tools/dev/envab_c5.sh MZLC_DEFER_WGRAD 1
for rep in 1 2 3; do for v in "" ${2:-1}; do
  echo -n "$1=$v: "; env ${v:+$1=$v} python tools/conv_learner_bench.py --hip-only --iters 5 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms frac %.4f'%(d['ms_hip'], d['mfma_frac']))"
done; done
