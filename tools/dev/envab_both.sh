# A/B of one create-time switch of the learner library on the Atari AND the C5 update: bash tools/dev/envab_both.sh MZLC_NO_SPLIT
for rep in 1 2; do for v in "" 1; do
  echo -n "$1=$v atari: "; env ${v:+$1=$v} python tools/conv_learner_bench.py --atari --chan 4 --planes 128 --blocks 8 --batch 128 --hip-only --iters 8 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms frac %.4f'%(d['ms_hip'], d['mfma_frac']))"
  echo -n "$1=$v c5:    "; env ${v:+$1=$v} python tools/conv_learner_bench.py --hip-only --iters 5 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms frac %.4f'%(d['ms_hip'], d['mfma_frac']))"
done; done
