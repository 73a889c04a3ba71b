for rep in 1 2; do
for arm in "MZLC_NO_HALO_IN=1 MZLC_NO_RING_ROWS=1" "MZLC_NO_RING_ROWS=1" "MZLC_X=1"; do
  echo -n "$arm: "; env $arm python tools/conv_learner_bench.py --atari --chan 4 --planes 128 --blocks 8 --batch 128 --hip-only --iters 8 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms frac %.4f'%(d['ms_hip'], d['mfma_frac']))"
done; done
