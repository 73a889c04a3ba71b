# round 6: images per staging round (MZLC_WGRAD_SG) x least images per workgroup (MZLC_WGRAD_MIN_IPW) of the 6 x 6 towers' weight gradient, Atari update
for sg in 16 7 6 5 4 3 2; do for ipw in 1 8 16; do
  echo -n "sg<=$sg min_ipw=$ipw: "; MZLC_WGRAD_SG=$sg MZLC_WGRAD_MIN_IPW=$ipw python tools/conv_learner_bench.py --atari --chan 4 --planes 128 --blocks 8 --batch 128 --hip-only --iters 8 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms frac %.4f'%(d['ms_hip'], d['mfma_frac']))"
done; done
