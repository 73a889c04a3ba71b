# round 6 A/B of the Atari learner's tile path and of the C5 net's update: learner library builds given as arguments (MZL_LIB_PATH), alternating
for rep in 1 2; do
for lib in "$@"; do
  echo -n "$lib atari: "; MZL_LIB_PATH=$lib python tools/conv_learner_bench.py --atari --chan 4 --planes 128 --blocks 8 --batch 128 --hip-only --iters 8 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms frac %.4f'%(d['ms_hip'], d['mfma_frac']))"
  echo -n "$lib c5:    "; MZL_LIB_PATH=$lib python tools/conv_learner_bench.py --hip-only --iters 5 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms frac %.4f'%(d['ms_hip'], d['mfma_frac']))"
done; done
