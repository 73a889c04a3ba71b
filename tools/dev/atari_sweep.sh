mkdir -p gpurun_out/cl
F='amdgpu.ids\|UserWarning\|Consider\|print('
run() { echo "== $*"; timeout 400 python tools/dev/conv_learner_check.py --atari --brief "$@" 2>&1 | grep -v "$F" | cut -c1-220; }
{
for s in 1 2 3 4 5 6 7 8; do run --chan 4 --planes 8 --blocks 1 --batch 3 --seed $s; done
} > gpurun_out/cl/atari_sweep.log 2>&1
grep "==\|closest\|represent" gpurun_out/cl/atari_sweep.log
