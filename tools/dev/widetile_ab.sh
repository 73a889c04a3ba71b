for rep in 1 2; do
for v in wide12x16 all12x12; do
  if [ $v = all12x12 ]; then export MZLC_NO_WIDE_TILES=1; else unset MZLC_NO_WIDE_TILES; fi
  a=$(timeout 300 python tools/conv_learner_bench.py --atari --chan 4 --planes 128 --blocks 8 --batch 128 --iters 10 --hip-only 2>&1 | tail -1 | grep -o '"ms_hip": [0-9.]*')
  echo "$v atari $a"
done; done
