for rep in 1 2; do
for v in gather mfma; do
  if [ $v = mfma ]; then export MZLC_ACT_MFMA=1; else unset MZLC_ACT_MFMA; fi
  a=$(timeout 300 python tools/conv_learner_bench.py --hip-only --iters 30 2>&1 | tail -1 | grep -o '"ms_hip": [0-9.]*')
  b=$(timeout 300 python tools/conv_learner_bench.py --board 9 --planes 128 --blocks 4 --batch 128 --iters 30 --hip-only 2>&1 | tail -1 | grep -o '"ms_hip": [0-9.]*')
  c=$(timeout 300 python tools/conv_learner_bench.py --atari --chan 4 --planes 128 --blocks 8 --batch 128 --iters 10 --hip-only 2>&1 | tail -1 | grep -o '"ms_hip": [0-9.]*')
  echo "$v C5 $a | 9x9/128/4 $b | atari $c"
done; done
