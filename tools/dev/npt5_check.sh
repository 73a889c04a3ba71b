# quick timings of the conv learner over geometries that pick different pixel tilings (run through gpurun; compare builds by hand)
for rep in 1 2; do
  a=$(timeout 300 python tools/conv_learner_bench.py --atari --chan 4 --planes 128 --blocks 8 --batch 128 --iters 10 --hip-only 2>&1 | tail -1 | grep -o '"ms_hip": [0-9.]*')
  b=$(timeout 300 python tools/conv_learner_bench.py --board 13 --planes 64 --blocks 4 --batch 128 --iters 30 --hip-only 2>&1 | tail -1 | grep -o '"ms_hip": [0-9.]*')
  c=$(timeout 300 python tools/conv_learner_bench.py --board 14 --planes 64 --blocks 4 --batch 128 --iters 30 --hip-only 2>&1 | tail -1 | grep -o '"ms_hip": [0-9.]*')
  d=$(timeout 300 python tools/conv_learner_bench.py --hip-only --iters 30 2>&1 | tail -1 | grep -o '"ms_hip": [0-9.]*')
  echo "atari $a | 13x13/64/4 $b | 14x14/64/4 $c | C5 $d"
done
