for rep in 1 2; do
  a=$(timeout 300 python tools/conv_learner_bench.py --atari --chan 4 --planes 128 --blocks 8 --batch 128 --iters 10 --hip-only 2>&1 | tail -1 | grep -o '"ms_hip": [0-9.]*')
  b=$(timeout 300 python tools/conv_learner_bench.py --board 5 --planes 64 --blocks 4 --batch 256 --iters 30 --hip-only 2>&1 | tail -1 | grep -o '"ms_hip": [0-9.]*')
  c=$(timeout 300 python tools/conv_learner_bench.py --board 9 --planes 128 --blocks 4 --batch 128 --iters 30 --hip-only 2>&1 | tail -1 | grep -o '"ms_hip": [0-9.]*')
  d=$(timeout 300 python tools/conv_learner_bench.py --board 3 --planes 16 --blocks 2 --batch 128 --iters 50 --hip-only 2>&1 | tail -1 | grep -o '"ms_hip": [0-9.]*')
  e=$(timeout 300 python tools/conv_learner_bench.py --board 8 --planes 64 --blocks 4 --batch 128 --iters 30 --hip-only 2>&1 | tail -1 | grep -o '"ms_hip": [0-9.]*')
  echo "atari $a | 5x5/64/4/b256 $b | 9x9/128/4 $c | 3x3/16/2 $d | 8x8/64/4 $e"
done
