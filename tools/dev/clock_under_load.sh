# shader clock while the conv learner's update runs (rocm-smi samples during tools/conv_learner_bench.py), and idle before it
echo "idle:"; rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2
python tools/conv_learner_bench.py --hip-only --iters 600 > /tmp/clb.log 2>&1 &
P=$!
sleep 14
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -3; sleep 0.7; done
wait $P; tail -1 /tmp/clb.log | cut -c1-200
