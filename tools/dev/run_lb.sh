out=gpurun_out/lb; mkdir -p $out
python -m pytest tests/test_gpu_hip_learner.py -x -q -m gpu > $out/learner_tests.txt 2>&1; tail -2 $out/learner_tests.txt
for i in 1 2; do
python tools/learner_bench.py --batches 4096,16384 --no-torch > $out/lb.txt 2>&1; grep '^{"batch' $out/lb.txt | cut -c1-120
python tools/learner_bench.py --batches 4096,16384 --no-torch --lib muzero_amd/lib/ab_learner_global.so > $out/lbg.txt 2>&1; grep '^{"batch' $out/lbg.txt | cut -c1-120
done
