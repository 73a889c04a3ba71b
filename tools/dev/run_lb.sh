out=gpurun_out/lb; mkdir -p $out
python -m pytest tests/test_gpu_hip_learner.py -x -q -m gpu > $out/learner_tests.txt 2>&1; tail -2 $out/learner_tests.txt
for v in 96 64 40 24; do echo chain_min=$v
MZL_CHAIN_MIN_TILES=$v python tools/learner_bench.py --batches 384,640,768,1024,1536,16384 --no-torch > $out/lb.txt 2>&1; grep '^{"batch' $out/lb.txt | cut -c1-100
done
