out=gpurun_out/lb; mkdir -p $out
python -m pytest tests/test_gpu_hip_learner.py tests/test_gpu_launcher_flow.py -x -q -m gpu > $out/learner_tests.txt 2>&1; tail -2 $out/learner_tests.txt
python tools/learner_bench.py --batches 128,256,1024,4096,16384 --no-torch > $out/lb.txt 2>&1; grep '^{"batch' $out/lb.txt | cut -c1-100
