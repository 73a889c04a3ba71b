out=gpurun_out/lb; mkdir -p $out
python -m pytest tests/test_gpu_hip_learner.py tests/test_gpu_launcher_flow.py -x -q -m gpu > $out/learner_tests.txt 2>&1; tail -2 $out/learner_tests.txt
MZ_FUZZ_LEARN_CASES=150 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k learner -x 2>&1 | tail -2
for v in 0 1; do echo no_overlap=$v
if [ $v = 1 ]; then export MZL_NO_OVERLAP=1; fi
python tools/learner_bench.py --batches 128,256,512,1024,1400 --no-torch > $out/lb.txt 2>&1; grep '^{"batch' $out/lb.txt | cut -c1-100
done
