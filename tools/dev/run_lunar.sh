out=gpurun_out/ac4; mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_selfplay.py tests/test_gpu_soak.py -x -q -m gpu > $out/parity.txt 2>&1; tail -3 $out/parity.txt
python bench.py --workload lunar --no-cpu-baseline > $out/lunar.json 2> $out/lunar.err
python -c "import json;d=json.load(open('$out/lunar.json'));print('lunar', d.get('value'), d.get('roofline',{}).get('frac'))"
for i in 1 2; do
python bench.py --no-cpu-baseline --no-sustained --no-e2e --no-configs --no-learner > $out/c2.json 2>$out/c2.err; python -c "import json;d=json.load(open('$out/c2.json'));print('c2', d['value'], d['roofline']['frac'])"
done
python bench.py --workload c3 --no-cpu-baseline --no-sustained > $out/c3.json 2>$out/c3.err; python -c "import json;d=json.load(open('$out/c3.json'));print('c3', d['value'], d['roofline']['frac'])"
python -m pytest tests/test_gpu_hip_learner.py -x -q -m gpu > $out/learner_tests.txt 2>&1; tail -5 $out/learner_tests.txt
python tools/learner_bench.py --batches 128,1024,4096,16384 --no-torch > $out/lb.txt 2>&1; grep '^{"batch' $out/lb.txt
