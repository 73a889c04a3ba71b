out=gpurun_out/c5f; mkdir -p $out
python -m pytest tests/test_gpu_conv.py -x -q -m gpu > $out/conv_tests.txt 2>&1; tail -3 $out/conv_tests.txt
for v in 1 0 1 0; do
MZ_ACTION_FUSE=$v python bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline --no-sustained > $out/c5_$v.json 2>$out/c5_$v.err
python -c "import json;d=json.load(open('$out/c5_$v.json'));print('c5 fuse=$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_step'))"
done
