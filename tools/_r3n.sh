out=gpurun_out/r3n; mkdir -p $out
timeout 400 python -m pytest tests/test_gpu_trained.py -x -q -m gpu 2>&1 | tail -5
