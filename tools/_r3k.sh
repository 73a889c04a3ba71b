out=gpurun_out/r3k; mkdir -p $out
timeout 500 python tools/ab_env.py "" "MZ_FORCE_GENERIC=1" 2>&1 | grep -v amdgpu.ids | tee $out/ab_generic.txt
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_selfplay.py tests/test_gpu_api_errors.py -x -q -m gpu 2>&1 | tail -2
