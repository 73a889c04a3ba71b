timeout 300 python -m pytest tests/test_gpu_selfplay.py tests/test_gpu_parity.py tests/test_gpu_soak.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tools/ab_env.py "MZ_LIB=muzero_amd/lib/libmz_nohw.so" "" 2>&1 | grep -v amdgpu.ids
