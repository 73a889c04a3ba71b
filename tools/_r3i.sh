out=gpurun_out/r3i; mkdir -p $out
timeout 400 python tools/ab_env.py "MZ_LIB=muzero_amd/lib/libmz_rd3.so" "MZ_LIB=muzero_amd/lib/libmz_rd4.so" "MZ_LIB=muzero_amd/lib/libmz_rd5.so" "MZ_LIB=muzero_amd/lib/libmz_rd3.so" 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
MZ_LIB=x timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tictactoe or lunar" 2>&1 | tail -2
