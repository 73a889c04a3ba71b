out=gpurun_out/r3e; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py tests/test_gpu_selfplay.py tests/test_gpu_rng.py -x -q -m gpu > $out/tests.txt 2>&1; tail -3 $out/tests.txt
timeout 120 python tools/ab_bench.py muzero_amd/lib/libmz_nohw.so muzero_amd/lib/libmzplanner_hip.so 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
