out=gpurun_out/r3d; mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py tests/test_gpu_selfplay.py tests/test_gpu_rng.py -x -q -m gpu > $out/tests.txt 2>&1; tail -3 $out/tests.txt
python tools/ab_bench.py muzero_amd/lib/libmz_nohw.so muzero_amd/lib/libmzplanner_hip.so 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
python tools/phase_profile.py cartpole 2>&1 | grep -v amdgpu.ids > $out/phase_c2.txt; cat $out/phase_c2.txt
python tools/phase_profile.py tictactoe 2>&1 | grep -v amdgpu.ids > $out/phase_c3.txt; cat $out/phase_c3.txt
