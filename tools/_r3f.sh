out=gpurun_out/r3f; mkdir -p $out
timeout 120 python tools/phase_profile.py cartpole 2>&1 | grep -v amdgpu.ids > $out/phase_c2.txt; cat $out/phase_c2.txt
timeout 120 python tools/phase_profile.py tictactoe 2>&1 | grep -v amdgpu.ids > $out/phase_c3.txt; cat $out/phase_c3.txt
