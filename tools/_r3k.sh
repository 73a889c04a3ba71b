out=gpurun_out/r3k; mkdir -p $out
timeout 400 python tools/ab_env.py "MZ_HWX=0" "MZ_HWX=1" "MZ_HWX=3" "" 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
