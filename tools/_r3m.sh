out=gpurun_out/r3m; mkdir -p $out
timeout 700 python tools/trained_parity.py --train-steps 15000 2>&1 | grep -v amdgpu.ids | tee $out/trained_parity.txt
