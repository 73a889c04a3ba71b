out=gpurun_out/r3c; mkdir -p $out
./tools/micro/mfma_valu > $out/mfma_valu.txt 2>&1; cat $out/mfma_valu.txt
