export MZ_FUZZ_SEED_OFFSET=1
for v in none MZLC_NO_TAPSETS MZLC_NO_WIDE_TILES MZLC_NO_WGRAD_STACK MZLC_DENSE_TILING MZLC_NO_XCD_REMAP MZLC_NO_PAIR; do
  echo "== switch: $v"
  if [ $v = none ]; then timeout 300 python tools/dev/atari_kf_probe.py 16 2>&1 | grep "conv_2.weight\|res_blocks_3.1.conv_block2.1.weight" | cut -c1-120
  else env $v=1 timeout 300 python tools/dev/atari_kf_probe.py 16 2>&1 | grep "conv_2.weight\|res_blocks_3.1.conv_block2.1.weight" | cut -c1-120; fi
done
