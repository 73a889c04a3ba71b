OUT=$GRAFT_REPO_ROOT/gpurun_out/deep
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
FP=$(python3 -c "from muzero_amd import build as b; print('planner sources', b.source_fingerprint(), '| learner sources', b.learner_fingerprint())")
for off in 1 2; do
{ echo "== learner fuzz, seed region $off: MZ_FUZZ_SEED_OFFSET=$off MZ_FUZZ_LEARN_CASES=300 MZ_FUZZ_CONV_LEARN_CASES=400 MZ_FUZZ_ATARI_LEARN_CASES=300 python -m pytest tests/test_gpu_fuzz.py -k learner -q -m gpu"; echo "== build: $FP"; date -u; } > $OUT/learnfuzz_offset$off.log
( export MZ_FUZZ_SEED_OFFSET=$off MZ_FUZZ_LEARN_CASES=300 MZ_FUZZ_CONV_LEARN_CASES=400 MZ_FUZZ_ATARI_LEARN_CASES=300; timeout 2400 python3 -m pytest tests/test_gpu_fuzz.py -k learner -q -m gpu --durations=3 2>&1 | grep -v "amdgpu.ids" | tail -25 ) >> $OUT/learnfuzz_offset$off.log
date -u >> $OUT/learnfuzz_offset$off.log
tail -2 $OUT/learnfuzz_offset$off.log
done
