OUT=$GRAFT_REPO_ROOT/gpurun_out/deep
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
FP=$(python3 -c "from muzero_amd import build as b; print('planner sources', b.source_fingerprint(), '| learner sources', b.learner_fingerprint())")
{ echo "== fuzz, second seed region: MZ_FUZZ_SEED_OFFSET=1 MZ_FUZZ_CASES=300 python -m pytest tests/test_gpu_fuzz.py -q -m gpu"; echo "== build: $FP"; date -u; } > $OUT/fuzz_offset1.log
( export MZ_FUZZ_SEED_OFFSET=1 MZ_FUZZ_CASES=300; timeout 1500 python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu --durations=5 2>&1 | grep -v "amdgpu.ids" | tail -30 ) >> $OUT/fuzz_offset1.log
date -u >> $OUT/fuzz_offset1.log
tail -3 $OUT/fuzz_offset1.log
{ echo "== soak, 48 seeds: MZ_SOAK_SEEDS=48 python -m pytest tests/test_gpu_soak.py -q -m gpu"; echo "== build: $FP"; date -u; } > $OUT/soak48.log
( export MZ_SOAK_SEEDS=48; timeout 1500 python3 -m pytest tests/test_gpu_soak.py -q -m gpu --durations=5 2>&1 | grep -v "amdgpu.ids" | tail -30 ) >> $OUT/soak48.log
date -u >> $OUT/soak48.log
tail -3 $OUT/soak48.log
