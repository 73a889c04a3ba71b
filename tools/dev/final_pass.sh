# the round's final verification pass on the GPU box: full -m gpu suite, rocprofv3 evidence of every workload, the driver-format line, the planner's deep parity jobs
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/full_gpu_tests.log
bash tools/profile_all.sh ${WORKLOADS:-c2 c3 c4 c5 lunar} > gpurun_out/profile_all.log 2>&1
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
JOBS="${DEEP_JOBS:-slow soak fuzz}" bash tools/dev/deep_parity.sh > /dev/null 2>&1
OUT=gpurun_out/deep
FP=$(python3 -c "from muzero_amd import build as b; print('planner sources', b.source_fingerprint(), '| learner sources', b.learner_fingerprint())")
{ echo "== fuzz, second seed region: MZ_FUZZ_SEED_OFFSET=1 MZ_FUZZ_CASES=300 python -m pytest tests/test_gpu_fuzz.py -q -m gpu"; echo "== build: $FP"; date -u; } > $OUT/fuzz_offset1.log
( export MZ_FUZZ_SEED_OFFSET=1 MZ_FUZZ_CASES=300; timeout 1500 python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -x --durations=5 2>&1 | grep -v "amdgpu.ids" | tail -40 ) >> $OUT/fuzz_offset1.log
{ echo "== soak, 48 seeds: MZ_SOAK_SEEDS=48 python -m pytest tests/test_gpu_soak.py -q -m gpu"; echo "== build: $FP"; date -u; } > $OUT/soak48.log
( export MZ_SOAK_SEEDS=48; timeout 1500 python3 -m pytest tests/test_gpu_soak.py -q -m gpu -x --durations=5 2>&1 | grep -v "amdgpu.ids" | tail -40 ) >> $OUT/soak48.log
tail -3 gpurun_out/full_gpu_tests.log
for f in slow soak fuzz fuzz_offset1 soak48; do tail -2 $OUT/$f.log | head -1; done
