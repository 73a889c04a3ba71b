out=gpurun_out/lb; mkdir -p $out
python -m pytest tests/test_gpu_hip_learner.py -x -q -m gpu > $out/learner_tests.txt 2>&1; tail -2 $out/learner_tests.txt
python tools/learner_bench.py --batches 1024,4096,16384 --no-torch > $out/lb.txt 2>&1; grep '^{"batch' $out/lb.txt | cut -c1-120
cd /tmp && export TMPDIR=/tmp
for B in 4096 16384; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lb/lprof$B -o lb -- python3 $GRAFT_REPO_ROOT/tools/learner_bench.py --batches $B --no-torch --iters 50 > $GRAFT_REPO_ROOT/gpurun_out/lb/lprof$B.log 2>&1
python $GRAFT_REPO_ROOT/tools/dev/ktrace.py $GRAFT_REPO_ROOT/gpurun_out/lb/lprof$B | tee $GRAFT_REPO_ROOT/gpurun_out/lb/ktrace$B.txt
done
find $GRAFT_REPO_ROOT/gpurun_out/lb -name "*kernel_trace.csv" -delete
