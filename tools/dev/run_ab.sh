out=gpurun_out/ab; mkdir -p $out
python tools/ab_bench.py muzero_amd/lib/ab_hid64.so muzero_amd/lib/ab_hid32.so muzero_amd/lib/ab_hid64.so muzero_amd/lib/ab_hid32.so 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab/lprof -o lb -- python3 $GRAFT_REPO_ROOT/tools/learner_bench.py --batches 4096,16384 --no-torch --iters 50 > $GRAFT_REPO_ROOT/gpurun_out/ab/lprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/ktrace.py gpurun_out/ab/lprof | tee $out/ktrace.txt
find gpurun_out/ab/lprof -name "*kernel_trace.csv" -delete
