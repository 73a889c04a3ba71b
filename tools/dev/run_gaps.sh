out=$GRAFT_REPO_ROOT/gpurun_out/gaps; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/t -o g -- python3 $GRAFT_REPO_ROOT/tools/learner_bench.py --batches 128 --no-torch --iters 300 > $out/log.txt 2>&1
grep '^{"batch' $out/log.txt | cut -c1-100
python3 $GRAFT_REPO_ROOT/tools/dev/gaps.py $out/t
find $out -name "*.csv" -size +1M -delete
