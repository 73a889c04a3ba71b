# kernel-trace statistics of the Atari learner update only (no PMC passes): gpurun_out/atari_trace/stats.csv
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}
OUT=$R/gpurun_out/atari_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/tools/conv_learner_bench.py --atari --chan 4 --planes 128 --blocks 8 --batch 128 --hip-only --iters 5 > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/stats.csv
find $OUT/trace -name "*_kernel_trace.csv" -delete
head -32 $OUT/stats.csv | cut -c1-150
