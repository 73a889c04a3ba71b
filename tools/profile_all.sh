#!/bin/bash
# rocprofv3 evidence for profiles/<round>/ (run on the GPU box through gpurun):
#   kernel trace + stats of the bench command per workload, then PMC passes (own runs, --kernel-trace only) on reduced
#   move counts -- counter collection serialises every dispatch.  Output: gpurun_out/profiles/<workload>/...
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
python3 -m muzero_amd.build > /dev/null  # never inside the profiled process (the .so normally travels with the snapshot)
cd /tmp && export TMPDIR=/tmp
run() {  # name, trace args, pmc args [, program (default bench.py)]
  local W=$1 TARGS=$2 PARGS=$3 PROG=${4:-bench.py}
  mkdir -p $OUT/$W
  timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$W/trace -o t -- python3 $R/$PROG $TARGS > $OUT/$W/trace.log 2>&1
  local i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"; do
    i=$((i+1))
    timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/$W/pmc$i -o p -- python3 $R/$PROG $PARGS > $OUT/$W/pmc$i.log 2>&1
  done
  python3 $R/tools/pmc_summary.py $OUT/$W "$TARGS" "$PARGS" $PROG > $OUT/$W/summary.json 2> $OUT/$W/summary.err
  # keep the merged output small: per-dispatch traces are not needed once summarised
  find $OUT/$W -name "*_kernel_trace.csv" -size +2M -delete
  find $OUT/$W -name "*_counter_collection.csv" -size +2M -delete
}
for W in "$@"; do
  case $W in
    c2) run c2 "--steps 10 --warmup 2 --no-cpu-baseline --no-sustained --no-e2e --no-configs --no-learner" "--steps 4 --warmup 1 --preheat 0 --no-cpu-baseline --no-sustained --no-e2e --no-configs --no-learner" ;;
    c3) run c3 "--workload c3 --steps 10 --warmup 2 --no-cpu-baseline --no-sustained" "--workload c3 --steps 4 --warmup 1 --preheat 0 --no-cpu-baseline --no-sustained" ;;
    c4) run c4 "--workload c4 --steps 1 --warmup 0 --no-cpu-baseline --no-sustained" "--workload c4 --steps 1 --warmup 0 --sims 3 --no-cpu-baseline --no-sustained" ;;
    learner) run learner "--batches 128,4096 --no-torch --iters 100" "--batches 128,4096 --no-torch --iters 10" tools/learner_bench.py ;;
    convlearner) run convlearner "--hip-only --iters 5" "--hip-only --iters 1" tools/conv_learner_bench.py ;;
    atarilearner) run atarilearner "--atari --chan 4 --planes 128 --blocks 8 --batch 128 --hip-only --iters 5" "--atari --chan 4 --planes 128 --blocks 8 --batch 128 --hip-only --iters 1" tools/conv_learner_bench.py ;;
    lunar) run lunar "--workload lunar --steps 10 --warmup 2 --no-cpu-baseline" "--workload lunar --steps 4 --warmup 1 --preheat 0 --no-cpu-baseline" ;;
    c5) run c5 "--workload c5 --steps 1 --warmup 0 --no-cpu-baseline --no-sustained" "--workload c5 --steps 1 --warmup 0 --sims 2 --no-cpu-baseline --no-sustained" ;;
  esac
done
ls -R $OUT | head -50
