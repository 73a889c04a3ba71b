#!/bin/bash
# kernel trace + PMC passes of one conv-workload move (run on the GPU box through gpurun)
W=${1:-c5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 1 --warmup 0 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc1 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 1 --warmup 0 > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/pmc2 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 1 --warmup 0 > $OUT/pmc2.log 2>&1
find $OUT -name "*.csv" | head -20
