#!/bin/bash
# round 6: is the search kernel's instruction stream served from the instruction cache?  (run on the GPU box through gpurun)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}
OUT=$R/gpurun_out/icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 1 --preheat 0 --no-cpu-baseline --no-sustained --no-e2e --no-configs --no-learner"
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/pmc$i -o p -- python3 $R/bench.py $ARGS > $OUT/pmc$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'k_search_fast' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, (v, n) in acc.items(): print(f, k, 'per launch %.4g' % (v / n), 'launches', n)
PY
