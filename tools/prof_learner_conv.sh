#!/bin/bash
# rocprofv3 kernel trace + stats (csv) of the conv learner's update at the C5 net size (run on the GPU box through gpurun).  Usage: tools/prof_learner_conv.sh <tag> [bench args]
tag=${1:-lc}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/conv_learner_bench.py --hip-only --iters 5 "$@" > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -24 $f | cut -c1-160
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        if "k_lc_conv" in n or "k_lc_wgrad" in n:
            agg[(n[:40], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(k, "n", len(v), "avg %.1f min %.1f max %.1f us  total %.1f ms" % (sum(v) / len(v), min(v), max(v), sum(v) / 1e3))
PY
find $OUT/trace -name "*_kernel_trace.csv" -size +2M -delete
