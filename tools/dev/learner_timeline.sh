#!/bin/bash
# Launch timeline of the MLP learner's batch-128 update (VERDICT r4 #3c): rocprofv3 kernel trace of tools/learner_bench.py, then per update the sum of
# kernel durations, the sum of the gaps between consecutive kernels and the span.  Run on the GPU box through gpurun.
OUT=$GRAFT_REPO_ROOT/gpurun_out/ltl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/learner_bench.py --batches 128 --no-torch --iters 200 > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("mzl::", "void mzl::"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# an update ends with k_learn_adam
ups, cur = [], []
for r in rows:
    cur.append(r)
    if "k_learn_adam" in r["Kernel_Name"]:
        ups.append(cur); cur = []
ups = ups[len(ups) // 2:]  # steady state
import statistics as st
def stats(u):
    dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in u) / 1e3
    gaps = sum(max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) for a, b in zip(u, u[1:])) / 1e3
    span = (int(u[-1]["End_Timestamp"]) - int(u[0]["Start_Timestamp"])) / 1e3
    return len(u), dur, gaps, span
S = [stats(u) for u in ups if len(u) > 5]
print("updates", len(S), "launches per update", st.median(s[0] for s in S))
print("per update (median, us): kernel time %.1f  gaps between kernels %.1f  span %.1f" % tuple(st.median(s[i] for s in S) for i in (1, 2, 3)))
between = [ (int(b[0]["Start_Timestamp"]) - int(a[-1]["End_Timestamp"])) / 1e3 for a, b in zip(ups, ups[1:]) ]
print("gap between updates (median, us): %.1f" % st.median(between))
u = ups[len(ups) // 2]
for a, b in zip(u, u[1:] + [None]):
    d = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 if b else 0.0
    print("  %-46s %6.1f us   then gap %5.1f us   grid %s x %s" % (a["Kernel_Name"][:46], d, g, a["Grid_Size_X"], a["Grid_Size_Y"]))
PY
find $OUT/trace -name "*_kernel_trace.csv" -size +2M -delete
