out=$GRAFT_REPO_ROOT/gpurun_out/pmcl; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS -d $out/p1 -o p -- python3 $GRAFT_REPO_ROOT/tools/learner_bench.py --batches 4096 --no-torch --iters 10 > $out/p1.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $out/p2 -o p -- python3 $GRAFT_REPO_ROOT/tools/learner_bench.py --batches 4096 --no-torch --iters 10 > $out/p2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmcl'
for p in ('p1','p2'):
    f=glob.glob(out+'/'+p+'/**/*counter_collection.csv', recursive=True)
    if not f: print(p,'no file'); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].split('(')[0][:44]+' g'+r['Grid_Size']
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        if 'mzl' not in k: continue
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
find $out -name "*.csv" -size +1M -delete
