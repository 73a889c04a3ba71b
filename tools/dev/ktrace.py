"""Per-kernel, per-grid average durations from a rocprofv3 kernel trace CSV (last 3/4 of the launches of each kind)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'mzl' not in r['Kernel_Name'] and (len(sys.argv) < 3 or sys.argv[2] not in r['Kernel_Name']):
        continue
    agg[(r['Kernel_Name'].split('(')[0][:40], r['Grid_Size_X'], r['Grid_Size_Y'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(agg.items()):
    v = v[len(v) // 4:]
    print(k, len(v), 'avg us %.2f' % (sum(v) / len(v) / 1e3))
