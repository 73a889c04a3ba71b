"""Launch gaps of the learner step from a rocprofv3 kernel trace: per update (k_learn_adam closes one) the span, the sum of kernel
durations and the idle time between consecutive kernels.   python tools/dev/gaps.py <trace dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]) for r in csv.DictReader(open(f)) if 'mzl' in r['Kernel_Name']))
steps, cur = [], []
for r in rows:
    cur.append(r)
    if 'k_learn_adam' in r[2]:
        steps.append(cur); cur = []
steps = [s for s in steps[len(steps) // 2:] if len(s) > 5]
n = len(steps)
span = sum(s[-1][1] - s[0][0] for s in steps) / n
busy = sum(sum(e - b for b, e, _ in s) for s in steps) / n
gaps = [steps[i][j + 1][0] - steps[i][j][1] for i in range(n) for j in range(len(steps[i]) - 1)]
inter = [steps[i + 1][0][0] - steps[i][-1][1] for i in range(n - 1)]
print('updates', n, 'kernels per update', len(steps[0]))
print('span us %.1f  busy us %.1f  mean gap inside an update us %.2f (max %.1f)  gap between updates us %.2f' % (span / 1e3, busy / 1e3, sum(gaps) / len(gaps) / 1e3, max(gaps) / 1e3, sum(inter) / len(inter) / 1e3))
for b, e, k in steps[0]: print('  %6.1f us  +%5.1f  %s' % ((b - steps[0][0][0]) / 1e3, (e - b) / 1e3, k))
