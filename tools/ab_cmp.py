"""Per-step kernel-time comparison of two rocprofv3 kernel_stats.csv files (tools/ab_prof.sh):  python tools/ab_cmp.py base.csv work.csv [rows]"""
import csv, re, sys


def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
        n = re.sub(r'\(.*$', '', n).replace('void ', '')
        c, t = d.get(n, (0, 0.0))
        d[n] = (c + int(r['Calls']), t + float(r['TotalDurationNs']) / 1e3)
    steps = [v[0] for k, v in d.items() if k.startswith('adam')][0]
    return {k: (v[0] / steps, v[1] / steps) for k, v in d.items()}


a, b = load(sys.argv[1]), load(sys.argv[2])
rows = sorted(((b.get(n, (0, 0))[1] - a.get(n, (0, 0))[1], n) for n in set(a) | set(b)), key=lambda r: -abs(r[0]))
for d, n in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    x, y = a.get(n, (0, 0)), b.get(n, (0, 0))
    print('%-60s base %5.1f x %7.1f = %8.1f | work %5.1f x %7.1f = %8.1f | d %+7.1f us/step' % (
        n[:60], x[0], x[1] / max(x[0], 1e-9), x[1], y[0], y[1] / max(y[0], 1e-9), y[1], d))
print('launches/step: base %.1f work %.1f; kernel us/step: base %.1f work %.1f' % (
    sum(v[0] for v in a.values()), sum(v[0] for v in b.values()), sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
