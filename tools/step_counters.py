"""Merge the four rocprofv3 runs of tools/step_counters.sh into one per-kernel table:
calls, mean duration, FETCH_SIZE (doubled: gfx950 reports half of a wide coalesced read stream, MI355X_MICROARCH.md HBM section)
and WRITE_SIZE per launch, achieved HBM TB/s, MFMA busy fraction, wave wait fractions.  usage: step_counters.py <dir> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    n = re.sub(r'\(anonymous namespace\)::', '', name)
    n = re.sub(r'^void ', '', n)
    n = n.split('(')[0]
    n = re.sub(r'at::native::', '', n)
    return n[:96]


def find(d, pat):
    files = glob.glob(d + '/**/' + pat, recursive=True)
    return files[0] if files else None


def pmc(d):
    f = find(d, '*counter_collection.csv')
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc


root, out = sys.argv[1], sys.argv[2]
stats = list(csv.DictReader(open(find(root + '/trace', '*kernel_stats.csv'))))
fetch, write, sq = pmc(root + '/fetch'), pmc(root + '/write'), pmc(root + '/sq')
mean = lambda xs: sum(xs) / len(xs) if xs else None
total_ns = sum(float(r['TotalDurationNs']) for r in stats)
rows = []
for r in stats:
    k = short(r['Name'])
    calls, avg = int(r['Calls']), float(r['AverageNs'])
    fe, wr = mean(fetch[k].get('FETCH_SIZE', [])), mean(write[k].get('WRITE_SIZE', []))  # KiB per launch
    row = {'kernel': k, 'calls': calls, 'avg_us': round(avg / 1e3, 2), 'share_of_kernel_time': round(float(r['TotalDurationNs']) / total_ns, 4)}
    if fe is not None and wr is not None:
        rb, wb = 2.0 * fe * 1024, wr * 1024
        row.update({'fetch_MB': round(rb / 1e6, 2), 'write_MB': round(wb / 1e6, 2), 'hbm_TBps': round((rb + wb) / avg / 1e3, 3)})
    s = sq.get(k)
    if s:
        busy, wave = mean(s.get('SQ_BUSY_CYCLES', [])), mean(s.get('SQ_WAVE_CYCLES', []))
        if busy and wave:
            dur = busy / 32.0  # 32 SEs report busy cycles
            row.update({'mfma_busy_frac': round(mean(s.get('SQ_VALU_MFMA_BUSY_CYCLES', [0])) / (1024.0 * dur), 4),
                        'wait_any_frac': round(mean(s.get('SQ_WAIT_ANY', [0])) / wave, 3),
                        'wait_inst_frac': round(mean(s.get('SQ_WAIT_INST_ANY', [0])) / wave, 3),
                        'active_inst_frac': round(mean(s.get('SQ_ACTIVE_INST_ANY', [0])) / wave, 3),
                        'valu_insts': mean(s.get('SQ_INSTS_VALU', [0])), 'lds_insts': mean(s.get('SQ_INSTS_LDS', [0]))})
    rows.append(row)
rows.sort(key=lambda r: -r['share_of_kernel_time'])
json.dump({'_comment': 'per launch means; fetch doubled per the gfx950 calibration note; hbm_TBps = (fetch + write) / duration; '
                       'mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32)', 'kernels': rows}, open(out, 'w'), indent=1)
for r in rows[:40]:
    print(json.dumps(r))
