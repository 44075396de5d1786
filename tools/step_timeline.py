#!/usr/bin/env python3
"""Per-queue timeline of ONE training step out of a rocprofv3 kernel trace (p_kernel_trace.csv of tools/step_counters.sh):
for each HIP queue the busy time, the idle time between consecutive kernels (launch gaps + waits on the other queue) and the
kernels sorted by time -- answers "what is on the critical stream and how much of the step is gaps".
    usage: python tools/step_timeline.py <p_kernel_trace.csv> [step_index_from_end=2]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    return name.split('(')[0][:80]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # a step ends with the fused Adam launches: split there
    marks = [i for i, r in enumerate(rows) if 'adam_multi_kernel' in r['Kernel_Name'] or 'FusedAdam' in r['Kernel_Name']]
    ends = [m for j, m in enumerate(marks) if j + 1 == len(marks) or marks[j + 1] - m > 8]
    a, b = ends[-back - 1] + 1, ends[-back] + 1
    step = rows[a:b]
    t0, t1 = int(step[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in step)
    print('step: %d kernels, %.3f ms wall (first start -> last end)' % (len(step), (t1 - t0) / 1e6))
    by_q = defaultdict(list)
    for r in step:
        by_q[r['Queue_Id']].append(r)
    for q, rs in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
        busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
        gaps = [int(n['Start_Timestamp']) - int(p['End_Timestamp']) for p, n in zip(rs, rs[1:])]
        pos = [g for g in gaps if g > 0]
        span = int(rs[-1]['End_Timestamp']) - int(rs[0]['Start_Timestamp'])
        print('\nqueue %s: %d kernels, busy %.3f ms, span %.3f ms, idle between kernels %.3f ms (median gap %.1f us, %d gaps > 20 us = %.3f ms)' % (
            q, len(rs), busy / 1e6, span / 1e6, sum(pos) / 1e6, sorted(pos)[len(pos) // 2] / 1e3 if pos else 0,
            sum(1 for g in pos if g > 20000), sum(g for g in pos if g > 20000) / 1e6))
        agg = defaultdict(lambda: [0, 0])
        for r in rs:
            k = short(r['Kernel_Name'])
            agg[k][0] += 1
            agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
            print('  %-82s x%3d %8.1f us' % (k, n, t / 1e3))


if __name__ == '__main__':
    main()
