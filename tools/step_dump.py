#!/usr/bin/env python3
"""One training step out of a rocprofv3 kernel trace, launch by launch: start offset, queue (M = the training stream), kernel, duration,
grid / workgroup size and -- on the training stream -- the gap to the previous kernel.  Complements tools/step_timeline.py (per-queue
totals): this is the view that shows which kernels of the backward pass overlap which side-stream launches.
    usage: python tools/step_dump.py <p_kernel_trace.csv> [step_index_from_end=2] [filter]"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return n.split('(')[0][:64]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    flt = sys.argv[3] if len(sys.argv) > 3 else ''
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'adam_multi_kernel' in r['Kernel_Name']]
    a, b = marks[-back - 1] + 1, marks[-back] + 1
    step = rows[a:b]
    t0 = int(step[0]['Start_Timestamp'])
    qs = defaultdict(int)
    for r in step:
        qs[r['Queue_Id']] += 1
    main_q = max(qs, key=qs.get)
    prev_end = None
    for r in step:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        on_main = r['Queue_Id'] == main_q
        gap = ''
        if on_main:
            gap = '%6.1f' % ((s - prev_end) / 1e3) if prev_end else ''
            prev_end = e
        name = short(r['Kernel_Name'])
        if flt and flt not in name:
            continue
        print('%8.1f %s %-64s %7.1f us  grid %8s wg %4s  gap %s' % ((s - t0) / 1e3, 'M' if on_main else ' ' + r['Queue_Id'][-1], name, (e - s) / 1e3,
                                                                    r.get('Grid_Size_X', '?'), r.get('Workgroup_Size_X', '?'), gap))
    print('step: %d launches, %d on the training stream, %.3f ms from first start to last end' % (
        len(step), qs[main_q], (max(int(r['End_Timestamp']) for r in step) - t0) / 1e6))


if __name__ == '__main__':
    main()
