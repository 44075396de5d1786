#!/usr/bin/env python3
"""profiles/rNN_step_traffic.json (tools/step_counters.py: rocprofv3 PMC passes of the training step) as a table: every kernel of the step
against the two rooflines that can bound it -- HBM (8 TB/s) and the dense bf16 MFMA pipe (busy fraction of 1024 SIMDs).
    python tools/traffic_table.py profiles/r04_step_traffic.json > profiles/r04_step_rooflines.md"""
import json
import sys

HBM_PEAK_TBPS = 8.0
d = json.load(open(sys.argv[1]))
rows = sorted(d['kernels'], key=lambda k: -k['calls'] * k['avg_us'])
steps = max(1, round(min(k['calls'] for k in rows if 'adam' in k['kernel'] or 'seg_loss_kernel' in k['kernel'])))
print('# Kernels of the training step against their rooflines ({})\n'.format(sys.argv[1]))
print(d.get('_comment', ''), '\n')
print('Counter passes over {} steps of `bench.py --train-only` at B = 32; time and traffic per LAUNCH, `per step` = launches x time / steps; '
      'HBM fraction against {} TB/s; MFMA = SQ_VALU_MFMA_BUSY_CYCLES share of the 1024 SIMDs.\n'.format(steps, HBM_PEAK_TBPS))
print('| kernel | launches / step | us / launch | us / step | fetch + write MB | TB/s | of HBM peak | MFMA busy | waiting on memory | issuing |')
print('|---|---|---|---|---|---|---|---|---|---|')
tot = 0.0
for k in rows:
    per = k['calls'] / steps
    us_step = per * k['avg_us']
    tot += us_step
    if us_step < 15:
        continue
    print('| `{}` | {:.2f} | {:.1f} | {:.0f} | {:.1f} | {:.2f} | {:.2f} | {:.2f} | {:.2f} | {:.2f} |'.format(
        k['kernel'][:70], per, k['avg_us'], us_step, k['fetch_MB'] + k['write_MB'], k['hbm_TBps'], k['hbm_TBps'] / HBM_PEAK_TBPS,
        k['mfma_busy_frac'], k.get('wait_any_frac', 0.0), k.get('active_inst_frac', 0.0)))
print('\nSum of kernel time over all streams: {:.2f} ms per step (kernels under 15 us per step left out of the table).'.format(tot / 1e3))
gb = sum(k['calls'] / steps * (k['fetch_MB'] + k['write_MB']) for k in rows) / 1e3
print('HBM traffic: {:.1f} GB per step.'.format(gb))
