#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection.csv for the dominant kernel: per-dispatch
means over the long dispatches only (the full-frame launches)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else 'resmlp'
by = collections.defaultdict(dict)
dur = {}
for r in rows:
    if pat in r['Kernel_Name']:
        by[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
if not dur:
    sys.exit('no dispatch matches ' + pat)
mx = max(dur.values())
keep = [d for d in dur if dur[d] > 0.7 * mx]
print(f'{len(keep)} long dispatches of {len(dur)}; mean duration {sum(dur[d] for d in keep)/len(keep):.3f} ms')
names = sorted({k for d in keep for k in by[d]})
mean = {n: sum(by[d].get(n, 0) for d in keep) / len(keep) for n in names}
for n in names:
    print(f'  {n:32s} {mean[n]:.4e}')
ms = sum(dur[d] for d in keep) / len(keep)
if 'GRBM_GUI_ACTIVE' in mean:
    print(f'  effective clock ~ {mean["GRBM_GUI_ACTIVE"]/8/(ms*1e-3)/1e9:.3f} GHz (GRBM_GUI_ACTIVE/8/duration)')
if 'SQ_WAVE_CYCLES' in mean:
    wc = mean['SQ_WAVE_CYCLES']
    for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS'):
        if n in mean:
            print(f'  {n}/SQ_WAVE_CYCLES = {mean[n]/wc:.3f}')
    for n in ('SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_LDS_CMD_FIFO_FULL',
              'SQ_LDS_DATA_FIFO_FULL'):
        if n in mean:
            print(f'  {n}/SQ_WAVE_CYCLES = {mean[n]/wc:.3f}')
    if 'SQ_VALU_MFMA_COEXEC_CYCLES' in mean:
        print(f'  SQ_VALU_MFMA_COEXEC_CYCLES / (4 SQ_WAVE_CYCLES) = {mean["SQ_VALU_MFMA_COEXEC_CYCLES"]/(4*wc):.3f}')
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in mean:
        print(f'  MFMA busy / wave cycles = {mean["SQ_VALU_MFMA_BUSY_CYCLES"]/(4*wc):.3f} (quad-cycle corrected)')
