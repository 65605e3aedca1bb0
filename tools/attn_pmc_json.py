#!/usr/bin/env python
"""Summarise the passes of tools/attn_pmc.sh / tools/sq_pmc.sh: python tools/attn_pmc_json.py OUTDIR [name-filter[,name-filter...]] [workload text] > pmc.json"""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
filters = sys.argv[2].split(',') if len(sys.argv) > 2 else ['attn_']
workload = sys.argv[3] if len(sys.argv) > 3 else ('tools/attn_pmc_driver.py (self-attention B=16: N4096 h8 d40, N4096 h5 d64; plain and pre-scaled-query kernels)')
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(out, 'p*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not any(x in k for x in filters):
            continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[(k, f)].add(r['Dispatch_Id'])
ndisp = {}
for (k, f), s in disp.items():
    ndisp[k] = max(ndisp.get(k, 0), len(s))
stats = {}
for f in glob.glob(os.path.join(out, 'st', '**', '*kernel_stats.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r['Name']] = dict(calls=int(r['Calls']), avg_us=float(r['AverageNs']) / 1e3)
# durations of the SAME (profiled) dispatches from pass 1's kernel trace -> effective clock = GRBM_GUI_ACTIVE / duration
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, 'p1', '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name']].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
res = {}
for k, c in sorted(agg.items()):
    n = ndisp[k]
    g = lambda name: c.get(name, 0.0) / n     # noqa: E731
    wc = g('SQ_WAVE_CYCLES')
    d = {'dispatches': n, 'avg_us': stats.get(k, {}).get('avg_us'), 'SQ_WAVE_CYCLES': wc,
         'wait_any_frac': g('SQ_WAIT_ANY') / wc if wc else None, 'wait_inst_any_frac': g('SQ_WAIT_INST_ANY') / wc if wc else None,
         'active_inst_any_frac': g('SQ_ACTIVE_INST_ANY') / wc if wc else None, 'active_valu_frac': g('SQ_ACTIVE_INST_VALU') / wc if wc else None,
         'active_lds_frac': g('SQ_ACTIVE_INST_LDS') / wc if wc else None, 'wait_inst_lds_frac': g('SQ_WAIT_INST_LDS') / wc if wc else None,
         'mfma_busy_over_busy_cycles': g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_BUSY_CYCLES') if g('SQ_BUSY_CYCLES') else None,
         'insts_valu': g('SQ_INSTS_VALU'), 'insts_mfma': g('SQ_INSTS_MFMA'), 'insts_lds': g('SQ_INSTS_LDS'), 'insts_salu': g('SQ_INSTS_SALU'),
         'valu_per_mfma': g('SQ_INSTS_VALU') / g('SQ_INSTS_MFMA') if g('SQ_INSTS_MFMA') else None,
         'profiled_avg_us': (sum(dur[k]) / len(dur[k]) / 1e3) if dur.get(k) else None,
         'effective_clock_ghz': (g('GRBM_GUI_ACTIVE') / (sum(dur[k]) / len(dur[k]))) if dur.get(k) and g('GRBM_GUI_ACTIVE') else None,
         'mfma_busy_cycles_per_simd_over_gui_active': g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024.0 / g('GRBM_GUI_ACTIVE') if g('GRBM_GUI_ACTIVE') else None,
         # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs (the 'effective clock' above reads 14-19 GHz = 8 x 1.8-2.4): per-XCD corrected figures
         'effective_clock_ghz_per_xcd': (g('GRBM_GUI_ACTIVE') / 8.0 / (sum(dur[k]) / len(dur[k]))) if dur.get(k) and g('GRBM_GUI_ACTIVE') else None,
         'mfma_busy_frac_per_simd': g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024.0 / (g('GRBM_GUI_ACTIVE') / 8.0) if g('GRBM_GUI_ACTIVE') else None,
         'lds_bank_conflict_frac': g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE') if g('SQ_LDS_IDX_ACTIVE') else None}
    res[k] = d
print(json.dumps({'source': 'rocprofv3 --pmc (two passes of 8 SQ counters, --kernel-trace only) + a --stats pass on ' + workload,
                  'units': 'SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES in cycles; per dispatch',
                  'kernels': res}, indent=1))
