#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/pass{1,2,3} (tools/pmc_round.sh) -> profiles/<tag>_pmc.json:
per kernel (counters SUMMED over all its dispatches of the run - a kernel launched once per
pyramid level has dispatches of very different sizes, and the last one alone, the smallest,
under-fills the chip), raw SQ/GRBM counters and derived figures:
  valu_per_wave          SQ_INSTS_VALU / SQ_WAVES
  valu_active_of_wave    SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   (quad-cycles both)
  parked_of_wave         SQ_WAIT_ANY / SQ_WAVE_CYCLES           (s_waitcnt / barrier)
  valu_pipe_utilisation  SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * kernel cycles), kernel cycles =
                         GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs)
Usage: python tools/summarize_pmc.py r02c"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
src = os.path.join(ROOT, 'gpurun_out', 'pmc_' + tag)
NSIMD = 256*4


def counters(sub):
    f = glob.glob(os.path.join(src, sub, '*', '*_counter_collection.csv'))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
    return agg


allc = collections.defaultdict(dict)
for p in ('pass1', 'pass2', 'pass3'):
    for k, d in counters(p).items():
        allc[k].update(d)
out = {}
for k, d in sorted(allc.items()):
    if not (k.startswith('void k_') or k.startswith('k_')):
        continue
    e = dict(d)
    if d.get('SQ_WAVES'):
        e['valu_per_wave'] = d.get('SQ_INSTS_VALU', 0)/d['SQ_WAVES']
    if d.get('SQ_WAVE_CYCLES'):
        e['valu_active_of_wave'] = d.get('SQ_ACTIVE_INST_VALU', 0)/d['SQ_WAVE_CYCLES']
        e['parked_of_wave'] = d.get('SQ_WAIT_ANY', 0)/d['SQ_WAVE_CYCLES']
        e['issue_stall_of_wave'] = d.get('SQ_WAIT_INST_ANY', 0)/d['SQ_WAVE_CYCLES']
    if d.get('GRBM_GUI_ACTIVE') and 'SQ_ACTIVE_INST_VALU' in d:
        cyc = d['GRBM_GUI_ACTIVE']/8
        e['kernel_cycles'] = cyc
        e['valu_pipe_utilisation'] = d['SQ_ACTIVE_INST_VALU']*4/(NSIMD*cyc)
    out[k] = e
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha        # noqa: E402
with open(os.path.join(ROOT, 'profiles', tag + '_pmc.json'), 'w') as f:
    json.dump({'what': __doc__, 'kernel_sources_sha': kernel_sources_sha(), 'kernels': out}, f, indent=1,
              sort_keys=True)
for k, e in out.items():
    print('%-52s' % k[:52], ' '.join('%s=%.3g' % (n, e[n]) for n in
          ('valu_per_wave', 'valu_active_of_wave', 'parked_of_wave', 'valu_pipe_utilisation') if n in e))
