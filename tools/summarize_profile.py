#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed
profiles/<tag>_* files: kernel stats CSV, bench lines and a traffic JSON with the
calibrated HBM bytes per launch of every transform kernel."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = os.path.join(ROOT, 'gpurun_out', 'prof_' + tag)
dst = os.path.join(ROOT, 'profiles')


def counters(sub, name):
    f = glob.glob(os.path.join(src, sub, '*', '*_counter_collection.csv'))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == name:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return agg


GIB = float(1 << 30)
cf = counters('calib_fetch', 'FETCH_SIZE')
cw = counters('calib_write', 'WRITE_SIZE')
# counters are in KiB-ish units of 1024 B (rocprofv3 derived metric): calibrate
f_dword = GIB/(cf[[k for k in cf if 'read_dword' in k][0]][-1]*1024)
f_int4 = GIB/(cf[[k for k in cf if 'read_int4' in k][0]][-1]*1024)
w_int4 = GIB/(cw[[k for k in cw if 'write_int4' in k][0]][-1]*1024)
fetch = counters('fetch', 'FETCH_SIZE')
write = counters('write', 'WRITE_SIZE')
out = {'calibration': {'fetch_factor_dword_loads': f_dword, 'fetch_factor_int4_loads': f_int4,
                       'write_factor_int4_stores': w_int4,
                       'method': 'known 1 GiB streamed by k_calib_* with the same access width; '
                                 'factor = true bytes / (counter*1024)'},
       'kernels': {}}
for k in sorted(fetch):
    if not k.startswith('void k_') and not k.startswith('k_'):
        continue
    if 'calib' in k or 'fill' in k:
        continue
    ff = f_dword if 'k_forward' in k else f_int4
    fb = fetch[k][-1]*1024*ff
    wb = write.get(k, [0])[-1]*1024*w_int4
    out['kernels'][k] = {'fetch_bytes': fb, 'write_bytes': wb, 'hbm_bytes': fb + wb,
                         'fetch_counter_kib': fetch[k][-1], 'write_counter_kib': write.get(k, [0])[-1]}
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha        # noqa: E402
out['kernel_sources_sha'] = kernel_sources_sha()     # bench.py flags a profile older than the kernels
with open(os.path.join(dst, tag + '_traffic.json'), 'w') as f:
    json.dump(out, f, indent=1, sort_keys=True)
st = glob.glob(os.path.join(src, 'stats', '*', '*_kernel_stats.csv'))[0]
shutil.copy(st, os.path.join(dst, tag + '_kernel_stats.csv'))
for name in ('bench.json', 'bench_profiled.json'):
    shutil.copy(os.path.join(src, name), os.path.join(dst, tag + '_' + name))
print(json.dumps(out['calibration'], indent=1))
for k, v in out['kernels'].items():
    print('%-48s fetch %8.1f MB  write %8.1f MB' % (k[:48], v['fetch_bytes']/1e6, v['write_bytes']/1e6))
