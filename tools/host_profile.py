#!/usr/bin/env python3
"""Flat host profile of the LIVE intra step (30 x 1080p, device feed, 30 workers) on the GPU box:
the integration library's sampling profiler (daala_amd/host/hip_prof.c: process CPU-time timer,
interrupted program counters) around two steps, samples mapped to symbols with nm (local symbols
included) and to libraries through /proc/self/maps.  Markdown on stdout.
  python tools/host_profile.py [--inter] > profiles/r03_host_profile.md"""
import argparse
import bisect
import collections
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np            # noqa: E402
import bench                  # noqa: E402
import daala_amd.hipenc as H  # noqa: E402


def symbols(path):
    for nm in ('nm', '/opt/rocm/lib/llvm/bin/llvm-nm'):
        try:
            out = subprocess.run([nm, '-n', '--defined-only', path], capture_output=True, text=True)
        except FileNotFoundError:
            continue
        if out.returncode == 0:
            tab = []
            for line in out.stdout.splitlines():
                f = line.split()
                if len(f) == 3 and f[1] in 'tTwW':
                    tab.append((int(f[0], 16), f[2]))
            return tab
    return []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--inter', action='store_true', help='profile a 1080p I P P P stream on one worker instead')
    a = ap.parse_args()
    lib = H.hipenc()
    lib.od_hipenc_prof_start.argtypes = [ctypes.c_long, ctypes.c_int]
    lib.od_hipenc_prof_stop.restype = ctypes.c_long
    lib.od_hipenc_prof_stop.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.c_long, ctypes.POINTER(ctypes.c_size_t)]
    W, Hh = 1920, 1080
    if a.inter:
        frames = bench.make_frames(4, seed0=1)
        buf, nf = H.pack_frames(frames, W, Hh), 4
        prm = H.Params(W, Hh, 20, 7, 1, 1, 0, 0, 30)
        what = '1080p I P P P, one worker, P-frame feed'
    else:
        frames = bench.make_frames(30, seed0=1)
        buf, nf = H.pack_frames(frames, W, Hh), 30
        prm = H.Params(W, Hh, 20, 7, 1, 30, 0, 30)
        what = 'intra step: 30 x 1080p, 30 workers, device feed'
    cap = 400000
    with H.Session(prm, use_device=1) as ses:
        ses.encode(buf, nf)
        assert lib.od_hipenc_prof_start(cap, 500) == 0
        for _ in range(a.steps):
            n, pk, st = ses.encode(buf, nf)
        pcs = (ctypes.c_size_t*cap)()
        base = ctypes.c_size_t()
        ticks = lib.od_hipenc_prof_stop(pcs, cap, ctypes.byref(base))
    got = min(ticks, cap)
    pcs = np.frombuffer(pcs, np.uint64)[:got]
    maps = []
    for line in open('/proc/self/maps'):
        f = line.split()
        if len(f) >= 6 and 'x' in f[1]:
            lo, hi = (int(v, 16) for v in f[0].split('-'))
            maps.append((lo, hi, int(f[2], 16), f[5]))
    maps.sort()
    symtabs = {}
    by_sym = collections.Counter()
    by_lib = collections.Counter()
    for pc in pcs.tolist():
        i = bisect.bisect_right([m[0] for m in maps], pc) - 1
        if i < 0 or pc >= maps[i][1]:
            by_lib['?'] += 1
            by_sym['?'] += 1
            continue
        lo, hi, off, path = maps[i]
        name = os.path.basename(path)
        by_lib[name] += 1
        if 'daala' not in name and 'libm' not in name:
            by_sym['[' + name + ']'] += 1
            continue
        if path not in symtabs:
            symtabs[path] = symbols(path)
        tab = symtabs[path]
        # file offset -> virtual address of a PIE/shared object: the first executable mapping's
        # offset equals its address offset for these libraries (no separate code segment alignment)
        first = min(m[0] - m[2] for m in maps if m[3] == path)
        addr = pc - first
        j = bisect.bisect_right([t[0] for t in tab], addr) - 1
        by_sym[(tab[j][1] if j >= 0 else '?') + ' (' + name.split('.')[0] + ')'] += 1
    print('# Host profile of the live seam (%s)\n' % what)
    print('Sampling profiler of the integration library (`daala_amd/host/hip_prof.c`, 500 us of process CPU time per '
          'sample; the kernel delivers one per scheduler tick, 250 per second of wall time), %d steps, %d samples; '
          'wall %.3f s per step.\n' % (a.steps, got, st.t_total_s))
    print('| library | samples | share |\n|---|---|---|')
    for k, v in by_lib.most_common(8):
        print('| %s | %d | %.1f %% |' % (k, v, 100.*v/got))
    print('\n| function | samples | share |\n|---|---|---|')
    for k, v in by_sym.most_common(45):
        print('| `%s` | %d | %.1f %% |' % (k, v, 100.*v/got))


if __name__ == '__main__':
    main()
