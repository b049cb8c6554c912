#!/usr/bin/env python3
"""A/B of integration-library builds on ONE box (the box-to-box spread of the host-bound number
is larger than most build-flag effects): alternates the 30-frame 1080p end-to-end encode through
a session between the libraries given, a few rounds each, and prints Mpixels/s per round.
  python tools/ab_hostlib.py daala_amd/host/build/libdaala_hipenc.so daala_amd/host/build_old/libdaala_hipenc.so
  python tools/ab_hostlib.py OD_HIP_SPIN_SYNC=0 OD_HIP_SPIN_SYNC=1        (environment variants)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r)
import daala_amd.hipenc as H, bench
frames = bench.make_frames(30, seed0=3)
buf = H.pack_frames(frames, bench.PIC_W, bench.PIC_H)
prm = H.Params(bench.PIC_W, bench.PIC_H, 20, 7, 1, 30, 0, 0)
S = H.Session(prm, use_device=1)
S.encode(buf, 30)
ts = []
for r in range(3):
    t0 = time.perf_counter(); n, pk, st = S.encode(buf, 30); ts.append(time.perf_counter() - t0)
S.close()
print(' '.join('%%.2f' %% (30*bench.PIC_W*bench.PIC_H/t/1e6) for t in ts))
''' % (ROOT, os.path.join(ROOT, 'tests'))

for rnd in range(2):
    for lib in sys.argv[1:]:
        if '=' in lib and not os.path.exists(lib):       # NAME=VALUE: an environment variant of the default build
            k, v = lib.split('=', 1)
            env = dict(os.environ, **{k: v})
        else:
            env = dict(os.environ, OD_HIPENC_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
        print(lib, out.stdout.strip() or out.stderr.strip()[-300:], flush=True)
