#!/usr/bin/env python3
"""Times the HOST path of the integration library (device off, one worker, two 1080p frames of
the bench content) and prints an md5 of the packets: the A/B tool for host build flags
(OD_HIPENC_LIB=<other build>/libdaala_hipenc.so python tools/time_host_path.py)."""
import sys, time, hashlib
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tools')
import daala_amd.hipenc as H, bench
frames = bench.make_frames(2, seed0=3)
buf = H.pack_frames(frames, 1920, 1080)
prm = H.Params(1920, 1080, 20, 7, 1, 1, 0, 0)
best = 1e9
for r in range(3):
    n, pk, st = H.encode(prm, buf, 2)
    best = min(best, st.t_total_s)
print('bytes', n, 'best %.3f s' % best, hashlib.md5(b''.join(pk)).hexdigest())
