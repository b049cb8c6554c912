#!/usr/bin/env python3
"""configs[3] through both seams at 1920x1080: I + (n - 1) P frames encoded by the session driver
(device feed, P-frame feed, device OBMC), decoded by the reference code on the host, by the
seam with the P frames' PVQ synthesis on the device and with it on the host (HIPDEC_SYNTH=0).
HIPDEC_DEBUG=2 prints every packet's time classes.  Also the workload of the inter kernel
profile (profiles/rNN_inter_kernel_stats.csv):
  rocprofv3 --kernel-trace --stats ... -- python3 tools/time_inter_decode.py 6"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import daala_amd.hipenc as H
from testlib import synth_plane
w, h, nf = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 8
base = [synth_plane(w + 64, h + 64, 31), synth_plane(w//2 + 32, h//2 + 32, 32, 1), synth_plane(w//2 + 32, h//2 + 32, 33, 1)]
frames = [[base[0][2*f:2*f + h, 3*f:3*f + w], base[1][f:f + h//2, (3*f)//2:(3*f)//2 + w//2],
           base[2][f:f + h//2, (3*f)//2:(3*f)//2 + w//2]] for f in range(nf)]
buf = H.pack_frames(frames, w, h)
prm = H.Params(w, h, 20, 7, 1, 1, 0, 0, 30)
t = time.time(); n, pk, st = H.encode(prm, buf, nf, use_device=1); print('encode %.2f s' % (time.time() - t), n, [len(p) for p in pk])
hdr = H.headers(prm)
n0, want, s0, _ = H.decode(prm, hdr, pk)
print('reference decode %.3f s' % s0)
for env, res in (('1', '1'), ('1', '0'), ('0', '1'), ('1', '1'), ('1', '0'), ('0', '1')):
    os.environ['HIPDEC_SYNTH'] = env
    os.environ['HIPDEC_REF_RESIDENT'] = res     # 0: references uploaded from the host image (rounds 2-3)
    nd, got, s1, d1 = H.decode(prm, hdr, pk, use_device=1)
    print('HIPDEC_SYNTH=%s HIPDEC_REF_RESIDENT=%s: %.3f s (device calls %.3f s) identical %s synth %s resident refs %d'
          % (env, res, s1, d1, np.array_equal(got, want), H.synth_stats(), H.ref_resident_frames()))
