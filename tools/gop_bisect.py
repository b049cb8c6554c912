#!/usr/bin/env python3
"""Diagnostic for tests/test_gpu_hipenc.py::test_inter_gop_golden_frames_and_second_keyframe: the
33-frame GOP stream through the encoder seam with the device stages switched off one at a time
(HIPENC_MV_SADS, HIPENC_PFEED), check mode on and off; prints which packets differ from the pure
reference build and every check counter.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import hipenc_lib as H
from test_hipenc_cpu import inter_stream_frames
from test_gpu_hipenc import ref_encode
w, h, nf = 176, 144, 33
buf = inter_stream_frames(w, h, nf)
want = ref_encode(w, h, buf, nf, keyrate=30)
for env in ({}, {'HIPENC_MV_SADS': '0'}, {'HIPENC_PFEED': '0'}, {'HIPENC_MV_SADS': '0', 'HIPENC_PFEED': '0'}):
    for k in ('HIPENC_MV_SADS', 'HIPENC_PFEED'):
        os.environ.pop(k, None)
    os.environ.update(env)
    for check in (1, 0):
        prm = H.Params(w, h, 20, 7, 1, 1, check, 0, 30)
        n, pk, st = H.encode(prm, buf, nf, use_device=1)
        bad = [f for f in range(nf) if pk[f] != want[f]]
        d = st.as_dict()
        print(env, 'check', check, 'bad frames', bad, {k: d[k] for k in ('pvq_check_fail', 'check_fail', 'fdct_check_fail', 'dering_check_fail', 'dist_check_fail', 'mv_check_fail', 'lost_sync', 'pfeed_frames', 'mv_dev_calls', 'g2_mismatch')}, flush=True)
