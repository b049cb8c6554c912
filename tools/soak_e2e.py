#!/usr/bin/env python3
"""Soak test of the live seams on the GPU box: repeats the 30-frame 1080p encode (all
packets compared with the plain reference search) and decode (all pictures compared with
the plain reference decoder) to flush out ordering races that a single pass can miss.
  python tools/soak_e2e.py [--rounds 4] [--workers 16]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=4)
    ap.add_argument('--workers', type=int, default=16)
    ap.add_argument('--frames', type=int, default=30)
    a = ap.parse_args()
    import daala_amd.hipenc as H
    import bench
    frames = bench.make_frames(a.frames, seed0=3)
    buf = H.pack_frames(frames, bench.PIC_W, bench.PIC_H)
    prm = H.Params(bench.PIC_W, bench.PIC_H, 20, 7, 1, a.workers, 0, 0)
    n0, want, _ = H.encode(prm, buf, a.frames)
    hdr = H.headers(prm)
    nd, pics0, _, _ = H.decode(prm, hdr, want)
    bad = 0
    for r in range(a.rounds):
        prm.batch = (0, 8, 5, 3)[r % 4]
        n, got, st = H.encode(prm, buf, a.frames, use_device=1)
        ok_e = n == n0 and got == want and st.lost_sync == 0
        nd1, pics1, sec, _ = H.decode(prm, hdr, want, use_device=1)
        ok_d = nd1 == a.frames and np.array_equal(pics0, pics1)
        print('round %d batch %d: encode %s (%.2f Mpix/s), decode %s (%.0f Mpix/s)'
              % (r, prm.batch, 'ok' if ok_e else 'MISMATCH',
                 a.frames*bench.PIC_W*bench.PIC_H/st.t_total_s/1e6, 'ok' if ok_d else 'MISMATCH',
                 a.frames*bench.PIC_W*bench.PIC_H/sec/1e6), flush=True)
        bad += (not ok_e) + (not ok_d)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
