#!/usr/bin/env python3
"""Soak test of the inter path on the GPU box: one process encodes and decodes the same inter
stream over and over through the live seams (every call creates and destroys its contexts,
prediction objects, P-frame feed, synthesis object, the motion search's thread-local window
buffers) - packets must stay equal to the first round's (which is compared with the pure
reference build), resident set and device memory in use must stay flat.
  python tools/soak_inter.py [--rounds 12] [--frames 8] [--width 640 --height 360]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def device_used_mib():
    hip = ctypes.CDLL('libamdhip64.so')
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    if hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) != 0:
        return -1
    return (total.value - free.value) >> 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=12)
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--height', type=int, default=360)
    a = ap.parse_args()
    import psutil
    import daala_amd.hipenc as H
    from configs_round import frames_of, reference
    w, h, nf = a.width, a.height, a.frames
    buf = H.pack_frames(frames_of(w, h, nf, 41, step=(2, 3)), w, h)
    prm = H.Params(w, h, 20, 7, 1, 1, 0, 0, 30)
    want = reference(w, h, buf, nf, 20, 1, 30)[0]
    hdr = H.headers(prm)
    proc = psutil.Process()
    bad = 0
    first = None
    for r in range(a.rounds):
        n, pk, st = H.encode(prm, buf, nf, use_device=1)
        nd, pics, _, _ = H.decode(prm, hdr, pk, use_device=1)
        ok = n > 0 and pk == want and nd == nf and st.mv_check_fail == 0 and st.mv_bma_windows > 0
        if first is None:
            first = pics
        ok = ok and all((x == y).all() for p, q in zip(pics, first) for x, y in zip(p, q))
        bad += not ok
        print('round %2d: %s  rss %d MiB  device %d MiB  windows %d hits %d misses %d' % (
            r, 'ok ' if ok else 'BAD', proc.memory_info().rss >> 20, device_used_mib(), st.mv_bma_windows,
            st.mv_bma_hits, st.mv_bma_misses), flush=True)
    print('%d rounds, %d bad' % (a.rounds, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
