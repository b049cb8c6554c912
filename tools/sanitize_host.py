#!/usr/bin/env python3
"""Workloads of tools/sanitize_host.sh: small streams through both seams with the device off
(the library named by OD_HIPENC_LIB is a sanitizer build).  Every line starting with "ok" is a
workload that ran to its end with packets / pictures as expected."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def frames(w, h, nf, seed):
    from testlib import synth_plane
    return [[synth_plane(w, h, seed + f), synth_plane(w//2, h//2, seed + f, 1),
             synth_plane(w//2, h//2, seed + f + 1, 1)] for f in range(nf)]


def main():
    import daala_amd.hipenc as H
    from test_hipenc_cpu import inter_stream_frames
    from test_gpu_hipenc import ref_encode
    if '--threads' in sys.argv:
        w, h, nf = 150, 100, 6
        buf = H.pack_frames(frames(w, h, nf, 9), w, h)
        prm = H.Params(w, h, 12, 7, 1, 3, 0, 0, 1)
        n, pk, st = H.encode(prm, buf, nf, use_device=0)
        nd = H.decode(prm, H.headers(prm), pk, use_device=0)[0]
        print('ok intra, 3 workers: %d bytes, %d pictures' % (n, nd), flush=True)
        return
    # inter streams in check mode, level-by-level EPZS walk (HIPENC_MV_EPZS=2 in the environment)
    for w, h, nf, kr in ((176, 144, 9, 4), (320, 200, 5, 30)):
        buf = inter_stream_frames(w, h, nf)
        prm = H.Params(w, h, 20, 7, 1, 1, 1, 0, kr)
        n, pk, st = H.encode(prm, buf, nf, use_device=0)
        same = pk == ref_encode(w, h, buf, nf, kr)
        nd = H.decode(prm, H.headers(prm), pk, use_device=0)[0]
        print('%s inter %dx%d x %d: packets equal %s, level walks %d, check failures %d, %d pictures'
              % ('ok' if same and st.mv_check_fail == 0 and nd == nf else 'BAD', w, h, nf, same, st.mv_level_walks,
                 st.mv_check_fail, nd), flush=True)
    w, h, nf = 160, 96, 2
    buf = H.pack_frames(frames(w, h, nf, 5), w, h)
    prm = H.Params(w, h, 0, 7, 1, 1, 0, 0, 1)
    n, pk, st = H.encode(prm, buf, nf, use_device=0)
    nd = H.decode(prm, H.headers(prm), pk, use_device=0)[0]
    print('ok lossless: %d bytes, %d pictures' % (n, nd), flush=True)
    for masking in (1, 0):
        w, h, nf = 150, 100, 4
        buf = H.pack_frames(frames(w, h, nf, 9), w, h)
        prm = H.Params(w, h, 12, 7, masking, 2, 0, 0, 1)
        n, pk, st = H.encode(prm, buf, nf, use_device=0)
        nd = H.decode(prm, H.headers(prm), pk, use_device=0)[0]
        print('ok intra masking %d, 2 workers: %d bytes, %d pictures' % (masking, n, nd), flush=True)


if __name__ == '__main__':
    main()
