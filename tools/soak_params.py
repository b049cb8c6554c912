#!/usr/bin/env python3
"""Parameter sweep of the live encoder/decoder seams in check mode (GPU box): quantizers,
activity masking, several picture sizes and seeds; every device answer is compared with
the C search/transform, packets with the plain reference, pictures with the reference
decoder.  Exit code 1 on any difference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    import daala_amd.hipenc as H
    from testlib import synth_plane
    bad = 0
    total = dict(dev_hits=0, g2=0, fdct=0)
    cases = [(1280, 720, q, m, s) for q in (3, 10, 20, 45, 100, 250) for m in (0, 1) for s in (0,)]
    cases += [(w, h, 20, 1, 7) for (w, h) in ((1920, 1080), (720, 576), (854, 480), (98, 50))]
    cases += [(640, 360, 0, 1, 3)]                      # lossless
    cases += [(640, 360, 20, 1, 5, c) for c in (0, 1, 4, 10)]   # OD_SET_COMPLEXITY (0, 1: no block-size RDO)
    for case in cases:
        (w, h, q, m, seed), cx = case[:5], (case[5] if len(case) > 5 else 7)
        cw, ch = (w + 1)//2, (h + 1)//2
        nf = 4
        fr = [[synth_plane(w, h, seed + f), synth_plane(cw, ch, seed + f, 1), synth_plane(cw, ch, seed + f + 1, 1)]
              for f in range(nf)]
        buf = H.pack_frames(fr, w, h)
        prm = H.Params(w, h, q, cx, m, 4, 1, 0)
        n0, want, _ = H.encode(prm, buf, nf)
        pure_ok = True
        if cx != 7:
            import configs_round as C0
            pure_ok = C0.reference(w, h, buf, nf, q, m, 1, cx)[0] == want
        n, got, st = H.encode(prm, buf, nf, use_device=1)
        hdr = H.headers(prm)
        _, p0, _, _ = H.decode(prm, hdr, want)
        _, p1, _, _ = H.decode(prm, hdr, want, use_device=1)
        ok = (pure_ok and n == n0 and got == want and st.check_fail == 0 and st.lost_sync == 0 and st.pvq_check_fail == 0
              and st.g2_mismatch == 0 and st.dist_check_fail == 0
              and st.fdct_check_fail == 0 and st.dering_check_fail == 0
              and (st.dering_dev_sbs > 0 if q > 0 else st.haar_hits > 0)
              and np.array_equal(p0, p1))
        total['dev_hits'] += st.dev_hits
        total['g2'] += st.g2_mismatch
        total['fdct'] += st.fdct_hits
        print('%4dx%-4d q=%-3d masking=%d complexity=%d: %s  (hits %d, g2 %d, fdct %d, bytes %d)'
              % (w, h, q, m, cx, 'ok' if ok else 'MISMATCH', st.dev_hits, st.g2_mismatch, st.fdct_hits, n),
              flush=True)
        bad += not ok
    # inter streams (configs[3]): one worker, frames in order, check mode (every device OBMC frame
    # and every keyframe answer compared), packets against the PURE reference build
    import configs_round as C
    for (w, h, q, m, keyrate, nf, cx) in ((640, 360, 20, 1, 3, 6, 7), (352, 288, 10, 0, 4, 5, 7), (98, 50, 60, 1, 2, 5, 7),
                                          (416, 240, 35, 1, 30, 4, 7), (352, 288, 20, 1, 3, 4, 2), (352, 288, 20, 1, 3, 4, 10)):
        buf = H.pack_frames(C.frames_of(w, h, nf, 50 + q, step=(2, 3)), w, h)
        want, _ = C.reference(w, h, buf, nf, q, m, keyrate, cx)
        prm = H.Params(w, h, q, cx, m, 1, 1, 0, keyrate)
        n, got, st = H.encode(prm, buf, nf, use_device=1)
        hdr = H.headers(prm)
        nd0, p0, _, _ = H.decode(prm, hdr, want)
        nd1, p1, _, _ = H.decode(prm, hdr, want, use_device=1)
        fdev, fbad = H.mc_stats()
        mdh, mdbad = H.md_stats()
        ok = (got == want and st.check_fail == 0 and st.pvq_check_fail == 0 and st.lost_sync == 0
              and st.fdct_check_fail == 0 and st.dering_check_fail == 0 and st.dist_check_fail == 0
              and nd0 == nf and nd1 == nf and np.array_equal(p0, p1) and fbad == 0 and mdbad == 0
              and mdh > 0 and H.tail_frames() == nf)
        print('%4dx%-4d q=%-3d masking=%d complexity=%d inter keyrate=%d: %s  (bytes %d, device OBMC frames in decode %d)'
              % (w, h, q, m, cx, keyrate, 'ok' if ok else 'MISMATCH', n, fdev), flush=True)
        bad += not ok
    print('total', total)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
