"""Test-side helpers on top of daala_amd/hipenc.py: an oracle-backed producer of
encoder-feed views (stand-in for od_hip_enc_feed_view in CPU tests, checker of the
device feed).  Test infrastructure."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from daala_amd.hipenc import *          # noqa: F401,F403
from daala_amd.hipenc import FeedLevel, level_params, p32, p16, pu8, pf64, I32P


def pulse_layout(off, nb):
    """(yo, ns) of every band in the feed's 16-bit pulse array (include/daala_hip.h 4b)."""
    return [0 if b == 0 else off[b] for b in range(nb)], [(off[b + 1] - off[b] + 1) & ~1 for b in range(nb)]

from testlib import oracle


class OracleFeed:
    """Encoder-feed views of one frame computed by the oracle (CPU): the stand-in for
    od_hip_enc_feed_view in CPU tests and the checker of the device feed."""

    def __init__(self, prm, luma_padded, lp=None):
        o = oracle()
        fh, fw = luma_padded.shape
        qm, q, beta = lp if lp is not None else level_params(prm)
        c = np.zeros((fh, fw), np.int32)
        lev = [np.zeros((fh, fw), np.int32) for _ in range(4)]
        levp = (I32P*4)(*[p32(a) for a in lev])
        o.orc_forward_pyramid_plane(p32(c), levp, 4, pu8(luma_padded), fw, fw//32, fh//32, 0,
                                    prm.pic_width, prm.pic_height)
        self.levels = []
        self.keep = []
        for l in range(4):
            n = 32 >> l
            nb = {32: 9, 16: 7, 8: 4, 4: 1}[n]
            off = [1, 16, 24, 32, 64, 96, 128, 256, 384, 512][:nb + 1]
            nbx, nblk = fw//n, (fw//n)*(fh//n)
            nrec = nb*nblk
            ncoded = min(n*n, 512)
            a = {'cg': np.zeros(nrec), 'g': np.zeros(nrec), 'ncand': np.zeros(nrec, np.int32),
                 'qg': np.zeros(2*nrec, np.int32), 'k': np.zeros(2*nrec, np.int32),
                 'cos_dist': np.zeros(2*nrec), 'y': np.zeros(2*nblk*(ncoded - 1), np.int32)}
            qi = np.ascontiguousarray(q[l].astype(np.int32))
            o.orc_feed_level(p32(lev[l]), fw, fh, n, p16(np.ascontiguousarray(qm[l])),
                             p32(qi), pf64(np.ascontiguousarray(beta[l])), pf64(a['cg']), pf64(a['g']),
                             p32(a['ncand']), p32(a['qg']), p32(a['k']), pf64(a['cos_dist']),
                             p32(a['y']))
            v = FeedLevel()
            v.n, v.nbands, v.nblk, v.nbx = n, nb, nblk, nbx
            for i, x in enumerate(off):
                v.off[i] = x
            v.cg, v.g, v.ncand, v.qg, v.k = pf64(a['cg']), pf64(a['g']), p32(a['ncand']), p32(a['qg']), p32(a['k'])
            # the oracle writes 32-bit pulses, band b at 2*nblk*(off[b] - 1), runs of n_b; the feed
            # carries 16-bit pulses, band b at 2*nblk*yo[b], runs of ns[b] (padded to even)
            yo, ns = pulse_layout(off, nb)
            y16 = np.zeros(2*nblk*ncoded, np.int16)
            for b in range(nb):
                nn = off[b + 1] - off[b]
                src = a['y'][2*nblk*(off[b] - 1): 2*nblk*(off[b] - 1) + 2*nblk*nn].reshape(2*nblk, nn)
                y16[2*nblk*yo[b]: 2*nblk*yo[b] + 2*nblk*ns[b]].reshape(2*nblk, ns[b])[:, :nn] = src
            a['y'] = y16
            v.cos_dist, v.y = pf64(a['cos_dist']), a['y'].ctypes.data_as(ctypes.POINTER(ctypes.c_int16))
            v.lev, v.lev_stride = p32(lev[l]), fw
            a['lev'] = lev[l]
            self.levels.append(v)
            self.keep.append(a)


