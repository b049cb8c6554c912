"""The decoder's whole pixel-domain stage on the device against a REAL decode by
the reference (oracle/_ref/dec_probe.so, built in the dev container, travels with
gpurun): from the dequantised coefficients, block sizes, skip maps, dering flags
and quantizers the reference decoder used, the device must reproduce the
reference decoder's output picture bit-exactly."""
import os

import numpy as np
import pytest

from test_decode_tail import PROBE, oracle_tail, reference_decode

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(PROBE), reason='oracle/_ref not built')]


@pytest.mark.parametrize('w,h,quant,masking,seed', ((176, 112, 20, 1, 3), (176, 112, 8, 1, 5),
                                                     (352, 288, 30, 0, 6), (355, 291, 20, 1, 41),
                                                     (130, 66, 60, 0, 41)))
def test_device_decode_tail_equals_reference_decoder(w, h, quant, masking, seed):
    import daala_amd.binding as b
    r = None
    for attempt in range(6):          # the reference's dering decisions vary run to run
        r = reference_decode(w, h, quant, masking, seed + 10*attempt)
        if r['flags'].any() and (r['bsize'] == 3).any():
            break
    ctx = b.DaalaHip(w, h, r['fw'], r['fh'], nplanes=3, xdec=(0, 1, 1), nslots=1)
    ctx.set_bsize(0, r['bsize'])
    for pli in range(3):
        ctx.upload_coeffs(0, pli, r['d'][pli])
    ctx.set_decode_info(0, r['flags'], r['bskip'])
    thr = [int(1.0*pow(q, 0.84182)) for q in r['q']]
    ctx.decode_tail(thr, r['q'], 1)
    rec_o = oracle_tail(r)
    for pli in range(3):
        got = ctx.download_recon(0, pli)
        hh, ww = r['out'][pli].shape
        assert np.array_equal(got[:hh, :ww], r['out'][pli]), ('vs reference decoder', pli)
        assert np.array_equal(got, rec_o[pli]), ('vs oracle incl. padding', pli)
    ctx.close()


def test_device_decode_tail_forced_flags(w=192, h=128):
    """Every superblock filtered and every superblock unfiltered, random skip maps:
    device vs oracle on the same inputs (covers frame edges and skip handling even
    when the reference encoder happened to switch deringing off)."""
    import daala_amd.binding as b
    rng = np.random.default_rng(12)
    r = reference_decode(w, h, 25, 1, 9)
    for mode in (0, 1):
        r['flags'][:] = mode
        for pli in range(3):
            r['bskip'][pli][:] = (rng.random(r['bskip'][pli].shape) < (.4 if mode else 0)).astype(np.uint8)
        ctx = b.DaalaHip(w, h, r['fw'], r['fh'], nplanes=3, xdec=(0, 1, 1), nslots=1)
        ctx.set_bsize(0, r['bsize'])
        for pli in range(3):
            ctx.upload_coeffs(0, pli, r['d'][pli])
        ctx.set_decode_info(0, r['flags'], r['bskip'])
        ctx.decode_tail([int(1.0*pow(q, 0.84182)) for q in r['q']], r['q'], 1)
        rec_o = oracle_tail(r)
        for pli in range(3):
            assert np.array_equal(ctx.download_recon(0, pli), rec_o[pli]), (mode, pli)
        ctx.close()
