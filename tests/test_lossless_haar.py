"""Lossless frames (BASELINE configs[4]): whole-superblock Haar planes.  The pin is a
REAL lossless encode + decode by the reference (oracle/_ref/dec_probe.so, quantizer
0): the coefficient planes its decoder reconstructed from must equal the forward
Haar of the input, and the inverse of them must equal its output picture (= the
input, the round trip is lossless).  CPU: oracle vs reference; GPU: device vs both."""
import ctypes
import os

import numpy as np
import pytest

from testlib import oracle, p32, pu8, synth_plane
from test_decode_tail import PROBE, reference_decode

needs_ref = pytest.mark.skipif(not os.path.exists(PROBE), reason='oracle/_ref not built')


def input_planes(w, h, seed):
    return [synth_plane(w, h, seed), synth_plane(w//2, h//2, seed, 1),
            synth_plane(w//2, h//2, seed + 1, 1)]


@needs_ref
@pytest.mark.parametrize('w,h,seed', ((128, 64, 3), (192, 128, 8)))
def test_oracle_haar_planes_equal_reference_lossless_codec(w, h, seed):
    o = oracle()
    r = reference_decode(w, h, 0, 1, seed)           # picture size == frame size: no padding
    assert r['q'][0] == 0 and (r['fw'], r['fh']) == (w, h)
    src = input_planes(w, h, seed)
    for pli in range(3):
        pw, ph, sb = w >> (pli > 0), h >> (pli > 0), 32 >> (pli > 0)
        assert np.array_equal(r['out'][pli], src[pli]), 'reference round trip is lossless'
        d = np.zeros((ph, pw), np.int32)
        o.orc_haar_forward_plane(p32(d), pu8(src[pli]), pw, ph, sb)
        assert np.array_equal(d, r['d'][pli]), ('forward vs reference decoder coefficients', pli)
        back = np.zeros((ph, pw), np.uint8)
        o.orc_haar_inverse_plane(pu8(back), p32(r['d'][pli]), pw, ph, sb)
        assert np.array_equal(back, r['out'][pli])


@pytest.mark.gpu
@needs_ref
def test_device_haar_planes_equal_reference_lossless_codec():
    import daala_amd.binding as b
    w, h, seed = 192, 128, 8
    r = reference_decode(w, h, 0, 1, seed)
    src = input_planes(w, h, seed)
    ctx = b.DaalaHip(w, h, w, h, nplanes=3, xdec=(0, 1, 1), nslots=2)
    ctx.upload_planes(1, src)
    ctx.forward_haar(1, 1)
    for pli in range(3):
        assert np.array_equal(ctx.download_coeffs(1, pli), r['d'][pli]), pli
    for pli in range(3):
        ctx.upload_coeffs(0, pli, r['d'][pli])
    ctx.inverse_haar(0, 1)
    for pli in range(3):
        assert np.array_equal(ctx.download_recon(0, pli), r['out'][pli]), pli
    ctx.close()


@pytest.mark.gpu
def test_device_haar_round_trip_1080p_and_oracle():
    """Full BASELINE size: device forward == oracle forward, inverse(forward(x)) == x
    (size-independent property), 3 frames, extreme content included."""
    import daala_amd.binding as b
    o = oracle()
    fw, fh = 1920, 1088
    rng = np.random.default_rng(5)
    frames = [input_planes(fw, fh, 1),
              [rng.integers(0, 256, (fh >> d, fw >> d), dtype=np.uint8) for d in (0, 1, 1)],
              [np.where(rng.random((fh >> d, fw >> d)) < .5, 0, 255).astype(np.uint8) for d in (0, 1, 1)]]
    ctx = b.DaalaHip(1920, 1080, fw, fh, nplanes=3, xdec=(0, 1, 1), nslots=3)
    for f, fr in enumerate(frames):
        ctx.upload_planes(f, fr)
    ctx.forward_haar()
    ctx.inverse_haar()
    for f, fr in enumerate(frames):
        for pli in range(3):
            assert np.array_equal(ctx.download_recon(f, pli), fr[pli]), (f, pli)
    for pli in range(3):
        pw, ph, sb = fw >> (pli > 0), fh >> (pli > 0), 32 >> (pli > 0)
        d = np.zeros((ph, pw), np.int32)
        o.orc_haar_forward_plane(p32(d), pu8(frames[2][pli]), pw, ph, sb)
        assert np.array_equal(ctx.download_coeffs(2, pli), d), pli
    ctx.close()
