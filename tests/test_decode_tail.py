"""End-to-end check of the decoder's whole pixel-domain stage against a REAL decode:
the reference encoder makes a keyframe, the reference decoder decodes it
(oracle/_ref/dec_probe.so) and hands over what its reconstruction consumed -
dequantised coefficients, block sizes, skip maps, dering flags, quantizers.  From
those the oracle (here) and the device (tests/test_gpu_decode_tail.py) must
reproduce the decoder's output picture bit-exactly:
  iDCT -> split post-filters -> frame post-filter -> deringing -> bilinear
  smoothing -> 8-bit clamp   (src/decode.c:1010-1155)."""
import ctypes
import os

import numpy as np
import pytest

from testlib import ORACLE_DIR, oracle, p32, pu8, synth_plane

PROBE = os.path.join(ORACLE_DIR, '_ref', 'dec_probe.so')
pytestmark = pytest.mark.skipif(not os.path.exists(PROBE), reason='oracle/_ref not built')


def reference_decode(w, h, quant, masking, seed):
    pr = ctypes.CDLL(PROBE)
    cw, ch = (w + 1)//2, (h + 1)//2
    frame = np.concatenate([synth_plane(w, h, seed).ravel(), synth_plane(cw, ch, seed, 1).ravel(),
                            synth_plane(cw, ch, seed + 1, 1).ravel()])
    fw, fh = (w + 63)//64*64, (h + 63)//64*64
    d = [np.zeros((fh >> x, fw >> x), np.int32) for x in (0, 1, 1)]
    bsize = np.zeros((fh//8, fw//8), np.uint8)
    bskip = [np.zeros((fh//4, fw//4), np.uint8) for _ in range(3)]
    flags = np.zeros((fh//32, fw//32), np.uint8)
    q = (ctypes.c_int*3)()
    out = [np.zeros((h, w), np.uint8), np.zeros((ch, cw), np.uint8), np.zeros((ch, cw), np.uint8)]
    fwo = ctypes.c_int(); fho = ctypes.c_int()
    rc = pr.probe_decode_dump(w, h, quant, masking, pu8(frame), ctypes.byref(fwo), ctypes.byref(fho),
                              p32(d[0]), p32(d[1]), p32(d[2]), pu8(bsize), pu8(bskip[0]),
                              pu8(bskip[1]), pu8(bskip[2]), pu8(flags), q, pu8(out[0]), pu8(out[1]),
                              pu8(out[2]))
    assert rc == 0 and (fwo.value, fho.value) == (fw, fh)
    return dict(w=w, h=h, fw=fw, fh=fh, d=d, bsize=bsize, bskip=bskip, flags=flags,
                q=[q[0], q[1], q[2]], out=out)


def oracle_tail(r):
    o = oracle()
    fw, fh = r['fw'], r['fh']
    xdec = (ctypes.c_int*3)(0, 1, 1)
    c = []
    for pli in range(3):
        dec = 1 if pli else 0
        pw, ph = fw >> dec, fh >> dec
        cp = np.zeros((ph, pw), np.int32)
        dummy = np.zeros((ph, pw), np.uint8)
        o.orc_inverse_plane(pu8(dummy), pw, p32(cp), p32(r['d'][pli]), fw//32, fh//32, dec,
                            pu8(r['bsize']), fw//8, r['w'], r['h'])
        c.append(cp)
    rec = [np.zeros_like(x, dtype=np.uint8) for x in c]
    thr = (ctypes.c_int*3)(*[int(1.0*pow(q, 0.84182)) for q in r['q']])
    qq = (ctypes.c_int*3)(*r['q'])
    I32PP = ctypes.POINTER(ctypes.c_int32)*3
    U8PP = ctypes.POINTER(ctypes.c_uint8)*3
    o.orc_decode_tail(I32PP(*[p32(x) for x in c]), U8PP(*[pu8(x) for x in rec]), 3, fw, fh, xdec,
                      pu8(r['flags']), U8PP(*[pu8(b) for b in r['bskip']]), fw//4, pu8(r['bsize']),
                      fw//8, thr, qq, 1)
    return rec


@pytest.mark.parametrize('quant,masking,seed', ((20, 1, 3), (40, 0, 4), (8, 1, 5)))
def test_oracle_reproduces_reference_decoder_output(quant, masking, seed):
    r = reference_decode(176, 112, quant, masking, seed)
    rec = oracle_tail(r)
    for pli in range(3):
        hh, ww = r['out'][pli].shape
        assert np.array_equal(rec[pli][:hh, :ww], r['out'][pli]), pli


def test_dering_and_smoothing_units_match_reference():
    """od_dering (luma with direction search, chroma reusing luma directions, frame
    edges, skipped neighbourhoods) and od_bilinear_smooth against the reference
    functions on seeded data."""
    o = oracle()
    ep = ctypes.CDLL(os.path.join(ORACLE_DIR, '_ref', 'enc_probe.so'))
    r = ctypes.CDLL(os.path.join(ORACLE_DIR, '_ref', 'libdaala_ref.so'))
    I16P = ctypes.POINTER(ctypes.c_int16)
    rng = np.random.default_rng(55)
    nhsb, nvsb = 3, 3
    for trial in range(60):
        q = int(rng.choice([30, 90, 217, 421, 900]))
        sbx, sby = int(rng.integers(0, nhsb)), int(rng.integers(0, nvsb))
        dirs = None
        for pli in (0, 1):
            xdec = pli
            ln = 5 - xdec
            n = 1 << ln
            w = (nhsb*32) >> xdec
            plane = (rng.normal(0, 300, size=((nvsb*32) >> xdec, w)) +
                     np.add.outer(np.arange((nvsb*32) >> xdec)*9, np.arange(w)*(-5))).astype(np.int16)
            sstride = (nhsb*32)//4
            bskip = (rng.random(((nvsb*32)//4, sstride)) < .3).astype(np.uint8)
            if trial % 5 == 0:
                bskip[:] = 1
            xs = plane[(sby << ln):, (sbx << ln):]
            x = np.ascontiguousarray(plane)
            off = (sby << ln)*w + (sbx << ln)
            boff = (sby << (3 - xdec))*sstride + (sbx << (3 - xdec))
            ya = np.zeros((n, n), np.int16); yb = np.zeros((n, n), np.int16)
            da = (ctypes.c_int*16)(*(dirs if dirs is not None else [0]*16))
            db = (ctypes.c_int*16)(*(dirs if dirs is not None else [0]*16))
            xp = ctypes.cast(ctypes.addressof(x.ctypes.data_as(I16P).contents) + 2*off, I16P)
            bp = ctypes.cast(bskip.ctypes.data + boff, ctypes.POINTER(ctypes.c_uint8))
            assert ep.probe_dering(ya.ctypes.data_as(I16P), n, xp, w, ln, sbx, sby, nhsb, nvsb, q,
                                   xdec, da, pli, bp, sstride) == 0
            thr = int(1.0*pow(q, 0.84182))
            o.orc_dering_sb(yb.ctypes.data_as(I16P), n, xp, w, ln, sbx, sby, nhsb, nvsb, thr, xdec,
                            db, pli, bp, sstride)
            assert list(da) == list(db)
            assert np.array_equal(ya, yb), (trial, pli)
            dirs = list(da)
    for trial in range(200):
        ln = int(rng.choice([4, 5]))
        n = 1 << ln
        pli = int(rng.integers(0, 3))
        q = int(rng.choice([30, 90, 217, 421, 2000]))
        base = np.add.outer(np.arange(n)*rng.integers(-40, 40), np.arange(n)*rng.integers(-40, 40))
        x = (base + rng.normal(0, rng.choice([2, 30, 400]), size=(n, n))).astype(np.int32)
        a, b = x.copy(), x.copy()
        r.od_bilinear_smooth(p32(a), ln, n, q, pli)
        o.orc_bilinear_smooth(p32(b), ln, n, q, pli)
        assert np.array_equal(a, b)
