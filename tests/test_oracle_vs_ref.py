"""Pins the CPU oracle (oracle/daala_oracle.c) bit-exactly to the REAL reference
compiled from /root/reference (oracle/_ref).  Skipped where oracle/_ref is not
built; test_oracle_golden.py pins the same functions through committed vectors."""
import ctypes

import numpy as np
import pytest

from testlib import (c_int, have_ref, oracle, p16, p32, pf64, pu8, random_bsize_map,
                     ref, synth_plane)

pytestmark = pytest.mark.skipif(not have_ref(), reason='oracle/_ref not built')

SIZES = (4, 8, 16, 32)


def rand_blocks(rng, n, count, amp):
    return rng.integers(-amp, amp + 1, size=(count, n, n), dtype=np.int32)


@pytest.mark.parametrize('n', SIZES)
def test_dct_2d_matches_reference(n):
    o, r = oracle(), ref()
    rng = np.random.default_rng(n)
    for amp in (255, 4096, 40000):
        for x in rand_blocks(rng, n, 300, amp):
            x = np.ascontiguousarray(x)
            yo = np.zeros_like(x); yr = np.zeros_like(x)
            o.orc_fdct_2d(n, p32(yo), n, p32(x), n)
            getattr(r, 'od_bin_fdct%dx%d' % (n, n))(p32(yr), n, p32(x), n)
            assert np.array_equal(yo, yr)
            xo = np.zeros_like(x); xr = np.zeros_like(x)
            o.orc_idct_2d(n, p32(xo), n, p32(x), n)          # arbitrary input
            getattr(r, 'od_bin_idct%dx%d' % (n, n))(p32(xr), n, p32(x), n)
            assert np.array_equal(xo, xr)
            o.orc_idct_2d(n, p32(xo), n, p32(yo), n)
            assert np.array_equal(xo, x)                      # perfect reconstruction


def test_dct_in_place_aliasing():
    # the reference vtable contract allows out == in (src/dct.h:61-62)
    o = oracle()
    rng = np.random.default_rng(3)
    for n in SIZES:
        x = np.ascontiguousarray(rand_blocks(rng, n, 1, 3000)[0])
        y = np.zeros_like(x)
        o.orc_fdct_2d(n, p32(y), n, p32(x), n)
        z = x.copy()
        o.orc_fdct_2d(n, p32(z), n, p32(z), n)
        assert np.array_equal(y, z)


def test_filter4_matches_reference():
    o, r = oracle(), ref()
    rng = np.random.default_rng(5)
    v = rng.integers(-70000, 70001, size=(20000, 4), dtype=np.int32)
    v[:4000] = rng.integers(-3, 4, size=(4000, 4))
    for x in v:
        x = np.ascontiguousarray(x)
        a = np.zeros(4, np.int32); b = np.zeros(4, np.int32)
        o.orc_pre_filter4(p32(a), p32(x)); r.od_pre_filter4(p32(b), p32(x))
        assert np.array_equal(a, b)
        o.orc_post_filter4(p32(a), p32(x)); r.od_post_filter4(p32(b), p32(x))
        assert np.array_equal(a, b)


@pytest.mark.parametrize('dec', (0, 1))
def test_frame_and_split_lapping(dec):
    o, r = oracle(), ref()
    rng = np.random.default_rng(7 + dec)
    nhsb, nvsb = 3, 2
    w, h = (nhsb*32) >> dec, (nvsb*32) >> dec
    c = rng.integers(-2048, 2048, size=(h, w), dtype=np.int32)
    a, b = c.copy(), c.copy()
    o.orc_prefilter_frame_sbs(p32(a), w, nhsb, nvsb, dec)
    r.od_apply_prefilter_frame_sbs(p32(b), w, nhsb, nvsb, dec, dec)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    o.orc_postfilter_frame_sbs(p32(a), w, nhsb, nvsb, dec)
    r.od_apply_postfilter_frame_sbs(p32(b), w, nhsb, nvsb, dec, dec, 0, None, 0)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    for bs in (1, 2, 3):
        n = 4 << bs
        for hf in (0, 1):
            for vf in (0, 1):
                blk = np.ascontiguousarray(c[:n, :n]); a, b = blk.copy(), blk.copy()
                o.orc_prefilter_split(p32(a), n, n, hf, vf)
                r.od_prefilter_split(p32(b), n, bs, 0, hf, vf)
                assert np.array_equal(a, b)
                o.orc_postfilter_split(p32(a), n, n, hf, vf)
                r.od_postfilter_split(p32(b), n, bs, 0, 0, None, 0, hf, vf)
                assert np.array_equal(a, b) and np.array_equal(a, blk)


def test_tf_and_cfl_resample():
    o, r = oracle(), ref()
    rng = np.random.default_rng(11)
    for n in (4, 8, 16):
        src = rng.integers(-5000, 5001, size=(2*n, 2*n), dtype=np.int32)
        for fn, args in (('tf_up_h_lp', (n, n)), ('tf_up_v_lp', (n, n)),
                         ('tf_up_hv_lp', (n, n, n))):
            a = np.zeros((2*n, 2*n), np.int32); b = a.copy()
            getattr(o, 'orc_' + fn)(p32(a), 2*n, p32(src), 2*n, *args)
            getattr(r, 'od_' + fn)(p32(b), 2*n, p32(src), 2*n, *args)
            assert np.array_equal(a, b), fn
        a = np.zeros((2*n, 2*n), np.int32); b = a.copy()
        o.orc_tf_up_hv(p32(a), 2*n, p32(src), 2*n, n)
        r.od_tf_up_hv(p32(b), 2*n, p32(src), 2*n, n)
        assert np.array_equal(a, b)
        a2 = np.zeros_like(a); b2 = a2.copy()
        o.orc_tf_down_hv(p32(a2), 2*n, p32(a), 2*n, 2*n)
        r.od_tf_down_hv(p32(b2), 2*n, p32(b), 2*n, 2*n)
        assert np.array_equal(a2, b2) and np.array_equal(a2, src)
    # od_resample_luma_coeffs (src/intra.c:72): 4:2:0, luma 4x4 group and copy case
    luma = rng.integers(-3000, 3001, size=(32, 64), dtype=np.int32)
    for bs, cbs in ((0, 0), (0, 1), (1, 2), (2, 3), (3, 3)):
        n = 4 << bs
        a = np.zeros((n, n), np.int32); b = a.copy()
        o.orc_resample_luma_coeffs(p32(a), n, p32(luma), 64, 1, 1, bs, cbs)
        r.od_resample_luma_coeffs(p32(b), n, p32(luma), 64, 1, 1, bs, cbs)
        assert np.array_equal(a, b)


@pytest.mark.parametrize('n', SIZES)
def test_coding_order(n):
    o, r = oracle(), ref()
    rng = np.random.default_rng(13)
    src = rng.integers(-9999, 9999, size=(n, n), dtype=np.int32)
    a = np.full(n*n, -1, np.int32); b = a.copy()
    o.orc_raster_to_coding_order(p32(a), n, p32(src), n)
    r.od_raster_to_coding_order(p32(b), n, p32(src), n)
    assert np.array_equal(a, b)
    ra = np.full((n, n), -5, np.int32); rb = ra.copy()
    o.orc_coding_order_to_raster(p32(ra), n, p32(a), n)
    r.od_coding_order_to_raster(p32(rb), n, p32(b), n)
    assert np.array_equal(ra, rb)


def test_haar():
    o, r = oracle(), ref()
    rng = np.random.default_rng(17)
    for ln in (2, 3, 4, 5):
        n = 1 << ln
        x = rng.integers(-255, 256, size=(n, n), dtype=np.int32)
        a = np.zeros_like(x); b = a.copy()
        o.orc_haar(p32(a), n, p32(x), n, ln); r.od_haar(p32(b), n, p32(x), n, ln)
        assert np.array_equal(a, b)
        xa = np.zeros_like(x); xb = xa.copy()
        o.orc_haar_inv(p32(xa), n, p32(a), n, ln); r.od_haar_inv(p32(xb), n, p32(b), n, ln)
        assert np.array_equal(xa, xb) and np.array_equal(xa, x)


def test_hv_intra_pred():
    o, r = oracle(), ref()
    rng = np.random.default_rng(19)
    w = 96
    d = rng.integers(-500, 501, size=(96, w), dtype=np.int32)
    bstride = 16
    for trial in range(200):
        bsz = rng.integers(0, 4, size=(16, bstride)).astype(np.uint8)
        bs = int(rng.integers(0, 4))
        nb = 1 << bs
        bx = int(rng.integers(0, 96//4//nb))*nb
        by = int(rng.integers(0, 96//4//nb))*nb
        n = 4 << bs
        a = np.full(n*n, 77, np.int32); b = a.copy()
        o.orc_hv_intra_pred(p32(a), p32(d), w, bx, by, pu8(bsz), bstride, bs)
        r.od_hv_intra_pred(p32(b), p32(d), w, bx, by, pu8(bsz), bstride, bs)
        assert np.array_equal(a, b)


@pytest.mark.parametrize('pli', (0, 1))
@pytest.mark.parametrize('keyframe', (0, 1))
def test_forward_plane_matches_reference_compute_dcts(pli, keyframe):
    """orc_forward_plane == od_ref_buf_to_coeff + od_apply_prefilter_frame_sbs +
    od_compute_dcts of the reference on a random valid block-size map; the
    picture is not a multiple of 32 so the hfilter/vfilter gating is hit."""
    o, pr = oracle(), ref('enc_probe')
    pic_w, pic_h = 150, 100           # pads to 192 x 128
    fw, fh = 192, 128
    nhsb, nvsb = fw//32, fh//32
    dec = 1 if pli else 0
    w, h = fw >> dec, fh >> dec
    pix = synth_plane(w, h, seed=3 + pli, chroma=dec)
    for seed in range(3):
        bmap = random_bsize_map(nhsb, nvsb, seed)
        c_r = np.zeros((h, w), np.int32); d_r = np.zeros((h, w), np.int32)
        assert pr.probe_forward_plane(pic_w, pic_h, pli, keyframe, pu8(pix), pu8(bmap),
                                      p32(c_r), p32(d_r)) == 0
        c_o = np.zeros((h, w), np.int32); d_o = np.zeros((h, w), np.int32)
        o.orc_forward_plane(p32(c_o), p32(d_o), pu8(pix), w, nhsb, nvsb, dec,
                            pu8(bmap), nhsb*4, pic_w, pic_h, keyframe)
        assert np.array_equal(c_o, c_r)
        assert np.array_equal(d_o, d_r)


def test_pyramid_levels_equal_uniform_maps_and_inverse_roundtrip():
    o = oracle()
    pic_w, pic_h, fw, fh = 150, 100, 192, 128
    nhsb, nvsb = fw//32, fh//32
    for dec in (0, 1):
        w, h = fw >> dec, fh >> dec
        pix = synth_plane(w, h, seed=9, chroma=dec)
        nlev = 4 - dec
        lev = [np.zeros((h, w), np.int32) for _ in range(nlev)]
        arr = (ctypes.POINTER(ctypes.c_int32)*nlev)(*[p32(a) for a in lev])
        c = np.zeros((h, w), np.int32)
        o.orc_forward_pyramid_plane(p32(c), arr, nlev, pu8(pix), w, nhsb, nvsb, dec,
                                    pic_w, pic_h)
        for k in range(nlev):
            bs = 3 - k
            bmap = np.full((nvsb*4, nhsb*4), bs, np.uint8)
            c2 = np.zeros((h, w), np.int32); d2 = np.zeros((h, w), np.int32)
            o.orc_forward_plane(p32(c2), p32(d2), pu8(pix), w, nhsb, nvsb, dec,
                                pu8(bmap), nhsb*4, pic_w, pic_h, 0)
            assert np.array_equal(lev[k], d2), (dec, k)
        # exact invertibility of the whole path on a mixed map
        bmap = random_bsize_map(nhsb, nvsb, 5)
        c2 = np.zeros((h, w), np.int32); d2 = np.zeros((h, w), np.int32)
        o.orc_forward_plane(p32(c2), p32(d2), pu8(pix), w, nhsb, nvsb, dec, pu8(bmap),
                            nhsb*4, pic_w, pic_h, 0)
        out = np.zeros((h, w), np.uint8); c3 = np.zeros((h, w), np.int32)
        o.orc_inverse_plane(pu8(out), w, p32(c3), p32(d2), nhsb, nvsb, dec, pu8(bmap),
                            nhsb*4, pic_w, pic_h)
        assert np.array_equal(out, pix)


# ---------------------------------------------------------------------------
# PVQ
def make_qm(rng, n):
    qm = rng.integers(9000, 32768, size=n).astype(np.int16)
    qm_inv = np.floor(.5 + 32768.*4096./qm.astype(np.float64)).astype(np.int16)
    return qm, qm_inv


def test_pvq_scalar_helpers():
    o, r = oracle(), ref()
    rng = np.random.default_rng(23)
    for _ in range(3000):
        qcg = float(rng.uniform(0, 40)) if rng.random() < .7 else float(rng.integers(0, 9))
        beta = 1.5 if rng.random() < .5 else 1.0
        n = int(rng.choice([7, 8, 14, 15, 16, 31, 32, 127, 128]))
        assert o.orc_pvq_compute_max_theta(qcg, beta) == r.od_pvq_compute_max_theta(qcg, beta)
        it = int(rng.integers(0, 12)); th = float(rng.uniform(0, 1.5))
        for noref in (0, 1):
            for nodesync in (0, 1):
                assert o.orc_pvq_compute_k(qcg, it, th, noref, n, beta, nodesync) == \
                    r.od_pvq_compute_k(qcg, it, th, noref, n, beta, nodesync)
        ts = int(rng.integers(0, 20))
        assert o.orc_pvq_compute_theta(it, ts) == r.od_pvq_compute_theta(it, ts)
        q0 = int(rng.integers(1, 400))
        assert o.orc_gain_expand(qcg, q0, beta) == r.od_gain_expand(qcg, q0, beta)


def test_pvq_gain_householder_synthesis():
    o, r = oracle(), ref()
    rng = np.random.default_rng(29)
    for _ in range(1500):
        n = int(rng.choice([8, 15, 16, 32, 128]))
        amp = int(rng.choice([3, 40, 600, 8000]))
        x = rng.integers(-amp, amp + 1, size=n, dtype=np.int32)
        qm, qm_inv = make_qm(rng, n)
        q0 = int(rng.integers(1, 300))
        beta = 1.5 if rng.random() < .5 else 1.0
        ga = ctypes.c_double(); gb = ctypes.c_double()
        ca = o.orc_pvq_compute_gain(p32(x), n, q0, ctypes.byref(ga), beta, p16(qm))
        cb = r.od_pvq_compute_gain(p32(x), n, q0, ctypes.byref(gb), beta, p16(qm))
        assert ca == cb and ga.value == gb.value
        rr = rng.normal(0, 50, size=n)
        ra, rb = rr.copy(), rr.copy()
        sa = c_int(); sb = c_int()
        ma = o.orc_compute_householder(pf64(ra), n, float(np.sqrt((rr*rr).sum())), ctypes.byref(sa))
        mb = r.od_compute_householder(pf64(rb), n, float(np.sqrt((rr*rr).sum())), ctypes.byref(sb))
        assert ma == mb and sa.value == sb.value and np.array_equal(ra, rb)
        xa = rng.normal(0, 30, size=n); xb = xa.copy()
        o.orc_apply_householder(pf64(xa), pf64(ra), n)
        r.od_apply_householder(pf64(xb), pf64(rb), n)
        assert np.array_equal(xa, xb)
        for noref in (0, 1):
            nn = n - (0 if noref else 1)
            y = rng.integers(-3, 4, size=n, dtype=np.int32)
            y[nn:] = 0
            g = float(rng.uniform(1, 5000)); th = float(rng.uniform(0, 1.5))
            outa = np.zeros(n, np.int32); outb = outa.copy()
            o.orc_pvq_synthesis_partial(p32(outa), p32(y), pf64(ra), n, noref, g, th, ma,
                                        sa.value, p16(qm_inv))
            r.od_pvq_synthesis_partial(p32(outb), p32(y), pf64(rb), n, noref, g, th, mb,
                                       sb.value, p16(qm_inv))
            assert np.array_equal(outa, outb)


def test_pvq_search_matches_reference_static():
    o, pr = oracle(), ref('pvq_probe')
    rng = np.random.default_rng(31)
    for _ in range(4000):
        n = int(rng.choice([7, 8, 14, 15, 16, 31, 32, 127, 128]))
        kind = rng.integers(0, 4)
        if kind == 0:
            x = rng.normal(0, 1, size=n)
        elif kind == 1:
            x = rng.laplace(0, 1, size=n)*np.exp(-np.arange(n)/8.)
        elif kind == 2:     # many exact ties
            x = rng.integers(-2, 3, size=n).astype(np.float64)
        else:
            x = rng.integers(-300, 301, size=n)*rng.integers(9000, 32768, size=n)*(1./32767)
        k = int(rng.choice([1, 2, 3, 4, 7, 10, 17, 40, 130, 300]))
        g2 = float(rng.uniform(0.1, 60))
        ya = np.zeros(n, np.int32); yb = ya.copy()
        ca = o.orc_pvq_search_rdo_double(pf64(x), n, k, p32(ya), g2)
        cb = pr.probe_pvq_search_rdo_double(pf64(x), n, k, p32(yb), g2)
        assert np.array_equal(ya, yb) and ca == cb
        assert np.abs(ya).sum() == k or not x.any()


def test_noref_candidates_reproduce_pvq_theta_decision():
    """State-free candidates + the reference's own rate term reproduce the full
    pvq_theta() outcome (keyframe luma, null reference => no-reference branch
    only, src/pvq_encoder.c:452-481): qg, k, y and the synthesised output."""
    o, pr = oracle(), ref('pvq_probe')
    rng = np.random.default_rng(37)
    lam = .147
    for trial in range(1500):
        bs = int(rng.integers(0, 4))
        n = int(rng.choice({0: [15], 1: [15, 8, 32], 2: [15, 8, 32, 128], 3: [15, 8, 32, 128]}[bs]))
        amp = int(rng.choice([2, 20, 200, 3000]))
        x0 = (rng.laplace(0, amp, size=n)).astype(np.int32)
        qm, qm_inv = make_qm(rng, n)
        q0 = int(rng.integers(2, 200))
        beta = 1.5 if (bs > 0 and rng.random() < .5) else 1.0
        r0 = np.zeros(n, np.int32)
        out_r = np.zeros(n, np.int32); y_r = np.zeros(n, np.int32)
        it = c_int(); mt = c_int(); vk = c_int(); sd = ctypes.c_double(0)
        qg_r = pr.probe_pvq_theta(p32(out_r), p32(x0.copy()), p32(r0), n, q0, p32(y_r),
                                  ctypes.byref(it), ctypes.byref(mt), ctypes.byref(vk), beta,
                                  ctypes.byref(sd), 1, 1, 0, bs, p16(qm), p16(qm_inv))
        cg = ctypes.c_double(); g = ctypes.c_double()
        qg = np.zeros(2, np.int32); k = np.zeros(2, np.int32)
        cd = np.zeros(2); dist = np.zeros(2); y = np.zeros((2, n), np.int32)
        nc = o.orc_pvq_noref_candidates(p32(x0), n, q0, beta, p16(qm), 1, ctypes.byref(cg),
                                        ctypes.byref(g), qg.ctypes.data_as(ctypes.POINTER(c_int)),
                                        k.ctypes.data_as(ctypes.POINTER(c_int)), pf64(cd),
                                        pf64(dist), p32(y))
        best_cost = 1.4*cg.value*cg.value
        best = -1
        for c in range(nc):
            rate = pr.probe_pvq_rate_reset(int(qg[c]), 0, -1, 0, p32(y[c]), int(k[c]), n, 1, 0, bs)
            cost = dist[c] + lam*rate
            if cost <= best_cost:
                best_cost = cost
                best = c
        if best < 0:
            assert qg_r == 0 and not out_r.any()
        else:
            assert qg_r == qg[best] and vk.value == k[best] and it.value == -1
            assert np.array_equal(y_r, y[best])
            gexp = o.orc_gain_expand(float(qg[best]), q0, beta)
            out_o = np.zeros(n, np.int32)
            o.orc_pvq_synthesis_partial(p32(out_o), p32(y[best]), pf64(np.zeros(n)), n, 1,
                                        gexp, 0., 0, 1, p16(qm_inv))
            assert np.array_equal(out_o, out_r)


@pytest.mark.parametrize('is_keyframe,pli', ((1, 0), (1, 1), (0, 0), (0, 2)))
def test_full_pvq_theta_decision_with_reference(is_keyframe, pli):
    """Complete pvq_theta (with-reference theta/gain search + no-reference search,
    skip logic, synthesis) reproduced from the oracle's state-free candidates and
    the reference's own od_pvq_rate: return code, itheta, max_theta, K, pulses and
    synthesised coefficients must all equal the reference's."""
    from testlib import ThetaOut, decide_pvq_theta
    o, pr = oracle(), ref('pvq_probe')
    o.orc_pvq_theta_candidates.argtypes = [ctypes.POINTER(ctypes.c_int32)]*2 + [
        c_int, c_int, ctypes.c_double, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_int16),
        ctypes.POINTER(ThetaOut), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    rng = np.random.default_rng(41 + 2*is_keyframe + pli)
    nsearched = 0
    for trial in range(1200):
        bs = int(rng.integers(0, 4))
        n = int(rng.choice({0: [15], 1: [15, 8, 32], 2: [15, 8, 32, 128], 3: [15, 8, 32, 128]}[bs]))
        amp = int(rng.choice([3, 30, 300, 3000]))
        x0 = (rng.laplace(0, amp, size=n)).astype(np.int32)
        kind = rng.integers(0, 4)
        if kind == 0:
            r0 = (x0*rng.uniform(.5, 1.5) + rng.laplace(0, amp/3. + 1, size=n)).astype(np.int32)
        elif kind == 1:
            r0 = (rng.laplace(0, amp, size=n)).astype(np.int32)
        elif kind == 2:
            r0 = x0.copy()
        else:
            r0 = np.zeros(n, np.int32)
        qm, qm_inv = make_qm(rng, n)
        q0 = int(rng.integers(2, 200))
        beta = 1.5 if (bs > 0 and pli == 0 and rng.random() < .5) else 1.0
        out_r = np.zeros(n, np.int32); y_r = np.zeros(n, np.int32)
        it = c_int(); mt = c_int(); vk = c_int(); sd = ctypes.c_double(0)
        ret_r = pr.probe_pvq_theta(p32(out_r), p32(x0.copy()), p32(r0.copy()), n, q0, p32(y_r),
                                   ctypes.byref(it), ctypes.byref(mt), ctypes.byref(vk), beta,
                                   ctypes.byref(sd), 1, is_keyframe, pli, bs, p16(qm), p16(qm_inv))
        t = ThetaOut()
        y_ref = np.zeros((12, n), np.int32); y_nr = np.zeros((2, n), np.int32)
        o.orc_pvq_theta_candidates(p32(x0), p32(r0), n, q0, beta, 1, is_keyframe, pli, p16(qm),
                                   ctypes.byref(t), p32(y_ref), p32(y_nr))
        nsearched += t.theta_searched

        def rate(qg, icgr, theta, ts, y, k):
            yy = np.ascontiguousarray(y) if y is not None else np.zeros(n, np.int32)
            return pr.probe_pvq_rate_reset(int(qg), int(icgr), int(theta), int(ts), p32(yy), int(k),
                                           n, is_keyframe, pli, bs)

        ret, itheta, max_theta, k, y, out = decide_pvq_theta(
            o, rate, t, y_ref, y_nr, x0, r0, n, q0, beta, is_keyframe, pli, qm, qm_inv)
        assert (ret, itheta, max_theta, k) == (ret_r, it.value, mt.value, vk.value), trial
        nn = n if itheta == -1 else n - 1
        assert np.array_equal(y[:nn], y_r[:nn]), trial
        assert np.array_equal(out, out_r), trial
        # skip_diff accumulates skip_dist - best_dist (src/pvq_encoder.c:505), bit for bit
        assert sd.value == 0. + (t.skip_dist - decide_pvq_theta.best_dist), trial
    assert nsearched > 200


def test_obmc_block_prediction_matches_reference():
    """F3: the oracle's OBMC of one prediction block (orc_mc_predict = four
    od_mc_predict1fmv8_c predictions + od_mc_blend_full8_c / od_mc_blend_full_split8_c) against
    the reference's od_mc_predict run through a live context's vtable: every block size, full
    and fractional vectors, one and several references, every (oc, s)."""
    o = oracle()
    p = ref('enc_probe')
    rng = np.random.default_rng(5)
    H, W, pad = 224, 256, 48
    refs = [rng.integers(0, 256, size=(H, W), dtype=np.uint8) for _ in range(3)]
    for r in refs[1:]:
        r[::3] = 255 - r[::3]//2                       # saturating content
    U8P, I32P = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int32)

    def at(k, x, y):
        return ctypes.cast(refs[k].ctypes.data + y*W + x, U8P)

    for trial in range(400):
        lx = int(rng.integers(2, 7))
        ly = lx if trial % 5 else int(rng.integers(2, 7))
        n, m = 1 << lx, 1 << ly
        x0, y0 = int(rng.integers(pad, W - pad - n)), int(rng.integers(pad, H - pad - m))
        mvx = rng.integers(-8*(pad - 8), 8*(pad - 8), size=4).astype(np.int32)
        mvy = rng.integers(-8*(pad - 8), 8*(pad - 8), size=4).astype(np.int32)
        if trial % 3 == 0:
            mvx &= ~7
        if trial % 4 == 0:
            mvy &= ~7
        if trial % 7 == 0:
            mvx[:] = mvx[0]; mvy[:] = mvy[0]
        ks = [0, 0, 0, 0] if trial % 2 else [int(v) for v in rng.integers(0, 3, size=4)]
        if lx != ly:
            mvx |= 1                                    # the full-pel copy asserts square blocks
        oc, s = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        want = np.zeros((m, n), np.uint8)
        got = np.zeros((m, n), np.uint8)
        assert p.probe_mc_predict(pu8(want), n, at(ks[0], x0, y0), at(ks[1], x0, y0), at(ks[2], x0, y0),
                                  at(ks[3], x0, y0), W, mvx.ctypes.data_as(I32P),
                                  mvy.ctypes.data_as(I32P), oc, s, lx, ly) == 0
        srcs = (U8P*4)(*[at(ks[k], x0, y0) for k in range(4)])
        o.orc_mc_predict(pu8(got), n, srcs, W, mvx.ctypes.data_as(I32P), mvy.ctypes.data_as(I32P),
                         oc, s, lx, ly)
        assert np.array_equal(got, want), (trial, lx, ly, oc, s)


def test_sad_satd_match_reference():
    """F3: orc_mc_sad8 / orc_mc_satd8 against the reference's od_mc_compute_sad8_NxN_c and
    od_mc_compute_satd8_NxN_c (src/mcenc.c:1349-1372, :1562-1612), distinct strides, extreme and
    near-identical content."""
    o = oracle()
    r = ref()
    rng = np.random.default_rng(77)
    U8P = ctypes.POINTER(ctypes.c_uint8)
    o.orc_mc_sad8.restype = o.orc_mc_satd8.restype = ctypes.c_int32
    for trial in range(600):
        lg = int(rng.integers(2, 7))
        n = 1 << lg
        ss, rs = n + int(rng.integers(0, 9)), n + int(rng.integers(0, 9))
        a = rng.integers(0, 256, size=(n, ss), dtype=np.uint8)
        if trial % 3 == 0:
            b = np.clip(a[:, :n].astype(np.int32) + rng.integers(-3, 4, size=(n, n)), 0, 255).astype(np.uint8)
        elif trial % 3 == 1:
            b = rng.integers(0, 256, size=(n, n), dtype=np.uint8)
        else:
            a[:] = 255*(rng.integers(0, 2, size=a.shape))
            b = (255 - a[:, :n]).astype(np.uint8)
        bb = np.zeros((n, rs), np.uint8)
        bb[:, :n] = b
        for kind, f in (('sad', o.orc_mc_sad8), ('satd', o.orc_mc_satd8)):
            g = getattr(r, 'od_mc_compute_%s8_%dx%d_c' % (kind, n, n))
            g.restype = ctypes.c_int32
            want = g(a.ctypes.data_as(U8P), ss, bb.ctypes.data_as(U8P), rs)
            assert f(a.ctypes.data_as(U8P), ss, bb.ctypes.data_as(U8P), rs, lg) == want, (trial, kind, n)


def test_mvest_calc_sads_matches_reference():
    """F3, second half: the oracle's restatement of od_mv_est_calc_sads against the reference's own
    function (oracle/ref_probe/mcenc_probe.c #includes src/mcenc.c) on the motion estimation
    context of a live encoder - another stream than the committed fixture (picture size that is
    not a multiple of the block sizes: the SAD's clip against the picture is exercised)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from configs_round import frames_of
    import daala_amd.hipenc as H
    from testlib import mvest_items, mvest_oracle_sads, mvest_split
    mp = ref('mcenc_probe')
    if mp is None:
        pytest.skip('oracle/_ref/mcenc_probe.so not built')
    o = oracle()
    U8P, I32P = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int32)
    w, h, nf = 150, 100, 3
    buf = H.pack_frames(frames_of(w, h, nf, 5, step=(1, 2)), w, h)
    assert mp.probe_mvest_open(w, h, nf, 30, 30, pu8(buf)) == 0
    dims = np.zeros(19, np.int32)
    assert mp.probe_mvest_dims(p32(dims)) == 0
    nh, nv, nimg, fw, fh = int(dims[0]), int(dims[1]), int(dims[4]), int(dims[5]), int(dims[6])
    g = {'dims': dims, 'pic': np.array([w, h], np.int32)}
    for k in ('gmvx', 'gmvy', 'gref'):
        g[k] = np.zeros((nv + 1, nh + 1), np.int32)
    for p in range(3):
        g['refs%d' % p] = np.zeros((nimg, int(dims[8 + 4*p]), int(dims[7 + 4*p])), np.uint8)
        g['src%d' % p] = np.zeros((fh >> (p > 0), fw >> (p > 0)), np.uint8)
    sad = [np.zeros((nv >> l, nh >> l, 4), np.int32) for l in range(3)]
    assert mp.probe_mvest_get(p32(g['gmvx']), p32(g['gmvy']), p32(g['gref']),
                              (U8P*3)(*[pu8(g['refs%d' % p]) for p in range(3)]),
                              (U8P*3)(*[pu8(g['src%d' % p]) for p in range(3)]),
                              (I32P*3)(*[p32(a) for a in sad])) == 0
    mp.probe_mvest_close()
    items, sizes, smax = mvest_items(o, g)
    got = mvest_split(mvest_oracle_sads(o, g, items), sizes, smax, dims)
    assert np.array_equal(got[1], sad[1]) and np.array_equal(got[2], sad[2])
