"""CPU suite: pins the oracle restatement (oracle/daala_oracle.c) to the golden
vectors committed under tests/golden/ (generated from the real reference by
tools/gen_golden.py).  Needs neither /root/reference nor a GPU."""
import ctypes
import os

import numpy as np
import pytest

from testlib import GOLDEN, c_int, oracle, p16, p32, pf64, pu8


def load(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.mark.parametrize('n', (4, 8, 16, 32))
def test_dct_golden(n):
    o = oracle()
    g = load('dct_vectors.npz')
    for x, yf, xi in zip(g['x%d' % n], g['fdct%d' % n], g['idct%d' % n]):
        x = np.ascontiguousarray(x)
        y = np.zeros_like(x)
        o.orc_fdct_2d(n, p32(y), n, p32(x), n)
        assert np.array_equal(y, yf)
        o.orc_idct_2d(n, p32(y), n, p32(x), n)
        assert np.array_equal(y, xi)


def test_dcttest_anchor_recorded():
    # the reference's own self-test passed on the oracle-of-record build and its
    # stdout hash is the one SURVEY.md section 4 records
    with open(os.path.join(GOLDEN, 'dcttest.md5')) as f:
        assert f.read().startswith('eaace07761b6fd646b83285dfbdbc5c2')


def test_filter_haar_cfl_golden():
    o = oracle()
    g = load('filter_vectors.npz')
    for x, a, b in zip(g['x'], g['pre'], g['post']):
        x = np.ascontiguousarray(x)
        y = np.zeros(4, np.int32)
        o.orc_pre_filter4(p32(y), p32(x))
        assert np.array_equal(y, a)
        o.orc_post_filter4(p32(y), p32(x))
        assert np.array_equal(y, b)
    for ln in (2, 3, 4, 5):
        n = 1 << ln
        for x, yy in zip(g['haar_x%d' % n], g['haar_y%d' % n]):
            x = np.ascontiguousarray(x)
            y = np.zeros_like(x)
            o.orc_haar(p32(y), n, p32(x), n, ln)
            assert np.array_equal(y, yy)
            xb = np.zeros_like(x)
            o.orc_haar_inv(p32(xb), n, p32(y), n, ln)
            assert np.array_equal(xb, x)
    luma = np.ascontiguousarray(g['cfl_luma'])
    for bs, cbs in ((0, 0), (0, 1), (1, 2), (2, 3)):
        n = 4 << bs
        a = np.zeros((n, n), np.int32)
        o.orc_resample_luma_coeffs(p32(a), n, p32(luma), 64, 1, 1, bs, cbs)
        assert np.array_equal(a, g['cfl_%d_%d' % (bs, cbs)])


def test_coding_order_golden():
    o = oracle()
    g = load('coding_order.npz')
    for n in (4, 8, 16, 32):
        src = np.arange(n*n, dtype=np.int32)
        dst = np.full(n*n, -1, np.int32)
        o.orc_raster_to_coding_order(p32(dst), n, p32(src), n)
        t = g['n%d' % n]
        assert np.array_equal(dst[:len(t)], t)


def test_forward_plane_and_pyramid_golden():
    o = oracle()
    g = load('plane_forward.npz')
    pic_w, pic_h, fw, fh = [int(v) for v in g['geom']]
    nhsb, nvsb = fw//32, fh//32
    bmap = np.ascontiguousarray(g['bsize'])
    for pli in (0, 1):
        dec = 1 if pli else 0
        w, h = fw >> dec, fh >> dec
        pix = np.ascontiguousarray(g['pix%d' % pli])
        for kf in (0, 1):
            c = np.zeros((h, w), np.int32); d = np.zeros((h, w), np.int32)
            o.orc_forward_plane(p32(c), p32(d), pu8(pix), w, nhsb, nvsb, dec, pu8(bmap),
                                nhsb*4, pic_w, pic_h, kf)
            assert np.array_equal(d, g['d%d_kf%d' % (pli, kf)])
            if kf == 0:
                assert np.array_equal(c, g['c%d' % pli])
        nlev = 4 - dec
        lev = [np.zeros((h, w), np.int32) for _ in range(nlev)]
        arr = (ctypes.POINTER(ctypes.c_int32)*nlev)(*[p32(a) for a in lev])
        c = np.zeros((h, w), np.int32)
        o.orc_forward_pyramid_plane(p32(c), arr, nlev, pu8(pix), w, nhsb, nvsb, dec, pic_w, pic_h)
        for k in range(nlev):
            assert np.array_equal(lev[k], g['lev%d_%d' % (pli, k)])
        # and the inverse path restores the pixels exactly (non-keyframe coefficients)
        d = np.ascontiguousarray(g['d%d_kf0' % pli])
        out = np.zeros((h, w), np.uint8); c3 = np.zeros((h, w), np.int32)
        o.orc_inverse_plane(pu8(out), w, p32(c3), p32(d), nhsb, nvsb, dec, pu8(bmap), nhsb*4,
                            pic_w, pic_h)
        assert np.array_equal(out, pix)


def test_pvq_search_golden():
    o = oracle()
    g = load('pvq_search.npz')
    for x, y, k, g2, cd, n in zip(g['x'], g['y'], g['k'], g['g2'], g['cos_dist'], g['n']):
        n = int(n)
        xx = np.ascontiguousarray(x[:n])
        yy = np.zeros(n, np.int32)
        c = o.orc_pvq_search_rdo_double(pf64(xx), n, int(k), p32(yy), float(g2))
        assert np.array_equal(yy, y[:n]) and c == cd


def noref_decide(o, x0, n, q, beta, qm, rate_fn):
    """Candidates from the oracle + a rate callback -> (qg, k, y) like pvq_theta."""
    cg = ctypes.c_double(); g = ctypes.c_double()
    qg = np.zeros(2, np.int32); k = np.zeros(2, np.int32)
    cd = np.zeros(2); dist = np.zeros(2); y = np.zeros((2, n), np.int32)
    nc = o.orc_pvq_noref_candidates(p32(x0), n, q, beta, p16(qm), 1, ctypes.byref(cg),
                                    ctypes.byref(g), qg.ctypes.data_as(ctypes.POINTER(c_int)),
                                    k.ctypes.data_as(ctypes.POINTER(c_int)), pf64(cd), pf64(dist),
                                    p32(y))
    return nc, cg.value, qg, k, cd, dist, y


def test_pvq_theta_noref_golden_candidates_contain_decision():
    """Without the (adaptive, host-side) rate term the oracle cannot pick, but the
    reference's decision must be one of its candidates (or the null vector), with
    identical pulses, and the synthesised coefficients must match."""
    o = oracle()
    g = load('pvq_theta_noref.npz')
    prm = load('encoder_params.npz')
    hits = 0
    for i in range(len(g['n'])):
        n, bs, off = int(g['n'][i]), int(g['bs'][i]), int(g['off'][i])
        tag = 'q20_m%d' % int(g['masking'][i])
        qm = np.ascontiguousarray(prm['qm_' + tag][bs*2048 + off:bs*2048 + off + n])
        qmi = np.ascontiguousarray(prm['qm_inv_' + tag][bs*2048 + off:bs*2048 + off + n])
        x0 = np.ascontiguousarray(g['x0'][i][:n])
        q, beta = int(g['q'][i]), float(g['beta'][i])
        nc, cg, qg, k, cd, dist, y = noref_decide(o, x0, n, q, beta, qm, None)
        qg_r, k_r = int(g['qg'][i]), int(g['k'][i])
        if qg_r == 0:
            assert not g['out'][i].any()
            continue
        sel = [c for c in range(nc) if qg[c] == qg_r]
        assert len(sel) == 1
        c = sel[0]
        assert k[c] == k_r and np.array_equal(y[c], g['y'][i][:n])
        gexp = o.orc_gain_expand(float(qg_r), q, beta)
        out = np.zeros(n, np.int32)
        o.orc_pvq_synthesis_partial(p32(out), p32(y[c]), pf64(np.zeros(n)), n, 1, gexp, 0., 0, 1,
                                    p16(qmi))
        assert np.array_equal(out, g['out'][i][:n])
        hits += 1
    assert hits > 50


def test_compute_dist_golden():
    o = oracle()
    o.orc_compute_dist.restype = ctypes.c_double
    g = load('compute_dist.npz')
    for bs in (1, 2, 3):
        n = 4 << bs
        for m in (0, 1):
            for x, y, d in zip(g['x_%d' % bs], g['y_%d' % bs], g['dist_%d_m%d' % (bs, m)]):
                got = o.orc_compute_dist(p32(np.ascontiguousarray(x)), p32(np.ascontiguousarray(y)), n,
                                         pf64(np.ascontiguousarray(g['mag2_%d' % bs])), m)
                assert got == d


def test_filters_8_16_32_against_reference_vectors():
    """A3, the n-point variants (only the reference's dcttest/tools reach them):
    oracle == reference outputs, and post(pre(x)) == x."""
    o = oracle()
    g = load('filter_n_vectors.npz')
    for n in (8, 16, 32):
        for x, a, b in zip(g['x%d' % n], g['pre%d' % n], g['post%d' % n]):
            x = np.ascontiguousarray(x)
            y = np.zeros(n, np.int32)
            o.orc_pre_filter_n(n, p32(y), p32(x))
            assert np.array_equal(y, a)
            z = np.zeros(n, np.int32)
            o.orc_post_filter_n(n, p32(z), p32(y))
            assert np.array_equal(z, x)
            o.orc_post_filter_n(n, p32(y), p32(x))
            assert np.array_equal(y, b)


def test_cfl_resample_other_decimations():
    """A9 for 4:2:2 / 4:4:0 / 4:4:4 chroma: oracle == reference vectors."""
    o = oracle()
    g = load('cfl_decimations.npz')
    luma = np.ascontiguousarray(g['luma'])
    for xdec, ydec in ((1, 0), (0, 1), (0, 0)):
        for bs, cbs in ((0, 0), (1, 1), (2, 2)):
            n = 4 << bs
            b = np.zeros((n, n), np.int32)
            o.orc_resample_luma_coeffs(p32(b), n, p32(luma), 64, xdec, ydec, bs, cbs)
            assert np.array_equal(b, g['p_%d%d_%d_%d' % (xdec, ydec, bs, cbs)]), (xdec, ydec, bs)


def test_obmc_blocks_golden():
    """F3: the oracle's OBMC block prediction against the reference outputs stored in
    tests/golden/mc_blocks.npz (od_mc_predict through a live context, tools/gen_golden.py)."""
    import ctypes
    g = load('mc_blocks.npz')
    o = oracle()
    refs, pad, dst = g['refs'], int(g['pad']), g['dst']
    rw = refs.shape[2]
    U8P, I32P = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int32)
    got = np.zeros_like(dst)
    for b in g['blocks']:
        x, y, lx, ly = (int(v) for v in b[:4])
        n, m = 1 << lx, 1 << ly
        mvx, mvy = np.ascontiguousarray(b[8:12]), np.ascontiguousarray(b[12:16])
        srcs = (U8P*4)(*[ctypes.cast(refs[int(k)].ctypes.data + (pad + y)*rw + pad + x, U8P) for k in b[4:8]])
        out = np.zeros((m, n), np.uint8)
        o.orc_mc_predict(pu8(out), n, srcs, rw, mvx.ctypes.data_as(I32P), mvy.ctypes.data_as(I32P),
                         int(b[16]), int(b[17]), lx, ly)
        got[y:y + m, x:x + n] = out
    assert np.array_equal(got, dst)


def test_sad_satd_pairs_golden():
    """F3: the oracle's SAD / SATD (orc_mc_sad8, orc_mc_satd8) against the reference's C vtable
    entries' outputs stored in tests/golden/mc_sad_pairs.npz (tools/gen_golden.py)."""
    import ctypes
    g = load('mc_sad_pairs.npz')
    o = oracle()
    src, rf = g['src'], g['ref']
    W = src.shape[1]
    U8P = ctypes.POINTER(ctypes.c_uint8)
    o.orc_mc_sad8.restype = o.orc_mc_satd8.restype = ctypes.c_int32
    got = []
    for sx, sy, rx, ry, lg, satd in g['pairs']:
        f = o.orc_mc_satd8 if satd else o.orc_mc_sad8
        got.append(f(ctypes.cast(src.ctypes.data + int(sy)*W + int(sx), U8P), W,
                     ctypes.cast(rf.ctypes.data + int(ry)*W + int(rx), U8P), W, int(lg)))
    assert np.array_equal(np.array(got, np.int32), g['out'])


def test_mvest_calc_sads_golden():
    """F3, second half: the oracle's od_mv_est_calc_sads (grid -> items -> OBMC of every plane +
    clipped SAD, chroma scaled) against sad_cache as the REAL od_mv_est_calc_sads wrote it inside a
    reference encoder (tests/golden/mvest_sads.npz, oracle/ref_probe/mcenc_probe.c)."""
    from testlib import mvest_items, mvest_oracle_sads, mvest_split
    g = load('mvest_sads.npz')
    o = oracle()
    items, sizes, smax = mvest_items(o, g)
    assert list(sizes) == [0, g['sad1'].size, g['sad2'].size] and list(smax) == [0, 4, 4]
    got = mvest_split(mvest_oracle_sads(o, g, items), sizes, smax, g['dims'])
    assert np.array_equal(got[1], g['sad1']) and np.array_equal(got[2], g['sad2'])
    assert g['sad1'].max() > 0


def test_mvest_bma_sad_golden():
    """F3: the oracle's od_mv_est_bma_sad (single-vector prediction of every plane + the SAD clipped
    against the picture on all four sides) against the REAL static function's values for 400 blocks
    centred on grid vertices - 50 of them hanging over the frame's left or top edge, others over
    the right / bottom edge - stored in tests/golden/mvest_sads.npz (bma_req, bma_sad)."""
    from testlib import BMA_REC, mvest_oracle_bma_windows
    g = load('mvest_sads.npz')
    o = oracle()
    req = g['bma_req']
    recs = np.zeros(len(req), BMA_REC)
    recs['bx'], recs['by'], recs['log_blk_sz'], recs['ref'] = req[:, 0], req[:, 1], req[:, 2], req[:, 3]
    recs['cx'], recs['cy'] = req[:, 4], req[:, 5]
    recs['xmin'] = recs['ymin'] = -(1 << 13)
    recs['xmax'] = recs['ymax'] = 1 << 13
    got = mvest_oracle_bma_windows(o, g, recs, 0)
    assert np.array_equal(got[:, 0], g['bma_sad'])
    assert (req[:, 0] < 0).sum() + (req[:, 1] < 0).sum() > 20 and g['bma_sad'].max() > 1000
