"""GPU parity tests: every HIP entry point of include/daala_hip.h, called through
the C ABI (ctypes), against the CPU oracle on the same seeded inputs and against
the committed golden vectors.  Integer paths must be bit-exact; the only
floating-point tolerance in this file is stated where it is used."""
import ctypes
import os

import numpy as np
import pytest

from testlib import GOLDEN, c_int, oracle, p16, p32, pf64, pu8, random_bsize_map, synth_plane

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    import daala_amd.binding as b
    lib = b.load()          # raises when the extension is not built: no fallback
    assert lib.od_hip_device_count() > 0, 'no HIP device visible'
    return b


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def orc_blocks(fn, n, blocks):
    o = oracle()
    out = np.zeros_like(blocks)
    for i in range(len(blocks)):
        getattr(o, fn)(n, p32(out[i]), n, p32(np.ascontiguousarray(blocks[i])), n)
    return out


@pytest.mark.parametrize('bs', (0, 1, 2, 3))
def test_dct_blocks_ieee1180_ranges(hip, bs):
    """The reference's own transform test recipe (dcttest, src/dct.c:3439-3469):
    ranges (-256,255), (-5,5), (-300,300), both signs - here at the coefficient
    scale (<<4) plus a wide range; forward, inverse on arbitrary data, and exact
    reconstruction."""
    n = 4 << bs
    rng = np.random.default_rng(100 + bs)
    parts = []
    for lo, hi in ((-256, 255), (-5, 5), (-300, 300)):
        v = rng.integers(lo, hi + 1, size=(400, n, n), dtype=np.int32) << 4
        parts += [v, -v]
    parts.append(rng.integers(-60000, 60001, size=(300, n, n), dtype=np.int32))
    x = np.concatenate(parts)
    y = hip.od_bin_fdct_blocks(bs, x)
    assert np.array_equal(y, orc_blocks('orc_fdct_2d', n, x))
    xi = hip.od_bin_idct_blocks(bs, x)
    assert np.array_equal(xi, orc_blocks('orc_idct_2d', n, x))
    assert np.array_equal(hip.od_bin_idct_blocks(bs, y), x)


def test_dct_golden_vectors(hip):
    g = golden('dct_vectors.npz')
    for bs, n in enumerate((4, 8, 16, 32)):
        assert np.array_equal(hip.od_bin_fdct_blocks(bs, g['x%d' % n]), g['fdct%d' % n])
        assert np.array_equal(hip.od_bin_idct_blocks(bs, g['x%d' % n]), g['idct%d' % n])


def test_dct_empty_and_ragged_batches(hip):
    for bs in range(4):
        n = 4 << bs
        assert hip.od_bin_fdct_blocks(bs, np.zeros((0, n, n), np.int32)).shape == (0, n, n)
        rng = np.random.default_rng(bs)
        for cnt in (1, 3, 256//n + 1, 1000 + 7):       # not multiples of the WG batch
            x = rng.integers(-4096, 4096, size=(cnt, n, n), dtype=np.int32)
            assert np.array_equal(hip.od_bin_fdct_blocks(bs, x), orc_blocks('orc_fdct_2d', n, x))


def test_vtable_dropins_strided_and_in_place(hip):
    """od_dct_func_2d contract (src/dct.h:61-62): element strides, out may alias in."""
    lib = hip.load()
    f = (ctypes.c_void_p*5)()
    i = (ctypes.c_void_p*5)()
    lib.od_hip_vtbl_fill.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    assert lib.od_hip_vtbl_fill(f, i) == 0
    assert all(f[k] and i[k] for k in range(4)) and not f[4] and not i[4]
    o = oracle()
    rng = np.random.default_rng(5)
    for bs, n in enumerate((4, 8, 16, 32)):
        plane = rng.integers(-3000, 3001, size=(40, 48), dtype=np.int32)
        ref_p = plane.copy()
        sub = plane[3:, 5:]
        subr = ref_p[3:, 5:]
        # in place, strided
        hip.vtable_call('od_hip_bin_fdct%dx%d' % (n, n), sub, 48, sub, 48)
        tmp = np.ascontiguousarray(subr[:n, :n])
        out = np.zeros_like(tmp)
        o.orc_fdct_2d(n, p32(out), n, p32(tmp), n)
        subr[:n, :n] = out
        assert np.array_equal(plane, ref_p)
        hip.vtable_call('od_hip_bin_idct%dx%d' % (n, n), sub, 48, sub, 48)
        o.orc_idct_2d(n, p32(tmp), n, p32(out), n)
        subr[:n, :n] = tmp
        assert np.array_equal(plane, ref_p)


def test_filter4_haar_cfl(hip):
    o = oracle()
    g = golden('filter_vectors.npz')
    assert np.array_equal(hip.od_pre_filter4(g['x']), g['pre'])
    assert np.array_equal(hip.od_post_filter4(g['x']), g['post'])
    rng = np.random.default_rng(9)
    v = rng.integers(-100000, 100001, size=(50001, 4), dtype=np.int32)
    pre = hip.od_pre_filter4(v)
    exp = np.zeros_like(v)
    for k in range(0, len(v), 97):
        o.orc_pre_filter4(p32(exp[k]), p32(np.ascontiguousarray(v[k])))
        assert np.array_equal(pre[k], exp[k])
    assert np.array_equal(hip.od_post_filter4(pre), v)        # exact inversion
    for bs, ln in enumerate((2, 3, 4, 5)):
        n = 1 << ln
        x = g['haar_x%d' % n]
        y = hip.od_haar_blocks(bs, x)
        assert np.array_equal(y, g['haar_y%d' % n])
        assert np.array_equal(hip.od_haar_blocks(bs, y, inverse=True), x)
    luma = g['cfl_luma']
    for bs, cbs in ((0, 0), (0, 1), (1, 2), (2, 3)):
        p = hip.od_resample_luma_coeffs_420(luma, 64, [0], bs, cbs)
        assert np.array_equal(p[0], g['cfl_%d_%d' % (bs, cbs)])
    # many blocks at different offsets against the oracle
    offs = [(y*8)*64 + x*8 for y in range(3) for x in range(7)]
    p = hip.od_resample_luma_coeffs_420(luma, 64, offs, 0, 0)
    for k, off in enumerate(offs):
        a = np.zeros((4, 4), np.int32)
        o.orc_resample_luma_coeffs(p32(a), 4, p32(np.ascontiguousarray(luma).ravel()[off:]), 64,
                                   1, 1, 0, 0)
        assert np.array_equal(p[k], a)


def oracle_pyramid(pix, fw, fh, dec, pic_w, pic_h):
    o = oracle()
    w, h = fw >> dec, fh >> dec
    nlev = 4 - dec
    lev = [np.zeros((h, w), np.int32) for _ in range(nlev)]
    arr = (ctypes.POINTER(ctypes.c_int32)*nlev)(*[p32(a) for a in lev])
    c = np.zeros((h, w), np.int32)
    o.orc_forward_pyramid_plane(p32(c), arr, nlev, pu8(pix), w, fw//32, fh//32, dec, pic_w, pic_h)
    return lev


def test_forward_pyramid_golden_and_oracle(hip):
    g = golden('plane_forward.npz')
    pic_w, pic_h, fw, fh = [int(v) for v in g['geom']]
    ctx = hip.DaalaHip(pic_w, pic_h, fw, fh, nplanes=2, xdec=(0, 1), nslots=2)
    planes = [g['pix0'], g['pix1']]
    ctx.upload_planes(0, planes)
    ctx.upload_planes(1, [p[::-1].copy() for p in planes])     # a second, different frame
    ctx.forward_pyramid()
    for pli in (0, 1):
        for k in range(ctx.nlevels(pli)):
            assert np.array_equal(ctx.download_level(0, pli, k), g['lev%d_%d' % (pli, k)])
        lev = oracle_pyramid(np.ascontiguousarray(planes[pli][::-1]), fw, fh, pli, pic_w, pic_h)
        for k in range(ctx.nlevels(pli)):
            assert np.array_equal(ctx.download_level(1, pli, k), lev[k])
    ctx.close()


def test_forward_known_golden_and_inverse_roundtrip(hip):
    g = golden('plane_forward.npz')
    pic_w, pic_h, fw, fh = [int(v) for v in g['geom']]
    ctx = hip.DaalaHip(pic_w, pic_h, fw, fh, nplanes=2, xdec=(0, 1), nslots=1)
    ctx.upload_planes(0, [g['pix0'], g['pix1']])
    ctx.set_bsize(0, g['bsize'])
    for kf in (1, 0):
        ctx.forward_known(keyframe=kf)
        for pli in (0, 1):
            assert np.array_equal(ctx.download_coeffs(0, pli), g['d%d_kf%d' % (pli, kf)])
    ctx.inverse()                      # coefficients of the last (non-keyframe) pass
    for pli in (0, 1):
        assert np.array_equal(ctx.download_recon(0, pli), g['pix%d' % pli])
    ctx.close()


def test_set_bsize_rejects_inconsistent_maps(hip):
    ctx = hip.DaalaHip(64, 64, nplanes=1, xdec=(0,), nslots=1)
    bad = np.zeros((8, 8), np.uint8)
    bad[0, 0] = 3
    with pytest.raises(hip.HipError):
        ctx.set_bsize(0, bad)
    with pytest.raises(hip.HipError):
        ctx.forward_pyramid(slot0=0, nslots=2)      # slot range
    ctx.close()


def test_inverse_of_quantised_coefficients_matches_oracle(hip):
    """The decoder case: coefficients that are NOT in the image of the forward
    path (coarsely quantised), mixed block sizes, picture not a multiple of 32."""
    o = oracle()
    pic_w, pic_h, fw, fh = 150, 100, 192, 128
    nhsb, nvsb = fw//32, fh//32
    ctx = hip.DaalaHip(pic_w, pic_h, fw, fh, nplanes=2, xdec=(0, 1), nslots=1)
    bmap = random_bsize_map(nhsb, nvsb, 123)
    ctx.set_bsize(0, bmap)
    planes = [synth_plane(fw, fh, 4), synth_plane(fw//2, fh//2, 4, 1)]
    ctx.upload_planes(0, planes)
    ctx.forward_known(keyframe=0)
    for pli in (0, 1):
        d = ctx.download_coeffs(0, pli)
        q = 96
        dq = (np.sign(d)*((np.abs(d) + q//2)//q)*q).astype(np.int32)
        ctx.upload_coeffs(0, pli, dq)
    ctx.inverse()
    for pli in (0, 1):
        h, w = ctx.plane_shape(pli)
        dq = ctx.download_coeffs(0, pli)
        out = np.zeros((h, w), np.uint8); c = np.zeros((h, w), np.int32)
        o.orc_inverse_plane(pu8(out), w, p32(c), p32(dq), nhsb, nvsb, pli, pu8(bmap), nhsb*4,
                            pic_w, pic_h)
        assert np.array_equal(ctx.download_recon(0, pli), out)
    ctx.close()


@pytest.mark.parametrize('fw,fh,amp', [(320, 224, 4000), (192, 128, 400000), (448, 96, 60000)])
def test_inverse_out_of_range_coefficients_and_odd_tilings(hip, fw, fh, amp):
    """The fused inverse (interior written as 8 bit, 2-sample edge strips as int16 with an
    int32 escape) on coefficients far outside what an encoder produces - reconstruction
    values that do not fit int16, and beyond the 24-bit multiplier's range - and on plane
    widths that end in a partial 64-wide tile (chroma 160, 96, 224): identical to the oracle."""
    o = oracle()
    nhsb, nvsb = fw//32, fh//32
    ctx = hip.DaalaHip(fw - 7, fh - 5, fw, fh, nplanes=2, xdec=(0, 1), nslots=1)
    bmap = random_bsize_map(nhsb, nvsb, 77)
    ctx.set_bsize(0, bmap)
    rng = np.random.default_rng(amp)
    ds = []
    for pli in (0, 1):
        h, w = ctx.plane_shape(pli)
        d = (rng.laplace(0, 1, size=(h, w))*amp).astype(np.int32)
        d[rng.random((h, w)) < .7] = 0
        ctx.upload_coeffs(0, pli, d)
        ds.append(d)
    ctx.inverse()
    for pli in (0, 1):
        h, w = ctx.plane_shape(pli)
        out = np.zeros((h, w), np.uint8); c = np.zeros((h, w), np.int32)
        o.orc_inverse_plane(pu8(out), w, p32(c), p32(ds[pli]), nhsb, nvsb, pli, pu8(bmap), nhsb*4,
                            fw - 7, fh - 5)
        assert np.array_equal(ctx.download_recon(0, pli), out), (pli, int(np.abs(c).max()))
        if amp >= 60000:
            assert np.abs(c).max() > 40000          # the escape path was exercised
    ctx.close()


def test_full_size_1080p_roundtrip_and_sampled_oracle(hip):
    """BASELINE config 2 geometry (1920x1080 4:2:0, padded to 1920x1088): the
    size-independent property forward(known) -> inverse == identity on the whole
    frame, plus the pyramid checked against the oracle on the full frame."""
    pic_w, pic_h, fw, fh = 1920, 1080, 1920, 1088
    ctx = hip.DaalaHip(pic_w, pic_h, fw, fh, nslots=2)
    planes = [synth_plane(fw, fh, 7), synth_plane(fw//2, fh//2, 7, 1), synth_plane(fw//2, fh//2, 8, 1)]
    for s in range(2):
        ctx.upload_planes(s, planes if s == 0 else [p[:, ::-1].copy() for p in planes])
        ctx.set_bsize(s, random_bsize_map(fw//32, fh//32, 40 + s))
    ctx.forward_known(keyframe=0)
    ctx.inverse()
    for pli in range(3):
        assert np.array_equal(ctx.download_recon(0, pli), planes[pli])
        assert np.array_equal(ctx.download_recon(1, pli), planes[pli][:, ::-1])
    ctx.forward_pyramid()
    for pli in (0, 1):
        lev = oracle_pyramid(planes[pli], fw, fh, 1 if pli else 0, pic_w, pic_h)
        for k in range(ctx.nlevels(pli)):
            assert np.array_equal(ctx.download_level(0, pli, k), lev[k])
    ctx.close()


# ---------------------------------------------------------------------------
def test_pvq_search_vectors(hip):
    o = oracle()
    g = golden('pvq_search.npz')
    for n in (8, 15, 32, 128):
        m = g['n'] == n
        y, cd = hip.pvq_search_vectors(g['x'][m][:, :n], g['k'][m], g['g2'][m])
        assert np.array_equal(y, g['y'][m][:, :n])
        assert np.array_equal(cd, g['cos_dist'][m])          # bit-exact doubles
    rng = np.random.default_rng(77)
    for n in (7, 8, 14, 15, 16, 31, 32, 127, 128):
        nv = 300
        x = rng.laplace(0, 1, size=(nv, n))*np.exp(-np.arange(n)/7.)
        x[::3] = rng.integers(-2, 3, size=(len(x[::3]), n))    # exact ties
        k = rng.choice([1, 2, 3, 4, 8, 13, 33, 100, 260], size=nv).astype(np.int32)
        g2 = rng.uniform(.05, 80, size=nv)
        y, cd = hip.pvq_search_vectors(x, k, g2)
        for i in range(nv):
            yo = np.zeros(n, np.int32)
            co = o.orc_pvq_search_rdo_double(pf64(np.ascontiguousarray(x[i])), n, int(k[i]), p32(yo),
                                             float(g2[i]))
            assert np.array_equal(y[i], yo), (n, i)
            assert cd[i] == co, (n, i)


def test_pvq_synthesis_noref(hip):
    o = oracle()
    rng = np.random.default_rng(81)
    for n in (8, 15, 32, 128):
        nv = 200
        y = rng.integers(-4, 5, size=(nv, n), dtype=np.int32)
        y[0] = 0
        g = rng.uniform(1, 9000, size=nv)
        qmi = rng.integers(3000, 16000, size=(nv, n)).astype(np.int16)
        out = hip.pvq_synthesis_noref(y, g, qmi)
        for i in range(nv):
            e = np.zeros(n, np.int32)
            o.orc_pvq_synthesis_partial(p32(e), p32(np.ascontiguousarray(y[i])), pf64(np.zeros(n)), n,
                                        1, float(g[i]), 0., 0, 1, p16(np.ascontiguousarray(qmi[i])))
            assert np.array_equal(out[i], e)


def level_params(prm, tag, pli, bs, xdec):
    q0 = int(prm['quantizer_' + tag][pli])
    pq = prm['pvq_qm_q4_' + tag][pli]
    off = {0: [1, 16], 1: [1, 16, 24, 32, 64], 2: [1, 16, 24, 32, 64, 96, 128, 256],
           3: [1, 16, 24, 32, 64, 96, 128, 256, 384, 512]}[bs]
    nb = len(off) - 1
    q = [max(1, q0*int(pq[bs*(bs + 1) + (b + 1) - (b + 1)//3]) >> 4) for b in range(nb)]
    masking = tag.endswith('m1')
    beta = [1.5 if (masking and pli == 0 and bs > 0) else 1.0]*nb
    n = 4 << bs
    base = bs*2048 + xdec*1024
    qm = prm['qm_' + tag][base:base + n*n]
    return off, q, beta, np.ascontiguousarray(qm)


@pytest.mark.parametrize('tag', ('q20_m0', 'q20_m1'))
def test_pvq_noref_level_vs_oracle(hip, tag):
    """Frame-wide no-reference candidates (state-free part of pvq_theta) for every
    band of every block of every pyramid level, real -v 20 QM: bit-exact in every field
    for beta == 1 and beta == 1.5 alike (the companding pow() of beta == 1.5 is the
    host's libm between the gain pass and the search pass)."""
    o = oracle()
    prm = golden('encoder_params.npz')
    pic_w, pic_h, fw, fh = 150, 100, 192, 128
    ctx = hip.DaalaHip(pic_w, pic_h, fw, fh, nplanes=2, xdec=(0, 1), nslots=1)
    planes = [synth_plane(fw, fh, 11), synth_plane(fw//2, fh//2, 11, 1)]
    ctx.upload_planes(0, planes)
    ctx.forward_pyramid()
    for pli in (0, 1):
        for level in range(ctx.nlevels(pli)):
            n = (32 >> pli) >> level
            bs = {4: 0, 8: 1, 16: 2, 32: 3}[n]
            off, q, beta, qm = level_params(prm, tag, pli, bs, pli)
            ctx.pvq_noref_search(pli, level, qm, q, beta)
            bands, ys = ctx.pvq_download(0, pli, level)
            lev = ctx.download_level(0, pli, level)
            h, w = lev.shape
            nbx = w//n
            rng = np.random.default_rng(level)
            for blk in rng.choice(len(bands), size=min(len(bands), 60), replace=False):
                by, bx = divmod(int(blk), nbx)
                block = np.ascontiguousarray(lev[by*n:by*n + n, bx*n:bx*n + n])
                co = np.zeros(n*n, np.int32)
                o.orc_raster_to_coding_order(p32(co), n, p32(block), n)
                for b in range(len(off) - 1):
                    nn = off[b + 1] - off[b]
                    x0 = np.ascontiguousarray(co[off[b]:off[b + 1]])
                    cg = ctypes.c_double(); g = ctypes.c_double()
                    qg = np.zeros(2, np.int32); k = np.zeros(2, np.int32)
                    cd = np.zeros(2); dist = np.zeros(2); y = np.zeros((2, nn), np.int32)
                    nc = o.orc_pvq_noref_candidates(
                        p32(x0), nn, q[b], beta[b], p16(np.ascontiguousarray(qm[off[b]:off[b + 1]])),
                        1, ctypes.byref(cg), ctypes.byref(g),
                        qg.ctypes.data_as(ctypes.POINTER(c_int)),
                        k.ctypes.data_as(ctypes.POINTER(c_int)), pf64(cd), pf64(dist), p32(y))
                    r = bands[blk, b]
                    assert r['g'] == g.value
                    assert r['ncand'] == nc
                    assert r['cg'] == cg.value
                    for c in range(nc):
                        assert r['qg'][c] == qg[c] and r['k'][c] == k[c]
                        assert np.array_equal(ys[blk, c, off[b]:off[b + 1]], y[c])
                        assert r['cos_dist'][c] == cd[c] and r['dist'][c] == dist[c]
    ctx.close()


def test_pvq_theta_golden_decisions_reachable(hip):
    """tests/golden/pvq_theta_noref.npz holds the REFERENCE's pvq_theta outcome
    for single bands; the device candidates for the same band must contain it."""
    g = golden('pvq_theta_noref.npz')
    prm = golden('encoder_params.npz')
    for n in (8, 15, 32, 128):
        idx = [i for i in range(len(g['n'])) if g['n'][i] == n and g['qg'][i] > 0 and g['beta'][i] == 1.0]
        if not idx:
            continue
        xs, ks, g2s = [], [], []
        for i in idx:
            bs, off = int(g['bs'][i]), int(g['off'][i])
            tag = 'q20_m%d' % int(g['masking'][i])
            qm = prm['qm_' + tag][bs*2048 + off:bs*2048 + off + n].astype(np.int64)
            x0 = g['x0'][i][:n].astype(np.int64)
            x1 = (x0*qm).astype(np.float64)*(1./32767)
            acc = 0.0
            for v, m in zip(x0, qm):
                acc += float(v)*float(v)*float(m)*(1./32767)*float(m)*(1./32767)
            cg = np.sqrt(acc)/int(g['q'][i])
            xs.append(x1); ks.append(int(g['k'][i])); g2s.append(float(g['qg'][i])*cg)
        y, _ = hip.pvq_search_vectors(np.array(xs), np.array(ks, np.int32), np.array(g2s))
        for j, i in enumerate(idx):
            assert np.array_equal(y[j], g['y'][i][:n])


def test_sb_row_strips_on_device_equal_full_frame(hip):
    """Config-3 style sharding: each 'rank' transforms its strip of superblock rows
    (+1 halo SB row each side) on the device; stacking the owned rows must equal the
    full-frame pyramid.  Ranks are emulated sequentially on the one GPU; the
    collective itself is covered by tests/test_sharding_gloo.py."""
    from daala_amd import sharding
    pic_w, pic_h, fw, fh = 150, 200, 192, 224
    planes = [synth_plane(fw, fh, 31), synth_plane(fw//2, fh//2, 31, 1)]
    xdec = (0, 1)
    full = sharding.hip_strip_compute()(planes, (pic_w, pic_h, fw, fh))
    for world in (2, 3, 7):
        parts = [sharding.strip_pyramid(sharding.SbRowShard(pic_w, pic_h, fw, fh, world, r), planes,
                                        xdec, sharding.hip_strip_compute()) for r in range(world)]
        for pli, d in enumerate(xdec):
            for k in range(4 - d):
                got = np.concatenate([p[pli][k] for p in parts], axis=0)
                assert np.array_equal(got, full[pli][k]), (world, pli, k)
    lev = oracle_pyramid(planes[0], fw, fh, 0, pic_w, pic_h)
    assert np.array_equal(full[0][3], lev[3])


def ulps(a, b):
    return abs(a - b)/np.spacing(max(abs(a), abs(b), 1e-300))


@pytest.mark.parametrize('is_keyframe,pli', ((1, 0), (1, 1), (0, 0)))
def test_pvq_theta_vectors_full_candidate_lists(hip, is_keyframe, pli):
    """Complete pvq_theta candidate enumeration (with-reference gain/theta search
    + no-reference search) on the device vs the oracle: EVERY field bit-exact - candidate
    lists, K, pulses, Householder axis/sign, gains, theta, cosine distances, distortions.
    (acos/sin/cos/pow are evaluated by the host's libm between the two device passes,
    DESIGN.md section 5; the device does only exactly rounded arithmetic.)"""
    from testlib import ThetaOut
    o = oracle()
    o.orc_pvq_theta_candidates.argtypes = [ctypes.POINTER(ctypes.c_int32)]*2 + [
        c_int, c_int, ctypes.c_double, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_int16),
        ctypes.POINTER(ThetaOut), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    rng = np.random.default_rng(300 + 2*is_keyframe + pli)
    for n in (8, 15, 32, 128):
        for beta in (1.0, 1.5):
            nv = 160
            amp = rng.choice([3, 30, 300, 3000], size=nv)
            x0 = (rng.laplace(0, 1, size=(nv, n))*amp[:, None]).astype(np.int32)
            r0 = x0.copy()
            kind = rng.integers(0, 4, size=nv)
            noise = (rng.laplace(0, 1, size=(nv, n))*(amp[:, None]/3. + 1)).astype(np.int32)
            r0 = np.where((kind == 0)[:, None], (x0*1.2).astype(np.int32) + noise, r0)
            r0 = np.where((kind == 1)[:, None], noise*3, r0)
            r0 = np.where((kind == 3)[:, None], 0, r0).astype(np.int32)
            qm = rng.integers(9000, 32768, size=n).astype(np.int16)
            q0 = rng.integers(2, 200, size=nv).astype(np.int32)
            out, y_ref, y_nr = hip.pvq_theta_vectors(x0, r0, qm, q0, beta, 1, is_keyframe, pli)
            nsearch = 0
            for v in range(nv):
                t = ThetaOut()
                yr = np.zeros((12, n), np.int32); yn = np.zeros((2, n), np.int32)
                o.orc_pvq_theta_candidates(p32(np.ascontiguousarray(x0[v])),
                                           p32(np.ascontiguousarray(r0[v])), n, int(q0[v]), beta, 1,
                                           is_keyframe, pli, p16(qm), ctypes.byref(t), p32(yr), p32(yn))
                d = out[v]
                nsearch += t.theta_searched
                assert (d.icgr, d.m, d.s, d.nref, d.nnoref, d.theta_searched, d.noref_searched) == \
                    (t.icgr, t.m, t.s, t.nref, t.nnoref, t.theta_searched, t.noref_searched), (n, v)
                assert d.g == t.g and d.gr == t.gr and d.corr == t.corr
                assert d.cg == t.cg and d.cgr == t.cgr and d.gain_offset == t.gain_offset
                assert d.theta == t.theta and d.skip_dist == t.skip_dist and d.null_dist == t.null_dist
                for c in range(t.nref):
                    assert (d.ref_qg[c], d.ref_itheta[c], d.ref_ts[c], d.ref_k[c]) == \
                        (t.ref_qg[c], t.ref_itheta[c], t.ref_ts[c], t.ref_k[c]), (n, v, c)
                    assert np.array_equal(y_ref[v, c, :n - 1], yr[c, :n - 1]), (n, v, c)
                    assert d.ref_qtheta[c] == t.ref_qtheta[c]
                    assert d.ref_cos_dist[c] == t.ref_cos_dist[c] and d.ref_dist[c] == t.ref_dist[c]
                for c in range(t.nnoref):
                    assert (d.nr_qg[c], d.nr_k[c]) == (t.nr_qg[c], t.nr_k[c])
                    assert np.array_equal(y_nr[v, c], yn[c])
                    assert d.nr_cos_dist[c] == t.nr_cos_dist[c] and d.nr_dist[c] == t.nr_dist[c]
            assert nsearch > 20


def test_pvq_synthesis_vectors_both_branches(hip):
    o = oracle()
    o.orc_pvq_synthesis.argtypes = [ctypes.POINTER(ctypes.c_int32)]*3 + [
        c_int, ctypes.c_double, c_int, ctypes.c_double, ctypes.c_double,
        ctypes.POINTER(ctypes.c_int16), ctypes.POINTER(ctypes.c_int16)]
    rng = np.random.default_rng(91)
    for n in (8, 15, 32, 128):
        nv = 300
        noref = rng.integers(0, 2, size=nv).astype(np.int32)
        y = rng.integers(-3, 4, size=(nv, n), dtype=np.int32)
        y[noref == 0, n - 1] = 0
        ref = rng.integers(-900, 901, size=(nv, n), dtype=np.int32)
        qm = rng.integers(9000, 32768, size=n).astype(np.int16)
        qmi = np.floor(.5 + 32768.*4096./qm.astype(np.float64)).astype(np.int16)
        gr = np.sqrt(((ref.astype(np.float64)*qm*(1./32767))**2).sum(axis=1))
        g = rng.uniform(1, 6000, size=nv)
        theta = rng.uniform(0, 1.5, size=nv)
        out = hip.pvq_synthesis_vectors(y, ref, gr, noref, g, theta, qm, qmi)
        for v in range(nv):
            e = np.zeros(n, np.int32)
            o.orc_pvq_synthesis(p32(e), p32(np.ascontiguousarray(y[v])), p32(np.ascontiguousarray(ref[v])),
                                n, float(gr[v]), int(noref[v]), float(g[v]), float(theta[v]), p16(qm),
                                p16(qmi))
            # both branches bit-exact: sin/cos of the with-reference branch are the host's
            assert np.array_equal(out[v], e), (n, v, int(noref[v]))


def test_hv_intra_pred_blocks(hip):
    o = oracle()
    rng = np.random.default_rng(95)
    w = h = 96
    d = rng.integers(-500, 501, size=(h, w), dtype=np.int32)
    for bs in range(4):
        nb = 1 << bs
        bsz = rng.integers(0, 4, size=(h//8, w//8)).astype(np.uint8)
        bsz[rng.random(bsz.shape) < .5] = bs
        bx = rng.integers(0, w//4//nb, size=150).astype(np.int32)*nb
        by = rng.integers(0, h//4//nb, size=150).astype(np.int32)*nb
        pred = hip.od_hv_intra_pred_blocks(d, bsz, bs, bx, by)
        n = 4 << bs
        for i in range(len(bx)):
            e = np.zeros(n*n, np.int32)
            o.orc_hv_intra_pred(p32(e), p32(d), w, int(bx[i]), int(by[i]), pu8(bsz), w//8, bs)
            assert np.array_equal(pred[i], e)


def test_compute_dist_blocks(hip):
    """od_compute_dist on the device vs the reference values in the golden fixture
    (generated by the real reference with this image's glibc): bit-exact - the activity
    pow(., -1/6) is the host's libm applied to the device's exact argument."""
    g = golden('compute_dist.npz')
    for bs in (1, 2, 3):
        for m in (0, 1):
            d = hip.od_compute_dist_blocks(bs, g['x_%d' % bs], g['y_%d' % bs], g['mag2_%d' % bs], m)
            e = g['dist_%d_m%d' % (bs, m)]
            assert np.array_equal(d, e), (bs, m)


def test_filters_8_16_32(hip):
    """od_pre/post_filter{4,8,16,32}: device vs the reference's vectors, exact inversion."""
    import daala_amd.binding as b
    g = golden('filter_n_vectors.npz')
    for n in (8, 16, 32):
        assert np.array_equal(b.od_filter_vectors(n, g['x%d' % n]), g['pre%d' % n])
        assert np.array_equal(b.od_filter_vectors(n, g['x%d' % n], inverse=True), g['post%d' % n])
        rng = np.random.default_rng(n)
        v = rng.integers(-200000, 200001, size=(20001, n), dtype=np.int32)
        assert np.array_equal(b.od_filter_vectors(n, b.od_filter_vectors(n, v), inverse=True), v)
    g4 = golden('filter_vectors.npz')
    assert np.array_equal(b.od_filter_vectors(4, g4['x']), g4['pre'])
    assert np.array_equal(b.od_filter_vectors(4, g4['x'], inverse=True), g4['post'])


def test_cfl_resample_other_decimations(hip):
    """od_resample_luma_coeffs for 4:2:2 (od_tf_up_h_lp), 4:4:0 (od_tf_up_v_lp), 4:4:4."""
    import daala_amd.binding as b
    g = golden('cfl_decimations.npz')
    luma = g['luma']
    for xdec, ydec in ((1, 0), (0, 1), (0, 0)):
        for bs, cbs in ((0, 0), (1, 1), (2, 2)):
            p = b.od_resample_luma_coeffs(luma, 64, [0], bs, cbs, xdec, ydec)
            assert np.array_equal(p[0], g['p_%d%d_%d_%d' % (xdec, ydec, bs, cbs)]), (xdec, ydec, bs)
    # many blocks at several offsets against the oracle, 4:2:2 and 4:4:0
    o = oracle()
    offs = [(y*8)*64 + x*4 for y in range(3) for x in range(7)]
    for xdec, ydec in ((1, 0), (0, 1)):
        p = b.od_resample_luma_coeffs(luma, 64, offs, 0, 0, xdec, ydec)
        for i, off in enumerate(offs):
            e = np.zeros((4, 4), np.int32)
            o.orc_resample_luma_coeffs(p32(e), 4, p32(np.ascontiguousarray(luma.ravel()[off:])), 64,
                                       xdec, ydec, 0, 0)
            assert np.array_equal(p[i], e), (xdec, ydec, i)


def test_coding_order_gather_and_scatter(hip):
    """A11 both directions on the device vs the oracle (the reference's permutation tables):
    od_raster_to_coding_order and od_coding_order_to_raster; a 32x32 block moves 512 of its
    1024 coefficients, the rest of the destination keeps the caller's values."""
    o = oracle()
    rng = np.random.default_rng(77)
    for bs in range(4):
        n = 4 << bs
        x = rng.integers(-5000, 5001, size=(40, n, n), dtype=np.int32)
        base = rng.integers(-9, 10, size=(40, n*n), dtype=np.int32)
        got = hip.od_coding_order_blocks(bs, x.reshape(40, n*n), dst=base)
        back = hip.od_coding_order_blocks(bs, got, to_raster=True, dst=base)
        for b in range(40):
            e = base[b].copy()
            o.orc_raster_to_coding_order(p32(e), n, p32(np.ascontiguousarray(x[b])), n)
            assert np.array_equal(got[b], e), (bs, b)
            r = np.ascontiguousarray(base[b].reshape(n, n).copy())
            o.orc_coding_order_to_raster(p32(r), n, p32(np.ascontiguousarray(got[b])), n)
            assert np.array_equal(back[b].reshape(n, n), r), (bs, b)


def test_obmc_prediction_blocks(hip):
    """F3, first inter kernel: od_hip_mc_predict_blocks against the reference's od_mc_predict
    outputs (tests/golden/mc_blocks.npz: 117 blocks of 4x4..32x32, one to three references,
    full and fractional vectors, every (oc, s)) and against the oracle on a random list."""
    import ctypes
    g = golden('mc_blocks.npz')
    refs, pad, dst = g['refs'], int(g['pad']), g['dst']
    blocks = [dict(x=int(b[0]), y=int(b[1]), lx=int(b[2]), ly=int(b[3]), ref=b[4:8], mvx=b[8:12],
                   mvy=b[12:16], oc=int(b[16]), s=int(b[17])) for b in g['blocks']]
    got = hip.od_mc_predict_blocks(list(refs), pad, pad, blocks, np.zeros_like(dst))
    assert np.array_equal(got, dst)
    # random (also non-square) blocks vs the oracle
    o = oracle()
    rng = np.random.default_rng(9)
    H, W, P = 128, 192, 40
    rr = [rng.integers(0, 256, size=(H + 2*P, W + 2*P), dtype=np.uint8) for _ in range(2)]
    rw = W + 2*P
    U8P, I32P = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int32)
    bl, want = [], np.zeros((H, W), np.uint8)
    for y in range(0, H, 64):
        for x in range(0, W, 64):
            lx, ly = int(rng.integers(2, 7)), int(rng.integers(2, 7))
            mvx = rng.integers(-8*(P - 6), 8*(P - 6), size=4).astype(np.int32) | 1
            mvy = rng.integers(-8*(P - 6), 8*(P - 6), size=4).astype(np.int32)
            ks = rng.integers(0, 2, size=4)
            oc, s = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            bl.append(dict(x=x, y=y, lx=lx, ly=ly, ref=ks, mvx=mvx, mvy=mvy, oc=oc, s=s))
            n, m = 1 << lx, 1 << ly
            out = np.zeros((m, n), np.uint8)
            srcs = (U8P*4)(*[ctypes.cast(rr[int(k)].ctypes.data + (P + y)*rw + P + x, U8P) for k in ks])
            o.orc_mc_predict(pu8(out), n, srcs, rw, mvx.ctypes.data_as(I32P), mvy.ctypes.data_as(I32P),
                             oc, s, lx, ly)
            want[y:y + m, x:x + n] = out
    got = hip.od_mc_predict_blocks(rr, P, P, bl, np.zeros((H, W), np.uint8))
    assert np.array_equal(got, want)


def test_mvest_calc_sads_fused_obmc_sad(hip):
    """F3, second half: od_hip_mc_sad_items (k_mc_sad_items: OBMC prediction of every plane fused
    with the clipped SAD, one wave per item, nothing written but the sums) against sad_cache as the
    REAL od_mv_est_calc_sads wrote it inside a reference encoder (tests/golden/mvest_sads.npz) and
    against the oracle on a random item list: all four block sizes, every (oc, s), one to three
    references, vectors that reach into the padding, picture sizes that clip blocks, luma only and
    three planes."""
    from testlib import mvest_items, mvest_oracle_sads, mvest_split, SAD_ITEM
    g = golden('mvest_sads.npz')
    o = oracle()
    dims = g['dims']
    items, sizes, smax = mvest_items(o, g)
    refs = [g['refs%d' % p] for p in range(3)]
    src = [g['src%d' % p] for p in range(3)]
    ox = [int(dims[9 + 4*p]) for p in range(3)]
    oy = [int(dims[10 + 4*p]) for p in range(3)]
    mc = hip.McSad(refs, ox, oy, src, [(0, 0), (1, 1), (1, 1)])
    got = mvest_split(mc.sad_items(items, int(g['pic'][0]), int(g['pic'][1])), sizes, smax, dims)
    assert np.array_equal(got[1], g['sad1']) and np.array_equal(got[2], g['sad2'])
    # random items vs the oracle
    rng = np.random.default_rng(31)
    fw, fh = src[0].shape[1], src[0].shape[0]
    nimg = refs[0].shape[0]
    for pic_w, pic_h, nplanes in ((int(g['pic'][0]), int(g['pic'][1]), 3), (fw - 21, fh - 37, 3), (fw, fh, 1)):
        it = np.zeros(700, SAD_ITEM)
        for i in range(len(it)):
            lg = int(rng.integers(3, 7))
            n = 1 << lg
            it[i]['log_blk_sz'] = lg
            it[i]['x'] = int(rng.integers(0, (fw - n)//8 + 1))*8
            it[i]['y'] = int(rng.integers(0, (fh - n)//8 + 1))*8
            it[i]['oc'], it[i]['s'] = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            one = rng.random() < .3
            it[i]['ref'] = int(rng.integers(0, nimg)) if one else rng.integers(0, nimg, size=4)
            reach = 8*(ox[0] - 8)
            it[i]['mvx'] = rng.integers(-reach, reach, size=4)
            it[i]['mvy'] = rng.integers(-reach, reach, size=4)
            if rng.random() < .25:
                it[i]['mvx'] &= ~7
            if rng.random() < .25:
                it[i]['mvy'] &= ~7
            if rng.random() < .2:
                it[i]['mvx'] = it[i]['mvx'][0]
                it[i]['mvy'] = it[i]['mvy'][0]
        gg = dict(g)
        gg['pic'] = np.array([pic_w, pic_h], np.int32)
        want = mvest_oracle_sads(o, gg, it, nplanes)
        got = mc.sad_items(it, pic_w, pic_h, nplanes)
        assert np.array_equal(got, want), (pic_w, pic_h, nplanes, np.flatnonzero(got != want)[:5])
    # bad operands are refused before anything is launched
    bad = it[:1].copy()
    bad[0]['x'] = fw
    with pytest.raises(hip.HipError):
        mc.sad_items(bad, fw, fh)
    bad = it[:1].copy()
    bad[0]['ref'][2] = nimg
    with pytest.raises(hip.HipError):
        mc.sad_items(bad, fw, fh)
    mc.close()


def test_mvest_bma_windows(hip):
    """F3, EPZS initialisation: od_hip_mc_bma_windows (k_mc_bma_windows: one wave per vertex and
    window offset) against the REAL od_mv_est_bma_sad's values (tests/golden/mvest_sads.npz: blocks
    centred on grid vertices, some hanging over the frame on every side) at the window centres, and
    against the oracle for every half-sample vector of 5x5 windows - with limits that cut some
    windows, luma only and three planes."""
    from testlib import BMA_REC, mvest_oracle_bma_windows
    g = golden('mvest_sads.npz')
    o = oracle()
    dims = g['dims']
    refs = [g['refs%d' % p] for p in range(3)]
    src = [g['src%d' % p] for p in range(3)]
    ox = [int(dims[9 + 4*p]) for p in range(3)]
    oy = [int(dims[10 + 4*p]) for p in range(3)]
    mc = hip.McSad(refs, ox, oy, src, [(0, 0), (1, 1), (1, 1)])
    req = g['bma_req']
    recs = np.zeros(len(req), BMA_REC)
    recs['bx'], recs['by'], recs['log_blk_sz'], recs['ref'] = req[:, 0], req[:, 1], req[:, 2], req[:, 3]
    recs['cx'], recs['cy'] = req[:, 4], req[:, 5]
    recs['xmin'] = recs['ymin'] = -(1 << 13)
    recs['xmax'] = recs['ymax'] = 1 << 13
    pw, ph = int(g['pic'][0]), int(g['pic'][1])
    got = mc.bma_windows(recs, 0, pw, ph)
    assert np.array_equal(got[:, 0], g['bma_sad'])
    rng = np.random.default_rng(4)
    cut = recs[:160].copy()                      # limits that cut the windows
    cut['xmin'] = cut['cx'] - rng.integers(0, 3, size=len(cut))
    cut['xmax'] = cut['cx'] + rng.integers(0, 3, size=len(cut))
    cut['ymin'] = cut['cy'] - rng.integers(0, 3, size=len(cut))
    cut['ymax'] = cut['cy'] + rng.integers(0, 3, size=len(cut))
    for nplanes in (3, 1):
        want = mvest_oracle_bma_windows(o, g, cut, 2, nplanes)
        got = mc.bma_windows(cut, 2, pw, ph, nplanes)
        assert (want == -1).any() and (want[:, 12] >= 0).all()
        assert np.array_equal(got, want), (nplanes, np.argwhere(got != want)[:5])
    # the one-wave-per-offset kernel (OD_HIP_BMA_V1=1, kept for A/B) gives the same windows
    os.environ['OD_HIP_BMA_V1'] = '1'
    try:
        for nplanes in (3, 1):
            want = mvest_oracle_bma_windows(o, g, cut, 2, nplanes)
            assert np.array_equal(mc.bma_windows(cut, 2, pw, ph, nplanes), want), nplanes
        assert np.array_equal(mc.bma_windows(recs, 0, pw, ph)[:, 0], g['bma_sad'])
    finally:
        del os.environ['OD_HIP_BMA_V1']
    # more records than one launch's grid takes (a 4K frame's finest level): launched in chunks
    small = cut[cut['log_blk_sz'] == cut['log_blk_sz'].min()]
    many = np.tile(small, 1 + 40000//len(small))[:40000]
    assert len(many) == 40000
    many['cx'] += rng.integers(-6, 7, size=len(many))
    many['cy'] += rng.integers(-6, 7, size=len(many))
    want = mvest_oracle_bma_windows(o, g, many, 1, 1)
    got = mc.bma_windows(many, 1, pw, ph, 1)
    assert np.array_equal(got, want)
    bad = recs[:1].copy()
    bad['ref'] = refs[0].shape[0]
    with pytest.raises(hip.HipError):
        mc.bma_windows(bad, 1, pw, ph)
    mc.close()


def test_reference_image_built_on_the_device_from_a_reconstruction(hip):
    """od_hip_mc_set_ref_ctx (k_mc_ref_from_rec): a context's reconstruction plane becomes a
    reference image of a prediction object ON the device, the padding filled as od_img_edge_ext
    fills it (src/state.c:1100-1171: the frame's edge samples replicated).  Reference: the same
    image padded on the host (numpy edge mode) and uploaded with od_hip_mc_set_ref; both objects
    must give identical SADs for vectors that reach deep into every side of the padding - and
    those must equal the oracle's."""
    from testlib import mvest_oracle_sads, SAD_ITEM
    o = oracle()
    fw, fh, pad = 192, 128, 96
    planes = [synth_plane(fw, fh, 5), synth_plane(fw//2, fh//2, 6, 1), synth_plane(fw//2, fh//2, 7, 1)]
    ctx = hip.DaalaHip(fw, fh, fw, fh, nplanes=3, xdec=(0, 1, 1), nslots=1)
    ctx.upload_planes(0, planes)
    ctx.forward_haar()                       # lossless round trip: the reconstruction IS the input
    ctx.inverse_haar()
    for pli in range(3):
        assert np.array_equal(ctx.download_recon(0, pli), planes[pli])
    refs = [np.pad(planes[p], pad >> (p > 0), mode='edge')[None] for p in range(3)]
    src = [synth_plane(fw, fh, 9), synth_plane(fw//2, fh//2, 10, 1), synth_plane(fw//2, fh//2, 11, 1)]
    org = [pad, pad//2, pad//2]
    dec = [(0, 0), (1, 1), (1, 1)]
    host = hip.McSad(refs, org, org, src, dec)
    dev = hip.McSad([np.zeros_like(r) for r in refs], org, org, src, dec)
    for pli in range(3):
        dev.set_ref_ctx(pli, 0, ctx, 0, refs[pli].shape[2], refs[pli].shape[1], org[pli], org[pli])
    rng = np.random.default_rng(77)
    it = np.zeros(400, SAD_ITEM)
    for i in range(len(it)):
        lg = int(rng.integers(3, 7))
        n = 1 << lg
        it[i]['log_blk_sz'] = lg
        it[i]['x'] = int(rng.choice([0, fw - n, int(rng.integers(0, (fw - n)//8 + 1))*8]))
        it[i]['y'] = int(rng.choice([0, fh - n, int(rng.integers(0, (fh - n)//8 + 1))*8]))
        it[i]['oc'], it[i]['s'] = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        it[i]['mvx'] = rng.integers(-8*(pad - 8), 8*(pad - 8), size=4)
        it[i]['mvy'] = rng.integers(-8*(pad - 8), 8*(pad - 8), size=4)
    a = host.sad_items(it, fw, fh)
    b = dev.sad_items(it, fw, fh)
    assert np.array_equal(a, b), np.flatnonzero(a != b)[:5]
    g = {'dims': np.array([fw//8, fh//8, 0, 4, 1, fw, fh] + sum([[refs[p].shape[2], refs[p].shape[1], org[p], org[p]]
                                                                  for p in range(3)], []), np.int32),
         'pic': np.array([fw, fh], np.int32)}
    for p in range(3):
        g['refs%d' % p], g['src%d' % p] = refs[p], src[p]
    assert np.array_equal(a, mvest_oracle_sads(o, g, it))
    host.close()
    dev.close()
    ctx.close()


def test_superblock_row_strips_in_c_equal_the_full_frame(hip):
    """SURVEY 8e in C: od_hip_set_strip restricts the forward pyramid and the PVQ passes to a
    strip of superblock rows (the kernels read their lapping halo from the pixels: no halo
    recomputation); three uneven strips computed one after the other reproduce the full
    frame bit for bit - planes, gains, candidates, pulses.  od_hip_gather_strips (RCCL,
    device to device) is exercised with a one-rank communicator."""
    prm = golden('encoder_params.npz')
    pic_w, pic_h, fw, fh = 300, 200, 320, 224                    # 7 superblock rows
    planes = [synth_plane(fw, fh, 21), synth_plane(fw//2, fh//2, 21, 1), synth_plane(fw//2, fh//2, 22, 1)]

    def run(strips):
        ctx = hip.DaalaHip(pic_w, pic_h, fw, fh, nplanes=3, xdec=(0, 1, 1), nslots=1)
        ctx.upload_planes(0, planes)
        for r0, r1 in strips:
            ctx.set_strip(r0, r1)
            ctx.forward_pyramid()
            for pli in (0, 1):
                for level in range(ctx.nlevels(pli)):
                    n = (32 >> (pli > 0)) >> level
                    bs = {4: 0, 8: 1, 16: 2, 32: 3}[n]
                    off, q, beta, qm = level_params(prm, 'q20_m1', pli, bs, int(pli > 0))
                    ctx.pvq_noref_search(pli, level, qm, q, beta)
        ctx.set_strip(0, fh//32)
        return ctx

    full = run([(0, 7)])
    part = run([(0, 3), (3, 4), (4, 7)])
    comm = hip.Comm(0, 1, 0, hip.comm_unique_id())
    part.gather_strips(comm, 0, [0, 7])
    comm.close()
    for pli in range(3):
        for level in range(full.nlevels(pli)):
            assert np.array_equal(full.download_level(0, pli, level), part.download_level(0, pli, level))
    for pli in (0, 1):
        for level in range(full.nlevels(pli)):
            (bf, yf), (bp, yp) = full.pvq_download(0, pli, level), part.pvq_download(0, pli, level)
            assert np.array_equal(yf, yp)
            for f in ('cg', 'g', 'cos_dist', 'dist', 'qg', 'k', 'ncand'):
                assert np.array_equal(bf[f], bp[f]), (pli, level, f)
    # the packed strip (what travels to the coding rank as ONE message per owner): a context that
    # computed rows 0..3 imports the strip of rows 3..7 another context computed
    a = run([(0, 3)])
    b = run([(3, 7)])
    blob = b.strip_export(0, 3, 7)
    assert blob.size > 0
    a.strip_import(0, 3, 7, blob)
    for pli in range(3):
        for level in range(full.nlevels(pli)):
            assert np.array_equal(full.download_level(0, pli, level), a.download_level(0, pli, level))
    for pli in (0, 1):
        for level in range(full.nlevels(pli)):
            (bf, yf), (bp, yp) = full.pvq_download(0, pli, level), a.pvq_download(0, pli, level)
            assert np.array_equal(yf, yp)
            for f in ('cg', 'g', 'cos_dist', 'qg', 'k', 'ncand'):
                assert np.array_equal(bf[f], bp[f]), (pli, level, f)
    a.close()
    b.close()
    full.close()
    part.close()


def test_pframe_feed_complete_candidate_lists_vs_oracle(hip):
    """The P-frame feed (include/daala_hip.h 4d): for an input frame and a prediction, every band
    of every block of every level of every plane gets pvq_theta's COMPLETE candidate list on the
    device (is_keyframe = 0: with-reference (gain, theta) candidates in the reference's loop
    order + no-reference ones).  Against the oracle's pvq_theta on the oracle's own pyramids:
    gains, companded gains, correlation, theta, which searches run, and per slot K, cosine
    distance and pulses - every field bit-exact."""
    from testlib import ThetaOut
    o = oracle()
    o.orc_pvq_theta_candidates.argtypes = [ctypes.POINTER(ctypes.c_int32)]*2 + [
        c_int, c_int, ctypes.c_double, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_int16),
        ctypes.POINTER(ThetaOut), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    prm = golden('encoder_params.npz')
    tag = 'q20_m1'
    pic_w, pic_h, fw, fh = 128, 96, 128, 96
    rng = np.random.default_rng(5)
    src = [synth_plane(fw + 8, fh + 8, 21), synth_plane(fw//2 + 8, fh//2 + 8, 21, 1), synth_plane(fw//2 + 8, fh//2 + 8, 22, 1)]
    inp = [np.ascontiguousarray(p[4:4 + (fh >> (i > 0)), 4:4 + (fw >> (i > 0))]) for i, p in enumerate(src)]
    # a "prediction": the frame displaced by a pixel plus a little noise; one flat region (null-ish reference)
    pred = []
    for i, p in enumerate(src):
        h, w = fh >> (i > 0), fw >> (i > 0)
        q = p[3:3 + h, 5:5 + w].astype(np.int32) + rng.integers(-3, 4, size=(h, w))
        q[:h//4, :w//4] = 128
        pred.append(np.clip(q, 0, 255).astype(np.uint8))
    pf = hip.PFeed(pic_w, pic_h, fw, fh)
    par = {}
    for pli in range(3):
        for level in range(pf.nlevels(pli)):
            n = (32 >> (pli > 0)) >> level
            bs = {4: 0, 8: 1, 16: 2, 32: 3}[n]
            off, q, beta, qm = level_params(prm, tag, pli, bs, int(pli > 0))
            par[pli, level] = (n, off, q, beta, qm)
            pf.set_level(pli, level, qm, q, beta)
    pf.run(inp, pred)
    checked = nsearch = nnoref = 0
    for pli in range(3):
        dec = int(pli > 0)
        w, h = fw >> dec, fh >> dec
        nlev = pf.nlevels(pli)
        pyr = []
        for planes in (inp, pred):
            lev = [np.zeros((h, w), np.int32) for _ in range(nlev)]
            arr = (ctypes.POINTER(ctypes.c_int32)*nlev)(*[p32(a) for a in lev])
            c = np.zeros((h, w), np.int32)
            o.orc_forward_pyramid_plane(p32(c), arr, nlev, pu8(planes[pli]), w, fw//32, fh//32, dec, pic_w, pic_h)
            pyr.append(lev)
        for level in range(nlev):
            n, off, q, beta, qm = par[pli, level]
            v = pf.view(pli, level)
            nb, nblk, nbx = v['nbands'], v['nblk'], v['nbx']
            assert v['nslots'] == 14 and v['nref_slots'] == 12
            ncoded = min(n*n, 512)
            for blk in range(nblk):
                bx, by = blk % nbx, blk//nbx
                cx = np.zeros(n*n, np.int32)
                cr = np.zeros(n*n, np.int32)
                o.orc_raster_to_coding_order(p32(cx), n, p32(pyr[0][level][by*n:, bx*n:][:n, :n].copy()), n)
                o.orc_raster_to_coding_order(p32(cr), n, p32(pyr[1][level][by*n:, bx*n:][:n, :n].copy()), n)
                for b in range(nb):
                    nn = off[b + 1] - off[b]
                    ns = (nn + 1) & ~1
                    yo = 0 if b == 0 else off[b]
                    r = b*nblk + blk
                    t = ThetaOut()
                    yr = np.zeros((12, nn), np.int32)
                    yn = np.zeros((2, nn), np.int32)
                    o.orc_pvq_theta_candidates(p32(np.ascontiguousarray(cx[off[b]:off[b + 1]])),
                                               p32(np.ascontiguousarray(cr[off[b]:off[b + 1]])), nn, int(q[b]),
                                               float(beta[b]), 1, 0, pli,
                                               p16(np.ascontiguousarray(qm[off[b]:off[b + 1]])),
                                               ctypes.byref(t), p32(yr), p32(yn))
                    assert v['g'][r] == t.g and v['gr'][r] == t.gr, (pli, level, blk, b)
                    assert v['cg'][r] == t.cg and v['cgr'][r] == t.cgr
                    assert v['corr'][r] == t.corr
                    assert (v['flags'][r] & 1) == t.theta_searched and ((v['flags'][r] >> 1) & 1) == t.noref_searched
                    if t.theta_searched:
                        assert v['theta'][r] == t.theta
                    ybase = 14*nblk*yo
                    for c in range(12):
                        k = v['k'][c, r]
                        if c < t.nref:
                            assert k == t.ref_k[c], (pli, level, blk, b, c, k, t.ref_k[c])
                            assert v['cos_dist'][c, r] == t.ref_cos_dist[c], (pli, level, blk, b, c)
                            yy = v['y'][ybase + (c*nblk + blk)*ns: ybase + (c*nblk + blk)*ns + nn - 1]
                            assert np.array_equal(yy, yr[c, :nn - 1]), (pli, level, blk, b, c)
                            nsearch += 1
                        else:
                            assert k == -1
                    for c in range(2):
                        k = v['k'][12 + c, r]
                        if c < t.nnoref:
                            assert k == t.nr_k[c]
                            assert v['cos_dist'][12 + c, r] == t.nr_cos_dist[c]
                            yy = v['y'][ybase + ((12 + c)*nblk + blk)*ns: ybase + ((12 + c)*nblk + blk)*ns + nn]
                            assert np.array_equal(yy, yn[c])
                            nnoref += 1
                        else:
                            assert k == -1
                    checked += 1
    pf.close()
    assert checked > 2000 and nsearch > 3000 and nnoref > 500, (checked, nsearch, nnoref)


def test_decoder_synthesis_frame_vs_oracle(hip):
    """od_hip_dsynth_* (header section 4e; pvq_synthesis src/pvq_decoder.c:104-118, od_init_skipped_coeffs
    src/state.c:1351-1357, od_coding_order_to_raster src/partition.c:176, od_pvq_compute_gain
    src/pvq.c:456-464): a whole frame of random block sizes, DC values and band decisions (copied
    from the reference, cleared, synthesised with and without reference; pulses within and beyond
    16 bits) against the oracle's band-by-band restatement on the same prediction pyramid."""
    import daala_amd.binding as b
    o = oracle()
    o.orc_pvq_synthesis.argtypes = [ctypes.POINTER(ctypes.c_int32)]*3 + [
        c_int, ctypes.c_double, c_int, ctypes.c_double, ctypes.c_double,
        ctypes.POINTER(ctypes.c_int16), ctypes.POINTER(ctypes.c_int16)]
    o.orc_pvq_compute_gain.restype = ctypes.c_double
    o.orc_pvq_compute_gain.argtypes = [ctypes.POINTER(ctypes.c_int32), c_int, c_int,
                                       ctypes.POINTER(ctypes.c_double), ctypes.c_double,
                                       ctypes.POINTER(ctypes.c_int16)]
    rng = np.random.default_rng(123)
    W, Hh = 128, 96
    ctx = b.DaalaHip(W, Hh, W, Hh, nplanes=3, xdec=(0, 1, 1), nslots=1)
    planes = [rng.integers(0, 256, size=(Hh, W), dtype=np.uint8),
              rng.integers(0, 256, size=(Hh//2, W//2), dtype=np.uint8),
              rng.integers(0, 256, size=(Hh//2, W//2), dtype=np.uint8)]
    ctx.upload_planes(0, planes)
    ctx.forward_pyramid(0, 1)
    ds = b.DSynth(ctx)
    OFF = [1, 16, 24, 32, 64, 96, 128, 256, 384, 512]
    NB = [1, 4, 7, 9]
    qm, qmi, md = {}, {}, {}
    for pli in range(3):
        for level in range(ctx.nlevels(pli)):
            n = (32 >> ctx.xdec[pli]) >> level
            q = rng.integers(9000, 32768, size=n*n).astype(np.int16)
            qm[pli, level] = q
            qmi[pli, level] = np.floor(.5 + 32768.*4096./q.astype(np.float64)).astype(np.int16)
            ds.set_level(pli, level, qm[pli, level], qmi[pli, level])
            md[pli, level] = ctx.download_level(0, pli, level)
    # reference gains of every band of every block size
    gr = ds.ref_gains()
    for (pli, level), g in gr.items():
        n = (32 >> ctx.xdec[pli]) >> level
        bs = {4: 0, 8: 1, 16: 2, 32: 3}[n]
        w = W >> ctx.xdec[pli]
        for blk in rng.integers(0, g.shape[1], size=12):
            by, bx = divmod(int(blk), w//n)
            cod = np.zeros(n*n, np.int32)
            o.orc_raster_to_coding_order(p32(cod), n, p32(np.ascontiguousarray(md[pli, level][by*n:(by + 1)*n, bx*n:(bx + 1)*n])), n)
            for band in range(NB[bs]):
                gg = ctypes.c_double()
                ref = np.ascontiguousarray(cod[OFF[band]:OFF[band + 1]])
                o.orc_pvq_compute_gain(p32(ref), ref.size, 1, ctypes.byref(gg), 1.0, p16(qm[pli, level][OFF[band]:]))
                assert gg.value == g[band, blk], (pli, level, band, blk)
    # a frame of records
    want = []
    nblocks = nbands = npulses = 0
    for pli in range(3):
        w, h = W >> ctx.xdec[pli], Hh >> ctx.xdec[pli]
        sb = 32 >> ctx.xdec[pli]
        exp = np.zeros((h, w), np.int32)
        for sy in range(0, h, sb):
            for sx in range(0, w, sb):
                level = int(rng.integers(0, ctx.nlevels(pli)))
                n = sb >> level
                bs = {4: 0, 8: 1, 16: 2, 32: 3}[n]
                for by in range(sy, sy + sb, n):
                    for bx in range(sx, sx + sb, n):
                        blk = ds.blocks[nblocks]
                        blk.org, blk.pli, blk.bs = by*w + bx, pli, bs
                        blk.dc = int(rng.integers(-5000, 5001))
                        mdb = np.ascontiguousarray(md[pli, level][by:by + n, bx:bx + n])
                        cod = np.zeros(n*n, np.int32)
                        o.orc_raster_to_coding_order(p32(cod), n, p32(mdb), n)
                        out = cod.copy()
                        for band in range(NB[bs]):
                            nn = OFF[band + 1] - OFF[band]
                            mode = int(rng.choice([-1, b.DSYNTH_ZERO, b.DSYNTH_NOREF, b.DSYNTH_REF]))
                            if mode < 0:
                                continue                                  # copied from the reference: no record
                            r = ds.bands[nbands]
                            r.block, r.band, r.yoff = nblocks, band, npulses
                            if mode == b.DSYNTH_ZERO:
                                r.mode = mode
                                out[OFF[band]:OFF[band + 1]] = 0
                                nbands += 1
                                continue
                            noref = mode == b.DSYNTH_NOREF
                            ny = nn - (not noref)
                            wide = rng.random() < .15
                            y = rng.integers(-3, 4, size=ny).astype(np.int32)
                            if wide:
                                y[int(rng.integers(0, ny))] = int(rng.choice([-1, 1]))*int(rng.integers(32768, 46000))   # y*y stays an int32
                            g = float(rng.uniform(1, 60000))
                            theta = float(rng.uniform(0, 1.5))
                            r.mode = mode | (b.DSYNTH_WIDE if wide else 0)
                            r.g, r.sin_theta, r.cos_theta = g, (0. if noref else np.sin(theta)), (0. if noref else np.cos(theta))
                            for i in range(ny):
                                if wide:
                                    v = int(y[i]) & 0xffffffff
                                    ds.pulses[npulses + 2*i] = ctypes.c_int16(v & 0xffff).value
                                    ds.pulses[npulses + 2*i + 1] = ctypes.c_int16(v >> 16).value
                                else:
                                    ds.pulses[npulses + i] = int(y[i])
                            npulses += ny*(2 if wide else 1)
                            nbands += 1
                            ref = np.ascontiguousarray(cod[OFF[band]:OFF[band + 1]])
                            ypad = np.zeros(nn, np.int32)
                            ypad[:ny] = y
                            gg = ctypes.c_double()
                            o.orc_pvq_compute_gain(p32(ref), nn, 1, ctypes.byref(gg), 1.0, p16(qm[pli, level][OFF[band]:]))
                            e = np.zeros(nn, np.int32)
                            o.orc_pvq_synthesis(p32(e), p32(ypad), p32(ref), nn, gg.value, int(noref), g, theta,
                                                p16(qm[pli, level][OFF[band]:]), p16(qmi[pli, level][OFF[band]:]))
                            out[OFF[band]:OFF[band + 1]] = e
                        out[0] = cod[0] + blk.dc
                        ras = mdb.copy()                 # od_init_skipped_coeffs: what coding order does not cover
                        o.orc_coding_order_to_raster(p32(ras), n, p32(out), n)
                        exp[by:by + n, bx:bx + n] = ras
                        nblocks += 1
        want.append(exp)
    ds.run(nblocks, nbands, npulses)
    for pli in range(3):
        got = ctx.download_coeffs(0, pli)
        assert np.array_equal(got, want[pli]), pli
    # a record that points outside its plane is refused on the host, not launched
    ds.blocks[0].org = (Hh*W) - 4
    with pytest.raises(b.HipError):
        ds.run(nblocks, nbands, npulses)
    ds.close()
