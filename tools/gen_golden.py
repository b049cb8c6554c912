#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference build (oracle/_ref, compiled
from /root/reference by oracle/Makefile).  Dev container only; the fixtures are
committed so that the GPU box and later rounds need neither the reference nor
this script.  Only DATA is stored: seeded inputs and the reference's outputs."""
import ctypes
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from testlib import (c_int, p16, p32, pf64, pu8, random_bsize_map, ref,  # noqa: E402
                     synth_plane)

G = os.path.join(ROOT, 'tests', 'golden')
r = ref()
pp = ref('pvq_probe')
ep = ref('enc_probe')
rng = np.random.default_rng(20261004)


def dct_vectors():
    out = {}
    for n in (4, 8, 16, 32):
        xs = []
        for amp in (255, 2047, 30000):
            xs.append(rng.integers(-amp, amp + 1, size=(6, n, n), dtype=np.int32))
        imp = np.zeros((2, n, n), np.int32)
        imp[0, 0, 0] = 255 << 4
        imp[1, n - 1, n - 1] = -(256 << 4)
        x = np.concatenate(xs + [imp])
        y = np.zeros_like(x)
        xi = np.zeros_like(x)
        for i in range(len(x)):
            getattr(r, 'od_bin_fdct%dx%d' % (n, n))(p32(y[i]), n, p32(x[i]), n)
            getattr(r, 'od_bin_idct%dx%d' % (n, n))(p32(xi[i]), n, p32(x[i]), n)
        out['x%d' % n] = x
        out['fdct%d' % n] = y
        out['idct%d' % n] = xi
    np.savez_compressed(os.path.join(G, 'dct_vectors.npz'), **out)


def filter_vectors():
    v = rng.integers(-40000, 40001, size=(512, 4), dtype=np.int32)
    v[:64] = rng.integers(-4, 5, size=(64, 4))
    pre = np.zeros_like(v)
    post = np.zeros_like(v)
    for i in range(len(v)):
        r.od_pre_filter4(p32(pre[i]), p32(v[i]))
        r.od_post_filter4(p32(post[i]), p32(v[i]))
    ln_out = {}
    for ln in (2, 3, 4, 5):
        n = 1 << ln
        x = rng.integers(-255, 256, size=(4, n, n), dtype=np.int32)
        y = np.zeros_like(x)
        for i in range(4):
            r.od_haar(p32(y[i]), n, p32(x[i]), n, ln)
        ln_out['haar_x%d' % n] = x
        ln_out['haar_y%d' % n] = y
    luma = rng.integers(-3000, 3001, size=(32, 64), dtype=np.int32)
    cfl = {}
    for bs, cbs in ((0, 0), (0, 1), (1, 2), (2, 3)):
        n = 4 << bs
        b = np.zeros((n, n), np.int32)
        r.od_resample_luma_coeffs(p32(b), n, p32(luma), 64, 1, 1, bs, cbs)
        cfl['cfl_%d_%d' % (bs, cbs)] = b
    np.savez_compressed(os.path.join(G, 'filter_vectors.npz'), x=v, pre=pre, post=post,
                        cfl_luma=luma, **ln_out, **cfl)


def filter_n_vectors():
    """od_pre/post_filter{8,16,32} (src/filter.c:306-1380): the variants only the
    reference's transform test tools reach; inputs + reference outputs as data."""
    out = {}
    for n in (8, 16, 32):
        v = rng.integers(-40000, 40001, size=(192, n), dtype=np.int32)
        v[:32] = rng.integers(-4, 5, size=(32, n))
        v[32:64] = rng.integers(-2040, 2041, size=(32, n))       # (pixel - 128) << 4 range
        pre = np.zeros_like(v)
        post = np.zeros_like(v)
        for i in range(len(v)):
            getattr(r, 'od_pre_filter%d' % n)(p32(pre[i]), p32(v[i]))
            getattr(r, 'od_post_filter%d' % n)(p32(post[i]), p32(v[i]))
        out['x%d' % n], out['pre%d' % n], out['post%d' % n] = v, pre, post
    np.savez_compressed(os.path.join(G, 'filter_n_vectors.npz'), **out)


def cfl_decimations():
    """od_resample_luma_coeffs (src/intra.c:72) for the chroma decimations other than
    4:2:0: 4:2:2 (od_tf_up_h_lp), 4:4:0 (od_tf_up_v_lp), 4:4:4 (copy)."""
    luma = rng.integers(-3000, 3001, size=(32, 64), dtype=np.int32)
    out = {'luma': luma}
    for xdec, ydec in ((1, 0), (0, 1), (0, 0)):
        for bs, cbs in ((0, 0), (1, 1), (2, 2)):
            n = 4 << bs
            b = np.zeros((n, n), np.int32)
            r.od_resample_luma_coeffs(p32(b), n, p32(luma), 64, xdec, ydec, bs, cbs)
            out['p_%d%d_%d_%d' % (xdec, ydec, bs, cbs)] = b
    np.savez_compressed(os.path.join(G, 'cfl_decimations.npz'), **out)


def plane_forward():
    pic_w, pic_h, fw, fh = 150, 100, 192, 128     # reference pads to multiples of 64
    nhsb, nvsb = fw//32, fh//32
    out = {'geom': np.array([pic_w, pic_h, fw, fh], np.int32)}
    bmap = random_bsize_map(nhsb, nvsb, 77)
    out['bsize'] = bmap
    for pli in (0, 1):
        dec = 1 if pli else 0
        w, h = fw >> dec, fh >> dec
        pix = synth_plane(w, h, seed=21 + pli, chroma=dec)
        out['pix%d' % pli] = pix
        for kf in (0, 1):
            c = np.zeros((h, w), np.int32)
            d = np.zeros((h, w), np.int32)
            assert ep.probe_forward_plane(pic_w, pic_h, pli, kf, pu8(pix), pu8(bmap), p32(c),
                                          p32(d)) == 0
            out['d%d_kf%d' % (pli, kf)] = d
            if kf == 0:
                out['c%d' % pli] = c
        # uniform maps = pyramid levels (see tests/test_oracle_vs_ref.py)
        for bs in range(4 - dec):
            um = np.full_like(bmap, 3 - bs)
            c = np.zeros((h, w), np.int32)
            d = np.zeros((h, w), np.int32)
            assert ep.probe_forward_plane(pic_w, pic_h, pli, 0, pu8(pix), pu8(um), p32(c),
                                          p32(d)) == 0
            out['lev%d_%d' % (pli, bs)] = d
    np.savez_compressed(os.path.join(G, 'plane_forward.npz'), **out)


def pvq_vectors():
    xs, ks, g2s, ys, cds, ns = [], [], [], [], [], []
    for n in (8, 15, 32, 128):
        for _ in range(40):
            kind = rng.integers(0, 3)
            if kind == 0:
                x = rng.laplace(0, 1, size=n)*np.exp(-np.arange(n)/9.)
            elif kind == 1:
                x = rng.integers(-3, 4, size=n).astype(np.float64)
            else:
                x = rng.integers(-300, 301, size=n)*rng.integers(9000, 32768, size=n)*(1./32767)
            k = int(rng.choice([1, 2, 3, 5, 9, 20, 64, 200]))
            g2 = float(rng.uniform(.1, 50))
            y = np.zeros(n, np.int32)
            cd = pp.probe_pvq_search_rdo_double(pf64(x), n, k, p32(y), g2)
            xp = np.zeros(128)
            xp[:n] = x
            yp = np.zeros(128, np.int32)
            yp[:n] = y
            xs.append(xp); ys.append(yp); ks.append(k); g2s.append(g2); cds.append(cd); ns.append(n)
    np.savez_compressed(os.path.join(G, 'pvq_search.npz'), x=np.array(xs), y=np.array(ys),
                        k=np.array(ks, np.int32), g2=np.array(g2s), cos_dist=np.array(cds),
                        n=np.array(ns, np.int32))


def encoder_params():
    out = {}
    for quant in (5, 20):
        for masking in (0, 1):
            q = (c_int*3)()
            pq = np.zeros(3*20, np.uint8)
            qm = np.zeros(4*2*1024, np.int16)
            qmi = np.zeros(4*2*1024, np.int16)
            assert ep.probe_encoder_params(quant, masking, q, pu8(pq), p16(qm), p16(qmi)) == 0
            tag = 'q%d_m%d' % (quant, masking)
            out['quantizer_' + tag] = np.array(list(q), np.int32)
            out['pvq_qm_q4_' + tag] = pq.reshape(3, 20)
            out['qm_' + tag] = qm
            out['qm_inv_' + tag] = qmi
    np.savez_compressed(os.path.join(G, 'encoder_params.npz'), **out)


def pvq_theta_decisions():
    """Keyframe-luma pvq_theta outcomes with a null reference (no-ref branch) and a
    freshly reset adaptation context, using the real QM of -v 20."""
    prm = np.load(os.path.join(G, 'encoder_params.npz'))
    recs = {'x0': [], 'n': [], 'bs': [], 'band': [], 'q': [], 'beta': [], 'qg': [], 'k': [],
            'y': [], 'out': [], 'off': []}
    offs = {0: [1, 16], 1: [1, 16, 24, 32, 64], 2: [1, 16, 24, 32, 64, 96, 128, 256],
            3: [1, 16, 24, 32, 64, 96, 128, 256, 384, 512]}
    for masking in (0, 1):
        tag = 'q20_m%d' % masking
        qm_all, qmi_all = prm['qm_' + tag], prm['qm_inv_' + tag]
        q0 = int(prm['quantizer_' + tag][0])
        pq = prm['pvq_qm_q4_' + tag][0]
        for bs in range(4):
            off = offs[bs]
            for band in range(len(off) - 1):
                n = off[band + 1] - off[band]
                qm = np.ascontiguousarray(qm_all[bs*2048 + off[band]:bs*2048 + off[band] + n])
                qmi = np.ascontiguousarray(qmi_all[bs*2048 + off[band]:bs*2048 + off[band] + n])
                idx = bs*(bs + 1) + (band + 1) - (band + 1)//3
                q = max(1, q0*int(pq[idx]) >> 4)
                beta = 1.5 if (masking and bs > 0) else 1.0
                for amp in (6, 60, 600):
                    x0 = (rng.laplace(0, amp, size=n)*np.exp(-np.arange(n)/(n/2.))).astype(np.int32)
                    out = np.zeros(n, np.int32); y = np.zeros(n, np.int32)
                    it = c_int(); mt = c_int(); vk = c_int(); sd = ctypes.c_double(0)
                    qg = pp.probe_pvq_theta(p32(out), p32(x0.copy()), p32(np.zeros(n, np.int32)), n,
                                            q, p32(y), ctypes.byref(it), ctypes.byref(mt),
                                            ctypes.byref(vk), beta, ctypes.byref(sd), 1, 1, 0, bs,
                                            p16(qm), p16(qmi))
                    pad = lambda a: np.concatenate([a, np.zeros(128 - n, np.int32)])  # noqa
                    recs['x0'].append(pad(x0)); recs['y'].append(pad(y)); recs['out'].append(pad(out))
                    recs['n'].append(n); recs['bs'].append(bs); recs['band'].append(band)
                    recs['q'].append(q); recs['beta'].append(beta); recs['qg'].append(qg)
                    recs['k'].append(vk.value); recs['off'].append(off[band])
                    recs.setdefault('masking', []).append(masking)
    np.savez_compressed(os.path.join(G, 'pvq_theta_noref.npz'),
                        **{k: np.array(v) for k, v in recs.items()})


def e2e_anchors():
    """Whole-encoder anchors: packet bytes + FNV-1a of the reference bitstream for a
    seeded synthetic CIF frame."""
    w, h = 352, 288
    fr = np.concatenate([synth_plane(w, h, 1).ravel(), synth_plane(w//2, h//2, 1, 1).ravel(),
                         synth_plane(w//2, h//2, 2, 1).ravel()])
    lines = []
    for quant, masking in ((20, 1), (20, 0), (5, 1)):
        fnv = ctypes.c_uint(); sec = ctypes.c_double()
        ep.probe_encode_frames.restype = ctypes.c_long
        nbytes = ep.probe_encode_frames(w, h, 1, quant, 7, masking, 1, pu8(fr), ctypes.byref(fnv),
                                        ctypes.byref(sec), None, 0)
        lines.append('cif synth seed1 q=%d masking=%d complexity=7: bytes=%d fnv1a=%08x'
                     % (quant, masking, nbytes, fnv.value))
    with open(os.path.join(G, 'e2e_anchors.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


def compute_dist_vectors():
    """od_compute_dist (static, src/encode.c:1032) through enc_probe, plus the
    per-coefficient weights it derives from the reference tables."""
    ep.probe_compute_dist.restype = ctypes.c_double
    r2 = np.random.default_rng(77)
    out = {}
    for bs in (1, 2, 3):
        n = 4 << bs
        mag = np.zeros(64)
        ep.probe_dist_weights(bs, pf64(mag))
        out['mag2_%d' % bs] = mag
        x = r2.integers(-2000, 2001, size=(24, n, n), dtype=np.int32)
        y = (x + r2.integers(-80, 81, size=x.shape)).astype(np.int32)
        y[:4] = x[:4]
        for m in (0, 1):
            out['dist_%d_m%d' % (bs, m)] = np.array([
                ep.probe_compute_dist(m, p32(np.ascontiguousarray(x[i])),
                                      p32(np.ascontiguousarray(y[i])), n, bs) for i in range(len(x))])
        out['x_%d' % bs] = x
        out['y_%d' % bs] = y
    np.savez_compressed(os.path.join(G, 'compute_dist.npz'), **out)


def mc_blocks():
    """F3: od_mc_predict (reference, through a live context's C vtable) for a list of
    prediction blocks covering a 96x64 picture area: inputs (3 padded reference planes, the
    block list) and the predicted plane."""
    U8P, I32P = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int32)
    g = np.random.default_rng(424242)
    pad, W, H = 40, 96, 64
    refs = np.stack([synth_plane(W + 2*pad, H + 2*pad, 70 + k) for k in range(3)])
    refs[1, ::5] = 255 - refs[1, ::5]//3
    rw = refs.shape[2]
    blocks = []
    # a quadtree-ish cover: 32x32 cells, some split down to 4x4
    def cover(x, y, lg):
        if lg > 2 and g.random() < (.55 if lg > 3 else .35):
            h = 1 << (lg - 1)
            for dy in (0, h):
                for dx in (0, h):
                    cover(x + dx, y + dy, lg - 1)
            return
        one = g.random() < .4
        ks = [int(g.integers(0, 3))]*4 if one else [int(v) for v in g.integers(0, 3, size=4)]
        mvx = g.integers(-8*(pad - 6), 8*(pad - 6), size=4)
        mvy = g.integers(-8*(pad - 6), 8*(pad - 6), size=4)
        if g.random() < .3:
            mvx &= ~7
        if g.random() < .3:
            mvy &= ~7
        if g.random() < .25:
            mvx[:] = mvx[0]
            mvy[:] = mvy[0]
        blocks.append([x, y, lg, lg] + ks + [int(v) for v in mvx] + [int(v) for v in mvy]
                      + [int(g.integers(0, 4)), int(g.integers(0, 4))])
    for y in range(0, H, 32):
        for x in range(0, W, 32):
            cover(x, y, 5)
    blocks = np.array(blocks, np.int32)
    dst = np.zeros((H, W), np.uint8)
    for b in blocks:
        x, y, lx, ly = (int(v) for v in b[:4])
        n, m = 1 << lx, 1 << ly
        ks, mvx, mvy = b[4:8], np.ascontiguousarray(b[8:12]), np.ascontiguousarray(b[12:16])
        out = np.zeros((m, n), np.uint8)
        at = [ctypes.cast(refs[int(k)].ctypes.data + (pad + y)*rw + pad + x, U8P) for k in ks]
        assert ep.probe_mc_predict(pu8(out), n, at[0], at[1], at[2], at[3], rw, mvx.ctypes.data_as(I32P),
                                   mvy.ctypes.data_as(I32P), int(b[16]), int(b[17]), lx, ly) == 0
        dst[y:y + m, x:x + n] = out
    np.savez_compressed(os.path.join(G, 'mc_blocks.npz'), refs=refs, pad=np.int32(pad), blocks=blocks,
                        dst=dst)
    print('mc_blocks:', len(blocks), 'blocks')


def mc_sad_pairs():
    """F3: the reference's SAD / SATD C entries (od_mc_compute_sad8_NxN_c, od_mc_compute_satd8_NxN_c,
    src/mcenc.c:1349-1372, :1562-1612) on a list of block pairs of two 160x128 planes."""
    U8P = ctypes.POINTER(ctypes.c_uint8)
    g = np.random.default_rng(5150)
    W, H = 160, 128
    src = synth_plane(W, H, 91)
    rf = synth_plane(W, H, 92)
    rf[40:90, 30:120] = np.clip(src[38:88, 33:123].astype(np.int32) + g.integers(-6, 7, size=(50, 90)), 0, 255)
    rf[::7] = 255 - rf[::7]
    pairs, want = [], []
    for i in range(600):
        lg = int(g.integers(2, 7))
        n = 1 << lg
        satd = int(g.integers(0, 2))
        sx, sy = int(g.integers(0, W - n + 1)), int(g.integers(0, H - n + 1))
        rx, ry = int(g.integers(0, W - n + 1)), int(g.integers(0, H - n + 1))
        f = getattr(r, 'od_mc_compute_%s8_%dx%d_c' % ('satd' if satd else 'sad', n, n))
        f.restype = ctypes.c_int32
        want.append(f(ctypes.cast(src.ctypes.data + sy*W + sx, U8P), W,
                      ctypes.cast(rf.ctypes.data + ry*W + rx, U8P), W))
        pairs.append([sx, sy, rx, ry, lg, satd])
    np.savez_compressed(os.path.join(G, 'mc_sad_pairs.npz'), src=src, ref=rf,
                        pairs=np.array(pairs, np.int32), out=np.array(want, np.int32))
    print('mc_sad_pairs:', len(pairs), 'pairs')


def mvest_sads():
    """F3, second half: the reference's od_mv_est_calc_sads (src/mcenc.c:3761) run by
    oracle/ref_probe/mcenc_probe.c on the motion estimation context of a real encoder after an
    I P P stream (176x144, moving content): what the call read - the vector grid (1/8-sample
    vectors, image index per vertex), every reference image of every plane with its padding, the
    encoder's padded input frame - and what it wrote, sad_cache of both evaluated block sizes."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    sys.path.insert(0, ROOT)
    from configs_round import frames_of
    import daala_amd.hipenc as H
    mp = ref('mcenc_probe')
    U8P, I32P = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int32)
    w, h, nf = 176, 144, 3
    fr = frames_of(w, h, nf, 77, step=(2, 3))
    g = np.random.default_rng(99)
    for f in range(nf):       # local motion on top of the global pan, and some noise
        fr[f] = [p.copy() for p in fr[f]]
        fr[f][0][40:90, 60 + 5*f:120 + 5*f] = fr[0][0][30:80, 20:80]
        fr[f][0] = np.clip(fr[f][0].astype(np.int32) + g.integers(-3, 4, size=fr[f][0].shape), 0, 255).astype(np.uint8)
    buf = H.pack_frames(fr, w, h)
    assert mp.probe_mvest_open(w, h, nf, 20, 30, pu8(buf)) == 0
    dims = np.zeros(19, np.int32)
    assert mp.probe_mvest_dims(p32(dims)) == 0
    nh, nv, nimg, fw, fh = int(dims[0]), int(dims[1]), int(dims[4]), int(dims[5]), int(dims[6])
    gmvx = np.zeros((nv + 1, nh + 1), np.int32)
    gmvy = np.zeros_like(gmvx)
    gref = np.zeros_like(gmvx)
    refs = [np.zeros((nimg, int(dims[8 + 4*p]), int(dims[7 + 4*p])), np.uint8) for p in range(3)]
    src = [np.zeros((fh >> (p > 0), fw >> (p > 0)), np.uint8) for p in range(3)]
    sad = [np.zeros((nv >> l, nh >> l, 4), np.int32) for l in range(3)]
    assert mp.probe_mvest_get(p32(gmvx), p32(gmvy), p32(gref), (U8P*3)(*[pu8(a) for a in refs]),
                              (U8P*3)(*[pu8(a) for a in src]), (I32P*3)(*[p32(a) for a in sad])) == 0
    # od_mv_est_bma_sad (the real static function, same state): block-matching SADs of blocks centred
    # on grid vertices - also on the frame's edges, where the block hangs over it - for half-sample
    # vectors around the vertex's own vector, both reference frame types
    mp.probe_mvest_bma_sad.restype = ctypes.c_int32
    req, want = [], []
    for i in range(400):
        lg = int(g.integers(0, 4))                       # log_mvb_sz: 8x8 .. 64x64
        n = 8 << lg
        vx = int(g.integers(0, nh + 1))//(1 << lg)*(1 << lg)
        vy = int(g.integers(0, nv + 1))//(1 << lg)*(1 << lg)
        rtype = int(g.integers(0, 2))
        bx, by = vx*8 - n//2, vy*8 - n//2
        mvx = int(gmvx[vy, vx])//4 + int(g.integers(-6, 7))
        mvy = int(gmvy[vy, vx])//4 + int(g.integers(-6, 7))
        img = ctypes.c_int()
        v = mp.probe_mvest_bma_sad(rtype, bx, by, mvx, mvy, lg, ctypes.byref(img))
        if img.value < 0:
            continue
        req.append([bx, by, lg + 3, img.value, mvx, mvy])
        want.append(v)
    mp.probe_mvest_close()
    assert len(np.unique(gref)) >= 2 and np.any(gmvx & 7) and np.any(gmvy & 7), 'grid too tame'
    assert len(req) > 300 and any(r[0] < 0 for r in req) and any(r[1] < 0 for r in req)
    np.savez_compressed(os.path.join(G, 'mvest_sads.npz'), dims=dims, pic=np.array([w, h], np.int32), gmvx=gmvx,
                        gmvy=gmvy, gref=gref, refs0=refs[0], refs1=refs[1], refs2=refs[2], src0=src[0],
                        src1=src[1], src2=src[2], sad1=sad[1], sad2=sad[2], bma_req=np.array(req, np.int32),
                        bma_sad=np.array(want, np.int32))
    print('mvest_sads: grid', gmvx.shape, 'images used', np.unique(gref), 'sad1', sad[1].shape, 'sad2', sad[2].shape)


def dcttest_md5():
    out = subprocess.run([os.path.join(ROOT, 'oracle', '_ref', 'dcttest')], capture_output=True)
    assert out.returncode == 0
    md5 = hashlib.md5(out.stdout).hexdigest()
    with open(os.path.join(G, 'dcttest.md5'), 'w') as f:
        f.write('%s  dcttest stdout (reference src/dct.c:2192-3951, -DOD_DCT_TEST '
                '-DOD_DCT_CHECK_OVERFLOW), %d lines\n' % (md5, out.stdout.count(b'\n')))
    print(md5)


if __name__ == '__main__':
    only = [a for a in sys.argv[1:] if not a.startswith('--')]
    if only:       # python tools/gen_golden.py mvest_sads: just the named fixtures
        for name in only:
            globals()[name]()
        sys.exit(0)
    dct_vectors()
    filter_vectors()
    filter_n_vectors()
    cfl_decimations()
    plane_forward()
    pvq_vectors()
    encoder_params()
    pvq_theta_decisions()
    e2e_anchors()
    compute_dist_vectors()
    mc_blocks()
    mc_sad_pairs()
    mvest_sads()
    if '--dcttest' in sys.argv:
        dcttest_md5()
    print('golden fixtures written to', G)
