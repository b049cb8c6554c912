"""Shared helpers for the test-suite: ctypes access to the CPU oracle
(oracle/libdaala_oracle.so, our restatement) and - when present - to the real
reference build (oracle/_ref/*.so).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, 'oracle')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
I32P = ctypes.POINTER(ctypes.c_int32)
I16P = ctypes.POINTER(ctypes.c_int16)
U8P = ctypes.POINTER(ctypes.c_uint8)
F64P = ctypes.POINTER(ctypes.c_double)
INTP = ctypes.POINTER(ctypes.c_int)
c_int = ctypes.c_int
c_double = ctypes.c_double


def p32(a):
    assert a.dtype == np.int32
    return a.ctypes.data_as(I32P)


def p16(a):
    assert a.dtype == np.int16
    return a.ctypes.data_as(I16P)


def pu8(a):
    assert a.dtype == np.uint8
    return a.ctypes.data_as(U8P)


def pf64(a):
    assert a.dtype == np.float64
    return a.ctypes.data_as(F64P)


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        so = os.path.join(ORACLE_DIR, 'libdaala_oracle.so')
        src = os.path.join(ORACLE_DIR, 'daala_oracle.c')
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(['make', '-C', ORACLE_DIR, 'libdaala_oracle.so'],
                                  stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(so)
        for name in ('orc_gain_compand', 'orc_pvq_compute_gain', 'orc_gain_expand',
                     'orc_pvq_compute_theta', 'orc_pvq_search_rdo_double'):
            getattr(lib, name).restype = c_double
        lib.orc_gain_compand.argtypes = [c_double, c_int, c_double]
        lib.orc_gain_expand.argtypes = [c_double, c_int, c_double]
        lib.orc_pvq_compute_gain.argtypes = [I32P, c_int, c_int, F64P, c_double, I16P]
        lib.orc_pvq_compute_max_theta.argtypes = [c_double, c_double]
        lib.orc_pvq_compute_theta.argtypes = [c_int, c_int]
        lib.orc_pvq_compute_k.argtypes = [c_double, c_int, c_double, c_int, c_int,
                                          c_double, c_int]
        lib.orc_pvq_search_rdo_double.argtypes = [F64P, c_int, c_int, I32P, c_double]
        lib.orc_pvq_synthesis_partial.argtypes = [I32P, I32P, F64P, c_int, c_int,
                                                  c_double, c_double, c_int, c_int, I16P]
        lib.orc_compute_householder.argtypes = [F64P, c_int, c_double, INTP]
        lib.orc_pvq_noref_candidates.argtypes = [I32P, c_int, c_int, c_double, I16P,
                                                 c_int, F64P, F64P, INTP, INTP, F64P,
                                                 F64P, I32P]
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(os.path.join(ORACLE_DIR, '_ref', 'libdaala_ref.so'))


_ref = {}


def ref(name='libdaala_ref'):
    """The REAL reference (or a probe built around it); None when not built."""
    if name not in _ref:
        so = os.path.join(ORACLE_DIR, '_ref', name + '.so')
        if not os.path.exists(so):
            _ref[name] = None
            return None
        lib = ctypes.CDLL(so)
        if name == 'libdaala_ref':
            lib.od_pvq_compute_gain.restype = c_double
            lib.od_pvq_compute_gain.argtypes = [I32P, c_int, c_int, F64P, c_double, I16P]
            lib.od_gain_expand.restype = c_double
            lib.od_gain_expand.argtypes = [c_double, c_int, c_double]
            lib.od_pvq_compute_theta.restype = c_double
            lib.od_pvq_compute_max_theta.argtypes = [c_double, c_double]
            lib.od_pvq_compute_k.argtypes = [c_double, c_int, c_double, c_int, c_int,
                                             c_double, c_int]
            lib.od_compute_householder.argtypes = [F64P, c_int, c_double, INTP]
            lib.od_pvq_synthesis_partial.argtypes = [I32P, I32P, F64P, c_int, c_int,
                                                     c_double, c_double, c_int, c_int,
                                                     I16P]
        if name == 'pvq_probe':
            lib.probe_pvq_search_rdo_double.restype = c_double
            lib.probe_pvq_search_rdo_double.argtypes = [F64P, c_int, c_int, I32P, c_double]
            lib.probe_pvq_rate_reset.restype = c_double
            lib.probe_pvq_rate_reset.argtypes = [c_int, c_int, c_int, c_int, I32P, c_int,
                                                 c_int, c_int, c_int, c_int]
            lib.probe_pvq_theta.argtypes = [I32P, I32P, I32P, c_int, c_int, I32P, INTP,
                                            INTP, INTP, c_double, F64P, c_int, c_int,
                                            c_int, c_int, I16P, I16P]
        _ref[name] = lib
    return _ref[name]


# ---------------------------------------------------------------------------
# Deterministic synthetic content (SURVEY.md section 8d recipe): smooth sinusoid
# + sharp rectangles + textured region + diagonal ramp, so all block sizes occur.
def synth_plane(w, h, seed, chroma=0):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    sc = 2.0 if chroma else 1.0
    v = 128 + 50*np.sin((xx*sc + 3*seed)*0.021*(1 + chroma))*np.cos(yy*sc*0.016)
    v += 18*np.sin((xx + yy)*sc*0.11)
    cell = ((xx*sc)//64 + (yy*sc)//48).astype(np.int64) & 1
    v += np.where(cell == 1, 25, -25)
    v += (xx + yy)*sc*0.01
    rng = np.random.default_rng(1000*seed + chroma)
    amp = np.where((xx*sc > w*sc*0.5) & (yy*sc > h*sc*0.4), 28, 4)
    v += rng.integers(-1000, 1001, size=(h, w))*amp/1000.0
    return np.clip(np.floor(v + .5), 0, 255).astype(np.uint8)


def random_bsize_map(nhsb, nvsb, seed):
    """Random valid luma block-size map: 1 byte per 8x8, quadtree-consistent."""
    rng = np.random.default_rng(seed)
    m = np.zeros((nvsb*4, nhsb*4), np.uint8)
    for sy in range(nvsb):
        for sx in range(nhsb):
            r = rng.integers(0, 4)
            if r == 3:
                m[sy*4:sy*4 + 4, sx*4:sx*4 + 4] = 3
                continue
            for qy in range(2):
                for qx in range(2):
                    r2 = rng.integers(0, 3)
                    y0, x0 = sy*4 + qy*2, sx*4 + qx*2
                    if r2 == 2:
                        m[y0:y0 + 2, x0:x0 + 2] = 2
                    else:
                        m[y0:y0 + 2, x0:x0 + 2] = rng.integers(0, 2, size=(2, 2))
    return m


# ---------------------------------------------------------------------------
class ThetaOut(ctypes.Structure):
    """Mirror of orc_theta_out (oracle/daala_oracle.c) and od_hip_pvq_theta_out."""
    _fields_ = [('cg', c_double), ('cgr', c_double), ('g', c_double), ('gr', c_double),
                ('corr', c_double), ('theta', c_double), ('gain_offset', c_double),
                ('skip_dist', c_double), ('null_dist', c_double),
                ('icgr', ctypes.c_int32), ('m', ctypes.c_int32), ('s', ctypes.c_int32),
                ('nref', ctypes.c_int32), ('nnoref', ctypes.c_int32),
                ('theta_searched', ctypes.c_int32), ('noref_searched', ctypes.c_int32),
                ('pad', ctypes.c_int32),
                ('ref_qg', ctypes.c_int32*12), ('ref_itheta', ctypes.c_int32*12),
                ('ref_ts', ctypes.c_int32*12), ('ref_k', ctypes.c_int32*12),
                ('ref_qtheta', c_double*12), ('ref_cos_dist', c_double*12),
                ('ref_dist', c_double*12),
                ('nr_qg', ctypes.c_int32*2), ('nr_k', ctypes.c_int32*2),
                ('nr_cos_dist', c_double*2), ('nr_dist', c_double*2)]


def neg_interleave(x, r):
    if x < r:
        return -2*(x - r) - 1
    if x < 2*r:
        return 2*(x - r)
    return x - 1


def decide_pvq_theta(o, rate_fn, t, y_ref, y_noref, x0, r0, n, q0, beta, is_keyframe, pli, qm,
                     qm_inv):
    """Replays the decision part of pvq_theta (reference src/pvq_encoder.c:373-511)
    on the state-free candidates `t` (ThetaOut) plus a rate callback
    rate_fn(qg, icgr, theta, ts, y, k).  Returns (ret, itheta, max_theta, k, y, out)."""
    lam = .147
    gw = 1.4
    cfl = bool(is_keyframe and pli != 0)
    qg = 0
    best_dist = gw*t.cg*t.cg
    best_cost = best_dist + lam*rate_fn(0, 0, -1, 0, None, 0)
    noref, best_k, itheta, max_theta, best_qtheta = 1, 0, -1, 0, 0.
    y = np.zeros(n, np.int32)
    if not is_keyframe:
        scgr = max(0., t.gain_offset)
        if t.icgr == 0:
            best_dist = gw*(t.cg - scgr)*(t.cg - scgr) + scgr*t.cg*(2 - 2*t.corr)
        best_cost = best_dist + lam*rate_fn(0, t.icgr, 0, 0, None, 0)
        itheta, max_theta, noref = 0, 0, 0
    for c in range(t.nref):
        cost = t.ref_dist[c] + lam*rate_fn(t.ref_qg[c], t.icgr, t.ref_itheta[c], t.ref_ts[c],
                                            y_ref[c], t.ref_k[c])
        if cost < best_cost:
            best_cost, best_dist = cost, t.ref_dist[c]
            qg, best_k, best_qtheta = t.ref_qg[c], t.ref_k[c], t.ref_qtheta[c]
            itheta, max_theta, noref = t.ref_itheta[c], t.ref_ts[c], 0
            y = y_ref[c].copy()
            y[n - 1:] = 0
    for c in range(t.nnoref):
        cost = t.nr_dist[c] + lam*rate_fn(t.nr_qg[c], 0, -1, 0, y_noref[c], t.nr_k[c])
        if cost <= best_cost:
            best_cost, best_dist = cost, t.nr_dist[c]
            qg, best_k, noref, itheta, max_theta = t.nr_qg[c], t.nr_k[c], 1, -1, 0
            y = y_noref[c].copy()
    skip = 0
    if noref:
        if qg == 0:
            skip = 1
    else:
        if not is_keyframe and qg == 0:
            skip = 1 if t.icgr else 2
        if qg == t.icgr and itheta == 0 and not cfl:
            skip = 2
    out = np.zeros(n, np.int32)
    if skip == 2:
        out = r0.copy()
    elif skip == 0:
        go = 0. if noref else t.gain_offset
        g = o.orc_gain_expand(qg + go, q0, beta)
        o.orc_pvq_synthesis.argtypes = [I32P, I32P, I32P, c_int, c_double, c_int, c_double,
                                        c_double, I16P, I16P]
        o.orc_pvq_synthesis(p32(out), p32(np.ascontiguousarray(y)), p32(r0), n, t.gr, noref, g,
                            best_qtheta, p16(qm), p16(qm_inv))
    if is_keyframe:
        ret = qg if noref else neg_interleave(qg, t.icgr)
    else:
        ret = qg - 1 if noref else neg_interleave(qg + 1, t.icgr + 1)
    decide_pvq_theta.best_dist = best_dist          # for the skip_diff identity
    return ret, itheta, max_theta, best_k, y, out


# ---------------------------------------------------------------------------
# F3, second half: od_mv_est_calc_sads through the oracle (orc_mv_est_calc_sads_items +
# orc_mv_est_sad_items) on a fixture of the mcenc probe's layout (tests/golden/mvest_sads.npz).
SAD_ITEM = np.dtype([('x', np.int32), ('y', np.int32), ('log_blk_sz', np.int32), ('oc', np.int32),
                     ('s', np.int32), ('ref', np.int32, 4), ('mvx', np.int32, 4), ('mvy', np.int32, 4),
                     ('reserved', np.int32)])


def mvest_items(o, g):
    """The item list of od_mv_est_calc_sads for the fixture's grid, and per block size the
    (count, smax) the reference's loops give."""
    dims = g['dims']
    nh, nv, lmin, lmax = (int(v) for v in dims[:4])
    gx, gy, gr = (np.ascontiguousarray(g[k], dtype=np.int32) for k in ('gmvx', 'gmvy', 'gref'))
    sizes, smax = np.zeros(3, np.int32), np.zeros(3, np.int32)
    n = o.orc_mv_est_calc_sads_items(nh, nv, lmin, lmax, p32(gx), p32(gy), p32(gr), None, p32(sizes), p32(smax))
    items = np.zeros(n, SAD_ITEM)
    assert o.orc_mv_est_calc_sads_items(nh, nv, lmin, lmax, p32(gx), p32(gy), p32(gr),
                                        ctypes.c_void_p(items.ctypes.data), p32(sizes), p32(smax)) == n
    return items, sizes, smax


def mvest_oracle_sads(o, g, items, nplanes=3):
    dims = g['dims']
    U8P = ctypes.POINTER(ctypes.c_uint8)
    refs = [np.ascontiguousarray(g['refs%d' % p]) for p in range(3)]
    src = [np.ascontiguousarray(g['src%d' % p]) for p in range(3)]
    rs = np.array([dims[7 + 4*p] for p in range(3)], np.int32)
    rh = np.array([dims[8 + 4*p] for p in range(3)], np.int32)
    ox = np.array([dims[9 + 4*p] for p in range(3)], np.int32)
    oy = np.array([dims[10 + 4*p] for p in range(3)], np.int32)
    ss = np.array([s.shape[1] for s in src], np.int32)
    dec = np.array([0, 1, 1], np.int32)
    out = np.zeros(len(items), np.int32)
    o.orc_mv_est_sad_items(ctypes.c_void_p(items.ctypes.data), len(items), nplanes, (U8P*3)(*[pu8(r) for r in refs]),
                           p32(rs), p32(rh), p32(ox), p32(oy), (U8P*3)(*[pu8(s) for s in src]), p32(ss),
                           p32(dec), p32(dec), int(g['pic'][0]), int(g['pic'][1]), p32(out))
    return out


def mvest_split(sad, sizes, smax, dims):
    """Item-order SADs -> {log_mvb_sz: [rows, cols, smax]} like est->sad_cache."""
    nh, nv = int(dims[0]), int(dims[1])
    out, o = {}, 0
    for l in range(3):
        if sizes[l]:
            out[l] = sad[o:o + sizes[l]].reshape(nv >> l, nh >> l, smax[l])
            o += int(sizes[l])
    return out


BMA_REC = np.dtype([('bx', np.int32), ('by', np.int32), ('log_blk_sz', np.int32), ('ref', np.int32),
                    ('cx', np.int32), ('cy', np.int32), ('xmin', np.int32), ('xmax', np.int32),
                    ('ymin', np.int32), ('ymax', np.int32)])


def mvest_oracle_bma_windows(o, g, recs, radius, nplanes=3):
    """orc_mv_est_bma_windows on a fixture of the mcenc probe's layout: [nrec][(2R+1)^2]."""
    dims = g['dims']
    U8P = ctypes.POINTER(ctypes.c_uint8)
    refs = [np.ascontiguousarray(g['refs%d' % p]) for p in range(3)]
    src = [np.ascontiguousarray(g['src%d' % p]) for p in range(3)]
    rs = np.array([dims[7 + 4*p] for p in range(3)], np.int32)
    rh = np.array([dims[8 + 4*p] for p in range(3)], np.int32)
    ox = np.array([dims[9 + 4*p] for p in range(3)], np.int32)
    oy = np.array([dims[10 + 4*p] for p in range(3)], np.int32)
    ss = np.array([s.shape[1] for s in src], np.int32)
    dec = np.array([0, 1, 1], np.int32)
    W = 2*radius + 1
    out = np.zeros((len(recs), W*W), np.int32)
    recs = np.ascontiguousarray(recs, dtype=BMA_REC)
    o.orc_mv_est_bma_windows(ctypes.c_void_p(recs.ctypes.data), len(recs), radius, nplanes,
                             (U8P*3)(*[pu8(r) for r in refs]), p32(rs), p32(rh), p32(ox), p32(oy),
                             (U8P*3)(*[pu8(s) for s in src]), p32(ss), p32(dec), p32(dec),
                             int(g['pic'][0]), int(g['pic'][1]), p32(out))
    return out
