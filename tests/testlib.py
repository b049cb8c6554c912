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
