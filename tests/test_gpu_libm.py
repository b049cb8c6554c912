"""Quantifies how the device's double-precision primitives relate to the host libm
the reference is linked against.  sqrt and divide MUST be bit-identical (IEEE
correctly rounded on both sides): every product kernel relies on it.  OCML's
pow/acos/sin/cos are sub-ulp accurate but NOT bit-identical to glibc's (which is not
correctly rounded either): that is why no product kernel calls them - the
transcendentals of the path are evaluated by the host's libm between device passes
(DESIGN.md section 5).  The measured disagreement is printed and bounded here."""
import ctypes
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def probe(fn, x, y=None):
    import daala_amd.binding as b
    lib = b.load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), dtype=np.float64)
    out = np.empty_like(x)
    P = ctypes.POINTER(ctypes.c_double)
    lib.od_hip_libm_probe.argtypes = [ctypes.c_int, ctypes.c_int, P, P, P]
    assert lib.od_hip_libm_probe(fn, len(x), x.ctypes.data_as(P), y.ctypes.data_as(P),
                                 out.ctypes.data_as(P)) == 0
    return out


def ulp_diff(a, b):
    return np.abs(a - b)/np.spacing(np.maximum(np.abs(a), np.abs(b)))


def test_sqrt_and_divide_are_correctly_rounded():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0, 1e12, 200000), rng.uniform(0, 4, 200000),
                        np.arange(1, 70000, dtype=np.float64), 10.0**rng.uniform(-300, 300, 50000)])
    assert np.array_equal(probe(4, x), np.sqrt(x))
    y = np.concatenate([rng.uniform(1e-3, 1e6, 400000), np.arange(1, 70000, dtype=np.float64),
                        10.0**rng.uniform(-100, 100, 50000)])
    assert np.array_equal(probe(5, x, y), x/y)


def test_transcendentals_agree_to_one_ulp():
    rng = np.random.default_rng(2)
    n = 200000
    # od_gain_compand: pow(g/4096, 1/1.5), g in a realistic range
    g = 10.0**rng.uniform(-2, 5, n)/4096.
    e = np.full(n, 1./1.5)
    host = np.array([math.pow(a, 1./1.5) for a in g])
    d = ulp_diff(probe(0, g, e), host)
    print('pow   : identical %.4f%%, max %g ulp' % (100*np.mean(d == 0), d.max()))
    assert d.max() <= 1
    c = rng.uniform(0, 1, n)
    host = np.array([math.acos(a) for a in c])
    d = ulp_diff(probe(1, c), host)
    print('acos  : identical %.4f%%, max %g ulp' % (100*np.mean(d == 0), d.max()))
    assert d.max() <= 1
    t = rng.uniform(-2, 2, n)
    for fn, f, name in ((2, math.sin, 'sin'), (3, math.cos, 'cos')):
        host = np.array([f(a) for a in t])
        d = ulp_diff(probe(fn, t), host)
        print('%-6s: identical %.4f%%, max %g ulp' % (name, 100*np.mean(d == 0), d.max()))
        assert d.max() <= 1
