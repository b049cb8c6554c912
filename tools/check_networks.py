"""Dev check (needs oracle/_ref): interpret the lifting networks with numpy and
compare 1-D forward and mechanically-derived inverse against the compiled
reference od_bin_fdctN / od_bin_idctN."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
from lifting_networks import NETWORKS, inverse_steps

def ev(r, e):
    if e[0] == 'r': return r[e[1]]
    if e[0] == 'h':
        a = r[e[1]]
        return (a + ((a.view(np.uint32) >> 31).astype(np.int32))) >> 1
    if e[0] == 'm':
        return (r[e[1]] * np.int32(e[2]) + np.int32(1 << (e[3] - 1))) >> e[3]

def run(steps, r):
    for st in steps:
        if st[0] == 'neg': r[st[1]] = -r[st[1]]
        elif st[0] == 'add': r[st[1]] = r[st[1]] + ev(r, st[2])
        elif st[0] == 'sub': r[st[1]] = r[st[1]] - ev(r, st[2])
        elif st[0] == 'rsb': r[st[1]] = ev(r, st[2]) - r[st[1]]
    return r

def fwd(net, x):   # x: [n, batch]
    r = [None] * net.n
    for i, p in enumerate(net.perm): r[p] = x[i].copy()
    return np.stack(run(net.steps, r))

def inv(net, y):
    r = run(inverse_steps(net.steps), [y[i].copy() for i in range(net.n)])
    return np.stack([r[p] for p in net.perm])

if __name__ == '__main__':
    ref = ctypes.CDLL(os.path.join(os.path.dirname(__file__), '..', 'oracle', '_ref', 'libdaala_ref.so'))
    rng = np.random.default_rng(1)
    P = ctypes.POINTER(ctypes.c_int32)
    for n, build in NETWORKS.items():
        net = build()
        B = 20000
        for amp in (300, 5000, 1 << 20):
            x = rng.integers(-amp, amp, size=(n, B), dtype=np.int32)
            y = fwd(net, x)
            yr = np.zeros((B, n), np.int32); xr = np.zeros((B, n), np.int32)
            xt = np.ascontiguousarray(x.T)
            f = getattr(ref, 'od_bin_fdct%d' % n); g = getattr(ref, 'od_bin_idct%d' % n)
            for b in range(B):
                f(yr[b].ctypes.data_as(P), xt[b].ctypes.data_as(P), 1)
                g(xr[b].ctypes.data_as(P), 1, xt[b].ctypes.data_as(P))
            assert np.array_equal(y.T, yr), (n, amp, 'fwd')
            assert np.array_equal(inv(net, x).T, xr), (n, amp, 'inv')
            assert np.array_equal(inv(net, y), x)
        print('N=%d ok: %d steps' % (n, len(net.steps)))
