#!/usr/bin/env python3
"""Sound worst-case magnitude bounds for every multiply operand of the lifting
DCTs, to justify the 24-bit integer multiplier (v_mad_i32_i24: full rate on
gfx950, while v_mul_lo_u32 is quarter rate) in the pixel-driven forward kernels.

Method: every register is tracked as an exact linear form in the N inputs plus an
accumulated rounding-error bound (each shift/rounding contributes <= 1).  For
inputs bounded by B, |register| <= L1(coefficients)*B + err.  mul24 is
bit-identical to the reference's wrapping int32 arithmetic iff the multiplied
register fits in a signed 24-bit integer (the constants are <= 2^15)."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lifting_networks import NETWORKS, inverse_steps  # noqa: E402


def analyse(steps, n, perm_in, B):
    """perm_in[i] = register holding input i at the start (None: register k holds
    input k).  Returns (max |mul operand|, max |output|)."""
    lin = [np.zeros(n) for _ in range(n)]
    err = [0.0]*n
    for i in range(n):
        lin[perm_in[i] if perm_in else i][i] = 1.0
    worst = 0.0

    def val(e):
        nonlocal worst
        if e[0] == 'r':
            return lin[e[1]].copy(), err[e[1]]
        if e[0] == 'h':
            return lin[e[1]]*0.5, err[e[1]]*0.5 + 1
        c = e[2]/float(1 << e[3])
        worst = max(worst, np.abs(lin[e[1]]).sum()*B + err[e[1]])
        return lin[e[1]]*c, err[e[1]]*c + 1

    for st in steps:
        x = st[1]
        if st[0] == 'neg':
            lin[x] = -lin[x]
            continue
        v, ve = val(st[2])
        if st[0] == 'add':
            lin[x] = lin[x] + v
        elif st[0] == 'sub':
            lin[x] = lin[x] - v
        else:
            lin[x] = v - lin[x]
        err[x] = err[x] + ve
    out = max(np.abs(lin[k]).sum()*B + err[k] for k in range(n))
    return worst, out


def lap4_gain():
    """L1 gain of od_pre_filter4 (src/filter.c:174-220) as a linear map + error."""
    x = np.eye(4)
    d3, d2 = x[0] - x[3], x[1] - x[2]
    s1, s0 = x[1] - d2/2, x[0] - d3/2
    d2, d3 = d2*85/64, d3*75/64
    d3 = d3 - d2*15/64
    d2 = d2 + d3*33/64
    s0, s1 = s0 + d3/2, s1 + d2/2
    rows = [s0, s1, s1 - d2, s0 - d3]
    return max(np.abs(r).sum() for r in rows), 8.0


def report():
    lines = []
    g, e = lap4_gain()
    pix = 2048.0                     # |(p - 128) << 4|
    # a sample is lapped at most once per axis: along one axis the frame filter
    # acts at SB-edge +-2 and the split filter of level k at (n_k/2) +-2, which are
    # disjoint positions; so at most two applications (one per axis)
    spatial = (pix*g + e)*g + e
    lines.append('lap4 L1 gain %.4f; spatial bound after lapping %.0f (2^%.2f)'
                 % (g, spatial, math.log2(spatial)))
    ok = True
    for n, build in sorted(NETWORKS.items()):
        net = build()
        w1, o1 = analyse(net.steps, n, net.perm, spatial)
        w2, o2 = analyse(net.steps, n, net.perm, o1)
        w = max(w1, w2)
        lines.append('fdct%-2d 2-D from pixels: max multiply operand 2^%.2f, max output 2^%.2f -> '
                     'mul24 %s' % (n, math.log2(w), math.log2(o2), 'SAFE' if w < 2**23 else 'UNSAFE'))
        ok = ok and w < 2**23
    for n, build in sorted(NETWORKS.items()):
        net = build()
        inv = inverse_steps(net.steps)
        lo, hi = 1.0, float(1 << 23)
        for _ in range(60):
            mid = (lo + hi)/2
            w1, o1 = analyse(inv, n, None, mid)
            w2, o2 = analyse(inv, n, None, o1)
            if max(w1, w2) < 2**23:
                lo = mid
            else:
                hi = mid
        lines.append('idct%-2d 2-D: mul24 safe while every |coefficient| <= %d (2^%.2f)'
                     % (n, int(lo), math.log2(lo)))
    return ok, lines


if __name__ == '__main__':
    ok, lines = report()
    print('\n'.join(lines))
    sys.exit(0 if ok else 1)
