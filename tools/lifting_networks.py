"""Lifting-network description of Daala's reversible integer DCTs (4/8/16/32 point).

This is OUR representation of the transforms: each 1-D transform is a list of
primitive, individually invertible integer "lifting" steps over a register file
r[0..N-1].  The forward step list is transcribed from the arithmetic of the
reference (file:line cited per builder below); the INVERSE transform is not
transcribed at all - it is derived mechanically by walking the step list
backwards and inverting every step.  Because every step is a bijection on
Z^N (mod 2^32) the inverse of the network is unique, hence equal to the
reference's hand-written od_bin_idct* as a function (checked bit-exactly against
the compiled reference in tests/ and tools/check_networks.py).

Step forms (x = register being updated, e = expression not involving x):
    ('add', x, e)   x += e
    ('sub', x, e)   x -= e
    ('rsb', x, e)   x  = e - x        (self-inverse)
    ('neg', x)      x  = -x           (self-inverse)
Expressions:
    ('r', a)            register a
    ('h', a)            OD_DCT_RSHIFT(r[a], 1): halve, rounding toward zero
                        (reference src/filter.h:40-43)
    ('m', a, C, S)      (r[a]*C + (1 << (S-1))) >> S   (arithmetic shift, int32)

Register index k of an N-point network holds output coefficient k at the end
(the reference names them t0..tv in base 32 - src/dct.c:1757-1857).  `perm[i]`
is the register that input sample i is loaded into.
"""


def reg(name):
    """'t0'..'tv' -> 0..31 (base-32 digit, the reference's naming)."""
    return int(name[1:], 32)


class Half(object):
    """A captured OD_DCT_RSHIFT(r[a],1) value; remembers the version of r[a]."""

    def __init__(self, a, version):
        self.a = a
        self.version = version


class Net(object):
    def __init__(self, n):
        self.n = n
        self.steps = []
        self.version = [0] * n
        self.perm = None

    # -- expression helpers -------------------------------------------------
    def H(self, name):
        a = reg(name)
        return Half(a, self.version[a])

    def _expr(self, e, x):
        if isinstance(e, str):
            a = reg(e)
            assert a != x
            return ('r', a)
        if isinstance(e, Half):
            # The halved value must still be recomputable from the live register
            # when the step executes (so the derived inverse can recompute it).
            assert self.version[e.a] == e.version, 'stale half of r%d' % e.a
            assert e.a != x
            return ('h', e.a)
        if isinstance(e, tuple) and e[0] == 'm':
            a = reg(e[1])
            assert a != x
            return ('m', a, e[2], e[3])
        raise ValueError(e)

    def _push(self, op, name, e=None):
        x = reg(name)
        if op == 'neg':
            self.steps.append(('neg', x))
        else:
            self.steps.append((op, x, self._expr(e, x)))
        self.version[x] += 1

    def add(self, x, e):
        self._push('add', x, e)

    def sub(self, x, e):
        self._push('sub', x, e)

    def rsb(self, x, e):
        self._push('rsb', x, e)

    def neg(self, x):
        self._push('neg', x)

    def load_order(self, names):
        self.perm = [reg(nm) for nm in names]
        assert sorted(self.perm) == list(range(self.n))


def M(name, c, s):
    return ('m', name, c, s)


# ---------------------------------------------------------------------------
# 4-point: reference od_bin_fdct4, src/dct.c:74-112
def build_fdct4():
    n = Net(4)
    n.load_order(['t0', 't2', 't1', 't3'])
    n.rsb('t3', 't0')
    n.add('t2', 't1')
    t2h = n.H('t2')
    n.rsb('t1', t2h)
    n.sub('t0', n.H('t3'))
    n.add('t0', t2h)
    n.rsb('t2', 't0')
    n.sub('t3', M('t1', 23013, 15))
    n.add('t1', M('t3', 21407, 15))
    n.sub('t3', M('t1', 18293, 14))
    return n


# 8-point: reference od_bin_fdct8, src/dct.c:151-269
def build_fdct8():
    n = Net(8)
    n.load_order(['t0', 't4', 't2', 't6', 't7', 't3', 't5', 't1'])
    n.rsb('t1', 't0')
    t1h = n.H('t1')
    n.sub('t0', t1h)
    n.add('t4', 't5')
    t4h = n.H('t4')
    n.sub('t5', t4h)
    n.rsb('t3', 't2')
    n.sub('t2', n.H('t3'))
    n.add('t6', 't7')
    t6h = n.H('t6')
    n.rsb('t7', t6h)
    # embedded 4-point DCT-II
    n.add('t0', t6h)
    n.rsb('t6', 't0')
    n.rsb('t2', t4h)
    n.rsb('t4', 't2')
    n.sub('t0', M('t4', 13573, 15))
    n.add('t4', M('t0', 11585, 14))
    n.sub('t0', M('t4', 13573, 15))
    n.sub('t6', M('t2', 21895, 15))
    n.add('t2', M('t6', 15137, 14))
    n.sub('t6', M('t2', 21895, 15))
    # embedded 4-point DST-IV
    n.add('t3', M('t5', 19195, 15))
    n.add('t5', M('t3', 11585, 14))
    n.sub('t3', M('t5', 7489, 13))
    n.rsb('t7', n.H('t5'))
    n.sub('t5', 't7')
    n.rsb('t3', t1h)
    n.sub('t1', 't3')
    n.add('t7', M('t1', 3227, 15))
    n.sub('t1', M('t7', 6393, 15))
    n.add('t7', M('t1', 3227, 15))
    n.add('t5', M('t3', 2485, 13))
    n.sub('t3', M('t5', 18205, 15))
    n.add('t5', M('t3', 2485, 13))
    return n


# 16-point: reference od_bin_fdct16, src/dct.c:349-640
def build_fdct16():
    n = Net(16)
    n.load_order(['t0', 't8', 't4', 'tc', 'te', 'ta', 't6', 't2',
                  't3', 'td', 't9', 'tf', 't1', 't7', 'tb', 't5'])
    n.rsb('t5', 't0')
    n.add('t8', 'tb')
    n.rsb('t7', 't4')
    n.add('tc', 't1')
    n.rsb('tf', 'te')
    n.add('ta', 't9')
    n.rsb('td', 't6')
    n.add('t2', 't3')
    n.sub('t0', n.H('t5'))
    t8h = n.H('t8')
    n.rsb('tb', t8h)
    n.sub('t4', n.H('t7'))
    tch = n.H('tc')
    n.rsb('t1', tch)
    n.sub('te', n.H('tf'))
    tah = n.H('ta')
    n.rsb('t9', tah)
    n.sub('t6', n.H('td'))
    t2h = n.H('t2')
    n.rsb('t3', t2h)
    # embedded 8-point DCT-II
    n.add('t0', t2h)
    n.rsb('t6', t8h)
    n.add('t4', tah)
    n.rsb('te', tch)
    n.rsb('t2', 't0')
    n.sub('t8', 't6')
    n.rsb('ta', 't4')
    n.sub('tc', 'te')
    # embedded 4-point DCT-II
    n.rsb('tc', 't0')
    n.add('t8', 't4')
    t8h = n.H('t8')
    n.rsb('t4', t8h)
    n.sub('t0', n.H('tc'))
    n.add('t0', t8h)
    n.rsb('t8', 't0')
    n.sub('tc', M('t4', 23013, 15))
    n.add('t4', M('tc', 10703, 14))
    n.sub('tc', M('t4', 9147, 13))
    # embedded 4-point DST-IV
    n.add('t6', M('ta', 13573, 15))
    n.sub('ta', M('t6', 11585, 14))
    n.add('t6', M('ta', 13573, 15))
    n.add('ta', 'te')
    n.add('t2', 't6')
    n.rsb('te', n.H('ta'))
    n.rsb('t6', n.H('t2'))
    n.add('te', M('t2', 2275, 11))
    n.sub('t2', M('te', 9041, 15))
    n.sub('te', M('t2', 2873, 11))
    n.sub('t6', M('ta', 8593, 14))
    n.add('ta', M('t6', 12873, 14))
    n.add('t6', M('ta', 7335, 15))
    # embedded 8-point DST-IV
    n.add('t3', M('t5', 1035, 11))
    n.sub('t5', M('t3', 14699, 14))
    n.sub('t3', M('t5', 851, 13))
    n.add('tb', M('td', 17515, 15))
    n.sub('td', M('tb', 20435, 14))
    n.add('tb', M('td', 4379, 14))
    n.add('t9', M('t7', 12905, 14))
    n.sub('t7', M('t9', 3363, 13))
    n.sub('t9', M('t7', 14101, 14))
    n.add('t1', M('tf', 5417, 13))
    n.sub('tf', M('t1', 23059, 14))
    n.add('t1', M('tf', 20055, 15))
    n.rsb('tf', 't3')
    n.add('td', 't9')
    tfh = n.H('tf')
    n.sub('t3', tfh)
    tdh = n.H('td')
    n.rsb('t9', tdh)
    n.add('t1', 't5')
    n.rsb('tb', 't7')
    t1h = n.H('t1')
    n.rsb('t5', t1h)
    tbh = n.H('tb')
    n.sub('t7', tbh)
    n.add('t3', tbh)
    n.rsb('t5', tdh)
    n.add('t9', tfh)
    n.rsb('t7', t1h)
    n.sub('tb', 't3')
    n.sub('td', 't5')
    n.rsb('tf', 't9')
    n.sub('t1', 't7')
    n.sub('t5', M('tb', 10947, 14))
    n.add('tb', M('t5', 15137, 14))
    n.sub('t5', M('tb', 10947, 14))
    n.add('td', M('t3', 21895, 15))
    n.sub('t3', M('td', 15137, 14))
    n.add('td', M('t3', 10947, 14))
    n.sub('t1', M('tf', 13573, 15))
    n.add('tf', M('t1', 11585, 14))
    n.sub('t1', M('tf', 13573, 15))
    return n


# ---------------------------------------------------------------------------
# 32-point: built from nested sub-networks, reference src/dct.c:790-1443 (the
# forward halves of the OD_F* macro family) and OD_FDCT_32 :1638-1701.
# Each helper takes register NAMES (and captured halves) so the same
# sub-network can be instantiated on any register subset.

def _fdct2(n, a, b):          # src/dct.c:790-803
    n.sub(a, M(b, 13573, 15))
    n.add(b, M(a, 5793, 13))
    n.sub(a, M(b, 3393, 13))


def _fdst2(n, a, b):          # src/dct.c:817-830
    n.sub(a, M(b, 10947, 14))
    n.add(b, M(a, 473, 9))
    n.sub(a, M(b, 10947, 14))


def _fdct4_asym(n, t0, t2, t2h, t1, t3, t3h):     # src/dct.c:844-854
    n.add(t0, t3h)
    n.rsb(t3, t0)
    n.rsb(t1, t2h)
    n.rsb(t2, t1)
    _fdct2(n, t0, t2)
    _fdst2(n, t3, t1)


def _fdst4_asym(n, t0, t0h, t2, t1, t3):          # src/dct.c:870-905
    n.sub(t2, M(t1, 7489, 13))
    n.add(t1, M(t2, 11585, 14))
    n.add(t2, M(t1, 19195, 15))
    n.add(t3, n.H(t2))
    n.sub(t2, t3)
    n.rsb(t1, t0h)
    n.sub(t0, t1)
    n.add(t3, M(t0, 6723, 13))
    n.sub(t0, M(t3, 8035, 13))
    n.add(t3, M(t0, 6723, 13))
    n.add(t2, M(t1, 8757, 14))
    n.sub(t1, M(t2, 6811, 13))
    n.add(t2, M(t1, 8757, 14))


def _fdct8(n, t0, t4, t2, t6, t1, t5, t3, t7):    # src/dct.c:936-956
    n.rsb(t7, t0)
    t7h = n.H(t7)
    n.sub(t0, t7h)
    n.add(t4, t3)
    t4h = n.H(t4)
    n.rsb(t3, t4h)
    n.rsb(t5, t2)
    n.sub(t2, n.H(t5))
    n.add(t6, t1)
    t6h = n.H(t6)
    n.rsb(t1, t6h)
    _fdct4_asym(n, t0, t4, t4h, t2, t6, t6h)
    _fdst4_asym(n, t7, t7h, t3, t5, t1)


def _fdst8(n, t0, t4, t2, t6, t1, t5, t3, t7):    # src/dct.c:977-1070
    n.sub(t6, M(t1, 13573, 15))
    n.add(t1, M(t6, 11585, 14))
    n.sub(t6, M(t1, 13573, 15))
    n.sub(t5, M(t2, 21895, 15))
    n.add(t2, M(t5, 15137, 14))
    n.sub(t5, M(t2, 10947, 14))
    n.sub(t4, M(t3, 3259, 14))
    n.add(t3, M(t4, 3135, 13))
    n.sub(t4, M(t3, 3259, 14))
    n.add(t7, t1)
    t7h = n.H(t7)
    n.sub(t1, t7h)
    n.rsb(t2, t3)
    t2h = n.H(t2)
    n.sub(t3, t2h)
    n.sub(t0, t6)
    t0h = n.H(t0)
    n.add(t6, t0h)
    n.rsb(t5, t4)
    t5h = n.H(t5)
    n.sub(t4, t5h)
    n.add(t1, t5h)
    n.rsb(t5, t1)
    n.add(t4, t0h)
    n.sub(t0, t4)
    n.sub(t6, t2h)
    n.add(t2, t6)
    n.sub(t3, t7h)
    n.add(t7, t3)
    n.neg(t7)
    n.sub(t0, M(t7, 7425, 13))
    n.add(t7, M(t0, 8153, 13))
    n.sub(t0, M(t7, 7425, 13))
    n.sub(t6, M(t1, 4861, 15))
    n.add(t1, M(t6, 1189, 12))
    n.sub(t6, M(t1, 4861, 15))
    n.sub(t2, M(t5, 2455, 12))
    n.add(t5, M(t2, 7225, 13))
    n.sub(t2, M(t5, 2455, 12))
    n.sub(t4, M(t3, 11725, 15))
    n.add(t3, M(t4, 5197, 13))
    n.sub(t4, M(t3, 11725, 15))


def _fdct16_asym(n, t0, t8, t8h, t4, tc, tch, t2, ta, tah, t6, te, teh,
                 t1, t9, t9h, t5, td, tdh, t3, tb, tbh, t7, tf, tfh):
    # src/dct.c:1146-1169
    n.add(t0, tfh)
    n.rsb(tf, t0)
    n.sub(t1, teh)
    n.add(te, t1)
    n.add(t2, tdh)
    n.rsb(td, t2)
    n.sub(t3, tch)
    n.add(tc, t3)
    n.add(t4, tbh)
    n.rsb(tb, t4)
    n.sub(t5, tah)
    n.add(ta, t5)
    n.add(t6, t9h)
    n.rsb(t9, t6)
    n.sub(t7, t8h)
    n.add(t8, t7)
    _fdct8(n, t0, t8, t4, tc, t2, ta, t6, te)
    _fdst8(n, tf, t7, tb, t3, td, t5, t9, t1)


def _fdst16_asym(n, t0, t0h, t8, t4, t4h, tc, t2, ta, t6, te,
                 t1, t9, t5, td, t3, tb, t7, t7h, tf):
    # src/dct.c:1204-1443
    n.neg(t8)
    n.neg(t9)
    n.neg(ta)
    n.neg(tb)
    n.neg(td)
    n.sub(t1, M(te, 13573, 14))
    n.add(te, M(t1, 11585, 15))
    n.sub(t1, M(te, 13573, 14))
    n.add(t2, M(td, 4161, 14))
    n.sub(td, M(t2, 15137, 14))
    n.add(t2, M(td, 14341, 14))
    n.sub(tc, M(t3, 14341, 14))
    n.add(t3, M(tc, 15137, 14))
    n.sub(tc, M(t3, 4161, 14))
    n.rsb(te, t0h)
    n.sub(t0, te)
    n.rsb(tf, n.H(t1))
    n.sub(t1, tf)
    n.neg(tc)
    n.rsb(t2, n.H(tc))
    n.sub(tc, t2)
    n.rsb(t3, n.H(td))
    n.rsb(td, t3)
    n.sub(t9, M(t6, 7489, 13))
    n.add(t6, M(t9, 11585, 14))
    n.add(t9, M(t6, 19195, 15))
    n.add(t8, n.H(t9))
    n.sub(t9, t8)
    n.rsb(t6, t7h)
    n.sub(t7, t6)
    n.add(t8, M(t7, 6723, 13))
    n.sub(t7, M(t8, 16069, 14))
    n.add(t8, M(t7, 6723, 13))
    n.add(t9, M(t6, 17515, 15))
    n.sub(t6, M(t9, 13623, 14))
    n.add(t9, M(t6, 17515, 15))
    n.add(t5, M(ta, 13573, 14))
    n.sub(ta, M(t5, 11585, 15))
    n.add(t5, M(ta, 13573, 14))
    n.add(tb, n.H(t5))
    n.rsb(t5, tb)
    n.add(ta, t4h)
    n.sub(t4, ta)
    n.add(ta, M(t5, 2485, 13))
    n.sub(t5, M(ta, 18205, 15))
    n.add(ta, M(t5, 2485, 13))
    n.sub(tb, M(t4, 6723, 13))
    n.add(t4, M(tb, 16069, 14))
    n.sub(tb, M(t4, 6723, 13))
    n.neg(t5)
    n.sub(tc, tf)
    tch = n.H(tc)
    n.add(tf, tch)
    n.add(t3, t0)
    t3h = n.H(t3)
    n.sub(t0, t3h)
    n.sub(td, t1)
    tdh = n.H(td)
    n.add(t1, tdh)
    n.add(t2, te)
    t2h = n.H(t2)
    n.sub(te, t2h)
    n.add(t8, t4)
    t8h = n.H(t8)
    n.rsb(t4, t8h)
    n.rsb(t7, tb)
    t7h = n.H(t7)
    n.rsb(tb, t7h)
    n.sub(t6, ta)
    t6h = n.H(t6)
    n.add(ta, t6h)
    n.rsb(t9, t5)
    t9h = n.H(t9)
    n.sub(t5, t9h)
    n.sub(t0, t7h)
    n.add(t7, t0)
    n.add(tf, t8h)
    n.sub(t8, tf)
    n.sub(te, t6h)
    n.add(t6, te)
    n.add(t1, t9h)
    n.sub(t9, t1)
    n.sub(tb, tch)
    n.add(tc, tb)
    n.add(t4, t3h)
    n.sub(t3, t4)
    n.sub(ta, tdh)
    n.add(td, ta)
    n.rsb(t5, t2h)
    n.sub(t2, t5)
    n.neg(t8)
    n.neg(t9)
    n.neg(ta)
    n.neg(tb)
    n.neg(tc)
    n.neg(td)
    n.neg(tf)
    n.sub(t0, M(tf, 7799, 13))
    n.add(tf, M(t0, 4091, 12))
    n.sub(t0, M(tf, 7799, 13))
    n.add(t1, M(te, 2417, 15))
    n.sub(te, M(t1, 601, 12))
    n.add(t1, M(te, 2417, 15))
    n.sub(t7, M(t8, 14525, 15))
    n.add(t8, M(t7, 3035, 12))
    n.sub(t7, M(t8, 7263, 14))
    n.sub(t2, M(td, 6393, 13))
    n.add(td, M(t2, 3973, 12))
    n.sub(t2, M(td, 6393, 13))
    n.sub(t5, M(ta, 9281, 14))
    n.add(ta, M(t5, 7027, 13))
    n.sub(t5, M(ta, 9281, 14))
    n.sub(t3, M(tc, 11539, 14))
    n.add(tc, M(t3, 7713, 13))
    n.sub(t3, M(tc, 11539, 14))
    n.sub(t4, M(tb, 10375, 14))
    n.add(tb, M(t4, 7405, 13))
    n.sub(t4, M(tb, 10375, 14))
    n.sub(t6, M(t9, 8247, 14))
    n.add(t9, M(t6, 1645, 11))
    n.sub(t6, M(t9, 8247, 14))


def build_fdct32():
    # reference od_bin_fdct32 src/dct.c:1757-1857 + OD_FDCT_32 :1638-1701
    n = Net(32)
    order = ['t0', 'tg', 't8', 'to', 't4', 'tk', 'tc', 'ts',
             't2', 'ti', 'ta', 'tq', 't6', 'tm', 'te', 'tu',
             't1', 'th', 't9', 'tp', 't5', 'tl', 'td', 'tt',
             't3', 'tj', 'tb', 'tr', 't7', 'tn', 'tf', 'tv']
    n.load_order(order)
    (t0, tg, t8, to, t4, tk, tc, ts, t2, ti, ta, tq, t6, tm, te, tu,
     t1, th, t9, tp, t5, tl, td, tt, t3, tj, tb, tr, t7, tn, tf, tv) = order
    n.rsb(tv, t0)
    tvh = n.H(tv)
    n.sub(t0, tvh)
    n.add(tu, t1)
    tuh = n.H(tu)
    n.rsb(t1, tuh)
    n.rsb(tt, t2)
    n.sub(t2, n.H(tt))
    n.add(ts, t3)
    tsh = n.H(ts)
    n.rsb(t3, tsh)
    n.rsb(tr, t4)
    n.sub(t4, n.H(tr))
    n.add(tq, t5)
    tqh = n.H(tq)
    n.rsb(t5, tqh)
    n.rsb(tp, t6)
    n.sub(t6, n.H(tp))
    n.add(to, t7)
    toh = n.H(to)
    n.rsb(t7, toh)
    n.rsb(tn, t8)
    tnh = n.H(tn)
    n.sub(t8, tnh)
    n.add(tm, t9)
    tmh = n.H(tm)
    n.rsb(t9, tmh)
    n.rsb(tl, ta)
    n.sub(ta, n.H(tl))
    n.add(tk, tb)
    tkh = n.H(tk)
    n.rsb(tb, tkh)
    n.rsb(tj, tc)
    n.sub(tc, n.H(tj))
    n.add(ti, td)
    tih = n.H(ti)
    n.rsb(td, tih)
    n.rsb(th, te)
    thh = n.H(th)
    n.sub(te, thh)
    n.add(tg, tf)
    tgh = n.H(tg)
    n.rsb(tf, tgh)
    _fdct16_asym(n, t0, tg, tgh, t8, to, toh, t4, tk, tkh, tc, ts, tsh,
                 t2, ti, tih, ta, tq, tqh, t6, tm, tmh, te, tu, tuh)
    _fdst16_asym(n, tv, tvh, tf, tn, tnh, t7, tr, tb, tj, t3,
                 tt, td, tl, t5, tp, t9, th, thh, t1)
    return n


NETWORKS = {4: build_fdct4, 8: build_fdct8, 16: build_fdct16, 32: build_fdct32}


def inverse_steps(steps):
    """Mechanical inverse: reversed order, add<->sub, rsb/neg self-inverse."""
    out = []
    for st in reversed(steps):
        if st[0] == 'add':
            out.append(('sub',) + st[1:])
        elif st[0] == 'sub':
            out.append(('add',) + st[1:])
        else:
            out.append(st)
    return out
