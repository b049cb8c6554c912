"""ctypes binding of include/daala_hip.h.

Names follow the reference's (od_bin_fdct8x8, od_pre_filter4,
od_resample_luma_coeffs, ...) so that parity tests read like the reference's own
tests.  Every call goes to the HIP library; if it is missing or no device is
usable the call raises - nothing here computes on the CPU."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
I32P = ctypes.POINTER(ctypes.c_int32)
I16P = ctypes.POINTER(ctypes.c_int16)
U8P = ctypes.POINTER(ctypes.c_uint8)
F64P = ctypes.POINTER(ctypes.c_double)
c_int = ctypes.c_int


class HipError(RuntimeError):
    pass


class Geometry(ctypes.Structure):
    _fields_ = [('pic_width', c_int), ('pic_height', c_int), ('frame_width', c_int),
                ('frame_height', c_int), ('nplanes', c_int), ('xdec', c_int*4),
                ('nslots', c_int)]


class PvqBand(ctypes.Structure):
    _fields_ = [('cg', ctypes.c_double), ('g', ctypes.c_double),
                ('cos_dist', ctypes.c_double*2), ('dist', ctypes.c_double*2),
                ('qg', ctypes.c_int32*2), ('k', ctypes.c_int32*2),
                ('ncand', ctypes.c_int32), ('pad', ctypes.c_int32)]


class FeedLevel(ctypes.Structure):
    """od_hip_feed_level (include/daala_hip.h section 4b)."""
    _fields_ = [('n', ctypes.c_int32), ('nbands', ctypes.c_int32), ('nblk', ctypes.c_int32),
                ('nbx', ctypes.c_int32), ('off', ctypes.c_int32*11), ('pad', ctypes.c_int32),
                ('cg', F64P), ('g', F64P), ('ncand', I32P), ('qg', I32P), ('k', I32P), ('cos_dist', F64P),
                ('y', I16P), ('lev', I32P), ('lev_stride', ctypes.c_int32), ('pad2', ctypes.c_int32)]


PVQ_BAND_DTYPE = np.dtype([('cg', 'f8'), ('g', 'f8'), ('cos_dist', 'f8', 2), ('dist', 'f8', 2),
                           ('qg', 'i4', 2), ('k', 'i4', 2), ('ncand', 'i4'), ('pad', 'i4')])


def lib_path():
    # OD_HIP_LIB: A/B builds of the same library (tuning only)
    return os.environ.get('OD_HIP_LIB') or os.path.join(_HERE, 'libdaala_hip.so')


_lib = None


def load():
    """Load libdaala_hip.so; raises HipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise HipError('%s not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                       'or `make -C daala_amd/csrc` (there is no CPU fallback)' % path)
    lib = ctypes.CDLL(path)
    lib.od_hip_last_error.restype = ctypes.c_char_p
    lib.od_hip_version.restype = ctypes.c_char_p
    lib.od_hip_ctx_create.restype = ctypes.c_void_p
    lib.od_hip_ctx_create.argtypes = [c_int, ctypes.POINTER(Geometry)]
    lib.od_hip_ctx_destroy.argtypes = [ctypes.c_void_p]
    for name in ('od_hip_bin_fdct4x4', 'od_hip_bin_fdct8x8', 'od_hip_bin_fdct16x16',
                 'od_hip_bin_fdct32x32', 'od_hip_bin_idct4x4', 'od_hip_bin_idct8x8',
                 'od_hip_bin_idct16x16', 'od_hip_bin_idct32x32'):
        f = getattr(lib, name)
        f.restype = None
        f.argtypes = [I32P, c_int, I32P, c_int]
    lib.od_hip_fdct_blocks.argtypes = [c_int, I32P, I32P, c_int]
    lib.od_hip_idct_blocks.argtypes = [c_int, I32P, I32P, c_int]
    lib.od_hip_haar_blocks.argtypes = [c_int, c_int, I32P, I32P, c_int]
    lib.od_hip_filter4_vectors.argtypes = [c_int, I32P, I32P, c_int]
    lib.od_hip_resample_luma_420.argtypes = [I32P, I32P, ctypes.c_size_t, c_int, I32P, c_int,
                                             c_int, c_int]
    vp = ctypes.c_void_p
    lib.od_hip_upload_planes.argtypes = [vp, c_int, ctypes.POINTER(U8P), ctypes.POINTER(c_int)]
    lib.od_hip_forward_pyramid.argtypes = [vp, c_int, c_int]
    lib.od_hip_set_bsize.argtypes = [vp, c_int, U8P, c_int]
    lib.od_hip_forward_known.argtypes = [vp, c_int, c_int, c_int]
    lib.od_hip_inverse.argtypes = [vp, c_int, c_int]
    lib.od_hip_download_level.argtypes = [vp, c_int, c_int, c_int, I32P]
    lib.od_hip_download_coeffs.argtypes = [vp, c_int, c_int, I32P]
    lib.od_hip_upload_coeffs.argtypes = [vp, c_int, c_int, I32P]
    lib.od_hip_download_recon.argtypes = [vp, c_int, c_int, U8P]
    lib.od_hip_band_offsets.argtypes = [c_int, ctypes.POINTER(c_int)]
    lib.od_hip_pvq_noref_search.argtypes = [vp, c_int, c_int, c_int, c_int, I16P, I32P, F64P]
    lib.od_hip_pvq_nblocks.argtypes = [vp, c_int, c_int]
    lib.od_hip_pvq_download.argtypes = [vp, c_int, c_int, c_int, ctypes.c_void_p, I32P]
    lib.od_hip_pvq_search_vectors.argtypes = [c_int, c_int, F64P, I32P, F64P, I32P, F64P]
    lib.od_hip_pvq_synthesis_noref.argtypes = [c_int, c_int, I32P, F64P, I16P, I32P]
    lib.od_hip_sync.argtypes = [vp]
    lib.od_hip_timing_reset.argtypes = [vp]
    lib.od_hip_timing_get.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(c_int), F64P]
    _lib = lib
    return lib


def _p32(a):
    return a.ctypes.data_as(I32P)


def _chk(rc):
    if rc < 0:
        raise HipError('daala_hip error %d: %s' % (rc, load().od_hip_last_error().decode()))
    return rc


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def od_filter_vectors(n, vecs, inverse=False):
    """od_pre_filter{n} (inverse=False) / od_post_filter{n} of [nvec, n] int32 vectors."""
    x = _c32(vecs)
    y = np.empty_like(x)
    load().od_hip_filter_vectors.argtypes = [c_int, c_int, I32P, I32P, c_int]
    _chk(load().od_hip_filter_vectors(n, int(inverse), _p32(y), _p32(x), x.shape[0]))
    return y


# -- section 1/2 of the header: stand-alone pieces --------------------------------
def od_bin_fdct_blocks(bs, blocks):
    """blocks: [nblocks, n, n] int32 -> forward transform of each (od_bin_fdctNxN)."""
    x = _c32(blocks)
    y = np.empty_like(x)
    _chk(load().od_hip_fdct_blocks(bs, _p32(y), _p32(x), x.shape[0]))
    return y


def od_bin_idct_blocks(bs, blocks):
    y = _c32(blocks)
    x = np.empty_like(y)
    _chk(load().od_hip_idct_blocks(bs, _p32(x), _p32(y), y.shape[0]))
    return x


def od_haar_blocks(bs, blocks, inverse=False):
    x = _c32(blocks)
    y = np.empty_like(x)
    _chk(load().od_hip_haar_blocks(bs, int(inverse), _p32(y), _p32(x), x.shape[0]))
    return y


def comm_unique_id():
    """The 128-byte RCCL id rank 0 creates and every rank passes to Comm()."""
    buf = (ctypes.c_ubyte*128)()
    _chk(load().od_hip_comm_unique_id(buf))
    return bytes(buf)


class Comm(object):
    """od_hip_comm: an RCCL communicator over the ranks of a superblock-row sharded frame."""

    def __init__(self, device, world, rank, uid):
        lib = load()
        lib.od_hip_comm_create.restype = ctypes.c_void_p
        lib.od_hip_comm_create.argtypes = [c_int, c_int, c_int, ctypes.c_char_p]
        lib.od_hip_comm_destroy.argtypes = [ctypes.c_void_p]
        self.lib = lib
        self.h = lib.od_hip_comm_create(device, world, rank, uid)
        if not self.h:
            raise HipError(lib.od_hip_last_error().decode())

    def close(self):
        if self.h:
            self.lib.od_hip_comm_destroy(self.h)
            self.h = None


class McBlock(ctypes.Structure):
    """od_hip_mc_block (include/daala_hip.h)."""
    _fields_ = [('x', ctypes.c_int32), ('y', ctypes.c_int32), ('log_xblk_sz', ctypes.c_int32),
                ('log_yblk_sz', ctypes.c_int32), ('ref', ctypes.c_int32*4),
                ('mvx', ctypes.c_int32*4), ('mvy', ctypes.c_int32*4), ('oc', ctypes.c_int32),
                ('s', ctypes.c_int32)]


def od_mc_predict_blocks(refs, org_x, org_y, blocks, dst):
    """F3: OBMC prediction of a list of blocks.  refs: list of equal-shape padded u8 planes;
    blocks: list of dicts (x, y, lx, ly, ref[4], mvx[4], mvy[4], oc, s); dst: u8 plane (copy)."""
    refs = [np.ascontiguousarray(r, dtype=np.uint8) for r in refs]
    out = np.ascontiguousarray(dst, dtype=np.uint8).copy()
    arr = (McBlock*len(blocks))()
    for i, b in enumerate(blocks):
        arr[i].x, arr[i].y, arr[i].log_xblk_sz, arr[i].log_yblk_sz = b['x'], b['y'], b['lx'], b['ly']
        for k in range(4):
            arr[i].ref[k], arr[i].mvx[k], arr[i].mvy[k] = int(b['ref'][k]), int(b['mvx'][k]), int(b['mvy'][k])
        arr[i].oc, arr[i].s = b['oc'], b['s']
    ptrs = (U8P*len(refs))(*[r.ctypes.data_as(U8P) for r in refs])
    lib = load()
    lib.od_hip_mc_predict_blocks.argtypes = [c_int, ctypes.POINTER(U8P), c_int, c_int, c_int, c_int,
                                             ctypes.POINTER(McBlock), c_int, U8P, c_int, c_int]
    _chk(lib.od_hip_mc_predict_blocks(len(refs), ptrs, refs[0].shape[1], refs[0].shape[0], org_x, org_y,
                                      arr, len(blocks), out.ctypes.data_as(U8P), out.shape[1],
                                      out.shape[0]))
    return out


class McSad(object):
    """od_hip_mc with the frame being coded resident too: the batched OBMC + SAD of the motion
    search (od_hip_mc_set_ref / od_hip_mc_set_src / od_hip_mc_sad_items).
    refs[pli]: [nref, rows, stride] u8 with the picture origin at (org_x[pli], org_y[pli]);
    src[pli]: the padded input plane; dec[pli]: (xdec, ydec)."""

    ITEM = np.dtype([('x', np.int32), ('y', np.int32), ('log_blk_sz', np.int32), ('oc', np.int32),
                     ('s', np.int32), ('ref', np.int32, 4), ('mvx', np.int32, 4), ('mvy', np.int32, 4),
                     ('reserved', np.int32)])

    def __init__(self, refs, org_x, org_y, src, dec, device=0):
        lib = load()
        lib.od_hip_mc_create.restype = ctypes.c_void_p
        lib.od_hip_mc_create.argtypes = [c_int, c_int]
        lib.od_hip_mc_destroy.argtypes = [ctypes.c_void_p]
        lib.od_hip_mc_set_ref.argtypes = [ctypes.c_void_p, c_int, c_int, U8P, c_int, c_int, c_int, c_int]
        lib.od_hip_mc_set_src.argtypes = [ctypes.c_void_p, c_int, U8P, c_int, c_int, c_int, c_int, c_int]
        lib.od_hip_mc_sad_items.argtypes = [ctypes.c_void_p, c_int, c_int, c_int, ctypes.c_void_p, c_int,
                                            ctypes.POINTER(ctypes.c_int32)]
        self.lib = lib
        self.nplanes = len(refs)
        self.h = lib.od_hip_mc_create(device, refs[0].shape[0])
        if not self.h:
            raise HipError(lib.od_hip_last_error().decode())
        for pli in range(self.nplanes):
            r = np.ascontiguousarray(refs[pli], dtype=np.uint8)
            for k in range(r.shape[0]):
                _chk(lib.od_hip_mc_set_ref(self.h, pli, k, r[k].ctypes.data_as(U8P), r.shape[2], r.shape[1],
                                           int(org_x[pli]), int(org_y[pli])))
            sp = np.ascontiguousarray(src[pli], dtype=np.uint8)
            _chk(lib.od_hip_mc_set_src(self.h, pli, sp.ctypes.data_as(U8P), sp.shape[1], sp.shape[1], sp.shape[0],
                                       int(dec[pli][0]), int(dec[pli][1])))

    def set_ref_ctx(self, pli, k, ctx, slot, ref_stride, ref_h, org_x, org_y):
        """od_hip_mc_set_ref_ctx: reference image k of plane pli from the reconstruction plane of
        a DaalaHip context's slot, padded on the device."""
        self.lib.od_hip_mc_set_ref_ctx.argtypes = [ctypes.c_void_p, c_int, c_int, ctypes.c_void_p, c_int, c_int,
                                                   c_int, c_int, c_int]
        _chk(self.lib.od_hip_mc_set_ref_ctx(self.h, pli, k, ctx.ctx, slot, ref_stride, ref_h, org_x, org_y))

    BMA_REC = np.dtype([('bx', np.int32), ('by', np.int32), ('log_blk_sz', np.int32), ('ref', np.int32),
                        ('cx', np.int32), ('cy', np.int32), ('xmin', np.int32), ('xmax', np.int32),
                        ('ymin', np.int32), ('ymax', np.int32)])

    def bma_windows(self, recs, radius, pic_w, pic_h, nplanes=None):
        """od_hip_mc_bma_windows: [nrec][(2 radius + 1)^2] block-matching SADs (-1: outside the limits)."""
        r = np.ascontiguousarray(recs, dtype=self.BMA_REC)
        W = 2*radius + 1
        out = np.zeros((len(r), W*W), np.int32)
        self.lib.od_hip_mc_bma_windows.argtypes = [ctypes.c_void_p, c_int, c_int, c_int, ctypes.c_void_p, c_int, c_int,
                                                   ctypes.POINTER(ctypes.c_int32)]
        _chk(self.lib.od_hip_mc_bma_windows(self.h, self.nplanes if nplanes is None else nplanes, pic_w, pic_h,
                                            r.ctypes.data, len(r), radius, _p32(out)))
        return out

    def sad_items(self, items, pic_w, pic_h, nplanes=None):
        it = np.ascontiguousarray(items, dtype=self.ITEM)
        out = np.zeros(len(it), np.int32)
        _chk(self.lib.od_hip_mc_sad_items(self.h, self.nplanes if nplanes is None else nplanes, pic_w, pic_h,
                                          it.ctypes.data, len(it), _p32(out)))
        return out

    def close(self):
        if self.h:
            self.lib.od_hip_mc_destroy(self.h)
            self.h = None


def od_coding_order_blocks(bs, blocks, to_raster=False, dst=None):
    """A11 gather (raster -> coding order) / scatter (coding order -> raster) of dense blocks."""
    x = _c32(blocks)
    y = np.zeros_like(x) if dst is None else _c32(dst).copy()
    _chk(load().od_hip_coding_order_blocks(bs, int(to_raster), _p32(y), _p32(x), x.shape[0]))
    return y


def od_pre_filter4(vectors):
    x = _c32(vectors)
    y = np.empty_like(x)
    _chk(load().od_hip_filter4_vectors(0, _p32(y), _p32(x), x.shape[0]))
    return y


def od_post_filter4(vectors):
    x = _c32(vectors)
    y = np.empty_like(x)
    _chk(load().od_hip_filter4_vectors(1, _p32(y), _p32(x), x.shape[0]))
    return y


def od_resample_luma_coeffs_420(luma, lstride, offsets, bs, chroma_bs):
    luma = _c32(luma).ravel()
    off = np.ascontiguousarray(offsets, dtype=np.int32)
    n = 4 << bs
    pred = np.empty((len(off), n, n), np.int32)
    _chk(load().od_hip_resample_luma_420(_p32(pred), _p32(luma), luma.size, lstride, _p32(off),
                                         len(off), bs, chroma_bs))
    return pred


def od_resample_luma_coeffs(luma, lstride, offsets, bs, chroma_bs, xdec, ydec):
    """od_resample_luma_coeffs (src/intra.c:72) for any chroma decimation."""
    luma = _c32(luma).ravel()
    off = np.ascontiguousarray(offsets, dtype=np.int32)
    n = 4 << bs
    pred = np.empty((len(off), n, n), np.int32)
    load().od_hip_resample_luma.argtypes = [I32P, I32P, ctypes.c_size_t, c_int, I32P, c_int, c_int,
                                            c_int, c_int, c_int]
    _chk(load().od_hip_resample_luma(_p32(pred), _p32(luma), luma.size, lstride, _p32(off),
                                     len(off), bs, chroma_bs, xdec, ydec))
    return pred


def vtable_call(name, out, ostride, inp, istride):
    """Call one of the od_dct_func_2d drop-ins on (possibly aliasing) host arrays."""
    getattr(load(), name)(_p32(out), ostride, _p32(inp), istride)


def pvq_search_vectors(x, k, g2):
    x = np.ascontiguousarray(x, dtype=np.float64)
    nvec, n = x.shape
    k = np.ascontiguousarray(k, dtype=np.int32)
    g2 = np.ascontiguousarray(g2, dtype=np.float64)
    y = np.empty((nvec, n), np.int32)
    cd = np.empty(nvec, np.float64)
    _chk(load().od_hip_pvq_search_vectors(n, nvec, x.ctypes.data_as(F64P), _p32(k),
                                          g2.ctypes.data_as(F64P), _p32(y),
                                          cd.ctypes.data_as(F64P)))
    return y, cd


def pvq_synthesis_noref(y, g, qm_inv):
    y = _c32(y)
    nvec, n = y.shape
    g = np.ascontiguousarray(g, dtype=np.float64)
    qm_inv = np.ascontiguousarray(qm_inv, dtype=np.int16)
    out = np.empty_like(y)
    _chk(load().od_hip_pvq_synthesis_noref(n, nvec, _p32(y), g.ctypes.data_as(F64P),
                                           qm_inv.ctypes.data_as(I16P), _p32(out)))
    return out


def band_offsets(bs):
    off = (c_int*11)()
    nb = _chk(load().od_hip_band_offsets(bs, off))
    return [off[i] for i in range(nb + 1)]


# -- section 3/4: device-resident frame pipeline ------------------------------------
class DaalaHip(object):
    """One od_hip_ctx: the HBM-resident buffers of `nslots` frames."""

    def __init__(self, pic_width, pic_height, frame_width=None, frame_height=None, nplanes=3,
                 xdec=(0, 1, 1), nslots=1, device=0):
        self.lib = load()
        fw = frame_width or (pic_width + 31)//32*32
        fh = frame_height or (pic_height + 31)//32*32
        g = Geometry()
        g.pic_width, g.pic_height, g.frame_width, g.frame_height = pic_width, pic_height, fw, fh
        g.nplanes, g.nslots = nplanes, nslots
        for i in range(nplanes):
            g.xdec[i] = xdec[i]
        self.geo = g
        self.nplanes = nplanes
        self.xdec = tuple(xdec[:nplanes])
        self.fw, self.fh, self.nslots = fw, fh, nslots
        self.ctx = self.lib.od_hip_ctx_create(device, ctypes.byref(g))
        if not self.ctx:
            raise HipError('od_hip_ctx_create failed: %s' % self.lib.od_hip_last_error().decode())

    def close(self):
        if getattr(self, 'feed', None):
            self.lib.od_hip_enc_feed_destroy(self.feed)
            self.feed = None
        if getattr(self, 'ctx', None):
            self.lib.od_hip_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        self.close()

    def plane_shape(self, pli):
        return (self.fh >> self.xdec[pli], self.fw >> self.xdec[pli])

    def nlevels(self, pli):
        return 4 - self.xdec[pli]

    def upload_planes(self, slot, planes):
        planes = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
        for pli, p in enumerate(planes):
            assert p.shape == self.plane_shape(pli), (p.shape, self.plane_shape(pli))
        ptrs = (U8P*len(planes))(*[p.ctypes.data_as(U8P) for p in planes])
        strides = (c_int*len(planes))(*[p.shape[1] for p in planes])
        _chk(self.lib.od_hip_upload_planes(self.ctx, slot, ptrs, strides))

    def set_bsize(self, slot, bsize):
        b = np.ascontiguousarray(bsize, dtype=np.uint8)
        assert b.shape == (self.fh//8, self.fw//8)
        _chk(self.lib.od_hip_set_bsize(self.ctx, slot, b.ctypes.data_as(U8P), b.shape[1]))

    def forward_pyramid(self, slot0=0, nslots=None):
        _chk(self.lib.od_hip_forward_pyramid(self.ctx, slot0, nslots or self.nslots - slot0))

    def forward_known(self, slot0=0, nslots=None, keyframe=1):
        _chk(self.lib.od_hip_forward_known(self.ctx, slot0, nslots or self.nslots - slot0,
                                           keyframe))

    def inverse(self, slot0=0, nslots=None):
        _chk(self.lib.od_hip_inverse(self.ctx, slot0, nslots or self.nslots - slot0))

    def download_level(self, slot, pli, level):
        out = np.empty(self.plane_shape(pli), np.int32)
        _chk(self.lib.od_hip_download_level(self.ctx, slot, pli, level, _p32(out)))
        return out

    def download_coeffs(self, slot, pli):
        out = np.empty(self.plane_shape(pli), np.int32)
        _chk(self.lib.od_hip_download_coeffs(self.ctx, slot, pli, _p32(out)))
        return out

    def upload_coeffs(self, slot, pli, d):
        d = _c32(d)
        assert d.shape == self.plane_shape(pli)
        _chk(self.lib.od_hip_upload_coeffs(self.ctx, slot, pli, _p32(d)))

    def download_recon(self, slot, pli):
        out = np.empty(self.plane_shape(pli), np.uint8)
        _chk(self.lib.od_hip_download_recon(self.ctx, slot, pli, out.ctypes.data_as(U8P)))
        return out

    def pvq_noref_search(self, pli, level, qm, q, beta, slot0=0, nslots=None):
        qm = np.ascontiguousarray(qm, dtype=np.int16)
        q = np.ascontiguousarray(q, dtype=np.int32)
        beta = np.ascontiguousarray(beta, dtype=np.float64)
        _chk(self.lib.od_hip_pvq_noref_search(self.ctx, slot0, nslots or self.nslots - slot0,
                                              pli, level, qm.ctypes.data_as(I16P), _p32(q),
                                              beta.ctypes.data_as(F64P)))

    def _pvq_call(self, fn, pli, level, qm, q, beta, slot0, nslots, with_qm=True):
        qa = np.ascontiguousarray(q, dtype=np.int32)
        ba = np.ascontiguousarray(beta, dtype=np.float64)
        args = [self.ctx, slot0, nslots or self.nslots - slot0, pli, level]
        if with_qm:
            qma = np.ascontiguousarray(qm, dtype=np.int16)
            fn.argtypes = [ctypes.c_void_p, c_int, c_int, c_int, c_int, I16P, I32P, F64P]
            args.append(qma.ctypes.data_as(I16P))
        else:
            fn.argtypes = [ctypes.c_void_p, c_int, c_int, c_int, c_int, I32P, F64P]
        _chk(fn(*args, _p32(qa), ba.ctypes.data_as(F64P)))

    def pvq_gains(self, pli, level, qm, q, beta, slot0=0, nslots=None):
        """Device pass 1: exact uncompanded gains of every band (od_hip_pvq_gains)."""
        self._pvq_call(self.lib.od_hip_pvq_gains, pli, level, qm, q, beta, slot0, nslots)

    def pvq_compand_level(self, pli, level, q, beta, slot0=0, nslots=None):
        """Host stage: g down, cg = od_gain_compand(g) with the host's libm, cg up."""
        self._pvq_call(self.lib.od_hip_pvq_compand_level, pli, level, None, q, beta, slot0, nslots,
                       with_qm=False)

    def pvq_search(self, pli, level, qm, q, beta, slot0=0, nslots=None):
        """Device pass 2: candidates, K, codeword searches with the uploaded cg."""
        self._pvq_call(self.lib.od_hip_pvq_search, pli, level, qm, q, beta, slot0, nslots)

    def set_strip(self, sb_row0, sb_row1):
        """Restrict the forward pyramid and the PVQ passes to superblock rows [r0, r1)."""
        self.lib.od_hip_set_strip.argtypes = [ctypes.c_void_p, c_int, c_int]
        _chk(self.lib.od_hip_set_strip(self.ctx, sb_row0, sb_row1))

    def gather_strips(self, comm, slot, sb_rows, with_pvq=True):
        rows = np.ascontiguousarray(sb_rows, dtype=np.int32)
        self.lib.od_hip_gather_strips.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_int, I32P, c_int]
        _chk(self.lib.od_hip_gather_strips(self.ctx, comm.h, slot, _p32(rows), int(with_pvq)))

    def strip_export(self, slot, r0, r1, with_pvq=True):
        """The packed results of superblock rows [r0, r1) as one host buffer."""
        self.lib.od_hip_strip_bytes.restype = ctypes.c_long
        self.lib.od_hip_strip_bytes.argtypes = [ctypes.c_void_p, c_int, c_int, c_int, c_int]
        n = self.lib.od_hip_strip_bytes(self.ctx, slot, r0, r1, int(with_pvq))
        if n < 0:
            raise HipError(self.lib.od_hip_last_error().decode())
        buf = np.zeros(max(n, 1), np.uint8)
        self.lib.od_hip_strip_export.argtypes = [ctypes.c_void_p, c_int, c_int, c_int, c_int, U8P, ctypes.c_long]
        _chk(self.lib.od_hip_strip_export(self.ctx, slot, r0, r1, int(with_pvq), buf.ctypes.data_as(U8P), n))
        return buf[:n]

    def strip_import(self, slot, r0, r1, blob, with_pvq=True):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self.lib.od_hip_strip_import.argtypes = [ctypes.c_void_p, c_int, c_int, c_int, c_int, U8P, ctypes.c_long]
        _chk(self.lib.od_hip_strip_import(self.ctx, slot, r0, r1, int(with_pvq), blob.ctypes.data_as(U8P), blob.size))

    def pvq_stats(self, enable):
        """Device work counters of the searches (greedy element steps, RDO element steps,
        candidates); enable=1 restarts them, 0 reads and stops."""
        out = (ctypes.c_ulonglong*3)()
        self.lib.od_hip_pvq_stats.argtypes = [ctypes.c_void_p, c_int, ctypes.POINTER(ctypes.c_ulonglong)]
        _chk(self.lib.od_hip_pvq_stats(self.ctx, int(enable), out))
        return [int(v) for v in out]

    def pvq_nblocks(self, pli, level):
        return _chk(self.lib.od_hip_pvq_nblocks(self.ctx, pli, level))

    def pvq_download(self, slot, pli, level):
        nblk = self.pvq_nblocks(pli, level)
        n = (32 >> self.xdec[pli]) >> level
        bs = {4: 0, 8: 1, 16: 2, 32: 3}[n]
        nb = len(band_offsets(bs)) - 1
        ncoded = min(n*n, 512)
        bands = np.zeros((nblk, nb), PVQ_BAND_DTYPE)
        y = np.zeros((nblk, 2, ncoded), np.int32)
        _chk(self.lib.od_hip_pvq_download(self.ctx, slot, pli, level,
                                          bands.ctypes.data_as(ctypes.c_void_p), _p32(y)))
        return bands, y

    def forward_haar(self, slot0=0, nslots=None):
        """Lossless frames: od_haar of every superblock (pixels -> coefficient planes)."""
        self.lib.od_hip_forward_haar.argtypes = [ctypes.c_void_p, c_int, c_int]
        _chk(self.lib.od_hip_forward_haar(self.ctx, slot0, nslots or self.nslots - slot0))

    def inverse_haar(self, slot0=0, nslots=None):
        """Lossless frames: od_haar_inv + 8-bit store (coefficient planes -> recon)."""
        self.lib.od_hip_inverse_haar.argtypes = [ctypes.c_void_p, c_int, c_int]
        _chk(self.lib.od_hip_inverse_haar(self.ctx, slot0, nslots or self.nslots - slot0))

    # -- encoder feed (header section 4b) -----------------------------------------
    def enc_feed_create(self, qm, q, beta):
        """qm [4][1024] int16, q [4][11] int32, beta [4][11] f64: per luma level."""
        lib = self.lib
        lib.od_hip_enc_feed_create.restype = ctypes.c_void_p
        lib.od_hip_enc_feed_create.argtypes = [ctypes.c_void_p]
        lib.od_hip_enc_feed_destroy.argtypes = [ctypes.c_void_p]
        lib.od_hip_enc_feed_set_level.argtypes = [ctypes.c_void_p, c_int, I16P, I32P, F64P]
        lib.od_hip_enc_feed_run.argtypes = [ctypes.c_void_p, c_int, c_int]
        lib.od_hip_enc_feed_view.argtypes = [ctypes.c_void_p, c_int, ctypes.POINTER(FeedLevel)]
        self.feed = lib.od_hip_enc_feed_create(self.ctx)
        if not self.feed:
            raise HipError('od_hip_enc_feed_create: %s' % lib.od_hip_last_error().decode())
        for l in range(4):
            qm_l = np.ascontiguousarray(qm[l], dtype=np.int16)
            q_l = np.ascontiguousarray(q[l], dtype=np.int32)
            b_l = np.ascontiguousarray(beta[l], dtype=np.float64)
            _chk(lib.od_hip_enc_feed_set_level(self.feed, l, qm_l.ctypes.data_as(I16P), _p32(q_l),
                                               b_l.ctypes.data_as(F64P)))

    def enc_feed_set_level_plane(self, pli, level, qm, q, beta):
        """od_hip_enc_feed_set_level_plane: puts plane pli into the feed (all its levels must be set)."""
        self.lib.od_hip_enc_feed_set_level_plane.argtypes = [ctypes.c_void_p, c_int, c_int, I16P, I32P, F64P]
        qm_l = np.ascontiguousarray(qm, dtype=np.int16)
        q_l = np.ascontiguousarray(q, dtype=np.int32)
        b_l = np.ascontiguousarray(beta, dtype=np.float64)
        _chk(self.lib.od_hip_enc_feed_set_level_plane(self.feed, pli, level, qm_l.ctypes.data_as(I16P), _p32(q_l),
                                                      b_l.ctypes.data_as(F64P)))

    def enc_feed_phases(self, slot0=0, nslots=None):
        """od_hip_enc_feed_gains -> _compand per slot -> _search (what enc_feed_run does), as
        separate calls so that a strip set with set_strip() is honoured by every phase."""
        n = nslots or self.nslots - slot0
        lib = self.lib
        for f in (lib.od_hip_enc_feed_gains, lib.od_hip_enc_feed_search, lib.od_hip_enc_feed_refresh):
            f.argtypes = [ctypes.c_void_p, c_int, c_int]
        lib.od_hip_enc_feed_compand.argtypes = [ctypes.c_void_p, c_int]
        _chk(lib.od_hip_enc_feed_gains(self.feed, slot0, n))
        for s in range(slot0, slot0 + n):
            _chk(lib.od_hip_enc_feed_compand(self.feed, s))
        _chk(lib.od_hip_enc_feed_search(self.feed, slot0, n))

    def enc_feed_refresh(self, slot0=0, nslots=None):
        self.lib.od_hip_enc_feed_refresh.argtypes = [ctypes.c_void_p, c_int, c_int]
        _chk(self.lib.od_hip_enc_feed_refresh(self.feed, slot0, nslots or self.nslots - slot0))

    def enc_feed_views_raw(self, slot):
        """The four od_hip_feed_level structs of a slot (pointers into the feed's pinned host
        mirrors, valid until the slot is reused) - what od_hipenc_encode_frames takes as views."""
        lev = (FeedLevel*4)()
        _chk(self.lib.od_hip_enc_feed_view(self.feed, slot, lev))
        return lev

    def enc_feed_run(self, slot0=0, nslots=None):
        _chk(self.lib.od_hip_enc_feed_run(self.feed, slot0, nslots or self.nslots - slot0))

    def enc_feed_view(self, slot, pli=0):
        """Host arrays (copies) of one slot of plane pli: list of dicts cg/ncand/qg/k/cos_dist/y,
        one per level of the plane."""
        lev = (FeedLevel*4)()
        self.lib.od_hip_enc_feed_view_plane.argtypes = [ctypes.c_void_p, c_int, c_int, ctypes.POINTER(FeedLevel)]
        _chk(self.lib.od_hip_enc_feed_view_plane(self.feed, slot, pli, lev))
        out = []
        for v in list(lev)[:self.nlevels(pli)]:
            nrec = v.nbands*v.nblk
            ny = 2*v.nblk*min(v.n*v.n, 512)      # int16, bands padded to even (daala_hip.h 4b)
            out.append({'n': v.n, 'nbands': v.nbands, 'nblk': v.nblk, 'nbx': v.nbx,
                        'off': list(v.off)[:v.nbands + 1],
                        'cg': np.ctypeslib.as_array(v.cg, (nrec,)).copy(),
                        'g': np.ctypeslib.as_array(v.g, (nrec,)).copy(),
                        'ncand': np.ctypeslib.as_array(v.ncand, (nrec,)).copy(),
                        'qg': np.ctypeslib.as_array(v.qg, (2*nrec,)).copy(),
                        'k': np.ctypeslib.as_array(v.k, (2*nrec,)).copy(),
                        'cos_dist': np.ctypeslib.as_array(v.cos_dist, (2*nrec,)).copy(),
                        'y': np.ctypeslib.as_array(v.y, (ny,)).copy(),
                        'lev': np.ctypeslib.as_array(v.lev, self.plane_shape(pli)).copy()})
        return out

    def set_decode_info(self, slot, dering_flags, bskip):
        """dering_flags: [nvsb, nhsb] u8; bskip: per plane [fh/4, fw/4] u8 (dense)."""
        fl = np.ascontiguousarray(dering_flags, dtype=np.uint8)
        bs = [np.ascontiguousarray(b, dtype=np.uint8) for b in bskip]
        ptrs = (U8P*len(bs))(*[b.ctypes.data_as(U8P) for b in bs])
        self.lib.od_hip_set_decode_info.argtypes = [ctypes.c_void_p, c_int, U8P,
                                                    ctypes.POINTER(U8P), c_int]
        _chk(self.lib.od_hip_set_decode_info(self.ctx, slot, fl.ctypes.data_as(U8P), ptrs,
                                             bs[0].shape[1]))

    def decode_tail(self, threshold, quantizer, is_keyframe, slot0=0, nslots=None):
        th = np.ascontiguousarray(threshold, dtype=np.int32)
        q = np.ascontiguousarray(quantizer, dtype=np.int32)
        self.lib.od_hip_decode_tail.argtypes = [ctypes.c_void_p, c_int, c_int, I32P, I32P, c_int]
        _chk(self.lib.od_hip_decode_tail(self.ctx, slot0, nslots or self.nslots - slot0, _p32(th),
                                         _p32(q), int(is_keyframe)))

    def sync(self):
        _chk(self.lib.od_hip_sync(self.ctx))

    def timing_reset(self):
        _chk(self.lib.od_hip_timing_reset(self.ctx))

    def timing_get(self, kernel):
        n = c_int()
        ms = ctypes.c_double()
        _chk(self.lib.od_hip_timing_get(self.ctx, kernel.encode(), ctypes.byref(n),
                                        ctypes.byref(ms)))
        return n.value, ms.value


class PvqThetaOut(ctypes.Structure):
    """od_hip_pvq_theta_out (include/daala_hip.h)."""
    _fields_ = [('cg', ctypes.c_double), ('cgr', ctypes.c_double), ('g', ctypes.c_double),
                ('gr', ctypes.c_double), ('corr', ctypes.c_double), ('theta', ctypes.c_double),
                ('gain_offset', ctypes.c_double), ('skip_dist', ctypes.c_double),
                ('null_dist', ctypes.c_double),
                ('icgr', ctypes.c_int32), ('m', ctypes.c_int32), ('s', ctypes.c_int32),
                ('nref', ctypes.c_int32), ('nnoref', ctypes.c_int32),
                ('theta_searched', ctypes.c_int32), ('noref_searched', ctypes.c_int32),
                ('pad', ctypes.c_int32),
                ('ref_qg', ctypes.c_int32*12), ('ref_itheta', ctypes.c_int32*12),
                ('ref_ts', ctypes.c_int32*12), ('ref_k', ctypes.c_int32*12),
                ('ref_qtheta', ctypes.c_double*12), ('ref_cos_dist', ctypes.c_double*12),
                ('ref_dist', ctypes.c_double*12),
                ('nr_qg', ctypes.c_int32*2), ('nr_k', ctypes.c_int32*2),
                ('nr_cos_dist', ctypes.c_double*2), ('nr_dist', ctypes.c_double*2)]


def pvq_theta_vectors(x0, r0, qm, q0, beta, robust, is_keyframe, pli):
    """pvq_theta candidates (no rate term) for [nvec][n] inputs/predictions."""
    lib = load()
    x0 = _c32(x0)
    r0 = _c32(r0)
    nvec, n = x0.shape
    qm = np.ascontiguousarray(qm, dtype=np.int16)
    q0 = np.ascontiguousarray(q0, dtype=np.int32)
    out = (PvqThetaOut*nvec)()
    y_ref = np.zeros((nvec, 12, n), np.int32)
    y_nr = np.zeros((nvec, 2, n), np.int32)
    lib.od_hip_pvq_theta_vectors.argtypes = [c_int, c_int, I32P, I32P, I16P, I32P,
                                             ctypes.c_double, c_int, c_int, c_int,
                                             ctypes.POINTER(PvqThetaOut), I32P, I32P]
    _chk(lib.od_hip_pvq_theta_vectors(n, nvec, _p32(x0), _p32(r0), qm.ctypes.data_as(I16P),
                                      _p32(q0), float(beta), int(robust), int(is_keyframe),
                                      int(pli), out, _p32(y_ref), _p32(y_nr)))
    return out, y_ref, y_nr


def pvq_synthesis_vectors(y, ref, gr, noref, g, theta, qm, qm_inv):
    lib = load()
    y = _c32(y)
    ref = _c32(ref)
    nvec, n = y.shape
    gr = np.ascontiguousarray(gr, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    noref = np.ascontiguousarray(noref, dtype=np.int32)
    qm = np.ascontiguousarray(qm, dtype=np.int16)
    qm_inv = np.ascontiguousarray(qm_inv, dtype=np.int16)
    out = np.empty_like(y)
    lib.od_hip_pvq_synthesis_vectors.argtypes = [c_int, c_int, I32P, I32P, F64P, I32P, F64P, F64P,
                                                 I16P, I16P, I32P]
    _chk(lib.od_hip_pvq_synthesis_vectors(n, nvec, _p32(y), _p32(ref), gr.ctypes.data_as(F64P),
                                          _p32(noref), g.ctypes.data_as(F64P),
                                          theta.ctypes.data_as(F64P), qm.ctypes.data_as(I16P),
                                          qm_inv.ctypes.data_as(I16P), _p32(out)))
    return out


def od_hv_intra_pred_blocks(d, bsize, bs, bx, by):
    """OD_CLEAR + od_hv_intra_pred for blocks (bx[i], by[i]) (4x4 units) of plane d."""
    lib = load()
    d = _c32(d)
    h, w = d.shape
    bsize = np.ascontiguousarray(bsize, dtype=np.uint8)
    bx = np.ascontiguousarray(bx, dtype=np.int32)
    by = np.ascontiguousarray(by, dtype=np.int32)
    n = 4 << bs
    pred = np.empty((len(bx), n*n), np.int32)
    lib.od_hip_hv_intra_pred_blocks.argtypes = [I32P, c_int, c_int, U8P, c_int, c_int, c_int, I32P,
                                                I32P, I32P]
    _chk(lib.od_hip_hv_intra_pred_blocks(_p32(d), w, h, bsize.ctypes.data_as(U8P), bsize.shape[1],
                                         bs, len(bx), _p32(bx), _p32(by), _p32(pred)))
    return pred


def od_compute_dist_blocks(bs, x, y, mag2, activity_masking):
    lib = load()
    x = _c32(x)
    y = _c32(y)
    mag2 = np.ascontiguousarray(mag2, dtype=np.float64)
    out = np.empty(x.shape[0], np.float64)
    lib.od_hip_compute_dist_blocks.argtypes = [c_int, c_int, I32P, I32P, F64P, c_int, F64P]
    _chk(lib.od_hip_compute_dist_blocks(bs, x.shape[0], _p32(x), _p32(y), mag2.ctypes.data_as(F64P),
                                        int(activity_masking), out.ctypes.data_as(F64P)))
    return out


class PFeedLevel(ctypes.Structure):
    """od_hip_pfeed_level (include/daala_hip.h section 4d)."""
    _fields_ = [('n', ctypes.c_int32), ('nbands', ctypes.c_int32), ('nblk', ctypes.c_int32),
                ('nbx', ctypes.c_int32), ('off', ctypes.c_int32*11), ('nslots', ctypes.c_int32),
                ('nref_slots', ctypes.c_int32), ('pad', ctypes.c_int32),
                ('g', F64P), ('gr', F64P), ('corr', F64P), ('isnull', I32P), ('cg', F64P), ('cgr', F64P),
                ('theta', F64P), ('flags', I32P), ('cos_dist', F64P), ('k', I32P), ('y', I16P)]


class PFeed:
    """od_hip_pfeed_*: the complete pvq_theta candidate lists of one inter frame (all planes,
    all levels) from its padded input planes and its motion-compensated prediction."""

    def __init__(self, pic_w, pic_h, fw, fh, nplanes=3, xdec=(0, 1, 1), device=0):
        self.lib = load()
        g = Geometry()
        g.pic_width, g.pic_height, g.frame_width, g.frame_height = pic_w, pic_h, fw, fh
        g.nplanes = nplanes
        for i, d in enumerate(xdec):
            g.xdec[i] = d
        g.nslots = 2
        self.lib.od_hip_pfeed_create.restype = ctypes.c_void_p
        self.lib.od_hip_pfeed_create.argtypes = [c_int, ctypes.POINTER(Geometry)]
        self.h = self.lib.od_hip_pfeed_create(device, ctypes.byref(g))
        if not self.h:
            raise HipError(self.lib.od_hip_last_error().decode())
        self.nplanes, self.xdec = nplanes, tuple(xdec)
        for name in ('od_hip_pfeed_destroy', 'od_hip_pfeed_set_level', 'od_hip_pfeed_gains',
                     'od_hip_pfeed_nrec', 'od_hip_pfeed_host_stage', 'od_hip_pfeed_search',
                     'od_hip_pfeed_view'):
            getattr(self.lib, name).argtypes = None
        self.lib.od_hip_pfeed_destroy.argtypes = [ctypes.c_void_p]
        self.lib.od_hip_pfeed_set_level.argtypes = [ctypes.c_void_p, c_int, c_int, I16P, I32P, F64P]
        self.lib.od_hip_pfeed_gains.argtypes = [ctypes.c_void_p, ctypes.POINTER(U8P), I32P,
                                                ctypes.POINTER(U8P), I32P]
        self.lib.od_hip_pfeed_nrec.argtypes = [ctypes.c_void_p, c_int, c_int]
        self.lib.od_hip_pfeed_host_stage.argtypes = [ctypes.c_void_p, c_int, c_int, ctypes.c_long, ctypes.c_long]
        self.lib.od_hip_pfeed_search.argtypes = [ctypes.c_void_p]
        self.lib.od_hip_pfeed_view.argtypes = [ctypes.c_void_p, c_int, c_int, ctypes.POINTER(PFeedLevel)]

    def nlevels(self, pli):
        return 4 - self.xdec[pli]

    def set_level(self, pli, level, qm, q, beta):
        qm = np.ascontiguousarray(qm, dtype=np.int16)
        q = np.ascontiguousarray(list(q) + [1]*(11 - len(q)), dtype=np.int32)
        beta = np.ascontiguousarray(list(beta) + [1.]*(11 - len(beta)), dtype=np.float64)
        _chk(self.lib.od_hip_pfeed_set_level(self.h, pli, level, qm.ctypes.data_as(I16P), _p32(q),
                                             beta.ctypes.data_as(F64P)))

    def run(self, planes_in, planes_pred):
        def pack(planes):
            pl = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
            return pl, (U8P*len(pl))(*[p.ctypes.data_as(U8P) for p in pl]), \
                np.ascontiguousarray([p.shape[1] for p in pl], dtype=np.int32)
        a, pa, sa = pack(planes_in)
        b, pb, sb = pack(planes_pred)
        _chk(self.lib.od_hip_pfeed_gains(self.h, pa, _p32(sa), pb, _p32(sb)))
        for pli in range(self.nplanes):
            for level in range(self.nlevels(pli)):
                n = self.lib.od_hip_pfeed_nrec(self.h, pli, level)
                _chk(self.lib.od_hip_pfeed_host_stage(self.h, pli, level, 0, n))
        _chk(self.lib.od_hip_pfeed_search(self.h))

    def view(self, pli, level):
        v = PFeedLevel()
        _chk(self.lib.od_hip_pfeed_view(self.h, pli, level, ctypes.byref(v)))
        nrec = v.nbands*v.nblk
        ncoded = min(v.n*v.n, 512)
        arr = np.ctypeslib.as_array
        return {'n': v.n, 'nbands': v.nbands, 'nblk': v.nblk, 'nbx': v.nbx, 'off': list(v.off)[:v.nbands + 1],
                'nslots': v.nslots, 'nref_slots': v.nref_slots,
                'g': arr(v.g, (nrec,)).copy(), 'gr': arr(v.gr, (nrec,)).copy(),
                'corr': arr(v.corr, (nrec,)).copy(), 'isnull': arr(v.isnull, (nrec,)).copy(),
                'cg': arr(v.cg, (nrec,)).copy(), 'cgr': arr(v.cgr, (nrec,)).copy(),
                'theta': arr(v.theta, (nrec,)).copy(), 'flags': arr(v.flags, (nrec,)).copy(),
                'cos_dist': arr(v.cos_dist, (v.nslots*nrec,)).copy().reshape(v.nslots, nrec),
                'k': arr(v.k, (v.nslots*nrec,)).copy().reshape(v.nslots, nrec),
                'y': arr(v.y, (v.nslots*v.nblk*ncoded,)).copy()}

    def close(self):
        if self.h:
            self.lib.od_hip_pfeed_destroy(self.h)
            self.h = None


class DSynthBlock(ctypes.Structure):
    """od_hip_dsynth_block (include/daala_hip.h section 4e)."""
    _fields_ = [('org', ctypes.c_int32), ('dc', ctypes.c_int32), ('pli', ctypes.c_uint8),
                ('bs', ctypes.c_uint8), ('pad', ctypes.c_uint8*2)]


class DSynthBand(ctypes.Structure):
    """od_hip_dsynth_band."""
    _fields_ = [('block', ctypes.c_uint32), ('band', ctypes.c_uint8), ('mode', ctypes.c_uint8),
                ('pad', ctypes.c_uint8*2), ('yoff', ctypes.c_uint32), ('pad2', ctypes.c_uint32),
                ('g', ctypes.c_double), ('sin_theta', ctypes.c_double), ('cos_theta', ctypes.c_double)]


DSYNTH_ZERO, DSYNTH_NOREF, DSYNTH_REF, DSYNTH_WIDE = 0, 1, 2, 4


class DSynth:
    """od_hip_dsynth_*: decoder-side PVQ synthesis of an inter frame into the coefficient planes
    of slot 0 of `ctx` (a DaalaHip whose slot 0 holds the forward pyramid of the prediction)."""

    def __init__(self, ctx):
        self.lib, self.ctx = ctx.lib, ctx
        L = self.lib
        L.od_hip_dsynth_create.restype = ctypes.c_void_p
        L.od_hip_dsynth_create.argtypes = [ctypes.c_void_p]
        L.od_hip_dsynth_destroy.argtypes = [ctypes.c_void_p]
        L.od_hip_dsynth_set_level.argtypes = [ctypes.c_void_p, c_int, c_int, I16P, I16P]
        L.od_hip_dsynth_ref_gains.argtypes = [ctypes.c_void_p, ctypes.POINTER(F64P*4)]
        L.od_hip_dsynth_buffers.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.POINTER(DSynthBlock)),
                                            ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.POINTER(DSynthBand)),
                                            ctypes.POINTER(ctypes.c_long), ctypes.POINTER(I16P),
                                            ctypes.POINTER(ctypes.c_long)]
        L.od_hip_dsynth_run.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_long]
        self.h = L.od_hip_dsynth_create(ctx.ctx)
        if not self.h:
            raise HipError(L.od_hip_last_error().decode())
        b, nb = ctypes.POINTER(DSynthBlock)(), ctypes.c_long()
        r, nr = ctypes.POINTER(DSynthBand)(), ctypes.c_long()
        y, ny = I16P(), ctypes.c_long()
        _chk(L.od_hip_dsynth_buffers(self.h, ctypes.byref(b), ctypes.byref(nb), ctypes.byref(r), ctypes.byref(nr),
                                     ctypes.byref(y), ctypes.byref(ny)))
        self.blocks, self.max_blocks = b, nb.value
        self.bands, self.max_bands = r, nr.value
        self.pulses, self.max_pulses = y, ny.value

    def close(self):
        if self.h:
            self.lib.od_hip_dsynth_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_level(self, pli, level, qm, qm_inv):
        qm = np.ascontiguousarray(qm, np.int16)
        qm_inv = np.ascontiguousarray(qm_inv, np.int16)
        _chk(self.lib.od_hip_dsynth_set_level(self.h, pli, level, qm.ctypes.data_as(I16P),
                                              qm_inv.ctypes.data_as(I16P)))

    def ref_gains(self):
        """{(pli, level): gr [nbands, nblk]} of the pyramid in slot 0."""
        arr = ((F64P*4)*3)()
        _chk(self.lib.od_hip_dsynth_ref_gains(self.h, arr))
        out = {}
        for pli in range(3):
            for level in range(self.ctx.nlevels(pli)):
                n = (32 >> self.ctx.xdec[pli]) >> level
                nb = {4: 1, 8: 4, 16: 7, 32: 9}[n]
                nblk = self.ctx.pvq_nblocks(pli, level)
                out[pli, level] = np.ctypeslib.as_array(arr[pli][level], shape=(nb, nblk)).copy()
        return out

    def run(self, nblocks, nbands, npulses):
        _chk(self.lib.od_hip_dsynth_run(self.h, nblocks, nbands, npulses))
