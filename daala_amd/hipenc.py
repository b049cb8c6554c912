"""ctypes binding of daala_amd/host/hip_enc_glue.h: the reference encoder/decoder with
the batched frame seam bound to libdaala_hip.so (INTEGRATION.md seam 2) - the drop-in
behind daala_encode_img_in() / daala_decode_packet_in().

The shared library (daala_amd/host/build/libdaala_hipenc.so) links the reference's own
host code (compiled where it lies under /root/reference by daala_amd/host/Makefile,
never copied) with our C glue; build/ is git-ignored and travels to the GPU box as a
binary.  Python is only the harness (tests, bench.py, tools/daala_hip_cli.py)."""
import ctypes
import os

import numpy as np

HIPENC_SO = os.environ.get('OD_HIPENC_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'host', 'build',
                                                             'libdaala_hipenc.so')
I32P = ctypes.POINTER(ctypes.c_int32)
I16P = ctypes.POINTER(ctypes.c_int16)
U8P = ctypes.POINTER(ctypes.c_uint8)
F64P = ctypes.POINTER(ctypes.c_double)


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

c_int = ctypes.c_int
c_int32 = ctypes.c_int32


class Params(ctypes.Structure):
    _fields_ = [('pic_width', c_int32), ('pic_height', c_int32), ('quant', c_int32),
                ('complexity', c_int32), ('masking', c_int32), ('nworkers', c_int32),
                ('check', c_int32), ('batch', c_int32), ('keyframe_rate', c_int32)]


class Stats(ctypes.Structure):
    _fields_ = [('dev_hits', ctypes.c_int64), ('cpu_noref_luma', ctypes.c_int64),
                ('cpu_other', ctypes.c_int64), ('g2_mismatch', ctypes.c_int64),
                ('lost_sync', ctypes.c_int64), ('check_fail', ctypes.c_int64),
                ('resampled', ctypes.c_int64), ('pvq_check_fail', ctypes.c_int64),
                ('search_cpu_s', ctypes.c_double), ('search_class_s', ctypes.c_double*4),
                ('fdct_hits', ctypes.c_int64), ('haar_hits', ctypes.c_int64),
                ('fdct_check_fail', ctypes.c_int64),
                ('dering_dev_sbs', ctypes.c_int64), ('dering_check_fail', ctypes.c_int64),
                ('dist_dev', ctypes.c_int64), ('dist_check_fail', ctypes.c_int64),
                ('pfeed_frames', ctypes.c_int64), ('t_pfeed_s', ctypes.c_double),
                ('rate_s', ctypes.c_double), ('rate_state_free_s', ctypes.c_double),
                ('rate_calls', ctypes.c_int64), ('frame_cpu_s', ctypes.c_double), ('pre_mc_s', ctypes.c_double),
                ('mv_stage_s', ctypes.c_double*8),
                ('mv_dev_calls', ctypes.c_int64), ('mv_dev_sads', ctypes.c_int64), ('mv_dev_wait_s', ctypes.c_double), ('mv_check_fail', ctypes.c_int64),
                ('mv_bma_calls', ctypes.c_int64), ('mv_bma_windows', ctypes.c_int64), ('mv_bma_hits', ctypes.c_int64), ('mv_bma_misses', ctypes.c_int64), ('mv_level_walks', ctypes.c_int64),
                ('t_setup_s', ctypes.c_double),
                ('t_upload_s', ctypes.c_double), ('t_launch_s', ctypes.c_double),
                ('t_compand_s', ctypes.c_double),
                ('t_total_s', ctypes.c_double), ('pkt_bytes_needed', ctypes.c_int64)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k in ('search_class_s', 'mv_stage_s') else getattr(self, k))
                for k, _ in self._fields_}


from daala_amd.binding import FeedLevel     # od_hip_feed_level: ONE mirror of the C struct


_lib = None


def have_hipenc():
    return os.path.exists(HIPENC_SO)


def hipenc():
    global _lib
    if _lib is None:
        if not have_hipenc():
            raise RuntimeError('daala_amd/host/build/libdaala_hipenc.so is not built (it needs the '
                               'reference sources: `make -C daala_amd/host` in the dev container)')
        lib = ctypes.CDLL(HIPENC_SO)
        lib.od_hipenc_encode_frames.restype = ctypes.c_long
        lib.od_hipenc_encode_frames.argtypes = [ctypes.POINTER(Params), c_int, U8P,
                                                ctypes.POINTER(FeedLevel), c_int, c_int, U8P,
                                                ctypes.c_long, ctypes.POINTER(Stats)]
        lib.od_hipenc_open.restype = ctypes.c_void_p
        lib.od_hipenc_open.argtypes = [ctypes.POINTER(Params), c_int, c_int, ctypes.POINTER(c_int)]
        lib.od_hipenc_encode.restype = ctypes.c_long
        lib.od_hipenc_encode.argtypes = [ctypes.c_void_p, c_int, ctypes.c_long, U8P,
                                         ctypes.POINTER(FeedLevel), U8P, ctypes.c_long,
                                         ctypes.POINTER(Stats)]
        lib.od_hipenc_close.argtypes = [ctypes.c_void_p]
        lib.od_hipenc_level_params.argtypes = [ctypes.POINTER(Params), I16P, I32P, F64P]
        lib.od_hipenc_pad_frame.argtypes = [ctypes.POINTER(Params), U8P,
                                            ctypes.POINTER(U8P), ctypes.POINTER(c_int),
                                            ctypes.POINTER(c_int)]
        _lib = lib
    return _lib


def pack_frames(frames, w, h):
    """frames: list of [Y, U, V] arrays (at least picture size) -> dense 4:2:0 buffer."""
    cw, ch = (w + 1)//2, (h + 1)//2
    for f in frames:     # a short plane would make every later frame start at the wrong offset
        assert f[0].shape[0] >= h and f[0].shape[1] >= w and min(f[1].shape[0], f[2].shape[0]) >= ch \
            and min(f[1].shape[1], f[2].shape[1]) >= cw, 'frame planes smaller than the picture'
    return np.ascontiguousarray(np.concatenate(
        [np.concatenate([f[0][:h, :w].ravel(), f[1][:ch, :cw].ravel(), f[2][:ch, :cw].ravel()])
         for f in frames]))


def split_packets(buf, count):
    out, o = [], 0
    for _ in range(count):
        n = int.from_bytes(buf[o:o + 4].tobytes(), 'little')
        out.append(buf[o + 4:o + 4 + n].tobytes())
        o += 4 + n
    return out


def level_params(prm):
    qm = np.zeros((4, 1024), np.int16)
    q = np.zeros((4, 11), np.int32)
    beta = np.zeros((4, 11), np.float64)
    rc = hipenc().od_hipenc_level_params(ctypes.byref(prm), p16(qm), p32(q), pf64(beta))
    assert rc == 0
    return qm, q, beta


def pad_frame(prm, frame):
    """The reference's padded input planes of one dense 4:2:0 frame."""
    lib = hipenc()
    fw, fh = c_int(), c_int()
    assert lib.od_hipenc_pad_frame(ctypes.byref(prm), pu8(frame), None, ctypes.byref(fw),
                                   ctypes.byref(fh)) == 0
    fw, fh = fw.value, fh.value
    planes = [np.zeros((fh, fw), np.uint8), np.zeros((fh//2, fw//2), np.uint8),
              np.zeros((fh//2, fw//2), np.uint8)]
    arr = (U8P*3)(*[pu8(p) for p in planes])
    assert lib.od_hipenc_pad_frame(ctypes.byref(prm), pu8(frame), arr, None, None) == 0
    return planes


ENOSPC = -11


def encode(prm, frames_buf, nframes, views=None, use_device=0, device=0, out_cap=None):
    lib = hipenc()
    out = np.zeros(max(1 << 20, frames_buf.size) if out_cap is None else out_cap, np.uint8)
    st = Stats()
    varr = None
    if views is not None:
        varr = (FeedLevel*(4*nframes))()
        for f in range(nframes):
            for l in range(4):
                varr[4*f + l] = views[f].levels[l]
    n = lib.od_hipenc_encode_frames(ctypes.byref(prm), nframes, pu8(frames_buf), varr,
                                    use_device, device, pu8(out), out.size, ctypes.byref(st))
    if n == ENOSPC and out_cap is None:
        # all-or-nothing: the call reports the size it needs (never a partial stream)
        return encode(prm, frames_buf, nframes, views, use_device, device, int(st.pkt_bytes_needed))
    if n < 0:
        return n, None, st
    assert st.pkt_bytes_needed == n + 4*nframes <= out.size
    return n, split_packets(out, nframes), st


class Session:
    """od_hipenc_open / od_hipenc_encode / od_hipenc_close: persistent workers + device
    context; encode() codes one stream of independent keyframes per call."""

    def __init__(self, prm, use_device=0, device=0):
        self.lib = hipenc()
        self.prm = prm
        err = c_int()
        self.h = self.lib.od_hipenc_open(ctypes.byref(prm), use_device, device, ctypes.byref(err))
        if not self.h:
            raise RuntimeError('od_hipenc_open failed: %d' % err.value)

    def encode(self, frames_buf, nframes, views=None, frame0=0, out=None):
        if out is None:
            out = np.zeros(max(1 << 20, frames_buf.size), np.uint8)
        st = Stats()
        varr = None
        if views is not None:
            varr = (FeedLevel*(4*nframes))()
            for f in range(nframes):
                for l in range(4):
                    varr[4*f + l] = views[f].levels[l]
        n = self.lib.od_hipenc_encode(self.h, nframes, frame0, pu8(frames_buf), varr, pu8(out),
                                      out.size, ctypes.byref(st))
        if n == ENOSPC:
            return self.encode(frames_buf, nframes, views, frame0,
                               np.zeros(int(st.pkt_bytes_needed), np.uint8))
        if n < 0:
            return n, None, st
        assert st.pkt_bytes_needed == n + 4*nframes <= out.size
        return n, split_packets(out, nframes), st

    def close(self):
        if self.h:
            self.lib.od_hipenc_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def headers(prm):
    lib = hipenc()
    lib.od_hipenc_headers.restype = ctypes.c_long
    lib.od_hipenc_headers.argtypes = [ctypes.POINTER(Params), U8P, ctypes.c_long]
    out = np.zeros(1 << 16, np.uint8)
    n = lib.od_hipenc_headers(ctypes.byref(prm), pu8(out), out.size)
    assert n > 0, n
    return out[:n].copy()


def join_packets(packets):
    return np.frombuffer(b''.join(len(p).to_bytes(4, 'little') + p for p in packets), np.uint8).copy()


def decode(prm, hdr, packets, use_device=0, device=0):
    """Decodes keyframe packets -> (n, frames [nframes, frame_bytes] u8, seconds, device_seconds)."""
    return decode_blob(prm, hdr, join_packets(packets), len(packets), use_device, device)


def mc_stats():
    """(inter frames predicted on the device, check-mode mismatches) of the last decode."""
    out = (ctypes.c_long*2)()
    hipenc().od_hipdec_mc_stats(out)
    return int(out[0]), int(out[1])


def md_stats():
    """(prediction-side transforms of P frames served from the device pyramid, check-mode
    mismatches) of the last decode."""
    out = (ctypes.c_long*2)()
    hipenc().od_hipdec_md_stats(out)
    return int(out[0]), int(out[1])


def synth_stats():
    """(P frames of the last decode whose PVQ synthesis ran on the device - the host parsed
    symbols only -, check-mode mismatches of reference gains / coefficient planes)."""
    out = (ctypes.c_long*3)()
    hipenc().od_hipdec_synth_stats(out)
    return int(out[0]), int(out[1])


def synth_wide_bands():
    """Bands of the last decode whose pulses did not fit 16 bits (two entries per pulse)."""
    out = (ctypes.c_long*3)()
    hipenc().od_hipdec_synth_stats(out)
    return int(out[2])


def ref_resident_frames():
    """Frames of the last decode whose reconstruction became a reference image on the device."""
    f = hipenc().od_hipdec_ref_resident_frames
    f.restype = ctypes.c_long
    return int(f())


def tail_frames():
    """Frames of the last decode whose pixel-domain stage (od_hip_decode_tail) ran on the device."""
    f = hipenc().od_hipdec_tail_frames
    f.restype = ctypes.c_long
    return int(f())


def decode_blob(prm, hdr, buf, nframes, use_device=0, device=0):
    """The same on a length-prefixed packet blob as od_hipenc_encode_frames writes it."""
    lib = hipenc()
    lib.od_hipdec_decode_frames.restype = ctypes.c_long
    lib.od_hipdec_decode_frames.argtypes = [ctypes.POINTER(Params), U8P, ctypes.c_long, c_int, U8P,
                                            ctypes.c_long, c_int, c_int, U8P, F64P, F64P]
    w, h = prm.pic_width, prm.pic_height
    fb = w*h + 2*((w + 1)//2)*((h + 1)//2)
    buf = np.ascontiguousarray(buf)
    hdr = np.ascontiguousarray(hdr)
    out = np.zeros((nframes, fb), np.uint8)
    sec, dsec = ctypes.c_double(), ctypes.c_double()
    n = lib.od_hipdec_decode_frames(ctypes.byref(prm), pu8(hdr), hdr.size, nframes, pu8(buf),
                                    buf.size, use_device, device, pu8(out), ctypes.byref(sec),
                                    ctypes.byref(dsec))
    return n, out, sec.value, dsec.value
