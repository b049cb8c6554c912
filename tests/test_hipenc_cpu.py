"""CPU tests of the reference-side glue (daala_amd/host/hip_enc_glue.c) with the
oracle standing in for the device feed: the call-sequence logic that maps
pvq_search_rdo_double calls to feed candidates, the multi-worker driver and the
frame-index priming must reproduce the sequential reference encoder byte for byte."""
import ctypes

import numpy as np
import pytest

from testlib import synth_plane, ref, pu8
import hipenc_lib as H

pytestmark = pytest.mark.skipif(not H.have_hipenc(), reason='daala_amd/host/build/libdaala_hipenc.so not built')


def setup_frames(w, h, seeds):
    fr = [[synth_plane(w, h, s), synth_plane(w//2, h//2, s, 1), synth_plane(w//2, h//2, s + 1, 1)]
          for s in seeds]
    return H.pack_frames(fr, w, h)


def reference_packets(buf, w, h, nf, masking):
    lib = ref('enc_probe')
    lib.probe_encode_frames.restype = ctypes.c_long
    out = np.zeros(1 << 22, np.uint8)
    fnv, sec = ctypes.c_uint(), ctypes.c_double()
    n = lib.probe_encode_frames(w, h, nf, 20, 7, masking, 1, pu8(buf), ctypes.byref(fnv),
                                ctypes.byref(sec), pu8(out), out.size)
    assert n > 0
    return H.split_packets(out, nf)


def test_workers_reproduce_sequential_stream():
    """12 frames on 3 independent contexts == one context fed in order (covers the
    golden-frame flag at frame 10)."""
    w, h, nf = 64, 64, 12
    buf = setup_frames(w, h, range(1, nf + 1))
    want = reference_packets(buf, w, h, nf, 1)
    prm = H.Params(w, h, 20, 7, 1, 3, 0, 0)
    n, got, st = H.encode(prm, buf, nf)
    assert got == want and st.dev_hits == 0


@pytest.mark.parametrize('masking', [1, 0])
def test_oracle_feed_gives_identical_packets(masking):
    w, h, nf = 352, 288, 2
    buf = setup_frames(w, h, [3, 4])
    want = reference_packets(buf, w, h, nf, masking)
    prm = H.Params(w, h, 20, 7, masking, 2, 1, 0)
    lp = H.level_params(prm)
    fb = w*h*3//2
    views = [H.OracleFeed(prm, H.pad_frame(prm, buf[f*fb:(f + 1)*fb])[0], lp) for f in range(nf)]
    n, got, st = H.encode(prm, buf, nf, views)
    assert got == want
    assert st.check_fail == 0 and st.lost_sync == 0 and st.g2_mismatch == 0
    assert st.dev_hits > 50000
    # the luma forward transforms of both passes (8x8 and larger) came from the feed pyramid
    assert st.fdct_hits > 4000 and st.fdct_check_fail == 0


def test_corrupt_feed_is_detected_not_trusted():
    """A feed whose K does not match the call falls out of step: the glue must run the
    C search for that block (packets stay identical) and count it."""
    w, h = 352, 288
    buf = setup_frames(w, h, [3])
    want = reference_packets(buf, w, h, 1, 0)
    prm = H.Params(w, h, 20, 7, 0, 1, 0, 0)
    view = H.OracleFeed(prm, H.pad_frame(prm, buf)[0])
    view.keep[1]['k'][::7] += 1
    n, got, st = H.encode(prm, buf, 1, [view])
    assert got == want and st.lost_sync > 0


def test_corrupt_pulses_are_detected_not_trusted():
    """A pulse vector that is not a K-pulse codeword (or a cosine outside [0, 1]) fails the
    integrity check of the feed consumer: the band is searched on the host (packets stay
    identical) and counted."""
    w, h = 352, 288
    buf = setup_frames(w, h, [3])
    want = reference_packets(buf, w, h, 1, 1)
    prm = H.Params(w, h, 20, 7, 1, 1, 0, 0)
    view = H.OracleFeed(prm, H.pad_frame(prm, buf)[0])
    y = view.keep[2]['y']
    idx = np.flatnonzero(y)[::5]
    y[idx] += np.sign(y[idx])                            # one pulse too many: no K-pulse codeword
    view.keep[0]['cos_dist'][::11] = 1.5
    n, got, st = H.encode(prm, buf, 1, [view])
    assert got == want and st.lost_sync > 100


def test_plausible_but_wrong_feed_is_caught_by_the_sampled_research(monkeypatch):
    """A cosine distance that is wrong but plausible passes the integrity checks; the
    sampled re-search (HIPENC_SAMPLE, here every 4th candidate) reports it."""
    w, h = 352, 288
    buf = setup_frames(w, h, [3])
    prm = H.Params(w, h, 20, 7, 1, 1, 0, 0)
    view = H.OracleFeed(prm, H.pad_frame(prm, buf)[0])
    monkeypatch.setenv('HIPENC_SAMPLE', '4')
    n, got, st = H.encode(prm, buf, 1, [view])
    assert st.resampled > 1000 and st.check_fail == 0
    cd = view.keep[1]['cos_dist']
    cd[cd > 0] *= 0.999
    n, got, st = H.encode(prm, buf, 1, [view])
    assert st.check_fail > 100


def test_corrupt_pyramid_is_caught_by_check_mode():
    """The transform coefficients are taken from the feed without an independent host
    value (unlike K): the OD_CHECKASM-style check mode is what catches a wrong pyramid."""
    w, h = 352, 288
    buf = setup_frames(w, h, [3])
    prm = H.Params(w, h, 20, 7, 0, 1, 1, 0)
    view = H.OracleFeed(prm, H.pad_frame(prm, buf)[0])
    view.keep[2]['lev'][40, 72] += 1
    n, got, st = H.encode(prm, buf, 1, [view])
    assert st.fdct_check_fail > 0


def test_no_device_is_loud():
    import daala_amd.binding as b
    if b.load().od_hip_device_count() > 0:
        pytest.skip('a HIP device is present')
    w, h = 64, 64
    buf = setup_frames(w, h, [1])
    prm = H.Params(w, h, 20, 7, 1, 1, 0, 0)
    n, got, st = H.encode(prm, buf, 1, use_device=1)
    assert n == -30 and got is None


def test_decoder_driver_host_path_matches_encoder_reconstruction():
    """od_hipdec_decode_frames without a device is the plain reference decoder on N
    workers: its last picture must equal the encoder's own reconstruction (the
    invariant of the reference's OD_ENCODER_CHECK), and the pixel-stage entry points
    bound by hip_dec_glue.c must forward to the reference's code on encoder threads
    (the packet tests above run through them)."""
    w, h, nf = 176, 144, 3
    buf = setup_frames(w, h, [5, 6, 7])
    prm = H.Params(w, h, 20, 7, 1, 2, 0, 0)
    n, pk, st = H.encode(prm, buf, nf)
    assert pk == reference_packets(buf, w, h, nf, 1)
    nd, out, sec, dsec = H.decode(prm, H.headers(prm), pk)
    assert nd == nf and dsec == 0
    lib = ref('enc_probe')
    lib.probe_encode_frames_vtbl.restype = ctypes.c_long
    rec = np.zeros(w*h*3//2, np.uint8)
    fnv, s = ctypes.c_uint(), ctypes.c_double()
    lib.probe_encode_frames_vtbl(w, h, nf, 20, 7, 1, 1, pu8(buf), ctypes.byref(fnv),
                                 ctypes.byref(s), None, 0, None, None, pu8(rec))
    assert np.array_equal(out[-1], rec)


def test_decoder_no_device_is_loud():
    import daala_amd.binding as b
    if b.load().od_hip_device_count() > 0:
        pytest.skip('a HIP device is present')
    w, h = 64, 64
    buf = setup_frames(w, h, [1])
    prm = H.Params(w, h, 20, 7, 1, 1, 0, 0)
    n, pk, st = H.encode(prm, buf, 1)
    nd, out, sec, dsec = H.decode(prm, H.headers(prm), pk, use_device=1)
    assert nd == -30


@pytest.mark.timeout(120)
def test_decoder_driver_many_frames_per_worker():
    """One context decoding many keyframes in a row (the reference's two-entry output
    queue must be advanced exactly once per packet) == frames spread over 4 contexts."""
    w, h, nf = 96, 64, 9
    buf = setup_frames(w, h, range(nf))
    prm = H.Params(w, h, 20, 7, 1, 3, 0, 0)
    n, pk, st = H.encode(prm, buf, nf)
    hdr = H.headers(prm)
    outs = []
    for nw in (1, 4):
        prm.nworkers = nw
        nd, out, sec, dsec = H.decode(prm, hdr, pk)
        assert nd == nf
        outs.append(out)
    assert np.array_equal(outs[0], outs[1])


def test_packet_buffer_too_small_is_an_error_not_a_partial_stream():
    """od_hipenc_encode_frames is all-or-nothing: a pkt_cap that cannot hold every packet
    fails with OD_HIP_ENOSPC, reports the size needed and writes nothing."""
    prm = H.Params(96, 64, 20, 7, 1, 2, 0, 0)
    frames = [[synth_plane(128, 64, 40 + f)[:64, :96], synth_plane(64, 32, 50 + f, 1)[:32, :48],
               synth_plane(64, 32, 60 + f, 1)[:32, :48]] for f in range(3)]
    buf = H.pack_frames(frames, 96, 64)
    n, pk, st = H.encode(prm, buf, 3)
    assert n > 0 and st.pkt_bytes_needed == n + 12
    n2, pk2, st2 = H.encode(prm, buf, 3, out_cap=int(st.pkt_bytes_needed) - 1)
    assert n2 == H.ENOSPC and pk2 is None and st2.pkt_bytes_needed == st.pkt_bytes_needed
    n3, pk3, _ = H.encode(prm, buf, 3, out_cap=int(st.pkt_bytes_needed))
    assert n3 == n and pk3 == pk


def test_truncated_or_mismatched_container_is_rejected():
    """The length-prefixed container is checked before the reference decoder sees it: a
    packet length that runs past the blob, a truncated header blob and dimensions that
    are not the stream's all fail with OD_HIP_EINVAL."""
    prm = H.Params(96, 64, 20, 7, 1, 1, 0, 0)
    frames = [[synth_plane(128, 64, 41)[:64, :96], synth_plane(64, 32, 51, 1)[:32, :48],
               synth_plane(64, 32, 61, 1)[:32, :48]]]
    buf = H.pack_frames(frames, 96, 64)
    n, pk, _ = H.encode(prm, buf, 1)
    hdr = H.headers(prm)
    nd, pics, _, _ = H.decode(prm, hdr, pk)
    assert nd == 1
    blob = H.join_packets(pk)
    assert H.decode_blob(prm, hdr, blob[:-5], 1)[0] == -10          # body shorter than its length
    bad = blob.copy()
    bad[0:4] = np.frombuffer((len(pk[0]) + 1000).to_bytes(4, 'little'), np.uint8)
    assert H.decode_blob(prm, hdr, bad, 1)[0] == -10
    assert H.decode_blob(prm, hdr[:-3], blob, 1)[0] == -10          # truncated header packet
    other = H.Params(128, 64, 20, 7, 1, 1, 0, 0)
    assert H.decode_blob(other, hdr, blob, 1)[0] == -10             # not the stream's picture size


def test_host_codeword_search_equals_reference_search():
    """hip_pvq_search.c against the reference's pvq_search_rdo_double: several (K, g2)
    candidates of ONE vector through one search context (what is shared between candidates is
    computed once, the greedy pulses once per distinct K - repeated K's are in the list),
    scalar scans and lane-wise scans + verified winner, on the sizes the host searches
    (with-reference bands have n - 1 coefficients), including inputs full of exact ties."""
    lib = H.hipenc()
    F64P, I32P = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
    lib.od_ref_pvq_search_rdo_double_cpu.restype = ctypes.c_double
    lib.od_ref_pvq_search_rdo_double_cpu.argtypes = [F64P, ctypes.c_int, ctypes.c_int, I32P, ctypes.c_double]
    lib.od_hip_pvq_search_multi.restype = None
    lib.od_hip_pvq_search_multi.argtypes = [F64P, ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P, F64P,
                                            I32P, F64P]
    rng = np.random.default_rng(7)
    for n in (7, 8, 14, 15, 31, 32, 127, 128):
        for trial in range(160):
            kind = trial % 4
            x = rng.laplace(0, 1, n)*rng.choice([.01, 1, 30, 900])
            if kind == 1:
                x = np.round(x)                      # integers: many exact ties and zeros
            elif kind == 2:
                x[rng.random(n) < .6] = 0
            elif kind == 3:
                x = np.full(n, x[0] if x[0] else 1.)  # everything tied
                x[rng.integers(n)] *= -1
            x = np.ascontiguousarray(x)
            nc = 14                                   # more candidates than the context caches
            ks = rng.choice([1, 2, 3, 5, 9, 17, 40, 120], size=nc).astype(np.int32)
            ks[1::3] = ks[0]                          # repeated K, different g2
            g2 = rng.uniform(.01, 50, size=nc)
            g2[4] = g2[1]                             # and an exact repeat
            want_y, want_c = np.zeros((nc, n), np.int32), np.zeros(nc)
            for c in range(nc):
                want_c[c] = lib.od_ref_pvq_search_rdo_double_cpu(
                    x.ctypes.data_as(F64P), n, int(ks[c]), want_y[c].ctypes.data_as(I32P), float(g2[c]))
            for lanes in (0, 1, 2, 3):                # bit 0: greedy phase, bit 1: RDO phase on lanes
                y, cd = np.zeros((nc, n), np.int32), np.zeros(nc)
                lib.od_hip_pvq_search_multi(x.ctypes.data_as(F64P), n, lanes, nc, ks.ctypes.data_as(I32P),
                                            g2.ctypes.data_as(F64P), y.ctypes.data_as(I32P),
                                            cd.ctypes.data_as(F64P))
                assert np.array_equal(y, want_y) and np.array_equal(cd, want_c), (n, trial, lanes)


def inter_stream_frames(w, h, nf):
    """nf dense 4:2:0 frames of content that pans (2, 3) samples per frame over a larger synthetic
    picture and jumps back every 30 frames (a scene cut inside the stream)."""
    base = [synth_plane(w + 96, h + 64, 31), synth_plane(w//2 + 48, h//2 + 32, 32, 1),
            synth_plane(w//2 + 48, h//2 + 32, 33, 1)]
    frames = []
    for f in range(nf):
        dy, dx = 2*(f % 30), 3*(f % 30)//2*2
        frames.append([base[0][dy:dy + h, dx:dx + w], base[1][dy//2:dy//2 + h//2, dx//2:dx//2 + w//2],
                       base[2][dy//2:dy//2 + h//2, dx//2:dx//2 + w//2]])
    buf = H.pack_frames(frames, w, h)
    assert buf.size == nf*(w*h + 2*((w + 1)//2)*((h + 1)//2))
    return buf


def inter_stream(w, h, nf, keyrate=4):
    """A short inter stream (keyframe every `keyrate` frames) from the pure reference encoder,
    of content that moves between frames, plus the encoder's reconstruction of the last frame."""
    base = [synth_plane(w + 64, h + 64, 31), synth_plane(w//2 + 32, h//2 + 32, 32, 1),
            synth_plane(w//2 + 32, h//2 + 32, 33, 1)]
    frames = [[base[0][2*f:2*f + h, 3*f:3*f + w], base[1][f:f + h//2, (3*f)//2:(3*f)//2 + w//2],
               base[2][f:f + h//2, (3*f)//2:(3*f)//2 + w//2]] for f in range(nf)]
    buf = H.pack_frames(frames, w, h)
    lib = ref('enc_probe')
    lib.probe_encode_frames_vtbl.restype = ctypes.c_long
    out = np.zeros(1 << 22, np.uint8)
    rec = np.zeros(w*h*3//2, np.uint8)
    fnv, s = ctypes.c_uint(), ctypes.c_double()
    n = lib.probe_encode_frames_vtbl(w, h, nf, 20, 7, 1, keyrate, pu8(buf), ctypes.byref(fnv),
                                     ctypes.byref(s), pu8(out), out.size, None, None, pu8(rec))
    assert n > 0
    return H.split_packets(out, nf), rec


def test_inter_stream_through_the_decoder_driver_host_path():
    """configs[3] plumbing: an inter stream (P frames: od_dec_mv_unpack + od_state_mc_predict +
    residual) decoded sequentially by one worker of od_hipdec_decode_frames; without a device
    every bound entry point (od_state_mc_predict among them) forwards to the reference's code,
    and the last picture equals the encoder's own reconstruction (OD_ENCODER_CHECK's invariant)."""
    w, h, nf = 96, 64, 5
    pk, rec = inter_stream(w, h, nf)
    prm = H.Params(w, h, 20, 7, 1, 1, 0, 0)
    nd, out, sec, dsec = H.decode(prm, H.headers(prm), pk)
    assert nd == nf
    assert np.array_equal(out[-1], rec)
    # asking for several workers must not split a stream that has P frames
    prm4 = H.Params(w, h, 20, 7, 1, 4, 0, 0)
    nd4, out4, _, _ = H.decode(prm4, H.headers(prm4), pk)
    assert nd4 == nf and np.array_equal(out4, out)


def test_rate_only_coder_equals_reference_od_pvq_rate():
    """od_hip_pvq_rate (hip_pvq_host.c: the (rng, bit count) recurrence alone) against the
    reference's od_pvq_rate (trial coding into a real range encoder) on random pulse vectors,
    sizes, K and adaptation states - including states far from the initial one."""
    lib = H.hipenc()
    I32P = ctypes.POINTER(ctypes.c_int32)
    for f in (lib.od_hip_pvq_rate, lib.od_ref_pvq_rate):
        f.restype = ctypes.c_double
        f.argtypes = [ctypes.c_int]*4 + [ctypes.c_void_p, I32P] + [ctypes.c_int]*5
    lib.od_hipenc_test_adapt.restype = ctypes.c_void_p
    lib.od_hipenc_test_adapt.argtypes = [ctypes.c_uint]
    rng = np.random.default_rng(11)
    checked = 0
    for trial in range(6000):
        adapt = lib.od_hipenc_test_adapt(int(rng.integers(0, 1 << 31)) if trial % 3 else 0)
        bs = int(rng.integers(0, 4))
        noref = int(rng.integers(0, 2))
        n = int(rng.choice([8, 15, 32, 128]))
        k = int(rng.choice([1, 1, 2, 3, 5, 9, 20, 60, 200]))
        nn = n - (0 if noref else 1)
        y = np.zeros(n, np.int32)
        pos = rng.integers(0, nn, size=k) if trial % 5 else rng.integers(0, max(1, nn//4), size=k)
        for p in pos:
            y[p] += 1
        y *= rng.choice([-1, 1], size=n).astype(np.int32)
        theta = -1 if noref else int(rng.integers(0, 6))
        qg = int(rng.integers(0, 4))
        icgr = int(rng.integers(0, 4))
        ts = int(rng.integers(1, 12))
        pli = int(rng.integers(0, 3))
        a = lib.od_hip_pvq_rate(qg, icgr, theta, ts, adapt, y.ctypes.data_as(I32P), k, n, 1, pli, bs)
        b = lib.od_ref_pvq_rate(qg, icgr, theta, ts, adapt, y.ctypes.data_as(I32P), k, n, 1, pli, bs)
        assert a == b, (trial, n, k, noref, bs)
        checked += 1
    assert checked == 6000


def test_inter_stream_encoded_by_the_driver_host_path():
    """keyframe_rate > 1: one worker codes the stream in order; without a device the packets
    of the integration library equal the pure reference encoder's inter stream."""
    w, h, nf = 176, 144, 5
    want, rec = inter_stream(w, h, nf, keyrate=4)
    base = [synth_plane(w + 64, h + 64, 31), synth_plane(w//2 + 32, h//2 + 32, 32, 1),
            synth_plane(w//2 + 32, h//2 + 32, 33, 1)]
    frames = [[base[0][2*f:2*f + h, 3*f:3*f + w], base[1][f:f + h//2, (3*f)//2:(3*f)//2 + w//2],
               base[2][f:f + h//2, (3*f)//2:(3*f)//2 + w//2]] for f in range(nf)]
    buf = H.pack_frames(frames, w, h)
    prm = H.Params(w, h, 20, 7, 1, 3, 0, 0, 4)          # 3 workers asked for, 1 used
    n, pk, st = H.encode(prm, buf, nf)
    assert n > 0 and pk == want


def test_motion_search_leaves_equal_reference():
    """hip_mc_host.c: the host-vector od_state_opt_vtbl leaves of the motion search
    (od_hipenc_mc_blend_full8, od_hipenc_mc_predict1fmv8) against the reference's C entries
    (src/mc.c:352-377, :94-203) - every block shape, every sub-pel phase pair, saturating
    content, strides that are not the block width."""
    lib = H.hipenc()
    U8P = ctypes.POINTER(ctypes.c_uint8)
    f = lib.od_hipenc_mc_leaves_test
    f.restype = None
    f.argtypes = [ctypes.c_int, U8P, ctypes.c_int, U8P, U8P, U8P, U8P, ctypes.c_int, ctypes.c_int32,
                  ctypes.c_int32, ctypes.c_int, ctypes.c_int]
    rng = np.random.default_rng(12)
    for trial in range(600):
        lx, ly = int(rng.integers(2, 7)), int(rng.integers(2, 7))
        n, m = 1 << lx, 1 << ly
        kind = trial % 3
        srcs = [rng.integers(0, 256, size=n*m, dtype=np.uint8) for _ in range(4)]
        if kind == 1:
            srcs = [(255*rng.integers(0, 2, size=n*m)).astype(np.uint8) for _ in range(4)]
        ds = n + int(rng.integers(0, 5))
        got, want = np.zeros((m, ds), np.uint8), np.zeros((m, ds), np.uint8)
        for which, o in ((0, got), (1, want)):
            f(which, pu8(o), ds, pu8(srcs[0]), pu8(srcs[1]), pu8(srcs[2]), pu8(srcs[3]), 0, 0, 0, lx, ly)
        assert np.array_equal(got, want), ('blend', trial, lx, ly)
        # the blend with unsplit edges (od_mc_blend_full_split8_c, src/mc.c:1104): every corner and
        # split state the motion search can ask for
        oc, ss = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        got[:] = 0
        want[:] = 0
        for which, o in ((5, got), (6, want)):
            f(which, pu8(o), ds, pu8(srcs[0]), pu8(srcs[1]), pu8(srcs[2]), pu8(srcs[3]), 0, oc, ss, lx, ly)
        assert np.array_equal(got, want), ('split blend', trial, lx, ly, oc, ss)
    W, Hh, pad = 160, 144, 24
    for trial in range(900):
        lx, ly = int(rng.integers(2, 7)), int(rng.integers(2, 7))
        n, m = 1 << lx, 1 << ly
        plane = rng.integers(0, 256, size=(Hh, W), dtype=np.uint8)
        if trial % 3 == 1:
            plane = (255*rng.integers(0, 2, size=(Hh, W))).astype(np.uint8)
        elif trial % 3 == 2:
            plane[:, ::2] = 255; plane[:, 1::2] = 0      # worst case for the 6-tap sums
        x0, y0 = int(rng.integers(pad, W - pad - n)), int(rng.integers(pad, Hh - pad - m))
        mvx, mvy = int(rng.integers(-8*(pad - 8), 8*(pad - 8))), int(rng.integers(-8*(pad - 8), 8*(pad - 8)))
        if trial % 5 == 0:
            mvx &= ~7
        elif trial % 5 == 1:
            mvy &= ~7
        if not (mvx & 7 or mvy & 7):
            mvx |= 3                                 # the full-pel copy goes through state->opt_vtbl
        src = ctypes.cast(plane.ctypes.data + y0*W + x0, U8P)
        got, want = np.zeros((m, n), np.uint8), np.zeros((m, n), np.uint8)
        again = np.zeros((m, n), np.uint8)
        for which, o in ((2, got), (3, want), (4, again)):     # 4: the repeat, from the cache
            f(which, pu8(o), n, src, src, src, src, W, mvx, mvy, lx, ly)
        assert np.array_equal(got, want), ('predict', trial, lx, ly, mvx & 7, mvy & 7)
        assert np.array_equal(again, want), ('cached', trial)
    lib.od_hipenc_mc_cache_stats.argtypes = [ctypes.POINTER(ctypes.c_int64)]*2
    hits, misses = ctypes.c_int64(), ctypes.c_int64()
    lib.od_hipenc_mc_cache_stats(ctypes.byref(hits), ctypes.byref(misses))
    assert hits.value >= 900 and misses.value >= 900


def test_host_vector_transforms_equal_reference():
    """hip_dct_host.c (the 2-D lifting DCTs that stay on the host, eight columns per vector,
    generated from the same step lists as the device kernels) against the reference's
    od_bin_fdctNxN / od_bin_idctNxN: random blocks at coefficient scale and far beyond,
    strides that are not the block width, in-place use."""
    lib = H.hipenc()
    r = ref()
    I32P = ctypes.POINTER(ctypes.c_int32)
    rng = np.random.default_rng(31)
    for n in (4, 8, 16, 32):
        fh = getattr(lib, 'od_hipenc_fdct%dx%d' % (n, n))
        ih = getattr(lib, 'od_hipenc_idct%dx%d' % (n, n))
        fr = getattr(r, 'od_bin_fdct%dx%d' % (n, n))
        ir = getattr(r, 'od_bin_idct%dx%d' % (n, n))
        for f in (fh, ih, fr, ir):
            f.restype = None
            f.argtypes = [I32P, ctypes.c_int, I32P, ctypes.c_int]
        for trial in range(300):
            amp = [255 << 4, 40000, 1 << 22][trial % 3]
            xs, ys = n + int(rng.integers(0, 7)), n + int(rng.integers(0, 7))
            x = rng.integers(-amp, amp + 1, size=(n, xs)).astype(np.int32)
            ya, yb = np.zeros((n, ys), np.int32), np.zeros((n, ys), np.int32)
            fh(ya.ctypes.data_as(I32P), ys, x.ctypes.data_as(I32P), xs)
            fr(yb.ctypes.data_as(I32P), ys, x.ctypes.data_as(I32P), xs)
            assert np.array_equal(ya[:, :n], yb[:, :n]), ('fdct', n, trial)
            xa, xb = np.zeros((n, xs), np.int32), np.zeros((n, xs), np.int32)
            ih(xa.ctypes.data_as(I32P), xs, yb.ctypes.data_as(I32P), ys)
            ir(xb.ctypes.data_as(I32P), xs, yb.ctypes.data_as(I32P), ys)
            assert np.array_equal(xa[:, :n], xb[:, :n]) and np.array_equal(xa[:, :n], x[:, :n]), ('idct', n, trial)
            # arbitrary (not transform-generated) coefficients through the inverse
            c = rng.integers(-amp, amp + 1, size=(n, ys)).astype(np.int32)
            ih(xa.ctypes.data_as(I32P), xs, c.ctypes.data_as(I32P), ys)
            ir(xb.ctypes.data_as(I32P), xs, c.ctypes.data_as(I32P), ys)
            assert np.array_equal(xa[:, :n], xb[:, :n]), ('idct arbitrary', n, trial)
            # in place (the encoder's od_compute_dist transforms a stack block in place? no - but
            # od_bin_* allow y == x only through z; exercise it anyway for the 2-D wrappers)
            za, zb = x[:, :n].copy(), x[:, :n].copy()
            fh(za.ctypes.data_as(I32P), n, za.ctypes.data_as(I32P), n)
            fr(zb.ctypes.data_as(I32P), n, zb.ctypes.data_as(I32P), n)
            assert np.array_equal(za, zb), ('in place', n, trial)


def test_motion_search_stage_timers_and_host_path():
    """The build recipe pipes src/mcenc.c through the same rebinding as encode.c (mcenc_head.h /
    mcenc_tail.c).  Without a device od_mv_est_calc_sads runs the reference's own loop - packets equal
    the pure reference encoder's - and the stage timers account for od_mv_est: every stage measured,
    their sum below the whole, no device call counted."""
    w, h, nf = 96, 64, 4
    want, rec = inter_stream(w, h, nf, keyrate=4)
    base = [synth_plane(w + 64, h + 64, 31), synth_plane(w//2 + 32, h//2 + 32, 32, 1),
            synth_plane(w//2 + 32, h//2 + 32, 33, 1)]
    frames = [[base[0][2*f:2*f + h, 3*f:3*f + w], base[1][f:f + h//2, (3*f)//2:(3*f)//2 + w//2],
               base[2][f:f + h//2, (3*f)//2:(3*f)//2 + w//2]] for f in range(nf)]
    buf = H.pack_frames(frames, w, h)
    prm = H.Params(w, h, 20, 7, 1, 1, 0, 0, 4)
    n, pk, st = H.encode(prm, buf, nf)
    assert n > 0 and pk == want
    s = list(st.mv_stage_s)
    assert s[7] > 0 and s[0] > 0 and s[2] > 0 and s[6] > 0          # whole, EPZS, calc_sads, sub-pel
    assert sum(s[:5]) + s[6] <= s[7]*1.001                            # the refinement loop is the rest
    assert st.mv_dev_calls == 0 and st.mv_dev_sads == 0 and st.mv_check_fail == 0


def test_motion_search_level_by_level_walk_equals_the_reference_walk(monkeypatch):
    """od_mv_est_init_mvs walks the vector grid block by block in the reference (src/mcenc.c:3036);
    with a device the integration library walks it LEVEL BY LEVEL (mcenc_tail.c) so that a level's
    block-matching windows are one device call.  HIPENC_MV_EPZS=2 forces that walk without a device
    (host SADs): the packets of inter streams - the 33-frame GOP across golden frames and the second
    keyframe among them - must equal the pure reference encoder's."""
    from test_gpu_hipenc import ref_encode
    monkeypatch.setenv('HIPENC_MV_EPZS', '2')
    for w, h, nf, keyrate in ((176, 144, 33, 30), (150, 100, 9, 3)):
        buf = inter_stream_frames(w, h, nf)
        want = ref_encode(w, h, buf, nf, keyrate)
        prm = H.Params(w, h, 20, 7, 1, 1, 0, 0, keyrate)
        n, pk, st = H.encode(prm, buf, nf)
        assert n > 0 and pk == want
        nkey = (nf + keyrate - 1)//keyrate
        assert st.mv_level_walks >= nf - nkey and st.mv_bma_windows == 0      # every P frame, at least the previous reference
