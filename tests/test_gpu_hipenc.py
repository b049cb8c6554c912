"""Batched frame seam, end to end on the device (INTEGRATION.md seam 2): the real
reference encoder (daala_amd/host/build/libdaala_hipenc.so, built in the dev container, runs
here as a prebuilt binary) takes every keyframe-luma no-reference PVQ search from the
device feed and must produce the packets of the pure-C reference, byte for byte."""
import ctypes
import os
import sys

import numpy as np
import pytest

from testlib import synth_plane, ref, pu8
import hipenc_lib as H

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not H.have_hipenc(), reason='daala_amd/host/build/libdaala_hipenc.so not built')]


def frames_of(w, h, seeds):
    return [[synth_plane(w, h, s), synth_plane(w//2, h//2, s, 1), synth_plane(w//2, h//2, s + 1, 1)]
            for s in seeds]


def reference_packets(buf, w, h, nf, masking):
    lib = ref('enc_probe')
    lib.probe_encode_frames.restype = ctypes.c_long
    out = np.zeros(max(1 << 22, buf.size), np.uint8)
    fnv, sec = ctypes.c_uint(), ctypes.c_double()
    n = lib.probe_encode_frames(w, h, nf, 20, 7, masking, 1, pu8(buf), ctypes.byref(fnv),
                                ctypes.byref(sec), pu8(out), out.size)
    assert n > 0
    return H.split_packets(out, nf)


def ref_encode(w, h, buf, nf, keyrate, quant=20):
    """Packets of the PURE reference encoder (oracle/_ref/enc_probe.so) for an inter stream."""
    lib = ref('enc_probe')
    lib.probe_encode_frames.restype = ctypes.c_long
    out = np.zeros(max(1 << 22, buf.size), np.uint8)
    fnv, sec = ctypes.c_uint(), ctypes.c_double()
    n = lib.probe_encode_frames(w, h, nf, quant, 7, 1, keyrate, pu8(buf), ctypes.byref(fnv),
                                ctypes.byref(sec), pu8(out), out.size)
    assert n > 0
    return H.split_packets(out, nf)


@pytest.mark.parametrize('masking', [1, 0])
def test_device_feed_equals_oracle_feed(masking):
    import daala_amd.binding as b
    w, h = 352, 288
    prm = H.Params(w, h, 20, 7, masking, 1, 0, 0)
    buf = H.pack_frames(frames_of(w, h, [5]), w, h)
    planes = H.pad_frame(prm, buf)
    lp = H.level_params(prm)
    want = H.OracleFeed(prm, planes[0], lp)
    fh, fw = planes[0].shape
    ctx = b.DaalaHip(w, h, fw, fh, nplanes=3, xdec=(0, 1, 1), nslots=2)
    ctx.upload_planes(1, planes)
    ctx.enc_feed_create(*lp)
    ctx.enc_feed_run(1, 1)
    got = ctx.enc_feed_view(1)
    for l in range(4):
        a, g = want.keep[l], got[l]
        nrec = g['nbands']*g['nblk']
        assert np.array_equal(a['lev'], g['lev']), ('pyramid plane', l)
        assert np.array_equal(a['ncand'], g['ncand'])
        assert np.array_equal(a['g'].view(np.int64), g['g'].view(np.int64)), ('gain', l)
        # beta = 1.5 too: the companding pow is the host's libm between the device passes
        assert np.array_equal(a['cg'].view(np.int64), g['cg'].view(np.int64)), ('cg', l)
        for c in range(2):
            live = a['ncand'] > c
            for key in ('qg', 'k'):
                assert np.array_equal(a[key][c*nrec:(c + 1)*nrec][live], g[key][c*nrec:(c + 1)*nrec][live])
            assert np.array_equal(a['cos_dist'][c*nrec:(c + 1)*nrec][live].view(np.int64),
                                  g['cos_dist'][c*nrec:(c + 1)*nrec][live].view(np.int64))
        # pulses of live candidates
        for bnd in range(g['nbands']):
            nn = (g['off'][bnd + 1] - g['off'][bnd] + 1) & ~1          # runs padded to even
            base = 2*g['nblk']*(0 if bnd == 0 else g['off'][bnd])
            for c in range(2):
                live = a['ncand'][bnd*g['nblk']:(bnd + 1)*g['nblk']] > c
                ya = a['y'][base + c*g['nblk']*nn: base + (c + 1)*g['nblk']*nn].reshape(-1, nn)
                yg = g['y'][base + c*g['nblk']*nn: base + (c + 1)*g['nblk']*nn].reshape(-1, nn)
                assert np.array_equal(ya[live], yg[live])
    ctx.close()


@pytest.mark.parametrize('masking', [1, 0])
def test_hip_encoder_packets_identical_cif(masking):
    w, h, nf = 352, 288, 3
    buf = H.pack_frames(frames_of(w, h, [1, 2, 3]), w, h)
    want = reference_packets(buf, w, h, nf, masking)
    prm = H.Params(w, h, 20, 7, masking, 3, 1, 2)     # 3 workers, check mode, batches of 2
    n, got, st = H.encode(prm, buf, nf, use_device=1)
    assert n > 0, n
    assert got == want
    assert st.dev_hits > 0 and st.check_fail == 0 and st.lost_sync == 0
    assert st.fdct_hits > 0 and st.fdct_check_fail == 0
    assert st.dering_dev_sbs > 0 and st.dering_check_fail == 0
    # A22 in the live encoder: both od_compute_dist calls of every deringed superblock of the
    # on/off loop come from the device pass, bit-identical to the reference's function
    assert st.dist_dev > 0 and st.dist_dev % 2 == 0 and st.dist_check_fail == 0
    assert st.g2_mismatch == 0 and st.pvq_check_fail == 0


def test_chroma_planes_in_the_keyframe_feed(monkeypatch):
    """Keyframes: the chroma planes' forward transforms and no-reference candidates come from the
    device like luma's (od_hip_enc_feed_set_level_plane).  Check mode re-computes every transform
    and re-searches every candidate taken from the feed on the host (fdct_check_fail, check_fail,
    g2_mismatch); packets identical to the pure reference with the planes in the feed and without
    (HIPENC_CHROMA_FEED=0), more searches and transforms served by the device with them."""
    w, h, nf = 352, 288, 2
    buf = H.pack_frames(frames_of(w, h, [11, 12]), w, h)
    for masking in (1, 0):
        want = reference_packets(buf, w, h, nf, masking)
        prm = H.Params(w, h, 20, 7, masking, 2, 1, 0)
        n1, got1, st1 = H.encode(prm, buf, nf, use_device=1)
        monkeypatch.setenv('HIPENC_CHROMA_FEED', '0')
        n0, got0, st0 = H.encode(prm, buf, nf, use_device=1)
        monkeypatch.delenv('HIPENC_CHROMA_FEED')
        assert n1 > 0 and got1 == want and got0 == want
        for st in (st0, st1):
            assert st.check_fail == 0 and st.fdct_check_fail == 0 and st.g2_mismatch == 0
            assert st.pvq_check_fail == 0 and st.lost_sync == 0
        assert st1.dev_hits > st0.dev_hits and st1.fdct_hits > st0.fdct_hits
        assert st1.dev_hits + st1.cpu_other + st1.cpu_noref_luma == st0.dev_hits + st0.cpu_other + st0.cpu_noref_luma


def test_hip_encoder_packets_identical_1080p():
    w, h, nf = 1920, 1080, 2
    buf = H.pack_frames(frames_of(w, h, [7, 8]), w, h)
    prm = H.Params(w, h, 20, 7, 1, 2, 0, 0)
    want = reference_packets(buf, w, h, nf, 1)           # the PURE reference build (oracle/_ref, -std=c89 -O2)
    n, got, st = H.encode(prm, buf, nf, use_device=1)
    assert n > 0 and got == want
    assert st.dev_hits > 0 and st.lost_sync == 0
    print('1080p x%d: device feed %.2fs; hits %d, host searches %d, g2 %d'
          % (nf, st.t_total_s, st.dev_hits, st.cpu_other + st.cpu_noref_luma, st.g2_mismatch))


@pytest.mark.parametrize('w,h,quant,masking', [(352, 288, 20, 1), (176, 112, 8, 0), (416, 240, 35, 1)])
def test_hip_decoder_pictures_identical(w, h, quant, masking):
    """daala_decode_packet_in with the pixel-domain stage on the device == the plain
    reference decoder, every output picture byte for byte (3 workers)."""
    nf = 4
    buf = H.pack_frames(frames_of(w, h, [11, 12, 13, 14]), w, h)
    prm = H.Params(w, h, quant, 7, masking, 3, 0, 0)
    n, pk, st = H.encode(prm, buf, nf)
    assert n > 0
    hdr = H.headers(prm)
    n0, want, s0, _ = H.decode(prm, hdr, pk, use_device=0)
    n1, got, s1, ds = H.decode(prm, hdr, pk, use_device=1)
    assert n0 == nf and n1 == nf
    assert np.array_equal(got, want)
    assert ds > 0


def test_hip_decoder_1080p():
    w, h, nf = 1920, 1080, 4
    buf = H.pack_frames(frames_of(w, h, [21, 22, 23, 24]), w, h)
    prm = H.Params(w, h, 20, 7, 1, 4, 0, 0)
    n, pk, st = H.encode(prm, buf, nf, use_device=1)
    assert n > 0
    hdr = H.headers(prm)
    n0, want, s0, _ = H.decode(prm, hdr, pk, use_device=0)
    n1, got, s1, ds = H.decode(prm, hdr, pk, use_device=1)
    assert np.array_equal(got, want)
    print('1080p decode x%d, 4 workers: reference %.3fs, device tail %.3fs (device calls %.3fs)'
          % (nf, s0, s1, ds))


def test_hip_encoder_packets_identical_4k_geometry():
    """BASELINE configs[2] geometry (3840x2160, padded 3840x2176): two frames, two
    workers, batches of one frame (slot reuse across batches) - packets identical to the
    pure reference encoder's."""
    w, h, nf = 3840, 2160, 2
    buf = H.pack_frames(frames_of(w, h, [31, 32]), w, h)
    prm = H.Params(w, h, 20, 7, 1, 2, 0, 1)
    want = reference_packets(buf, w, h, nf, 1)           # the PURE reference build, ~6 s
    n, got, st = H.encode(prm, buf, nf, use_device=1)
    assert n > 0 and got == want
    assert st.dev_hits > 0 and st.lost_sync == 0
    print('4K x%d: device feed %.2fs' % (nf, st.t_total_s))


@pytest.mark.parametrize('w,h,quant,masking', [(355, 291, 20, 1), (64, 48, 1, 1), (130, 66, 60, 0),
                                               (32, 32, 400, 1), (200, 120, 8, 1), (17, 9, 20, 1)])
def test_hip_encoder_odd_geometry_and_quantizers(w, h, quant, masking):
    """Picture sizes that are not multiples of the block sizes (padding, edge gating of
    the split lapping) and quantizer extremes: device feed == plain reference search,
    packets byte for byte, and the decoder seam returns the reference decoder's pictures."""
    nf = 2
    cw, ch = (w + 1)//2, (h + 1)//2
    fr = [[synth_plane(w, h, s), synth_plane(cw, ch, s, 1), synth_plane(cw, ch, s + 1, 1)] for s in (41, 42)]
    buf = H.pack_frames(fr, w, h)
    prm = H.Params(w, h, quant, 7, masking, 2, 1, 0)
    n0, want, st0 = H.encode(prm, buf, nf)
    n, got, st = H.encode(prm, buf, nf, use_device=1)
    assert n == n0 and got == want
    assert st.check_fail == 0 and st.lost_sync == 0
    hdr = H.headers(prm)
    nd0, pics0, _, _ = H.decode(prm, hdr, got, use_device=0)
    nd1, pics1, _, ds = H.decode(prm, hdr, got, use_device=1)
    assert nd0 == nf and nd1 == nf and np.array_equal(pics0, pics1)


def _content(kind, w, h, seed):
    rng = np.random.default_rng(seed)
    cw, ch = (w + 1)//2, (h + 1)//2

    def plane(pw, ph):
        if kind == 'noise':
            return rng.integers(0, 256, (ph, pw), dtype=np.uint8)
        if kind == 'flat':
            return np.full((ph, pw), 128 + seed % 3, np.uint8)
        if kind == 'bilevel':        # saturated 0/255 checker of random cell size: largest coefficients
            c = int(rng.integers(1, 9))
            yy, xx = np.mgrid[0:ph, 0:pw]
            return (((yy//c + xx//c) & 1)*255).astype(np.uint8)
        if kind == 'ramp':
            yy, xx = np.mgrid[0:ph, 0:pw]
            return ((xx*3 + yy*2) & 255).astype(np.uint8)
        if kind == 'sparse':         # flat with a few isolated spikes: mostly zero bands
            p = np.full((ph, pw), 100, np.uint8)
            idx = rng.integers(0, ph*pw, 40)
            p.ravel()[idx] = 255
            return p
        raise ValueError(kind)
    return [plane(w, h), plane(cw, ch), plane(cw, ch)]


@pytest.mark.parametrize('kind', ['noise', 'flat', 'bilevel', 'ramp', 'sparse'])
@pytest.mark.parametrize('quant,masking', [(5, 1), (40, 0), (160, 1)])
def test_hip_encoder_stress_content(kind, quant, masking):
    """Degenerate and extreme pictures (all-small-block noise, all-zero bands, saturated
    0/255 patterns = the largest coefficients the 24-bit multiplier proof covers, isolated
    spikes) through the live encoder seam in check mode: packets identical, every device
    answer equal to the C search, never out of step."""
    w, h, nf = 160, 96, 2
    buf = H.pack_frames([_content(kind, w, h, s) for s in (1, 2)], w, h)
    prm = H.Params(w, h, quant, 7, masking, 2, 1, 0)
    n0, want, st0 = H.encode(prm, buf, nf)
    n, got, st = H.encode(prm, buf, nf, use_device=1)
    assert n == n0 and got == want
    assert st.check_fail == 0 and st.lost_sync == 0 and st.fdct_check_fail == 0
    assert st.dering_check_fail == 0
    if masking == 0:
        assert st.g2_mismatch == 0
    hdr = H.headers(prm)
    nd0, pics0, _, _ = H.decode(prm, hdr, got, use_device=0)
    nd1, pics1, _, _ = H.decode(prm, hdr, got, use_device=1)
    assert nd0 == nf and nd1 == nf and np.array_equal(pics0, pics1)


def test_inter_stream_motion_compensation_on_the_device():
    """configs[3], first live step: an inter stream from the pure reference encoder decoded by
    one worker with the device - every frame's pixel-domain stage through od_hip_decode_tail
    (P frames too: same stage minus the keyframe smoothing), every P frame's
    od_state_mc_predict (OBMC of the whole frame) through od_hip_mc_predict_blocks, in check
    mode (the reference's own prediction is computed beside it and compared).  Pictures must
    equal the plain reference decode, the last one the encoder's reconstruction."""
    from test_hipenc_cpu import inter_stream
    w, h, nf = 352, 288, 6
    pk, rec = inter_stream(w, h, nf, keyrate=4)
    prm = H.Params(w, h, 20, 7, 1, 1, 1, 0)
    hdr = H.headers(prm)
    n0, want, _, _ = H.decode(prm, hdr, pk)
    nd, got, sec, dsec = H.decode(prm, hdr, pk, use_device=1)
    assert n0 == nf and nd == nf
    frames, bad = H.mc_stats()
    assert frames == 4 and bad == 0          # frames 1, 2, 3 and 5 are P frames
    assert H.tail_frames() == nf             # I and P frames: inverse + filters + clamp on the device
    hits, md_bad = H.md_stats()
    assert hits > 0 and md_bad == 0          # check mode: device prediction pyramid == host transforms
    # the P frames' PVQ synthesis (Householder + od_pvq_synthesis_partial + scatter of every coded
    # band, DC, skipped coefficients) ran on the device from parsed symbols; check mode compared
    # every reference gain and the finished coefficient planes with the reference's host path
    synth, synth_bad = H.synth_stats()
    assert synth == 4 and synth_bad == 0
    assert np.array_equal(got, want)
    assert np.array_equal(got[-1], rec)
    # without check mode the prediction side of every P frame is the device pyramid alone: no
    # level plane comes to the host, no coefficient plane goes to the device
    prm.check = 0
    nd, got2, _, _ = H.decode(prm, hdr, pk, use_device=1)
    hits2, _ = H.md_stats()
    assert nd == nf and hits2 == hits and np.array_equal(got2, want)
    assert H.synth_stats() == (4, 0)


def test_inter_decode_1080p_synthesis_on_the_device(monkeypatch):
    """configs[3] at its own size, decoder side: 1920x1080 I P P from the pure reference encoder.
    The P frames' prediction (OBMC), its forward pyramid, the PVQ synthesis of every band and the
    whole pixel-domain stage run on the device; the host parses symbols.  Pictures identical to
    the reference decoder's; the same with the synthesis left on the host (HIPDEC_SYNTH=0)."""
    from test_hipenc_cpu import inter_stream
    w, h, nf = 1920, 1080, 3
    pk, rec = inter_stream(w, h, nf, keyrate=30)
    prm = H.Params(w, h, 20, 7, 1, 1, 0, 0)
    hdr = H.headers(prm)
    n0, want, s0, _ = H.decode(prm, hdr, pk)
    nd, got, s1, _ = H.decode(prm, hdr, pk, use_device=1)
    assert n0 == nf and nd == nf and np.array_equal(got, want) and np.array_equal(got[-1], rec)
    assert H.synth_stats() == (2, 0)
    nd, got, s1, _ = H.decode(prm, hdr, pk, use_device=1)       # warm
    monkeypatch.setenv('HIPDEC_SYNTH', '0')
    nd2, got2, s2, _ = H.decode(prm, hdr, pk, use_device=1)
    nd2, got2, s2, _ = H.decode(prm, hdr, pk, use_device=1)
    assert nd2 == nf and np.array_equal(got2, want) and H.synth_stats() == (0, 0)
    print('1080p I P P decode: reference %.3f s, device with synthesis %.3f s, synthesis on the host %.3f s'
          % (s0, s1, s2))


def test_inter_quantizer_1_single_frequency_patterns():
    """Quantizer 1 on full-swing single-frequency patterns (flat, checkerboard, vertical and
    horizontal stripes in turn - nothing a motion vector can predict): all of a block's energy
    sits in one coefficient and K runs into the tens of thousands: the encoder-side feeds leave
    candidates whose K does not fit their 16-bit pulse records to the host, the decoder's synthesis
    records switch to two entries per pulse when one does not fit (that path itself:
    tests/test_gpu_parity.py::test_decoder_synthesis_frame_vs_oracle).  Packets identical to the
    pure reference encoder's, pictures to its decoder's."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import configs_round as C
    w, h, nf = 96, 64, 4
    yy, xx = np.mgrid[0:h, 0:w]
    pats = [np.full((h, w), 128), 255*((xx + yy) & 1), 255*(xx & 1), 255*(yy & 1)]
    frames = [[p.astype(np.uint8), p[:h//2, :w//2].astype(np.uint8), p[:h//2, :w//2].astype(np.uint8)] for p in pats]
    buf = H.pack_frames(frames, w, h)
    want, _ = C.reference(w, h, buf, nf, 1, 0, 30, 7)
    prm = H.Params(w, h, 1, 7, 0, 1, 1, 0, 30)
    n, got, st = H.encode(prm, buf, nf, use_device=1)
    assert n > 0 and got == want and st.check_fail == 0 and st.pvq_check_fail == 0
    hdr = H.headers(prm)
    n0, p0, _, _ = H.decode(prm, hdr, want)
    for check in (1, 0):
        prm.check = check
        nd, p1, _, _ = H.decode(prm, hdr, want, use_device=1)
        assert nd == nf and np.array_equal(p0, p1)
        assert H.synth_stats() == (3, 0)


def test_stream_ordering_regression_many_workers_repeated():
    """Regression for the round-1 stream-ordering race (a null-stream hipMemset could zero
    dering flags AFTER the non-blocking context stream had uploaded them: rare +-1 pixels in
    multi-worker decodes).  Six workers, each with its own context and
    stream, decode 18 odd-sized frames three times over; every picture of every pass must
    equal the reference decode - a late memset or copy anywhere shows up as a difference."""
    w, h, nf = 200, 136, 18
    buf = H.pack_frames(frames_of(w, h, list(range(40, 40 + nf))), w, h)
    prm = H.Params(w, h, 12, 7, 1, 6, 0, 0)
    n, pk, st = H.encode(prm, buf, nf)
    assert n > 0
    hdr = H.headers(prm)
    n0, want, _, _ = H.decode(prm, hdr, pk, use_device=0)
    assert n0 == nf
    for rep in range(3):
        n1, got, _, ds = H.decode(prm, hdr, pk, use_device=1)
        assert n1 == nf and ds > 0
        assert np.array_equal(got, want), rep


def test_lossless_frames_through_both_seams():
    """BASELINE configs[4]: lossless mode (quantizer 0: Haar wavelet of whole superblocks).
    Encoder seam: od_haar calls answered from the device's Haar planes (check mode compares
    each with the reference's od_haar), packets identical to the pure reference encoder.
    Decoder seam: od_haar_inv + od_coeff_to_ref_plane = one od_hip_inverse_haar per frame;
    pictures identical to the reference decoder AND to the input (lossless round trip)."""
    w, h, nf = 352, 288, 3
    fr = frames_of(w, h, [61, 62, 63])
    buf = H.pack_frames(fr, w, h)
    lib = ref('enc_probe')
    lib.probe_encode_frames.restype = ctypes.c_long
    out = np.zeros(1 << 23, np.uint8)
    fnv, sec = ctypes.c_uint(), ctypes.c_double()
    n = lib.probe_encode_frames(w, h, nf, 0, 7, 1, 1, pu8(buf), ctypes.byref(fnv), ctypes.byref(sec),
                                pu8(out), out.size)
    assert n > 0
    want_pk = H.split_packets(out, nf)
    prm = H.Params(w, h, 0, 7, 1, 2, 1, 0)
    n1, pk, st = H.encode(prm, buf, nf, use_device=1)
    assert n1 > 0 and pk == want_pk
    fw, fh = (w + 63)//64*64, (h + 63)//64*64
    assert st.haar_hits >= nf*((fw//32)*(fh//32) + 2*(fw//32)*(fh//32)) and st.fdct_check_fail == 0
    hdr = H.headers(prm)
    n0, want, _, _ = H.decode(prm, hdr, pk, use_device=0)
    n2, got, _, ds = H.decode(prm, hdr, pk, use_device=1)
    assert n0 == nf and n2 == nf and ds > 0
    assert np.array_equal(got, want)
    assert np.array_equal(got.ravel(), buf)                # lossless: the input itself


def test_inter_stream_encoded_through_the_seam():
    """configs[3] on the ENCODER side: a session with keyframe_rate 4 (one worker, frames in
    order).  Keyframes take the device feed, deringing and distortions as usual; every P frame's
    od_state_mc_predict call (od_predict_frame, src/encode.c:2219: the frame the residual is taken
    from; the calls in src/mcenc.c exist only in OD_ANIMATE builds) runs on the device in check mode.
    The packets must equal the pure reference encoder's inter stream."""
    from test_hipenc_cpu import inter_stream
    w, h, nf = 176, 144, 5
    want, rec = inter_stream(w, h, nf, keyrate=4)
    base = [synth_plane(w + 64, h + 64, 31), synth_plane(w//2 + 32, h//2 + 32, 32, 1),
            synth_plane(w//2 + 32, h//2 + 32, 33, 1)]
    frames = [[base[0][2*f:2*f + h, 3*f:3*f + w], base[1][f:f + h//2, (3*f)//2:(3*f)//2 + w//2],
               base[2][f:f + h//2, (3*f)//2:(3*f)//2 + w//2]] for f in range(nf)]
    buf = H.pack_frames(frames, w, h)
    prm = H.Params(w, h, 20, 7, 1, 3, 1, 0, 4)
    n, pk, st = H.encode(prm, buf, nf, use_device=1)
    assert n > 0 and pk == want
    assert st.pvq_check_fail == 0 and st.check_fail == 0
    # the P frames' bands took the complete candidate lists of the P-frame feed (4d): in check
    # mode every candidate taken from it was searched again on the host and compared
    assert st.pfeed_frames == 3 and st.resampled > 1000
    # od_mv_est_calc_sads of every P frame came from the device (one fused OBMC + SAD call per
    # frame); in check mode the reference's own loop ran beside it on the same grid
    assert st.mv_dev_calls == 3 and st.mv_dev_sads > 0 and st.mv_check_fail == 0
    # the EPZS initialisation walked levels >= 1 against device block-matching windows; in check
    # mode every SAD read from a window was computed again by the reference's od_mv_est_bma_sad
    assert st.mv_level_walks >= 3 and st.mv_bma_windows > 0 and st.mv_bma_hits > st.mv_bma_misses


def test_inter_stream_1080p_pframe_feed_packets_identical():
    """configs[3] at its own size: 1920x1080, I P P (keyframe rate 30), the P frames' pvq_theta
    candidates - with-reference and no-reference, every band of every block size of every plane
    outside the padded superblock row - from the device's P-frame feed.  Packets identical to
    the pure reference encoder's; fewer than a tenth of the P frames' searches left on the host."""
    from test_hipenc_cpu import inter_stream
    w, h, nf = 1920, 1080, 3
    want, rec = inter_stream(w, h, nf, keyrate=30)
    base = [synth_plane(w + 64, h + 64, 31), synth_plane(w//2 + 32, h//2 + 32, 32, 1),
            synth_plane(w//2 + 32, h//2 + 32, 33, 1)]
    frames = [[base[0][2*f:2*f + h, 3*f:3*f + w], base[1][f:f + h//2, (3*f)//2:(3*f)//2 + w//2],
               base[2][f:f + h//2, (3*f)//2:(3*f)//2 + w//2]] for f in range(nf)]
    buf = H.pack_frames(frames, w, h)
    prm = H.Params(w, h, 20, 7, 1, 1, 0, 0, 30)
    n1, pk1, st1 = H.encode(prm, buf, 1, use_device=1)            # the keyframe alone: its host searches
    n, pk, st = H.encode(prm, buf, nf, use_device=1)
    assert n > 0 and pk == want
    assert st.pfeed_frames == 2 and st.lost_sync == st1.lost_sync
    assert st.mv_dev_calls == 2 and st.mv_dev_sads == 2*(4*(120*68 + 60*34) + 30*17)   # two block sizes x four split states + the top-level blocks
    host_p = (st.cpu_other + st.cpu_noref_luma) - (st1.cpu_other + st1.cpu_noref_luma)
    dev_p = st.dev_hits - st1.dev_hits
    print('1080p I P P: P-frame searches from the feed %d, on the host %d; feed wait %.3f s; total %.2f s'
          % (dev_p, host_p, st.t_pfeed_s, st.t_total_s))
    assert host_p < 0.1*(host_p + dev_p)


def test_inter_gop_golden_frames_and_second_keyframe():
    """configs[3] over a whole GOP and beyond: 33 frames of 176x144, keyframe rate 30, one worker,
    check mode.  The stream crosses the golden P frames 10 and 20 (ip_frame_count % 10 == 0,
    src/encode.c:2958-2963: quantizer boost, the golden reference is replaced -> the resident
    reference set's dirty marking), the second keyframe (frame 30) and the first P frames after
    it (whose EPZS candidates still hold the previous GOP's vectors, src/mcenc.c:2757-2762).
    Encoder seam: packets == the pure reference encoder's, every check counter 0; decoder seam
    (check mode too): pictures == the reference decoder's for every frame."""
    from test_hipenc_cpu import inter_stream_frames
    w, h, nf = 176, 144, 33
    buf = inter_stream_frames(w, h, nf)
    want = ref_encode(w, h, buf, nf, keyrate=30)
    prm = H.Params(w, h, 20, 7, 1, 1, 1, 0, 30)
    n, pk, st = H.encode(prm, buf, nf, use_device=1)
    assert n > 0
    bad = [f for f in range(nf) if pk[f] != want[f]]
    d = st.as_dict()
    info = {k: d[k] for k in ('pvq_check_fail', 'check_fail', 'fdct_check_fail', 'dering_check_fail',
                              'dist_check_fail', 'mv_check_fail', 'lost_sync', 'g2_mismatch')}
    assert not bad, (bad, [(len(pk[f]), len(want[f])) for f in bad], info)
    assert st.pvq_check_fail == 0 and st.check_fail == 0 and st.fdct_check_fail == 0
    assert st.dering_check_fail == 0 and st.dist_check_fail == 0 and st.mv_check_fail == 0
    assert st.pfeed_frames == nf - 2 and st.mv_dev_calls == nf - 2
    assert st.mv_level_walks >= nf - 2 and st.mv_bma_windows > 0
    hdr = H.headers(prm)
    n0, pics0, _, _ = H.decode(prm, hdr, want, use_device=0)
    n1, pics1, _, _ = H.decode(prm, hdr, want, use_device=1)
    assert n0 == nf and n1 == nf
    diff = [f for f in range(nf) if not np.array_equal(pics0[f], pics1[f])]
    assert not diff, diff
    frames, mism = H.mc_stats()
    assert frames == nf - 2 and mism == 0        # check mode: every device prediction == the host's
    done, smis = H.synth_stats()
    assert smis == 0
    assert H.tail_frames() == nf
    # every frame after the first became a reference ON the device (od_hip_mc_set_ref_ctx: tail ->
    # edge extension -> resident reference set), the second keyframe and the golden frames too;
    # the predictions checked above were made from those
    assert H.ref_resident_frames() == nf - 1


def test_haar_frames_with_a_quantizer_decode_on_the_host_path(monkeypatch):
    """The Haar-wavelet flag of a frame is a decoded bit, independent of the quantizer
    (src/decode.c:1206).  A stream with the flag set and quantizer 20 (I P P, written by the
    integration library's own reference encoder code through a test hook) never reaches the DCT
    hooks: the decoder glue must notice at the first od_haar / od_haar_inv call, rebuild what it
    had skipped for the device (the prediction planes of a P frame) and let the reference's
    host code decode the frame - pictures identical to the plain decode, no abort, no error."""
    w, h, nf = 176, 144, 3
    buf = H.pack_frames(frames_of(w, h, [71, 72, 73]), w, h)
    monkeypatch.setenv('HIPENC_TEST_HAAR', '1')
    prm = H.Params(w, h, 20, 7, 1, 1, 0, 0, 4)
    n, pk, st = H.encode(prm, buf, nf)
    monkeypatch.delenv('HIPENC_TEST_HAAR')
    assert n > 0
    hdr = H.headers(prm)
    n0, want, _, _ = H.decode(prm, hdr, pk, use_device=0)
    n1, got, _, _ = H.decode(prm, hdr, pk, use_device=1)
    assert n0 == nf and n1 == nf
    assert np.array_equal(got, want)
    assert H.tail_frames() == 0                 # no frame of this stream took the device tail
    frames, bad = H.mc_stats()
    assert frames == 2 and bad == 0             # the P frames' prediction still came from the device


def test_device_failures_surface_as_error_codes_not_aborts():
    """A failed device pass fails the call (daala_decode_packet_in returns OD_EFAULT inside the
    driver, od_hipdec_decode_frames / od_hipenc_encode_frames return a negative code); the
    process lives, and the next call - nothing injected - works."""
    from test_hipenc_cpu import inter_stream
    w, h, nf = 176, 144, 4
    pk, rec = inter_stream(w, h, nf, keyrate=4)
    prm = H.Params(w, h, 20, 7, 1, 1, 0, 0)
    hdr = H.headers(prm)
    lib = H.hipenc()
    for nth in (1, 2):                          # the keyframe's tail pass; the first P frame's prediction
        lib.od_hipdec_test_fail_after(nth)
        n, _, _, _ = H.decode(prm, hdr, pk, use_device=1)
        assert n < 0
    lib.od_hipdec_test_fail_after(0)
    n, got, _, _ = H.decode(prm, hdr, pk, use_device=1)
    assert n == nf and np.array_equal(got[-1], rec)
    # encoder side: the P frame's device prediction fails -> the job fails, the session survives
    base = [synth_plane(w + 64, h + 64, 31), synth_plane(w//2 + 32, h//2 + 32, 32, 1),
            synth_plane(w//2 + 32, h//2 + 32, 33, 1)]
    frames = [[base[0][2*f:2*f + h, 3*f:3*f + w], base[1][f:f + h//2, (3*f)//2:(3*f)//2 + w//2],
               base[2][f:f + h//2, (3*f)//2:(3*f)//2 + w//2]] for f in range(nf)]
    buf = H.pack_frames(frames, w, h)
    eprm = H.Params(w, h, 20, 7, 1, 1, 0, 0, 4)
    with H.Session(eprm, use_device=1) as ses:
        lib.od_hipdec_test_fail_after(1)
        n, _, _ = ses.encode(buf, nf)
        assert n < 0
        lib.od_hipdec_test_fail_after(0)
        n, pk2, _ = ses.encode(buf, nf)       # a new job on the same session: fresh GOP, complete stream
        assert n > 0 and pk2 == pk
