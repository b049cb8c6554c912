#!/usr/bin/env python3
"""All five BASELINE.json configs through the live seams on the GPU box, bounded samples of each,
parity against the pure reference build (oracle/_ref/enc_probe.so) and timings, one JSON object
on stdout (committed as profiles/rNN_configs.json).
  python tools/configs_round.py [--k4-frames 60] [--inter-frames 4] [--workers 30]
configs[2] is run on ONE GPU here (whole frames; its 8-GPU superblock-row shard is the strip API,
tests/test_gpu_parity.py); configs[3] is a bounded GOP prefix (the reference's P frames take
tens of seconds each at 1080p)."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

U8P = ctypes.POINTER(ctypes.c_uint8)


def frames_of(w, h, count, seed0, step=(3, 5)):
    from testlib import synth_plane
    fw, fh = (w + 31) & ~31, (h + 31) & ~31
    base = [synth_plane(fw + 64, fh + 64, seed0), synth_plane(fw//2 + 32, fh//2 + 32, seed0, 1),
            synth_plane(fw//2 + 32, fh//2 + 32, seed0 + 1, 1)]
    out = []
    for f in range(count):
        dy, dx = step[0]*f % 60, step[1]*f % 60
        out.append([base[0][dy:dy + h, dx:dx + w], base[1][dy//2:dy//2 + h//2, dx//2:dx//2 + w//2],
                    base[2][dy//2:dy//2 + h//2, dx//2:dx//2 + w//2]])
    return out


def reference(w, h, buf, nframes, quant, masking, keyrate, complexity=7):
    import daala_amd.hipenc as H
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', '_ref', 'enc_probe.so'))
    lib.probe_encode_frames.restype = ctypes.c_long
    out = np.zeros(max(1 << 22, buf.size*2), np.uint8)
    fnv, sec = ctypes.c_uint(), ctypes.c_double()
    n = lib.probe_encode_frames(w, h, nframes, quant, complexity, masking, keyrate, buf.ctypes.data_as(U8P),
                                ctypes.byref(fnv), ctypes.byref(sec), out.ctypes.data_as(U8P), out.size)
    assert n > 0
    return H.split_packets(out, nframes), sec.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--k4-frames', type=int, default=60)
    ap.add_argument('--k4-ref-frames', type=int, default=2)
    ap.add_argument('--inter-frames', type=int, default=4)
    ap.add_argument('--lossless-frames', type=int, default=4)
    ap.add_argument('--workers', type=int, default=30)
    ap.add_argument('--no-device', action='store_true', help='dry run of the script on a CPU-only box')
    ap.add_argument('--scale', type=int, default=1, help='divide picture sizes (dry runs)')
    a = ap.parse_args()
    import daala_amd.hipenc as H
    res = {}
    DEV = 0 if a.no_device else 1
    sc = a.scale

    def mpix(w, h, n, s):
        return round(n*w*h/s/1e6, 3)

    # configs[0]: one CIF frame
    w, h = 352, 288
    buf = H.pack_frames(frames_of(w, h, 1, 11), w, h)
    want, rs = reference(w, h, buf, 1, 20, 1, 1)
    prm = H.Params(w, h, 20, 7, 1, 1, 0, 0)
    n, got, st = H.encode(prm, buf, 1, use_device=DEV)
    res['configs0_cif_1frame'] = {'packets_equal_reference': got == want, 'lost_sync': int(st.lost_sync),
                                  'reference_1thread_s': round(rs, 3), 'seam_s': round(st.t_total_s, 3)}
    print('configs[0]', res['configs0_cif_1frame'], file=sys.stderr, flush=True)

    # configs[2]: 4K, whole frames on one GPU
    w, h, nf = 3840//sc, 2160//sc//8*8, a.k4_frames
    fr = frames_of(w, h, nf, 23)
    buf = H.pack_frames(fr, w, h)
    prm = H.Params(w, h, 20, 7, 1, a.workers, 0, 0)
    t0 = time.time()
    n0, host, st0 = H.encode(prm, buf, nf)
    th = time.time() - t0
    n1, got, st = H.encode(prm, buf, nf, use_device=DEV)
    want, rs = reference(w, h, buf[:(w*h*3//2)*a.k4_ref_frames], a.k4_ref_frames, 20, 1, 1)
    res['configs2_4k'] = {
        'frames': nf, 'workers': a.workers,
        'all_packets_equal_reference_code_without_device': got == host,
        'first_packets_equal_pure_reference': got[:a.k4_ref_frames] == want, 'pure_reference_frames': a.k4_ref_frames,
        'lost_sync': int(st.lost_sync), 'g2_mismatch': int(st.g2_mismatch),
        'Mpixels_per_s': mpix(w, h, nf, st.t_total_s), 'host_only_Mpixels_per_s': mpix(w, h, nf, st0.t_total_s),
        'reference_1thread_Mpixels_per_s': mpix(w, h, a.k4_ref_frames, rs),
        'what': 'one GPU, whole frames; the superblock-row shard of configs[2] is od_hip_set_strip + '
                'od_hip_gather_strips (tests/test_gpu_parity.py), unmeasured without an 8-GPU node'}
    hdr = H.headers(prm)
    nd8 = min(8, nf)
    nd0, pics0, s0, _ = H.decode(prm, hdr, got[:nd8])
    nd1, pics1, s1, _ = H.decode(prm, hdr, got[:nd8], use_device=DEV)
    res['configs2_4k']['decode_first_pictures_identical'] = bool(nd0 == nd8 and nd1 == nd8 and np.array_equal(pics0, pics1))
    print('configs[2]', res['configs2_4k'], file=sys.stderr, flush=True)
    del fr, buf, host, got, pics0, pics1

    # configs[3]: inter, 1080p, bounded GOP prefix (I P P ...), frames in order on one worker
    w, h, nf = 1920//sc, 1080//sc//8*8, a.inter_frames
    buf = H.pack_frames(frames_of(w, h, nf, 31, step=(2, 3)), w, h)
    want, rs = reference(w, h, buf, nf, 20, 1, 30)
    prm = H.Params(w, h, 20, 7, 1, 1, 0, 0, 30)
    n, got, st = H.encode(prm, buf, nf, use_device=DEV)
    hdr = H.headers(prm)
    nd0, pics0, s0, _ = H.decode(prm, hdr, want)
    nd1, pics1, s1, _ = H.decode(prm, hdr, want, use_device=DEV)
    frames_dev, bad = H.mc_stats()
    md_hits, _ = H.md_stats()
    tail = H.tail_frames()
    res['configs3_inter_1080p'] = {
        'frames': nf, 'keyframe_rate': 30, 'packets_equal_pure_reference': got == want,
        'reference_1thread_s': round(rs, 2), 'seam_s': round(st.t_total_s, 2),
        'decode_pictures_identical': bool(nd0 == nf and nd1 == nf and np.array_equal(pics0, pics1)),
        'decode_reference_s': round(s0, 3), 'decode_seam_s': round(s1, 3),
        'mc_frames_on_device_in_decode': int(frames_dev), 'mc_check_fail': int(bad),
        'tail_frames_on_device_in_decode': int(tail), 'prediction_transforms_from_device_pyramid': int(md_hits),
        'references_resident_on_device_in_decode': int(H.ref_resident_frames()),
        'mv_est_calc_sads_device_calls': int(st.mv_dev_calls), 'mv_est_block_sads_on_device': int(st.mv_dev_sads),
        'mv_est_seconds_per_P_frame': round(st.mv_stage_s[7]/max(1, nf - 1), 4),
        'decode_seam_ms_per_frame': round(1e3*s1/nf, 2), 'decode_reference_ms_per_frame': round(1e3*s0/nf, 2),
        'what': 'P frames: od_state_mc_predict (OBMC, all planes) on the device in both seams; decoder: '
                'forward pyramid of the prediction and the whole pixel-domain stage on the device too; '
                'encoder: deringing and its distortions on the device, P-frame feed, od_mv_est_calc_sads as one '
                'fused OBMC + SAD call per frame; the EPZS initialisation of od_mv_est reads the block-matching SADs of levels >= 1 from device windows; its DP refinement stays reference host code'}
    print('configs[3]', res['configs3_inter_1080p'], file=sys.stderr, flush=True)

    # configs[4]: lossless
    w, h, nf = 1920//sc, 1080//sc//8*8, a.lossless_frames
    buf = H.pack_frames(frames_of(w, h, nf, 41), w, h)
    want, rs = reference(w, h, buf, nf, 0, 1, 1)
    prm = H.Params(w, h, 0, 7, 1, min(nf, a.workers), 0, 0)
    n, got, st = H.encode(prm, buf, nf, use_device=DEV)
    hdr = H.headers(prm)
    nd1, pics1, s1, _ = H.decode(prm, hdr, got, use_device=DEV)
    res['configs4_lossless_1080p'] = {
        'frames': nf, 'packets_equal_pure_reference': got == want, 'haar_planes_from_device': int(st.haar_hits),
        'decoded_equals_input': bool(nd1 == nf and np.array_equal(pics1.ravel(), buf)),
        'reference_1thread_Mpixels_per_s': mpix(w, h, nf, rs), 'seam_Mpixels_per_s': mpix(w, h, nf, st.t_total_s),
        'decode_seam_Mpixels_per_s': mpix(w, h, nf, s1)}
    print('configs[4]', res['configs4_lossless_1080p'], file=sys.stderr, flush=True)
    print(json.dumps(res))
    ok = (res['configs0_cif_1frame']['packets_equal_reference']
          and res['configs2_4k']['all_packets_equal_reference_code_without_device']
          and res['configs2_4k']['first_packets_equal_pure_reference']
          and res['configs3_inter_1080p']['packets_equal_pure_reference']
          and res['configs4_lossless_1080p']['packets_equal_pure_reference']
          and res['configs4_lossless_1080p']['decoded_equals_input'])
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
