#!/usr/bin/env python3
"""End-to-end intra encode of BASELINE configs[1] through the batched frame seam:
30 synthetic 1080p 4:2:0 frames, device feed + N host workers running the reference
encoder's serial stage (daala_amd/host/build/libdaala_hipenc.so).  Prints one JSON object.
  python tools/e2e_encode.py [--frames 30] [--workers 16] [--ref-frames 2] [--no-verify]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=30)
    ap.add_argument('--workers', type=int, default=16)
    ap.add_argument('--ref-frames', type=int, default=3)
    ap.add_argument('--batch', type=int, default=0)
    ap.add_argument('--masking', type=int, default=1)
    ap.add_argument('--size', default='1920x1080', help='picture size WxH (default: BASELINE configs[1])')
    ap.add_argument('--host-only', action='store_true', help='same driver, plain C search (no device)')
    ap.add_argument('--decode', action='store_true', help='also decode the packets (device tail vs reference)')
    args = ap.parse_args()
    import daala_amd.hipenc as H
    import bench
    PIC_W, PIC_H = (int(v) for v in args.size.split('x'))
    if (PIC_W, PIC_H) != (bench.PIC_W, bench.PIC_H):
        bench.FW, bench.FH = (PIC_W + 63)//64*64, (PIC_H + 63)//64*64     # generator works on padded planes
    frames = bench.make_frames(args.frames, seed0=1)
    buf = H.pack_frames(frames, PIC_W, PIC_H)
    res = {'frames': args.frames, 'workers': args.workers, 'pic': [PIC_W, PIC_H]}
    px = PIC_W*PIC_H
    # single-thread reference on the first frames (same driver, 1 worker, plain C search)
    p1 = H.Params(PIC_W, PIC_H, 20, 7, args.masking, 1, 0, 0)
    n0, pk0, st0 = H.encode(p1, buf, args.ref_frames)
    res['reference_1thread'] = {'Mpixels_per_s': round(args.ref_frames*px/st0.t_total_s/1e6, 3),
                                'frames': args.ref_frames, 's': round(st0.t_total_s, 3),
                                'search_share': round(st0.search_cpu_s/st0.t_total_s, 3)}
    prm = H.Params(PIC_W, PIC_H, 20, 7, args.masking, args.workers, 0, args.batch)
    t0 = time.perf_counter()
    n, pk, st = H.encode(prm, buf, args.frames, use_device=0 if args.host_only else 1)
    wall = time.perf_counter() - t0
    if n < 0:
        raise SystemExit('encode failed: %d' % n)
    res['hip'] = {'Mpixels_per_s': round(args.frames*px/st.t_total_s/1e6, 3),
                  'packet_bytes': int(n), 'wall_incl_setup_s': round(wall, 3), **{
                      k: (round(v, 4) if isinstance(v, float) else [round(x, 3) for x in v] if isinstance(v, list) else v)
                      for k, v in st.as_dict().items()}}
    res['bit_exact_vs_reference'] = pk[:args.ref_frames] == pk0
    res['speedup_vs_1thread'] = round(res['hip']['Mpixels_per_s']/res['reference_1thread']['Mpixels_per_s'], 2)
    if args.decode:
        sys.stderr.write('decode: headers\n'); sys.stderr.flush()
        hdr = H.headers(prm)
        p1.nworkers = 1
        sys.stderr.write('decode: reference 1 thread\n'); sys.stderr.flush()
        n1, want, s1, _ = H.decode(p1, hdr, pk[:4])
        sys.stderr.write('decode: device %d workers\n' % args.workers); sys.stderr.flush()
        nd, pics, sd, dsec = H.decode(prm, hdr, pk, use_device=0 if args.host_only else 1)
        sys.stderr.write('decode: done\n'); sys.stderr.flush()
        res['decode'] = {'reference_1thread_Mpixels_per_s': round(4*px/s1/1e6, 2),
                         'hip_Mpixels_per_s': round(args.frames*px/sd/1e6, 2), 'seconds': round(sd, 3),
                         'device_call_seconds': round(dsec, 3),
                         'identical': bool(nd == args.frames and np.array_equal(pics[:4], want))}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
