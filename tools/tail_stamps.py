#!/usr/bin/env python3
"""Where a wave of k_decode_tail spends its life (diagnostic build of the library with
-DTAIL_STAMPS: s_memtime stamps per phase of the luma plane, one wave in 16 sampled).
  OD_HIP_LIB=build_ab/libtailstamps.so python tools/tail_stamps.py"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PHASES = ['tile load', 'direction search', 'thresholds + skip test', 'direction filter', 'orthogonal filter',
          'smoothing', 'clamp + pack + store']


def main():
    import numpy as np
    import bench
    import daala_amd.binding as b
    lib = b.load()
    frames = bench.make_frames(bench.FRAMES, seed0=1)
    ds, _ = bench.device_step(0, frames, 0, 1, 1, True, 1)      # runs the decode tail section too
    out = (ctypes.c_ulonglong*16)()
    lib.od_hip_tail_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    assert lib.od_hip_tail_stamps(out) == 0
    n = out[15]
    res = {'waves_sampled': int(n), 'decode_tail': ds.get('decode_tail'),
           'cycles_per_wave': {PHASES[i]: round(out[i]/max(n, 1)) for i in range(7)}}
    tot = sum(out[i] for i in range(7))
    res['share'] = {PHASES[i]: round(out[i]/max(tot, 1), 3) for i in range(7)}
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
