#!/usr/bin/env python3
"""The device-only step of bench.py (per-kernel HIP-event spans) before and after a 30-worker
device session in the same process: the experiment behind DESIGN.md section 4's "60-100 us per
launch once a session with >= 8 workers has run".
  OD_HIP_STREAM_POOL=0 python tools/launch_gap_probe.py    one stream per worker object (rounds 2-3)
  OD_HIP_STREAM_POOL=4 python tools/launch_gap_probe.py    worker objects lease from a pool of 4
Prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def brief(ds):
    k = ds['kernels']
    return {'ms_per_step': ds['ms_per_step'], 'kernel_spans_ms': ds['kernel_spans_ms_per_step'],
            'host_issue_ms': ds['host_issue_ms_per_step'],
            'pyramid_luma_ms': k['k_forward_pyramid_luma']['avg_ms'], 'gain15_ms': k['k_pvq_gain<15>']['avg_ms'],
            'gain8_ms': k['k_pvq_gain<8>']['avg_ms'], 'pvq_phase_ms': ds['pvq']['ms_per_step']}


def main():
    import bench
    import daala_amd.hipenc as H
    workers = int(os.environ.get('PROBE_WORKERS', '30'))
    frames = bench.make_frames(bench.FRAMES, seed0=1)
    res = {'OD_HIP_STREAM_POOL': os.environ.get('OD_HIP_STREAM_POOL', '(default 4)'), 'workers': workers}
    ds, _ = bench.device_step(0, frames, 0, 5, 2, False, 1)
    res['before'] = brief(ds)
    buf = H.pack_frames(frames, bench.PIC_W, bench.PIC_H)
    prm = H.Params(bench.PIC_W, bench.PIC_H, 20, 7, 1, workers, 0, bench.FRAMES)
    with H.Session(prm, use_device=1, device=0) as ses:
        t = time.perf_counter()
        n, pk, st = ses.encode(buf, bench.FRAMES)
        res['session_step_s'] = round(time.perf_counter() - t, 3)
        ds, _ = bench.device_step(0, frames, 0, 5, 2, False, 1)
        res['after_session_open'] = brief(ds)
    ds, _ = bench.device_step(0, frames, 0, 5, 2, False, 1)
    res['after_session_closed'] = brief(ds)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
