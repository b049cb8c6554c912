#!/usr/bin/env python3
"""Which HIP runtime serves the process, and what a device-only step costs with it.
torch 2.10+rocm7.0 bundles its own libamdhip64.so (soname libamdhip64.so.7, the same soname as
the image's ROCm 7.2 runtime): whichever is mapped first serves every HIP client of the process.
  PROBE_ORDER=torch_first | lib_first | no_torch   python tools/runtime_probe.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def mapped():
    out = set()
    for line in open('/proc/self/maps'):
        if 'libamdhip64' in line or 'libhsa-runtime64' in line:
            out.add(line.split()[-1])
    return sorted(out)


def main():
    order = os.environ.get('PROBE_ORDER', 'no_torch')
    if order == 'torch_first':
        import torch
        torch.cuda.is_available()
    import daala_amd.binding as b
    b.load()
    if order == 'lib_first':
        import torch
        torch.cuda.is_available()
        torch.cuda.synchronize()
    import bench
    frames = bench.make_frames(bench.FRAMES, seed0=1)
    ds, _ = bench.device_step(0, frames, 0, 5, 2, False, 1)
    print(json.dumps({'order': order, 'runtime': mapped(), 'ms_per_step': ds['ms_per_step'],
                      'host_issue_ms_per_step': ds['host_issue_ms_per_step'],
                      'kernel_spans_ms_per_step': ds['kernel_spans_ms_per_step'],
                      'pvq_phase_ms': ds['pvq']['ms_per_step']}))


if __name__ == '__main__':
    main()
