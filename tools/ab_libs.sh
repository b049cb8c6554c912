#!/bin/bash
# A/B of kernel build variants on the GPU box: runs the device-only bench section once per
# library given (build_ab/*.so, built with different -D switches) and prints the per-kernel
# average durations - first with the PVQ launches serialised on one stream (OD_HIP_PVQ_STREAMS=0:
# clean per-kernel times), then as the product runs them (side streams: the phase time counts).
# Usage (inside gpurun): bash tools/ab_libs.sh daala_amd/libdaala_hip.so build_ab/libB.so ...
for lib in "$@"; do
  for streams in 0 3; do
  OD_HIP_PVQ_STREAMS=$streams OD_HIP_LIB=$PWD/$lib python3 bench.py --device-only --device-steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ds=d['device_step']
print('$lib', 'streams=$streams', 'step %.3f ms' % ds['ms_per_step'], 'pvq %.3f ms' % ds['pvq']['ms_per_step'], ' '.join('%s=%.3f' % (k.replace('k_pvq_',''), v['avg_ms']) for k,v in ds['kernels'].items() if 'pvq' in k), 'tail %.3f' % ds['decode_tail']['ms_per_30_frames'], 'frac %.4f' % ds['pvq']['roofline']['frac'])
"
  done
done
