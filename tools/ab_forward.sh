#!/bin/bash
# A/B of transform-kernel build variants on the GPU box: the device-only bench section once per
# library, alternating, printing the average duration of every transform kernel.
# Usage (inside gpurun): bash tools/ab_forward.sh daala_amd/libdaala_hip.so build_ab/libB.so [rounds]
A=$1; B=$2; R=${3:-2}
for r in $(seq 1 $R); do
  for lib in $A $B; do
    OD_HIP_LIB=$PWD/$lib python3 bench.py --device-only --device-steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ds=d['device_step']
print('$lib', 'step %.3f ms' % ds['ms_per_step'], ' '.join('%s=%.4f' % (k.replace('k_',''), v['avg_ms']) for k,v in ds['kernels'].items() if 'pvq' not in k), 'tail %.4f' % ds['decode_tail']['ms_per_30_frames'], 'roofline %.4f' % d['roofline']['frac'])
"
  done
done
