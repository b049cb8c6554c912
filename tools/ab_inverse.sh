#!/bin/bash
# like ab_libs.sh, for the inverse kernels
for lib in "$@"; do
  OD_HIP_LIB=$PWD/$lib python3 bench.py --device-only --device-steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['device_step']['kernels']
print('$lib', ' '.join('%s=%.4f' % (n.replace('k_inverse_',''), v['avg_ms']) for n,v in k.items() if 'inverse' in n))"
done
