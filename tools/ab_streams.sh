for st in 3 6 8 12 16; do
OD_HIP_PVQ_STREAMS=$st OD_HIP_LIB=$PWD/build_ab/libB.so python3 bench.py --device-only --device-steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ds=d['device_step']
print('streams=$st', 'step %.3f ms' % ds['ms_per_step'], 'pvq %.3f ms' % ds['pvq']['ms_per_step'], 'frac %.4f' % ds['pvq']['roofline']['frac'])
"
done
