#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of the device-only bench section for one library variant
# (OD_HIP_LIB), into gpurun_out/traffic_$1/{fetch,write}.  Usage: tools/traffic_ab.sh tag lib.so
set -e
TAG=$1; LIB=$2
OUT=gpurun_out/traffic_$TAG
mkdir -p $OUT
export OD_HIP_LIB=$PWD/$LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --device-only --device-steps 1 > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --device-only --device-steps 1 > /dev/null 2> $OUT/write.err
python3 - <<PY
import csv,glob,collections
for kind in ('fetch','write'):
    f=glob.glob('$OUT/'+kind+'/*/*_counter_collection.csv')[0]
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if 'inverse' in r['Kernel_Name']:
            agg[r['Kernel_Name']]=float(r['Counter_Value']); n[r['Kernel_Name']]+=1
    for k,v in sorted(agg.items()): print('$TAG', kind, k[:44], '%.1f' % v, 'raw units (last dispatch)')
PY
