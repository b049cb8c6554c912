#!/bin/bash
# Run on the GPU box: one rocprofv3 --pmc pass of the bench (counters given as args),
# output under gpurun_out/pmc_$TAG.  Usage: tools/pmc_pass.sh TAG COUNTER [COUNTER ...]
set -e
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k[:60].ljust(60), ' '.join('%s=%.4g' % (c, sum(v)/len(v)) for c, v in sorted(d.items())))
PY
