#!/bin/bash
# Run on the GPU box (via gpurun): collects the per-round profile set into
# gpurun_out/prof_$1/ :  kernel-trace stats of the default bench, FETCH_SIZE and
# WRITE_SIZE in SEPARATE --pmc passes (TCC slots, MI355X_MICROARCH.md), plus the
# counter calibration run.  Usage: tools/profile_round.sh r01
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --device-only --device-steps 3 > $OUT/bench_profiled.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --device-only --device-steps 1 > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --device-only --device-steps 1 > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- python3 tools/calibrate_traffic.py > /dev/null 2> $OUT/calib_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- python3 tools/calibrate_traffic.py > /dev/null 2> $OUT/calib_write.err
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "profile set written to $OUT"
