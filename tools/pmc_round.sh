#!/bin/bash
# Run on the GPU box (via gpurun): SQ / GRBM counter passes of the device-only bench section
# (one step, 30 frames), each in its own rocprofv3 --pmc run with --kernel-trace only, into
# gpurun_out/pmc_$1/pass{1,2,3}.  tools/summarize_pmc.py turns them into
# profiles/$1_pmc.json (per-kernel instruction counts, wave cycles, VALU utilisation).
set -e
TAG=${1:-r02}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pass1 -- python3 bench.py --device-only --device-steps 1 > $OUT/pass1.json 2> $OUT/pass1.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pass2 -- python3 bench.py --device-only --device-steps 1 > $OUT/pass2.json 2> $OUT/pass2.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pass3 -- python3 bench.py --device-only --device-steps 1 > $OUT/pass3.json 2> $OUT/pass3.err
echo "pmc passes written to $OUT"
