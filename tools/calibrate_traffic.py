#!/usr/bin/env python3
"""Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`: streams 1 GiB with
each access width so the counters can be calibrated (tools/profile_round.sh)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daala_amd.binding as b
lib = b.load()
lib.od_hip_calibrate_traffic.argtypes = [ctypes.c_int, ctypes.c_size_t]
for mode in (0, 1, 2):
    assert lib.od_hip_calibrate_traffic(mode, 1 << 30) == 0
print('calibration kernels ran on 1 GiB')
