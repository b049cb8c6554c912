"""N > 1 host logic on CPU: superblock-row sharding with halo rows + all_gather of
the strips (gloo), and the independent-frame partition.  Compute is the oracle."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from daala_amd import sharding

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partitions():
    for nvsb in (1, 7, 34, 68):
        for world in (1, 2, 3, 8):
            rows = []
            for r in range(world):
                a, b = sharding.sb_row_partition(nvsb, world, r)
                rows += list(range(a, b))
            assert rows == list(range(nvsb))
    sh = sharding.SbRowShard(1920, 1080, 1920, 1088, 8, 7)
    pw, ph, fw, fh = sh.strip_geometry()
    assert fh == (sh.h1 - sh.h0)*32 and 0 <= ph <= fh
    assert sh.own_rows_in_frame(0)[1] == 1088


@pytest.mark.parametrize('world', (2, 3))
def test_sb_row_sharding_and_gather_gloo(world):
    port = free_port()
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, MASTER_ADDR='127.0.0.1')
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, '_shard_worker.py'), str(r),
                                   str(world), str(port), d], env=env) for r in range(world)]
        for p in procs:
            assert p.wait(timeout=300) == 0
        for r in range(world):
            assert open(os.path.join(d, 'rank%d.txt' % r)).read() == 'ok'
