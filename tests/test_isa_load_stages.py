"""Regression guard for a performance property that no GPU is needed to check: the load stages of
the HBM-side kernels issue ALL their global loads before the first wait.

Round 4 found the row-tile kernels and the decoder tail waiting for their loads one by one (a load
behind a lane-dependent branch, or converted where it was loaded, makes the compiler wait on the
spot): 6 to 25 memory round trips in a wave's life, 10-20 % of the kernels' time
(profiles/r04x_ab_load_stages.log).  The kernels were rewritten; this test compiles the device
code to gfx950 assembly (hipcc cross-compiles here, ~20 s) and counts, per kernel, the batches of
global loads that are followed by a wait before the next load leaves."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = '/opt/rocm/bin/hipcc'


def waited_load_batches(asm):
    """{mangled kernel name: (global loads, batches of loads closed by an s_waitcnt vmcnt)}"""
    lines = asm.split('\n')
    out = {}
    starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
    for i, name in starts:
        loads = batches = pending = 0
        for l in lines[i + 1:]:
            st = l.strip()
            if st.startswith('.Lfunc_end'):
                break
            if st.startswith('global_load'):
                loads += 1
                pending += 1
            elif st.startswith('s_waitcnt') and 'vmcnt' in st and pending:
                batches += 1
                pending = 0
        out[name] = (loads, batches)
    return out


@pytest.fixture(scope='module')
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC) or shutil.which('c++filt') is None:
        pytest.skip('needs hipcc and c++filt')
    out = tmp_path_factory.mktemp('isa')/'daala_hip.s'
    src = os.path.join(ROOT, 'daala_amd', 'csrc', 'daala_hip.hip')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-fast-math',
                    '-Wno-unused-function', '--cuda-device-only', '-S', '-o', str(out), src],
                   check=True, capture_output=True, timeout=600)
    stats = waited_load_batches(out.read_text())
    names = list(stats)
    dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True, check=True).stdout.split('\n')
    return {d.strip(): stats[n] for n, d in zip(names, dem)}


def test_load_stages_leave_in_one_batch(isa):
    def batches(prefix):
        hits = [v for k, v in isa.items() if k.startswith(prefix)]
        assert len(hits) == 1, (prefix, [k for k in isa if prefix[:20] in k])
        return hits[0]
    # row-tile kernels: pixels (+ the block-size byte) / the whole coefficient tile = ONE batch
    for k in ('void k_forward_rt<32, 4, false>(', 'void k_forward_rt<32, 4, true>(', 'void k_forward_rt<16, 3, false>(',
              'void k_forward_rt<16, 3, true>(', 'void k_inverse_rt_fused<32, 4>(', 'void k_inverse_rt_fused<16, 3>(',
              'void k_inverse_rt<32, 4>(', 'void k_inverse_rt<16, 3>('):
        loads, nb = batches(k)
        assert loads >= 3 and nb == 1, (k, loads, nb)
    # strip kernel: one batch of raw strip values, then the escape reads (each its own conditional batch)
    for k in ('void k_inverse_strips<32>(', 'void k_inverse_strips<16>('):
        loads, nb = batches(k)
        assert nb <= 8, (k, loads, nb)
    # decoder tail, 4:2:0 instantiation: both input formats (decoder int32 / encoder int16) are in the
    # kernel, each with a head batch and one batch per plane tile; the first single-wave version had 60+
    loads, nb = batches('void k_decode_tail<true>(')
    assert loads >= 100 and nb <= 16, (loads, nb)
    # lossless Haar planes: the superblock's loads in one batch
    for k in ('void k_haar_forward_plane<32>(', 'void k_haar_forward_plane<16>(', 'void k_haar_inverse_plane<32>(',
              'void k_haar_inverse_plane<16>('):
        loads, nb = batches(k)
        assert nb == 1, (k, loads, nb)
    # the K-order kernels read their chunk's gains in one batch (+ the histogram read)
    for k in ('k_pvq_order_count(', 'k_pvq_order_scatter('):
        loads, nb = batches(k)
        assert nb <= 2, (k, loads, nb)
