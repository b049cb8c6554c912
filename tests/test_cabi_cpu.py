"""CPU checks of the C ABI: the shared library loads, exports every symbol that
include/daala_hip.h declares, and - with no GPU in this container - refuses compute
calls loudly instead of falling back to any CPU path."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import daala_amd.binding as b
    if not os.path.exists(b.lib_path()):
        import subprocess
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'daala_amd', 'csrc')])
    return b.load()


def declared_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'daala_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = re.findall(r'^(?:int|long|void|const char \*|od_hip_\w+ \*)\s*(od_hip_\w+)\s*\(', hdr, re.M)
    assert len(names) > 85
    return sorted(set(names))


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert getattr(lib, name) is not None, name


def test_header_is_plain_c():
    import subprocess
    src = '#include "daala_hip.h"\nint main(void) { return OD_HIP_SUCCESS; }\n'
    r = subprocess.run(['gcc', '-std=c89', '-pedantic', '-Wall', '-Werror', '-fsyntax-only', '-x', 'c',
                        '-I', os.path.join(ROOT, 'include'), '-'], input=src.encode(),
                       capture_output=True)
    assert r.returncode == 0, r.stderr.decode()


def test_no_cpu_fallback_without_a_device(lib):
    import daala_amd.binding as b
    if lib.od_hip_device_count() > 0:
        pytest.skip('a GPU is visible: the refusal path cannot be exercised')
    x = np.zeros((1, 4, 4), np.int32)
    with pytest.raises(b.HipError) as e:
        b.od_bin_fdct_blocks(0, x)
    assert 'no HIP device' in str(e.value)
    with pytest.raises(b.HipError):
        b.DaalaHip(64, 64, nplanes=1, xdec=(0,), nslots=1)
    with pytest.raises(b.HipError):
        b.pvq_search_vectors(np.zeros((1, 8)), np.ones(1, np.int32), np.ones(1))
    # argument validation happens before any device work
    assert lib.od_hip_fdct_blocks(7, None, None, 1) < 0


def test_new_objects_refuse_without_a_device(lib):
    """The round-3 objects (resident motion compensation, P-frame feed) have no CPU path either."""
    import daala_amd.binding as b
    if lib.od_hip_device_count() > 0:
        pytest.skip('a device is present')
    lib.od_hip_mc_create.restype = ctypes.c_void_p
    assert not lib.od_hip_mc_create(0, 4)
    g = b.Geometry()
    g.pic_width, g.pic_height, g.frame_width, g.frame_height, g.nplanes, g.nslots = 64, 64, 64, 64, 3, 2
    g.xdec[1] = g.xdec[2] = 1
    lib.od_hip_pfeed_create.restype = ctypes.c_void_p
    assert not lib.od_hip_pfeed_create(0, ctypes.byref(g))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under daala_amd/ or include/ may
    reference it."""
    for base in ('daala_amd', 'include'):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith(('.py', '.h', '.hpp', '.hip', '.c', '.cpp', 'Makefile')):
                    txt = open(os.path.join(dirpath, f), errors='ignore').read()
                    assert 'oracle' not in txt.lower() or f == 'sharding.py', os.path.join(dirpath, f)


def test_export_map():
    """include/daala_hip.h's EXPORT MAP: the entries it lists as VECTOR (parity tests only) are
    bound by no seam - nothing under daala_amd/host references them - and every one of them is a
    real export."""
    import re
    hdr = open(os.path.join(ROOT, 'include', 'daala_hip.h')).read()
    m = re.search(r'\*   VECTOR - (.*?)\n \*/', hdr, re.S)
    assert m
    names = set(re.findall(r'od_hip_[a-z0-9_]+', m.group(1)))
    names.add('od_hip_resample_luma_420')
    assert len(names) >= 16
    decl = set(re.findall(r'\b(od_hip_[a-z0-9_]+)\(', hdr))
    assert names <= decl, names - decl
    host = ''
    hd = os.path.join(ROOT, 'daala_amd', 'host')
    for f in os.listdir(hd):
        if f.endswith(('.c', '.h')):
            host += open(os.path.join(hd, f), errors='ignore').read()
    used = set(re.findall(r'od_hip_[a-z0-9_]+', host))
    assert not (names & used), names & used
