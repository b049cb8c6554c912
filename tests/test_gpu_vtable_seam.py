"""Drop-in seam 1 exercised end to end on the GPU: the REAL reference encoder
(oracle/_ref/enc_probe.so, compiled from /root/reference in the dev container;
the .so travels with gpurun) with its od_state_opt_vtbl fdct_2d/idct_2d entries
replaced by the HIP kernels must emit byte-identical packets and an identical
reconstruction.  This is the OD_CHECKASM idea (src/x86/x86dct.h:305-315) applied
to a whole frame: every DCT of the RDO pass, the real pass and od_compute_dist
runs on the device."""
import ctypes
import os

import numpy as np
import pytest

from testlib import ORACLE_DIR, pu8, synth_plane

pytestmark = pytest.mark.gpu
PROBE = os.path.join(ORACLE_DIR, '_ref', 'enc_probe.so')


@pytest.mark.skipif(not os.path.exists(PROBE), reason='oracle/_ref not built')
@pytest.mark.parametrize('quant,masking', ((20, 1), (5, 0)))
def test_reference_encoder_with_hip_vtable_is_bit_identical(quant, masking):
    import daala_amd.binding as b
    hip = b.load()
    assert hip.od_hip_device_count() > 0
    ep = ctypes.CDLL(PROBE)
    ep.probe_encode_frames_vtbl.restype = ctypes.c_long
    w, h = 176, 112                      # pads to 192 x 128: split-filter edge gating is hit
    frame = np.concatenate([synth_plane(w, h, 5).ravel(), synth_plane(w//2, h//2, 5, 1).ravel(),
                            synth_plane(w//2, h//2, 6, 1).ravel()])
    f = (ctypes.c_void_p*5)()
    i = (ctypes.c_void_p*5)()
    hip.od_hip_vtbl_fill.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    assert hip.od_hip_vtbl_fill(f, i) == 0

    def run(fd, idc):
        fnv = ctypes.c_uint()
        sec = ctypes.c_double()
        pk = np.zeros(1 << 20, np.uint8)
        rec = np.zeros(w*h*3//2, np.uint8)
        n = ep.probe_encode_frames_vtbl(w, h, 1, quant, 7, masking, 1, pu8(frame),
                                        ctypes.byref(fnv), ctypes.byref(sec), pu8(pk),
                                        ctypes.c_long(pk.size), fd, idc, pu8(rec))
        assert n > 0
        return n, fnv.value, pk[:n + 4].copy(), rec

    n_c, fnv_c, pk_c, rec_c = run(None, None)
    n_h, fnv_h, pk_h, rec_h = run(f, i)
    assert (n_c, fnv_c) == (n_h, fnv_h)
    assert np.array_equal(pk_c, pk_h)
    assert np.array_equal(rec_c, rec_h)
