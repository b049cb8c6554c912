"""Y4M reader/writer and the CLI round trip (host-only drivers on CPU; the device
path of the same drivers is covered by tests/test_gpu_hipenc.py).  BASELINE configs[0]:
a single 352x288 4:2:0 Y4M frame through encode -> decode."""
import ctypes
import os

import numpy as np
import pytest

from testlib import synth_plane, ref, pu8
import hipenc_lib as H
import importlib.util
_spec = importlib.util.spec_from_file_location('daala_hip_cli', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'daala_hip_cli.py'))
cli = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(cli)
from daala_amd.y4m import Y4MReader, Y4MWriter, Y4MError

needs = pytest.mark.skipif(not H.have_hipenc(), reason='daala_amd/host/build/libdaala_hipenc.so not built')


def write_clip(path, w, h, nf, chroma='C420jpeg'):
    with open(path, 'wb') as f:
        f.write(('YUV4MPEG2 W%d H%d F25:1 Ip A1:1 %s XYSCSS=420JPEG\n' % (w, h, chroma)).encode())
        frames = []
        for s in range(nf):
            fr = np.concatenate([synth_plane(w, h, s).ravel(), synth_plane((w + 1)//2, (h + 1)//2, s, 1).ravel(),
                                 synth_plane((w + 1)//2, (h + 1)//2, s + 1, 1).ravel()])
            f.write(b'FRAME\n')
            f.write(fr.tobytes())
            frames.append(fr)
    return frames


def test_y4m_roundtrip_and_refusals(tmp_path):
    p = str(tmp_path/'a.y4m')
    frames = write_clip(p, 46, 34, 3, 'C420mpeg2')       # odd chroma size
    rd = Y4MReader(p)
    assert (rd.width, rd.height, rd.fps, rd.chroma) == (46, 34, (25, 1), '420mpeg2')
    got = list(rd.frames())
    assert len(got) == 3 and all(np.array_equal(a, b) for a, b in zip(got, frames))
    q = str(tmp_path/'b.y4m')
    wr = Y4MWriter(q, 46, 34, rd.fps)
    for fr in got:
        wr.write(fr)
    wr.close()
    assert all(np.array_equal(a, b) for a, b in zip(Y4MReader(q).frames(), frames))
    bad = str(tmp_path/'c.y4m')
    open(bad, 'wb').write(b'YUV4MPEG2 W16 H16 F30:1 Ip C444\n')
    with pytest.raises(Y4MError):
        Y4MReader(bad)
    open(bad, 'wb').write(b'YUV4MPEG2 W16 H16 F30:1 It C420jpeg\n')
    with pytest.raises(Y4MError):
        Y4MReader(bad)
    trunc = str(tmp_path/'d.y4m')
    open(trunc, 'wb').write(open(p, 'rb').read()[:-5])
    with pytest.raises(Y4MError):
        list(Y4MReader(trunc).frames())


@needs
def test_cli_cif_frame_packets_are_the_reference_encoders(tmp_path):
    """configs[0]: one CIF frame; the CLI's stream holds exactly the reference
    encoder's packet, and decoding it gives the encoder's reconstruction."""
    w, h = 352, 288
    y4m, out, back = (str(tmp_path/n) for n in ('in.y4m', 'out.dhip', 'back.y4m'))
    frames = write_clip(y4m, w, h, 1)
    cli.main(['encode', y4m, out, '-v', '20', '--workers', '1', '--no-device'])
    cli.main(['decode', out, back, '--workers', '1', '--no-device'])
    _, _, _, _, _, nf, hdr, pk = cli.read_container(out)
    lib = ref('enc_probe')
    lib.probe_encode_frames_vtbl.restype = ctypes.c_long
    pkt = np.zeros(1 << 20, np.uint8)
    rec = np.zeros(w*h*3//2, np.uint8)
    fnv, sec = ctypes.c_uint(), ctypes.c_double()
    buf = np.ascontiguousarray(frames[0])
    n = lib.probe_encode_frames_vtbl(w, h, 1, 20, 7, 1, 1, pu8(buf), ctypes.byref(fnv), ctypes.byref(sec),
                                     pu8(pkt), pkt.size, None, None, pu8(rec))
    assert nf == 1 and pk == H.split_packets(pkt, 1) and len(pk[0]) == n
    dec = list(Y4MReader(back).frames())
    assert len(dec) == 1 and np.array_equal(dec[0], rec)


@needs
def test_cli_without_device_fails_loudly(tmp_path):
    import daala_amd.binding as b
    if b.load().od_hip_device_count() > 0:
        pytest.skip('a HIP device is present')
    y4m = str(tmp_path/'in.y4m')
    write_clip(y4m, 64, 64, 1)
    with pytest.raises(SystemExit) as e:
        cli.main(['encode', y4m, str(tmp_path/'o.dhip'), '--workers', '1'])
    assert 'no HIP device' in str(e.value)


@pytest.mark.gpu
@needs
def test_cli_device_path_equals_host_path(tmp_path):
    """Y4M -> stream -> Y4M through the CLI with the device (feed + decoder tail) and with
    --no-device: identical stream bytes, identical decoded pictures."""
    w, h = 320, 176
    y4m = str(tmp_path/'in.y4m')
    write_clip(y4m, w, h, 4)
    a, b_, pa, pb = (str(tmp_path/n) for n in ('a.dhip', 'b.dhip', 'a.y4m', 'b.y4m'))
    cli.main(['encode', y4m, a, '-v', '20', '--workers', '2'])
    cli.main(['encode', y4m, b_, '-v', '20', '--workers', '2', '--no-device'])
    assert open(a, 'rb').read() == open(b_, 'rb').read()
    cli.main(['decode', a, pa, '--workers', '2'])
    cli.main(['decode', a, pb, '--workers', '2', '--no-device'])
    assert open(pa, 'rb').read() == open(pb, 'rb').read()
