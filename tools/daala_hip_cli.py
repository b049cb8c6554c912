"""Command line front end of the MI355X Daala path:

  python tools/daala_hip_cli.py encode in.y4m out.dhip [-v 20] [-k 1] [--workers N] [--no-device]
  python tools/daala_hip_cli.py decode out.dhip out.y4m [--workers N] [--no-device]

(The drivers live in the reference-side integration library, which is built from the
reference sources in the dev container - see INTEGRATION.md "Seam 2, live" - so this
front end sits beside it in tools/, not in the product package.)

encode: with -k 1 (default) every frame is a keyframe coded by the reference encoder's serial
stage on N host workers with the device feed answering the state-free PVQ searches
(daala_amd/host/hip_enc_glue.c); -k K > 1 codes an inter stream (keyframe every K frames)
in order on one worker; -v 0 codes lossless frames (Haar planes from the device).  The
packets are byte-identical to the reference encoder's.  decode: reference symbol parse on the host workers, pixel-domain stage on
the device.  The reference's examples write Ogg; libogg is not part of this path, so
the container is minimal: b"DHIP1\\n", then little-endian u32 fields (width, height,
quant, masking, fps_n, fps_d, nframes, header bytes), the header-packet blob and the
length-prefixed video packets.  --no-device runs the same drivers with the plain
reference code (for comparison); without it a missing GPU is an error."""
import argparse
import struct
import sys
import time

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import daala_amd.hipenc as H                              # noqa: E402
from daala_amd.y4m import Y4MReader, Y4MWriter          # noqa: E402

MAGIC = b'DHIP1\n'


def cmd_encode(a):
    rd = Y4MReader(a.input)
    frames = list(rd.frames(a.limit))
    if not frames:
        raise SystemExit('no frames in %s' % a.input)
    buf = np.ascontiguousarray(np.concatenate(frames))
    prm = H.Params(rd.width, rd.height, a.quant, 7, 0 if a.no_masking else 1, a.workers, 0, a.batch,
                   a.keyframe_rate)
    t0 = time.perf_counter()
    n, pk, st = H.encode(prm, buf, len(frames), use_device=0 if a.no_device else 1, device=a.device)
    if n < 0:
        raise SystemExit('encode failed (%d)%s' % (n, ': no HIP device' if n == -30 else ''))
    hdr = H.headers(prm)
    with open(a.output, 'wb') as f:
        f.write(MAGIC)
        f.write(struct.pack('<8I', rd.width, rd.height, a.quant, 0 if a.no_masking else 1,
                            rd.fps[0], rd.fps[1], len(frames), hdr.size))
        f.write(hdr.tobytes())
        f.write(H.join_packets(pk).tobytes())
    px = rd.width*rd.height*len(frames)
    sys.stderr.write('%d frames %dx%d -> %d bytes, %.2f Mpixels/s (%.2fs; searches: %d device, %d host)\n'
                     % (len(frames), rd.width, rd.height, n, px/st.t_total_s/1e6,
                        time.perf_counter() - t0, st.dev_hits, st.cpu_other + st.cpu_noref_luma))


def read_container(path):
    with open(path, 'rb') as f:
        if f.read(len(MAGIC)) != MAGIC:
            raise SystemExit('%s: not a DHIP1 stream' % path)
        w, h, quant, masking, fn, fd, nf, hb = struct.unpack('<8I', f.read(32))
        hdr = np.frombuffer(f.read(hb), np.uint8).copy()
        rest = np.frombuffer(f.read(), np.uint8).copy()
    return w, h, quant, masking, (fn, fd), nf, hdr, H.split_packets(rest, nf)


def cmd_decode(a):
    w, h, quant, masking, fps, nf, hdr, pk = read_container(a.input)
    prm = H.Params(w, h, quant, 7, masking, a.workers, 0, 0)
    n, pics, sec, dsec = H.decode(prm, hdr, pk, use_device=0 if a.no_device else 1, device=a.device)
    if n < 0:
        raise SystemExit('decode failed (%d)%s' % (n, ': no HIP device' if n == -30 else ''))
    wr = Y4MWriter(a.output, w, h, fps)
    for f in range(nf):
        wr.write(pics[f])
    wr.close()
    sys.stderr.write('%d frames %dx%d decoded, %.2f Mpixels/s\n' % (nf, w, h, w*h*nf/sec/1e6))


def main(argv=None):
    ap = argparse.ArgumentParser(prog='daala_hip_cli', description=__doc__.split('\n')[0])
    sub = ap.add_subparsers(dest='cmd', required=True)
    e = sub.add_parser('encode')
    e.add_argument('input')
    e.add_argument('output')
    e.add_argument('-v', '--quant', type=int, default=20, help='OD_SET_QUANT (0 = lossless)')
    e.add_argument('-k', '--keyframe-rate', type=int, default=1, help='OD_SET_KEYFRAME_RATE (1 = intra only)')
    e.add_argument('--no-masking', action='store_true')
    e.add_argument('--limit', type=int, default=None)
    e.add_argument('--batch', type=int, default=0, help='frames resident on the device at once (0: all)')
    d = sub.add_parser('decode')
    d.add_argument('input')
    d.add_argument('output')
    for p in (e, d):
        p.add_argument('--workers', type=int, default=16)
        p.add_argument('--device', type=int, default=0)
        p.add_argument('--no-device', action='store_true')
    a = ap.parse_args(argv)
    if a.cmd == 'encode':
        if a.quant < 0 or a.keyframe_rate < 1:
            raise SystemExit('quant must be >= 0 and the keyframe rate >= 1')
        cmd_encode(a)
    else:
        cmd_decode(a)


if __name__ == '__main__':
    main()
