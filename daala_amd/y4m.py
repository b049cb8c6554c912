"""YUV4MPEG2 reader / writer for the formats the device path codes: 8-bit planar
4:2:0, progressive.  Mirrors what the reference's tools accept for those streams
(header tags: examples/encoder_example.c:173-394 `id_y4m_file` / y4m_parse_tags,
frame read :448-488; writer header line: examples/dump_video.c:380) - the chroma
siting variants C420jpeg / C420mpeg2 / C420paldv / C420 are all read as-is, like the
reference does.  Anything else (4:4:4, 4:2:2, >8 bit, interlaced) is refused loudly."""
import numpy as np

_OK_CHROMA = ('420jpeg', '420mpeg2', '420paldv', '420')


class Y4MError(ValueError):
    pass


class Y4MReader:
    def __init__(self, path):
        self.f = open(path, 'rb')
        line = self.f.readline(256)
        if not line.startswith(b'YUV4MPEG2') or not line.endswith(b'\n'):
            raise Y4MError('%s: not a YUV4MPEG2 file' % path)
        self.width = self.height = 0
        self.fps = (30, 1)
        self.aspect = (1, 1)
        self.interlace = 'p'
        self.chroma = '420jpeg'
        for tag in line[9:].split():
            t, v = chr(tag[0]), tag[1:].decode('ascii')
            if t == 'W':
                self.width = int(v)
            elif t == 'H':
                self.height = int(v)
            elif t == 'F':
                n, d = v.split(':')
                self.fps = (int(n), int(d))
            elif t == 'A':
                n, d = v.split(':')
                self.aspect = (int(n), int(d))
            elif t == 'I':
                self.interlace = v
            elif t == 'C':
                self.chroma = v
        if self.width <= 0 or self.height <= 0:
            raise Y4MError('%s: missing W/H tags' % path)
        if self.interlace not in ('p', '?'):
            raise Y4MError('interlaced input is not supported (the reference refuses it too)')
        if self.chroma not in _OK_CHROMA:
            raise Y4MError('chroma format C%s is not supported by the device path (8-bit 4:2:0 only)'
                           % self.chroma)
        cw, ch = (self.width + 1)//2, (self.height + 1)//2
        self.frame_bytes = self.width*self.height + 2*cw*ch

    def frames(self, limit=None):
        """Yields dense 4:2:0 frames (Y then U then V) as uint8 arrays."""
        n = 0
        while limit is None or n < limit:
            line = self.f.readline(256)
            if not line:
                return
            if not line.startswith(b'FRAME'):
                raise Y4MError('bad frame header %r' % line[:16])
            buf = self.f.read(self.frame_bytes)
            if len(buf) != self.frame_bytes:
                raise Y4MError('truncated frame %d' % n)
            yield np.frombuffer(buf, np.uint8)
            n += 1

    def close(self):
        self.f.close()


class Y4MWriter:
    def __init__(self, path, width, height, fps=(30, 1), aspect=(1, 1)):
        self.f = open(path, 'wb')
        self.f.write(('YUV4MPEG2 W%d H%d F%d:%d Ip A%d:%d C420jpeg\n'
                      % (width, height, fps[0], fps[1], aspect[0], aspect[1])).encode('ascii'))

    def write(self, frame):
        self.f.write(b'FRAME\n')
        self.f.write(np.ascontiguousarray(frame, dtype=np.uint8).tobytes())

    def close(self):
        self.f.close()
