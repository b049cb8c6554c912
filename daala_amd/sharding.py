"""Multi-GPU sharding of the hot path.  One process per GPU.

Two granularities, both WITHOUT a collective on the compute path:

* independent intra frames  -> `frame_partition`: frame f goes to rank f % world
  (adaptation is reset per frame, reference src/encode.c:3083, so keyframes are
  independent; what bench.py --gpus N measures).
* superblock rows of one frame (BASELINE config 3) -> `SbRowShard`: every rank
  transforms a contiguous strip of superblock rows.  The transform path is
  SB-local except for the frame lapping, which reaches 2 samples across an SB
  boundary (src/filter.c:1566-1584), so a strip is computed together with ONE
  halo superblock row above and below (recomputed redundantly, discarded), from
  the replicated input frame.  The only exchange is the final gather of the
  strips (`gather_rows`: torch.distributed all_gather = RCCL over xGMI on GPUs,
  gloo on CPU for the tests).

The compute itself is passed in as a callable so that the same host logic is
exercised by the CPU tests (oracle as compute, gloo) and by the GPU path
(`hip_strip_pyramid`)."""
import numpy as np


def frame_partition(nframes, world, rank):
    """Frames handled by `rank` (round robin keeps every rank busy for any count)."""
    return list(range(rank, nframes, world))


def sb_row_partition(nvsb, world, rank):
    """Contiguous strip [r0, r1) of superblock rows owned by `rank`."""
    base, rem = divmod(nvsb, world)
    r0 = rank*base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


class SbRowShard(object):
    """Geometry of one rank's strip, including the halo superblock rows."""

    def __init__(self, pic_w, pic_h, fw, fh, world, rank):
        self.pic_w, self.pic_h, self.fw, self.fh = pic_w, pic_h, fw, fh
        self.nvsb = fh//32
        self.r0, self.r1 = sb_row_partition(self.nvsb, world, rank)
        self.h0 = max(self.r0 - 1, 0)                   # first SB row incl. halo
        self.h1 = min(self.r1 + 1, self.nvsb)           # one past the last SB row incl. halo
        self.empty = self.r1 <= self.r0

    def strip_geometry(self):
        """(pic_w, pic_h, fw, fh) of the strip: the picture height is shifted so
        that the edge gating `(by + 1)*n <= pic_height` (src/encode.c:1319) sees
        the same truth values as in the full frame."""
        fh = (self.h1 - self.h0)*32
        ph = min(max(self.pic_h - self.h0*32, 0), fh)
        return self.pic_w, ph, self.fw, fh

    def input_rows(self, dec):
        sb = 32 >> dec
        return self.h0*sb, self.h1*sb

    def own_rows_in_strip(self, dec):
        sb = 32 >> dec
        return (self.r0 - self.h0)*sb, (self.r1 - self.h0)*sb

    def own_rows_in_frame(self, dec):
        sb = 32 >> dec
        return self.r0*sb, self.r1*sb


def strip_pyramid(shard, planes, xdec, compute):
    """Runs `compute(strip_planes, (pic_w, pic_h, fw, fh)) -> [plane][level] arrays`
    on this rank's strip (with halo) and returns only the rows the rank owns:
    result[pli][level] has shape (own_rows, plane_width)."""
    if shard.empty:
        return [[np.zeros((0, shard.fw >> d), np.int32) for _ in range(4 - d)] for d in xdec]
    strips = []
    for p, d in zip(planes, xdec):
        a, b = shard.input_rows(d)
        strips.append(np.ascontiguousarray(p[a:b]))
    lev = compute(strips, shard.strip_geometry())
    out = []
    for pli, d in enumerate(xdec):
        a, b = shard.own_rows_in_strip(d)
        out.append([np.ascontiguousarray(l[a:b]) for l in lev[pli]])
    return out


def gather_rows(local, total_rows, dist=None, device='cpu'):
    """All-gather row strips (each rank contributes `local`, shape (rows_r, width),
    ranks ordered top to bottom) into the full (total_rows, width) array on every
    rank.  Strips may have different heights: they are padded to the tallest one
    for the collective and trimmed afterwards.  Without a process group
    (dist is None) the input must already be the whole plane."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert local.shape[0] == total_rows
        return local
    import torch
    world = dist.get_world_size()
    width = local.shape[1]
    rows = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    all_rows = [torch.zeros_like(rows) for _ in range(world)]
    dist.all_gather(all_rows, rows)
    all_rows = [int(r.item()) for r in all_rows]
    assert sum(all_rows) == total_rows, (all_rows, total_rows)
    mx = max(all_rows)
    buf = torch.zeros((mx, width), dtype=torch.int32, device=device)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(local).to(device)
    parts = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return np.concatenate([p[:r].cpu().numpy() for p, r in zip(parts, all_rows)], axis=0)


def hip_strip_compute(device=0):
    """compute() for strip_pyramid backed by the HIP library (one context per call
    geometry; GPU only - raises if the library or a device is missing)."""
    from . import binding

    def compute(strips, geom):
        pic_w, pic_h, fw, fh = geom
        xdec = tuple(0 if s.shape[1] == fw else 1 for s in strips)
        ctx = binding.DaalaHip(pic_w, pic_h, fw, fh, nplanes=len(strips), xdec=xdec, nslots=1,
                               device=device)
        ctx.upload_planes(0, strips)
        ctx.forward_pyramid()
        out = [[ctx.download_level(0, pli, k) for k in range(ctx.nlevels(pli))]
               for pli in range(len(strips))]
        ctx.close()
        return out

    return compute
