"""Worker of tests/test_sharding_gloo.py: one rank of a gloo process group.
Compute = the CPU oracle (test infrastructure), so this exercises exactly the
host-side sharding/gather logic that the GPU path uses."""
import ctypes
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from testlib import oracle, p32, pu8, synth_plane  # noqa: E402
from daala_amd import sharding  # noqa: E402


def oracle_compute(strips, geom):
    pic_w, pic_h, fw, fh = geom
    o = oracle()
    out = []
    for p in strips:
        dec = 0 if p.shape[1] == fw else 1
        w, h = fw >> dec, fh >> dec
        nlev = 4 - dec
        lev = [np.zeros((h, w), np.int32) for _ in range(nlev)]
        arr = (ctypes.POINTER(ctypes.c_int32)*nlev)(*[p32(a) for a in lev])
        c = np.zeros((h, w), np.int32)
        o.orc_forward_pyramid_plane(p32(c), arr, nlev, pu8(np.ascontiguousarray(p)), w, fw//32,
                                    fh//32, dec, pic_w, pic_h)
        out.append(lev)
    return out


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % port, rank=rank,
                            world_size=world)
    pic_w, pic_h, fw, fh = 150, 200, 192, 224          # 7 SB rows: uneven strips, pic_h inside
    planes = [synth_plane(fw, fh, 31), synth_plane(fw//2, fh//2, 31, 1)]
    xdec = (0, 1)
    shard = sharding.SbRowShard(pic_w, pic_h, fw, fh, world, rank)
    local = sharding.strip_pyramid(shard, planes, xdec, oracle_compute)
    ok = True
    full = oracle_compute(planes, (pic_w, pic_h, fw, fh))
    for pli, d in enumerate(xdec):
        for k in range(4 - d):
            g = sharding.gather_rows(local[pli][k], fh >> d, dist)
            ok = ok and np.array_equal(g, full[pli][k])
    # frame partition covers every frame exactly once
    frames = sharding.frame_partition(11, world, rank)
    allf = [None]*world
    dist.all_gather_object(allf, frames)
    ok = ok and sorted(sum(allf, [])) == list(range(11))
    with open(os.path.join(outdir, 'rank%d.txt' % rank), 'w') as f:
        f.write('ok' if ok else 'FAIL')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
