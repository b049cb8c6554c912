#!/usr/bin/env python3
"""Benchmark of the MI355X Daala transform + PVQ hot path.

One "step" = one pass of the device hot path over one batch of synthetic input
already resident in HBM: BASELINE.json configs[1], 30 intra frames of 1920x1080
4:2:0 (padded to 1920x1088) per GPU:

  1. forward pyramid of all planes (A1 u8->coeff, A4 frame lapping, A5 split
     lapping, A6 fDCT of every block of every size)       [block-size RDO input]
  2. no-reference PVQ candidates (A13 gain, A18 K, A15 codeword search, distortion)
     for every band of every block of every pyramid level of every plane
  3. forward with known block sizes (od_compute_dcts) of all planes
  4. inverse (A7 iDCT, split + frame post-filters, A2 clamp) of all planes

The serial entropy coder / RDO argmin stay on the host and are NOT part of the
step (out of scope); `value` is therefore hot-path Mpixels/s,
not bitstream Mpixels/s.  Launch: python bench.py --gpus N --steps K --warmup W
(N > 1 under torch.distributed.run, one rank per GPU, independent frames per rank,
no data-path collective: "weak" scaling).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PIC_W, PIC_H, FW, FH = 1920, 1080, 1920, 1088
FRAMES = 30
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

BAND_OFF = {0: [1, 16], 1: [1, 16, 24, 32, 64], 2: [1, 16, 24, 32, 64, 96, 128, 256],
            3: [1, 16, 24, 32, 64, 96, 128, 256, 384, 512]}


def level_params(prm, tag, pli, bs, xdec):
    q0 = int(prm['quantizer_' + tag][pli])
    pq = prm['pvq_qm_q4_' + tag][pli]
    off = BAND_OFF[bs]
    nb = len(off) - 1
    q = [max(1, q0*int(pq[bs*(bs + 1) + (b + 1) - (b + 1)//3]) >> 4) for b in range(nb)]
    masking = tag.endswith('m1')
    beta = [1.5 if (masking and pli == 0 and bs > 0) else 1.0]*nb
    n = 4 << bs
    base = bs*2048 + xdec*1024
    return q, beta, np.ascontiguousarray(prm['qm_' + tag][base:base + n*n])


def make_frames(count, seed0):
    from testlib import synth_plane
    base = [synth_plane(FW, FH, seed0), synth_plane(FW//2, FH//2, seed0, 1),
            synth_plane(FW//2, FH//2, seed0 + 1, 1)]
    frames = []
    for f in range(count):
        frames.append([np.ascontiguousarray(np.roll(p, (3*f, 5*f), axis=(0, 1))) for p in base])
    return frames


def cpu_port_baseline(frames, bmaps, prm, tag, nframes):
    """Oracle (our C restatement) doing the same per-frame work as one device step,
    single thread.  Checker code used as a reported baseline only."""
    from testlib import oracle
    o = oracle()
    o.orc_bench_frame.restype = ctypes.c_long
    U8P = ctypes.POINTER(ctypes.c_uint8)
    qm = np.ascontiguousarray(prm['qm_' + tag])
    q0 = (ctypes.c_int*3)(*[int(v) for v in prm['quantizer_' + tag]])
    pq = np.ascontiguousarray(prm['pvq_qm_q4_' + tag]).ravel()
    t0 = time.perf_counter()
    for f in range(nframes):
        planes = (U8P*3)(*[p.ctypes.data_as(U8P) for p in frames[f]])
        o.orc_bench_frame(planes, FW, FH, PIC_W, PIC_H, bmaps[f].ctypes.data_as(U8P),
                          qm.ctypes.data_as(ctypes.POINTER(ctypes.c_int16)), q0,
                          pq.ctypes.data_as(U8P), 1 if tag.endswith('m1') else 0)
    dt = time.perf_counter() - t0
    return nframes*PIC_W*PIC_H/dt/1e6, dt


def cpu_reference_encoder(frames, nframes):
    """The real reference encoder (oracle/_ref, built from /root/reference in the
    dev container; the .so travels to the GPU box) on the same content, -v 20,
    complexity 7, masking on: whole bitstream encode, 1 thread."""
    so = os.path.join(ROOT, 'oracle', '_ref', 'enc_probe.so')
    if not os.path.exists(so):
        return None
    lib = ctypes.CDLL(so)
    lib.probe_encode_frames.restype = ctypes.c_long
    buf = np.concatenate([np.concatenate([frames[f][0][:PIC_H, :PIC_W].ravel(),
                                          frames[f][1][:PIC_H//2, :PIC_W//2].ravel(),
                                          frames[f][2][:PIC_H//2, :PIC_W//2].ravel()])
                          for f in range(nframes)])
    fnv = ctypes.c_uint()
    sec = ctypes.c_double()
    nbytes = lib.probe_encode_frames(PIC_W, PIC_H, nframes, 20, 7, 1, 1,
                                     buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                     ctypes.byref(fnv), ctypes.byref(sec), None, 0)
    if nbytes <= 0:
        return None
    return {'value': round(nframes*PIC_W*PIC_H/sec.value/1e6, 4), 'unit': 'Mpixels/s',
            'cores': 1, 'frames': nframes, 'packet_bytes': int(nbytes),
            'what': 'full reference encoder incl. entropy coding + RDO (oracle/_ref)'}


def host_cpu_budget():
    """CPUs this process may actually use: the cgroup quota when there is one
    (the GPU boxes expose 256 logical CPUs but cap a job at 16)."""
    n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(quota)//int(period)))
    except (OSError, ValueError):
        pass
    return n


def e2e_encode(frames, device, ref_mpix):
    """END-TO-END bit-exact intra encode through the batched frame seam
    (INTEGRATION.md seam 2, daala_amd/host/hip_enc_glue.c): the device feed answers
    every keyframe-luma no-reference PVQ search, N host workers run the reference
    encoder's serial entropy/RDO stage (daala_amd/host/build/libdaala_hipenc.so = the reference
    compiled in the dev container + our glue).  Reported beside `value`, not as it."""
    try:
        import daala_amd.hipenc as H
    except ImportError:
        return None
    if not H.have_hipenc():
        return None
    nw = min(16, host_cpu_budget(), len(frames))
    buf = H.pack_frames(frames, PIC_W, PIC_H)
    prm = H.Params(PIC_W, PIC_H, 20, 7, 1, nw, 0, 0)
    n, pk, st = H.encode(prm, buf, len(frames), use_device=1, device=device)
    if n < 0:
        return {'error': int(n)}
    # bit-exactness of the first two packets against the plain reference search
    p1 = H.Params(PIC_W, PIC_H, 20, 7, 1, 2, 0, 0)
    n0, pk0, st0 = H.encode(p1, buf, 2)
    mp = len(frames)*PIC_W*PIC_H/st.t_total_s/1e6
    out = {'Mpixels_per_s': round(mp, 3), 'frames': len(frames), 'host_workers': nw,
           'seconds': round(st.t_total_s, 3), 'packet_bytes': int(n),
           'bit_exact_vs_reference_packets': bool(pk[:2] == pk0),
           'searches_from_device': int(st.dev_hits),
           'searches_on_host': int(st.cpu_other + st.cpu_noref_luma),
           'g2_mismatch_host_recomputed': int(st.g2_mismatch), 'lost_sync': int(st.lost_sync),
           'host_search_seconds_all_workers': round(st.search_cpu_s, 3),
           'what': 'daala_encode_img_in/packet_out of 30 keyframes, device feed + host workers; '
                   'time from first frame in to last packet out (context creation excluded)'}
    if ref_mpix:
        out['x_single_thread_reference'] = round(mp/ref_mpix, 2)
    # decoder side of the seam on the packets just produced: reference parse on the host
    # workers + od_hip_decode_tail per frame, against the plain reference decoder
    hdr = H.headers(prm)
    nd, pics, sec, dsec = H.decode(prm, hdr, pk, use_device=1, device=device)
    p1.nworkers = 1
    n1, want, sec1, _ = H.decode(p1, hdr, pk[:4])
    if nd == len(pk) and n1 == 4:
        out['decode'] = {'Mpixels_per_s': round(len(pk)*PIC_W*PIC_H/sec/1e6, 2),
                         'host_workers': nw, 'seconds': round(sec, 3),
                         'device_call_seconds_all_workers': round(dsec, 3),
                         'pictures_identical_to_reference_decoder': bool(np.array_equal(pics[:4], want)),
                         'reference_decoder_1thread_Mpixels_per_s': round(4*PIC_W*PIC_H/sec1/1e6, 2),
                         'what': 'daala_decode_packet_in of the 30 packets; symbol parse on host '
                                 'workers, pixel-domain stage (iDCT, post-filters, deringing, '
                                 'smoothing, clamp) = one od_hip_decode_tail per frame'}
    return out


def e2e_encode_ranks(frames, device, world, dist, torch, red_dev='cuda'):
    """N > 1: every rank encodes its own 30 frames through the live seam on its own GPU
    with its share of the host CPUs; aggregate = all frames / slowest rank.  (No
    collective on the data path: frames are independent.)"""
    try:
        import daala_amd.hipenc as H
        ok = H.have_hipenc()
    except ImportError:
        ok = False
    sec = -1.0
    nw = max(1, min(16, host_cpu_budget()//world))
    if ok:
        buf = H.pack_frames(frames, PIC_W, PIC_H)
        prm = H.Params(PIC_W, PIC_H, 20, 7, 1, min(nw, len(frames)), 0, 0)
        dist.barrier()
        n, pk, st = H.encode(prm, buf, len(frames), use_device=1, device=device)
        if n >= 0:
            sec = st.t_total_s
    t = torch.tensor([sec, -sec], dtype=torch.float64, device=red_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    slowest, fastest = float(t[0].item()), -float(t[1].item())
    if fastest < 0:
        return None                      # some rank could not run it
    return {'Mpixels_per_s': round(world*len(frames)*PIC_W*PIC_H/slowest/1e6, 3),
            'frames_per_gpu': len(frames), 'host_workers_per_gpu': nw,
            'seconds_slowest_rank': round(slowest, 3),
            'what': 'every rank: 30 keyframes through the live seam on its own GPU; aggregate'}


KERNEL_SYMBOL = {     # bench kernel label -> substring of the device kernel name
    'k_forward_pyramid_luma': 'k_forward_rt<32, 4, false>',
    'k_forward_pyramid_chroma': 'k_forward_rt<16, 3, false>',
    'k_forward_known_luma': 'k_forward_rt<32, 4, true>',
    'k_forward_known_chroma': 'k_forward_rt<16, 3, true>',
    'k_inverse_sb_luma': 'k_inverse_rt<32, 4>',
    'k_inverse_sb_chroma': 'k_inverse_rt<16, 3>',
    'k_postfilter_clamp_luma': 'k_postfilter_clamp<32>',
    'k_postfilter_clamp_chroma': 'k_postfilter_clamp<16>',
}


def measured_traffic(label):
    """HBM bytes per launch of a kernel from the newest committed PMC profile
    (profiles/*_traffic.json: FETCH_SIZE and WRITE_SIZE from separate rocprofv3
    --pmc passes of this same bench, calibrated with known-size streams;
    tools/profile_round.sh + tools/summarize_profile.py).  None if absent: the
    counters cannot be read from inside an un-profiled run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_traffic.json')))
    if not files:
        return None
    with open(files[-1]) as f:
        t = json.load(f)
    for name, v in t['kernels'].items():
        if KERNEL_SYMBOL.get(label, '\0') in name:
            return {'hbm_bytes_per_launch': int(v['hbm_bytes']), 'fetch_bytes': int(v['fetch_bytes']),
                    'write_bytes': int(v['write_bytes']), 'source': os.path.basename(files[-1])}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--skip-pvq', action='store_true', help='profiling aid: transforms only (INVALID as a bench result)')
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # Rehearsal of the N > 1 code path on a 1-GPU box (tools/rehearse_multirank.sh):
    # BENCH_REHEARSE=1 puts every rank on device 0 and uses gloo for the barrier and the
    # max-time reduce (RCCL refuses two ranks on one GPU).  Never set by the driver.
    rehearse = os.environ.get('BENCH_REHEARSE') == '1'
    if rehearse:
        local_rank = 0
    red_dev = 'cpu' if rehearse else 'cuda'
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearse:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))

    import daala_amd.binding as b
    from testlib import random_bsize_map
    lib = b.load()                     # no fallback: raises if the HIP library is missing
    if lib.od_hip_device_count() <= local_rank:
        raise SystemExit('HIP device %d not available' % local_rank)

    prm = np.load(os.path.join(ROOT, 'tests', 'golden', 'encoder_params.npz'))
    tag = 'q20_m1'
    frames = make_frames(FRAMES, seed0=1 + rank)
    bmaps = [random_bsize_map(FW//32, FH//32, 1000*rank + f) for f in range(FRAMES)]
    ctx = b.DaalaHip(PIC_W, PIC_H, FW, FH, nplanes=3, xdec=(0, 1, 1), nslots=FRAMES,
                     device=local_rank)
    for f in range(FRAMES):
        ctx.upload_planes(f, frames[f])
        ctx.set_bsize(f, bmaps[f])
    lvl = []
    for pli in range(3):
        for level in range(ctx.nlevels(pli)):
            n = (32 >> ctx.xdec[pli]) >> level
            bs = {4: 0, 8: 1, 16: 2, 32: 3}[n]
            lvl.append((pli, level) + level_params(prm, tag, pli, bs, ctx.xdec[pli]))

    def step():
        ctx.forward_pyramid(0, FRAMES)
        if not args.skip_pvq:
            for pli, level, q, beta, qm in lvl:
                ctx.pvq_noref_search(pli, level, qm, q, beta, 0, FRAMES)
        ctx.forward_known(0, FRAMES, keyframe=1)
        ctx.inverse(0, FRAMES)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    ctx.sync()
    ctx.timing_reset()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel device time (HIP events on the context's own stream)
    s_y, s_c = FW*FH, (FW//2)*(FH//2)
    alg_bytes = {        # ALGORITHMIC bytes per launch (SURVEY.md 8d x units per launch)
        'k_forward_pyramid_luma': FRAMES*s_y*17,
        'k_forward_pyramid_chroma': FRAMES*s_c*13,
        'k_forward_known_luma': FRAMES*s_y*5,
        'k_forward_known_chroma': FRAMES*s_c*5,
        'k_inverse_sb_luma': FRAMES*s_y*5,          # inverse+postfilter+clamp is 5 B/sample
        'k_inverse_sb_chroma': FRAMES*s_c*5,        #   in total; split over two kernels here
        'k_postfilter_clamp_luma': FRAMES*s_y*5,
        'k_postfilter_clamp_chroma': FRAMES*s_c*5,
    }
    kernels = {}
    pvq_names = ['k_pvq_noref<15>', 'k_pvq_noref<8>', 'k_pvq_noref<32>', 'k_pvq_noref<128>']
    pvq_phase = ctx.timing_get('pvq_phase')
    for name in list(alg_bytes) + pvq_names:
        n, ms = ctx.timing_get(name)
        if n:
            kernels[name] = {'launches': n, 'avg_ms': ms/n}
            if name in alg_bytes:
                kernels[name]['GBps'] = alg_bytes[name]/(ms/n*1e-3)/1e9
    # Extra (not part of `value`): the decoder's pixel-domain tail on the same 30 frames
    # (iDCT + post-filters + deringing on every superblock + smoothing + clamp).
    decode_extra = None
    if world == 1:
        q0 = [int(v) for v in prm['quantizer_' + tag]]
        thr = [int(1.0*pow(q, 0.84182)) for q in q0]
        flags = np.ones((FH//32, FW//32), np.uint8)
        bsk = [np.zeros((FH//4, FW//4), np.uint8) for _ in range(3)]
        for f in range(FRAMES):
            ctx.set_decode_info(f, flags, bsk)
        ctx.decode_tail(thr, q0, 1, 0, FRAMES)
        ctx.sync()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            ctx.decode_tail(thr, q0, 1, 0, FRAMES)
        ctx.sync()
        dt = (time.perf_counter() - t1)/reps
        decode_extra = {'Mpixels_per_s': round(FRAMES*PIC_W*PIC_H/dt/1e6, 1),
                        'ms_per_30_frames': round(dt*1e3, 3),
                        'what': 'od_hip_decode_tail: coefficients -> 8-bit picture, deringing '
                                'forced on for every superblock (worst case)'}
    # Extra: lossless configuration (BASELINE configs[4]) - whole-superblock Haar planes,
    # forward (u8 -> int32) and inverse (int32 -> u8), 5 B/sample algorithmic each.
    lossless_extra = None
    if world == 1:
        ctx.timing_reset()
        for _ in range(3):
            ctx.forward_haar(0, FRAMES)
            ctx.inverse_haar(0, FRAMES)
        ctx.sync()
        hk = {}
        for nm, smp in (('k_haar_forward_luma', FW*FH), ('k_haar_forward_chroma', (FW//2)*(FH//2)),
                        ('k_haar_inverse_luma', FW*FH), ('k_haar_inverse_chroma', (FW//2)*(FH//2))):
            n, ms = ctx.timing_get(nm)
            if n:
                hk[nm] = {'avg_ms': round(ms/n, 4), 'GBps': round(FRAMES*smp*5/(ms/n*1e-3)/1e9, 1)}
        tot = sum(v['avg_ms'] for k, v in hk.items() if 'luma' in k) + \
            2*sum(v['avg_ms'] for k, v in hk.items() if 'chroma' in k)
        lossless_extra = {'kernels': hk, 'Mpixels_per_s_fwd_plus_inv': round(FRAMES*PIC_W*PIC_H/(tot*1e-3)/1e6, 1),
                          'what': 'od_hip_forward_haar + od_hip_inverse_haar of 30 frames (round trip exact)'}
    line = None
    if rank == 0:
        px = world*FRAMES*PIC_W*PIC_H*args.steps
        value = px/elapsed/1e6
        # dominant HBM-bound (transform) kernel by device time
        hb = {k: v for k, v in kernels.items() if k in alg_bytes}
        dom = max(hb, key=lambda k: hb[k]['avg_ms']*hb[k]['launches'])
        ach = hb[dom]['GBps']
        traffic = measured_traffic(dom)
        line = {
            'metric': 'encode Mpixels/s (intra hot path: lapping+DCT pyramid, PVQ no-ref '
                      'search, known-size forward, inverse; bit-exact vs oracle)',
            'value': round(value, 3), 'unit': 'Mpixels/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed/args.steps*1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int32+f64', 'data': 'synthetic',
            'config': {'workload': '1920x1080 4:2:0 synthetic, 30 intra frames per GPU '
                                   '(BASELINE configs[1]), q=20, activity masking on, '
                                   'random valid block-size maps',
                       'frames_per_gpu': FRAMES, 'parallelism': 'independent frames per GPU'},
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': round(ach, 2),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach/HBM_PEAK_GBS, 5),
                         'algorithmic_bytes_per_launch': alg_bytes[dom],
                         # HBM bytes per launch from the committed PMC passes (FETCH_SIZE +
                         # WRITE_SIZE, calibrated; profiles/*_traffic.json), or null
                         'traffic': traffic['hbm_bytes_per_launch'] if traffic else None,
                         'traffic_detail': traffic},
            'kernels': {k: {kk: round(vv, 4) if isinstance(vv, float) else vv
                            for kk, vv in v.items()} for k, v in kernels.items()},
        }
        if any(nm in kernels for nm in pvq_names):
            # band vectors searched per second (each band = up to 2 gain candidates)
            nb = {0: 1, 1: 4, 2: 7, 3: 9}
            bands = 0
            for pli, level, q, beta, qm in lvl:
                n = (32 >> ctx.xdec[pli]) >> level
                bands += ctx.pvq_nblocks(pli, level)*nb[{4: 0, 8: 1, 16: 2, 32: 3}[n]]
            tot_ms = sum(kernels[nm]['avg_ms']*kernels[nm]['launches'] for nm in pvq_names
                         if nm in kernels)
            # The PVQ kernels of a step run concurrently on side streams (their tails
            # overlap), so their individual spans overlap too: the batch's wall time is
            # the library's `pvq_phase` span (first PVQ launch -> join).
            nph, ph_ms = pvq_phase
            concurrent = nph > 0
            if concurrent:
                tot_ms = ph_ms
            other_ms = sum(v['avg_ms']*v['launches'] for k, v in kernels.items() if k not in pvq_names)
            line['pvq'] = {'bands_per_s': round(bands*FRAMES*args.steps/(tot_ms*1e-3), 1),
                           'bands_per_frame': bands,
                           'ms_per_step': round(tot_ms/args.steps, 3),
                           'share_of_device_time': round(tot_ms/(tot_ms + other_ms), 4),
                           'concurrent_side_streams': bool(concurrent)}
        if decode_extra:
            line['decode_tail'] = decode_extra
        if lossless_extra:
            line['lossless_haar'] = lossless_extra
        if world == 1 and not args.no_cpu_baseline:
            nf = 20                  # ~11 s of single-core work (the contract asks for 10-30 s)
            v, dt = cpu_port_baseline(frames, bmaps, prm, tag, nf)
            line['cpu_baseline'] = {'value': round(v, 4), 'unit': 'Mpixels/s', 'cores': 1,
                                    'kind': 'port',
                                    'sample': '%d of the 30 1080p frames, same per-frame work as '
                                              'the device step, oracle C code, %.1f s' % (nf, dt)}
            refenc = cpu_reference_encoder(frames, 3)
            if refenc:
                line['cpu_reference_encoder'] = refenc
            ctx.close()          # free the HBM slots before the end-to-end run
            e2e = e2e_encode(frames, local_rank, refenc['value'] if refenc else None)
            if e2e:
                line['e2e_encode'] = e2e
        if args.skip_pvq:
            line['INVALID'] = 'profiling run with --skip-pvq'
    ctx.close()
    if world > 1 and not args.no_cpu_baseline:
        # every rank takes part (after rank 0 has read everything it needs from its ctx)
        e2e_multi = e2e_encode_ranks(frames, local_rank, world, dist, torch, red_dev)
        if rank == 0 and e2e_multi:
            line['e2e_encode'] = e2e_multi
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
