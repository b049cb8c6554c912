#!/usr/bin/env python3
"""Host time classes of the live seams on the GPU box (HIPENC_TIME=1 timers inside the
integration library), written as markdown (profiles/rNN_host_time_classes.md):
  * the intra step of the bench (30 x 1080p, device feed): share of the workers' CPU time in the
    rate-only pricing of codewords, in the host searches that remain, and - the measured answer
    to VERDICT r02 #9 - in the part of the pricing that does not depend on the adaptation state;
  * an inter sample (1080p I P P): where a P frame's time goes before the prediction exists
    (input copy + od_mv_est), in the P-frame feed wait, in pricing.
Usage: python tools/host_time_classes.py > profiles/r03_host_time_classes.md"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['HIPENC_TIME'] = '1'
import bench                      # noqa: E402
import daala_amd.hipenc as H      # noqa: E402

W, Hh = 1920, 1080
frames = bench.make_frames(30, seed0=1)
buf = H.pack_frames(frames, W, Hh)
nw = 30
prm = H.Params(W, Hh, 20, 7, 1, nw, 0, 30)
with H.Session(prm, use_device=1) as ses:
    ses.encode(buf, 30)
    n, pk, st = ses.encode(buf, 30)
cpu = st.frame_cpu_s
print('# Host time classes, live seams, round 3 (HIPENC_TIME=1: the per-call classes time one call in 61 with the time stamp counter and scale)\n')
print('## Intra step: 30 x 1920x1080, -v 20, masking on, device feed, %d workers\n' % nw)
print('| class | seconds, all workers | share of the workers\' frame time |')
print('|---|---|---|')
print('| frame time (daala_encode_img_in + packet_out) | %.2f | 100 %% |' % cpu)
print('| rate-only pricing of codewords (%d calls) | %.2f | %.1f %% |' % (st.rate_calls, st.rate_s, 100*st.rate_s/cpu))
print('| - of which a state-free skeleton of the same work | %.2f | %.1f %% (= %.1f %% of the pricing) |'
      % (st.rate_state_free_s, 100*st.rate_state_free_s/cpu, 100*st.rate_state_free_s/max(st.rate_s, 1e-9)))
print('| host codeword searches (with-reference luma, chroma) | %.2f | %.1f %% |' % (st.search_cpu_s, 100*st.search_cpu_s/cpu))
cls = list(st.search_class_s)
print('| - luma no-ref / luma with-ref / chroma no-ref / chroma with-ref | %.2f / %.2f / %.2f / %.2f | |' % tuple(cls))
print('\nsearches from the device %d, on the host %d; wall %.3f s per step\n' % (st.dev_hits, st.cpu_other + st.cpu_noref_luma, st.t_total_s))
fr = frames[:3]
b3 = H.pack_frames(fr, W, Hh)
p3 = H.Params(W, Hh, 20, 7, 1, 1, 0, 0, 30)
for pfeed in ('1', '0'):
    os.environ['HIPENC_PFEED'] = pfeed
    with H.Session(p3, use_device=1) as ses:
        ses.encode(b3, 3)
        n, pk, s3 = ses.encode(b3, 3)
    with H.Session(p3, use_device=1) as ses:
        ses.encode(b3, 1)
        n1, pk1, s1 = ses.encode(b3, 1)
    pt = s3.t_total_s - s1.t_total_s
    print('## Inter sample: 1920x1080 I P P, one worker, P-frame feed %s\n' % ('on' if pfeed == '1' else 'off (HIPENC_PFEED=0)'))
    print('| class (the two P frames = stream minus its keyframe) | seconds | share |')
    print('|---|---|---|')
    print('| both P frames, wall | %.3f | 100 %% |' % pt)
    print('| frame start -> prediction exists (input copy, od_mv_est, device OBMC) | %.3f | %.1f %% |' % (s3.pre_mc_s, 100*s3.pre_mc_s/pt))
    print('| waiting for the P-frame feed (device passes + libm stage on helpers) | %.3f | %.1f %% |' % (s3.t_pfeed_s, 100*s3.t_pfeed_s/pt))
    print('| rate-only pricing | %.3f | %.1f %% |' % (s3.rate_s - s1.rate_s, 100*(s3.rate_s - s1.rate_s)/pt))
    print('| host codeword searches | %.3f | %.1f %% |' % (s3.search_cpu_s - s1.search_cpu_s, 100*(s3.search_cpu_s - s1.search_cpu_s)/pt))
    print('\nP-frame searches from the feed %d, on the host %d\n' % (s3.dev_hits - s1.dev_hits,
          (s3.cpu_other + s3.cpu_noref_luma) - (s1.cpu_other + s1.cpu_noref_luma)))
