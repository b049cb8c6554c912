#!/usr/bin/env python3
"""od_mv_est (src/mcenc.c:6390) split by stage on a 1080p I P P ... stream through the live
encoder seam (stage timers of daala_amd/host/mcenc_tail.c; one worker, frames in order).
  python tools/mvest_stages.py [--frames 4] [--no-device] [--scale 1]
Writes a markdown table to stdout (committed as profiles/r04_mvest_stages*.md)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

NAMES = ['EPZS initialisation, previous reference (`od_mv_est_init_mvs`, :3036)',
         'EPZS initialisation, golden reference',
         '`od_mv_est_calc_sads` (:3761)',
         '`od_mv_est_init_dus` without calc_sads (:3970)',
         'decimation loop (`od_mv_est_decimate` without init_dus, :4024)',
         'refinement loop (`od_mv_est_refine`, DP rows/columns, :6176)',
         '`od_mv_subpel_refine` (:6325)']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--no-device', action='store_true')
    ap.add_argument('--scale', type=int, default=1)
    ap.add_argument('--quant', type=int, default=20)
    a = ap.parse_args()
    import daala_amd.hipenc as H
    from configs_round import frames_of
    w, h, nf = 1920//a.scale, 1080//a.scale//8*8, a.frames
    buf = H.pack_frames(frames_of(w, h, nf, 31, step=(2, 3)), w, h)
    prm = H.Params(w, h, a.quant, 7, 1, 1, 0, 0, 30)
    os.environ['HIPENC_TIME'] = '1'
    n, got, st = H.encode(prm, buf, nf, use_device=0 if a.no_device else 1)
    assert n > 0, n
    s = list(st.mv_stage_s)
    tot = s[7]
    s[5] = tot - sum(s[:5]) - s[6]
    npf = nf - 1
    print('# `od_mv_est` by stage: %dx%d, 1 keyframe + %d P frames, one worker, device %s\n'
          % (w, h, npf, 'off' if a.no_device else 'on'))
    print('Whole stream %.3f s; frame CPU time %.3f s; `od_mv_est` %.3f s = %.3f s per P frame.\n'
          % (st.t_total_s, st.frame_cpu_s, tot, tot/max(1, npf)))
    print('| stage | s per P frame | share of `od_mv_est` |')
    print('|---|---|---|')
    for i in range(7):
        print('| %s | %.4f | %.1f %% |' % (NAMES[i], s[i]/max(1, npf), 100*s[i]/max(tot, 1e-9)))
    if st.mv_dev_calls:
        print('\nDevice batches: %d calls, %d block SADs, %.4f s waiting per P frame.'
              % (st.mv_dev_calls, st.mv_dev_sads, st.mv_dev_wait_s/max(1, npf)))
    if st.mv_level_walks:
        print('\nEPZS initialisation walked level by level %d times; %d device calls built the 9x9 block-matching '
              'windows of %d vertices; `od_mv_est_bma_sad` answered from a window %d times, on the host %d times '
              '(level 0 and vectors outside a window).'
              % (st.mv_level_walks, st.mv_bma_calls, st.mv_bma_windows, st.mv_bma_hits, st.mv_bma_misses))


if __name__ == '__main__':
    main()
