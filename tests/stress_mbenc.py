#!/usr/bin/env python3
"""Determinism stress of the macroblock pipeline (GPU box): the same P picture encoded again and again with identical inputs must give identical records -- any difference
is a race.  Several configurations (search modes, 8x8 transform, 4:2:2, slices of a few macroblocks in separate launches / side by side, streamed records).
usage: python tests/stress_mbenc.py <repeats per configuration>"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, d)
import synth_motion
import test_gpu_mbenc as T
from oracle import pyjmo

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
for (W, H, R, refs, qp, sm, t8, yuv, cabac, slice_mbs, together) in [
        (160, 96, 4, 2, 37, 3, 1, 1, 0, 3, False), (160, 96, 4, 2, 37, 3, 1, 1, 0, 3, True), (160, 96, 16, 2, 30, 3, 1, 1, 1, 0, False), (160, 96, 16, 1, 30, 3, 0, 1, 0, 0, False),
        (160, 96, 16, 2, 30, 0, 1, 1, 0, 7, True), (160, 96, 16, 2, 30, 1, 1, 2, 1, 0, False), (320, 192, 16, 2, 28, 3, 1, 2, 1, 0, False), (320, 192, 32, 1, 28, 0, 0, 1, 0, 0, False)]:
    f = int(192 * 2 ** ((qp - 28) / 6))
    lam = {2: ([f] * 3, f), 0: ([f, f + 3, f + 5], f + 1)}
    frames = synth_motion.motion_clip(W, H, 3, 5129, yuv422=yuv == 2)
    dev = T.DevSeqEncoder(W, H, qp, R, refs, lam, slice_mbs, together=together, cabac=cabac, search_mode=sm, transform8x8=t8, yuv_format=yuv)
    L, J = dev.L, dev.J
    for raw in frames[:2]:
        dev.encode(raw, W, H)
    # the third picture, again and again: the slices' launches exactly as DevSeqEncoder.encode makes them, blocking and streamed in turn
    nmb = (W // 16) * (H // 16)
    import mbenc_util
    J.set_current_frame(frames[2], W, H)
    nref = min(refs, len(dev.refs))
    slices = mbenc_util.slices_of(nmb, slice_mbs)
    first = None
    diffs = 0
    for it in range(N):
        recs = np.zeros(nmb, L.MB_RECORD)
        for sn, (fm, num) in enumerate(slices):
            cfg = pyjmo.mbenc_cfg(W, H, 0, fm, num, qp, R, nref, *lam[0], cabac=cabac, search_mode=sm, transform8x8=t8, yuv_format=yuv)
            prm = T.slice_params(L, cfg, sn, [r[0] for r in dev.refs[:nref]], [r[1] for r in dev.refs[:nref]], 0, None, 2 * dev.npic)
            if together and len(slices) > 1:
                prm["num_slices"] = len(slices)
                recs[:] = J.encode_slice_streamed(prm) if it & 1 else J.encode_slice(prm)
                break
            recs[fm:fm + num] = J.encode_slice_streamed(prm) if it & 1 else J.encode_slice(prm)
        b = recs.tobytes()
        if first is None:
            first, first_recs = b, recs.copy()
        elif b != first:
            diffs += 1
            k = [i for i in range(nmb) if recs[i].tobytes() != first_recs[i].tobytes()]
            if diffs <= 3:
                print("   differs at iteration", it, "macroblocks", k[:8], "fields", [n for n in recs.dtype.names if not np.array_equal(recs[k[0]][n], first_recs[k[0]][n])])
    print(f"{W}x{H} R{R} refs{refs} qp{qp} mode{sm} t8={t8} yuv={yuv} cabac={cabac} slices={slice_mbs}{'(one launch)' if together else ''}: {diffs} of {N - 1} repeats differ")
    bad += diffs
    J.close()
print("stress_mbenc:", "DETERMINISTIC" if bad == 0 else f"{bad} differing repeats")
sys.exit(1 if bad else 0)
