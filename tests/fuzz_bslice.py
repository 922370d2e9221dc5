#!/usr/bin/env python3
"""Randomised parity run of the B pictures (GPU box): jmhip_encode_slice with slice_type 1 (jm_amd/csrc/mbpipe_b.inc) against the oracle's restatement (oracle/jmo_mbenc_b.inc,
itself pinned to the real encoder's B slices by tests/golden/mb_low_*b*.npz and by tests/fuzz_oracle.py from seed 1 000 000) on seeded random configurations: picture size, search
range, full search / fast full search, one or two B pictures between the references, the lengths of both lists, QPs, slices, CAVLC / CABAC, 8x8 transform, 4:2:0 / 4:2:2,
direct_8x8_inference, the bi-predictive search's switches (BiPredMotionEstimation, BiPredSearch16x16 / 16x8 / 8x16, refinements, range, sub-pel levels), B/PSliceSearch* switches.
Every macroblock record of every picture and the reconstruction before and after the loop filter must be identical.
TEST INFRASTRUCTURE (uses oracle/).   usage: python tests/fuzz_bslice.py <seconds> [first seed]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, d)
import mb_tap, mbenc_util, synth_motion
import test_gpu_mbenc as T
import test_gpu_bslice as TB
from oracle import pyjmo

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
offsets = pyjmo.load_q_offsets(os.path.join(ROOT, "tests", "golden", "q_offset.cfg"))
t0, done, nb_pics, with_bipred = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    sm = int(rng.choice([0, 1]))                              # pyjmo search_mode: 0 full search, 1 fast full search
    R = int(rng.choice([4, 8, 16, 32])) if rng.integers(0, 3) else int(rng.integers(2, 33))
    big = R > 16
    W = 16 * int(rng.integers(3, 9 if big else 14)); H = 16 * int(rng.integers(3, 7 if big else 10))
    keep = int(rng.integers(1, 4 if big else 5))              # NumberReferenceFrames
    nb = int(rng.choice([1, 1, 2]))                           # NumberBFrames
    qp, qp_p, qp_b = (int(rng.integers(8, 46)) for _ in range(3))
    t8, yuv, cabac = int(rng.integers(0, 2)), int(rng.choice([1, 1, 1, 2])), int(rng.integers(0, 2))
    nmb = (W // 16) * (H // 16)
    slice_mbs = 0 if rng.integers(0, 3) else int(rng.integers(2, nmb))
    offs = offsets if rng.integers(0, 3) == 0 else None
    b = dict(direct_8x8_inference=int(rng.integers(0, 2)), direct_temporal=0, bipred_me=int(rng.integers(0, 3) > 0), bipred_search=[int(rng.integers(0, 4) > 0), int(rng.integers(0, 2)), int(rng.integers(0, 2)), 0],
             bipred_refinements=int(rng.integers(0, 4)), bipred_range=int(rng.choice([r for r in (2, 4, 8, 16) if r <= R])), bipred_subpel=int(rng.integers(0, 3)))
    if seed >= 2500000 and rng.integers(0, 2):                # temporal direct (needs direct_8x8_inference on the device: one reference per 8x8 block in the record)
        b.update(direct_temporal=1, direct_8x8_inference=1)
    iv = [1] * 8 if rng.integers(0, 4) else [1, 1] + [int(rng.integers(0, 2)) for _ in range(6)]
    ivb = None if rng.integers(0, 3) else [int(rng.integers(0, 3) > 0)] + [int(rng.integers(0, 4) > 0) for _ in range(7)]
    if t8:                                                    # the library turns Transform8x8Mode 1 without the 8x8 partition away (as the adapter does)
        iv[4] = 1
        if ivb:
            ivb[4] = 1
    n0, n1 = int(rng.integers(1, keep + 1)), int(rng.integers(1, 3))       # B(List0/1)References
    ngop = int(rng.integers(1, 4))
    nfr = 1 + ngop * (nb + 1)
    kind = str(rng.choice(["motion", "motion", "motion", "noise", "stripes", "still"]))
    if kind == "motion" or yuv == 2:
        frames = synth_motion.motion_clip(W, H, nfr, seed, yuv422=yuv == 2)
    else:
        frames = T.hard_clip(kind, W, H, nfr, seed)
    lam_of = lambda q: int(192 * 2 ** ((q - 28) / 6))
    f, fp, fb = lam_of(qp), lam_of(qp_p), lam_of(qp_b)
    lam = {2: ([f] * 3, f), 0: ([fp, fp + 3, fp + 5], fp + 1)}
    lam_b = ([fb + 1, fb + 2, fb + 6], fb + 2)
    desc = dict(seed=seed, W=W, H=H, R=R, keep=keep, nb=nb, qp=(qp, qp_p, qp_b), mode=sm, t8=t8, yuv=yuv, cabac=cabac, slice_mbs=slice_mbs, offsets=offs is not None, clip=kind, b=b, iv=iv, ivb=ivb, lists=(n0, n1), gops=ngop)
    args = (W, H, qp, R, keep, lam, slice_mbs)
    kw = dict(cabac=cabac, search_mode=sm, transform8x8=t8, yuv_format=yuv, offsets=offs, inter_valid=iv, qp_p=qp_p)
    try:
        dev = TB.DevSeqEncoderB(*args, **kw)
    except Exception as e:                                   # a configuration the library turns away (LDS budget): say so and go on
        print("skipped", desc, str(e)[:120]); seed += 1; continue
    ora = mbenc_util.SeqEncoder(*args, **kw)
    order = [(0, 2)]                                          # (display number, slice type) in coding order: I, then per group the P picture and the B pictures before it
    for g in range(ngop):
        p = (g + 1) * (nb + 1)
        order += [(p, 0)] + [(p - nb + k, 1) for k in range(nb)]
    stored = []                                               # picture order counts of the stored reference pictures, most recent first
    try:
        for disp, st in order:
            raw, poc = frames[disp], 2 * disp
            src = pyjmo.load_frame(raw, W, H, W, H, yuv)
            if st == 1:
                past, future = sorted([p for p in stored if p < poc], reverse=True), sorted(p for p in stored if p > poc)
                l0, l1 = (past + future)[:n0], (future + past)[:n1]
                recs, pre, post = dev.encode_b(raw, W, H, l0, l1, lam_b, qp_b, b, inter_valid_b=ivb, poc=poc)
                orecs, _, opre, opost = ora.encode_b(src, poc, l0, l1, lam_b, qp_b, b, inter_valid_b=ivb)
                nb_pics += 1
            else:
                recs, pre, post = dev.encode_ref(raw, W, H, poc)
                orecs, _, opre, opost = ora.encode(src, poc=poc)
                stored = ([poc] + stored)[:keep]
            d = T.first_difference(mb_tap.canonical(orecs, bslice=st == 1), mb_tap.canonical(T.as_oracle_records(recs), bslice=st == 1))
            assert d is None, ("records", disp, st, d[:3])
            assert all(np.array_equal(a, x.astype(np.uint8)) for a, x in zip(pre, opre)), ("reconstruction before the loop filter", disp, st)
            assert all(np.array_equal(a, x.astype(np.uint8)) for a, x in zip(post, opost)), ("reconstruction after the loop filter", disp, st)
    except Exception as e:
        print("FAILED", desc, repr(e)[:800])
        sys.exit(1)
    finally:
        dev.J.close()
    done += 1; with_bipred += b["bipred_me"]; seed += 1
print(f"fuzz_bslice: {done} random sequences with {nb_pics} B pictures identical to the oracle in {time.time() - t0:.0f} s ({with_bipred} with the bi-predictive search); next seed {seed}")
