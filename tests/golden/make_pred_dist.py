#!/usr/bin/env python3
"""Make tests/golden/pred_dist.npz: known answers of the REAL reference's candidate-distortion functions.

TEST INFRASTRUCTURE; runs only where /root/reference exists (it needs oracle/_ref/libjmrefcall.so = the unmodified JM 19.0 lencod
objects + oracle/ref_call.c, built by `make -f oracle/Makefile.ref call`).  It calls, through that shim,
    computeSAD / computeSSE / computeSATD, compute*WP, computeBiPred*1, computeBiPred*2      (lencod/src/me_distortion.c:349-1530)
on seeded random blocks, candidates (far outside the picture included, so that UMVLine4X's origin clamps are reached), weights,
offsets, weight denominators and early-exit thresholds, and stores inputs and the values the reference returned.

The 16 sub-pel planes handed to the reference are made by the oracle's getSubImagesLuma restatement, which tests/test_oracle_golden.py
pins plane by plane against the reference's own planes (qcif_fs.npz); the fixture stores only the two source pictures.

    python tests/golden/make_pred_dist.py
"""
import ctypes as C
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyjmo as J          # noqa: E402  (only for the planes)

W, H, N = 96, 64, 900
SIZES = [(16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)]


def picture(rng, shift):
    base = np.kron(rng.integers(0, 256, (H // 8 + 2, W // 8 + 2)), np.ones((8, 8)))
    k = np.ones(5) / 5
    sm = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, base)
    sm = np.apply_along_axis(lambda c: np.convolve(c, k, "same"), 0, sm)
    y = sm[shift[1]:shift[1] + H, shift[0]:shift[0] + W] + rng.normal(0, 3, (H, W))
    return np.clip(np.rint(y), 0, 255).astype(np.uint8)


def main():
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libjmrefcall.so"))
    ref.refcall_pred_dist.restype = C.c_longlong
    ref.refcall_pred_dist.argtypes = [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_int] * 8 + [C.c_longlong] + [C.c_int] * 4
    rng = np.random.default_rng(20260929)
    cur = picture(np.random.default_rng(7), (4, 4))
    r1 = picture(np.random.default_rng(7), (2, 3))
    r2 = picture(np.random.default_rng(7), (6, 5))
    p1, p2 = J.RefPic(r1), J.RefPic(r2)
    rec = np.zeros((N, 20), np.int64)
    for n in range(N):
        pred, metric = int(rng.integers(0, 4)), int(rng.choice([0, 1, 2]))
        bsx, bsy = SIZES[int(rng.integers(0, len(SIZES)))]
        test8x8 = int(metric == 2 and bsx >= 8 and bsy >= 8 and rng.integers(0, 2))
        px, py = int(rng.integers(0, (W - bsx) // 4 + 1)) * 4, int(rng.integers(0, (H - bsy) // 4 + 1)) * 4
        far = rng.random() < 0.3                                       # candidates that leave the padded area: origin clamps
        span = 260 if far else 40
        c = [int(v) for v in rng.integers(-span, span + 1, 4)]
        denom = int(rng.integers(0, 8))
        wp_round = (1 << (denom - 1)) if denom else 0                  # slice.c: wp_luma_round
        if rng.random() < 0.5:                                         # implicit-style pair (sum 64 at denom 5) or anything explicit
            denom, wp_round = 5, 16
            w1 = int(rng.integers(-64, 129)); w2 = 64 - w1
        else:
            w1, w2 = int(rng.integers(-128, 128)), int(rng.integers(-128, 128))
        off = int(rng.integers(-128, 128)) if rng.random() < 0.6 else 0
        orig = np.ascontiguousarray(cur[py:py + bsy, px:px + bsx].astype(np.uint16))
        args = (pred, metric, W, H, J._p(p1.planes), J._p(p2.planes), J._p(orig), bsx, bsy, test8x8, w1, w2, off, denom, wp_round)
        cand = (px * 4 + c[0], py * 4 + c[1], px * 4 + c[2], py * 4 + c[3])
        full = int(ref.refcall_pred_dist(*args, J.DIST_MAX, *cand))
        mode = rng.random()                                            # early-exit threshold: none, just below, exactly at, just above
        thr = J.DIST_MAX if mode < 0.4 else max(0, full + int(rng.choice([-64, -32, -1, 0, 31, 32, 200])) * (1 if mode < 0.9 else 40))
        got = int(ref.refcall_pred_dist(*args, thr, *cand))
        rec[n] = [pred, metric, bsx, bsy, test8x8, px, py, c[0], c[1], c[2], c[3], w1, w2, off, denom, wp_round, thr, got, full, 0]
    np.savez_compressed(os.path.join(HERE, "pred_dist.npz"), cur=cur, ref1=r1, ref2=r2, records=rec,
                        columns=np.array("pred metric bsx bsy test8x8 pos_x pos_y c1x c1y c2x c2y w1 w2 offset log_denom wp_round "
                                         "min_mcost result full_result reserved".split()))
    print("pred_dist.npz:", N, "records;", int((rec[:, 17] != rec[:, 18]).sum()), "early exits")


if __name__ == "__main__":
    main()
