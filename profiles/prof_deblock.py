#!/usr/bin/env python3
"""Time the deblocking row pipeline on frame shapes that separate its two cost terms:
   T = (W/16) * C  +  (H/16) * lag        C = one macroblock step inside a row, lag = row-to-row hand-off.
Run on the GPU box: python profiles/prof_deblock.py"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jm_amd import JmHip
from jm_amd.lib import DB_MB, DB_MOTION


def run(w, h, reps=20, busy=True, smooth=False, real=False):
    dev = torch.device("cuda", 0)
    ctx = JmHip(w, h, search_range=16, num_ref_slots=1, yuv_format=1, stream=torch.cuda.current_stream().cuda_stream)
    ctx.enable_timing(True)
    rng = np.random.default_rng(1)
    nmb = (w // 16) * (h // 16)
    if smooth:   # blocky-smooth content: most alpha/beta tests pass, the filters really run (as on a reconstructed frame)
        base = np.kron(rng.integers(60, 200, (h // 16 + 1, w // 16 + 1)), np.ones((16, 16), np.int64))[:h, :w]
        yy = (base + rng.integers(-3, 4, (h, w))).clip(0, 255).astype(np.uint8)
    else:
        yy = rng.integers(0, 256, (h, w)).astype(np.uint8)
    y = torch.from_numpy(yy).to(dev)
    c = torch.from_numpy(rng.integers(100, 140, (2, h // 2, w // 2)).astype(np.uint8)).to(dev)
    mbs = np.zeros(nmb, DB_MB)
    if busy:
        mbs["mb_type"] = rng.choice([0, 1, 1, 2, 3, 8, 8, 9, 10], nmb)
        mbs["cbp_blk"] = rng.integers(0, 1 << 16, nmb) * (rng.integers(0, 3, nmb) > 0)
    else:
        mbs["mb_type"] = 1
    mbs["qp"], mbs["qpc"] = 28, 27
    mbs["cbp"] = np.where(mbs["cbp_blk"] != 0, 15, 0)
    mot = np.zeros((h // 4) * (w // 4), DB_MOTION)
    mot["ref_id"][:, 1] = -1
    if real:   # the P picture of BASELINE configs[1] as JM produced it (tests/golden/g2_sideinfo.npz); 1920x1088 only
        from jm_amd.lib import db_arrays_from_tap
        g2 = np.load(os.path.join(ROOT, "tests", "golden", "g2_sideinfo.npz"))
        mbs, mot = db_arrays_from_tap(g2["p_mbs"].astype(np.int32), g2["p_mot"].astype(np.int32))
        mbs, mot = mbs.copy(), mot.reshape(-1).copy()
    d_mbs = torch.from_numpy(mbs.view(np.uint8).reshape(nmb, -1)).to(dev)
    d_mot = torch.from_numpy(mot.view(np.uint8).reshape(len(mot), -1)).to(dev)
    wy, wc = y.clone(), c.clone()
    ts = []
    for _ in range(reps):
        wy.copy_(y); wc.copy_(c)
        ctx.deblock_frame_dev(wy.data_ptr(), w, wc[0].data_ptr(), wc[1].data_ptr(), w // 2, d_mbs.data_ptr(), d_mot.data_ptr(), 1)
        torch.cuda.synchronize()
        ts.append(ctx.last_kernel_ms(4))
    ctx.close()
    return float(np.median(ts[3:]))


def phases():
    """JMHIP_LIB must point at a library built by profiles/build_dbprof.sh"""
    import ctypes
    lib = ctypes.CDLL(os.environ["JMHIP_LIB"], mode=ctypes.RTLD_GLOBAL)
    names = ["wait at barrier A", "V phase / mover part 1", "wait at barrier B", "H phase / mover part 2"]
    for label, kw in (("real P-picture side info", dict(real=True, smooth=True)), ("made-up intra-heavy mix", dict(busy=True, smooth=True))):
        t = run(1920, 1088, reps=8, **kw)
        out = np.zeros(64 * 2 * 8, np.uint64)
        lib.jmhip_debug_read_db_prof(out.ctypes.data_as(ctypes.c_void_p))
        out = out.reshape(64, 2, 8).astype(np.float64)
        print(f"1920x1088, {label}: {t * 1e3:.1f} us; s_memtime ticks per step (123 steps per band)")
        t0 = out[0, 1, 4]
        print("  time line of the mover waves, us after band 0 entered its loop (100 MHz wall clock): band: loop entry, step 1, step 64, end")
        print("   " + "  ".join("%d: %.1f %.1f %.1f %.1f" % (b, *[(out[b, 1, k] - t0) / 100.0 for k in (4, 5, 6, 7)]) for b in range(17)))
        for b in (0, 1, 8, 16):
            for wv in (0, 1):
                print("  band %2d %s: " % (b, "filter" if wv == 0 else "mover ") + "  ".join("%s %5.0f" % (n, out[b, wv, i] / 123.0) for i, n in enumerate(names)))


if __name__ == "__main__":
    if "--phases" in sys.argv:
        phases()
        sys.exit(0)
    for (w, h) in [(1920, 16), (1920, 32), (16, 1088), (1920, 1088), (3840, 2160 // 16 * 16), (3840, 272)]:
        print(f"{w}x{h}: smooth+busy {run(w, h, busy=True, smooth=True) * 1e3:.1f} us   noise+busy {run(w, h, busy=True) * 1e3:.1f} us   "
              f"nothing to filter {run(w, h, busy=False) * 1e3:.1f} us")
