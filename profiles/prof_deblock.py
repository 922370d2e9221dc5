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


def run(w, h, reps=20, busy=True, smooth=False):
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


if __name__ == "__main__":
    for (w, h) in [(1920, 16), (1920, 32), (16, 1088), (1920, 1088), (3840, 2160 // 16 * 16), (3840, 272)]:
        print(f"{w}x{h}: smooth+busy {run(w, h, busy=True, smooth=True) * 1e3:.1f} us   noise+busy {run(w, h, busy=True) * 1e3:.1f} us   "
              f"nothing to filter {run(w, h, busy=False) * 1e3:.1f} us")
