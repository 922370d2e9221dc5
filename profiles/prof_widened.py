#!/usr/bin/env python3
"""Times the kernels of the widened rows on one 1080p picture's worth of work, inputs resident in HBM:
   k_mc_luma (every macroblock as one 16x16 block, and as sixteen 4x4 blocks), k_mc_chroma (eight 4x4 blocks per macroblock),
   k_tq_luma16x16, k_tq_luma8x8, k_tq_chroma.  Prints time, algorithmic bytes and the fraction of the 8 TB/s HBM peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jm_amd import JmHip
from jm_amd.lib import MC_LUMA_BLK, MC_CHROMA_BLK, TQ16_OUT, TQ8_OUT, TQC_OUT, TQC_MB

W, H = 1920, 1088
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
ctx = JmHip(W, H, search_range=32, num_ref_slots=2, yuv_format=1, device=0, stream=stream.cuda_stream)
rng = np.random.default_rng(3)
for s in range(2):
    ctx.set_reference(s, rng.integers(0, 256, (H, W)).astype(np.uint8))
    ctx.set_reference_chroma(s, rng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8), rng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8))
nmb = (W // 16) * (H // 16)
mbx, mby = np.tile(np.arange(W // 16) * 16, H // 16), np.repeat(np.arange(H // 16) * 16, W // 16)


def timed(label, fn, alg_bytes, reps=20):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(stream); fn(); b.record(stream)
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    print(f"{label:44s} {ms * 1e3:8.1f} us   {alg_bytes / 1e6:7.2f} MB algorithmic   {alg_bytes / (ms * 1e-3) / 8e12 * 100:5.1f} % of HBM peak")


def dev_arr(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(len(a), -1)).to(dev)


# ---- luma prediction
for label, bs, per in (("k_mc_luma, 16x16 blocks, one list", 16, 1), ("k_mc_luma, 4x4 blocks, one list", 4, 16)):
    n = nmb * per
    b = np.zeros(n, MC_LUMA_BLK)
    k = np.arange(n) % per
    b["x"] = np.repeat(mbx, per) + (k % 4) * 4 * (bs == 4); b["y"] = np.repeat(mby, per) + (k // 4) * 4 * (bs == 4)
    b["w"] = b["h"] = bs
    b["mv"][:, 0] = np.array([12, 8]) + rng.integers(-6, 7, (n, 2))
    d_b, d_o = dev_arr(b), torch.empty((n, 256), dtype=torch.uint8, device=dev)
    timed(label, lambda: ctx.mc_luma_dev(d_b.data_ptr(), n, d_o.data_ptr()), n * (2 * bs * bs + 20))
n = nmb
b = np.zeros(n, MC_LUMA_BLK)
b["x"], b["y"], b["w"], b["h"], b["dir"] = mbx, mby, 16, 16, 2
b["slot"][:, 1] = 1
b["mv"] = np.array([12, 8]) + rng.integers(-6, 7, (n, 2, 2))
d_b, d_o = dev_arr(b), torch.empty((n, 256), dtype=torch.uint8, device=dev)
timed("k_mc_luma, 16x16 blocks, both lists", lambda: ctx.mc_luma_dev(d_b.data_ptr(), n, d_o.data_ptr()), n * (3 * 256 + 20))
# ---- chroma prediction: 4 blocks per plane per macroblock
n = nmb * 8
c = np.zeros(n, MC_CHROMA_BLK)
k = np.arange(n) % 8
c["x"] = np.repeat(mbx // 2, 8) + (k % 2) * 4; c["y"] = np.repeat(mby // 2, 8) + ((k // 2) % 2) * 4; c["plane"] = k // 4
c["mv"][:, 0] = np.array([12, 8]) + rng.integers(-6, 7, (n, 4, 2, 2))
d_c, d_oc = dev_arr(c), torch.empty((n, 16), dtype=torch.uint8, device=dev)
timed("k_mc_chroma, 4x4 blocks, one list", lambda: ctx.mc_chroma_dev(d_c.data_ptr(), n, d_oc.data_ptr()), n * (25 + 16 + 72))
# ---- Intra16x16
q = np.zeros((16, 3), np.int32)
sc, ds = {0: 8192, 1: 3355, 2: 5243}, {0: 16, 1: 25, 2: 20}
for j in range(4):
    for i in range(4):
        cl = 0 if (i % 2 == 0 and j % 2 == 0) else (1 if (i % 2 and j % 2) else 2)
        q[j * 4 + i] = (682 << (15 + 4 - 11), sc[cl], ds[cl] << 4)
prm = ctx.tq_params(q, 4, cavlc=1, adaptive_rounding=1, adapt_rnd_weight=4)
orig = rng.integers(0, 256, (nmb, 256)).astype(np.uint8)
pred = np.clip(orig.astype(np.int32) + rng.integers(-12, 13, (nmb, 256)), 0, 255).astype(np.uint8)
d_or, d_pr = torch.from_numpy(orig).to(dev), torch.from_numpy(pred).to(dev)
d_o16 = torch.empty((nmb, TQ16_OUT.itemsize), dtype=torch.uint8, device=dev)
timed("k_tq_luma16x16, every macroblock", lambda: ctx.tq_luma16x16_dev(prm, d_or.data_ptr(), d_pr.data_ptr(), nmb, d_o16.data_ptr()), nmb * (512 + TQ16_OUT.itemsize))
ctx.close()
