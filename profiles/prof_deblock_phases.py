#!/usr/bin/env python3
"""Per-phase cycle counts of the deblocking row pipeline (luma rows).  Needs a profiling build:
   hipcc ... -DJMHIP_DB_PROFILE (see the command at the bottom) -> /tmp/libjmhip_prof.so; JMHIP_LIB selects it."""
import ctypes as C
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jm_amd import JmHip, lib as L
from jm_amd.lib import DB_MB, DB_MOTION

w, h = 1920, 1088
dev = torch.device("cuda", 0)
ctx = JmHip(w, h, search_range=16, num_ref_slots=1, yuv_format=1, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
nmb = (w // 16) * (h // 16)
base = np.kron(rng.integers(60, 200, (h // 16 + 1, w // 16 + 1)), np.ones((16, 16), np.int64))[:h, :w]
y = torch.from_numpy((base + rng.integers(-3, 4, (h, w))).clip(0, 255).astype(np.uint8)).to(dev)
c = torch.from_numpy(rng.integers(100, 140, (2, h // 2, w // 2)).astype(np.uint8)).to(dev)
mbs = np.zeros(nmb, DB_MB)
mbs["mb_type"] = rng.choice([0, 1, 1, 2, 3, 8, 8, 9, 10], nmb)
mbs["cbp_blk"] = rng.integers(0, 1 << 16, nmb) * (rng.integers(0, 3, nmb) > 0)
mbs["qp"], mbs["qpc"] = 28, 27
mbs["cbp"] = np.where(mbs["cbp_blk"] != 0, 15, 0)
mot = np.zeros((h // 4) * (w // 4), DB_MOTION); mot["ref_id"][:, 1] = -1
d_mbs = torch.from_numpy(mbs.view(np.uint8).reshape(nmb, -1)).to(dev)
d_mot = torch.from_numpy(mot.view(np.uint8).reshape(len(mot), -1)).to(dev)
for _ in range(3):
    wy, wc = y.clone(), c.clone()
    ctx.deblock_frame_dev(wy.data_ptr(), w, wc[0].data_ptr(), wc[1].data_ptr(), w // 2, d_mbs.data_ptr(), d_mot.data_ptr(), 1)
    torch.cuda.synchronize()
mb_h = h // 16
buf = np.zeros(64 + mb_h * 2 * 6 * 8, np.uint8)
lib = L.load_library()
lib.jmhip_debug_read_db_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert lib.jmhip_debug_read_db_sync(ctx.h, buf.ctypes.data, buf.nbytes) == 0
acc = buf[64:].view(np.uint64).reshape(2, mb_h, 6)[0].astype(np.float64) / (w // 16)     # cycles per step (100 MHz counter?)
names = ["sync@top", "loads+V", "gran-load+H", "handover+stores+tile", "await+tile-top", "-"]
for r in (0, 1, 2, 30, 67):
    print("row", r, " ".join(f"{n}={acc[r][k]:.1f}" for k, n in enumerate(names[:5])), "total", acc[r][:5].sum())
print("mean rows 1..66:", " ".join(f"{n}={acc[1:67, k].mean():.1f}" for k, n in enumerate(names[:5])), "total", acc[1:67, :5].sum(1).mean())
