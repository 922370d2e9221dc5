"""Measurement aid: configs[1] (1080p, SR 32, one reference, QP 28, G2r's flags) as a real IPPP sequence with `depth` pictures in flight (jmhip_seq_*).
usage: python profiles/seq_probe.py [pictures] [depth,depth,...] [workgroups]"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from jm_amd import JmHip  # noqa: E402
from jm_amd.lib import SLICE_PARAMS  # noqa: E402

npic = int(sys.argv[1]) if len(sys.argv) > 1 else 24
depths = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
wg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mode = sys.argv[4] if len(sys.argv) > 4 else "fs"           # fs | ffs | epzs (configs[2]'s switches, CABAC, 8x8 transform off) | epzs5 (the same with five references) | epzs8 (configs[2]: 8x8 transform on) | ffs3 (three references)
W, H = 1920, 1088
nmb = (W // 16) * (H // 16)
frames = bench.yuv_frames(npic)
for depth in depths:
    nslots = depth + (4 if mode == "ffs3" else (6 if mode == "epzs5" else 2))
    ctx = JmHip(W, H, search_range=32, num_ref_slots=nslots, yuv_format=1)
    ctx.seq_open(depth, wg, ready=True)
    ctx.enable_timing(True)

    def prm(k):
        nref = 0 if k == 0 else (min(k, 3) if mode == "ffs3" else (min(k, 5) if mode == "epzs5" else 1))
        p = bench.slice_params(SLICE_PARAMS, 2 if k == 0 else 0, 0, nmb, 0, nref)
        for r in range(nref):
            p["ref_slot"][0, r] = (k - 1 - r) % nslots
            p["ref_id"][0, r] = k - 1 - r
            p["poc_ref"][0, r] = 2 * (k - 1 - r)
        p["poc_cur"] = 2 * k
        if mode.startswith("ffs"):
            p["search_mode"] = 1
        if mode in ("epzs", "epzs5"):
            p["search_mode"], p["symbol_mode"] = 3, 1
            for kk, v in dict(pattern=2, dual=3, fixed=2, aggressive=0, temporal=1, spatial_mem=1, blocktype=1, min_scale=0, med_scale=1, max_scale=2, sub_scale=2).items():
                p["epzs_" + kk] = v
        if mode == "epzs8":
            p = bench.configs2_params(p, 2 if k == 0 else 0, 2 * k)
        return p
    # all source pictures resident before the clock starts: one context entry per picture would be the product's ring; here the frames are re-uploaded per entry, untimed first pass
    for rep in range(2):
        ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(npic):
            e = k % depth
            if k >= depth:
                ctx.seq_wait(e)
            ctx.seq_set_frame(e, frames[k], 1920, 1080)
            ctx.seq_encode(e, prm(k), k % nslots, 1, False)
        for k in range(max(0, npic - depth), npic):
            ctx.seq_wait(k % depth)
        dt = time.perf_counter() - t0
    kms = [ctx.seq_kernel_ms(e) for e in range(min(depth, npic))]
    print(f"   launches' durations (HIP events, the last in each entry): {' '.join(f'{x:.1f}' for x in kms)} ms")
    print(f"{mode} depth {depth} workgroups/picture {wg or min(80, 256 // depth)}: {npic} pictures in {dt * 1e3:.1f} ms = {dt / npic * 1e3:.2f} ms per picture = {nmb * npic / dt / 1e3:.0f} k macroblocks/s", flush=True)
    ctx.close()
