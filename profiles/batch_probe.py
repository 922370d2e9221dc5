"""Measurement aid: configs[1] (1080p, SR 32, one reference, QP 28, G2r's flags) as a real IPPP sequence, the P pictures in ONE launch (jmhip_seq_batch).
usage: python profiles/batch_probe.py [pictures] [slots,slots,...] [workgroups,workgroups,...] [fs|ffs|ffs3]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from jm_amd import JmHip  # noqa: E402
from jm_amd.lib import SLICE_PARAMS, MB_RECORD  # noqa: E402

npic = int(sys.argv[1]) if len(sys.argv) > 1 else 33
slots = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [20]
wgs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [256]
mode = sys.argv[4] if len(sys.argv) > 4 else "fs"
W, H = 1920, 1088
nmb = (W // 16) * (H // 16)
frames = bench.yuv_frames(npic)
d_raw = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in frames]
d_rec = torch.zeros(npic * nmb * MB_RECORD.itemsize, dtype=torch.uint8, device="cuda")
nref_max = 3 if mode == "ffs3" else 1
for nslots in slots:
    for wg in wgs:
        ctx = JmHip(W, H, search_range=32, num_ref_slots=nslots, yuv_format=1)
        ctx.seq_open(1)
        ctx.set_pipeline_workgroups(wg)
        ctx.enable_timing(True)

        def prm(k, nref):
            p = bench.slice_params(SLICE_PARAMS, 2 if k == 0 else 0, 0, nmb, 0, nref)
            for r in range(nref):
                p["ref_slot"][0, r] = (k - 1 - r) % nslots
                p["ref_id"][0, r] = k - 1 - r
            if mode.startswith("ffs"):
                p["search_mode"] = 1
            return p
        for rep in range(2):
            for k in range(nref_max):                             # the I picture (and the P pictures with fewer references), untimed
                ctx.seq_set_frame_dev(0, d_raw[k].data_ptr(), 1920, 1080)
                ctx.seq_encode(0, prm(k, min(k, nref_max)), k % nslots, 1, False)
                ctx.seq_wait(0)
            pics = [dict(d_raw=d_raw[k].data_ptr(), src_w=1920, src_h=1080, out_slot=k % nslots, ref_slot=[(k - 1 - r) % nslots for r in range(nref_max)],
                         ref_id=[k - 1 - r for r in range(nref_max)], d_records=d_rec.data_ptr() + k * nmb * MB_RECORD.itemsize) for k in range(nref_max, npic)]
            ctx.synchronize()
            t0 = time.perf_counter()
            ctx.seq_batch(prm(nref_max, nref_max), pics)
            ctx.synchronize()
            dt = time.perf_counter() - t0
        n = npic - nref_max
        print(f"{mode} one launch, {nslots} slots, {wg} workgroups: {n} P pictures in {dt * 1e3:.1f} ms (kernel {ctx.last_kernel_ms(5):.1f} ms) = {dt / n * 1e3:.2f} ms per picture = {nmb * n / dt / 1e3:.0f} k macroblocks/s", flush=True)
        ctx.close()
