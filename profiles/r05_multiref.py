#!/usr/bin/env python3
"""P pictures with several references at 1080p (SearchRange 32, CAVLC, 4x4 transform, QP 28; the clip of bench.py): k_mb_pipe's launch alone and with eight pictures in
flight, full search (search_mode 0) and fast full search (1), 1 / 2 / 3 / 5 references.  usage: python profiles/r05_multiref.py  (GPU box; JMHIP_MB_NO_HELP=1: the waves' fixed
roles of rounds 2-4)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
from jm_amd import JmHip
from jm_amd.lib import SLICE_PARAMS, MB_RECORD
import hashlib

W, H, nmb = bench.W, bench.H, 8160
frames = bench.yuv_frames(8)
d_raw = torch.from_numpy(np.stack(frames)).cuda()
for sm in (0, 1):
    for nref in (1, 2, 3, 5):
        depth, npic = 8, 25
        nslots = nref + depth + 1
        ctx = JmHip(W, H, search_range=32, num_ref_slots=nslots, yuv_format=1)
        ctx.enable_timing(True)
        def prm(k, st):
            n = min(nref, k) if st == 0 else 0
            q = bench.slice_params(SLICE_PARAMS, st, 0, nmb, 0, n)
            q["search_mode"] = sm
            for r in range(n):
                q["ref_slot"][0, r], q["ref_id"][0, r] = (k - 1 - r) % nslots, k - 1 - r
            return q
        # picture after picture up to picture nref + 1 (the first with all its references), that one's launch alone three times
        for k in range(nref + 2):
            ctx.set_current_frame(frames[k % len(frames)], W, bench.H_SRC)
            ms = []
            for rep in range(3 if k == nref + 1 else 1):
                recs = ctx.encode_slice(prm(k, 2 if k == 0 else 0))
                ms.append(ctx.last_kernel_ms(5))
            ctx.deblock_picture_dev(1)
            ctx.reference_from_recon(k % nslots)
        alone = float(np.mean(ms[1:]))
        md5 = hashlib.md5(recs.tobytes()).hexdigest()[:12]
        ctx.seq_open(depth, 0, ready=True)
        d_recs = torch.zeros((npic, nmb * MB_RECORD.itemsize), dtype=torch.uint8, device="cuda")
        def run():
            for k in range(npic):
                e = k % depth
                ctx.seq_set_frame_dev(e, d_raw[k % len(frames)].data_ptr(), W, bench.H_SRC)
                ctx.seq_encode(e, prm(k, 2 if k == 0 else 0), k % nslots, 1, False, d_recs[k].data_ptr())
        run(); torch.cuda.synchronize(); ctx.synchronize()
        t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ctx.synchronize()
        md5f = hashlib.md5(d_recs[nref + 1].cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"search_mode {sm}  references {nref}: launch alone {alone:7.2f} ms   in flight ({depth}) {dt / npic * 1e3:6.2f} ms per picture   records md5 {md5} / in flight {md5f}", flush=True)
        ctx.seq_close(); ctx.close()
