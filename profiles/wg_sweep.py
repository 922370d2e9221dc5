#!/usr/bin/env python3
"""k_mb_pipe on configs[1]'s P picture with different numbers of persistent workgroups (jmhip_set_pipeline_workgroups): kernel time per setting.
Under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` the per-dispatch counters (Grid_Size tells the setting) show what the scratch footprint
(workgroups x 512 lanes x scratch bytes) costs in HBM traffic once it no longer fits the L2s.   usage: python profiles/wg_sweep.py 64 96 128 192 256"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from jm_amd import JmHip
from jm_amd.lib import SLICE_PARAMS

W, H, HS = bench.W, bench.H, bench.H_SRC
nmb = (W // 16) * (H // 16)
ctx = JmHip(W, H, search_range=bench.R, num_ref_slots=2, yuv_format=1)
raw0, raw1 = bench.yuv_frames(2)
ctx.set_current_frame(raw0, W, HS)
ctx.encode_slice_dev(bench.slice_params(SLICE_PARAMS, 2, 0, nmb, 0, 0))
ctx.deblock_picture_dev(1)
ctx.reference_from_recon(0)
ctx.set_current_frame(raw1, W, HS)
prm = bench.slice_params(SLICE_PARAMS, 0, 0, nmb, 0, 1)
prm["ref_slot"][0, 0] = 0
ctx.enable_timing(True)
ref = None
for wg in [int(a) for a in sys.argv[1:]] or [256]:
    ctx.set_pipeline_workgroups(wg)
    ms = []
    for i in range(4):
        ctx.encode_slice_dev(prm)
        ctx.synchronize()
        ms.append(ctx.last_kernel_ms(5))
    recs = ctx.encode_slice(prm).tobytes()
    ref = ref or recs
    print(f"workgroups {wg:4d}: k_mb_pipe {np.mean(ms[1:]):7.3f} ms (launches {', '.join(f'{m:.2f}' for m in ms)}), records equal to the first setting's: {recs == ref}")
