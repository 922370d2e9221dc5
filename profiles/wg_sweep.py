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

if os.environ.get("SWEEP_2160P") == "1":                 # configs[3]: 3840x2160, 8 slices of 4080 macroblocks in one launch
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synclip
    W, H, HS, per, ns = 3840, 2160, 2160, 4080, 8
    with tempfile.TemporaryDirectory() as t:
        synclip.syn2160p(os.path.join(t, "s.yuv"), 2)
        data = np.fromfile(os.path.join(t, "s.yuv"), np.uint8)
    fs = W * H * 3 // 2
    raw0, raw1 = data[:fs].copy(), data[fs:].copy()
else:
    W, H, HS = bench.W, bench.H, bench.H_SRC
    per, ns = (W // 16) * (H // 16), 0
    raw0, raw1 = bench.yuv_frames(2)
nmb = (W // 16) * (H // 16)
ctx = JmHip(W, H, search_range=bench.R, num_ref_slots=2, yuv_format=1)
ctx.set_current_frame(raw0, W, HS)
ctx.encode_slice_dev(bench.slice_params(SLICE_PARAMS, 2, 0, per, 0, 0, num_slices=ns))
ctx.deblock_picture_dev(1)
ctx.reference_from_recon(0)
ctx.set_current_frame(raw1, W, HS)
prm = bench.slice_params(SLICE_PARAMS, 0, 0, per, 0, 1, num_slices=ns)
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
