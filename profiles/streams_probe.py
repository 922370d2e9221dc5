"""How several sequences share one MI355X: k_mb_pipe alone and the whole step, S contexts on their own HIP streams.  gpu only."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from jm_amd import JmHip
from jm_amd.lib import SLICE_PARAMS
W, H, R = bench.W, bench.H, bench.R
f = bench.yuv_frames(2)
raw0, raw1 = f[0], f[1]
import types
# bench.main's slice_prm is a closure; rebuild the same record here through a tiny stand-in
def slice_prm(st, first, num, nr, nref):
    src = open(os.path.join(ROOT, "bench.py")).read()
    a = src.index("    def slice_prm("); b = src.index("    raw0, src_h = tall(0)")
    ns = {"np": np, "SLICE_PARAMS": SLICE_PARAMS, "QP": bench.QP, "R": R, "N": 1}
    exec("def _f():\n" + src[a:b] + "    return slice_prm\n", ns)
    return ns["_f"]()(st, first, num, nr, nref)
def run(S, share, whole, steps=8):
    streams = [torch.cuda.Stream() for _ in range(S)]
    ctxs = []
    for st in streams:
        c = JmHip(W, H, search_range=R, num_ref_slots=2, yuv_format=1, stream=st.cuda_stream)
        c.set_pipeline_workgroups(share)
        c.set_current_frame(raw0, W, bench.H_SRC); c.encode_slice_dev(slice_prm(2, 0, 8160, 0, 0)); c.deblock_picture_dev(1); c.reference_from_recon(0)
        c.set_current_frame(raw1, W, bench.H_SRC)
        ctxs.append(c)
    prm = slice_prm(0, 0, 8160, 0, 1); prm["ref_slot"][0, 0] = 0
    def rnd():
        for c in ctxs:
            c.encode_slice_dev(prm)
            if whole:
                c.deblock_picture_dev(1); c.reference_from_recon(1)
    rnd(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): rnd()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    for c in ctxs: c.synchronize(); c.close()
    print(f"{S} streams x {share} workgroups, {'whole step' if whole else 'k_mb_pipe only'}: {dt*1e3:.1f} ms per round = {S*8160/dt/1e3:.0f} k macroblocks/s", flush=True)
cases = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(1, 256), (1, 64), (1, 32), (2, 64), (4, 64), (4, 32), (8, 32)]
for S, share in cases:
    run(S, share, False)
for S, share in cases[-2:]:
    run(S, share, True)
