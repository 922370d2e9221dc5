"""Measurement aid: where a macroblock's time goes with pictures in flight (JMHIP_MB_PROF=1 time stamps, 100 MHz wall clock), the last picture of a short sequence.
usage: python profiles/seq_prof.py <depth> <mode fs|epzs|ffs>"""
import ctypes as C
import os
import sys
os.environ["JMHIP_MB_PROF"] = "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from jm_amd import JmHip
from jm_amd.lib import SLICE_PARAMS
depth, mode = int(sys.argv[1]), sys.argv[2]
npic = 2 * depth + 2
W, H = 1920, 1088
nmb = (W // 16) * (H // 16)
frames = bench.yuv_frames(npic)
nslots = depth + 2
ctx = JmHip(W, H, search_range=32, num_ref_slots=nslots, yuv_format=1)
ctx.seq_open(depth, 0, ready=True)
for k in range(npic):
    p = bench.slice_params(SLICE_PARAMS, 2 if k == 0 else 0, 0, nmb, 0, 0 if k == 0 else 1)
    if k:
        p["ref_slot"][0, 0], p["ref_id"][0, 0], p["poc_ref"][0, 0] = (k - 1) % nslots, k - 1, 2 * (k - 1)
    p["poc_cur"] = 2 * k
    if mode == "ffs":
        p["search_mode"] = 1
    if mode == "epzs":
        p["search_mode"], p["symbol_mode"] = 3, 1
        for kk, v in dict(pattern=2, dual=3, fixed=2, aggressive=0, temporal=1, spatial_mem=1, blocktype=1, min_scale=0, med_scale=1, max_scale=2, sub_scale=2).items():
            p["epzs_" + kk] = v
    e = k % depth
    if k >= depth:
        ctx.seq_wait(e)
    ctx.seq_set_frame(e, frames[k], 1920, 1080)
    ctx.seq_encode(e, p, k % nslots, 1, False)
for k in range(npic - depth, npic):
    ctx.seq_wait(k % depth)
st = np.zeros((nmb, 32), np.uint64)
assert ctx.lib.jmhip_debug_read_mb_prof(ctx.h, st.ctypes.data_as(C.c_void_p), st.nbytes) == 0
st = st.astype(np.int64)
ok = (st[:, 0] > 0) & (st[:, 17] > st[:, 0])
d = lambda a, b: np.median((st[ok, b] - st[ok, a]) / 100.0)        # microseconds
print(f"{mode} depth {depth}: macroblocks with stamps {ok.sum()}; median us: ticket->staged {d(0, 1):.1f}, staged->state in {d(1, 2):.1f}, ->wave 0's chain done {d(2, 8):.1f}, "
      f"->decided and coded {d(8, 16):.1f}, ->published {d(16, 17):.1f}; ticket->published {d(0, 17):.1f}; post stage: published->its start {d(17, 27):.1f}, waiting for the neighbours' post flags {d(27, 28):.1f}, DeblockMb {d(28, 29):.1f}, planes {d(29, 30):.1f}")
ctx.close()
