"""Measurement aid: where a macroblock's time goes when the P pictures share one launch (jmhip_seq_batch; JMHIP_MB_PROF=1 time stamps, 100 MHz wall clock; every macroblock
address keeps the stamps of the last picture that wrote them).   usage: python profiles/batch_prof.py [pictures] [fs|ffs] [1|11]"""
import ctypes as C
import os
import sys
os.environ["JMHIP_MB_PROF"] = sys.argv[3] if len(sys.argv) > 3 else "1"     # 1: time stamps; 11: the integer searches' absolute differences as issued
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from jm_amd import JmHip
from jm_amd.lib import SLICE_PARAMS, MB_RECORD
npic = int(sys.argv[1]) if len(sys.argv) > 1 else 25
mode = sys.argv[2] if len(sys.argv) > 2 else "fs"
W, H = 1920, 1088
nmb = (W // 16) * (H // 16)
frames = bench.yuv_frames(npic)
d_raw = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in frames]
d_rec = torch.zeros(npic * nmb * MB_RECORD.itemsize, dtype=torch.uint8, device="cuda")
nslots = 24
ctx = JmHip(W, H, search_range=32, num_ref_slots=nslots, yuv_format=1)
ctx.seq_open(1)


def prm(k):
    p = bench.slice_params(SLICE_PARAMS, 2 if k == 0 else 0, 0, nmb, 0, 0 if k == 0 else 1)
    if k:
        p["ref_slot"][0, 0], p["ref_id"][0, 0] = (k - 1) % nslots, k - 1
    if mode == "ffs":
        p["search_mode"] = 1
    return p
ctx.seq_set_frame_dev(0, d_raw[0].data_ptr(), 1920, 1080)
ctx.seq_encode(0, prm(0), 0, 1, False)
ctx.seq_wait(0)
ctx.seq_batch(prm(1), [dict(d_raw=d_raw[k].data_ptr(), src_w=1920, src_h=1080, out_slot=k % nslots, ref_slot=[(k - 1) % nslots], ref_id=[k - 1],
                            d_records=d_rec.data_ptr() + k * nmb * MB_RECORD.itemsize) for k in range(1, npic)])
import time
ctx.synchronize()
t0 = time.perf_counter()
ctx.seq_batch(prm(1), [dict(d_raw=d_raw[k].data_ptr(), src_w=1920, src_h=1080, out_slot=k % nslots, ref_slot=[(k - 1) % nslots], ref_id=[k - 1],
                            d_records=d_rec.data_ptr() + k * nmb * MB_RECORD.itemsize) for k in range(1, npic)])
ctx.synchronize()
dt = time.perf_counter() - t0
st = np.zeros((nmb, 32), np.uint64)
assert ctx.lib.jmhip_debug_read_mb_prof(ctx.h, st.ctypes.data_as(C.c_void_p), st.nbytes) == 0
st = st.astype(np.int64)
if os.environ["JMHIP_MB_PROF"] == "11":
    issued = 16 * st[:, 22:30].sum()                             # one picture's worth: every address holds the counts of the last picture that coded it
    per_wave = 16 * st[:, 22:30].sum(axis=0)
    jm = 7 * 256 * 65 * 65 * nmb
    rate = issued * (npic - 1) / dt
    print(f"{mode}, {npic - 1} P pictures in one launch, {dt / (npic - 1) * 1e3:.2f} ms per picture (counting on): absolute differences the integer searches issued per picture {issued / 1e9:.3f} G "
          f"(JM's full search visits {jm / 1e9:.2f} G: 7 block types x 256 samples x 4225 positions x 8160 macroblocks; the cost bound of me_fullsearch.c:83 leaves the device {100 * issued / jm:.1f} % of them); "
          f"per role 8x8 8x4 4x8 4x4 16x16 16x8 8x16 intra: {' '.join(f'{x / 1e6:.0f}M' for x in per_wave)}")
    print(f"   issued rate {rate / 1e12:.2f} T abs-diff/s = {100 * rate / 148.4e12:.2f} % of the measured v_sad_u8 peak (148.4 T/s, profiles/r01_valu_rates.txt); "
          f"JM-equivalent rate {jm * (npic - 1) / dt / 1e12:.2f} T/s = {100 * jm * (npic - 1) / dt / 148.4e12:.2f} %")
    ctx.close()
    sys.exit(0)
ok = (st[:, 0] > 0) & (st[:, 17] > st[:, 0])
d = lambda a, b: np.median((st[ok, b] - st[ok, a]) / 100.0)        # microseconds
print(f"{mode}, {npic - 1} P pictures in one launch: macroblocks with stamps {ok.sum()}; median us: ticket->staged {d(0, 1):.1f}, staged->state in {d(1, 2):.1f}, ->wave 0's chain done {d(2, 8):.1f}, "
      f"->decided and coded {d(8, 16):.1f}, ->published {d(16, 17):.1f}; ticket->published {d(0, 17):.1f}; post stage: published->its start {d(17, 27):.1f}, waiting for the neighbours' post flags {d(27, 28):.1f}, "
      f"DeblockMb {d(28, 29):.1f}, planes {d(29, 30):.1f}; ticket->post done {d(0, 30):.1f}")
for a, b in ((2, 3), (3, 4), (4, 5), (5, 6), (6, 8), (8, 9), (8, 10), (8, 11), (8, 12), (8, 13), (8, 14), (8, 15), (8, 31), (31, 16)):
    print(f"   stamps {a}->{b}: median {d(a, b):.1f} us")
q = lambda a, b: (st[ok, b] - st[ok, a]) / 100.0
print(f"   the launch: {dt / (npic - 1) * 1e3:.3f} ms per picture with the stamps on ({nmb * (npic - 1) / dt / 1e6:.3f} M macroblocks/s)")
for name, a, b in (("ticket->staged (incl. the wait for the neighbours' vectors)", 0, 1), ("chain (2->6)", 2, 6), ("ticket->published", 0, 17), ("vectors out after the ticket (0->31)", 0, 31), ("vectors out after the chain's start (2->31)", 2, 31)):
    v = q(a, b)
    print(f"   {name}: mean {v.mean():.1f}, median {np.median(v):.1f}, p90 {np.percentile(v, 90):.1f}, p99 {np.percentile(v, 99):.1f} us")
ctx.close()
