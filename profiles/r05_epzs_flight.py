"""configs[2]'s search (EPZS, CABAC, 8x8 transform) at 1080p with sixteen pictures in flight: ms per picture (bench.py's configs2.in_flight leg alone).  python profiles/r05_epzs_flight.py [depth] [pictures]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from jm_amd import JmHip
from jm_amd.lib import SLICE_PARAMS, MB_RECORD
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 49
W, nmb, nslots = bench.W, 8160, 24
frames = bench.yuv_frames(8)
d_raw = torch.from_numpy(np.stack(frames)).cuda()
ctx = JmHip(W, bench.H, search_range=32, num_ref_slots=nslots, yuv_format=1)
ctx.seq_open(depth, 0, ready=True)
d_r2 = torch.zeros((nq, nmb * MB_RECORD.itemsize), dtype=torch.uint8, device="cuda")
def estep(k):
    st = 2 if k == 0 else 0
    q = bench.configs2_params(bench.slice_params(SLICE_PARAMS, st, 0, nmb, 0, 0 if k == 0 else 1), st, 2 * k)
    if k:
        q["ref_slot"][0, 0], q["ref_id"][0, 0], q["poc_ref"][0, 0] = (k - 1) % nslots, k - 1, 2 * (k - 1)
    ctx.seq_set_frame_dev(k % depth, d_raw[k % 8].data_ptr(), W, bench.H_SRC)
    ctx.seq_encode(k % depth, q, k % nslots, 1, False, d_r2[k].data_ptr())
estep(0); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(1, nq):
    estep(k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for e in range(depth):
    ctx.seq_wait(e)
print(f"JMHIP_DBG_SEQ={os.environ.get('JMHIP_DBG_SEQ', '0')} depth {depth}: {dt / (nq - 1) * 1e3:.2f} ms per picture ({nq - 1} pictures)")
