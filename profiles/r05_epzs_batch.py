"""Measurement aid (GPU box): configs[2]'s P pictures (1080p, EPZS, CABAC, 8x8 transform, one reference) in ONE launch (jmhip_seq_batch, round 5): ms per picture against
the queue lag (JMHIP_EPZS_BATCH_LAG) and the launch's workgroups.   usage: python profiles/r05_epzs_batch.py [pictures]"""
import os
import sys
import time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from jm_amd import JmHip
from jm_amd.lib import SLICE_PARAMS, MB_RECORD
npic = int(sys.argv[1]) if len(sys.argv) > 1 else 49
W, H = 1920, 1088
nmb = (W // 16) * (H // 16)
frames = bench.yuv_frames(int(os.environ.get("FRAMES", min(npic, 44))))            # FRAMES=24: the clip of bench.py --steps 20 (its 64 P pictures turn round more often)
d_raw = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in frames]
d_rec = torch.zeros((npic, nmb * MB_RECORD.itemsize), dtype=torch.uint8, device="cuda")
nslots = 24


def pp(k):
    """the clip forwards, then backwards, ... (no scene cut where it would start again: bench.py's order)"""
    n = len(frames)
    m = k % (2 * (n - 1))
    return m if m < n else 2 * (n - 1) - m


def prm(st, nref, poc):
    return bench.configs2_params(bench.slice_params(SLICE_PARAMS, st, 0, nmb, 0, nref), st, poc)


first = None
REPS = int(os.environ.get('REPS', '2'))
NP = npic - int(os.environ.get('REFS', '1'))                   # P pictures of a launch
for lag, wg in [(int(a), 0) for a in os.environ['LAGS'].split(',')] if os.environ.get('LAGS') else ((0, 0), (8, 0), (10, 0), (16, 0), (20, 0), (26, 0), (0, 128), (0, 192), (0, 320)):
    if lag:
        os.environ["JMHIP_EPZS_BATCH_LAG"] = str(lag)
    else:
        os.environ.pop("JMHIP_EPZS_BATCH_LAG", None)
    ctx = JmHip(W, H, search_range=32, num_ref_slots=nslots, yuv_format=1)
    ctx.seq_open(1)
    ctx.seq_batch_reserve(npic - 1)
    ctx.set_pipeline_workgroups(wg)
    out, voids = [], 0
    NREF = int(os.environ.get("REFS", "1"))                      # REFS=5: configs[2] with five references (the first NREF pictures, which have fewer, launch by launch)
    for rep in range(REPS):
        for k in range(NREF):
            ctx.seq_set_frame_dev(0, d_raw[k].data_ptr(), 1920, 1080)
            qk = prm(2 if k == 0 else 0, k, 2 * k)
            for r in range(k):
                qk["ref_slot"][0, r], qk["ref_id"][0, r], qk["poc_ref"][0, r] = k - 1 - r, k - 1 - r, 2 * (k - 1 - r)
            ctx.seq_encode(0, qk, k, 1, False)
            ctx.seq_wait(0)
        q = prm(0, NREF, 2 * NREF)
        for r in range(NREF):
            q["ref_slot"][0, r], q["ref_id"][0, r], q["poc_ref"][0, r] = NREF - 1 - r, NREF - 1 - r, 2 * (NREF - 1 - r)
        pics = [dict(d_raw=d_raw[pp(k)].data_ptr(), src_w=1920, src_h=1080, out_slot=k % nslots, ref_slot=[(k - 1 - r) % nslots for r in range(NREF)],
                     ref_id=[k - 1 - r for r in range(NREF)], poc_offset=2 * (k - NREF), d_records=d_rec[k].data_ptr()) for k in range(NREF, npic)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.seq_batch(q, pics)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        try:
            ctx.synchronize()
            res = "ok"
            out.append(dt)
        except Exception as ex:
            res = "void (%s)" % getattr(ex, "code", "?")
            voids += 1
    same = ""
    if res == "ok":
        r = d_rec[1:].cpu()
        if first is None:
            first = r
        else:
            same = ", records equal the first run's" if torch.equal(first, r) else ", RECORDS DIFFER from the first run's"
    print(f"lag {lag or 'library'} workgroups {wg or 'library'}: {(min(out) if out else 0) / NP * 1e3:.3f} ms per picture ({NP} P pictures, the fastest of {REPS} launches; {voids} of them given up), last: {res}{same}", flush=True)
    ctx.close()
