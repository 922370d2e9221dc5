"""Times the macroblock pipeline on BASELINE configs[3]'s picture (3840x2160, SR 32, one reference, RDO off, QP 28), one MI355X:
the picture as ONE slice, and as 8 slices of 4050 macroblocks launched together (eight independent dependency chains side by side). gpu only."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synclip
from test_gpu_mbenc import DevSeqEncoder, load_case
c = load_case("g2r")                      # lambda tables, QP and vector limits of the same configuration
W, H, SH = 3840, 2160, 2160
with tempfile.TemporaryDirectory() as t:
    synclip.syn2160p(os.path.join(t, "s.yuv"), 3)
    data = np.fromfile(os.path.join(t, "s.yuv"), np.uint8)
fs = W * SH * 3 // 2
nmb = (W // 16) * (H // 16)
for slice_mbs, name in ((0, "one slice"), (nmb // 8, "8 slices of 4050 macroblocks, one launch")):
    enc = DevSeqEncoder(W, H, c["qp"], c["R"], 1, c["lam"], slice_mbs, c["mv_limit"], 0, together=True)
    for n in range(3):
        tm = []
        recs, pre, post = enc.encode(data[n * fs:(n + 1) * fs], W, SH, timing=tm)
        types = np.bincount(recs["mb_type"].astype(int), minlength=11)
        print(f"{name}: picture {n} ({'I' if n == 0 else 'P'}): kernel {tm[0]:.1f} ms = {nmb / tm[0] / 1e3:.2f} M macroblocks/s; mb types {types.tolist()}", flush=True)
    del enc
