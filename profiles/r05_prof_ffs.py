"""Where a macroblock's time goes in k_mb_pipe with the full search around every block's own centre (search_mode 0) and with the fast full search's one centre per macroblock and
reference (search_mode 1): JMHIP_MB_PROF time stamps, configs[1]'s P picture.  python profiles/r05_prof_ffs.py <prof mode> <search_mode>"""
import os, sys, ctypes as C
MODE = sys.argv[1] if len(sys.argv) > 1 else "1"
SM = int(sys.argv[2]) if len(sys.argv) > 2 else 1
os.environ["JMHIP_MB_PROF"] = MODE
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench, tempfile
from test_gpu_mbenc import DevSeqEncoder, load_case
c = load_case("g2r")
with tempfile.TemporaryDirectory() as t:
    bench.write_yuv(os.path.join(t, "s.yuv"), 2)
    data = np.fromfile(os.path.join(t, "s.yuv"), np.uint8)
fs = c["sw"] * c["sh"] * 3 // 2
enc = DevSeqEncoder(c["W"], c["H"], c["qp"], c["R"], c["num_ref"], c["lam"], c["slice_mbs"], c["mv_limit"], c["didc"], search_mode=SM)
nmb = 8160
for n in range(2):
    tm = []
    recs, pre, post = enc.encode(data[n * fs:(n + 1) * fs], c["sw"], c["sh"], timing=tm)
    st = np.zeros((nmb, 32), np.uint64)
    assert enc.J.lib.jmhip_debug_read_mb_prof(enc.J.h, st.ctypes.data_as(C.c_void_p), st.nbytes) == 0
    st = st.astype(np.int64)
    us = lambda a, b: (st[:, b] - st[:, a]) / 100.0
    if n == 0:
        continue
    print(f"search_mode {SM}, prof mode {MODE}: kernel {tm[0]:.1f} ms; per macroblock (median / mean us):")
    for name, a, b in [("ticket -> neighbours done + staged", 0, 1), ("phase 0", 2, 3), ("phase 1", 3, 4), ("phase 2", 4, 5), ("phase 3", 5, 6), ("decision + coding", 6, 16), ("after neighbours: edge .. publish", 1, 17)]:
        d = us(a, b)
        print(f"  {name:45s} {np.median(d):8.1f} {d.mean():8.1f}")
    print("  phase 0 per wave (waves 0-3: 8x8, 8x4, 4x8, 4x4 of block 0; 4-6: 16x16, 16x8, 8x16; 7: intra):", " ".join(f"{np.median((st[:, 8 + w] - st[:, 2]) / 100.0):7.1f}" for w in range(8)))
    if MODE in ("1", "5", "6", "7", "8", "9"):
        nm = {"1": "4x4", "5": "16x8", "6": "16x16", "7": "8x8", "8": "8x4", "9": "4x8"}[MODE]
        print("  first %s search: predictor %.2f, centre + block %.2f, row table %.2f, column loop %.2f, column 64 %.2f, wave minimum %.2f, half-pel %.2f, quarter-pel + skip %.2f us" % ((nm,) + tuple(
            np.median(us(a, b)) for a, b in ((18, 19), (19, 25), (25, 24), (24, 7), (7, 23), (23, 20), (20, 21), (21, 22)))))
        w = st[:, 26]
        print("  ... that search: window rows read by the sliding lanes: median %d (mean %.1f); candidates of step 2: median %d (mean %.1f); rows of step 1: median %d (mean %.1f)" % (
            np.median(w & 0xffff), (w & 0xffff).mean(), np.median((w >> 16) & 0xffff), ((w >> 16) & 0xffff).mean(), np.median(w >> 32), (w >> 32).mean()))
