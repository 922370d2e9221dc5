"""Where a macroblock's time goes inside k_mb_pipe (JMHIP_MB_PROF=1: 100 MHz time stamps per macroblock), configs[1] P picture.  gpu only."""
import os, sys, ctypes as C
MODE = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ["JMHIP_MB_PROF"] = sys.argv[1] if len(sys.argv) > 1 else "1"
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
enc = DevSeqEncoder(c["W"], c["H"], c["qp"], c["R"], c["num_ref"], c["lam"], c["slice_mbs"], c["mv_limit"], c["didc"])
nmb = 8160
for n in range(2):
    tm = []
    recs, pre, post = enc.encode(data[n * fs:(n + 1) * fs], c["sw"], c["sh"], timing=tm)
    st = np.zeros((nmb, 32), np.uint64)
    assert enc.J.lib.jmhip_debug_read_mb_prof(enc.J.h, st.ctypes.data_as(C.c_void_p), st.nbytes) == 0
    st = st.astype(np.int64)
    us = lambda a, b: (st[:, b] - st[:, a]) / 100.0
    print(f"picture {n}: kernel {tm[0]:.1f} ms; per macroblock (median / mean us):")
    for name, a, b in [("ticket -> neighbours done + staged", 0, 1), ("edge records", 1, 2), ("phase 0 (8x8 block 0 + modes 1-3 + intra)", 2, 3), ("phase 1", 3, 4), ("phase 2", 4, 5),
                       ("phase 3", 5, 6), ("decision + coding the winner", 6, 16), ("publish", 16, 17), ("whole macroblock", 0, 17), ("after neighbours: edge .. publish", 1, 17)]:
        d = us(a, b)
        print(f"  {name:45s} {np.median(d):8.1f} {d.mean():8.1f}")
    print("  phase 0 per wave (from phase start to the wave's end; waves 0-3: 8x8, 8x4, 4x8, 4x4 of block 0; 4-6: 16x16, 16x8, 8x16; 7: intra):")
    print("   ", " ".join(f"{np.median((st[:, 8 + w] - st[:, 2]) / 100.0):7.1f}" for w in range(8)))
    if MODE == "1" and n == 1:
        m = (st[:, 27] > 0) & (st[:, 30] > st[:, 27])
        f = lambda a, b: np.median((st[m, b] - st[m, a]) / 100.0)
        print("  final stage, inter luma (wave 0): decision %.2f, prediction fetch %.2f, transform/quant %.2f, thresholds + stores %.2f, rest up to the barrier %.2f us" % (f(6, 27), f(27, 28), f(28, 29), f(29, 30), f(30, 16)))
    if MODE == "1" and n == 1:
        print("  ... of the integer search: centre + source block %.2f, row table %.2f, column loop %.2f, column 64 %.2f, wave minimum %.2f us" % tuple(np.median(us(a, b)) for a, b in ((19, 25), (25, 24), (24, 7), (7, 23), (23, 20))))
    if MODE == "2":
        print("  intra wave (7): the neighbours' samples + the Intra4x4 chain %.1f us (median)" % np.median(us(2, 7)))
    if MODE == "3":
        d = (st[:, 19] - st[:, 18]) / np.maximum(us(0, 17), 1e-9)
        print("  shader clock while the kernel runs (s_memtime ticks per microsecond of s_memrealtime): median %.0f MHz" % np.median(d))
    if MODE == "2":
        print("  Intra4x4 block 5: neighbours + values %.2f, nine predictions + SATD + minimum %.2f, transform/quant/reconstruction %.2f, stores %.2f us (medians)" % tuple(
            np.median(us(a, b)) for a, b in ((18, 19), (19, 20), (20, 21), (21, 22))))
    if n == 1 and MODE in ("5", "6", "7", "8", "9"):
        print("  first %s search: predictor %.2f, centre + block %.2f, row table %.2f, column loop %.2f, column 64 %.2f, wave minimum %.2f, half-pel %.2f, quarter-pel + skip %.2f us" % (({"5": "16x8", "6": "16x16", "7": "8x8", "8": "8x4", "9": "4x8"}[MODE],) + tuple(
            np.median(us(a, b)) for a, b in ((18, 19), (19, 25), (25, 24), (24, 7), (7, 23), (23, 20), (20, 21), (21, 22)))))
    if n == 1 and MODE in ("1", "5", "6", "7", "8", "9"):
        w = st[:, 26]
        print("  ... that search: window rows read by the sliding lanes: median %d (mean %.1f); candidates of step 2: median %d (mean %.1f); rows of step 1: median %d (mean %.1f)" % (
            np.median(w & 0xffff), (w & 0xffff).mean(), np.median((w >> 16) & 0xffff), ((w >> 16) & 0xffff).mean(), np.median(w >> 32), (w >> 32).mean()))
    if n == 1 and MODE == "11":                                # what makes a first 8x8 block slow: rows read by the sliding lanes / candidates of step 2, per wave
        ph = us(2, 3)
        slow, fast = ph >= np.percentile(ph, 90), ph <= np.percentile(ph, 50)
        for w, nm in enumerate(("8x8", "8x4", "4x8", "4x4")):
            v = st[:, 18 + w]; rows, items = v & 0xffffffff, v >> 32
            tw = (st[:, 8 + w] - st[:, 2]) / 100.0
            print("  wave %s: time fast half %.1f / slow tenth %.1f us; rows read %.0f / %.0f; step-2 candidates %.0f / %.0f" % (nm, tw[fast].mean(), tw[slow].mean(), rows[fast].mean(), rows[slow].mean(), items[fast].mean(), items[slow].mean()))
    if n == 1 and MODE == "10":
        print("  first 4x4 search, the sliding part in detail: first rows %.2f, their minimum + the sub-pel fetch started %.2f, what remains decided %.2f, steps 1 %.2f, B1 + keys widened %.2f; step 2 %.2f, wave minimum %.2f us" % tuple(
            np.median(us(a, b)) for a, b in ((24, 27), (27, 28), (28, 29), (29, 30), (30, 7), (7, 23), (23, 20))))
    if n == 1 and MODE in ("1", "10"):
        d = us(18, 22)
        print("  first 4x4 search, whole: percentiles 10/50/90/99/max " + " ".join("%.2f" % np.percentile(d, q) for q in (10, 50, 90, 99, 100)) + "; its integer part: " + " ".join("%.2f" % np.percentile(us(19, 20), q) for q in (10, 50, 90, 99, 100)))
        print("  first 4x4 search of the macroblock (wave 3): predictor %.2f, integer search %.2f, half-pel stage %.2f, quarter-pel stage + clip %.2f us (medians)" % tuple(
            np.median(us(a, b)) for a, b in ((18, 19), (19, 20), (20, 21), (21, 22))))
    if n == 1:                                                 # the hand-over: from the last neighbour's flag store to this macroblock's start
        wmb = c["W"] // 16
        PUB = 31 if st[:, 31].any() else 17                   # the stamp after which the neighbours' searches may start (the vectors' flag)
        t17 = st[:, PUB].reshape(-1, wmb); t1 = st[:, 1].reshape(-1, wmb); t0 = st[:, 0].reshape(-1, wmb)
        last = np.zeros_like(t17)
        last[:, 1:] = np.maximum(last[:, 1:], t17[:, :-1]); last[1:, :] = np.maximum(last[1:, :], t17[:-1, :]); last[1:, :-1] = np.maximum(last[1:, :-1], t17[:-1, 1:])
        ho = (t1 - last)[last > 0] / 100.0
        early = (t0 < last)[last > 0]
        print("  hand-over (last neighbour's flag store -> start, macroblocks whose ticket was taken before that store: %.0f %%): median %.2f, mean %.2f, 90th percentile %.2f us" % (
            100.0 * early.mean(), np.median(ho[early]), ho[early].mean(), np.percentile(ho[early], 90)))
    if n == 1:
        d = us(1, 17)
        print("  edge .. publish percentiles 10/50/90/99/max: " + " ".join("%.1f" % np.percentile(d, q) for q in (10, 50, 90, 99, 100)))
        for nm, a, b in (("phase 0", 2, 3), ("phase 1", 3, 4), ("phase 2", 4, 5), ("phase 3", 5, 6), ("wait for the free waves", 6, 6), ("decision + coding", 6, 16)):
            if a != b: print("    %-26s 10/50/90/99: " % nm + " ".join("%.1f" % np.percentile(us(a, b), q) for q in (10, 50, 90, 99)))
        types = recs["mb_type"] if "mb_type" in recs.dtype.names else None
        if types is not None:
            for t in np.unique(types): print("    mb_type %2d: %5d macroblocks, median %.1f us" % (t, (types == t).sum(), np.median(d[types == t])))
    if n == 1:                                                 # the critical path: from the last macroblock back through whichever neighbour finished last
        hmb = c["H"] // 16
        x, y = wmb - 1, hmb - 1
        path = []
        while True:
            path.append(y * wmb + x)
            cand = [(t17[yy, xx], xx, yy) for xx, yy in ((x - 1, y), (x, y - 1), (x + 1, y - 1), (x - 1, y - 1)) if 0 <= xx < wmb and yy >= 0]
            if not cand: break
            _, x, y = max(cand)
        path = np.array(path)
        dp = us(1, PUB)[path]
        print("  critical path: %d macroblocks, start .. vectors published on it: median %.1f, mean %.1f us (all macroblocks: mean %.1f); sum %.2f ms" % (len(path), np.median(dp), dp.mean(), us(1, PUB).mean(), dp.sum() / 1000))
        for nm, a, b in (("phase 0", 2, 3), ("phase 1", 3, 4), ("phase 2", 4, 5), ("phase 3", 5, 6), ("final barrier .. vectors out", 6, PUB), ("final barrier .. end", 6, 16), ("publish", 16, 17)):
            print("    on the path, %-22s mean %.1f (all: %.1f)" % (nm, us(a, b)[path].mean(), us(a, b).mean()))
        print("    on the path, first 8x8 block, waves 0-3 (8x8, 8x4, 4x8, 4x4) from phase start to the wave's end: " + " ".join("%.1f" % ((st[path, 8 + w] - st[path, 2]) / 100.0).mean() for w in range(4)) +
              "  (all: " + " ".join("%.1f" % ((st[:, 8 + w] - st[:, 2]) / 100.0).mean() for w in range(4)) + ")")
        wl = np.argmax(np.stack([st[path, 8 + w] for w in range(4)]), 0)
        print("    ... the last of the four to finish, share of the path's macroblocks: " + " ".join("%s %.0f %%" % (nm, 100.0 * (wl == w).mean()) for w, nm in enumerate(("8x8", "8x4", "4x8", "4x4"))))
    span = (st[:, 17].max() - st[:, 0].min()) / 100.0
    print(f"  first ticket -> last publish {span / 1000:.2f} ms; sum of per-macroblock busy time / span = {us(1, 17).sum() / span:.1f} macroblocks in flight on average")
