"""Where a macroblock's time goes inside k_mb_pipe_epzs (JMHIP_MB_PROF = 20 + block type: 100 MHz time stamps of that block type's first search on reference 0), g3e P picture.  gpu only."""
import os, sys, ctypes as C
BT = int(sys.argv[1]) if len(sys.argv) > 1 else 1
os.environ["JMHIP_MB_PROF"] = str(20 + BT)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench, tempfile
from test_gpu_mbenc import DevSeqEncoder, load_case
c = load_case("g3e")
with tempfile.TemporaryDirectory() as t:
    bench.write_yuv(os.path.join(t, "s.yuv"), 2)
    data = np.fromfile(os.path.join(t, "s.yuv"), np.uint8)
fs = c["sw"] * c["sh"] * 3 // 2
enc = DevSeqEncoder(c["W"], c["H"], c["qp"], c["R"], c["num_ref"], c["lam"], c["slice_mbs"], c["mv_limit"], c["didc"], cabac=c["cabac"], search_mode=3, epzs=c["epzs"])
nmb = 8160
for n in range(2):
    tm = []
    recs, pre, post = enc.encode(data[n * fs:(n + 1) * fs], c["sw"], c["sh"], timing=tm)
    if n == 0:
        continue
    st = np.zeros((nmb, 32), np.uint64)
    assert enc.J.lib.jmhip_debug_read_mb_prof(enc.J.h, st.ctypes.data_as(C.c_void_p), st.nbytes) == 0
    st = st.astype(np.int64)
    us = lambda a, b, m=None: ((st[:, b] - st[:, a]) / 100.0)[m if m is not None else slice(None)]
    print(f"P picture: kernel {tm[0]:.1f} ms; per macroblock (median / mean us):")
    for name, a, b in [("ticket -> neighbours' vectors + source staged", 0, 1), ("EPZS state import + tile staging", 1, 2), ("modes 1-3 (5 searches x refs)", 2, 3), ("P8x8 block 0", 3, 4), ("block 1", 4, 5), ("block 2", 5, 6), ("block 3", 6, 7),
                       ("export", 7, 8), ("wave 0 whole search phase", 2, 8), ("decision + coding", 8, 16), ("publish", 16, 17), ("whole macroblock", 0, 17), ("after neighbours", 1, 17)]:
        d = us(a, b)
        print(f"  {name:48s} {np.median(d):8.2f} {d.mean():8.2f}")
    full = st[:, 24] > 0                  # searches that went through the predictor list
    early = ~full
    print(f"  first search of block type {BT} on reference 0: {full.sum()} through the predictor list, {early.sum()} left before it")
    for name, a, b, m in [("neighbours, predictor, centre", 18, 19, None), ("centre SAD (one candidate)", 19, 20, None), ("predictor slots -> list", 20, 21, full), ("unique + visited map + compaction", 21, 22, full),
                          ("the list's SADs", 22, 23, full), ("costs, best two", 23, 24, full), ("pattern refinement", 24, 25, full), ("integer search whole", 19, 26, None), ("... when it ends early", 19, 26, early),
                          ("sub-pel search", 26, 9, None), ("skip vector cost (16x16)", 9, 10, None), ("whole search", 18, 10, None)]:
        d = us(a, b, m)
        if len(d):
            print(f"    {name:46s} {np.median(d):8.2f} {d.mean():8.2f}")
    print(f"    pattern steps per search (through the list): median {np.median(st[full, 13]):.0f} mean {st[full, 13].mean():.2f}; candidates evaluated / listed: mean {(st[full, 14] & 0xffff).mean():.1f} / {(st[full, 14] >> 16).mean():.1f}")
