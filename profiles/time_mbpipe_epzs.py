"""Times jmhip_encode_slice with EPZS on BASELINE configs[2]'s search (g3e: 1080p, SR 32, five references configured, RDO off): I picture, then P pictures. gpu only."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench, tempfile
from test_gpu_mbenc import DevSeqEncoder, load_case
c = load_case("g3e")
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nref = int(sys.argv[2]) if len(sys.argv) > 2 else c["num_ref"]
with tempfile.TemporaryDirectory() as t:
    bench.write_yuv(os.path.join(t, "s.yuv"), nfr)
    data = np.fromfile(os.path.join(t, "s.yuv"), np.uint8)
fs = c["sw"] * c["sh"] * 3 // 2
enc = DevSeqEncoder(c["W"], c["H"], c["qp"], c["R"], nref, c["lam"], c["slice_mbs"], c["mv_limit"], c["didc"], cabac=c["cabac"], search_mode=3, epzs=c["epzs"])
for n in range(nfr):
    tm = []
    t0 = time.time()
    recs, pre, post = enc.encode(data[n * fs:(n + 1) * fs], c["sw"], c["sh"], timing=tm)
    t1 = time.time()
    types = np.bincount(recs["mb_type"].astype(int), minlength=11)
    print(f"picture {n}: kernel {tm} ms, whole call {1000*(t1-t0):.1f} ms, mb types {types.tolist()}", flush=True)
