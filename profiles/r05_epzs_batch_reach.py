"""Measurement aid (GPU box): EPZS in the one queue (jmhip_seq_batch, search_mode 3) -- with which queue lag / how few workgroups does a search reach past the queue's order
(JMHIP_EREACH), and is a launch that is NOT given up identical to the pictures coded one after another?   usage: python profiles/r05_epzs_batch_reach.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_seq as T
from test_gpu_mbenc import LAMBDAS, synthetic_clip, hard_clip

for (W, H, R, kind) in ((320, 192, 16, "synthetic"), (256, 160, 32, "stripes"), (640, 368, 32, "noise")):
    frames = synthetic_clip(W, H, 9, 5) if kind == "synthetic" else hard_clip(kind, W, H, 9, 77)
    want = T.classic(W, H, 28, R, 1, LAMBDAS, frames, search_mode=3)
    for lag in (None, 4, 6):
        for wg in (0, 1, 8):
            if lag is None:
                os.environ.pop("JMHIP_EPZS_BATCH_LAG", None)
            else:
                os.environ["JMHIP_EPZS_BATCH_LAG"] = str(lag)
            be = T.BatchEncoder(W, H, 28, R, 1, LAMBDAS, [8], 10, workgroups=wg, search_mode=3)
            try:
                got = be.run(frames, W, H)
                T.compare(want, got, (W, H, lag, wg))
                res = "identical"
            except T.pytest.fail.Exception as ex:
                res = "DIFFERENT " + repr(ex)[:200]
            except AssertionError as ex:
                res = "DIFFERENT " + repr(ex)[:200]
            except Exception as ex:
                res = "void: code %s" % getattr(ex, "code", "?") + (" " + repr(ex)[:120] if getattr(ex, "code", 0) != -6 else "")
            try:
                be.J.close()
            except Exception:
                pass
            print(f"{W}x{H} R {R} {kind:9s} lag {lag} workgroups {wg}: {res}", flush=True)
