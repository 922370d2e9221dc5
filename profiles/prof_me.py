#!/usr/bin/env python3
"""Runs only the full-search kernel on the bench workload (1080p, SR=32, 8160 window jobs x 41 partitions),
for rocprofv3 --kernel-trace / --pmc passes.  usage: prof_me.py [iters]
With --phases it first builds profiles/microbench/libjmhip_meprof.so (me_fast.hip with -DME_PROF; needs hipcc, run it where the
repository is writable) if that file is missing, and with JMHIP_LIB pointing at such a library prints the per-phase ticks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from jm_amd import JmHip
from jm_amd.lib import ME_JOB, ME_RESULT, NPART

iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
W, H, R = bench.W, bench.H, bench.R
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
ctx = JmHip(W, H, search_range=R, num_ref_slots=1, yuv_format=1, device=0, stream=stream.cuda_stream)
frames = bench.synth_luma(2)
ctx.set_reference(0, frames[0]); ctx.set_current(frames[1])
nmb = (W // 16) * (H // 16)
rng = np.random.default_rng(7)
jobs = np.zeros(nmb, ME_JOB)
jobs["mb_x"] = np.tile(np.arange(W // 16) * 16, H // 16); jobs["mb_y"] = np.repeat(np.arange(H // 16) * 16, W // 16)
jobs["search_range"], jobs["lambda"], jobs["part_mask"] = R, 187, np.uint64((1 << NPART) - 1)
jobs["pred"] = np.array([12, 8], np.int16) + rng.integers(-2, 3, (nmb, NPART, 2)).astype(np.int16)
jobs["center_x"], jobs["center_y"] = 12, 8
d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(nmb, -1)).to(dev)
d_res = torch.zeros((nmb, ME_RESULT.itemsize), dtype=torch.uint8, device=dev)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
for i in range(iters):
    ev[i][0].record(stream); ctx.me_fullsearch_dev(0, d_jobs.data_ptr(), nmb, d_res.data_ptr()); ev[i][1].record(stream)
torch.cuda.synchronize()
print("me_fullsearch ms:", [round(a.elapsed_time(b), 4) for a, b in ev])

import ctypes
lib = ctypes.CDLL(os.environ["JMHIP_LIB"], mode=ctypes.RTLD_GLOBAL) if os.environ.get("JMHIP_LIB") else None
if lib is not None and hasattr(lib, "jmhip_debug_read_me_prof"):
    out = np.zeros(8192 * 8 * 8, np.uint32)
    lib.jmhip_debug_read_me_prof(out.ctypes.data_as(ctypes.c_void_p))
    out = out.reshape(8192, 8, 8)[:nmb, :4].astype(np.float64)
    names = ["barrier after staging", "main pass (row pairs)", "65th column", "barrier (other waves)", "merge rounds", "final"]
    tot = out.sum(axis=2).mean()
    for i, n in enumerate(names):
        print("  %-24s mean %7.0f  min %7.0f  max %7.0f ticks  %5.1f %%   per wave: %s" % (n, out[:, :, i].mean(), out[:, :, i].min(), out[:, :, i].max(),
              100 * out[:, :, i].mean() / tot, " ".join("%6.0f" % v for v in out[:, :, i].mean(axis=0))))
