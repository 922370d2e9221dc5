#!/usr/bin/env python3
"""bench.py -- encoded macroblocks/sec of the MI355X hot path on synthetic 1080p (BASELINE.json configs[1]).

One "step" = the next P picture of ONE IPPP sequence of configs[1] (1920x1080 coded as 1920x1088 = 8160 macroblocks, Baseline, FullSearch SR = 32, one reference = the
picture before, QP 28, RDOptimization = 0 / AdaptiveRounding = 0: the configuration whose output the tests prove bit-identical to CPU JM), inputs already in HBM:
  jmhip_seq_set_frame_dev    read_one_frame + pad_borders from the file's bytes (k_load_frame)
  jmhip_seq_encode           ONE launch of k_mb_pipe: encode_one_macroblock_low of every macroblock on the x + 2y wavefront -- MV prediction, 41 full searches at each block's own
                             centre, sub-pel refinement, mode decision, transform / quantisation / reconstruction, the record JM's entropy coder reads -- and, behind each
                             macroblock, its DeblockMb and its share of getSubImagesLuma's sixteen planes (jm_amd/csrc/mbpipe_post.inc)
Up to --flight (8) consecutive pictures are in flight side by side: a macroblock of picture n + 1 starts as soon as picture n is filtered and interpolated five macroblocks to its
right and below.  The timed region starts and ends with an idle device.  After it every picture's records are compared with the REAL reference encoder's
(tests/golden/mb_low_g6r.npz, six pictures) and with the same sequence coded picture after picture: `records_equal_jm` and `records_equal_picture_after_picture` must be true or
the number is void.
`end_to_end` is the unmodified lencod with this path linked in (oracle/_ref/lencod_hip.exe) on the same clip and flags, next to CPU JM
(`cpu_baseline`, oracle/_ref/lencod.exe, one thread) -- both print their own per-picture times; the .264 md5s are compared.

python bench.py --gpus N --steps K --warmup W     (N > 1: launched by torch.distributed.run, one rank per GPU)
N > 1: the same step on every GPU, one closed GOP (I + P pictures) per GPU, no collective on the data path: weak scaling, `value` = all GPUs' macroblocks / the slowest
rank's time.  BASELINE configs[3] with RDO off -- one 2160p picture, 8 slices dealt to the N GPUs with one RCCL all-gather per picture (multi_gpu_configs3 below: strong scaling
of one picture) -- is measured in the same run (`configs3_slice_split`).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")         # before the HIP runtime starts: a hardware queue per stream (the pictures in flight; the concurrent_streams figure); the default is 4

W, H_SRC, H = 1920, 1080, 1088
R = 32
QP = 28
G2R_FLAGS = ("InputFile=syn1080p.yuv", "SourceWidth=1920", "SourceHeight=1080", "OutputWidth=1920", "OutputHeight=1080", "SearchMode=-1", "SearchRange=32",
             "NumberReferenceFrames=1", "LevelIDC=51", "RDOptimization=0", "AdaptiveRounding=0", "OutputFile=o.264", "ReconFile=o_rec.yuv", "TraceFile=/dev/null")
G2R_MD5 = "04ce4cdee722defe8c3c7c0b249eda7e"      # CPU JM's .264 for two frames of the clip with these flags
# HBM-side bytes of k_mb_pipe PER PICTURE: 2 x FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes of this command (--steps 20 --warmup 5), corrected as
# MI355X_MICROARCH.md prescribes (profiles/collect5.sh, profiles/r04_v7_kernel_stats.md).  The counters sit between the L2s and the fabric: they also count what the 256 MB
# Infinity Cache serves.  Most of it is by design: every reference sample is read with an sc1 load (past the XCD's L2 -- another XCD may have written it microseconds ago), 670 MB
# of reads per picture; WRITE_SIZE counts 32-byte sectors (a 16-byte write-through store counts twice: profiles/r04_write_calib.txt).  DESIGN.md section 0a.
PIPE_TRAFFIC_BYTES = {"batch": 810710640, "pictures": 464000000}           # per picture: one launch for all timed pictures (r06_final) / a launch per picture, eight in flight (r04_v1); the fallback when traffic_live cannot run
# absolute differences the integer searches of one P picture of the clip ISSUE (the kernel's own counters, JMHIP_MB_PROF=11: profiles/batch_prof.py 25 fs 11): 81.5 % of the 61.78 G
# JM's full search visits -- the cost bound of me_fullsearch.c:83 skips the rest for the small blocks
VALU_ISSUED_PER_PICTURE = 50.366e9
VALU_ISSUED_SOURCE = "profiles/r04_valu_issued.txt (the kernel's own counters over one picture of this clip; not re-measured in this run)"
PIPE_TRAFFIC_SOURCE = "profiles/r06_final_kernel_stats.md (--launch batch) / profiles/r04_v1_kernel_stats.md (--launch pictures): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command, per picture x the pictures of a launch; not re-measured in this run"


def synth_luma(n_frames, seed=1234):
    """SURVEY.md Appendix A generator (luma only), coded height 1088 by replicating the last row."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(H_SRC // 8 + 8, W // 8 + 8)).astype(np.float32)
    base = np.kron(base, np.ones((8, 8), np.float32))
    k = 5
    b = np.cumsum(np.cumsum(np.pad(base, ((k, k), (k, k)), mode="edge"), 0), 1)
    sm = (b[2 * k:, 2 * k:] - b[:-2 * k, 2 * k:] - b[2 * k:, :-2 * k] + b[:-2 * k, :-2 * k]) / (4 * k * k)
    sm = sm[:H_SRC + 64, :W + 64]
    frames = []
    for n in range(n_frames):
        dx, dy = 3 * n, 2 * n
        y = sm[dy:dy + H_SRC, dx:dx + W] + rng.normal(0, 2, size=(H_SRC, W))
        y = np.clip(np.rint(y), 0, 255).astype(np.uint8)
        frames.append(np.concatenate([y, np.repeat(y[-1:], H - H_SRC, 0)], 0))
    return frames


def write_yuv(path, n_frames, seed=1234):
    """the same clip as 4:2:0 YUV for lencod (Appendix A incl. chroma)."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(H_SRC // 8 + 8, W // 8 + 8)).astype(np.float32)
    base = np.kron(base, np.ones((8, 8), np.float32))
    k = 5
    b = np.cumsum(np.cumsum(np.pad(base, ((k, k), (k, k)), mode="edge"), 0), 1)
    sm = (b[2 * k:, 2 * k:] - b[:-2 * k, 2 * k:] - b[2 * k:, :-2 * k] + b[:-2 * k, :-2 * k]) / (4 * k * k)
    sm = sm[:H_SRC + 64, :W + 64]
    with open(path, "wb") as f:
        for n in range(n_frames):
            m = n % 42 if n % 42 <= 21 else 42 - n % 42            # Appendix A's pan for the first 22 frames (all that G2 / G4 and the goldens use); longer clips pan back and forth
            dx, dy = 3 * m, 2 * m
            y = sm[dy:dy + H_SRC, dx:dx + W] + rng.normal(0, 2, size=(H_SRC, W))
            y = np.clip(np.rint(y), 0, 255).astype(np.uint8)
            u = np.clip(np.rint(128 + 0.25 * (y[::2, ::2].astype(np.float32) - 128)), 0, 255).astype(np.uint8)
            v = np.clip(np.rint(128 - 0.25 * (y[::2, ::2].astype(np.float32) - 128)), 0, 255).astype(np.uint8)
            f.write(y.tobytes()); f.write(u.tobytes()); f.write(v.tobytes())



def yuv_frames(n_frames, seed=1234):
    """the clip of write_yuv as per-frame byte arrays (planar 4:2:0 at 1920x1080)"""
    with tempfile.TemporaryDirectory() as tmp:
        write_yuv(os.path.join(tmp, "s.yuv"), n_frames, seed)
        data = np.fromfile(os.path.join(tmp, "s.yuv"), np.uint8)
    fs = W * H_SRC * 3 // 2
    return [data[k * fs:(k + 1) * fs].copy() for k in range(n_frames)]


G3E_FLAGS = ("InputFile=syn1080p.yuv", "SourceWidth=1920", "SourceHeight=1080", "OutputWidth=1920", "OutputHeight=1080", "SearchMode=3", "SearchRange=32",
             "NumberReferenceFrames=5", "LevelIDC=51", "RDOptimization=0", "AdaptiveRounding=0", "SymbolMode=1", "ProfileIDC=100", "Transform8x8Mode=1", "OutputFile=o.264",
             "ReconFile=o_rec.yuv", "TraceFile=/dev/null")        # tests/golden/mb_low_g3h.npz: BASELINE configs[2] as stated (CABAC, 8x8 transform on, EPZS), P pictures only


def run_lencod(exe, frames, timeout, flags=None, clip=None, cfg_name="jm_baseline.cfg"):
    """(per-frame {type: [ms]}, md5 of the .264, adapter report line or None, wall seconds) of one encoder run on the clip with the G2r flags (or `flags`)"""
    import hashlib
    import shutil
    cfg = os.path.join(ROOT, "tests", "golden", cfg_name)
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copyfile(os.path.join(ROOT, "tests", "golden", "q_offset.cfg"), os.path.join(tmp, "q_offset.cfg"))      # read when a .cfg sets OffsetMatrixPresentFlag
        if clip:
            clip(tmp)
        else:
            write_yuv(os.path.join(tmp, "syn1080p.yuv"), frames)
        args = [exe, "-d", cfg]
        for kv in (flags or G2R_FLAGS) + (f"FramesToBeEncoded={frames}",):
            args += ["-p", kv]
        t0 = time.time()
        # (lencod_hip.exe: leave without the teardown once the files are closed -- a quarter of a two-picture run, profiles/r05_init_prof.txt; opt-in since round 6)
        r = subprocess.run(args, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=dict(os.environ, JMHIP_ADAPTER_FAST_EXIT="1"))
        wall = time.time() - t0
        if r.returncode != 0:
            return None
        out, err = r.stdout.decode(errors="replace"), r.stderr.decode(errors="replace")
        times = {}
        for m in re.finditer(r"^\s*\d+\(\s*(IDR|I|P|B)\s*\)\s+\d+\s+\d+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+(\d+)\s+(\d+)", out, re.M):
            times.setdefault(m.group(1) if m.group(1) in ("P", "B") else "I", []).append(int(m.group(2)))
        rep = re.search(r"jmhip adapter: macroblock pipeline: .*", err)
        return times, hashlib.md5(open(os.path.join(tmp, "o.264"), "rb").read()).hexdigest(), rep.group(0) if rep else None, wall


def cpu_baseline(max_seconds=120):
    """JM's own CPU lencod (oracle/_ref/lencod.exe, built from the reference: kind "reference"), single thread, on a bounded sample of the same
    workload with the same flags as the device path (so that both write the same bitstream): the first two pictures (I + P) of the clip.
    Falls back to the oracle's C restatement of the full search (kind "port") on a sample of macroblocks when the reference binary did not travel."""
    exe = os.path.join(ROOT, "oracle", "_ref", "lencod.exe")
    if os.path.exists(exe) and os.access(exe, os.X_OK):
        try:
            r = run_lencod(exe, 2, max_seconds)
            r2 = run_lencod(exe, 2, max_seconds)          # the wall time as end_to_end takes lencod_hip.exe's: the shorter of two runs
            if r and r2 and r2[3] < r[3]:
                r = r2
        except subprocess.TimeoutExpired:
            r = None
        if r and r[0].get("P"):
            times, md5, _, wall = r
            p_ms = times["P"][0]
            return {"value": round(8160 / (p_ms / 1000.0), 1), "unit": "macroblocks/s", "cores": 1, "kind": "reference", "p_frame_ms": p_ms, "wall_s_two_pictures": round(wall, 2),
                    "md5_264": md5, "md5_is_g2r": md5 == G2R_MD5,
                    "sample": f"JM 19.0 lencod -O3, 1 thread, same flags as the device path (FullSearch SR=32, 1 ref, RDOptimization=0, CAVLC): "
                              f"P picture of syn1080p (I + P encoded, {wall:.1f} s wall): {p_ms} ms"}
    from oracle import pyjmo as J
    from jm_amd.lib import PARTITIONS
    frames = synth_luma(2)
    ref, cur = J.RefPic(frames[0]), frames[1]
    t0, n = time.time(), 0
    for mby in range(256, 256 + 16 * 2, 16):
        for mbx in range(256, 256 + 16 * 6, 16):
            for (bt, bx, by, w, h) in PARTITIONS:
                J.full_search(ref, cur, mbx + bx, mby + by, w, h, (12, 8), (12, 8), R, 187)
            n += 1
    dt = time.time() - t0
    return {"value": round(n / dt, 1), "unit": "macroblocks/s", "cores": 1, "kind": "port",
            "sample": f"oracle jmo_full_search (C, -O2), integer-pel ME only, {n} macroblocks x 41 partitions, SR=32"}


def end_to_end(cpu, max_seconds=300):
    """The unmodified encoder with the device path behind encode_one_macroblock_low (oracle/_ref/lencod_hip.exe): picture times as lencod prints them."""
    exe = os.path.join(ROOT, "oracle", "_ref", "lencod_hip.exe")
    if not (os.path.exists(exe) and os.access(exe, os.X_OK)):
        return {"available": False, "why": "oracle/_ref/lencod_hip.exe did not travel (it is built from the reference's sources where /root/reference exists)"}
    try:
        two = run_lencod(exe, 2, max_seconds)            # the golden md5 is of two pictures
        again = run_lencod(exe, 2, max_seconds)          # (its wall time once more: a process started while the driver still clears the previous one away waits for that, profiles/r05_init_prof.txt)
        if two and again and again[3] < two[3]:
            two = again
        more = run_lencod(exe, 6, max_seconds)           # steady state: later P pictures no longer pay first-launch costs
    except subprocess.TimeoutExpired:
        return {"available": False, "why": "timeout"}
    if not two or not more or not more[0].get("P"):
        return {"available": False, "why": "lencod_hip.exe failed"}
    p = more[0]["P"]
    p_ms = float(np.median(p[1:])) if len(p) > 1 else float(p[0])
    out = {"available": True, "p_frame_ms": p_ms, "p_frame_ms_all": p, "i_frame_ms": more[0].get("I", [None])[0], "macroblocks_per_s": round(8160 / (p_ms / 1000.0), 1),
           "md5_264_two_frames": two[1], "md5_ok": two[1] == G2R_MD5, "adapter": more[2],
           # whole runs, process start to exit (context creation, first launches, the I picture, file I/O): what a two-picture job really gains
           "sequence_wall_s": {"pictures_2_hip": round(two[3], 2), "pictures_2_hip_is": "the shorter of two runs (CPU JM's too)", "pictures_2_cpu_jm": cpu.get("wall_s_two_pictures") if cpu else None, "pictures_6_hip": round(more[3], 2),
                               "speedup_2_pictures": round(cpu["wall_s_two_pictures"] / two[3], 1) if cpu and cpu.get("wall_s_two_pictures") else None,
                               "note": "p_frame_ms is the median of the later P pictures of the six-picture run; the first P picture and the I picture also pay first-launch costs (p_frame_ms_all, i_frame_ms)"},
           "config": "lencod_hip.exe -d jm_baseline.cfg " + " ".join("-p " + f for f in G2R_FLAGS[:11]) + ": unmodified JM 19.0 host code, entropy coding on the host, one thread"}
    if cpu and cpu.get("kind") == "reference":
        out["speedup_vs_cpu_jm_p_frame"] = round(cpu["p_frame_ms"] / p_ms, 2)
    return out


def configs2_end_to_end(max_seconds=300):
    """BASELINE configs[2]'s search (1080p, Main profile, CABAC, EPZS with the shipped switches, RDO off, P pictures, 4x4 transform): CPU JM and the
    drop-in encoder side by side on three pictures; every EPZS search runs inside k_mb_pipe_epzs (no per-candidate calls)."""
    cpu_exe, hip_exe = os.path.join(ROOT, "oracle", "_ref", "lencod.exe"), os.path.join(ROOT, "oracle", "_ref", "lencod_hip.exe")
    if not all(os.path.exists(e) and os.access(e, os.X_OK) for e in (cpu_exe, hip_exe)):
        return {"available": False, "why": "oracle/_ref/lencod.exe / lencod_hip.exe did not travel"}
    try:
        c = run_lencod(cpu_exe, 3, max_seconds, G3E_FLAGS)
        h = run_lencod(hip_exe, 3, max_seconds, G3E_FLAGS)
    except subprocess.TimeoutExpired:
        return {"available": False, "why": "timeout"}
    if not c or not h:
        return {"available": False, "why": "an encoder failed"}
    gold = str(np.load(os.path.join(ROOT, "tests", "golden", "mb_low_g3h.npz"))["md5_264"])
    out = {"available": True, "p_frame_ms_cpu_jm": c[0].get("P"), "p_frame_ms_hip": h[0].get("P"), "wall_s_cpu_jm": round(c[3], 2), "wall_s_hip": round(h[3], 2),
           "speedup_p_frames": round(sum(c[0]["P"]) / max(1, sum(h[0]["P"])), 2), "md5_equal": c[1] == h[1], "md5_is_g3h": h[1] == gold, "adapter": h[2],
           "config": "lencod -d jm_baseline.cfg " + " ".join("-p " + f for f in G3E_FLAGS[:14]) + " -p FramesToBeEncoded=3"}
    # three pictures are the committed golden's, but too few for the pictures in flight to show (the second P picture is the first one launched ahead): nine pictures, both encoders
    try:
        c9 = run_lencod(cpu_exe, 9, max_seconds, G3E_FLAGS)
        h9 = run_lencod(hip_exe, 9, max_seconds, G3E_FLAGS)
        if c9 and h9:
            cp, hp = c9[0]["P"][2:], h9[0]["P"][2:]                  # the later P pictures: five references, the pipeline filled
            out["nine_pictures"] = {"p_frame_ms_cpu_jm": c9[0]["P"], "p_frame_ms_hip": h9[0]["P"], "speedup_later_p_frames": round(sum(cp) / max(1, sum(hp)), 2),
                                    "wall_s_cpu_jm": round(c9[3], 2), "wall_s_hip": round(h9[3], 2), "md5_equal": c9[1] == h9[1]}
    except subprocess.TimeoutExpired:
        out["nine_pictures"] = {"available": False, "why": "timeout"}
    return out


def configs4_end_to_end(max_seconds=400):
    """BASELINE configs[4] at its own size with RDO off and P pictures only (tests/golden/mb_low_g4y.npz): encoder_yuv422.cfg -- High 4:2:2, CABAC, 8x8 transform on, fast full
    search SR 32, five references configured, q_offset.cfg -- on 1920x1080 4:2:2 synthetic input, I + 2 P pictures; CPU JM and the drop-in encoder side by side."""
    cpu_exe, hip_exe = os.path.join(ROOT, "oracle", "_ref", "lencod.exe"), os.path.join(ROOT, "oracle", "_ref", "lencod_hip.exe")
    if not all(os.path.exists(e) and os.access(e, os.X_OK) for e in (cpu_exe, hip_exe)):
        return {"available": False, "why": "oracle/_ref/lencod.exe / lencod_hip.exe did not travel"}
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synclip
    z = np.load(os.path.join(ROOT, "tests", "golden", "mb_low_g4y.npz"))
    flags = tuple(str(f) for f in z["overrides"] if not str(f).startswith("FramesToBeEncoded")) + ("OutputFile=o.264", "ReconFile=o_rec.yuv", "TraceFile=/dev/null")
    clip = lambda tmp: synclip.syn1080p422(os.path.join(tmp, "syn1080p422.yuv"), 3)
    try:
        c = run_lencod(cpu_exe, 3, max_seconds, flags, clip=clip, cfg_name=str(z["cfg"]))
        h = run_lencod(hip_exe, 3, max_seconds, flags, clip=clip, cfg_name=str(z["cfg"]))
    except subprocess.TimeoutExpired:
        return {"available": False, "why": "timeout"}
    if not c or not h:
        return {"available": False, "why": "an encoder failed"}
    return {"available": True, "workload": "configs[4]: 1080p 4:2:2 (High 4:2:2 profile), CABAC, 8x8 transform on, fast full search SR 32, five references configured (1 and 2 exist), RDO off, "
                                           "P pictures only, q_offset.cfg's quantiser offsets", "kernel": "k_mb_pipe_t8",
            "p_frame_ms_cpu_jm": c[0].get("P"), "p_frame_ms_hip": h[0].get("P"), "wall_s_cpu_jm": round(c[3], 2), "wall_s_hip": round(h[3], 2),
            "speedup_p_frames": round(sum(c[0]["P"]) / max(1, sum(h[0]["P"])), 2), "md5_equal": c[1] == h[1], "md5_is_g4y": h[1] == str(z["md5_264"]), "adapter": h[2]}


def b_pictures_end_to_end(max_seconds=400):
    """encoder_main.cfg with RDO off at 1080p through the drop-in encoder (tests/golden/mb_low_g3b.npz: I P B, the .264's md5 is CPU JM's), CPU JM beside it on the same three
    pictures; then nine pictures (I P B P B ...) for the later pictures' times -- the adapter launches P and B pictures ahead of time, the B pictures beside the P pictures after them."""
    cpu_exe, hip_exe = os.path.join(ROOT, "oracle", "_ref", "lencod.exe"), os.path.join(ROOT, "oracle", "_ref", "lencod_hip.exe")
    if not all(os.path.exists(e) and os.access(e, os.X_OK) for e in (cpu_exe, hip_exe)):
        return {"available": False, "why": "oracle/_ref/lencod.exe / lencod_hip.exe did not travel"}
    z = np.load(os.path.join(ROOT, "tests", "golden", "mb_low_g3b.npz"))
    flags = tuple(str(f) for f in z["overrides"] if not str(f).startswith("FramesToBeEncoded")) + ("OutputFile=o.264", "ReconFile=o_rec.yuv", "TraceFile=/dev/null")
    try:
        c = run_lencod(cpu_exe, 3, max_seconds, flags, cfg_name=str(z["cfg"]))
        h = run_lencod(hip_exe, 3, max_seconds, flags, cfg_name=str(z["cfg"]))
        h9 = run_lencod(hip_exe, 9, max_seconds, flags, cfg_name=str(z["cfg"]))
    except subprocess.TimeoutExpired:
        return {"available": False, "why": "timeout"}
    if not c or not h or not h9:
        return {"available": False, "why": "an encoder failed"}
    return {"available": True, "p_frame_ms_cpu_jm": c[0].get("P"), "b_frame_ms_cpu_jm": c[0].get("B"), "p_frame_ms_hip": h[0].get("P"), "b_frame_ms_hip": h[0].get("B"),
            "wall_s_cpu_jm": round(c[3], 2), "wall_s_hip": round(h[3], 2), "md5_equal": c[1] == h[1], "md5_is_g3b": h[1] == str(z["md5_264"]),
            "speedup_b_frame": round(sum(c[0].get("B", [0])) / max(1, sum(h[0].get("B", [1]))), 2),
            "nine_pictures": {"p_frame_ms_hip": h9[0].get("P"), "b_frame_ms_hip": h9[0].get("B"), "wall_s_hip": round(h9[3], 2)}, "adapter": h9[2]}


def configs3_end_to_end(max_seconds=400):
    """BASELINE configs[3] at its own size with RDO off (G4r: 3840x2160, 8 slices of 4080 macroblocks, FullSearch SR 32, one reference): the drop-in encoder on two
    pictures -- the eight slices' wavefronts run side by side in one launch on ONE GPU; the md5 is CPU JM's (tests/golden/md5.json)."""
    hip_exe = os.path.join(ROOT, "oracle", "_ref", "lencod_hip.exe")
    if not (os.path.exists(hip_exe) and os.access(hip_exe, os.X_OK)):
        return {"available": False, "why": "oracle/_ref/lencod_hip.exe did not travel"}
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synclip
    e = json.load(open(os.path.join(ROOT, "tests", "golden", "md5.json")))["G4r"]
    flags = tuple(f"{k}={v}" for k, v in e["overrides"].items() if k != "FramesToBeEncoded") + ("OutputFile=o.264", "ReconFile=o_rec.yuv", "TraceFile=/dev/null")
    try:
        h = run_lencod(hip_exe, int(e["overrides"].get("FramesToBeEncoded", 2)), max_seconds, flags, clip=lambda tmp: synclip.syn2160p(os.path.join(tmp, "syn2160p.yuv")))
    except subprocess.TimeoutExpired:
        return {"available": False, "why": "timeout"}
    if not h:
        return {"available": False, "why": "lencod_hip.exe failed"}
    out = {"available": True, "macroblocks_per_picture": 32400, "slices": 8, "p_frame_ms_hip": h[0].get("P"), "i_frame_ms_hip": h[0].get("I"), "wall_s_hip": round(h[3], 2),
           "macroblocks_per_s_p_frame": round(32400 / (h[0]["P"][0] / 1000.0), 1) if h[0].get("P") else None, "md5_is_g4r": h[1] == e["md5_264"], "adapter": h[2]}
    # ... and eight pictures of the same sequence: the adapter launches the next pictures ahead of time (SliceMode 1 pictures too since round 4: one launch per picture in the
    # picture's wavefront order), JM entropy-codes picture k while the device is pictures ahead; no CPU JM beside it (a minute per picture)
    try:
        h8 = run_lencod(hip_exe, 8, max_seconds, flags, clip=lambda tmp: synclip.syn2160p(os.path.join(tmp, "syn2160p.yuv"), 8))
        if h8 and h8[0].get("P"):
            later = sorted(h8[0]["P"][1:])
            out["eight_pictures"] = {"p_frame_ms_hip": h8[0]["P"], "p_frame_ms_later_median": later[len(later) // 2], "wall_s_hip": round(h8[3], 2),
                                     "macroblocks_per_s_later_p_frames": round(32400 / (later[len(later) // 2] / 1000.0), 1)}
    except subprocess.TimeoutExpired:
        pass
    return out


def configs3_device(device):
    """BASELINE configs[3]'s P picture on ONE GPU, device-resident: 3840x2160, 8 slices of 4080 macroblocks side by side in one launch of k_mb_pipe (HIP events), and the whole
    step (+ DeblockFrame + getSubImagesLuma)."""
    import torch
    from jm_amd import JmHip
    from jm_amd.lib import SLICE_PARAMS
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synclip
    W4, H4, per, ns = 3840, 2160, 4080, 8
    with tempfile.TemporaryDirectory() as t:
        synclip.syn2160p(os.path.join(t, "s.yuv"), 2)
        data = np.fromfile(os.path.join(t, "s.yuv"), np.uint8)
    fs = W4 * H4 * 3 // 2
    ctx = JmHip(W4, H4, search_range=R, num_ref_slots=2, yuv_format=1, device=device, stream=torch.cuda.current_stream().cuda_stream)
    ctx.set_current_frame(data[:fs].copy(), W4, H4)
    ctx.encode_slice_dev(slice_params(SLICE_PARAMS, 2, 0, per, 0, 0, num_slices=ns))
    ctx.deblock_picture_dev(1)
    ctx.reference_from_recon(0)
    ctx.set_current_frame(data[fs:2 * fs].copy(), W4, H4)
    prm = slice_params(SLICE_PARAMS, 0, 0, per, 0, 1, num_slices=ns)
    prm["ref_slot"][0, 0] = 0
    ctx.enable_timing(True)
    ms = []
    for i in range(4):
        ctx.encode_slice_dev(prm)
        ctx.synchronize()
        ms.append(ctx.last_kernel_ms(5))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5):
        ctx.encode_slice_dev(prm)
        ctx.deblock_picture_dev(1)
        ctx.reference_from_recon(1)
    ctx.synchronize()
    step = (time.perf_counter() - t0) / 5
    ctx.close()
    k = float(np.mean(ms[1:]))
    out = {"kernel": "k_mb_pipe", "slices_in_one_launch": ns, "avg_kernel_ms": round(k, 3), "ms_per_step": round(step * 1e3, 3), "macroblocks_per_s_step": round(32400 / step, 1)}
    # ... and 2160p as ONE slice per picture, an IPPP sequence with its P pictures in one launch (jmhip_seq_batch): not configs[3] (eight slices), the same picture size with four times
    # 1080p's macroblocks to run at a time; the first P pictures' records against the picture-after-picture path
    from jm_amd.lib import MB_RECORD
    nmb4, npic, nslots, ncheck = 32400, 13, 16, 3
    with tempfile.TemporaryDirectory() as t:
        synclip.syn2160p(os.path.join(t, "s.yuv"), npic)
        data = np.fromfile(os.path.join(t, "s.yuv"), np.uint8).reshape(npic, fs)
    dev = torch.device("cuda", device)
    d_raw = torch.from_numpy(data).to(dev)
    d_rec = torch.zeros((npic, nmb4 * MB_RECORD.itemsize), dtype=torch.uint8, device=dev)
    for key, per, note in (("one_slice_sequence_in_one_launch", nmb4, "3840x2160 IPPP, one slice per picture, SR 32, one reference: the P pictures in one launch; not the BASELINE configuration (eight slices)"),
                           ("eight_slice_sequence_in_one_launch", 4080, "3840x2160 IPPP, configs[3]'s eight slices per picture (SliceArgument 4080), SR 32, one reference, the loop filter across the slices' "
                            "edges: the P pictures in one launch, every picture in ITS wavefront order (the records against the slices' wavefronts side by side, picture after picture)")):
        ctx = JmHip(W4, H4, search_range=R, num_ref_slots=nslots, yuv_format=1, device=device, stream=torch.cuda.current_stream().cuda_stream)
        ctx.seq_open(1)

        def prm1(kk):
            q = slice_params(SLICE_PARAMS, 2 if kk == 0 else 0, 0, per, 0, 0 if kk == 0 else 1)
            if per < nmb4:
                q["num_slices"] = (nmb4 + per - 1) // per                   # the picture's slices in the one launch (SliceMode 1)
            if kk:
                q["ref_slot"][0, 0], q["ref_id"][0, 0] = (kk - 1) % nslots, kk - 1
            return q

        def launch(k0, k1):
            ctx.seq_batch(prm1(k0), [dict(d_raw=d_raw[kk].data_ptr(), src_w=W4, src_h=H4, out_slot=kk % nslots, ref_slot=[(kk - 1) % nslots], ref_id=[kk - 1],
                                          d_records=d_rec[kk].data_ptr()) for kk in range(k0, k1)])
        ctx.seq_set_frame_dev(0, d_raw[0].data_ptr(), W4, H4)
        ctx.seq_encode(0, prm1(0), 0, 1, False, d_rec[0].data_ptr())
        launch(1, 3)                                                          # warm-up
        ctx.seq_wait(0)
        ctx.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        launch(3, npic)
        ctx.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        recs = d_rec.cpu().numpy().view(MB_RECORD).reshape(npic, nmb4)
        ctx.seq_close()
        same = True
        for kk in range(1 + ncheck):                                          # the same pictures one after the other
            ctx.set_current_frame(data[kk], W4, H4)
            q = prm1(kk)
            if kk:
                q["ref_slot"][0, 0] = (kk - 1) & 1
            same = same and ctx.encode_slice(q).tobytes() == recs[kk].tobytes()
            ctx.deblock_picture_dev(1)
            ctx.reference_from_recon(kk & 1)
        ctx.close()
        out[key] = {"pictures": npic - 3, "ms_per_picture": round(dt / (npic - 3) * 1e3, 3), "macroblocks_per_s": round(nmb4 * (npic - 3) / dt, 1),
                                                   "records_equal_picture_after_picture": bool(same), "pictures_checked": 1 + ncheck,
                                                   "note": note}
    return out


def concurrent_streams(S, raw0, raw1, src_h, slice_prm, device, steps):
    """An extra figure, never `value`: S independent 1080p sequences (the same step each: P picture through k_mb_pipe + DeblockFrame + getSubImagesLuma)
    on S contexts with their own HIP streams, each slice's launch limited to its share of the chip (jmhip_set_pipeline_workgroups).  One sequence
    is a dependency chain that keeps ~27 of 256 compute units busy; a server encodes several at once."""
    import torch
    from jm_amd import JmHip
    share = max(16, (256 - 32) // S)                                     # 32 compute units stay free for the short kernels (loop filter, interpolation) of all streams
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(S - 1)]      # the stream the bench already owns + S - 1 more: a hardware queue each
    ctxs = []
    for st in streams:
        c = JmHip(W, H, search_range=R, num_ref_slots=2, yuv_format=1, device=device, stream=st.cuda_stream)
        c.set_pipeline_workgroups(share)
        c.set_current_frame(raw0, W, src_h)
        c.encode_slice_dev(slice_prm(2, 0, 8160, 0, 0))
        c.deblock_picture_dev(1)
        c.reference_from_recon(0)
        c.set_current_frame(raw1, W, src_h)
        ctxs.append(c)
    prm = slice_prm(0, 0, 8160, 0, 1)
    prm["ref_slot"][0, 0] = 0

    def round_():
        for c in ctxs:
            c.encode_slice_dev(prm)
            c.deblock_picture_dev(1)
            c.reference_from_recon(1)
    round_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        round_()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.synchronize()                                                  # device-side error words
        c.close()
    return {"streams": S, "workgroups_per_stream": share, "steps": steps, "ms_per_round": round(dt / steps * 1e3, 3),
            "macroblocks_per_s": round(S * 8160 * steps / dt, 1),
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "note": "S sequences side by side on one MI355X, one context + HIP stream each (GPU_MAX_HW_QUEUES raised from its default 4 so that the streams get a hardware queue each); the same step as `value` per sequence; not the BASELINE metric (one sequence)"}


def configs2_params(q, slice_type, poc_cur):
    """configs[2]'s switches on top of slice_params' record: EPZS as the .cfg files ship it, CABAC, High profile (8x8 transform, Intra8x8)"""
    q["search_mode"], q["symbol_mode"] = 3, 1
    q["transform8x8"], q["intra8_valid"] = 1, 1                  # High profile: q_params_8x8 at QP 28 (qp % 6 == 4 rows of quant_coef8 / dequant_coef8, q_matrix.c:38-167)
    s8, d8 = [8192, 7346, 13159, 7740, 10486, 9777], [32, 28, 51, 30, 40, 38]

    def cls8(j, i):
        i4, j4 = i & 3, j & 3
        if i4 == 0 and j4 == 0:
            return 0
        if (i & 1) and (j & 1):
            return 1
        if i4 == 2 and j4 == 2:
            return 2
        if (i4 == 0 and (j & 1)) or ((i & 1) and j4 == 0):
            return 3
        if (i4 == 0 and j4 == 2) or (i4 == 2 and j4 == 0):
            return 4
        return 5
    for intra in range(2):
        off = 682 if (intra and slice_type == 2) else 342
        for j in range(8):
            for i in range(8):
                q["q_luma8"][0, intra, j * 8 + i] = (off << (16 + QP // 6 - 11), s8[cls8(j, i)], d8[cls8(j, i)] << 4)
    for k, v in dict(pattern=2, dual=3, fixed=2, aggressive=0, temporal=1, spatial_mem=1, blocktype=1, min_scale=0, med_scale=1, max_scale=2, sub_scale=2).items():
        q["epzs_" + k] = v                                        # the shipped .cfg files' switches
    q["poc_cur"] = poc_cur
    return q


QUANT_COEF = [(13107, 5243, 8066), (11916, 4660, 7490), (10082, 4194, 6554), (9362, 3647, 5825), (8192, 3355, 5243), (7282, 2893, 4559)]      # quant_coef, by qp % 6 (q_matrix.c:20-27)
DEQUANT_COEF = [(10, 16, 13), (11, 18, 14), (13, 20, 16), (14, 23, 18), (16, 25, 20), (18, 29, 23)]                                                # dequant_coef (q_matrix.c:29-36)


def slice_params(SLICE_PARAMS, slice_type, first, num, slice_nr, num_ref, num_slices=0, qp=QP, qpc=QP, lam=192):
    """jmhip_slice_params with JM's own values for configs[1] / configs[3] with RDO off (tests/golden/mb_low_g2r.npz holds what the encoder used); qp / qpc / lam: a B picture of
    encoder_main.cfg (QPBSlice 30: chroma 29, lambda factor 256 -- tests/golden/mb_low_g3b.npz)"""
    p = np.zeros(1, SLICE_PARAMS)
    p["slice_type"], p["first_mb"], p["num_mb"], p["slice_nr"], p["qp"], p["qpc"] = slice_type, first, num, slice_nr, qp, qpc
    p["search_range"], p["num_ref"], p["num_slices"] = R, num_ref, num_slices
    p["lambda_mf"], p["lambda_mdfp"] = [lam, lam, lam], lam               # lambda_mf / LAMBDA_FACTOR(lambda_md) with RDOptimization = 0: 192 at QP 28
    p["max_mvd"] = 1023                                                  # mv_search.c:327 at SearchRange 32
    p["mv_limit"] = [-8192, 8191, -2048, 2047]                           # level 5.1
    p["inter_valid"], p["intra4_valid"], p["intra16_valid"], p["subpel"], p["start_qp"] = 1, 1, 1, 1, 1
    bits = [1, 3, 3] + [5] * 4 + [7] * 8 + [9]
    p["refbits"] = bits
    for intra in range(2):
        off = 682 if (intra and slice_type == 2) else 342                # q_offsets.c:135-162 default offsets
        for j in range(4):
            for i in range(4):
                c = 0 if (i % 2 == 0 and j % 2 == 0) else (1 if (i % 2 and j % 2) else 2)
                p["q_luma"][0, intra, j * 4 + i] = (off << (15 + qp // 6 - 11), QUANT_COEF[qp % 6][c], DEQUANT_COEF[qp % 6][c] << 4)
                p["q_chroma"][0, :, intra, j * 4 + i] = (off << (15 + qpc // 6 - 11), QUANT_COEF[qpc % 6][c], DEQUANT_COEF[qpc % 6][c] << 4)
    p["df_disable_idc"] = 0                                              # DeblockFrame filters across slice edges (the .cfg files' DFDisableIdc = 0)
    return p


def traffic_live(args, max_seconds=240):
    """HBM-side traffic of the timed launch, measured in THIS run (outside the clock): the same command once more under `rocprofv3 --pmc FETCH_SIZE` and once under
    `--pmc WRITE_SIZE` (separate passes, kernel trace only: the MI355X guide's recipe), as child processes; the timed launch is the third k_mb_pipe dispatch (I picture, warm-up
    launch, timed launch).  Bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the counters are in KB, and FETCH_SIZE counts half of what is fetched on gfx950
    (profiles/microbench/fetch_calib.hip, profiles/r01_v3_kernel_stats.md).  None when rocprofv3 is not there or a pass fails."""
    import csv
    import glob
    import shutil
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    vals = {}
    t_end = time.time() + max_seconds
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "t", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", str(args.steps), "--warmup", str(args.warmup), "--launch", args.launch, "--slots", str(args.slots), "--workgroups", str(args.workgroups),
                   "--flight", str(args.flight), "--no-cpu-baseline", "--no-end-to-end", "--streams", "0", "--no-traffic"]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=max(30.0, t_end - time.time()))
            except (subprocess.TimeoutExpired, OSError):
                return None
            if r.returncode != 0:
                return None
            rows = []
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                rows += list(csv.DictReader(open(f)))
            v = timed_launch_counter(rows, counter)
            if v is None:
                return None
            vals[counter] = v
    return traffic_bytes(vals["FETCH_SIZE"], vals["WRITE_SIZE"])


def timed_launch_counter(rows, counter):
    """rows of rocprofv3's counter_collection.csv -> the counter's value for the timed launch: the THIRD k_mb_pipe dispatch of a bench.py run (the I picture, the warm-up launch,
    the timed launch; the instrumented twin k_mb_pipe_prof and the other instances have other names), or None"""
    mine = [q for q in rows if q.get("Kernel_Name", "").startswith("k_mb_pipe(") and q.get("Counter_Name", counter) == counter]
    mine.sort(key=lambda q: int(q["Dispatch_Id"]))
    return float(mine[2]["Counter_Value"]) if len(mine) >= 3 else None


def traffic_bytes(fetch_kb, write_kb):
    """FETCH_SIZE / WRITE_SIZE (KB; FETCH_SIZE counts half of what is fetched on gfx950: profiles/microbench/fetch_calib.hip) -> bytes"""
    return {"bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024)), "fetch_size_kb": fetch_kb, "write_size_kb": write_kb}


def valu_issued_live(local, d_raw, src_h, nmb, npic=7):
    """Absolute differences the integer searches of ONE P picture of the clip issue, from the kernel's own counters (JMHIP_MB_PROF=11: every wave adds up the window rows its
    sliding lanes read and the candidates of its one-lane passes, mbpipe.hip fs_wave): a second context made with the counting on codes the clip's first pictures in one launch, outside
    any clock; every macroblock address then holds the counts of the last picture that coded it.  None when it cannot be measured (the bench line then carries the constant)."""
    import ctypes as C
    from jm_amd import JmHip
    from jm_amd.lib import SLICE_PARAMS
    os.environ["JMHIP_MB_PROF"] = "11"
    try:
        c = JmHip(W, H, search_range=R, num_ref_slots=npic + 1, yuv_format=1, device=local)
    finally:
        del os.environ["JMHIP_MB_PROF"]
    try:
        c.seq_open(1)

        def prm(k):
            q = slice_params(SLICE_PARAMS, 2 if k == 0 else 0, 0, nmb, 0, 0 if k == 0 else 1)
            if k:
                q["ref_slot"][0, 0], q["ref_id"][0, 0] = k - 1, k - 1
            return q
        c.seq_set_frame_dev(0, d_raw[0].data_ptr(), W, src_h)
        c.seq_encode(0, prm(0), 0, 1, False)
        c.seq_wait(0)
        import torch
        from jm_amd.lib import MB_RECORD
        d_rec = torch.zeros((npic, nmb * MB_RECORD.itemsize), dtype=torch.uint8, device=d_raw.device)
        c.seq_batch(prm(1), [dict(d_raw=d_raw[k].data_ptr(), src_w=W, src_h=src_h, out_slot=k, ref_slot=[k - 1], ref_id=[k - 1], d_records=d_rec[k].data_ptr()) for k in range(1, npic)])
        c.seq_wait(0)
        c.synchronize()
        st = np.zeros((nmb, 32), np.uint64)
        if c.lib.jmhip_debug_read_mb_prof(c.h, st.ctypes.data_as(C.c_void_p), st.nbytes) != 0:
            return None
        return float(16 * st[:, 22:30].astype(np.int64).sum())
    finally:
        c.close()


def b_pictures_leg(local, frames, src_h, nmb, flight):
    """encoder_main.cfg's settings with RDO off at 1080p (the shipped file's B picture: NumberBFrames 1, fast full search SR 32, CABAC, two references, spatial direct, the
    bi-predictive search with three refinements / range 16 / two sub-pel levels -- tests/golden/mb_low_g3b.npz is the real encoder's I P B of this clip):
    the B picture's launch alone (k_mb_pipe_b, records against the real encoder's), then I + 12 x (P B) with pictures in flight (jmhip_seq_encode: P pictures follow their
    references inside the device, a B picture starts when both its references are complete and runs beside the P pictures after them)."""
    import torch
    from jm_amd import JmHip
    from jm_amd.lib import SLICE_PARAMS, MB_RECORD
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import mb_tap
    if len(frames) < 3:
        frames = yuv_frames(3)
    bsw = 1 | 2 | (7 << 2) | (3 << 8) | (16 << 16) | (2 << 24)          # direct_8x8_inference, BiPredMotionEstimation, 16x16 / 16x8 / 8x16, 3 refinements, range 16, sub-pel 2
    ngop = 12
    # entries: the P pictures' and the B pictures' own (a B macroblock takes ~600 us of a workgroup's time, a P macroblock of this search ~200; the B pictures are what fills
    # the chip, the P pictures only have to stay ahead of them).  JMHIP_BENCH_B="P entries,B entries,workgroups per P picture,per B picture": measurement aid
    n_p, n_b, wg_p, wg_b = (int(x) for x in os.environ.get("JMHIP_BENCH_B", "4,6,32,32").split(","))
    depth = n_p + n_b
    nring = 2 + n_p + 2                                                  # reference pictures: the window of two + those in flight + two
    nslots = nring + n_b + 1                                             # ... and the B pictures' own slots behind them
    ctx = JmHip(W, H, search_range=R, num_ref_slots=nslots, yuv_format=1, device=local)

    def prm(st, refs0, refs1=()):
        q = slice_params(SLICE_PARAMS, st, 0, nmb, 0, len(refs0), **(dict(qp=30, qpc=29, lam=256) if st == 1 else {}))
        q["search_mode"], q["symbol_mode"] = 1, 1
        for r, (slot, pid) in enumerate(list(refs0) + list(refs1)):
            q["ref_slot"][0, r], q["ref_id"][0, r] = slot, pid
        if st == 1:
            q["num_ref1"], q["b_switches"] = len(refs1), bsw
        return q
    # ---- picture after picture: I (frame 0), P (frame 2), B (frame 1) against the real encoder's records
    ctx.enable_timing(True)
    ctx.set_current_frame(frames[0], W, src_h); r_i = ctx.encode_slice(prm(2, [])); ctx.deblock_picture_dev(1); ctx.reference_from_recon(0)
    ctx.set_current_frame(frames[2], W, src_h); r_p = ctx.encode_slice(prm(0, [(0, 0)])); ctx.deblock_picture_dev(1); ctx.reference_from_recon(1)
    ctx.set_current_frame(frames[1], W, src_h)
    pb = prm(1, [(0, 0), (1, 1)], [(1, 1)])
    bms = []
    for i in range(3):
        ctx.encode_slice_dev(pb)
        ctx.synchronize()
        bms.append(ctx.last_kernel_ms(5))
    r_b = ctx.encode_slice(pb)
    gold = mb_tap.widen(np.load(os.path.join(ROOT, "tests", "golden", "mb_low_g3b.npz"))["records"])
    eq = all(all(a.tobytes() == b.tobytes() for a, b in zip(mb_tap.canonical(np.frombuffer(r.tobytes(), gold.dtype).copy(), bslice=k == 2), gold[k * nmb:(k + 1) * nmb]))
             for k, r in enumerate((r_i, r_p, r_b)))
    types = np.bincount(r_b["mb_type"].astype(int), minlength=14)
    # ---- the sequence with pictures in flight
    ctx.seq_open(depth, wg_p, ready=True)
    ctx.seq_b_workgroups(wg_b)
    d_raw = torch.from_numpy(np.stack(frames)).to(f"cuda:{local}")
    npic = 1 + 2 * ngop
    d_recs = torch.zeros((npic, nmb * MB_RECORD.itemsize), dtype=torch.uint8, device=f"cuda:{local}")
    order = [(0, 2)] + [x for g in range(ngop) for x in ((2 * g + 2, 0), (2 * g + 1, 1))]      # (display number, slice type) in coding order
    nf = len(frames)

    def run():
        stored, nref_pic, nb_pic = [], 0, 0                              # (slot, picture id) of the stored references, most recent first
        for k, (disp, st) in enumerate(order):
            e = n_p + nb_pic % n_b if st == 1 else nref_pic % n_p
            ctx.seq_set_frame_dev(e, d_raw[disp % nf].data_ptr(), W, src_h)
            if st == 1:
                past, future = stored[1], stored[0]                      # the window of two: the picture before and the picture after this one
                ctx.seq_encode(e, prm(1, [past, future], [future]), nring + nb_pic % (n_b + 1), 1, False, d_recs[k].data_ptr())
                nb_pic += 1
            else:
                slot = nref_pic % nring
                ctx.seq_encode(e, prm(st, stored[:2] if st == 0 else []), slot, 1, False, d_recs[k].data_ptr())
                stored = ([(slot, k)] + stored)[:2]
                nref_pic += 1
    run()
    torch.cuda.synchronize(); ctx.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.synchronize()
    last_ms = [round(ctx.seq_kernel_ms(e), 1) for e in range(depth)]    # the coding launch each entry ran last (P and B pictures alternate over the entries)
    r3 = d_recs[:3].cpu().numpy().view(MB_RECORD).reshape(3, nmb)
    eq_fl = all(a.tobytes() == b.tobytes() for a, b in zip((r_i, r_p, r_b), r3))
    ctx.seq_close()
    ctx.close()
    return {"workload": "encoder_main.cfg with RDO off at 1080p: I P B P B ... (NumberBFrames 1, non-reference B pictures, spatial direct), fast full search SR 32, CABAC, two references, "
                        "BiPredMotionEstimation 1 (3 refinements, range 16, sub-pel 2; 16x16 / 16x8 / 8x16), QP 28 / 28 / 30",
            "b_picture_alone": {"kernel": "k_mb_pipe_b", "avg_kernel_ms": round(float(np.mean(bms[1:])), 3), "macroblocks_per_s": round(nmb / (float(np.mean(bms[1:])) * 1e-3), 1),
                                "mb_types_direct_16x16_16x8_8x16_p8x8_i4_i16": [int(types[k]) for k in (0, 1, 2, 3, 8, 9, 10)]},
            "records_equal_jm_i_p_b": bool(eq),
            "in_flight": {"pictures": npic, "b_pictures": ngop, "pictures_in_flight": depth, "entries_p_b": [n_p, n_b], "workgroups_per_p_b_picture": [wg_p, wg_b], "ms_per_picture": round(dt / npic * 1e3, 3), "macroblocks_per_s": round(nmb * npic / dt, 1),
                          "last_launch_ms_by_entry": last_ms, "records_equal_picture_after_picture_first_three": bool(eq_fl)}}


class _DevMem:
    """device memory of the library as a torch tensor (for the collective)"""
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def multi_gpu_configs3(args, world, rank, local, dev, one_gpu):
    """N > 1: BASELINE configs[3] with RDO off -- ONE 3840x2160 picture in 8 slices of 4080 macroblocks (SliceMode 1: bands of 17 ... 17, 16
    macroblock rows), the slices dealt to the N GPUs in order (8 / N each, their wavefronts side by side in one launch), DFDisableIdc = 0 as the .cfg files
    ship it: DeblockFrame filters across slice edges, so the picture is deblocked AFTER the exchange.  A step:
      jmhip_encode_slice_dev       this rank's slices (k_mb_pipe), reconstruction and loop filter side information left in the rank's picture buffers
      ONE all-gather (RCCL)        every rank's un-deblocked rows of Y / U / V + its rows of jmhip_db_mb / jmhip_db_motion (jm_amd.shard.BandGather)
      jmhip_deblock_picture_dev    the whole picture, on every rank (redundant by design: 0.3 ms against a halo protocol)
      jmhip_reference_from_recon   getSubImagesLuma of the whole picture: every rank holds the whole reference for the next picture's searches
    Total work is fixed as N grows ("strong").  After the timed region rank 0 hashes the I and the P picture's reconstruction: it must be the md5 CPU JM's
    -o file has for this clip and these flags (tests/golden/md5.json G4r), or the number is void.  `independent_sequences` is the other way to use N GPUs --
    one 1080p sequence (configs[1]'s step) per GPU, no collective at all -- measured in the same run."""
    import hashlib
    import torch
    import torch.distributed as dist
    from jm_amd import JmHip, shard
    from jm_amd.lib import SLICE_PARAMS
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synclip
    W4, H4 = 3840, 2160
    mbw, mbh = W4 // 16, H4 // 16
    nmb = mbw * mbh
    n_slices = 8 if 8 % world == 0 else world
    rows = -(-mbh // n_slices)                              # 17 macroblock rows per slice: SliceArgument = 4080
    per = shard.slice_argument(mbh, mbw, n_slices)
    spr = n_slices // world                                 # slices per rank
    k = rows * spr                                          # macroblock rows per rank (the last rank's band is one row short)
    first = rank * spr * per
    mine = min(spr * per, nmb - first)

    stream = torch.cuda.current_stream()
    ctx = JmHip(W4, H4, search_range=R, num_ref_slots=2, yuv_format=1, device=local, stream=stream.cuda_stream)
    with tempfile.TemporaryDirectory() as tmp:
        synclip.syn2160p(os.path.join(tmp, "s.yuv"), 2)
        data = np.fromfile(os.path.join(tmp, "s.yuv"), np.uint8)
    fs = W4 * H4 * 3 // 2
    raw0, raw1 = data[:fs].copy(), data[fs:2 * fs].copy()

    def flat(planes):
        return b"".join(np.ascontiguousarray(p).astype(np.uint8).tobytes() for p in planes)

    # the reference: the I picture, all eight slices on every rank (setup, untimed)
    ctx.set_current_frame(raw0, W4, H4)
    ctx.encode_slice_dev(slice_params(SLICE_PARAMS, 2, 0, per, 0, 0, num_slices=n_slices))
    ctx.deblock_picture_dev(1)
    rec_i = flat(ctx.get_recon()) if rank == 0 else None
    ctx.reference_from_recon(0)
    ctx.set_current_frame(raw1, W4, H4)
    ctx.synchronize()
    prm = slice_params(SLICE_PARAMS, 0, first, per if spr > 1 else mine, rank * spr, 1, num_slices=spr)
    prm["ref_slot"][0, 0] = 0

    py, pitch, pu, pv, pc = ctx.recon_planes_dev()
    pm, nm, po, no = ctx.deblock_side_info_dev()
    planes = [(torch.as_tensor(_DevMem(py, pitch * H4), device=dev).view(H4, pitch), 16 * k),
              (torch.as_tensor(_DevMem(pu, pc * H4 // 2), device=dev).view(H4 // 2, pc), 8 * k),
              (torch.as_tensor(_DevMem(pv, pc * H4 // 2), device=dev).view(H4 // 2, pc), 8 * k),
              (torch.as_tensor(_DevMem(pm, nm), device=dev).view(mbh, nm // mbh), k),                 # jmhip_db_mb: one row per macroblock row
              (torch.as_tensor(_DevMem(po, no), device=dev).view(4 * mbh, no // (4 * mbh)), 4 * k)]   # jmhip_db_motion: four rows per macroblock row
    if one_gpu:
        host = [(t.cpu(), kk) for t, kk in planes]
        gather = shard.BandGather(host, world, rank)
    else:
        gather = shard.BandGather(planes, world, rank)
    exchanged_bytes = int(gather.all.numel())

    def exchange():
        if one_gpu:
            for (h, _), (t, _) in zip(host, planes):
                h.copy_(t)
            gather()
            for (h, _), (t, _) in zip(host, planes):
                t.copy_(h)
        else:
            gather()

    def step():
        ctx.encode_slice_dev(prm)
        exchange()
        ctx.deblock_picture_dev(1)
        ctx.reference_from_recon(1)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ctx.enable_timing(True)
    for i in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    ctx.synchronize()
    ms = []
    for i in range(3):
        ctx.encode_slice_dev(prm)
        ctx.synchronize()
        ms.append(ctx.last_kernel_ms(5))
    pipe_ms = max_over_ranks(float(np.mean(ms)))
    step()                                                   # the picture buffers hold a whole deblocked P picture again
    ctx.synchronize()
    md5_ok = None
    if rank == 0:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "md5.json")))["G4r"]["md5_recon"]
        md5_ok = hashlib.md5(rec_i + flat(ctx.get_recon())).hexdigest() == gold
    ctx.close()

    if rank == 0:
        alg_mb = 6656 + 328 + 128 + 2900 + 1216 + 384       # as at N = 1
        alg = alg_mb * mine
        out = {
            "metric": "encoded macroblocks/sec (bit-exact vs CPU JM), ONE 2160p picture's 8 slices dealt to the GPUs, SR=32 (BASELINE configs[3], strong scaling)",
            "value": round(nmb * args.steps / dt, 1), "unit": "macroblocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"configs[3] with RDO off (G4r): ONE 3840x2160 4:2:0 synthetic picture ({nmb} MB) in {n_slices} slices of {per} macroblocks (SliceMode 1), "
                                   "Baseline IPPP, FullSearch SR=32, 1 ref, QP 28, DFDisableIdc = 0; P picture through the RDO-off macroblock pipeline + DeblockFrame + getSubImagesLuma",
                       "macroblocks_per_step": nmb, "macroblocks_per_step_rank0": mine, "search_range": R,
                       "parallelism": f"{spr} slice(s) per GPU on {world} GPUs; per step ONE all-gather of {exchanged_bytes} bytes (un-deblocked Y/U/V rows + loop filter side information), "
                                      "then every GPU deblocks and interpolates the whole picture and keeps the whole reference"
                                      + (" [UNMEASURED debugging path: all ranks on one GPU, exchange over gloo on host copies]" if one_gpu else ""),
                       "recon_md5_equals_cpu_jm": md5_ok,
                       "note": "one GPU already runs all eight slices side by side (configs3 object of the N = 1 line): a slice is a chain of 240 + 2 x 16 dependent macroblocks "
                               "whatever the number of GPUs, so the split cannot go below one slice's chain plus the exchange; N GPUs pay off for N closed GOPs (this line's `value`)"},
            "roofline": {"kernel": "k_mb_pipe", "bound": "latency", "achieved": round(alg / (pipe_ms * 1e-3) / 1e9, 3), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(alg / (pipe_ms * 1e-3) / 8e12, 6), "traffic": None, "avg_kernel_ms": round(pipe_ms, 3), "algorithmic_bytes_per_launch": alg,
                         "note": "the slowest rank's launch over its own slices"},
        }
        return out
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes that measure roofline.traffic (the committed constant is reported instead)")
    ap.add_argument("--flight", type=int, default=8, help="pictures of the sequence in flight side by side (jmhip_seq_open; 1 = one launch at a time)")
    ap.add_argument("--workgroups", type=int, default=0, help="workgroups per picture in flight (0: 256 / flight); with --launch batch: workgroups of the launch (0: 256)")
    ap.add_argument("--launch", choices=["batch", "pictures"], default="batch", help="batch: the timed P pictures in ONE launch, every picture's macroblocks from one queue (jmhip_seq_batch); "
                    "pictures: a launch per picture, --flight of them side by side (jmhip_seq_encode)")
    ap.add_argument("--slots", type=int, default=24, help="--launch batch: reference slots the pictures go to in turn (a picture starts once the last reader of its slot is done)")
    ap.add_argument("--streams", type=int, default=8, help="sequences encoded side by side for the extra concurrent_streams figure (0: skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from jm_amd import JmHip
    from jm_amd.lib import MB_RECORD, SLICE_PARAMS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    # JMHIP_BENCH_ONE_GPU=1 (debugging aid, never used for a reported number): all ranks share GPU 0 and the exchange goes through
    # gloo on host copies, so the N > 1 code path can be exercised on a single-GPU box
    one_gpu = os.environ.get("JMHIP_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from jm_amd import shard
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if one_gpu else "nccl", **({} if one_gpu else {"device_id": dev}))
    # N > 1: the SAME step on every GPU -- each rank codes its own closed GOP (an I picture and its P pictures: what JM writes with IDRPeriod = the GOP's length needs nothing
    # of the GOP before it), no collective on the data path (weak scaling); the slice split of configs[3] with its all-gather is measured beside it (configs3_slice_split)
    N, HP = world, H
    mbw, mbh = W // 16, H // 16
    nmb = mbw * mbh
    depth = max(1, min(8, args.flight))
    if one_gpu and world > 1:
        depth = max(1, 4 // world)                                        # the ranks share one GPU: the launches in flight of ALL of them must leave room for each rank's oldest picture
    batch = args.launch == "batch"
    nslots = max(depth + 2, min(32, args.slots)) if batch else depth + 2   # one reference + the pictures in flight + one: no launch ever waits for a slot
    nseq = 1 + args.warmup + args.steps                                    # the I picture, the warm-up and the timed P pictures: one IPPP sequence

    stream = torch.cuda.current_stream()
    ctx = JmHip(W, HP, search_range=R, num_ref_slots=nslots, yuv_format=1, device=local, stream=stream.cuda_stream)
    ctx.seq_open(1 if batch else depth, 0 if batch else args.workgroups, ready=True)

    def slice_prm(slice_type, first, num, slice_nr, num_ref):
        return slice_params(SLICE_PARAMS, slice_type, first, num, slice_nr, num_ref)

    def seq_prm(k):
        q = slice_prm(2 if k == 0 else 0, 0, nmb, 0, 0 if k == 0 else 1)
        if k:
            q["ref_slot"][0, 0], q["ref_id"][0, 0] = (k - 1) % nslots, k - 1
        return q

    # ---------------- inputs: the clip (continued), every source picture resident in HBM as the file holds it before the clock starts
    src_h = H_SRC
    frames = yuv_frames(nseq, seed=1234 + rank)                          # rank 0: the clip the golden records were made from
    raw0, raw1 = frames[0], frames[1]
    d_raw = torch.from_numpy(np.stack(frames)).to(dev)
    d_recs = torch.zeros((nseq, nmb * MB_RECORD.itemsize), dtype=torch.uint8, device=dev)      # every picture's records (the entries' own are reused)
    ctx.enable_timing(True)

    def step(k):
        """picture k of the sequence: read_one_frame / pad_borders -> encode_one_macroblock_low of every macroblock, DeblockMb and the sixteen sub-pel planes behind
        each of them (k_load_frame, k_mb_pipe); asynchronous -- up to `depth` pictures are in flight"""
        e = k % depth
        ctx.seq_set_frame_dev(e, d_raw[k].data_ptr(), W, src_h)
        ctx.seq_encode(e, seq_prm(k), k % nslots, 1, False, d_recs[k].data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def over_ranks(x, op):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if one_gpu else dev)
        dist.all_reduce(t, op=op)
        return float(t.item())

    def steps_in_one_launch(k0, k1):
        """pictures k0 .. k1 - 1 of the sequence, the same work per picture, in ONE launch of k_mb_pipe: the launch's workgroups draw the macroblocks of all of them from one queue
        ordered by wavefront index + 16 x picture (jmhip_seq_batch); asynchronous"""
        ctx.seq_batch(seq_prm(k0), [dict(d_raw=d_raw[k].data_ptr(), src_w=W, src_h=src_h, out_slot=k % nslots, ref_slot=[(k - 1) % nslots], ref_id=[k - 1],
                                         d_records=d_recs[k].data_ptr()) for k in range(k0, k1)])

    with_io = None
    if batch:
        ctx.set_pipeline_workgroups(args.workgroups or (256 // world if one_gpu else 256))
        ctx.seq_batch_reserve(max(args.steps, args.warmup, 1))           # the launches' scratch (per-picture edge records, flags, side information) allocated before the clock starts, as JM allocates per sequence
        step(0)                                                            # the I picture
        if args.warmup:
            steps_in_one_launch(1, 1 + args.warmup)
        barrier()
        t0 = time.perf_counter()
        steps_in_one_launch(1 + args.warmup, nseq)
        barrier()
        dt = over_ranks(time.perf_counter() - t0, dist.ReduceOp.MAX)
        kernel_ms = [ctx.last_kernel_ms(5)]                                # the timed launch
        ctx.seq_wait(0)
        ctx.synchronize()                                                  # reads the device-side error word (sticky: an incomplete picture cannot go unnoticed)
        # ---- the same launch with the I/O inside the clock (an extra figure; `value` keeps the contract's HBM-resident inputs): the source pictures lie in PINNED HOST memory as the
        # file holds them and k_load_frame reads them over PCIe, the records are written by the kernel straight into pinned host memory as each macroblock finishes -- both
        # overlapped with the coding by construction (no separate copy).  The sequence is coded again from its I picture so that every slot holds what the first run found there.
        if N == 1:
            try:
                h_raw = torch.from_numpy(np.stack(frames)).pin_memory()
                h_recs = torch.zeros((nseq, nmb * MB_RECORD.itemsize), dtype=torch.uint8).pin_memory()
                step(0)
                if args.warmup:
                    steps_in_one_launch(1, 1 + args.warmup)
                barrier()
                t0 = time.perf_counter()
                ctx.seq_batch(seq_prm(1 + args.warmup), [dict(d_raw=h_raw[k].data_ptr(), src_w=W, src_h=src_h, out_slot=k % nslots, ref_slot=[(k - 1) % nslots], ref_id=[k - 1],
                                                              d_records=h_recs[k].data_ptr()) for k in range(1 + args.warmup, nseq)])
                barrier()
                dt_io = time.perf_counter() - t0
                ctx.seq_wait(0)
                ctx.synchronize()
                io_same = h_recs[1 + args.warmup:].numpy().tobytes() == d_recs[1 + args.warmup:].cpu().numpy().tobytes()
                up, down = int(frames[0].nbytes), nmb * MB_RECORD.itemsize
                with_io = {"value": round(nmb * args.steps / dt_io, 1), "unit": "macroblocks/s", "ms_per_step": round(dt_io / args.steps * 1e3, 4), "ratio_to_value": round(dt / dt_io, 4),
                           "records_equal": bool(io_same), "bytes_up_per_picture": up, "bytes_down_per_picture": down,
                           "note": "the timed launch again with every source picture read from pinned host memory (k_load_frame over PCIe) and every record written by the kernel into "
                                   "pinned host memory: the PCIe traffic of an encoder that keeps nothing on the device but the references, inside the clock"}
            except Exception as ex:                                           # (an extra figure must not cost the line)
                with_io = {"error": repr(ex)[:300]}
        ctx.set_pipeline_workgroups(0)
    else:
        for k in range(1 + args.warmup):                                   # the I picture and the warm-up P pictures
            step(k)
        barrier()
        t0 = time.perf_counter()
        for k in range(1 + args.warmup, nseq):
            step(k)
        barrier()
        dt = over_ranks(time.perf_counter() - t0, dist.ReduceOp.MAX)
        kernel_ms = [ctx.seq_kernel_ms(e) for e in range(min(depth, args.steps))]      # the last launches, in flight together
        for e in range(depth):
            ctx.seq_wait(e)                                                # reads the device-side error words: the pipeline's is sticky (a launch that finds it set does nothing), so an incomplete picture cannot go unnoticed
        ctx.synchronize()
    pipe_ms = over_ranks(float(np.mean(kernel_ms)), dist.ReduceOp.MAX)
    recs_all = d_recs.cpu().numpy().view(MB_RECORD).reshape(nseq, nmb)

    # ---------------- the same sequence picture after picture (k_mb_pipe, then DeblockFrame, then getSubImagesLuma, each waiting for the one before): the round-3 step.
    # Every picture's records must be the same bytes; the launch alone gives the kernel's solo time.
    ctx.seq_close()
    ctx.enable_timing(True)
    solo_ms, same = [], True
    t1 = time.perf_counter()
    for k in range(nseq):
        ctx.set_current_frame(frames[k], W, src_h)
        q = seq_prm(k)
        if k:
            q["ref_slot"][0, 0] = (k - 1) & 1
        recs = ctx.encode_slice(q)
        if k:
            solo_ms.append(ctx.last_kernel_ms(5))
        else:
            i_ms = ctx.last_kernel_ms(5)                                  # the I picture's launch (no searches: the Intra4x4 chains on two waves), alone
        same = same and recs.tobytes() == recs_all[k].tobytes()
        ctx.deblock_picture_dev(1)
        ctx.reference_from_recon(k & 1)
    ctx.synchronize()
    classic_s = time.perf_counter() - t1
    same = over_ranks(1.0 if same else 0.0, dist.ReduceOp.MIN) == 1.0    # every rank's GOP

    # ---------------- the records against the real encoder's (the pictures the committed dumps hold)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import mb_tap
    gname = "mb_low_g6r.npz" if os.path.exists(os.path.join(ROOT, "tests", "golden", "mb_low_g6r.npz")) else "mb_low_g2r.npz"
    gold = mb_tap.widen(np.load(os.path.join(ROOT, "tests", "golden", gname))["records"])
    ngold = min(len(gold) // nmb, nseq) if rank == 0 else 0              # the golden records are of rank 0's clip
    equal = True
    for k in range(ngold):
        mine = np.frombuffer(recs_all[k].tobytes(), gold.dtype).copy()
        equal = equal and all(a.tobytes() == b.tobytes() for a, b in zip(mb_tap.canonical(mine), gold[k * nmb:(k + 1) * nmb]))
    types = np.bincount(recs_all[1 + args.warmup:]["mb_type"].astype(int).ravel(), minlength=14)

    # ---------------- configs[2]'s search on the same pictures: the P picture through k_mb_pipe_epzs (EPZS, CABAC), records against the real encoder's (g3e)
    slice_split = None
    if world > 1:
        ctx.close()
        slice_split = multi_gpu_configs3(args, world, rank, local, dev, one_gpu)
    configs2_device = {}
    # (with --no-end-to-end the configs[2..4] figures are not reported: the profiler's passes -- whose counter collection serialises the launches -- then see no pictures in
    # flight launch by launch)
    if N == 1 and rank == 0 and not args.no_end_to_end:
        def epzs_prm(slice_type, num_ref, poc_cur):
            return configs2_params(slice_prm(slice_type, 0, nmb, 0, num_ref), slice_type, poc_cur)
        ctx.set_current_frame(raw0, W, src_h)
        ctx.encode_slice_dev(epzs_prm(2, 0, 0))
        ctx.deblock_picture_dev(1)
        ctx.reference_from_recon(0)
        ctx.set_current_frame(raw1, W, src_h)
        pe = epzs_prm(0, 1, 2)
        pe["ref_slot"][0, 0] = 0
        ems = []
        for i in range(4):
            ctx.encode_slice_dev(pe)
            ctx.synchronize()
            ems.append(ctx.last_kernel_ms(5))
        erecs = ctx.encode_slice(pe)
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import mb_tap
        g3 = mb_tap.widen(np.load(os.path.join(ROOT, "tests", "golden", "mb_low_g3h.npz"))["records"][nmb:2 * nmb])
        mine = np.frombuffer(erecs.tobytes(), g3.dtype).copy()
        configs2_device = {"workload": "configs[2]: 1080p, High profile (CABAC, 8x8 transform on: Transform8x8Mode 1, Intra8x8), EPZS (pattern 2, dual 3, fixed 2, temporal, "
                                       "spatial memory, block type, sub-pel grid), RDO off, P picture with one reference", "kernel": "k_mb_pipe_epzs4_t8 (a launch alone)",
                           "avg_kernel_ms": round(float(np.mean(ems[1:])), 3), "macroblocks_per_s": round(nmb / (float(np.mean(ems[1:])) * 1e-3), 1),
                           "records_equal_jm": bool(all(a.tobytes() == b.tobytes() for a, b in zip(mb_tap.canonical(mine), g3)))}
        # ... and the same search as a sequence with pictures in flight (every search asks for what it reaches of a reference in the making): I + 16 P pictures of the clip, one
        # reference; the first three pictures' records against the real encoder's (mb_low_g3h holds three: I, P with one reference, P with two -- the third differs by its reference count, so two are compared)
        # EPZS P pictures run as four-wave workgroups, two to a compute unit: sixteen pictures in flight x 2 x 16 workgroups fill the chip (profiles/r04_epzs_four_wave.txt)
        nq = 65                                                          # an I picture and 64 P pictures: the clip's pictures forwards and backwards (four fills of the sixteen entries; a launch takes ~20 x the steady state's time per picture)
        depth_e = depth if (one_gpu and world > 1) else max(1, min(16, nslots - 2, 2 * args.flight))
        ctx.seq_open(depth_e, 0 if batch else args.workgroups, ready=True)
        d_r2 = torch.zeros((nq, nmb * MB_RECORD.itemsize), dtype=torch.uint8, device=dev)

        def pp(k):
            """the clip's pictures forwards, then backwards, ...: more pictures than the clip has without a scene cut where it would start again (a cut is where EPZS reaches furthest,
            and a launch of several pictures is then given up every other run: profiles/r05_epzs_batch.txt)"""
            m = k % (2 * (nseq - 1)) if nseq > 1 else 0
            return m if m < nseq else 2 * (nseq - 1) - m

        def estep(k):
            q = epzs_prm(2 if k == 0 else 0, 0 if k == 0 else 1, 2 * k)
            if k:
                q["ref_slot"][0, 0], q["ref_id"][0, 0], q["poc_ref"][0, 0] = (k - 1) % nslots, k - 1, 2 * (k - 1)
            ctx.seq_set_frame_dev(k % depth_e, d_raw[pp(k)].data_ptr(), W, src_h)
            ctx.seq_encode(k % depth_e, q, k % nslots, 1, False, d_r2[k].data_ptr())
        estep(0)
        barrier()
        te = time.perf_counter()
        for k in range(1, nq):
            estep(k)
        barrier()
        te = time.perf_counter() - te
        for e in range(depth_e):
            ctx.seq_wait(e)
        # ... and with the P pictures in ONE launch (jmhip_seq_batch, round 5: the queue ordered for what EPZS usually reaches, every search checked against it -- a search
        # that reaches further voids the launch with JMHIP_EREACH and the pictures go through the launches above): the same pictures, their records compared with that run's
        one_launch = None
        try:
            d_r3 = torch.zeros((nq, nmb * MB_RECORD.itemsize), dtype=torch.uint8, device=dev)
            ctx.seq_batch_reserve(nq - 1)
            ctx.set_pipeline_workgroups(args.workgroups if batch else 0)
            estep(0)
            ctx.seq_wait(0)
            q1 = epzs_prm(0, 1, 2)
            q1["poc_ref"][0, 0] = 0
            pics = [dict(d_raw=d_raw[pp(k)].data_ptr(), src_w=W, src_h=src_h, out_slot=k % nslots, ref_slot=[(k - 1) % nslots], ref_id=[k - 1], poc_offset=2 * (k - 1),
                         d_records=d_r3[k].data_ptr()) for k in range(1, nq)]
            given_up = []
            for lag in (0, 20, 32):                                       # the library's lag; a launch that is given up (JMHIP_EREACH) once more with its pictures further apart
                ctx.seq_batch_lag(lag)
                if given_up:
                    estep(0)
                    ctx.seq_wait(0)
                barrier()
                tb = time.perf_counter()
                ctx.seq_batch(q1, pics)
                barrier()
                tb = time.perf_counter() - tb
                kms = ctx.last_kernel_ms(5)
                try:
                    ctx.synchronize()
                    same3 = bool(torch.equal(d_r3[1:], d_r2[1:]))
                    one_launch = {"pictures": nq - 1, "kernel": "k_mb_pipe_epzs4_t8, one launch (jmhip_seq_batch)", "ms_per_picture": round(tb / (nq - 1) * 1e3, 3), "macroblocks_per_s": round(nmb * (nq - 1) / tb, 1),
                                  "kernel_ms": round(kms, 3), "records_equal_pictures_in_flight": same3, "queue_lag": lag or "the library's (13 at SearchRange 32)", "given_up_with_lags": given_up}
                    break
                except Exception as ex:
                    given_up.append(lag)
                    one_launch = {"void": repr(ex)[:300], "given_up_with_lags": given_up, "note": "every launch was given up (JMHIP_EREACH): these pictures are coded by the launches in flight above"}
            ctx.seq_batch_lag(0)
            ctx.set_pipeline_workgroups(0)
        except Exception as ex:                                           # (an extra figure must not cost the line)
            one_launch = {"error": repr(ex)[:300]}
        ctx.seq_close()
        g3all = mb_tap.widen(np.load(os.path.join(ROOT, "tests", "golden", "mb_low_g3h.npz"))["records"])
        r2 = d_r2[:2].cpu().numpy().view(MB_RECORD).reshape(2, nmb)
        eq2 = all(all(a.tobytes() == b.tobytes() for a, b in zip(mb_tap.canonical(np.frombuffer(r2[k].tobytes(), g3all.dtype).copy()), g3all[k * nmb:(k + 1) * nmb])) for k in range(2))
        configs2_device["in_flight"] = {"pictures": nq - 1, "pictures_in_flight": depth_e, "kernel": "k_mb_pipe_epzs4_t8 (four waves per workgroup, two workgroups per compute unit)", "ms_per_picture": round(te / (nq - 1) * 1e3, 3), "macroblocks_per_s": round(nmb * (nq - 1) / te, 1),
                                        "records_equal_jm_first_two_pictures": bool(eq2)}
        configs2_device["one_launch"] = one_launch

    issued_live = None
    if rank == 0 and N == 1:
        try:
            issued_live = valu_issued_live(local, d_raw, src_h, nmb)
        except Exception:                                             # (an extra figure must not cost the line)
            issued_live = None
    traffic = None
    if rank == 0 and N == 1 and world == 1 and batch and not args.no_traffic:
        try:
            traffic = traffic_live(args)
        except Exception:                                             # (an extra figure must not cost the line)
            traffic = None
    if rank == 0:
        total_mb = nmb * N * args.steps
        steps_chain = W // 16 + 2 * (H // 16 - 1)
        # algorithmic bytes per macroblock: SURVEY 8d's 6656 + 328 B per macroblock-reference for the search, the source
        # macroblock's chroma (128 B), ~2.9 KB of transform/quant traffic for the coded mode, the 1216-byte record, 384 B of reconstruction; with the
        # loop filter and the interpolation inside the launch also the filtered macroblock read back with its halo (21 x 24 B) and its share of the sixteen planes (4096 B)
        alg_mb = 6656 + 328 + 128 + 2900 + 1216 + 384 + 504 + 4096
        alg = alg_mb * nmb
        sad_ops = 7 * 256 * (2 * R + 1) ** 2 * nmb                       # seven block types x 256 samples x 4225 positions per macroblock-reference
        per_launch = args.steps if batch else 1                          # pictures a launch codes
        conc = pipe_ms * 1e-3 * (args.steps / per_launch) / dt           # launches in flight at a time, on average over the timed region
        solo = float(np.mean(solo_ms)) if solo_ms else None
        roof = {"kernel": "k_mb_pipe", "bound": "latency", "achieved": round(alg * args.steps / dt / 1e9, 3), "peak": 8000.0, "unit": "GB/s",
                "frac": round(alg * args.steps / dt / 8e12, 6), "traffic": (traffic["bytes_per_launch"] if traffic else PIPE_TRAFFIC_BYTES[args.launch] * per_launch) if N == 1 else None,
                "traffic_source": ("measured in this run, outside the clock: this command once more under rocprofv3 --pmc FETCH_SIZE and once under --pmc WRITE_SIZE (child processes, "
                                   "kernel trace only), the timed launch's dispatch; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- KB units, FETCH_SIZE x 2 on gfx950; "
                                   f"FETCH_SIZE {traffic['fetch_size_kb']:.0f} KB, WRITE_SIZE {traffic['write_size_kb']:.0f} KB" if traffic else PIPE_TRAFFIC_SOURCE),
                "traffic_over_algorithmic": round((traffic["bytes_per_launch"] if traffic else PIPE_TRAFFIC_BYTES[args.launch] * per_launch) / (alg * per_launch), 2) if N == 1 else None, "avg_kernel_ms": round(pipe_ms, 3), "algorithmic_bytes_per_launch": alg * per_launch,
                "pictures_per_launch": per_launch, "launches_in_flight": round(conc, 2),
                "per_launch": {"achieved": round(alg * per_launch / (pipe_ms * 1e-3) / 1e9, 3), "frac": round(alg * per_launch / (pipe_ms * 1e-3) / 8e12, 6),
                               "one_picture_launch_alone_ms": round(solo, 3) if solo else None,
                               "note": ("one launch = the timed region's pictures: every picture's macroblocks from one queue (jmhip_seq_batch)" if batch else
                                        "one launch = one picture; `depth` launches overlap, each slower than alone because they share the chip") +
                                       " -- achieved / frac above are the timed region's: algorithmic bytes of its launches over its wall time"},
                "abs_diff_per_s_jm_equivalent": round(sad_ops * args.steps / dt / 1e12, 3), "valu_frac_jm_equivalent": round(sad_ops * args.steps / dt / 148.4e12, 5),
                "valu_frac_issued": round((issued_live or VALU_ISSUED_PER_PICTURE) * args.steps / dt / 148.4e12, 5),
                "valu_issued_per_picture": issued_live or VALU_ISSUED_PER_PICTURE,
                "valu_issued_source": ("measured in this run: the kernel's own counters (JMHIP_MB_PROF=11) over a P picture of this clip coded once more by a second context, outside the clock"
                                       if issued_live else VALU_ISSUED_SOURCE),
                "critical_path": {"steps": steps_chain, "us_per_step_alone": round(solo * 1e3 / steps_chain, 1) if solo else None,
                                  "note": "a macroblock waits for its left and upper-right neighbours' vectors: one picture is a chain of mb_w + 2 (mb_h - 1) macroblocks whatever the "
                                          "chip's width, with 27 of 8160 macroblocks in flight on average.  What fills the chip is the NEXT pictures: macroblock (X, r) of picture n + 1 "
                                          "only needs picture n filtered and interpolated up to macroblock (X + 5, r + 5), so consecutive pictures follow each other 16 diagonals apart "
                                          "(jm_amd/csrc/mbpipe_post.inc)"},
                "note": "dependency (latency) bound wavefronts, not an HBM stream: frac prices the algorithmic bytes against 8 TB/s as the contract asks; valu_frac_jm_equivalent counts every "
                        "candidate JM's full search visits -- the device skips the ones JM's own cost bound excludes -- over the measured v_sad_u8 peak of 148.4 T/s "
                        "(profiles/r01_valu_rates.txt).  DESIGN.md sections 0, 4"}
        out = {
            "metric": "encoded macroblocks/sec (bit-exact vs CPU JM), 1080p IPPP SR=32",
            "value": round(total_mb / dt, 1), "unit": "macroblocks/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("" if N == 1 else f"{N} GPUs, one closed GOP of the workload below per GPU (no collective on the data path; a step = one picture on EVERY GPU).  ") +
                                   "configs[1]: 1080p 4:2:0 synthetic (1920x1088 coded, 8160 MB), Baseline IPPP, one sequence: a step is the next P picture of it -- read_one_frame / pad_borders from the "
                                   "file's bytes in HBM, encode_one_macroblock_low of every macroblock on the device (FullSearch SR=32 at every block's own centre, 1 ref = the picture before, QP 28, "
                                   "mode decision, transform/quant, reconstruction), DeblockFrame, getSubImagesLuma -- " +
                                   ("the timed P pictures in one launch whose workgroups draw every picture's macroblocks from one queue, a picture 16 wavefront steps behind the one it refers to" if batch else
                                    "with up to `pictures_in_flight` consecutive pictures in flight, a launch each") + "; the timed region "
                                   "starts and ends with an idle device (pipeline fill and drain are inside it); entropy coding is the host's and is outside the step (see end_to_end)",
                       "macroblocks_per_step_per_gpu": nmb, "search_range": R, "launch": args.launch,
                       **({"pictures_per_launch": args.steps, "reference_slots": nslots, "workgroups": args.workgroups or 256} if batch else
                          {"pictures_in_flight": depth, "workgroups_per_picture": args.workgroups or min(80, 256 // depth)}),
                       "parallelism": "1 GPU" if N == 1 else f"{N} GPUs x one GOP each (JM with IDRPeriod = GOP length codes the same pictures: closed GOPs are independent)",
                       "records_equal_jm": bool(equal), "pictures_checked_against_jm": ngold,
                       "records_equal_picture_after_picture": bool(same), "pictures_checked_against_picture_after_picture": nseq,
                       "picture_after_picture_ms_per_picture": round(classic_s / nseq * 1e3, 2), "i_picture_kernel_ms_alone": round(i_ms, 3),
                       "mb_types_pskip_16x16_16x8_8x16_p8x8_i4_i16": [int(types[k]) for k in (0, 1, 2, 3, 8, 9, 10)]},
            "roofline": roof,
        }
        if batch and with_io is not None:
            out["value_with_io"] = with_io
        if slice_split is not None:
            out["configs3_slice_split"] = slice_split
        cpu = None
        if not args.no_cpu_baseline and N == 1:                       # rank 0 at N = 1 only
            cpu = cpu_baseline()
            out["cpu_baseline"] = cpu
        if not args.no_end_to_end and N == 1:
            out["end_to_end"] = end_to_end(cpu)
        if not args.no_end_to_end and N == 1:
            out["configs2"] = dict(configs2_device, end_to_end=configs2_end_to_end())
            out["b_pictures"] = dict(b_pictures_leg(local, frames, src_h, nmb, args.flight), end_to_end=b_pictures_end_to_end())
            out["configs3"] = dict(configs3_end_to_end(), device=configs3_device(local))
            out["configs4"] = configs4_end_to_end()
        if args.streams > 1 and N == 1:
            out["concurrent_streams"] = concurrent_streams(args.streams, raw0, raw1, src_h, slice_prm, local, min(args.steps, 20))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
