#!/usr/bin/env python3
"""bench.py -- encoded macroblocks/sec of the MI355X hot path on synthetic 1080p (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one P frame whose inputs are already resident in HBM:
  K5  sub-pel planes of the reference      (getSubImagesLuma)
  K1-3 full search, SR=32, 41 partitions    (full_search_motion_estimation, one window job per macroblock)
  K4  9+9 sub-pel refinement, SATD          (sub_pel_motion_estimation)
  MC  luma prediction of every macroblock with its refined 16x16 vector (luma_prediction), device resident
  K7/8 4x4 transform/quant/reconstruct      (residual_transform_quant_luma_4x4, 16 luma blocks per macroblock) on source - prediction,
       the reconstructed blocks assembled into the picture
  the same for both chroma planes           (chroma_prediction_4x4, residual_transform_quant_chroma_4x4)
  K9/10 deblocking of that reconstructed picture, luma and chroma (DeblockFrame)
Data flows from stage to stage on the device as it does in the encoder (reference -> search -> refinement -> prediction -> residual
-> reconstruction -> loop filter).  What stays synthetic: the MV predictors (the sequential mode decision that produces them in JM
stays on the host) and the choice "every macroblock is P16x16"; the deblocking side
information is the P picture's of this very configuration as JM produced it (tests/golden/g2_sideinfo.npz).

python bench.py --gpus N --steps K --warmup W     (N > 1: launched by torch.distributed.run, one rank per GPU)
N > 1 shards one tall frame of N 1080p bands (slices) one band per GPU; each step all-gathers the
reconstructed bands (luma + chroma, one collective) over RCCL (the reference-picture exchange of SURVEY.md 8e) -- weak scaling.
Prints ONE JSON line on rank 0.
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

W, H_SRC, H = 1920, 1080, 1088
R = 32
QP = 28
# HBM bytes per launch, per stage (K5, K1-K3, K4, K7/K8, K9/K10): 2 x FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes of this command, corrected as
# MI355X_MICROARCH.md prescribes (profiles/r01_v13_kernel_stats.md).  The full search moves far less than its algorithmic 57.0 MB: neighbouring windows overlap
# and, with the XCD-aware job order, meet in the same L2
TRAFFIC_BYTES = [44921952, 12328672, 84850992, 57237008, 9934464]
MAX_VMV = 512       # level-4/5.1 vertical MV limit in pels (lencod/src/conformance.c:604-631): a search centre can sit this far away


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
            dx, dy = 3 * n, 2 * n
            y = sm[dy:dy + H_SRC, dx:dx + W] + rng.normal(0, 2, size=(H_SRC, W))
            y = np.clip(np.rint(y), 0, 255).astype(np.uint8)
            u = np.clip(np.rint(128 + 0.25 * (y[::2, ::2].astype(np.float32) - 128)), 0, 255).astype(np.uint8)
            v = np.clip(np.rint(128 - 0.25 * (y[::2, ::2].astype(np.float32) - 128)), 0, 255).astype(np.uint8)
            f.write(y.tobytes()); f.write(u.tobytes()); f.write(v.tobytes())


def cpu_baseline(max_seconds=240):
    """JM's own CPU lencod (oracle/_ref/lencod.exe, built from the reference: kind "reference"), single thread,
    on a bounded sample of the same workload: the first two frames (I + P) of the synthetic 1080p clip with
    the BASELINE.json configs[1] settings.  Falls back to the oracle's C restatement of the full search
    (kind "port") on a sample of macroblocks when the reference binary did not travel."""
    exe = os.path.join(ROOT, "oracle", "_ref", "lencod.exe")
    cfg = os.path.join(ROOT, "tests", "golden", "jm_baseline.cfg")
    if os.path.exists(exe) and os.access(exe, os.X_OK):
        with tempfile.TemporaryDirectory() as tmp:
            write_yuv(os.path.join(tmp, "syn1080p.yuv"), 2)
            args = [exe, "-d", cfg]
            for kv in ("InputFile=syn1080p.yuv", "SourceWidth=1920", "SourceHeight=1080", "OutputWidth=1920", "OutputHeight=1080",
                       "FramesToBeEncoded=2", "SearchMode=-1", "SearchRange=32", "NumberReferenceFrames=1", "LevelIDC=51",
                       "OutputFile=o.264", "ReconFile=o_rec.yuv", "TraceFile=/dev/null"):
                args += ["-p", kv]
            try:
                t0 = time.time()
                r = subprocess.run(args, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=max_seconds)
                wall = time.time() - t0
                log = r.stdout.decode(errors="replace")
                # per-frame line: "00001(P ) bits QP SnrY SnrU SnrV Time(ms) MET(ms) ..."
                m = re.search(r"^\s*0*1\(\s*P\s*\)\s+\d+\s+\d+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+(\d+)\s+(\d+)", log, re.M)
                if r.returncode == 0 and m:
                    p_ms, me_ms = int(m.group(1)), int(m.group(2))
                    return {"value": round(8160 / (p_ms / 1000.0), 1), "unit": "macroblocks/s", "cores": 1, "kind": "reference",
                            "sample": f"JM 19.0 lencod -O3, 1 thread: P frame of syn1080p (I+P encoded, {wall:.1f} s wall): "
                                      f"{p_ms} ms total, {me_ms} ms ME; FullSearch SR=32, 1 ref, RDO on, CAVLC"}
            except subprocess.TimeoutExpired:
                pass
    # port: the oracle's full search over the 41 partitions of a sample of macroblocks
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--deblock-load", choices=["real", "worst"], default="real",
                    help="deblocking side information: the P picture of configs[1] as JM produced it (default) or an intra-heavy made-up mix")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from jm_amd import JmHip
    from jm_amd.lib import ME_JOB, ME_RESULT, TQ_OUT, DB_MB, DB_MOTION, NPART

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
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    N = world
    band = shard.band_of(rank, N, (H // 16) * N)        # one tall picture of N 1080p bands, one band (slice) per GPU
    halo = shard.halo_rows(R, MAX_VMV) if N > 1 else 0               # rows of neighbouring bands a band's search windows can read: 576
    HL = H + 2 * halo                                   # rows of the local reference (own band + halos)

    stream = torch.cuda.current_stream()
    ctx = JmHip(W, HL, search_range=R, num_ref_slots=1, yuv_format=1, device=local, stream=stream.cuda_stream)

    # ---------------- synthetic inputs, resident in HBM before the timed region
    frames = synth_luma(2, seed=1234 + rank)
    def chroma_of(y):                                                   # U = 128 + (Y/2 - 128)/4, V = 128 - (Y/2 - 128)/4
        d = 0.25 * (y[::2, ::2].astype(np.float32) - 128)
        return np.clip(np.rint(128 + d), 0, 255).astype(np.uint8), np.clip(np.rint(128 - d), 0, 255).astype(np.uint8)
    ref_u, ref_v = chroma_of(frames[0])
    cur_u, cur_v = chroma_of(frames[1])
    ref_band = torch.from_numpy(frames[0]).to(dev)                      # this band's reconstructed reference: luma ...
    ref_band_u, ref_band_v = torch.from_numpy(ref_u).to(dev), torch.from_numpy(ref_v).to(dev)   # ... and 4:2:0 chroma
    ref_packed = shard.packed_band(ref_band, ref_band_u, ref_band_v) if N > 1 else None
    yuv_exchange = shard.YuvExchange(band, halo, N * H, W, N, "cpu" if one_gpu else dev) if N > 1 else None
    cur_local = np.zeros((HL, W), np.uint8); cur_local[halo:halo + H] = frames[1]
    d_cur = torch.from_numpy(cur_local).to(dev)
    ctx.set_current_dev(d_cur.data_ptr(), W)
    local_ref = torch.empty((HL, W), dtype=torch.uint8, device=dev)
    local_ref_c = torch.empty((HL // 2, W), dtype=torch.uint8, device=dev)      # U | V side by side, as the exchange delivers them

    mbw, mbh = W // 16, H // 16
    nmb = mbw * mbh
    rng = np.random.default_rng(7 + rank)
    jobs = np.zeros(nmb, ME_JOB)
    jobs["mb_x"] = np.tile(np.arange(mbw) * 16, mbh)
    jobs["mb_y"] = np.repeat(np.arange(mbh) * 16, mbw) + halo
    jobs["search_range"], jobs["lambda"], jobs["part_mask"] = R, 187, np.uint64((1 << NPART) - 1)
    # predictors: the clip's global motion (3,2) px = (12,8) quarter-pel +- 2 quarter-pels; one centre per MB
    jobs["pred"] = np.array([12, 8], np.int16) + rng.integers(-2, 3, (nmb, NPART, 2)).astype(np.int16)
    jobs["center_x"], jobs["center_y"] = 12, 8
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(nmb, -1)).to(dev)
    d_int = torch.zeros((nmb, ME_RESULT.itemsize), dtype=torch.uint8, device=dev)
    d_fin = torch.zeros((nmb, ME_RESULT.itemsize), dtype=torch.uint8, device=dev)
    rprm = ctx.refine_params(187, 187, 2, 2, 0, 0, 0)

    # transform/quant input: the band's 4x4 luma blocks, prediction = co-located reference block
    def blocks_of(img):
        return np.ascontiguousarray(img.reshape(H // 4, 4, W // 4, 4).transpose(0, 2, 1, 3).reshape(-1, 16))
    d_orig = torch.from_numpy(blocks_of(frames[1])).to(dev)
    nblk = d_orig.shape[0]
    d_tq = torch.zeros((nblk, TQ_OUT.itemsize), dtype=torch.uint8, device=dev)
    q = np.zeros((16, 3), np.int32)
    sc, ds = {0: 8192, 1: 3355, 2: 5243}, {0: 16, 1: 25, 2: 20}           # qp % 6 == 4 rows of quant_coef / dequant_coef (q_matrix.c:20-36)
    for j in range(4):
        for i in range(4):
            c = 0 if (i % 2 == 0 and j % 2 == 0) else (1 if (i % 2 and j % 2) else 2)
            q[j * 4 + i] = (342 << (15 + QP // 6 - 11), sc[c], ds[c] << 4)
    tqp = ctx.tq_params(q, QP // 6, cavlc=1, adaptive_rounding=1, adapt_rnd_weight=4)
    # chroma: both planes of every macroblock as items (2 * macroblock + plane) of 8x8 samples in rows of 8 (jmhip_tq_chroma_dev's layout)
    from jm_amd.lib import TQC_MB, TQC_OUT
    def items_of(u, v):
        a = np.zeros((nmb, 2, 16, 8), np.uint8)
        for p, pl in enumerate((u, v)):
            a[:, p, :8] = pl.reshape(mbh, 8, mbw, 8).transpose(0, 2, 1, 3).reshape(nmb, 8, 8)
        return a.reshape(2 * nmb, 128)
    d_origc = torch.from_numpy(items_of(cur_u, cur_v)).to(dev)
    d_predc = torch.zeros((2 * nmb, 128), dtype=torch.uint8, device=dev)
    cmbs = np.zeros(2 * nmb, TQC_MB); cmbs["uv"] = np.arange(2 * nmb) % 2
    d_cmbs = torch.from_numpy(cmbs.view(np.uint8).reshape(2 * nmb, -1)).to(dev)
    d_tqc = torch.zeros((2 * nmb, TQC_OUT.itemsize), dtype=torch.uint8, device=dev)
    tqcp = ctx.tqc_params(1, q, q[0], QP // 6, QP // 6, cavlc=1, adaptive_rounding=1, adapt_rnd_weight=4)   # chroma qp 28 for luma qp 28 (QP_SCALE_CR)

    # deblocking input: pre-filter reconstruction (the current frame stands in for it) + synthetic side information
    ch, cw = H // 2, W // 2
    pre_y = torch.from_numpy(frames[1]).to(dev)
    work_y, work_c = torch.empty_like(pre_y), torch.empty((2, ch, cw), dtype=torch.uint8, device=dev)
    # side information: what JM's DeblockFrame was given for the P picture of this very configuration (tests/golden/g2_sideinfo.npz,
    # captured from the reference encoder by tests/golden/make_g2_sideinfo.py): 78 % skipped macroblocks, 8 % with coefficients,
    # 0.2 % intra, 86 % of the 4x4 blocks on the clip's global motion vector.  --deblock-load worst swaps in a made-up intra-heavy mix.
    from jm_amd.lib import db_arrays_from_tap
    if args.deblock_load == "real":
        g2 = np.load(os.path.join(ROOT, "tests", "golden", "g2_sideinfo.npz"))
        mbs, mot = db_arrays_from_tap(g2["p_mbs"].astype(np.int32), g2["p_mot"].astype(np.int32))
        mbs, mot = mbs.copy(), mot.reshape(-1).copy()
    else:
        mbs = np.zeros(nmb, DB_MB)
        mbs["mb_type"] = rng.choice([0, 1, 1, 2, 3, 8, 8, 9, 10], nmb)
        mbs["qp"], mbs["qpc"] = QP, QP - 1
        mbs["cbp_blk"] = rng.integers(0, 1 << 16, nmb) * (rng.integers(0, 3, nmb) > 0)
        mbs["cbp"] = np.where(mbs["cbp_blk"] != 0, 15, 0)
        mot = np.zeros((H // 4) * (W // 4), DB_MOTION)
        mot["mv"][:, 0, :] = np.array([12, 8], np.int16) + rng.integers(-5, 6, (len(mot), 2)).astype(np.int16)
        mot["ref_id"][:, 0], mot["ref_id"][:, 1] = 0, -1
    mbs["df_disable_idc"] = 2 if N > 1 else 0
    d_mbs = torch.from_numpy(mbs.view(np.uint8).reshape(nmb, -1)).to(dev)
    d_mot = torch.from_numpy(mot.view(np.uint8).reshape(len(mot), -1)).to(dev)
    # deblock context works on the band itself (height H), not on the haloed reference
    dctx = ctx if halo == 0 else JmHip(W, H, search_range=R, num_ref_slots=1, yuv_format=1, device=local, stream=stream.cuda_stream)

    # HIP events on the launch stream around every stage of every timed step: 6 marks per step
    STAGES = ["k_subplanes (K5)", "k_me_fs_fast (K1-K3)", "k_me_refine_mb (K4)", "prediction + transform/quant + reconstruction, luma and chroma (MC, K7/K8)",
              "k_deblock_prep + k_deblock_tasks + k_deblock_sparse | k_deblock_rows (K9/K10)"]
    marks = [[torch.cuda.Event(enable_timing=True) for _ in range(7)] for _ in range(args.steps)]
    SPANS = [(0, 1), (1, 2), (2, 3), (3, 4), (5, 6)]
    d_predb = torch.zeros((nblk, 16), dtype=torch.uint8, device=dev)      # prediction in 4x4-block order, written by the MC stage

    def step(i, timed):
        if N > 1 and one_gpu:                                           # debugging path: the same exchange on host copies over gloo
            ly, lu, lv = yuv_exchange(ref_packed.cpu())
            local_ref.copy_(ly); local_ref_c[:, :W // 2].copy_(lu); local_ref_c[:, W // 2:].copy_(lv)
        elif N > 1:                                                     # reference-picture exchange over xGMI (RCCL): the one collective (luma + chroma)
            ly, lu, lv = yuv_exchange(ref_packed)
        if N > 1 and one_gpu:
            ly, lu, lv = local_ref, local_ref_c[:, :W // 2], local_ref_c[:, W // 2:]
        if timed:
            marks[i][0].record(stream)
        if N > 1:
            ctx.set_reference_dev(0, ly.data_ptr(), W)                                   # K5
            ctx.set_reference_chroma_dev(0, lu.data_ptr(), lv.data_ptr(), W)
        else:
            ctx.set_reference_dev(0, ref_band.data_ptr(), W)                             # K5
            ctx.set_reference_chroma_dev(0, ref_band_u.data_ptr(), ref_band_v.data_ptr(), W // 2)
        if timed:
            marks[i][1].record(stream)
        ctx.me_fullsearch_dev(0, d_jobs.data_ptr(), nmb, d_int.data_ptr())           # K1-K3
        if timed:
            marks[i][2].record(stream)
        ctx.me_refine_dev(0, d_jobs.data_ptr(), nmb, d_int.data_ptr(), rprm, d_fin.data_ptr())   # K4
        if timed:
            marks[i][3].record(stream)
        # P16x16 reconstruction path, device resident: prediction with each macroblock's refined 16x16 vector, residual transform /
        # quantisation / reconstruction of the sixteen 4x4 blocks, reconstructed blocks assembled into the picture the deblocking reads
        # (one launch: jmhip_mb16_recon_luma_dev = jmhip_mc_mb16_dev + jmhip_tq_luma4x4_dev + jmhip_tq_rec_to_plane_dev)
        ctx.mb16_recon_luma_dev(0, tqp, d_jobs.data_ptr(), d_fin.data_ptr(), nmb, halo, W // 4, d_orig.data_ptr(), d_tq.data_ptr(), 0,
                                work_y.data_ptr(), W)                                            # MC + K7/K8
        # the same for both chroma planes (chroma_prediction_4x4, residual_transform_quant_chroma_4x4)
        ctx.mc_mb16_chroma_dev(0, d_jobs.data_ptr(), d_fin.data_ptr(), nmb, d_predc.data_ptr())
        ctx.tq_chroma_dev(tqcp, d_cmbs.data_ptr(), d_origc.data_ptr(), d_predc.data_ptr(), 2 * nmb, d_tqc.data_ptr())
        ctx.tqc_rec_to_planes_dev(d_jobs.data_ptr(), d_tqc.data_ptr(), nmb, halo, work_c[0].data_ptr(), work_c[1].data_ptr(), cw)
        if timed:
            marks[i][4].record(stream)
        if timed:
            marks[i][5].record(stream)
        dctx.deblock_frame_dev(work_y.data_ptr(), W, work_c[0].data_ptr(), work_c[1].data_ptr(), cw, d_mbs.data_ptr(), d_mot.data_ptr(), 1)  # K9/K10
        if timed:
            marks[i][6].record(stream)

    def barrier():
        if N > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    barrier()
    dt = time.perf_counter() - t0
    if N > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stage_ms = [float(np.mean([m[a].elapsed_time(m[b]) for m in marks])) for (a, b) in SPANS]
    fs_ms = stage_ms[1]

    # sanity: the search found the clip's motion for the 16x16 partition of interior macroblocks
    res = d_fin.cpu().numpy().view(ME_RESULT).reshape(nmb)
    interior = (jobs["mb_x"] > 64) & (jobs["mb_x"] < W - 80) & (jobs["mb_y"] - halo > 64) & (jobs["mb_y"] - halo < H_SRC - 80)
    mv16 = res["best"][interior, 0]
    motion_ok = float(np.mean((np.abs(mv16["mv_x"] - 12) <= 2) & (np.abs(mv16["mv_y"] - 8) <= 2)))
    # sanity: the filtered reconstruction at the end of the chain is a faithful picture of the source (QP 28)
    def psnr(a, b):
        mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
        return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    psnr_y = psnr(work_y.cpu().numpy(), frames[1])
    psnr_u = psnr(work_c[0].cpu().numpy(), cur_u)

    if rank == 0:
        total_mb = nmb * N * args.steps
        # algorithmic bytes per launch and HBM traffic per launch from the PMC passes (profiles/r01_v13_kernel_stats.md)
        alg = [W * H + 16 * (W + 64) * (HL + 40),                               # K5: one plane in, 16 padded planes out
               (256 + (2 * R + 16) ** 2 + 328) * nmb,                           # K1-K3: SURVEY.md 8d per MB-reference: 6656 in + 328 out at R=32
               7 * 256 * 19 * nmb,                                              # K4: 18 candidate blocks + the current block, 7 block types
               (16 + 32 + 104 + 16) * nblk + (64 + 128 + 808 + 64) * 2 * nmb,   # MC + K7/K8: per luma 4x4 block 16 B reference, 32 B in, 104 B out, 16 B picture;
                                                                                # per chroma plane of a macroblock 64 B reference, 128 B in, 808 B out, 64 B picture
               int(1.5 * W * H * 2) + (192 + 28 + 16 * 16) * nmb]               # K9/K10: every sample once in, once out + records
        kernels = [{"kernel": STAGES[k], "ms": round(stage_ms[k], 4), "algorithmic_bytes": alg[k], "hbm_traffic_bytes": TRAFFIC_BYTES[k],
                    "hbm_frac": round(alg[k] / (stage_ms[k] * 1e-3) / 8e12, 5)} for k in range(len(STAGES))]
        dom = int(np.argmax(stage_ms))
        alg_bytes = alg[dom]
        roof = {"kernel": STAGES[dom], "bound": "hbm", "achieved": round(alg_bytes / (stage_ms[dom] * 1e-3) / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(alg_bytes / (stage_ms[dom] * 1e-3) / 8e12, 5), "traffic": TRAFFIC_BYTES[dom], "avg_kernel_ms": round(stage_ms[dom], 4),
                "algorithmic_bytes_per_launch": alg_bytes}
        sad_rate = nmb * (2 * R + 1) ** 2 * 256 / (fs_ms * 1e-3)
        if dom == 1:
            roof["valu_frac"] = round(sad_rate / 148.4e12, 4)
            roof["note"] = ("VALU-bound, not HBM-bound (155 abs-diff per algorithmic byte): valu_frac = achieved abs-diff/s over the measured v_sad_hi_u8 "
                            "peak of 148.4 T abs-diff/s (profiles/r01_valu_rates.txt); DESIGN.md section 3")
        elif dom == 4:
            roof["note"] = ("latency-bound by construction: JM's raster-order filter is a chain of (W/16 + H/16) macroblock steps of eight dependent edge "
                            "filters each; DESIGN.md section 3 (K9/K10)")
        out = {
            "metric": "encoded macroblocks/sec (bit-exact vs CPU JM), 1080p IPPP SR=32",
            "value": round(total_mb / dt, 1), "unit": "macroblocks/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[1]: 1080p 4:2:0 synthetic (1920x1088 coded, 8160 MB), Baseline IPPP P-frame hot path, FullSearch SR=32, 1 ref, QP 28",
                       "macroblocks_per_step_per_gpu": nmb, "search_range": R, "partitions": NPART,
                       "parallelism": "1 GPU" if N == 1 else f"{N} slices (1080p bands) one per GPU, RCCL all-gather of reconstructed bands per step",
                       "kernel_path_only": "MV predictors are synthetic inputs and every macroblock is reconstructed as P16x16 from its refined vector; the deblocking side information is "
                                           + ("the P picture's as JM produced it for this configuration" if args.deblock_load == "real" else "a made-up intra-heavy mix")
                                           + "; mode decision and entropy coding stay on the host",
                       "motion_found_frac": round(motion_ok, 4), "recon_psnr_y_db": round(psnr_y, 2), "recon_psnr_u_db": round(psnr_u, 2)},
            "roofline": roof,
            "kernels": kernels,
            "me_fullsearch": {"ms": round(fs_ms, 4), "abs_diff_per_s": round(sad_rate / 1e12, 2), "unit": "T abs-diff/s", "valu_frac": round(sad_rate / 148.4e12, 4)},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if N > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
