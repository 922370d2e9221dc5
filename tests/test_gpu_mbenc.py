"""gpu: jmhip_encode_slice (jm_amd/csrc/mbpipe.hip, the RDO-off macroblock pipeline of SURVEY.md 8f row 1) through the C ABI, against
  * the committed dumps of the REAL reference encoder's encode_one_macroblock_low (tests/golden/mb_low_*.npz, oracle/ref_tap_mb.c), and
  * the oracle (oracle/jmo_mbenc.c) on seeded synthetic clips, slices, several references, both search ranges.
The device encodes whole sequences on its own: its reconstruction is deblocked on the device and becomes the next picture's reference without
leaving HBM, exactly as the product uses it; every macroblock record and every picture before and after the loop filter must be identical."""
import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, G)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mb_tap  # noqa: E402
import mbenc_util  # noqa: E402
from oracle import pyjmo  # noqa: E402

pytestmark = pytest.mark.gpu


def slice_params(L, cfg, slice_nr, ref_slots, ref_ids, disable_idc=0, epzs=None, poc_cur=0):
    """jmhip_slice_params from the oracle's configuration record (same meaning field by field) plus JM's quantiser tables.
    epzs: the EPZS switches (dict, pyjmo.EPZS_DEFAULTS' keys) when cfg.search_mode is 3; picture order counts are 2 x the picture ids."""
    p = np.zeros(1, L.SLICE_PARAMS)
    if cfg.search_mode == 1:
        p["search_mode"] = 1
    if cfg.search_mode == 3:
        p["search_mode"] = 3
        for k, v in dict(pyjmo.EPZS_DEFAULTS, **(epzs or {})).items():
            p["epzs_" + k] = v
        p["poc_cur"] = poc_cur
        for r, i in enumerate(ref_ids):
            p["poc_ref"][0, r] = 2 * i
    for k in ("slice_type", "first_mb", "num_mb", "qp", "qpc", "search_range", "num_ref", "lambda_mdfp", "max_mvd", "intra4_valid", "intra16_valid", "subpel", "start_qp"):
        p[k] = getattr(cfg, k)
    p["slice_nr"] = slice_nr
    p["symbol_mode"] = cfg.cabac
    p["lambda_mf"] = list(cfg.lambda_mf)
    p["mv_limit"] = list(cfg.mv_limit)
    p["inter_valid"] = list(cfg.inter_valid)
    p["refbits"] = list(cfg.refbits)
    for r, (s, i) in enumerate(zip(ref_slots, ref_ids)):
        p["ref_slot"][0, r] = s
        p["ref_id"][0, r] = i
    if cfg.transform8x8:
        p["transform8x8"], p["intra8_valid"] = cfg.transform8x8, cfg.intra8_valid
        for intra in range(2):
            p["q_luma8"][0, intra] = pyjmo.qparams_8x8(cfg.qp, intra, list(cfg.off8[intra]))
    for intra in range(2):
        p["q_luma"][0, intra] = pyjmo.qparams_4x4(cfg.qp, intra, list(cfg.off4[0][intra]))
        for uv in range(2):
            p["q_chroma"][0, uv, intra] = pyjmo.qparams_4x4(cfg.qpc + uv * cfg.qpc_cr_delta, intra, list(cfg.off4[1 + uv][intra]))
            p["q_chroma_dc"][0, uv, intra] = pyjmo.qparams_4x4(cfg.qpc + uv * cfg.qpc_cr_delta + 3, intra, list(cfg.off4[1 + uv][intra]))[0]     # 4:2:2: the chroma DC quantiser (qpc + 3)
    p["qpc_cr_delta"] = cfg.qpc_cr_delta
    p["df_disable_idc"] = disable_idc
    return p


class DevSeqEncoder:
    """IPPP on the device: the counterpart of mbenc_util.SeqEncoder (the oracle), same parameters."""

    def __init__(self, W, H, qp, R, num_ref, lambdas, slice_mbs=0, level_mv=(-8192, 8191, -2048, 2047), disable_idc=0, together=False, cabac=0, search_mode=-1, epzs=None, transform8x8=0, yuv_format=1, offsets=None, inter_valid=None, qpc=None, qpc_cr_delta=0, qp_p=None, keep=0):
        self.cabac = cabac
        self.qp_p = qp if qp_p is None else qp_p
        self.yuv_format, self.offsets, self.inter_valid, self.qpc, self.qpc_cr_delta = yuv_format, offsets, inter_valid, qpc, qpc_cr_delta
        self.search_mode, self.epzs = search_mode, dict(epzs or {})
        self.transform8x8 = transform8x8
        import jm_amd.lib as L
        self.together = together      # all slices of a picture in one launch (num_slices), as the adapter does for SliceMode 1
        self.L = L
        self.W, self.H, self.qp, self.R, self.num_ref, self.lambdas = W, H, qp, R, num_ref, lambdas
        self.slice_mbs, self.level_mv, self.disable_idc = slice_mbs, level_mv, disable_idc
        self.J = L.JmHip(W, H, search_range=max(R, 1), num_ref_slots=max(num_ref, keep) + 1, yuv_format=yuv_format)
        self.refs = []          # most recent first: (slot, picture id)
        self.log = []           # per picture: what it was launched with (the runs with pictures in flight replay it: tests/test_gpu_bslice.py)
        self.keep = keep        # stored reference pictures when that is more than P's list (B pictures with a longer list 1)
        self.npic = 0

    def encode(self, raw, sw, sh, timing=None):
        L, J = self.L, self.J
        nmb = (self.W // 16) * (self.H // 16)
        st = 2 if self.npic == 0 else 0
        nref = min(self.num_ref, len(self.refs)) if st == 0 else 0
        J.set_current_frame(raw, sw, sh)
        recs = np.zeros(nmb, L.MB_RECORD)
        slices = mbenc_util.slices_of(nmb, self.slice_mbs)
        for sn, (first, num) in enumerate(slices):
            lam_mf, lam_md = self.lambdas[st]
            cfg = pyjmo.mbenc_cfg(self.W, self.H, st, first, num, self.qp if st == 2 else self.qp_p, self.R, nref, lam_mf, lam_md, level_mv=self.level_mv, cabac=self.cabac, search_mode=self.search_mode, transform8x8=self.transform8x8,
                                  yuv_format=self.yuv_format, offsets=self.offsets, inter_valid=self.inter_valid, qpc=self.qpc, qpc_cr_delta=self.qpc_cr_delta)
            prm = slice_params(L, cfg, sn, [r[0] for r in self.refs[:nref]], [r[1] for r in self.refs[:nref]], self.disable_idc, self.epzs, 2 * self.npic)
            if sn == 0:
                self.log.append(dict(st=st, raw=raw, sw=sw, sh=sh, prm=prm.copy(), nslices=len(slices), pic_id=self.npic, d8=1))
            if timing is not None:
                J.enable_timing(True)
            if self.together and len(slices) > 1:
                prm["num_slices"] = len(slices)
                recs[:] = J.encode_slice_streamed(prm) if self.npic & 1 else J.encode_slice(prm)
                if timing is not None:
                    timing.append(J.last_kernel_ms(5))
                break
            recs[first:first + num] = J.encode_slice(prm)
            if timing is not None:
                timing.append(J.last_kernel_ms(5))
        pre = J.get_recon()
        J.deblock_picture_dev(1)
        post = J.get_recon()
        used = {r[0] for r in self.refs[:max(max(self.num_ref, self.keep) - 1, 0)]}
        slot = [s for s in range(max(self.num_ref, self.keep) + 1) if s not in used][0]
        J.reference_from_recon(slot)
        J.synchronize()
        self.refs.insert(0, (slot, self.npic))
        self.refs = self.refs[:max(self.num_ref, self.keep)]
        self.npic += 1
        return recs, pre, post


def as_oracle_records(recs):
    return np.frombuffer(np.ascontiguousarray(recs).tobytes(), pyjmo.MB_RECORD).copy()


def load_case(tag):
    z = np.load(os.path.join(G, f"mb_low_{tag}.npz"))
    ov = dict(s.split("=") for s in z["overrides"])
    sw, sh, W, H = [int(x) for x in z["size"]]
    lam = {2: ([int(x) for x in z["lambda_i"][:3]], int(z["lambda_i"][3])), 0: ([int(x) for x in z["lambda_p"][:3]], int(z["lambda_p"][3]))}
    return dict(z=z, sw=sw, sh=sh, W=W, H=H, lam=lam, qp=int(z["qp"]), R=int(z["search_range"]), num_ref=int(z["num_ref"]),
                slice_mbs=int(ov.get("SliceArgument", 0)) if ov.get("SliceMode", "0") == "1" else 0, mv_limit=[int(x) for x in z["mv_limit"]],
                didc=int(ov.get("DFDisableRefPSlice", 0)), nfr=len(z["slice_type"]), records=mb_tap.widen(z["records"]), cabac=int(ov.get("SymbolMode", 0)),
                search_mode={-1: 0, 0: 1, 3: 3}[int(ov.get("SearchMode", -1))], epzs={k: int(ov[n]) for k, n in EPZS_KEYS.items() if n in ov}, t8=int(ov.get("Transform8x8Mode", 0)),
                yuv=int(z["yuv_format"]) if "yuv_format" in z.files else 1,
                offsets=pyjmo.load_q_offsets(os.path.join(G, "q_offset.cfg")) if ov.get("OffsetMatrixPresentFlag", "0") == "1" else None,
                inter_valid=[int(ov.get(k, 1)) for k in PSLICE_KEYS] if any(k in ov for k in PSLICE_KEYS) else None,
                qpc=int(z["qpc"]), qpc_cr_delta=(int(z["qpc_v"]) - int(z["qpc"])) if "qpc_v" in z.files else 0, qp_p=int(z["qp_p"]) if "qp_p" in z.files else None)


PSLICE_KEYS = ("PSliceSkip", "PSliceSearch16x16", "PSliceSearch16x8", "PSliceSearch8x16", "PSliceSearch8x8", "PSliceSearch8x4", "PSliceSearch4x8", "PSliceSearch4x4")
EPZS_KEYS = dict(pattern="EPZSPattern", dual="EPZSDualRefinement", fixed="EPZSFixedPredictors", aggressive="EPZSAggressiveWindow", temporal="EPZSTemporal",
                 spatial_mem="EPZSSpatialMem", blocktype="EPZSBlockType", min_scale="EPZSMinThresScale", med_scale="EPZSMedThresScale", max_scale="EPZSMaxThresScale",
                 sub_scale="EPZSSubPelThresScale")


def clip_bytes(tag, c):
    clip = str(c["z"]["clip"]) if "clip" in c["z"].files else ""
    if clip == "syn422":
        import synclip
        import tempfile
        with tempfile.TemporaryDirectory() as t:
            synclip.syn1080p422(os.path.join(t, "s.yuv"), c["nfr"])
            data = np.fromfile(os.path.join(t, "s.yuv"), np.uint8)
        assert hashlib.md5(data.tobytes()).hexdigest() == str(c["z"]["clip_md5"])
        return data
    if clip.startswith("motion"):
        import synth_motion
        data = np.concatenate(synth_motion.motion_clip(c["sw"], c["sh"], c["nfr"], int(clip.split(":")[1]), yuv422=clip.startswith("motion422")))
        assert hashlib.md5(data.tobytes()).hexdigest() == str(c["z"]["clip_md5"]), "the generated clip is not the one the golden records were made from"
        return data
    if tag in ("g2r", "g6r", "g6e") or clip == "True":
        import tempfile
        import bench
        with tempfile.TemporaryDirectory() as t:
            bench.write_yuv(os.path.join(t, "s.yuv"), c["nfr"])
            return np.fromfile(os.path.join(t, "s.yuv"), np.uint8)
    return np.fromfile(os.path.join(G, "foreman_part_qcif_422.yuv" if c["yuv"] == 2 else "foreman_part_qcif.yuv"), np.uint8)


def first_difference(want, got):
    bad = [k for k in range(len(want)) if got[k].tobytes() != want[k].tobytes()]
    return (len(bad), bad[:6], mb_tap.diff_fields(want[bad[0]], got[bad[0]]), want[bad[0]], got[bad[0]]) if bad else None


@pytest.mark.parametrize("tag", ["q1r", "q5r", "q4r", "q4s", "g2r", "g6r", "g6e", "q1c", "q0c", "q0r", "q1e", "m5e", "m2c", "m3p", "m2t", "g3e", "q1h", "q2hc", "m3h", "m2he", "m1hq", "g3h", "q5f", "m5f", "m3fh", "g5f", "q5y", "q2yv", "m3y", "m2yq", "g4y", "m2pd", "q1pd", "m3pe", "m2cq", "m2yc", "m3fl", "m3fm", "m2sl", "m2el", "m2es", "m5es"])
def test_encode_slice_equals_the_reference_encoder(tag):
    """The device against what JM's own encode_one_macroblock_low left behind: QCIF with one / five references, three slices, slices that start
    mid-row with two references and DFDisableIdc = 2, and BASELINE configs[1] with RDO off at 1920x1080 (SURVEY 8c G2r, 16 320 macroblocks).
    EPZS (SearchMode 3): the reference's clip (q1e), five references (m5e), CABAC + slices that start mid-row (m2c), the other patterns and window set
    (m3p), every optional predictor set off (m2t), and BASELINE configs[2]'s search at 1920x1080 (g3e: Main profile, CABAC, 24 480 macroblocks).
    High profile (Transform8x8Mode 1: transform decisions, the tr8x8 pass of P8x8, Intra8x8): CAVLC (q1h, m1hq), CABAC (q2hc, m3h), with EPZS (m2he, m1hq),
    and BASELINE configs[2] as stated at 1920x1080 (g3h: CABAC, 8x8 transform on, EPZS).
    Fast full search (SearchMode 0, encoder_baseline.cfg's): QCIF with five references (q5f), the motion clip (m5f), High profile (m3fh), and 1920x1080 with
    up to three references (g5f: 32 640 macroblocks).
    4:2:2 (High 4:2:2 profile; 8 x 16 chroma samples per macroblock, the 2x4 DC transform, q_offset.cfg's quantiser offsets): BASELINE configs[4] but for RDO / adaptive
    rounding / B pictures on the reference's clip (q5y) and at 1920x1080 (g4y: 24 480 macroblocks), CAVLC with the 4x4 transform only (q2yv), EPZS at QP 36 with
    slices that start mid-row (m3y), QP 12 (m2yq)."""
    c = load_case(tag)
    enc = DevSeqEncoder(c["W"], c["H"], c["qp"], c["R"], c["num_ref"], c["lam"], c["slice_mbs"], c["mv_limit"], c["didc"], cabac=c["cabac"],
                        search_mode=c["search_mode"], epzs=c["epzs"], transform8x8=c["t8"], yuv_format=c["yuv"], offsets=c["offsets"], inter_valid=c["inter_valid"], qpc=c["qpc"] if c["qp_p"] in (None, c["qp"]) else None, qpc_cr_delta=c["qpc_cr_delta"], qp_p=c["qp_p"])
    nmb = (c["W"] // 16) * (c["H"] // 16)
    data = clip_bytes(tag, c)
    fs = c["sw"] * c["sh"] * (4 if c["yuv"] == 2 else 3) // 2
    z = c["z"]
    for n in range(c["nfr"]):
        recs, pre, post = enc.encode(data[n * fs:(n + 1) * fs], c["sw"], c["sh"])
        got = mb_tap.canonical(as_oracle_records(recs))
        want = c["records"][n * nmb:(n + 1) * nmb]
        d = first_difference(want, got)
        assert d is None, (tag, n, d)
        for p, m in zip(pre, z["md5_pre_deblock"][n]):
            assert hashlib.md5(np.ascontiguousarray(p).tobytes()).hexdigest() == m, (tag, n, "reconstruction before the loop filter")


def synthetic_clip(W, H, nfr, seed):
    """moving blurred blocks + noise (the generator of SURVEY Appendix A at a small size), planar 4:2:0 bytes per frame"""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, ((H + 64) // 8 + 1, (W + 64) // 8 + 1)).astype(np.float64)
    big = np.kron(base, np.ones((8, 8)))
    k = 5
    pad = np.pad(big, k // 2, mode="edge")
    blur = sum(pad[i:i + big.shape[0], j:j + big.shape[1]] for i in range(k) for j in range(k)) / (k * k)
    frames = []
    for n in range(nfr):
        y = blur[2 * n:2 * n + H, 3 * n:3 * n + W] + rng.normal(0, 2 + 3 * (seed % 3), (H, W))
        y = np.clip(np.rint(y), 0, 255).astype(np.uint8)
        yd = y.reshape(H // 2, 2, W // 2, 2).mean(axis=(1, 3))
        u = np.clip(np.rint(128 + 0.25 * (yd - 128)), 0, 255).astype(np.uint8)
        v = np.clip(np.rint(128 - 0.25 * (yd - 128)), 0, 255).astype(np.uint8)
        frames.append(np.concatenate([y.ravel(), u.ravel(), v.ravel()]))
    return frames


LAMBDAS = {2: ([192, 192, 192], 192), 0: ([192, 192, 192], 192)}     # JM's tables at QP 28 with RDOptimization = 0 (tests/golden/mb_low_*.npz)


@pytest.mark.parametrize("W,H,R,num_ref,slice_mbs,qp,seed", [
    (320, 192, 32, 3, -50, 36, 3),    # negative slice size: the slices of a picture in ONE launch (num_slices), their wavefronts side by side
    (208, 160, 8, 1, -13, 20, 4),
    (256, 128, 16, 2, -100, 28, 6),   # the last slice shorter than the others
    (176, 144, 16, 1, 0, 28, 1),
    (64, 48, 32, 2, 0, 28, 2),        # a picture smaller than the search window: every window clamps on all sides
    (320, 192, 32, 3, 50, 36, 3),     # slices that start mid-row, three references, coarse quantiser
    (208, 160, 8, 1, 13, 20, 4),      # one slice per macroblock row, fine quantiser
    (16, 16, 16, 1, 0, 28, 5),        # a single macroblock
    (240, 176, 8, 5, 0, 28, 8),       # five references at SearchRange 8 ...
    (240, 176, 16, 5, 0, 32, 9),      # ... and 16, beyond QCIF
    (192, 128, 32, 4, 0, 24, 10),     # four references at SearchRange 32
])
def test_encode_slice_vs_oracle(W, H, R, num_ref, slice_mbs, qp, seed):
    f = int(192 * 2 ** ((qp - 28) / 6))          # any positive factors serve: they are inputs of both sides
    lam = LAMBDAS if qp == 28 else {2: ([f] * 3, f), 0: ([f, f + 3, f + 5], f + 1)}
    nfr = 3 if num_ref < 3 else (4 if num_ref < 4 else num_ref + 1)
    frames = synthetic_clip(W, H, nfr, seed)
    together, slice_mbs = slice_mbs < 0, abs(slice_mbs)
    dev = DevSeqEncoder(W, H, qp, R, num_ref, lam, slice_mbs, together=together)
    ora = mbenc_util.SeqEncoder(W, H, qp, R, num_ref, lam, slice_mbs)
    for n, raw in enumerate(frames):
        recs, pre, post = dev.encode(raw, W, H)
        cur = pyjmo.load_frame(raw, W, H, W, H, 1)
        orecs, _, opre, opost = ora.encode(cur)
        d = first_difference(mb_tap.canonical(orecs), mb_tap.canonical(as_oracle_records(recs)))
        assert d is None, (n, d)
        for a, b in zip(pre, opre):
            assert np.array_equal(a, b.astype(np.uint8)), (n, "reconstruction before the loop filter")
        for a, b in zip(post, opost):
            assert np.array_equal(a, b.astype(np.uint8)), (n, "reconstruction after the loop filter")


def hard_clip(kind, W, H, nfr, seed):
    """content chosen against the search's pruning and tie-breaking: everything equal, nothing alike, exact periodic ties, zero distortion at the predictor"""
    rng = np.random.default_rng(seed)
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    frames = []
    still = rng.integers(0, 256, (H, W))
    for n in range(nfr):
        if kind == "flat":
            y = np.full((H, W), 128)
            y[8 + 3 * n:16 + 3 * n, 20 + 5 * n:28 + 5 * n] = 230          # one small object crossing macroblock borders
        elif kind == "noise":
            y = rng.integers(0, 256, (H, W))
        elif kind == "stripes":
            y = 40 + 170 * (((xs + n) // 2) % 2) + 20 * (((ys + 2 * n) // 8) % 2)      # period 4 across, 16 down: whole families of equal SADs
        elif kind == "still":
            y = still
        elif kind == "chroma_step":
            y = still                                                       # luma predicts itself; the chroma planes jump by 240 (below)
        elif kind == "ramp":
            y = (3 * xs + 2 * ys + 7 * n) % 256                             # a gradient: SAD grows linearly with the displacement
        y = np.clip(y, 0, 255).astype(np.uint8)
        yd = y.reshape(H // 2, 2, W // 2, 2).mean(axis=(1, 3))
        u = np.clip(np.rint(128 + 0.25 * (yd - 128)), 0, 255).astype(np.uint8)
        v = np.clip(np.rint(128 - 0.25 * (yd - 128)), 0, 255).astype(np.uint8)
        if kind == "chroma_step":                                          # chroma DC levels beyond CAVLC's 2063 at QP 0: clamped (CAVLC) or not (CABAC)
            u[:] = 250 if n == 0 else 10
            v[:] = 250 if n == 0 else 10
        frames.append(np.concatenate([y.ravel(), u.ravel(), v.ravel()]))
    return frames


@pytest.mark.parametrize("kind,R,num_ref,qp", [("flat", 32, 1, 44), ("flat", 16, 2, 28), ("noise", 32, 1, 12), ("noise", 16, 2, 40), ("stripes", 32, 2, 28),
                                               ("stripes", 8, 1, 20), ("still", 32, 1, 36), ("still", 16, 2, 16), ("ramp", 32, 1, 28), ("ramp", 16, 1, 48),
                                               ("chroma_step", 16, 1, 0), ("chroma_step", 16, 1, -1), ("noise", 16, 1, -1)])   # qp -1: QP 0 with CABAC
def test_encode_slice_vs_oracle_hard_content(kind, R, num_ref, qp):
    """The search skips candidates by JM's own cost bound (rows near the predictor, then only what the best cost so far leaves: mbpipe.hip fs_wave);
    the result must stay JM's whatever the content does to that bound: pictures where every candidate ties (flat, periodic stripes: the spiral
    order decides), where nothing matches (noise: the bound excludes nothing), where the distortion at the predictor is zero (a still picture:
    the bound excludes almost everything), and a gradient; lambda from very small to very large."""
    W, H = 96, 80
    cabac, qp = int(qp < 0), max(qp, 0)
    f = max(1, int(192 * 2 ** ((qp - 28) / 6)))
    lam = {2: ([f] * 3, f), 0: ([f, f + 3, f + 5], f + 1)}
    frames = hard_clip(kind, W, H, 3 if num_ref == 1 else 4, 11)
    dev = DevSeqEncoder(W, H, qp, R, num_ref, lam, 0, cabac=cabac)
    ora = mbenc_util.SeqEncoder(W, H, qp, R, num_ref, lam, 0, cabac=cabac)
    for n, raw in enumerate(frames):
        recs, pre, post = dev.encode(raw, W, H)
        orecs, _, opre, opost = ora.encode(pyjmo.load_frame(raw, W, H, W, H, 1))
        d = first_difference(mb_tap.canonical(orecs), mb_tap.canonical(as_oracle_records(recs)))
        assert d is None, (kind, n, d)
        for a, b in zip(post, opost):
            assert np.array_equal(a, b.astype(np.uint8)), (kind, n, "reconstruction after the loop filter")
        if kind == "chroma_step" and n == 1:                               # the case is what it claims to be
            big = int(np.abs(recs["chroma_dc"].astype(int)).max())
            assert big == 2063 if not cabac else big > 2063, (kind, n, big, cabac)


@pytest.mark.parametrize("lam_f,kind", [(13999, "noise"), (14000, "ramp"), (60000, "noise"), (60000, "stripes"), (60000, "still")])
def test_encode_slice_key_limits(lam_f, kind):
    """jmhip_encode_slice takes lambda factors up to 60000; the searches' 32-bit (cost << 7 | rank) keys and the sub-pel scans' (cost << 4 | position)
    keys must hold them: cost < 2^25 resp. 2^27 (SAD << 5 < 2^21, rate = lambda x at most 46 bits).  13999 / 14000 is where the per-call kernels hand
    over to their 64-bit path; the pipeline has one path, pinned here against the oracle at the largest factors it accepts."""
    W, H, R = 96, 80, 32
    lam = {2: ([lam_f] * 3, lam_f), 0: ([lam_f, lam_f - 1, lam_f], lam_f)}
    frames = hard_clip(kind, W, H, 3, 5)
    dev = DevSeqEncoder(W, H, 40, R, 2, lam, 0)
    ora = mbenc_util.SeqEncoder(W, H, 40, R, 2, lam, 0)
    for n, raw in enumerate(frames):
        recs, pre, post = dev.encode(raw, W, H)
        orecs, _, opre, opost = ora.encode(pyjmo.load_frame(raw, W, H, W, H, 1))
        d = first_difference(mb_tap.canonical(orecs), mb_tap.canonical(as_oracle_records(recs)))
        assert d is None, (lam_f, kind, n, d)


@pytest.mark.parametrize("W,H,R,num_ref,slice_mbs,qp,seed,epzs", [
    (208, 160, 32, 3, 0, 28, 31, {}),                                   # the shipped switches
    (208, 160, 16, 2, 26, 34, 32, dict(pattern=1, dual=1)),             # square / small diamond, slices of two macroblock rows
    (64, 48, 32, 2, 0, 28, 33, dict(pattern=3, dual=4, fixed=1)),       # a picture smaller than the search range: every range check bites
    (176, 144, 8, 5, -33, 22, 34, dict(pattern=5, dual=5, aggressive=1)),   # five references, the extended window set (95 predictors), slices in one launch
    (16, 16, 16, 1, 0, 28, 35, {}),                                     # a single macroblock
    (320, 64, 32, 1, 0, 40, 36, dict(temporal=0, blocktype=0)),
    (208, 160, 32, 3, 0, 28, 37, dict(waves=8)),                        # the eight-wave form of the P slices forced on (by default: four waves, two workgroups per compute unit)
    (208, 160, 16, 6, 0, 28, 38, {}),                                   # six references: two workgroups' LDS no longer fit a compute unit -- the eight-wave form by itself
])
def test_encode_slice_epzs_vs_oracle(W, H, R, num_ref, slice_mbs, qp, seed, epzs, monkeypatch):
    epzs = dict(epzs)
    if epzs.pop("waves", 0) == 8:
        monkeypatch.setenv("JMHIP_EPZS_WAVES", "8")
    """EPZS inside the pipeline (k_mb_pipe_epzs) against the oracle's restatement (pinned to the real encoder by the m* / q1e / g3e records) on
    tests/golden/synth_motion.py's clips: objects with their own velocities, so predictor sets, early exits, both refinement rounds and several
    references all get used."""
    import synth_motion
    f = int(192 * 2 ** ((qp - 28) / 6))
    lam = LAMBDAS if qp == 28 else {2: ([f] * 3, f), 0: ([f, f + 3, f + 5], f + 1)}
    nfr = max(4, num_ref + 2)
    frames = synth_motion.motion_clip(W, H, nfr, seed) if W >= 64 and H >= 48 else synthetic_clip(W, H, nfr, seed)
    together, slice_mbs = slice_mbs < 0, abs(slice_mbs)
    dev = DevSeqEncoder(W, H, qp, R, num_ref, lam, slice_mbs, together=together, search_mode=3, epzs=epzs)
    ora = mbenc_util.SeqEncoder(W, H, qp, R, num_ref, lam, slice_mbs, search_mode=3, epzs=epzs)
    for n, raw in enumerate(frames):
        recs, pre, post = dev.encode(raw, W, H)
        orecs, _, opre, opost = ora.encode(pyjmo.load_frame(raw, W, H, W, H, 1))
        d = first_difference(mb_tap.canonical(orecs), mb_tap.canonical(as_oracle_records(recs)))
        assert d is None, (n, d)
        for a, b in zip(post, opost):
            assert np.array_equal(a, b.astype(np.uint8)), (n, "reconstruction after the loop filter")
    assert all(a == 0 for _, a in ora.epzs_stats)


def test_sequences_side_by_side_on_their_own_streams():
    """Three contexts, each on its own HIP stream with a share of the chip (jmhip_set_pipeline_workgroups), their slices launched without waiting
    for each other: every sequence's records and reconstruction equal what it produces alone with the default width.  (With one workgroup a
    slice still completes: tickets are drawn only by running workgroups.)"""
    import torch
    W, H, R = 208, 160, 16
    clips = [synthetic_clip(W, H, 3, 20 + k) for k in range(3)]
    alone = []
    for k in range(3):
        enc = DevSeqEncoder(W, H, 28, R, 1, LAMBDAS, 0)
        alone.append([enc.encode(raw, W, H) for raw in clips[k]])
        enc.J.close()
    streams = [torch.cuda.Stream() for _ in range(3)]
    encs = [DevSeqEncoder(W, H, 28, R, 1, LAMBDAS, 0) for _ in range(3)]
    for e, st, wg in zip(encs, streams, (1, 7, 40)):
        e.J.set_stream(st.cuda_stream)
        e.J.set_pipeline_workgroups(wg)
    L = encs[0].L
    nmb = (W // 16) * (H // 16)
    for n in range(3):
        prms = []
        for k, e in enumerate(encs):                       # everything of picture n queued on every stream before anything is read back
            st = 2 if n == 0 else 0
            e.J.set_current_frame(clips[k][n], W, H)
            cfg = pyjmo.mbenc_cfg(W, H, st, 0, nmb, 28, R, min(n, 1), *LAMBDAS[st])
            prm = slice_params(L, cfg, 0, [r[0] for r in e.refs[:min(n, 1)]], [r[1] for r in e.refs[:min(n, 1)]], 0)
            e.J.encode_slice_dev(prm)
            prms.append(prm)
        for k, e in enumerate(encs):
            e.J.synchronize()
            recs = e.J.encode_slice(prms[k])               # the same launch once more, blocking: the records (the first launch's are device-resident only)
            want = alone[k][n][0]
            assert recs.tobytes() == want.tobytes(), (k, n)
            e.J.deblock_picture_dev(1)
            post = e.J.get_recon()
            for a, b in zip(post, alone[k][n][2]):
                assert np.array_equal(a, b), (k, n)
            slot = n & 1
            e.J.reference_from_recon(slot)
            e.J.synchronize()
            e.refs = [(slot, n)]
    for e in encs:
        e.J.close()


@pytest.mark.parametrize("W,H,n_slices,world,seed", [(640, 368, 8, 2, 41), (320, 240, 4, 4, 42), (3840, 2160, 8, 8, 43)])
def test_slices_dealt_to_ranks_equal_the_one_launch_picture(W, H, n_slices, world, seed):
    """bench.py's N > 1 leg (BASELINE configs[3]: 2160p, 8 slices of 4080 macroblocks = bands of 17 ... 17, 16 rows; DFDisableIdc 0) without the collective:
    every "rank" is a context of its own that codes only its slices of the P picture; its records, its rows of the un-deblocked reconstruction and its rows
    of the loop filter's side information must equal the ones of the launch that codes all slices at once, and the picture assembled from the ranks' rows
    (what jm_amd.shard.BandGather leaves in every rank's buffers) must deblock to the same picture."""
    import torch
    import bench
    from jm_amd import shard
    R, qp = 16 if W < 3840 else 32, 28
    mbw, mbh = W // 16, H // 16
    nmb = mbw * mbh
    per = shard.slice_argument(mbh, mbw, n_slices)
    spr = n_slices // world
    k = -(-mbh // n_slices) * spr                            # macroblock rows per rank
    if W == 3840:
        import synclip
        import tempfile
        with tempfile.TemporaryDirectory() as t:
            synclip.syn2160p(os.path.join(t, "s.yuv"), 2)
            data = np.fromfile(os.path.join(t, "s.yuv"), np.uint8)
        clip = [data[:W * H * 3 // 2], data[W * H * 3 // 2:]]
    else:
        clip = synthetic_clip(W, H, 2, seed)

    def planes_of(J):
        py, pitch, pu, pv, pc = J.recon_planes_dev()
        pm, nm, po, no = J.deblock_side_info_dev()
        dev = torch.device("cuda", 0)
        return [(torch.as_tensor(bench._DevMem(py, pitch * H), device=dev).view(H, pitch), 16 * k),
                (torch.as_tensor(bench._DevMem(pu, pc * H // 2), device=dev).view(H // 2, pc), 8 * k),
                (torch.as_tensor(bench._DevMem(pv, pc * H // 2), device=dev).view(H // 2, pc), 8 * k),
                (torch.as_tensor(bench._DevMem(pm, nm), device=dev).view(mbh, nm // mbh), k),
                (torch.as_tensor(bench._DevMem(po, no), device=dev).view(4 * mbh, no // (4 * mbh)), 4 * k)]

    def coded(first, num, num_slices, slice_nr):
        """a fresh context: the I picture in all its slices, then the P picture's macroblocks [first, ...) only; (context, records)"""
        J = L.JmHip(W, H, search_range=R, num_ref_slots=2, yuv_format=1)
        J.set_current_frame(clip[0], W, H)
        cfg = pyjmo.mbenc_cfg(W, H, 2, 0, per, qp, R, 0, *LAMBDAS[2])
        prm = slice_params(L, cfg, 0, [], [])
        prm["num_slices"] = n_slices
        J.encode_slice_dev(prm)
        J.deblock_picture_dev(1)
        J.reference_from_recon(0)
        J.set_current_frame(clip[1], W, H)
        cfg = pyjmo.mbenc_cfg(W, H, 0, first, num, qp, R, 1, *LAMBDAS[0])
        prm = slice_params(L, cfg, slice_nr, [0], [0])
        prm["num_slices"] = num_slices
        recs = J.encode_slice(prm)
        J.synchronize()
        return J, recs

    import jm_amd.lib as L
    J, all_recs = coded(0, per, n_slices, 0)
    want = [t.cpu() for t, _ in planes_of(J)]                # un-deblocked reconstruction + side information of the one-launch picture
    J.deblock_picture_dev(1)
    post = J.get_recon()
    J.close()
    assert len(all_recs) == nmb
    got = [torch.full_like(t, 0xEE) for t in want]
    for r in range(world):
        first = r * spr * per
        mine = min(spr * per, nmb - first)
        Jr, recs = coded(first, per if spr > 1 else mine, spr, r * spr)
        assert len(recs) == mine and recs.tobytes() == all_recs[first:first + mine].tobytes(), ("records of rank", r)
        pl = planes_of(Jr)
        for (t, kk), w, g in zip(pl, want, got):
            assert torch.equal(t[r * kk:(r + 1) * kk].cpu(), w[r * kk:(r + 1) * kk]), ("rows of rank", r, kk)
            g[r * kk:(r + 1) * kk] = t[r * kk:(r + 1) * kk].cpu()
        if r < world - 1:
            Jr.close()
    for g, w in zip(got, want):
        assert torch.equal(g, w)                                 # every row has exactly one owner
    for (t, _), g in zip(pl, got):                               # the last rank's buffers after the exchange
        t.copy_(g)
    torch.cuda.synchronize()
    Jr.deblock_picture_dev(1)
    for a, b in zip(Jr.get_recon(), post):
        assert np.array_equal(a, b)
    Jr.close()


@pytest.mark.parametrize("W,H,n_slices,world,seed", [(640, 368, 8, 2, 51), (320, 240, 4, 4, 52), (1920, 1088, 8, 8, 53)])
def test_allgather_bands_through_the_c_abi(W, H, n_slices, world, seed):
    """jmhip_allgather_bands (SURVEY 8b viii): one process, `world` contexts -- all of them on this box's one device, which is what a single GPU allows; between devices the
    same calls are peer copies over xGMI --, each codes its share of a picture's slices, the call gives every context every band, and then EVERY context deblocks and
    interpolates the whole picture: filtered planes and sub-pel planes equal the ones of the context that coded all slices itself.  Two pictures, so that the second
    one's searches read a reference every context built from exchanged bands."""
    import jm_amd.lib as L
    from jm_amd import shard
    R, qp = 16, 28
    mbw, mbh = W // 16, H // 16
    nmb = mbw * mbh
    per = shard.slice_argument(mbh, mbw, n_slices)
    spr = n_slices // world
    band = -(-mbh // n_slices) * spr
    clip = synthetic_clip(W, H, 3, seed)

    def prm_for(st, first, num, num_slices, slice_nr, nref):
        cfg = pyjmo.mbenc_cfg(W, H, st, first, num, qp, R, nref, *LAMBDAS[st])
        q = slice_params(L, cfg, slice_nr, [n & 1 for n in range(nref)] if nref else [], [0] * nref)
        q["num_slices"] = num_slices
        return q
    one = L.JmHip(W, H, search_range=R, num_ref_slots=2, yuv_format=1)
    ctxs = [L.JmHip(W, H, search_range=R, num_ref_slots=2, yuv_format=1) for _ in range(world)]
    for n, raw in enumerate(clip):
        st, nref = (2, 0) if n == 0 else (0, 1)
        one.set_current_frame(raw, W, H)
        q = prm_for(st, 0, per, n_slices, 0, nref)
        if nref:
            q["ref_slot"][0, 0] = (n - 1) & 1
        all_recs = one.encode_slice(q)
        one.deblock_picture_dev(1)
        want_post = one.get_recon()
        one.reference_from_recon(n & 1)
        want_planes = one.get_subplanes(n & 1)
        for r, J in enumerate(ctxs):
            first = r * spr * per
            mine = min(spr * per, nmb - first)
            J.set_current_frame(raw, W, H)
            q = prm_for(st, first, per if spr > 1 else mine, spr, r * spr, nref)
            if nref:
                q["ref_slot"][0, 0] = (n - 1) & 1
            recs = J.encode_slice(q)
            assert recs.tobytes() == all_recs[first:first + mine].tobytes(), ("picture", n, "records of context", r)
        L.allgather_bands(ctxs, band)
        for r, J in enumerate(ctxs):
            J.deblock_picture_dev(1)
            for a, b in zip(J.get_recon(), want_post):
                assert np.array_equal(a, b), ("picture", n, "filtered picture of context", r)
            J.reference_from_recon(n & 1)
            assert np.array_equal(J.get_subplanes(n & 1), want_planes), ("picture", n, "sub-pel planes of context", r)
    with pytest.raises(L.JmHipError):
        L.allgather_bands(ctxs, band + 1 if world * (band + 1) - (band + 1) >= mbh else mbh + 1)      # bands that do not make up the picture
    for J in ctxs + [one]:
        J.close()


def test_streamed_records_equal_the_blocking_call():
    """jmhip_encode_slice_begin / jmhip_slice_record / jmhip_encode_slice_end (what the adapter uses: JM's entropy coder reads each record in raster
    order while the device is still encoding) hands over the same records as jmhip_encode_slice, picture after picture, with mid-row slices."""
    import jm_amd.lib as L
    W, H, R = 320, 192, 16
    frames = synthetic_clip(W, H, 3, 9)
    J = L.JmHip(W, H, search_range=R, num_ref_slots=2, yuv_format=1)
    nmb = (W // 16) * (H // 16)
    for n, raw in enumerate(frames):
        J.set_current_frame(raw, W, H)
        st, nref = (2, 0) if n == 0 else (0, 1)
        for sn, (first, num) in enumerate(mbenc_util.slices_of(nmb, 70)):
            cfg = pyjmo.mbenc_cfg(W, H, st, first, num, 28, R, nref, LAMBDAS[st][0], LAMBDAS[st][1])
            prm = slice_params(L, cfg, sn, [n & 1], [n - 1])
            a = J.encode_slice(prm)
            b = J.encode_slice_streamed(prm)
            assert a.tobytes() == b.tobytes(), (n, sn)
        J.deblock_picture_dev(1)
        J.reference_from_recon((n + 1) & 1)
    with pytest.raises(L.JmHipError):
        J._ck(J.lib.jmhip_encode_slice_end(J.h))             # nothing to end


def test_encode_slice_rejects_what_it_does_not_cover():
    import jm_amd.lib as L
    J = L.JmHip(64, 48, search_range=16, num_ref_slots=2, yuv_format=1)
    cfg = pyjmo.mbenc_cfg(64, 48, 0, 0, 12, 28, 16, 1, [187] * 3, 1097)
    prm = slice_params(L, cfg, 0, [0], [0])
    with pytest.raises(L.JmHipError):       # no current picture
        J.encode_slice(prm)
    J.set_current_frame(np.zeros(64 * 48 * 3 // 2, np.uint8), 64, 48)
    with pytest.raises(L.JmHipError):       # reference slot without chroma planes
        J.encode_slice(prm)
    bad = prm.copy(); bad["slice_type"] = 1
    with pytest.raises(L.JmHipError):
        J.encode_slice(bad)
    bad = prm.copy(); bad["num_mb"] = 13
    with pytest.raises(L.JmHipError):
        J.encode_slice(bad)
    bad = prm.copy(); bad["search_range"] = 32
    with pytest.raises(L.JmHipError):
        J.encode_slice(bad)
    bad = prm.copy(); bad["symbol_mode"] = 2
    with pytest.raises(L.JmHipError):       # CAVLC or CABAC, nothing else
        J.encode_slice(bad)
    with pytest.raises(L.JmHipError):
        J.set_pipeline_workgroups(-1)
    J4 = L.JmHip(64, 48, search_range=16, num_ref_slots=1, yuv_format=2)
    with pytest.raises(L.JmHipError):
        J4.encode_slice(prm)


def test_fast_full_search_is_refused_when_the_level_cuts_into_the_range():
    """me_fullfast.c:318-326 clips the search centre to limit -+ range after making sure (0,0) is inside; with a vertical limit of 255 quarter-pels (levels 1 / 1b; JM also maps
    level_idc 11 of the High profiles there) and SearchRange 32 the centre can become 127 -- off the sample grid --, the (0,0) position is then not found (:354-365) and JM uses the
    pos_00 an EARLIER macroblock left: raster-order state the wavefront cannot reproduce, so the call is refused and the adapter turns such sequences away (found by
    tests/fuzz_dropin.py: 3 of 2051 configurations).  SearchRange 16 with the same limits, and level 1.1's own limits (goldens m3fl, m3fm), are served."""
    import jm_amd.lib as L
    W, H, tight = 80, 112, (-8192, 8191, -256, 255)
    clip = synthetic_clip(W, H, 2, 77)
    J = L.JmHip(W, H, search_range=32, num_ref_slots=2, yuv_format=1)
    J.set_current_frame(clip[0], W, H)
    nmb = (W // 16) * (H // 16)
    J.encode_slice_dev(slice_params(L, pyjmo.mbenc_cfg(W, H, 2, 0, nmb, 28, 32, 0, *LAMBDAS[2], level_mv=tight, search_mode=1), 0, [], []))
    J.deblock_picture_dev(1); J.reference_from_recon(0)
    J.set_current_frame(clip[1], W, H)
    with pytest.raises(L.JmHipError, match="cuts into the search range"):
        J.encode_slice_dev(slice_params(L, pyjmo.mbenc_cfg(W, H, 0, 0, nmb, 28, 32, 1, *LAMBDAS[0], level_mv=tight, search_mode=1), 0, [0], [0]))
    for R, mode in ((16, 1), (32, 0), (32, 3)):                  # the same limits with SearchRange 16, and with full search / EPZS, are served
        J.encode_slice_dev(slice_params(L, pyjmo.mbenc_cfg(W, H, 0, 0, nmb, 28, R, 1, *LAMBDAS[0], level_mv=tight, search_mode=mode), 0, [0], [0]))
    J.synchronize()
    J.close()


def test_random_configurations_vs_oracle():
    """Thirty seconds of tests/fuzz_mbenc.py: seeded random configurations (size, search mode and range, references, QP, slices, entropy mode, 8x8 transform, 4:2:0 / 4:2:2,
    quantiser offsets, clip kind) through jmhip_encode_slice and the oracle; records and reconstructions identical.  (A ten-minute run of the same script: profiles/r03_fuzz.txt.)"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_mbenc.py"), "30", "424242"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "identical to the oracle" in out, out[-2000:]
