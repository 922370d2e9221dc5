"""gpu: B pictures through jmhip_encode_slice (jm_amd/csrc/mbpipe_b.inc: slice_type 1) against the committed dumps of the REAL reference encoder's B slices
(tests/golden/mb_low_*b*.npz: oracle/ref_tap_mb.c on the unmodified lencod, NumberBFrames 1) and against the oracle (oracle/jmo_mbenc_b.inc, itself pinned to the same dumps).

The device codes the sequence in the encoder's order (I P B P B ...): a P picture's reconstruction is filtered and interpolated on the device and becomes a reference, a B picture
reads its two lists from the slots (list 1 behind list 0 in ref_slot) and the co-located picture's "does not move" map, is filtered, and is nobody's reference."""
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
import test_gpu_mbenc as TG  # noqa: E402
import test_oracle_mbenc as TO  # noqa: E402

pytestmark = pytest.mark.gpu


def b_switch_word(b):
    """jmhip_slice_params::b_switches from the switches of pyjmo.encode_slice_b"""
    w = int(b.get("direct_8x8_inference", 1)) & 1
    if b.get("bipred_me", 0):
        w |= 2
        for i in range(3):
            w |= (int(b["bipred_search"][i]) & 1) << (2 + i)
        w |= (int(b["bipred_refinements"]) & 15) << 8 | (int(b["bipred_range"]) & 255) << 16 | (int(b["bipred_subpel"]) & 3) << 24
    return w


class DevSeqEncoderB(TG.DevSeqEncoder):
    """DevSeqEncoder with non-reference B pictures: references are kept by picture order count."""

    def __init__(self, *a, qpc_p=None, qpc_cr_delta_p=None, **k):
        super().__init__(*a, **k)
        self.poc_of = {}        # slot -> picture order count
        self.qpc_i, self.qpc_cr_delta_i = self.qpc, self.qpc_cr_delta
        self.qpc_pp, self.qpc_cr_delta_pp = (self.qpc, self.qpc_cr_delta) if qpc_p is None else (qpc_p, qpc_cr_delta_p)

    def encode_ref(self, raw, sw, sh, poc):
        self.qpc, self.qpc_cr_delta = (self.qpc_i, self.qpc_cr_delta_i) if self.npic == 0 else (self.qpc_pp, self.qpc_cr_delta_pp)      # the P pictures' chroma QPs (QPPSlice != QPISlice)
        recs, pre, post = self.encode(raw, sw, sh)
        self.poc_of[self.refs[0][0]] = poc
        return recs, pre, post

    def encode_b(self, raw, sw, sh, l0_pocs, l1_pocs, lambdas_b, qp_b, b, qpc_b=None, qpc_cr_delta_b=0, inter_valid_b=None):
        L, J = self.L, self.J
        nmb = (self.W // 16) * (self.H // 16)
        by_poc = {self.poc_of[s]: (s, i) for s, i in self.refs}
        L0, L1 = [by_poc[p] for p in l0_pocs], [by_poc[p] for p in l1_pocs]
        J.set_current_frame(raw, sw, sh)
        recs = np.zeros(nmb, L.MB_RECORD)
        for sn, (first, num) in enumerate(mbenc_util.slices_of(nmb, self.slice_mbs)):
            cfg = pyjmo.mbenc_cfg(self.W, self.H, 1, first, num, qp_b, self.R, len(L0), lambdas_b[0], lambdas_b[1], level_mv=self.level_mv, cabac=self.cabac, search_mode=self.search_mode,
                                  transform8x8=self.transform8x8, yuv_format=self.yuv_format, offsets=self.offsets, inter_valid=inter_valid_b, qpc=qpc_b, qpc_cr_delta=qpc_cr_delta_b)
            prm = TG.slice_params(L, cfg, sn, [r[0] for r in L0 + L1], [r[1] for r in L0 + L1], self.disable_idc)
            prm["num_ref1"] = len(L1)
            prm["b_switches"] = b_switch_word(b)
            recs[first:first + num] = J.encode_slice(prm)
        pre = J.get_recon()
        J.deblock_picture_dev(int(b.get("direct_8x8_inference", 1)))
        post = J.get_recon()
        J.synchronize()
        self.npic += 1
        return recs, pre, post


def run_case_b(tag, check_oracle_post=True):
    c = TO.load_case(tag)
    z = c["z"]
    ov = dict(s.split("=") for s in z["overrides"])
    args = (c["W"], c["H"], c["qp"], c["R"], c["num_ref"], c["lam"], c["slice_mbs"], c["mv_limit"], c["didc"])
    kw = dict(cabac=c.get("cabac", 0), search_mode=c["search_mode"], transform8x8=c["t8"], yuv_format=c["yuv"], offsets=c["offsets"], inter_valid=c["inter_valid"],
              qpc=c["qpc"] if c["qp_p"] in (None, c["qp"]) or c["qpc_p"] is not None else None, qpc_cr_delta=c["qpc_cr_delta"], qp_p=c["qp_p"])
    dev = DevSeqEncoderB(*args, qpc_p=c["qpc_p"], qpc_cr_delta_p=c["qpc_cr_delta_p"], keep=TO.stored_refs(ov, z), **kw)
    orc = mbenc_util.SeqEncoder(*args, epzs=c["epzs"], qpc_p=c["qpc_p"], qpc_cr_delta_p=c["qpc_cr_delta_p"], **kw)
    orc.keep = dev.keep
    nmb = (c["W"] // 16) * (c["H"] // 16)
    src = TO.source_frames(c, tag)
    raw = raw_frames(c, tag)
    lam_b = ([int(x) for x in z["lambda_b"][:3]], int(z["lambda_b"][3]))
    bsw = TO.b_switches(ov, z)
    ivb = [int(ov.get(k, 1)) for k in TO.BSLICE_KEYS] if any(k in ov for k in TO.BSLICE_KEYS) else None
    for n in range(len(z["slice_type"])):
        st, poc = int(z["slice_type"][n]), int(z["poc"][n])
        if st == 1:
            l0 = [int(p) for p in z["ref_poc"][n][:int(z["num_ref_pic"][n])]]
            l1 = [int(p) for p in z["poc_l1"][n][:int(z["num_ref1_pic"][n])]]
            kb = dict(qpc_b=int(z["qpc_b"]), qpc_cr_delta_b=int(z["qpc_v_b"]) - int(z["qpc_b"]), inter_valid_b=ivb)
            recs, pre, post = dev.encode_b(raw[poc // 2], c["sw"], c["sh"], l0, l1, lam_b, int(z["qp_b"]), bsw, **kb)
            orecs, _, opre, opost = orc.encode_b(src[poc // 2], poc, l0, l1, lam_b, int(z["qp_b"]), bsw, **kb)
        else:
            recs, pre, post = dev.encode_ref(raw[poc // 2], c["sw"], c["sh"], poc)
            orecs, _, opre, opost = orc.encode(src[poc // 2], poc=poc)
        want = c["records"][n * nmb:(n + 1) * nmb]
        got = mb_tap.canonical(TG.as_oracle_records(recs), bslice=st == 1)
        bad = [k for k in range(nmb) if got[k].tobytes() != want[k].tobytes()]
        assert not bad, (tag, n, st, len(bad), bad[:8], [(f, want[bad[0]][f].tolist(), got[bad[0]][f].tolist()) for f in mb_tap.diff_fields(want[bad[0]], got[bad[0]])][:6])
        for p, m in zip(pre, z["md5_pre_deblock"][n]):
            assert hashlib.md5(np.ascontiguousarray(p).tobytes()).hexdigest() == m, (tag, n, "reconstruction before the loop filter")
        if check_oracle_post:
            for a, b_ in zip(post, opost):
                assert np.array_equal(a, np.asarray(b_, np.uint8)), (tag, n, st, "the filtered picture differs from the oracle's")


def raw_frames(c, tag):
    """the clip's frames as the bytes of the input file (jmhip_set_current_frame reads those)"""
    z = c["z"]
    clip = str(z["clip"]) if "clip" in z.files else ""
    yuv = c["yuv"]
    fs = c["sw"] * c["sh"] * (2 if yuv == 2 else 3) // (1 if yuv == 2 else 2)
    if clip.startswith("motion"):
        import synth_motion
        fr = synth_motion.motion_clip(c["sw"], c["sh"], c["nfr"], int(clip.split(":")[1]), yuv422=clip.startswith("motion422"))
        return [np.ascontiguousarray(f, np.uint8) for f in fr]
    if clip == "True":
        import bench
        import tempfile
        with tempfile.TemporaryDirectory() as t:
            bench.write_yuv(os.path.join(t, "s.yuv"), c["nfr"])
            data = np.fromfile(os.path.join(t, "s.yuv"), np.uint8)
    else:
        data = np.fromfile(os.path.join(G, "foreman_part_qcif_422.yuv" if yuv == 2 else "foreman_part_qcif.yuv"), np.uint8)
    return [data[n * fs:(n + 1) * fs] for n in range(c["nfr"])]


# without the bi-predictive motion search (BiPredMotionEstimation 0): spatial direct, LIST_0 / LIST_1 / BI_PRED per partition, the direct 8x8 sub-mode; CAVLC full search (q1b0),
# CABAC + 8x8 transform + fast full search + three references + slices that start mid-row (m3b0)
@pytest.mark.parametrize("tag", ["q1b0", "m3b0"])
def test_b_pictures_equal_the_reference_encoder(tag):
    run_case_b(tag)


# with the bi-predictive motion search as the shipped files have it (BiPredMotionEstimation 1: BiPredBlockMotionSearch mv_search.c:1033, BI_PRED_L0 / _L1 in the decision):
# encoder_main.cfg with RDO off (q1b: fast full search SR 32, CABAC), High profile on a clip with motion (m3b: 8x8 transform), one refinement / range 8 / one sub-pel level,
# CAVLC, two list-1 references (m2b4), encoder_yuv422.cfg with its B picture (q5yb: 4:2:2, q_offset.cfg's B lists)
@pytest.mark.parametrize("tag", ["q1b", "m3b", "m2b4", "q5yb"])
def test_b_pictures_with_the_bipredictive_search_equal_the_reference_encoder(tag):
    run_case_b(tag)


def test_b_picture_1080p_equals_the_reference_encoder():
    """encoder_main.cfg's search and B settings at 1080p (fast full search SR 32, CABAC, the bi-predictive search), RDO off: I P B of the synthetic clip (g3b: 24 480 macroblocks
    of the real encoder); the filtered pictures are not compared with the oracle here (minutes of CPU: tests/test_oracle_mbenc.py does that under JMO_LONG)"""
    run_case_b("g3b", check_oracle_post=False)
