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
    if b.get("direct_temporal", 0):
        w |= 32
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

    def encode_b(self, raw, sw, sh, l0_pocs, l1_pocs, lambdas_b, qp_b, b, qpc_b=None, qpc_cr_delta_b=0, inter_valid_b=None, poc=0):
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
            if b.get("direct_temporal", 0):                  # picture order counts: the temporal direct mode's scales (list 1 behind list 0)
                prm["poc_cur"] = poc
                for r, p in enumerate(list(l0_pocs) + list(l1_pocs)):
                    prm["poc_ref"][0, r] = p
            if sn == 0:
                self.log.append(dict(st=1, raw=raw, sw=sw, sh=sh, prm=prm.copy(), nslices=len(mbenc_util.slices_of(nmb, self.slice_mbs)), pic_id=None, d8=int(b.get("direct_8x8_inference", 1))))
            recs[first:first + num] = J.encode_slice(prm)
        pre = J.get_recon()
        J.deblock_picture_dev(int(b.get("direct_8x8_inference", 1)))
        post = J.get_recon()
        J.synchronize()
        self.npic += 1
        return recs, pre, post


def run_case_b(tag, check_oracle_post=True, flight=None):
    """flight = (depth, workgroups, streamed records): the sequence once more with that many pictures in flight (jmhip_seq_*); records and filtered pictures must be the same"""
    classic = []
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
            recs, pre, post = dev.encode_b(raw[poc // 2], c["sw"], c["sh"], l0, l1, lam_b, int(z["qp_b"]), bsw, poc=poc, **kb)
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
        classic.append((recs, post, want, [np.asarray(x, np.uint8) for x in opost], st))
    dev.J.close()
    if flight:
        got = replay_in_flight(dev, *flight)
        for n, ((r0, p0, want, opost, st), (r1, p1)) in enumerate(zip(classic, got)):
            # the pictures in flight DIRECTLY against the real encoder's records and the oracle's filtered pictures (round 6) ...
            d = TG.first_difference(want, mb_tap.canonical(TG.as_oracle_records(r1), bslice=st == 1))
            assert d is None, (tag, n, "records of the picture in flight against the reference encoder", d[:3])
            assert all(np.array_equal(a, b_) for a, b_ in zip(opost, p1)), (tag, n, "filtered picture of the picture in flight against the oracle")
            # ... and byte for byte against the picture-after-picture path (the fields a canonical record drops too)
            assert r0.tobytes() == r1.tobytes(), (tag, n, "records of the picture in flight", TG.first_difference(TG.as_oracle_records(r0), TG.as_oracle_records(r1)))
            assert all(np.array_equal(a, b_) for a, b_ in zip(p0, p1)), (tag, n, "filtered picture of the picture in flight")


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


def replay_in_flight(dev, depth, workgroups=0, stream_records=False, b_workgroups=0):
    """The pictures of a finished picture-after-picture run (dev.log) once more through jmhip_seq_* with `depth` pictures in flight: the P pictures follow their references
    macroblocks apart inside the device, a B picture starts when both its references are complete and runs beside the P pictures after them.  Returns per picture
    (records, filtered planes)."""
    L = dev.L
    nmb = (dev.W // 16) * (dev.H // 16)
    nslots = max(dev.num_ref, dev.keep) + depth + 1
    J = L.JmHip(dev.W, dev.H, search_range=max(dev.R, 1), num_ref_slots=nslots, yuv_format=dev.yuv_format)
    J.seq_open(depth, workgroups)
    if b_workgroups:
        J.seq_b_workgroups(b_workgroups)               # the B pictures' own share (jmhip_seq_b_workgroups)
    slot_of, stored, pending, out = {}, [], {}, {}          # picture id -> slot; ids of the stored references (most recent first); picture number -> slot, not collected yet
    keep = max(dev.num_ref, dev.keep)

    def collect(k):
        e = k % depth
        if stream_records:
            recs = J.seq_records_streamed(e, 0, nmb)
            J.seq_wait(e)
        else:
            J.seq_wait(e)
            recs = J.seq_records(e)
        out[k] = (recs, J.seq_get_recon(pending.pop(k)))

    try:
        for k, p in enumerate(dev.log):
            if k >= depth:
                collect(k - depth)
            prm = p["prm"].copy()
            nref_all = 0 if p["st"] == 2 else int(prm["num_ref"][0]) + (int(prm["num_ref1"][0]) if p["st"] == 1 else 0)
            for r in range(nref_all):
                prm["ref_slot"][0, r] = slot_of[int(prm["ref_id"][0, r])]
            if p["nslices"] > 1:
                prm["num_slices"] = p["nslices"]
            busy = {slot_of[i] for i in stored} | set(pending.values())      # (the reconstruction is written while the oldest reference is still read)
            slot = [s for s in range(nslots) if s not in busy][0]
            J.seq_set_frame(k % depth, p["raw"], p["sw"], p["sh"])
            J.seq_encode(k % depth, prm, slot, p["d8"], stream_records)
            pending[k] = slot
            if p["st"] != 1:
                slot_of[p["pic_id"]] = slot
                stored = ([p["pic_id"]] + stored)[:keep]
        for k in sorted(pending):
            collect(k)
        J.synchronize()
    finally:
        J.close()
    return [out[k] for k in range(len(dev.log))]


# without the bi-predictive motion search (BiPredMotionEstimation 0): spatial direct, LIST_0 / LIST_1 / BI_PRED per partition, the direct 8x8 sub-mode; CAVLC full search (q1b0),
# CABAC + 8x8 transform + fast full search + three references + slices that start mid-row (m3b0)
@pytest.mark.parametrize("tag", ["q1b0", "m3b0"])
def test_b_pictures_equal_the_reference_encoder(tag):
    run_case_b(tag)


# with the bi-predictive motion search as the shipped files have it (BiPredMotionEstimation 1: BiPredBlockMotionSearch mv_search.c:1033, BI_PRED_L0 / _L1 in the decision):
# encoder_main.cfg with RDO off (q1b: fast full search SR 32, CABAC), High profile on a clip with motion (m3b: 8x8 transform), one refinement / range 8 / one sub-pel level,
# CAVLC, two list-1 references (m2b4), encoder_yuv422.cfg with its B picture (q5yb: 4:2:2, q_offset.cfg's B lists)
@pytest.mark.parametrize("tag", ["q1b", "m3b", "m2b4", "q5yb", "q1bt", "m3bt"])          # *bt: DirectModeType 0 (temporal direct)
def test_b_pictures_with_the_bipredictive_search_equal_the_reference_encoder(tag):
    run_case_b(tag)


def test_b_picture_1080p_equals_the_reference_encoder():
    """encoder_main.cfg's search and B settings at 1080p (fast full search SR 32, CABAC, the bi-predictive search), RDO off: I P B of the synthetic clip (g3b: 24 480 macroblocks
    of the real encoder); the filtered pictures are not compared with the oracle here (minutes of CPU: tests/test_oracle_mbenc.py does that under JMO_LONG)"""
    run_case_b("g3b", check_oracle_post=False)


# B pictures in flight (jmhip_seq_encode with slice_type 1): the sequences of the goldens once more with 3 / 6 pictures in flight -- P pictures following their references inside
# the device, B pictures beside the P pictures after them --: every record and every filtered picture as picture after picture (which the tests above pin to the real encoder)
@pytest.mark.parametrize("tag,flight", [("m3b", (3, 0, False)), ("m3b0", (6, 20, True, 48)), ("m2b4", (4, 0, True, 7)), ("q5yb", (3, 0, False)), ("q1b", (2, 0, False)), ("m3bt", (4, 0, False))])
def test_b_pictures_in_flight_equal_picture_after_picture(tag, flight):
    run_case_b(tag, check_oracle_post=False, flight=flight)


def test_many_b_pictures_in_flight_with_few_slots_and_a_starved_b_share():
    """The slots' readers and writers under stress (round 5's bug class: a P picture overwrote a slot a B picture still read -- found by bench.py, not by a test): 41 pictures
    I P B P B ... of a clip with motion, two references, every B picture given ONE workgroup (jmhip_seq_b_workgroups(1): a B launch then takes many P launches' time, the host runs
    entries ahead of the device and every slot is written again while launches queued long ago still read it), as few slots as the sequence needs.  Picture after picture the
    device equals the ORACLE (records, filtered pictures); in flight every picture must equal those again -- twice, with 3 and with 6 pictures in flight."""
    import synth_motion
    W, H, R, keep, nfr = 176, 144, 8, 2, 41
    qp, qp_b = 27, 29
    lam_of = lambda q: int(192 * 2 ** ((q - 28) / 6))
    f, fb = lam_of(qp), lam_of(qp_b)
    lam = {2: ([f] * 3, f), 0: ([f, f + 3, f + 5], f + 1)}
    lam_b = ([fb + 1, fb + 2, fb + 6], fb + 2)
    b = dict(direct_8x8_inference=1, direct_temporal=0, bipred_me=1, bipred_search=[1, 1, 1, 0], bipred_refinements=1, bipred_range=8, bipred_subpel=1)
    args = (W, H, qp, R, keep, lam, 0)
    kw = dict(cabac=1, search_mode=-1, transform8x8=0, yuv_format=1)
    dev = DevSeqEncoderB(*args, **kw)
    ora = mbenc_util.SeqEncoder(*args, **kw)
    frames = synth_motion.motion_clip(W, H, nfr, 606)
    order = [(0, 2)]
    for g in range((nfr - 1) // 2):
        order += [(2 * g + 2, 0), (2 * g + 1, 1)]
    stored, classic = [], []
    for disp, st in order:
        raw, poc = frames[disp], 2 * disp
        src = pyjmo.load_frame(raw, W, H, W, H, 1)
        if st == 1:
            l0, l1 = sorted([p for p in stored if p < poc], reverse=True)[:1], sorted(p for p in stored if p > poc)[:1]
            recs, pre, post = dev.encode_b(raw, W, H, l0, l1, lam_b, qp_b, b, poc=poc)
            orecs, _, opre, opost = ora.encode_b(src, poc, l0, l1, lam_b, qp_b, b)
        else:
            recs, pre, post = dev.encode_ref(raw, W, H, poc)
            orecs, _, opre, opost = ora.encode(src, poc=poc)
            stored = ([poc] + stored)[:keep]
        d = TG.first_difference(mb_tap.canonical(orecs, bslice=st == 1), mb_tap.canonical(TG.as_oracle_records(recs), bslice=st == 1))
        assert d is None, ("picture after picture against the oracle", disp, st, d[:3])
        assert all(np.array_equal(a, np.asarray(x, np.uint8)) for a, x in zip(post, opost)), ("filtered picture against the oracle", disp, st)
        classic.append((recs, post))
    dev.J.close()
    for depth, wg in ((3, 0), (6, 12)):
        got = replay_in_flight(dev, depth, wg, False, 1)
        for n, ((r0, p0), (r1, p1)) in enumerate(zip(classic, got)):
            assert r0.tobytes() == r1.tobytes(), (depth, n, "records of the picture in flight", TG.first_difference(TG.as_oracle_records(r0), TG.as_oracle_records(r1)))
            assert all(np.array_equal(a, b_) for a, b_ in zip(p0, p1)), (depth, n, "filtered picture of the picture in flight")
