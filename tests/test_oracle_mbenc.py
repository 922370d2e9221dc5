"""not gpu: oracle/jmo_mbenc.c (the CPU restatement of encode_one_macroblock_low and its callees) against the REAL reference encoder.

tests/golden/mb_low_*.npz hold, per macroblock, what JM's own encode_one_macroblock_low left behind (oracle/ref_tap_mb.c on the unmodified lencod,
tests/golden/make_mb_golden.py).  The oracle encodes the same clips picture by picture -- its own reconstruction, loop filter and sub-pel planes
feed the next picture, as in lencod -- and every macroblock record, the per-search motion costs and every picture's reconstruction before the loop
filter must equal the reference's.  Configurations: one reference (q1r), five references SR 32 (q5r), three slices (q4r), slices that start mid-row
with DFDisableIdc = 2 and two references (q4s), and BASELINE configs[1] with RDO off at full size (g2r = SURVEY 8c G2r, 8160 macroblocks, slow)."""
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


PSLICE_KEYS = ("PSliceSkip", "PSliceSearch16x16", "PSliceSearch16x8", "PSliceSearch8x16", "PSliceSearch8x8", "PSliceSearch8x4", "PSliceSearch4x8", "PSliceSearch4x4")
EPZS_KEYS = dict(pattern="EPZSPattern", dual="EPZSDualRefinement", fixed="EPZSFixedPredictors", aggressive="EPZSAggressiveWindow", temporal="EPZSTemporal",
                 spatial_mem="EPZSSpatialMem", blocktype="EPZSBlockType", min_scale="EPZSMinThresScale", med_scale="EPZSMedThresScale", max_scale="EPZSMaxThresScale",
                 sub_scale="EPZSSubPelThresScale")


def load_case(tag):
    z = np.load(os.path.join(G, f"mb_low_{tag}.npz"))
    ov = dict(s.split("=") for s in z["overrides"])
    sw, sh, W, H = [int(x) for x in z["size"]]
    lam = {2: ([int(x) for x in z["lambda_i"][:3]], int(z["lambda_i"][3])), 0: ([int(x) for x in z["lambda_p"][:3]], int(z["lambda_p"][3]))}
    didc = int(ov.get("DFDisableRefPSlice", 0))
    return dict(z=z, sw=sw, sh=sh, W=W, H=H, lam=lam, qp=int(z["qp"]), R=int(z["search_range"]), num_ref=int(z["num_ref"]),
                slice_mbs=int(ov.get("SliceArgument", 0)) if ov.get("SliceMode", "0") == "1" else 0, mv_limit=[int(x) for x in z["mv_limit"]], didc=didc,
                nfr=len(z["slice_type"]), records=mb_tap.widen(z["records"]), cabac=int(ov.get("SymbolMode", 0)), search_mode={-1: 0, 0: 1, 3: 3}[int(ov.get("SearchMode", -1))],
                epzs={k: int(ov[n]) for k, n in EPZS_KEYS.items() if n in ov}, t8=int(ov.get("Transform8x8Mode", 0)),
                yuv=int(z["yuv_format"]) if "yuv_format" in z.files else 1,
                offsets=pyjmo.load_q_offsets(os.path.join(G, "q_offset.cfg")) if ov.get("OffsetMatrixPresentFlag", "0") == "1" else None,
                inter_valid=[int(ov.get(k, 1)) for k in PSLICE_KEYS] if any(k in ov for k in PSLICE_KEYS) else None,
                qpc=int(z["qpc"]), qpc_cr_delta=(int(z["qpc_v"]) - int(z["qpc"])) if "qpc_v" in z.files else 0, qp_p=int(z["qp_p"]) if "qp_p" in z.files else None,
                qpc_p=int(z["qpc_p"]) if "qpc_p" in z.files else None, qpc_cr_delta_p=(int(z["qpc_v_p"]) - int(z["qpc_p"])) if "qpc_p" in z.files else None)


def source_frames(c, tag):
    clip = str(c["z"]["clip"]) if "clip" in c["z"].files else ""
    yuv = c["yuv"]
    if clip.startswith("motion"):
        import synth_motion
        fr = synth_motion.motion_clip(c["sw"], c["sh"], c["nfr"], int(clip.split(":")[1]), yuv422=clip.startswith("motion422"))
        assert hashlib.md5(np.concatenate(fr).tobytes()).hexdigest() == str(c["z"]["clip_md5"]), "the generated clip is not the one the golden records were made from"
        return [pyjmo.load_frame(f, c["sw"], c["sh"], c["W"], c["H"], yuv) for f in fr]
    if clip == "syn422":
        import synclip
        import tempfile
        with tempfile.TemporaryDirectory() as t:
            synclip.syn1080p422(os.path.join(t, "s.yuv"), c["nfr"])
            data = np.fromfile(os.path.join(t, "s.yuv"), np.uint8)
        assert hashlib.md5(data.tobytes()).hexdigest() == str(c["z"]["clip_md5"])
    elif tag == "g2r" or clip == "True":
        import bench
        data = bench.synthetic_frames(c["nfr"]) if hasattr(bench, "synthetic_frames") else None
        if data is None:
            import tempfile
            with tempfile.TemporaryDirectory() as t:
                bench.write_yuv(os.path.join(t, "s.yuv"), c["nfr"])
                data = np.fromfile(os.path.join(t, "s.yuv"), np.uint8)
    else:
        data = np.fromfile(os.path.join(G, "foreman_part_qcif_422.yuv" if yuv == 2 else "foreman_part_qcif.yuv"), np.uint8)
    fs = c["sw"] * c["sh"] * (2 if yuv == 2 else 3) // (1 if yuv == 2 else 2)
    return [pyjmo.load_frame(data[n * fs:(n + 1) * fs], c["sw"], c["sh"], c["W"], c["H"], yuv) for n in range(c["nfr"])]


BSLICE_KEYS = ("BSliceDirect", "BSliceSearch16x16", "BSliceSearch16x8", "BSliceSearch8x16", "BSliceSearch8x8", "BSliceSearch8x4", "BSliceSearch4x8", "BSliceSearch4x4")


def b_switches(ov, z):
    """the B slices' switches of pyjmo.encode_slice_b from a case's overrides (defaults: the shipped .cfg files')"""
    return dict(direct_8x8_inference=int(z["direct_8x8_inference"]), direct_temporal=int(int(ov.get("DirectModeType", 1)) == 0), bipred_me=int(ov.get("BiPredMotionEstimation", 1)),
                bipred_search=[int(ov.get(k, d)) for k, d in (("BiPredSearch16x16", 1), ("BiPredSearch16x8", 1), ("BiPredSearch8x16", 1), ("BiPredSearch8x8", 0))],
                bipred_refinements=int(ov.get("BiPredMERefinements", 3)), bipred_range=int(ov.get("BiPredMESearchRange", 16)), bipred_subpel=int(ov.get("BiPredMESubPel", 2)))


def stored_refs(ov, z):
    """reference pictures the sliding window keeps: NumberReferenceFrames (B lists may be shorter: BList0References / BList1References) -- at least the longest list a golden holds"""
    return max(int(ov.get("NumberReferenceFrames", 0)), int(z["num_ref_pic"].max()), int(z["num_ref1_pic"].max()))


def run_case_b(tag, nmax=None):
    """A sequence with B pictures (coding order I P B P B ...): the pictures in the order and with the reference lists the real encoder used (picture order counts in the golden)."""
    c = load_case(tag)
    z = c["z"]
    ov = dict(s.split("=") for s in z["overrides"])
    enc = mbenc_util.SeqEncoder(c["W"], c["H"], c["qp"], c["R"], c["num_ref"], c["lam"], c["slice_mbs"], c["mv_limit"], c["didc"], cabac=c.get("cabac", 0),
                                search_mode=c["search_mode"], epzs=c["epzs"], transform8x8=c["t8"], yuv_format=c["yuv"], offsets=c["offsets"], inter_valid=c["inter_valid"],
                                qpc=c["qpc"] if c["qp_p"] in (None, c["qp"]) or c["qpc_p"] is not None else None, qpc_cr_delta=c["qpc_cr_delta"], qp_p=c["qp_p"], qpc_p=c["qpc_p"], qpc_cr_delta_p=c["qpc_cr_delta_p"])
    enc.keep = stored_refs(ov, z)
    nmb = (c["W"] // 16) * (c["H"] // 16)
    src = source_frames(c, tag)
    lam_b = ([int(x) for x in z["lambda_b"][:3]], int(z["lambda_b"][3]))
    bsw = b_switches(ov, z)
    ivb = [int(ov.get(k, 1)) for k in BSLICE_KEYS] if any(k in ov for k in BSLICE_KEYS) else None
    for n in range(len(z["slice_type"]) if nmax is None else nmax):
        st, poc = int(z["slice_type"][n]), int(z["poc"][n])
        cur = src[poc // 2]
        if st == 1:
            l0 = [int(p) for p in z["ref_poc"][n][:int(z["num_ref_pic"][n])]]
            l1 = [int(p) for p in z["poc_l1"][n][:int(z["num_ref1_pic"][n])]]
            recs, dbg, pre, post = enc.encode_b(cur, poc, l0, l1, lam_b, int(z["qp_b"]), bsw, debug=True, qpc_b=int(z["qpc_b"]), qpc_cr_delta_b=int(z["qpc_v_b"]) - int(z["qpc_b"]), inter_valid_b=ivb)
        else:
            recs, dbg, pre, post = enc.encode(cur, debug=True, poc=poc)
        want = c["records"][n * nmb:(n + 1) * nmb]
        got = mb_tap.canonical(recs, bslice=st == 1)
        if st != 2 and z["motion_cost"].size:
            # the encoder never clears p_Vid->motion_cost: a mode this slice type does not search (P/BSliceSearch* 0) holds what the last picture of the other type left there
            iv = (ivb if st == 1 else c["inter_valid"]) or [1] * 8
            ms = [m - 1 for m in range(1, 8) if iv[m]]
            mc = z["motion_cost"][n * nmb:(n + 1) * nmb]
            badc = [k for k in range(nmb) if not np.array_equal(dbg["motion_cost"][k, 1:, :][ms], mc[k][ms])]
            assert not badc, (tag, n, "motion costs, list 0", badc[:5], dbg["motion_cost"][badc[0], 1:, :].tolist(), mc[badc[0]].tolist())
            if st == 1:
                mc1 = z["motion_cost1"][n * nmb:(n + 1) * nmb]
                badc = [k for k in range(nmb) if not np.array_equal(dbg["motion_cost1"][k, 1:, :][ms], mc1[k][ms])]
                assert not badc, (tag, n, "motion costs, list 1", badc[:5], dbg["motion_cost1"][badc[0], 1:, :].tolist(), mc1[badc[0]].tolist())
        bad = [k for k in range(nmb) if got[k].tobytes() != want[k].tobytes()]
        assert not bad, (tag, n, st, len(bad), bad[:5], [(f, want[bad[0]][f].tolist(), got[bad[0]][f].tolist()) for f in mb_tap.diff_fields(want[bad[0]], got[bad[0]])][:6])
        for p, m in zip(pre, z["md5_pre_deblock"][n]):
            assert hashlib.md5(np.ascontiguousarray(p.astype(np.uint8)).tobytes()).hexdigest() == m, (tag, n, "reconstruction before the loop filter")
    return enc


# B pictures (jmo_mbenc_b.inc): spatial direct, LIST_0 / LIST_1 / BI_PRED per partition, the direct 8x8 sub-mode, with (q1b, m3b, m2b4, q5yb) and without (q1b0, m3b0) the bi-predictive
# motion search; CAVLC / CABAC, 4x4 / 8x8 transform, full search / fast full search, slices, two list-1 references (m2b4), 4:2:2 with q_offset.cfg's B lists (q5yb)
@pytest.mark.parametrize("tag", ["q1b0", "q1b", "m3b0", "m3b", "m2b4", "q5yb", "q1bt", "m3bt"])
def test_oracle_b_pictures_equal_the_reference_encoder(tag):
    run_case_b(tag)


@pytest.mark.skipif(os.environ.get("JMO_LONG") != "1", reason="minutes of oracle searches: set JMO_LONG=1 (q1b pins the same configuration at QCIF)")
def test_oracle_b_picture_1080p_full_size():
    """encoder_main.cfg's search and B settings at 1080p, RDO off: I P B of the synthetic clip (g3b)"""
    run_case_b("g3b")


def run_case(tag):
    c = load_case(tag)
    enc = mbenc_util.SeqEncoder(c["W"], c["H"], c["qp"], c["R"], c["num_ref"], c["lam"], c["slice_mbs"], c["mv_limit"], c["didc"], cabac=c.get("cabac", 0),
                                search_mode=c["search_mode"], epzs=c["epzs"], transform8x8=c["t8"], yuv_format=c["yuv"], offsets=c["offsets"], inter_valid=c["inter_valid"], qpc=c["qpc"] if c["qp_p"] in (None, c["qp"]) or c["qpc_p"] is not None else None, qpc_cr_delta=c["qpc_cr_delta"], qp_p=c["qp_p"], qpc_p=c["qpc_p"], qpc_cr_delta_p=c["qpc_cr_delta_p"])
    nmb = (c["W"] // 16) * (c["H"] // 16)
    z = c["z"]
    for n, cur in enumerate(source_frames(c, tag)):
        recs, dbg, pre, post = enc.encode(cur, debug=True)
        want = c["records"][n * nmb:(n + 1) * nmb]
        got = mb_tap.canonical(recs)
        bad = [k for k in range(nmb) if got[k].tobytes() != want[k].tobytes()]
        assert not bad, (tag, n, bad[:5], mb_tap.diff_fields(want[bad[0]], got[bad[0]]))
        for p, m in zip(pre, z["md5_pre_deblock"][n]):
            assert hashlib.md5(np.ascontiguousarray(p.astype(np.uint8)).tobytes()).hexdigest() == m, (tag, n, "reconstruction before the loop filter")
        if int(z["slice_type"][n]) == 0 and z["motion_cost"].size:
            assert np.array_equal(dbg["motion_cost"][:, 1:, :], z["motion_cost"][n * nmb:(n + 1) * nmb]), (tag, n, "motion costs")
    return enc


@pytest.mark.parametrize("tag", ["q1r", "q5r", "q4r", "q4s", "q1c", "q0c", "q0r"])     # q1c / q0c: CABAC (Main profile) at QP 28 / 0; q0r: CAVLC at QP 0
def test_oracle_macroblock_pipeline_equals_the_reference_encoder(tag):
    run_case(tag)


# EPZS (SearchMode = 3; oracle/jmo_mbenc_epzs.inc): the reference's clip with the shipped switches (q1e); on tests/golden/synth_motion.py's clips five
# references (m5e), CABAC + slices that start mid-row (m2c), the other patterns / the aggressive window set / other thresholds (m3p), every optional
# predictor set off (m2t)
@pytest.mark.parametrize("tag", ["q1e", "m5e", "m2c", "m3p", "m2t"])
def test_oracle_epzs_pipeline_equals_the_reference_encoder(tag):
    enc = run_case(tag)
    assert enc.epzs_stats and all(a == 0 for _, a in enc.epzs_stats)        # no search count near 65535 here: JM's 16-bit map stamp cannot alias


def test_oracle_macroblock_pipeline_configs1_full_size():
    """BASELINE configs[1] with RDOptimization = 0 (G2r): I + P picture of the synthetic 1080p clip, 16 320 macroblocks (about half a minute)."""
    run_case("g2r")


# High profile, Transform8x8Mode = 1: transform_decision for 16x16 / 16x8 / 8x16, the tr8x8 and tr4x4 passes of P8x8, Intra8x8, 8x8 Hadamard SATD in the sub-pel
# searches, residual_transform_quant_luma_8x8 with CAVLC (q1h, m1hq) and CABAC (q2hc, m3h: three references, slices that start mid-row), with EPZS (m2he, m1hq)
@pytest.mark.parametrize("tag", ["q1h", "q2hc", "m3h", "m2he", "m1hq"])
def test_oracle_high_profile_pipeline_equals_the_reference_encoder(tag):
    run_case(tag)


def test_oracle_epzs_configs2_full_size():
    """BASELINE configs[2] as far as the pipeline goes (1080p, Main profile, CABAC, EPZS, RDO off, P pictures only, 4x4 transform): I + 2 P pictures, 24 480 macroblocks,
    a million EPZS searches; the stamp of JM's visited map wraps round ten times per picture without a single aliased candidate on this clip."""
    enc = run_case("g3e")
    assert [s for s, _ in enc.epzs_stats] == [334560, 669120] and all(a == 0 for _, a in enc.epzs_stats)


def test_oracle_epzs_six_pictures_full_size():
    """Six pictures of the same search (g6e): five P pictures with up to five references -- the temporal predictors, the spatial memory and JM's 16-bit visited-map stamp
    over a chain of 4.7 million searches (the stamp wraps round seventy times) without a single aliased candidate."""
    enc = run_case("g6e")
    assert len(enc.epzs_stats) == 5 and sum(s for s, _ in enc.epzs_stats) > 4000000 and all(s > 65535 for s, _ in enc.epzs_stats) and all(a == 0 for _, a in enc.epzs_stats)


def test_oracle_epzs_eight_pictures_one_reference_full_size():
    """Eight 1080p pictures of the same search with ONE reference (g8e): the sequence whose seven P pictures the device codes in one launch (tests/test_gpu_seq.py)."""
    enc = run_case("g8e")
    assert len(enc.epzs_stats) == 7 and all(a == 0 for _, a in enc.epzs_stats)


def test_oracle_epzs_forty_pictures():
    """Forty pictures of EPZS with two references (m2e40): JM's 16-bit visited-map stamp wraps round several times over the sequence; every picture's records equal the real
    encoder's and the oracle meets no aliased candidate -- the sequence the device's long EPZS runs (in flight, in one launch) are pinned to."""
    enc = run_case("m2e40")
    assert len(enc.epzs_stats) == 39 and sum(s for s, _ in enc.epzs_stats) > 4 * 65536 and all(a == 0 for _, a in enc.epzs_stats), enc.epzs_stats[-3:]


@pytest.mark.skipif(os.environ.get("JMO_LONG") != "1", reason="three minutes of oracle full searches: set JMO_LONG=1 (g2r pins the first two pictures of the same sequence)")
def test_oracle_macroblock_pipeline_configs1_six_pictures():
    """BASELINE configs[1] with RDOptimization = 0, six pictures (g6r): what the device's pictures in flight are checked against (tests/test_gpu_seq.py, bench.py)."""
    run_case("g6r")


def test_oracle_configs2_as_stated_full_size():
    """BASELINE configs[2] as stated -- 1080p, CABAC, 8x8 transform on (High profile), EPZS -- with RDO off and P pictures only (g3h): 24 480 macroblocks."""
    enc = run_case("g3h")
    assert all(a == 0 for _, a in enc.epzs_stats)


# 4:2:2 (High 4:2:2 profile, encoder_yuv422.cfg): 8 x 16 chroma samples per macroblock -- eight 4x4 blocks per plane, the 2x4 DC transform with the quantiser of qpc + 3,
# the vector of the luma block at the same row -- and the quantiser offsets of q_offset.cfg (OffsetMatrixPresentFlag 1).  q5y = BASELINE configs[4] on its own clip but for
# RDO / adaptive rounding / B pictures (CABAC, 8x8 transform, fast full search SR 32, five references); q2yv CAVLC, 4x4 transform only, default offsets; m3y EPZS, QP 36 (qpc != qp),
# slices that start mid-row; m2yq QP 12, CAVLC
@pytest.mark.parametrize("tag", ["q5y", "q2yv", "m3y", "m2yq"])
def test_oracle_yuv422_pipeline_equals_the_reference_encoder(tag):
    run_case(tag)


# partitions switched off (PSliceSearch*): EPZS's block-type predictors read currSlice->all_mv of types that are never searched -- the zeros it was allocated with
@pytest.mark.parametrize("tag", ["m2pd", "m3pe", "q1pd"])
def test_oracle_with_partitions_switched_off(tag):
    run_case(tag)


# level 1.1: the vertical vector limit (-256 .. 255 quarter-pels) cuts into SearchRange 32 -- full search, fast full search, EPZS
@pytest.mark.parametrize("tag", ["m3fl", "m3fm", "m2sl", "m2el"])
def test_oracle_at_a_small_level(tag):
    run_case(tag)


# EPZS at SearchRange 2: the window predictor set is empty and its searchPoints of -1 takes the last predictor before it off the list (me_epzs_common.c:343-414, :1661-1673)
@pytest.mark.parametrize("tag", ["m2es", "m5es"])
def test_oracle_epzs_at_search_range_two(tag):
    run_case(tag)


# CbQPOffset != CrQPOffset (High profiles): the two chroma planes are quantised (and loop-filtered) with different QPs; 4:2:0 (m2cq) and 4:2:2 (m2yc)
@pytest.mark.parametrize("tag", ["m2cq", "m2yc", "m2cp"])
def test_oracle_with_different_chroma_qps(tag):
    run_case(tag)


@pytest.mark.skipif(os.environ.get("JMO_LONG") != "1", reason="ten minutes of oracle searches: set JMO_LONG=1 (q5y pins the same configuration at QCIF)")
def test_oracle_configs4_1080p_full_size():
    """BASELINE configs[4] at its own size: 1080p 4:2:2 synthetic, encoder_yuv422.cfg but for RDO / adaptive rounding / B pictures (g4y): I + 2 P pictures."""
    run_case("g4y")


@pytest.mark.skipif(os.environ.get("JMO_LONG") != "1", reason="six minutes of oracle full searches: set JMO_LONG=1 (q5f / m5f / m3fh pin the same search at small sizes)")
def test_oracle_fast_full_search_1080p_full_size():
    """encoder_baseline.cfg's search as shipped (SearchMode 0, five references configured) at 1080p, RDO off: I + 3 P pictures with 1, 2 and 3 references (g5f)."""
    run_case("g5f")


def test_record_layout():
    assert pyjmo.MB_RECORD.itemsize == 1296


def test_oracle_equals_the_tapped_encoder_on_random_configurations():
    """Twelve seconds of tests/fuzz_oracle.py: seeded random RDO-off configurations through the tapped REAL encoder (oracle/_ref/lencod_tapmb.exe, built from /root/reference by
    oracle/Makefile.ref) and through the oracle -- records, motion costs, reconstructions equal.  Only where the reference's build exists (the build container); the committed goldens
    pin the same restatement everywhere else.  (35 minutes of the same script: profiles/r03_fuzz_oracle.txt.)"""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "lencod_tapmb.exe")):
        pytest.skip("needs oracle/_ref/lencod_tapmb.exe (the reference's build: /root/reference)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_oracle.py"), "12", "950000", "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and " 0 NOT equal" in out, out[-3000:]
