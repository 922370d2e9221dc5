"""Regenerates tests/golden/mb_low_*.npz: what the REAL reference encoder's encode_one_macroblock_low (lencod/src/md_low.c:104) leaves behind per
macroblock, for the configurations the RDO-off macroblock pipeline is pinned on.  TEST INFRASTRUCTURE; needs /root/reference (oracle/Makefile.ref,
target `tapmb`, builds oracle/_ref/lencod_tapmb.exe from the unmodified lencod objects + oracle/ref_tap_mb.c).

  python tests/golden/make_mb_golden.py

Per configuration: canonical macroblock records of every picture (tests/golden/mb_tap.py), the slice parameters the encoder used (lambda tables come
from its double arithmetic and are inputs, never recomputed), md5 of the bitstream and of the reconstruction file, and the md5 of every picture's
reconstruction before the loop filter (assembled from the per-macroblock samples)."""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

G = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(G))
sys.path.insert(0, ROOT)
sys.path.insert(0, G)
import mb_tap  # noqa: E402

EXE = os.path.join(ROOT, "oracle", "_ref", "lencod_tapmb.exe")
RDO_OFF = {"RDOptimization": "0", "AdaptiveRounding": "0"}
SYN1080 = {"InputFile": "syn1080p.yuv", "SourceWidth": "1920", "SourceHeight": "1080", "OutputWidth": "1920", "OutputHeight": "1080",
           "FramesToBeEncoded": "2", "SearchMode": "-1", "SearchRange": "32", "NumberReferenceFrames": "1", "LevelIDC": "51"}
CASES = {
    # tag: (overrides on tests/golden/jm_baseline.cfg -- or on the .cfg named as a fifth element --, source size, frames, synthetic clip?)
    "q1r": (dict(RDO_OFF, SearchMode="-1", SearchRange="16", NumberReferenceFrames="1"), (176, 144), 3, False),
    "q5r": (dict(RDO_OFF, SearchMode="-1", SearchRange="32"), (176, 144), 3, False),                      # five references
    "q4r": (dict(RDO_OFF, SearchMode="-1", SearchRange="16", SliceMode="1", SliceArgument="33"), (176, 144), 3, False),
    "q4s": (dict(RDO_OFF, SearchMode="-1", SearchRange="16", SliceMode="1", SliceArgument="40", NumberReferenceFrames="2", DFParametersFlag="1",
                 DFDisableRefISlice="2", DFDisableNRefISlice="2", DFDisableRefPSlice="2", DFDisableNRefPSlice="2"), (176, 144), 3, False),
    # CABAC (Main profile): the quantiser no longer clamps levels to CAVLC_LEVEL_LIMIT; QP 0 is where levels beyond 2063 occur (Intra16x16 DC)
    "q1c": (dict(RDO_OFF, SearchMode="-1", SearchRange="16", NumberReferenceFrames="1", SymbolMode="1", ProfileIDC="77"), (176, 144), 3, False),
    "q0c": (dict(RDO_OFF, SearchMode="-1", SearchRange="16", NumberReferenceFrames="1", SymbolMode="1", ProfileIDC="77", QPISlice="0", QPPSlice="0"), (176, 144), 3, False),
    "q0r": (dict(RDO_OFF, SearchMode="-1", SearchRange="16", NumberReferenceFrames="1", QPISlice="0", QPPSlice="0"), (176, 144), 3, False),
    "g2r": (dict(RDO_OFF, **SYN1080), (1920, 1080), 2, True),                                          # SURVEY 8c G2r = BASELINE configs[1], RDO off
    # the same, six pictures: what a sequence with consecutive pictures in flight (jmhip_seq_*) must leave, picture by picture (bench.py checks its timed sequence against it)
    "g6r": (dict(RDO_OFF, **dict(SYN1080, FramesToBeEncoded="6")), (1920, 1080), 6, True),
    # six pictures of configs[2]'s search (EPZS, five references configured, CABAC): the temporal predictors and spatial memory over a longer chain (a million searches a picture)
    "g6e": (dict(RDO_OFF, **dict(SYN1080, SearchMode="3", NumberReferenceFrames="5", FramesToBeEncoded="6", SymbolMode="1", ProfileIDC="77")), (1920, 1080), 6, True),
    # eight pictures of the same search with ONE reference: every P picture alike, so that the P pictures go through ONE launch (jmhip_seq_batch) at configs[2]'s size (round 6)
    "g8e": (dict(RDO_OFF, **dict(SYN1080, SearchMode="3", NumberReferenceFrames="1", FramesToBeEncoded="8", SymbolMode="1", ProfileIDC="77")), (1920, 1080), 8, True),
    # EPZS (SearchMode = 3) with the shipped EPZS switches (pattern 2, dual 3, fixed 2, temporal, spatial memory, block type, sub-pel grid, EPZS sub-pel search)
    # BASELINE configs[2] without its 8x8 transform and B pictures: 1080p, Main profile, CABAC, EPZS, five references configured (two exist by the third picture)
    "g3e": (dict(RDO_OFF, **dict(SYN1080, SearchMode="3", NumberReferenceFrames="5", FramesToBeEncoded="3", SymbolMode="1", ProfileIDC="77")), (1920, 1080), 3, True),
    # fast full search (SearchMode = 0, encoder_baseline.cfg's own): one search centre per macroblock and reference, the (0,0) vector first
    # encoder_baseline.cfg as north_star names it (fast full search, SearchRange 32, five references) at 1080p, but for RDO / adaptive rounding: four pictures
    "g5f": (dict(RDO_OFF, **dict(SYN1080, SearchMode="0", NumberReferenceFrames="5", FramesToBeEncoded="4")), (1920, 1080), 4, True),
    "q5f": (dict(RDO_OFF, SearchMode="0", SearchRange="32"), (176, 144), 3, False),                                   # encoder_baseline.cfg as shipped but for RDO / adaptive rounding
    "m5f": (dict(RDO_OFF, SearchMode="0", SearchRange="16", FramesToBeEncoded="6", SliceMode="1", SliceArgument="50"), (208, 160), 6, "motion:31"),     # five references, slices mid-row
    "m3fh": (dict(RDO_OFF, SearchMode="0", SearchRange="32", NumberReferenceFrames="3", Transform8x8Mode="1", ProfileIDC="100", SymbolMode="1", FramesToBeEncoded="4", QPISlice="32", QPPSlice="32"), (176, 144), 4, "motion:32"),
    # High profile: the 8x8 transform beside the 4x4 one (Transform8x8Mode = 1): transform_decision, the tr8x8 pass of P8x8, Intra8x8
    "q1h": (dict(RDO_OFF, SearchMode="-1", SearchRange="16", NumberReferenceFrames="1", Transform8x8Mode="1", ProfileIDC="100"), (176, 144), 3, False),     # CAVLC
    "q2hc": (dict(RDO_OFF, SearchMode="-1", SearchRange="16", NumberReferenceFrames="2", Transform8x8Mode="1", ProfileIDC="100", SymbolMode="1"), (176, 144), 3, False),   # CABAC
    "m3h": (dict(RDO_OFF, SearchMode="-1", SearchRange="32", NumberReferenceFrames="3", Transform8x8Mode="1", ProfileIDC="100", SymbolMode="1", SliceMode="1", SliceArgument="45",
                 FramesToBeEncoded="5", QPISlice="24", QPPSlice="24"), (208, 160), 5, "motion:21"),
    "m2he": (dict(RDO_OFF, SearchMode="3", SearchRange="32", NumberReferenceFrames="2", Transform8x8Mode="1", ProfileIDC="100", SymbolMode="1", FramesToBeEncoded="4"), (176, 144), 4, "motion:22"),   # EPZS + 8x8: configs[2]
    "m1hq": (dict(RDO_OFF, SearchMode="3", SearchRange="16", NumberReferenceFrames="1", Transform8x8Mode="1", ProfileIDC="100", FramesToBeEncoded="3", QPISlice="40", QPPSlice="40"), (176, 144), 3, "motion:23"),   # CAVLC, coarse
    # BASELINE configs[2] as stated (High profile so that the 8x8 transform exists: CABAC, 8x8 transform on, EPZS), P pictures only, RDO off
    "g3h": (dict(RDO_OFF, **dict(SYN1080, SearchMode="3", NumberReferenceFrames="5", FramesToBeEncoded="3", SymbolMode="1", ProfileIDC="100", Transform8x8Mode="1")), (1920, 1080), 3, True),
    "q1e": (dict(RDO_OFF, SearchMode="3", SearchRange="16", NumberReferenceFrames="1"), (176, 144), 3, False),          # the reference's own clip
    "m5e": (dict(RDO_OFF, SearchMode="3", SearchRange="32", FramesToBeEncoded="6"), (208, 160), 6, "motion:11"),         # five references: the ref > 0 exits, scaled predictors
    "m2c": (dict(RDO_OFF, SearchMode="3", SearchRange="32", NumberReferenceFrames="2", SymbolMode="1", ProfileIDC="77", SliceMode="1", SliceArgument="40", FramesToBeEncoded="4"), (176, 144), 4, "motion:12"),   # CABAC, slices that start mid-row
    "m3p": (dict(RDO_OFF, SearchMode="3", SearchRange="8", NumberReferenceFrames="3", EPZSPattern="4", EPZSDualRefinement="6", EPZSFixedPredictors="3", EPZSAggressiveWindow="1",
                 EPZSMinThresScale="1", EPZSSubPelThresScale="1", FramesToBeEncoded="4", QPISlice="36", QPPSlice="36"), (176, 144), 4, "motion:13"),    # the other patterns / window set
    "m2t": (dict(RDO_OFF, SearchMode="3", SearchRange="16", NumberReferenceFrames="2", EPZSPattern="0", EPZSDualRefinement="0", EPZSFixedPredictors="0", EPZSTemporal="0",
                 EPZSSpatialMem="0", EPZSBlockType="0", FramesToBeEncoded="4"), (192, 128), 4, "motion:14"),                                            # every optional predictor set off
    # partitions switched off (PSliceSearch*): a mode that is never searched leaves currSlice->all_mv at the zeros it was allocated with, and EPZS's block-type predictors read them
    "m2pd": (dict(RDO_OFF, SearchMode="3", SearchRange="16", NumberReferenceFrames="2", FramesToBeEncoded="4", PSliceSearch16x8="0", PSliceSearch8x4="0"), (176, 144), 4, "motion:51"),
    "m3pe": (dict(RDO_OFF, SearchMode="3", SearchRange="8", NumberReferenceFrames="3", FramesToBeEncoded="5", PSliceSearch16x8="0", PSliceSearch8x16="0", PSliceSearch8x4="0", PSliceSearch4x8="0",
                  Transform8x8Mode="1", ProfileIDC="100", SymbolMode="1", SliceMode="1", SliceArgument="27", QPISlice="26", QPPSlice="26"), (160, 96), 5, "motion:52"),
    "q1pd": (dict(RDO_OFF, SearchMode="-1", SearchRange="16", NumberReferenceFrames="1", PSliceSearch8x16="0", PSliceSearch4x8="0", PSliceSearch4x4="0"), (176, 144), 3, False),
    # a small level: the vertical vector limit (level 1.1: -256 .. 255 quarter-pels) cuts into the search range; fast full search clips its centre to limit -+ range, which need not be a whole sample
    "m3fl": (dict(RDO_OFF, SearchMode="0", SearchRange="32", NumberReferenceFrames="3", FramesToBeEncoded="5", LevelIDC="11", QPISlice="27", QPPSlice="27"), (72, 112), 5, "motion:71"),
    "m3fm": (dict(RDO_OFF, SearchMode="0", SearchRange="16", NumberReferenceFrames="3", FramesToBeEncoded="5", LevelIDC="11", QPISlice="27", QPPSlice="27"), (72, 112), 5, "motion:74"),
    "m2sl": (dict(RDO_OFF, SearchMode="-1", SearchRange="32", NumberReferenceFrames="2", FramesToBeEncoded="4", LevelIDC="11"), (120, 96), 4, "motion:72"),
    "m2el": (dict(RDO_OFF, SearchMode="3", SearchRange="32", NumberReferenceFrames="2", FramesToBeEncoded="4", LevelIDC="11"), (120, 96), 4, "motion:73"),
    # EPZS at SearchRange 2: EPZSWindowPredictorInit has no level to fill, searchPoints stays -1, and EPZSWindowPredictors -- adding it to the count -- drops the predictor before it
    "m2es": (dict(RDO_OFF, SearchMode="3", SearchRange="2", NumberReferenceFrames="2", FramesToBeEncoded="4"), (176, 144), 4, "motion:81"),
    "m5es": (dict(RDO_OFF, SearchMode="3", SearchRange="2", NumberReferenceFrames="5", FramesToBeEncoded="6", EPZSPattern="4", EPZSDualRefinement="2", EPZSAggressiveWindow="1",
                  QPISlice="30", QPPSlice="30"), (128, 64), 6, "motion:82"),
    # different chroma QP offsets for Cb and Cr (High profile): qpc[0] != qpc[1]
    "m2cq": (dict(RDO_OFF, SearchMode="-1", SearchRange="16", NumberReferenceFrames="2", FramesToBeEncoded="4", ProfileIDC="100", Transform8x8Mode="1", SymbolMode="1", CbQPOffset="3", CrQPOffset="-4",
                  QPISlice="33", QPPSlice="33"), (176, 144), 4, "motion:61"),
    # ... with QPPSlice != QPISlice as well: the P pictures' chroma QPs are not the I picture's (the harness takes both from the tap: qpc_p / qpc_v_p)
    "m2cp": (dict(RDO_OFF, SearchMode="-1", SearchRange="8", NumberReferenceFrames="2", FramesToBeEncoded="4", ProfileIDC="100", Transform8x8Mode="1", SymbolMode="0", CbQPOffset="5", CrQPOffset="-3",
                  QPISlice="27", QPPSlice="34"), (176, 144), 4, "motion:63"),
    "m2yc": (dict(RDO_OFF, NumberBFrames="0", SearchRange="16", NumberReferenceFrames="2", FramesToBeEncoded="4", CbQPOffset="-5", CrQPOffset="2", QPISlice="31", QPPSlice="31"),
             (176, 144), 4, "motion422:62", "jm_yuv422.cfg"),
    # 4:2:2 (High 4:2:2 profile): 8 x 16 chroma samples per macroblock, the 2x4 chroma DC transform with the quantiser of qpc + 3, vectors of the luma block at the same row
    # BASELINE configs[4] = encoder_yuv422.cfg (CABAC, 8x8 transform on, fast full search SR 32, five references) on its own clip, but for RDO / adaptive rounding / B pictures
    "q5y": (dict(RDO_OFF, NumberBFrames="0"), (176, 144), 3, False, "jm_yuv422.cfg"),
    "q2yv": (dict(RDO_OFF, YUVFormat="2", ProfileIDC="122", InputFile="foreman_part_qcif_422.yuv", SearchMode="-1", SearchRange="16", NumberReferenceFrames="2"), (176, 144), 3, False),   # CAVLC, 4x4 transform only
    "m3y": (dict(RDO_OFF, NumberBFrames="0", SearchMode="3", NumberReferenceFrames="3", SliceMode="1", SliceArgument="45", FramesToBeEncoded="5", QPISlice="36", QPPSlice="36"),
            (208, 160), 5, "motion422:41", "jm_yuv422.cfg"),                                                                                 # EPZS, qpc != qp, slices that start mid-row
    "m2yq": (dict(RDO_OFF, NumberBFrames="0", SearchMode="-1", SearchRange="16", NumberReferenceFrames="2", FramesToBeEncoded="4", QPISlice="12", QPPSlice="12", SymbolMode="0"),
             (176, 144), 4, "motion422:42", "jm_yuv422.cfg"),                                                                                # fine quantiser, CAVLC
    # BASELINE configs[4] at its own size: 1080p 4:2:2 synthetic, encoder_yuv422.cfg but for RDO / adaptive rounding / B pictures (and the level: five references of 1080p need 5.1)
    "g4y": (dict(RDO_OFF, NumberBFrames="0", InputFile="syn1080p422.yuv", SourceWidth="1920", SourceHeight="1080", OutputWidth="1920", OutputHeight="1080", FramesToBeEncoded="3", LevelIDC="51"),
            (1920, 1080), 3, "syn422", "jm_yuv422.cfg"),
    # EPZS with slices of whole macroblock rows (three per picture), temporal predictors on: what JMHIP_DEVICES deals to several contexts -- a band's last row reads the
    # co-located vectors of the next band's first row (me_epzs_common.c:1575-1602), so the bands' motion has to be exchanged with the samples
    "m2ed": (dict(RDO_OFF, SearchMode="3", SearchRange="16", NumberReferenceFrames="2", SliceMode="1", SliceArgument="33", FramesToBeEncoded="5"), (176, 144), 5, "motion:95"),
    # forty pictures of EPZS (two references, the shipped switches): JM's visited map is stamped with a 16-bit search count that wraps round every 65536 searches and is never
    # cleared -- a stamp left 65536 searches ago would make JM skip a candidate; the oracle counts such hits (none here), the device has no such state.  What pins a LONG EPZS
    # sequence in flight and in one launch (tests/test_gpu_seq.py)
    "m2e40": (dict(RDO_OFF, SearchMode="3", SearchRange="16", NumberReferenceFrames="2", FramesToBeEncoded="40"), (176, 144), 40, "motion:97"),
    # ---- B pictures (NumberBFrames 1, non-reference, spatial direct): coding order I P B P B ...; Main profile and up.  *b0: BiPredMotionEstimation 0; *b: as the shipped
    # encoder_main.cfg / encoder_yuv422.cfg have it (BiPredMotionEstimation 1, three refinements, range 16, sub-pel 2, 16x16 / 16x8 / 8x16)
    "q1b0": (dict(RDO_OFF, SearchMode="-1", SearchRange="16", NumberReferenceFrames="2", SymbolMode="0", BiPredMotionEstimation="0"), (176, 144), 3, False, "jm_main.cfg"),
    "q1b": (dict(RDO_OFF), (176, 144), 3, False, "jm_main.cfg"),                                      # encoder_main.cfg as shipped but for RDO / adaptive rounding
    "m3b0": (dict(RDO_OFF, SearchMode="0", SearchRange="16", NumberReferenceFrames="3", ProfileIDC="100", Transform8x8Mode="1", BiPredMotionEstimation="0",
                  FramesToBeEncoded="7", SliceMode="1", SliceArgument="50", QPISlice="26", QPPSlice="27", QPBSlice="29"), (208, 160), 7, "motion:91", "jm_main.cfg"),
    "m3b": (dict(RDO_OFF, SearchMode="0", SearchRange="32", NumberReferenceFrames="3", ProfileIDC="100", Transform8x8Mode="1", FramesToBeEncoded="7", QPISlice="30", QPPSlice="30", QPBSlice="32"),
            (208, 160), 7, "motion:92", "jm_main.cfg"),
    "m2b4": (dict(RDO_OFF, SearchMode="-1", SearchRange="8", NumberReferenceFrames="2", SymbolMode="0", BiPredMERefinements="1", BiPredMESearchRange="8", BiPredMESubPel="1",
                  FramesToBeEncoded="5", QPISlice="22", QPPSlice="22", QPBSlice="22", BList1References="2"), (176, 144), 5, "motion:93", "jm_main.cfg"),
    # encoder_yuv422.cfg with its B picture (4:2:2, CABAC, 8x8 transform, fast full search, q_offset.cfg), RDO off
    "q5yb": (dict(RDO_OFF, NumberBFrames="1", NumberReferenceFrames="4"), (176, 144), 3, False, "jm_yuv422.cfg"),
    # DirectModeType 0 (temporal direct; DirectInferenceFlag 1 as every level from 3 on requires): the shipped file otherwise (q1bt), High profile with three references, slices
    # that start mid-row and two B pictures between the references (m3bt)
    "q1bt": (dict(RDO_OFF, DirectModeType="0"), (176, 144), 3, False, "jm_main.cfg"),
    "m3bt": (dict(RDO_OFF, DirectModeType="0", SearchMode="0", SearchRange="16", NumberReferenceFrames="3", ProfileIDC="100", Transform8x8Mode="1", NumberBFrames="2",
                  FramesToBeEncoded="7", SliceMode="1", SliceArgument="50", QPISlice="27", QPPSlice="27", QPBSlice="30"), (208, 160), 7, "motion:94", "jm_main.cfg"),
    # encoder_main.cfg's search and B settings at 1080p (fast full search SR 32, CABAC, BiPredMotionEstimation 1), two references: I P B
    "g3b": (dict(RDO_OFF, **dict(SYN1080, SearchMode="0", NumberReferenceFrames="2", FramesToBeEncoded="3")), (1920, 1080), 3, True, "jm_main.cfg"),
}


def md5(b):
    return hashlib.md5(b).hexdigest()


def run(tag):
    ov, (sw, sh), nfr, syn = CASES[tag][:4]
    cfg = CASES[tag][4] if len(CASES[tag]) > 4 else "jm_baseline.cfg"
    W, H = (sw + 15) // 16 * 16, (sh + 15) // 16 * 16
    if cfg != "jm_baseline.cfg":                             # the tests read a case's settings from its overrides: spell out the ones this .cfg sets differently
        base = {}
        for name in ("jm_baseline.cfg", cfg):
            for line in open(os.path.join(G, name)):
                kv = line.split("#")[0].split("=")
                if len(kv) == 2:
                    base.setdefault(name, {})[kv[0].strip()] = kv[1].strip().strip('"')
        for k in ("SymbolMode", "SearchMode", "SearchRange", "Transform8x8Mode", "NumberReferenceFrames", "ProfileIDC", "LevelIDC", "YUVFormat", "InputFile", "FramesToBeEncoded",
                  "SliceMode", "SliceArgument", "QPISlice", "QPPSlice", "DFDisableRefPSlice", "OffsetMatrixPresentFlag", "NumberBFrames", "BiPredMotionEstimation", "BiPredMERefinements",
                  "BiPredMESearchRange", "BiPredMESubPel", "BiPredSearch16x16", "BiPredSearch16x8", "BiPredSearch8x16", "BiPredSearch8x8", "DirectInferenceFlag", "DirectModeType", "QPBSlice",
                  "BList0References", "BList1References"):
            if k in base[cfg] and k not in ov:
                ov = dict(ov, **{k: base[cfg][k]})
    tmp = tempfile.mkdtemp(prefix="mbgold_")
    try:
        for f in ("foreman_part_qcif.yuv", "foreman_part_qcif_422.yuv", "q_offset.cfg"):
            shutil.copyfile(os.path.join(G, f), os.path.join(tmp, f))
        clip_md5 = ""
        if syn is True:
            import bench
            bench.write_yuv(os.path.join(tmp, "syn1080p.yuv"), nfr)
        elif syn == "syn422":
            import synclip
            synclip.syn1080p422(os.path.join(tmp, "syn1080p422.yuv"), nfr)
            clip_md5 = md5(open(os.path.join(tmp, "syn1080p422.yuv"), "rb").read())
        elif syn:                                            # "motion:<seed>" / "motion422:<seed>": tests/golden/synth_motion.py at the case's size
            import synth_motion
            data = np.concatenate(synth_motion.motion_clip(sw, sh, nfr, int(syn.split(":")[1]), yuv422=syn.startswith("motion422")))
            data.tofile(os.path.join(tmp, "motion.yuv"))
            clip_md5 = md5(data.tobytes())
            ov = dict(ov, InputFile="motion.yuv", SourceWidth=str(sw), SourceHeight=str(sh), OutputWidth=str(sw), OutputHeight=str(sh))
        args = [EXE, "-d", os.path.join(G, cfg)]
        for k, v in dict(ov, OutputFile="o.264", ReconFile="o_rec.yuv", TraceFile="/dev/null").items():
            args += ["-p", f"{k}={v}"]
        subprocess.run(args, cwd=tmp, env=dict(os.environ, JM_TAP_DIR=tmp), check=True, stdout=subprocess.DEVNULL)
        tap = mb_tap.read(os.path.join(tmp, "mb_low.bin"))
        nmb = (W // 16) * (H // 16)
        assert len(tap) == nmb * nfr, (len(tap), nmb, nfr)
        pre = []
        for n in range(nfr):
            T = tap[n * nmb:(n + 1) * nmb]
            y = T["rec_y"].reshape(H // 16, W // 16, 16, 16).transpose(0, 2, 1, 3).reshape(H, W)
            if int(T["yuv_format"][0]) == 2:
                u = np.concatenate([T["rec_u"], T["rec_u2"]], axis=1).reshape(H // 16, W // 16, 16, 8).transpose(0, 2, 1, 3).reshape(H, W // 2)
                v = np.concatenate([T["rec_v"], T["rec_v2"]], axis=1).reshape(H // 16, W // 16, 16, 8).transpose(0, 2, 1, 3).reshape(H, W // 2)
            else:
                u = T["rec_u"].reshape(H // 16, W // 16, 8, 8).transpose(0, 2, 1, 3).reshape(H // 2, W // 2)
                v = T["rec_v"].reshape(H // 16, W // 16, 8, 8).transpose(0, 2, 1, 3).reshape(H // 2, W // 2)
            pre.append([md5(np.ascontiguousarray(p).tobytes()) for p in (y, u, v)])
        t0 = tap[0]
        lam = {int(t["slice_type"]): list(t["lambda_mf"]) + [int(t["lambda_mdfp"])] for t in tap[::nmb]}
        bfirst = [t for t in tap[::nmb] if int(t["slice_type"]) == 1]
        bextra = {}
        if bfirst:                                          # B pictures: the lists as the encoder ordered them (picture order counts), the B slices' QPs and lambdas
            bextra = dict(lambda_b=np.array(lam[1], np.int32), qp_b=int(bfirst[0]["qp"]), qpc_b=int(bfirst[0]["qpc"]), qpc_v_b=int(bfirst[0]["qpc_v"]),
                          num_ref_pic=tap["num_ref"][::nmb].astype(np.int32), num_ref1_pic=tap["num_ref1"][::nmb].astype(np.int32), poc_l1=tap["poc_l1"][::nmb].astype(np.int32),
                          frame_no=tap["frame_no"][::nmb].astype(np.int32), direct_8x8_inference=int(bfirst[0]["direct_8x8_inference"]),
                          motion_cost1=tap["motion_cost1"][:, 1:, 0, :].astype(np.int64) if sw * sh < 200000 else np.zeros(0, np.int64))
        np.savez_compressed(os.path.join(G, f"mb_low_{tag}.npz"),
                            records=mb_tap.tap_to_records(tap, int(ov.get("SymbolMode", 0))), slice_type=tap["slice_type"][::nmb].astype(np.int32), slice_nr=tap["slice_nr"].astype(np.int16),
                            lambda_i=np.array(lam.get(2, [0, 0, 0, 0]), np.int32), lambda_p=np.array(lam.get(0, [0, 0, 0, 0]), np.int32),
                            qp=int(t0["qp"]), qpc=int(t0["qpc"]), search_range=int(t0["search_range"]), max_mvd=int(t0["max_mvd"]),
                            mv_limit=t0["mv_limit"].astype(np.int32), num_ref=int(tap["num_ref"].max()), size=np.array([sw, sh, W, H], np.int32),
                            poc=tap["poc"][::nmb].astype(np.int32), ref_poc=tap["ref_poc"][::nmb].astype(np.int32),
                            motion_cost_ref=tap["motion_cost_ref"][:, 1:, :, :].astype(np.int64) if ov.get("SearchMode") == "3" else np.zeros(0, np.int64),
                            motion_cost=tap["motion_cost"][:, 1:, :].astype(np.int64) if sw * sh < 200000 else np.zeros(0, np.int64),
                            md5_264=md5(open(os.path.join(tmp, "o.264"), "rb").read()), md5_recon=md5(open(os.path.join(tmp, "o_rec.yuv"), "rb").read()),
                            md5_pre_deblock=np.array(pre), overrides=np.array(sorted(f"{k}={v}" for k, v in ov.items())), clip=str(syn), clip_md5=clip_md5, cfg=cfg, yuv_format=int(t0["yuv_format"]), qpc_v=int(t0["qpc_v"]),
                            qp_p=int(([int(t["qp"]) for t in tap[::nmb] if int(t["slice_type"]) == 0] or [int(t0["qp"])])[0]),
                            qpc_p=int(([int(t["qpc"]) for t in tap[::nmb] if int(t["slice_type"]) == 0] or [int(t0["qpc"])])[0]),      # the P pictures' chroma QPs (QPPSlice != QPISlice with chroma offsets)
                            qpc_v_p=int(([int(t["qpc_v"]) for t in tap[::nmb] if int(t["slice_type"]) == 0] or [int(t0["qpc_v"])])[0]), **bextra)
        print(tag, "records", len(tap), "md5", md5(open(os.path.join(tmp, "o.264"), "rb").read()))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-f", os.path.join(ROOT, "oracle", "Makefile.ref"), "all", "tapmb"])
    for tag in (sys.argv[1:] or CASES):
        run(tag)
