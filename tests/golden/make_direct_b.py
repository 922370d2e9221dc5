#!/usr/bin/env python3
"""Golden vectors for the spatial direct mode of B slices (oracle/jmo_direct.c): per macroblock of every B slice what the REAL encoder's Get_Direct_MV_Spatial_Normal
(lencod/src/mv_direct.c:522) read -- the neighbours A, B, C and the co-located picture's motion -- and what it left (direct_ref_idx, direct_pdir, the vectors), dumped by the tap
oracle/ref_tap_mb.c (tap_b_slice; oracle/_ref/lencod_tapmb.exe is the unmodified lencod objects + that file, `make -f oracle/Makefile.ref tapmb`).
Runs only where /root/reference exists; writes tests/golden/direct_b.npz.     usage: python tests/golden/make_direct_b.py"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

G = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(G))
sys.path.insert(0, G)
EXE = os.path.join(ROOT, "oracle", "_ref", "lencod_tapmb.exe")
# mirrors MBREC_B of oracle/ref_tap_mb.c (natural alignment: two bytes of padding at the end)
MBREC_B = np.dtype([("frame_no", "<i4"), ("mb_addr", "<i4"), ("slice_nr", "<i4"), ("direct_8x8_inference", "<i4"), ("weighted_bipred_idc", "<i4"), ("num_ref", "<i4", 2),
                    ("col_long_term", "<i4"), ("nb_avail", "i1", 4), ("nb_ref", "i1", (3, 2)), ("nb_mv", "<i2", (3, 2, 2)), ("direct_ref_idx", "i1", (16, 2)), ("direct_pdir", "i1", 16),
                    ("direct_mv", "<i2", (16, 2, 2)), ("col_ref", "i1", (16, 2)), ("col_mv", "<i2", (16, 2, 2)), ("pad_", "u1", 2)])
assert MBREC_B.itemsize == 404
COMMON = dict(ProfileIDC="77", RDOptimization="0", AdaptiveRounding="0", BiPredMotionEstimation="0", DirectModeType="1", SearchMode="-1", SearchRange="16", LevelIDC="40")
# tag: (overrides, clip) -- clip None = the reference's foreman_part_qcif.yuv, else (width, height, frames, seed) of tests/golden/synth_motion.py
CASES = {
    "qcif_b1": (dict(FramesToBeEncoded="3", NumberBFrames="1", NumberReferenceFrames="2", DirectInferenceFlag="1"), None),
    "motion_b1": (dict(FramesToBeEncoded="7", NumberBFrames="1", NumberReferenceFrames="3", DirectInferenceFlag="1"), (208, 160, 7, 71)),
    "motion_b2_4x4": (dict(FramesToBeEncoded="7", NumberBFrames="2", NumberReferenceFrames="2", DirectInferenceFlag="0", LevelIDC="21", SliceMode="1", SliceArgument="50"), (208, 160, 7, 72)),
    "motion_cabac": (dict(FramesToBeEncoded="5", NumberBFrames="1", NumberReferenceFrames="5", DirectInferenceFlag="1", SymbolMode="1", SearchRange="32"), (176, 144, 5, 73)),
}


def run(tag):
    ov, clip = CASES[tag]
    ov = dict(COMMON, **ov)
    tmp = tempfile.mkdtemp(prefix="directb_")
    try:
        shutil.copyfile(os.path.join(G, "q_offset.cfg"), os.path.join(tmp, "q_offset.cfg"))
        if clip is None:
            shutil.copyfile(os.path.join(G, "foreman_part_qcif.yuv"), os.path.join(tmp, "foreman_part_qcif.yuv"))
            ov["InputFile"] = "foreman_part_qcif.yuv"
        else:
            import synth_motion
            w, h, n, seed = clip
            np.concatenate(synth_motion.motion_clip(w, h, n, seed)).tofile(os.path.join(tmp, "motion.yuv"))
            ov.update(InputFile="motion.yuv", SourceWidth=str(w), SourceHeight=str(h), OutputWidth=str(w), OutputHeight=str(h))
        args = [EXE, "-d", os.path.join(G, "jm_baseline.cfg")]
        for k, v in dict(ov, OutputFile="o.264", ReconFile="o_rec.yuv", TraceFile="/dev/null").items():
            args += ["-p", f"{k}={v}"]
        subprocess.run(args, cwd=tmp, env=dict(os.environ, JM_TAP_DIR=tmp, JM_TAPMB_B="1"), check=True, stdout=subprocess.DEVNULL)
        return np.fromfile(os.path.join(tmp, "mb_low_b.bin"), MBREC_B), [f"{k}={v}" for k, v in sorted(ov.items())]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    out = {}
    for tag in CASES:
        recs, ov = run(tag)
        out[tag] = recs
        out[tag + "_overrides"] = np.array(ov)
        print(tag, len(recs), "macroblocks of B slices; pdir histogram (-1, 0, 1, 2):", np.bincount(recs["direct_pdir"].ravel() + 1, minlength=4).tolist(),
              "references used:", np.unique(recs["direct_ref_idx"]).tolist())
    np.savez_compressed(os.path.join(G, "direct_b.npz"), **out)
