#!/usr/bin/env python3
"""tests/golden/g2_sideinfo.npz: what JM's DeblockFrame was given for BASELINE.json configs[1] (G2: synthetic 1080p, FullSearch
SR=32, I + P) -- the per-macroblock records (mb_type, slice_type, qp, qpc[2], cbp, cbp_blk, slice_nr, DFDisableIdc,
DFAlphaC0Offset, DFBetaOffset, transform_8x8) and the per-4x4 motion (mv_x, mv_y, reference identity per list) of the I and the P
picture, captured by the tap of oracle/ref_tap.c in the unmodified reference encoder.  bench.py feeds the P picture's records to
the deblocking stage, so that stage works on the real mode / coefficient / motion statistics of the configuration the metric is
quoted on (78 % skipped macroblocks, 8 % with coefficients, 0.2 % intra) instead of made-up ones.  Data only; runs in the build
container (needs /root/reference and oracle/_ref/); ~20 s."""
import hashlib, os, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import bench
import make_golden as G


def main():
    import json
    with tempfile.TemporaryDirectory(prefix="jmg2_") as tmp:
        bench.write_yuv(os.path.join(tmp, "syn1080p.yuv"), 2)
        G.run(G.TAP, "encoder_baseline.cfg", dict(InputFile="syn1080p.yuv", SourceWidth=1920, SourceHeight=1080, OutputWidth=1920, OutputHeight=1080,
                                                  FramesToBeEncoded=2, SearchMode=-1, SearchRange=32, NumberReferenceFrames=1, LevelIDC=51), tmp, tap=True, tap_max=10)
        got = G.md5(os.path.join(tmp, "o.264"))
        want = json.load(open(os.path.join(HERE, "md5.json")))["G2"]
        want = want["md5_264"]
        assert got == want, (got, want)
        recs = G.read_deblock(os.path.join(tmp, "deblock.bin"))
        assert len(recs) == 2 and recs[1]["w"] == 1920 and recs[1]["h"] == 1088
        d = {}
        for tag, r in (("i", recs[0]), ("p", recs[1])):
            d[tag + "_mbs"] = r["mbs"].astype(np.int16)
            d[tag + "_mot"] = r["mot"].astype(np.int16)
            d[tag + "_d8"] = np.int32(r["d8"])
            d[tag + "_sha_pre_post"] = np.array([hashlib.sha256(p.tobytes()).hexdigest() for p in r["pre"] + r["post"]])
        # the P picture's source as JM holds it after read_one_frame + pad_borders (1080 -> 1088 rows; dumped when its first motion search
        # starts: clip frame 1): sha256 of Y, U, V
        r = G.Reader(os.path.join(tmp, "cur_yuv.bin"))
        idx, fmt = r.i32(), r.i32()
        assert (idx, fmt) == (1, 1) and r.b[r.o:r.o + 8] == np.array([1088, 1920], np.int32).tobytes()
        d["p_cur_yuv_sha"] = np.array([hashlib.sha256(r.plane().astype(np.uint8).tobytes()).hexdigest() for _ in range(3)])
        np.savez_compressed(os.path.join(HERE, "g2_sideinfo.npz"), **d)
        print("wrote", os.path.join(HERE, "g2_sideinfo.npz"), os.path.getsize(os.path.join(HERE, "g2_sideinfo.npz")), "bytes")


if __name__ == "__main__":
    main()
