#!/usr/bin/env python3
"""Randomised pinning of the ORACLE (CPU only, needs /root/reference's build in oracle/_ref: the build container): seeded random RDO-off configurations through the tapped real
encoder (tests/golden/make_mb_golden.py: oracle/_ref/lencod_tapmb.exe) and through oracle/jmo_mbenc.c (tests/test_oracle_mbenc.run_case) -- every macroblock's record, motion
costs, reconstruction must be equal.  The committed goldens pin the oracle on fixed cases; this checks it on configurations nobody chose.  TEST INFRASTRUCTURE.
usage: python tests/fuzz_oracle.py <seconds> [first seed] [workers]"""
import multiprocessing as mp
import os
import subprocess
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT); sys.path.insert(0, G); sys.path.insert(0, os.path.join(ROOT, "tests"))


def config(seed):
    rng = np.random.default_rng(seed)
    yuv = int(rng.choice([1, 1, 2]))
    t8, cabac = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    prof = 122 if yuv == 2 else (100 if t8 else (77 if cabac else 66))
    sw, sh = 16 * int(rng.integers(3, 12)) - int(rng.choice([0, 0, 2, 8])), 16 * int(rng.integers(3, 9)) - int(rng.choice([0, 0, 2, 6]))
    nfr = int(rng.integers(3, 6))
    if seed >= 800000 and rng.integers(0, 5) == 0:           # now and then a CIF-sized picture or a longer sequence (all five references in use)
        if rng.integers(0, 2):
            sw, sh = 16 * int(rng.integers(16, 23)), 16 * int(rng.integers(12, 19))
        else:
            nfr = int(rng.integers(6, 10))
    sm = int(rng.choice([-1, -1, 0, 3, 3]))
    R = int(rng.choice([4, 8, 16, 32])) if rng.integers(0, 3) else int(rng.integers(2, 33))
    qp = int(rng.integers(10, 45))
    ov = dict(RDOptimization=0, AdaptiveRounding=0, NumberBFrames=0, FramesToBeEncoded=nfr, YUVFormat=yuv, ProfileIDC=prof, LevelIDC=40, SymbolMode=cabac, Transform8x8Mode=t8, SearchMode=sm,
              SearchRange=R, NumberReferenceFrames=int(rng.integers(1, 5 if R > 16 else 6)), QPISlice=qp, QPPSlice=qp if rng.integers(0, 2) else int(rng.integers(10, 45)),
              OffsetMatrixPresentFlag=int(yuv == 2 and rng.integers(0, 2)), SliceMode=0, SliceArgument=50, DFDisableRefPSlice=0)
    if rng.integers(0, 3) == 0:
        ov.update(SliceMode=1, SliceArgument=int(rng.integers(3, ((sw + 15) // 16) * ((sh + 15) // 16))))
    if rng.integers(0, 4) == 0:
        for k in ("16x8", "8x16", "8x4", "4x8", "4x4"):
            ov["PSliceSearch" + k] = int(rng.integers(0, 2))
    if sm == 3:
        ov.update(EPZSPattern=int(rng.integers(0, 6)), EPZSDualRefinement=int(rng.integers(0, 7)), EPZSFixedPredictors=int(rng.integers(0, 4)), EPZSTemporal=int(rng.integers(0, 2)),
                  EPZSSpatialMem=int(rng.integers(0, 2)), EPZSBlockType=int(rng.integers(0, 2)), EPZSAggressiveWindow=int(rng.integers(0, 2)))
    if prof >= 100 and rng.integers(0, 3) == 0:
        ov.update(CbQPOffset=int(rng.integers(-6, 7)), CrQPOffset=int(rng.integers(-6, 7)))
    if sm != 0 and rng.integers(0, 4) == 0:
        ov.update(LevelIDC=int(rng.choice([11, 20, 30])))
    if seed >= 900000 and rng.integers(0, 3) == 0:           # the loop filter switched off / kept inside the slices (as golden q4s)
        d = int(rng.choice([1, 2]))
        ov.update(DFParametersFlag=1, DFDisableRefISlice=d, DFDisableNRefISlice=d, DFDisableRefPSlice=d, DFDisableNRefPSlice=d)
    clip = ("motion422:" if yuv == 2 else "motion:") + str(seed)
    case = ({k: str(v) for k, v in ov.items()}, (sw, sh), nfr, clip)
    return case + (("jm_yuv422.cfg",) if yuv == 2 else ())


def config_b(seed):
    """seeds from 1 000 000: sequences with non-reference B pictures (spatial direct; jm_main.cfg / jm_yuv422.cfg as the base), every B switch the oracle restates drawn at random"""
    rng = np.random.default_rng(seed)
    yuv = int(rng.choice([1, 1, 1, 2]))
    t8, cabac = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    prof = 122 if yuv == 2 else (100 if t8 else 77)
    sw, sh = 16 * int(rng.integers(3, 12)) - int(rng.choice([0, 0, 2, 8])), 16 * int(rng.integers(3, 9)) - int(rng.choice([0, 0, 2, 6]))
    nb = int(rng.choice([1, 1, 2]))
    nfr = int(rng.integers(3, 8))
    sm = int(rng.choice([-1, 0]))
    R = int(rng.choice([4, 8, 16, 32])) if rng.integers(0, 3) else int(rng.integers(2, 33))
    qp = int(rng.integers(10, 45))
    nref = int(rng.integers(1, 4 if R > 16 else 5))
    ov = dict(RDOptimization=0, AdaptiveRounding=0, NumberBFrames=nb, FramesToBeEncoded=nfr, YUVFormat=yuv, ProfileIDC=prof, LevelIDC=40, SymbolMode=cabac, Transform8x8Mode=t8, SearchMode=sm,
              SearchRange=R, NumberReferenceFrames=nref, QPISlice=qp, QPPSlice=qp if rng.integers(0, 2) else int(rng.integers(10, 45)), QPBSlice=int(rng.integers(10, 45)),
              OffsetMatrixPresentFlag=int(yuv == 2 and rng.integers(0, 2)), SliceMode=0, SliceArgument=50, DFDisableRefPSlice=0, DirectModeType=1 if seed < 1100000 else int(rng.integers(0, 2)),
              DirectInferenceFlag=int(rng.integers(0, 2)), BiPredMotionEstimation=int(rng.integers(0, 3) > 0))
    if ov["BiPredMotionEstimation"]:
        ov.update(BiPredMERefinements=int(rng.integers(0, 4)), BiPredMESearchRange=int(rng.choice([2, 4, 8, 16])), BiPredMESubPel=int(rng.integers(0, 3)),
                  BiPredSearch16x16=int(rng.integers(0, 4) > 0), BiPredSearch16x8=int(rng.integers(0, 2)), BiPredSearch8x16=int(rng.integers(0, 2)), BiPredSearch8x8=0)
    if rng.integers(0, 2):
        ov.update(BList0References=int(rng.integers(0, nref + 1)), BList1References=int(rng.integers(0, 3)))
    if rng.integers(0, 3) == 0:
        ov.update(SliceMode=1, SliceArgument=int(rng.integers(3, ((sw + 15) // 16) * ((sh + 15) // 16))))
    if rng.integers(0, 4) == 0:
        for k in ("16x8", "8x16", "8x4", "4x8", "4x4"):
            ov["PSliceSearch" + k] = int(rng.integers(0, 2))
    if rng.integers(0, 3) == 0:
        for k in ("16x16", "16x8", "8x16", "8x8", "8x4", "4x8", "4x4"):
            ov["BSliceSearch" + k] = int(rng.integers(0, 4) > 0)
        ov["BSliceDirect"] = int(rng.integers(0, 3) > 0)
    if prof >= 100 and rng.integers(0, 3) == 0:
        ov.update(CbQPOffset=int(rng.integers(-6, 7)), CrQPOffset=int(rng.integers(-6, 7)))
    if rng.integers(0, 4) == 0:
        d = int(rng.choice([1, 2]))
        ov.update(DFParametersFlag=1, DFDisableRefISlice=d, DFDisableNRefISlice=d, DFDisableRefPSlice=d, DFDisableNRefPSlice=d, DFDisableRefBSlice=d, DFDisableNRefBSlice=d)
    clip = ("motion422:" if yuv == 2 else "motion:") + str(seed)
    return ({k: str(v) for k, v in ov.items()}, (sw, sh), nfr, clip, "jm_yuv422.cfg" if yuv == 2 else "jm_main.cfg")


def one(seed):
    import make_mb_golden as M
    import test_oracle_mbenc as T
    tag = f"zz{seed}"
    M.CASES[tag] = config_b(seed) if seed >= 1000000 else config(seed)
    path = os.path.join(G, f"mb_low_{tag}.npz")
    try:
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            M.run(tag)
        (T.run_case_b if seed >= 1000000 else T.run_case)(tag)
        return seed, None
    except subprocess.CalledProcessError:                    # JM itself refuses the configuration (a level too small for the picture / the references): not a case
        return seed, "refused"
    except BaseException as e:                               # an assertion of run_case
        return seed, f"{type(e).__name__}: {str(e)[:300]} {M.CASES[tag][0]}"
    finally:
        if os.path.exists(path):
            os.remove(path)


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 700000
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    t0, done, bad, refused = time.time(), 0, 0, 0
    with mp.Pool(workers) as pool:
        while time.time() - t0 < budget:
            for s, err in pool.map(one, range(seed, seed + 4 * workers)):
                if err == "refused":
                    refused += 1
                    continue
                done += 1
                if err:
                    bad += 1
                    print("FAILED", s, err, flush=True)
            seed += 4 * workers
    print(f"fuzz_oracle: {done} random configurations through the tapped encoder and the oracle in {time.time() - t0:.0f} s, {bad} NOT equal ({refused} more refused by JM itself); next seed {seed}")
