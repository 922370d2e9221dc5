#!/usr/bin/env python3
"""Randomised end-to-end parity run (GPU box): the unmodified encoder with the library linked in (oracle/_ref/lencod_hip.exe) against CPU JM (oracle/_ref/lencod.exe) on
seeded random RDO-off configurations -- the .264 and the reconstruction file must be byte-identical whether the macroblock pipeline takes the sequence or the adapter turns it
away (then JM's own function or the per-call kernels run).  Varied: search mode / range, references, QPs of I and P slices, chroma QP offset, entropy coder, 8x8 transform, 4:2:0 /
4:2:2, slices, loop filter parameters and disable flags, partition switches, sub-pel on / off, intra period, picture size (cropped sources), EPZS switches; from seed 500000 on also
intra modes switched off, PList0References, UseMVLimits, EPZS threshold scales, explicit lambda weights, reference reordering, PicOrderCntType 2, IDRPeriod, ChangeQPFrame.
TEST INFRASTRUCTURE.   usage: python tests/fuzz_dropin.py <seconds> [first seed]"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, G)
import synth_motion

CPU, HIP = os.path.join(ROOT, "oracle", "_ref", "lencod.exe"), os.path.join(ROOT, "oracle", "_ref", "lencod_hip.exe")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()
t0, done, piped, failed = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    MORE = seed >= 500000
    yuv = int(rng.choice([1, 1, 2]))
    t8 = int(rng.integers(0, 2))
    cabac = int(rng.integers(0, 2))
    prof = 122 if yuv == 2 else (100 if t8 else (77 if cabac else 66))
    sw, sh = 16 * int(rng.integers(3, 12)) - int(rng.choice([0, 0, 2, 8])), 16 * int(rng.integers(3, 9)) - int(rng.choice([0, 0, 2, 6]))
    if rng.integers(0, 12) == 0:                              # now and then a CIF-sized picture
        sw, sh = 16 * int(rng.integers(16, 23)), 16 * int(rng.integers(12, 19))
    nfr = int(rng.integers(3, 6)) if rng.integers(0, 10) else int(rng.integers(6, 10))
    sm = int(rng.choice([-1, -1, 0, 3]))
    ov = dict(RDOptimization=0, AdaptiveRounding=0, InputFile="clip.yuv", SourceWidth=sw, SourceHeight=sh, OutputWidth=sw, OutputHeight=sh, FramesToBeEncoded=nfr, YUVFormat=yuv,
              ProfileIDC=prof, LevelIDC=40, SymbolMode=cabac, Transform8x8Mode=t8, SearchMode=sm, SearchRange=int(rng.choice([4, 8, 16, 32])), NumberReferenceFrames=int(rng.integers(1, 6)),
              QPISlice=int(rng.integers(10, 45)), QPPSlice=int(rng.integers(10, 45)), ChromaQPOffset=int(rng.choice([0, 0, 0, -3, 4])), DisableSubpelME=int(rng.choice([0, 0, 0, 1])),
              IntraPeriod=int(rng.choice([0, 0, 2, 3])), NumberBFrames=0)
    if rng.integers(0, 3) == 0:
        nmb = ((sw + 15) // 16) * ((sh + 15) // 16)
        ov.update(SliceMode=1, SliceArgument=int(rng.integers(3, nmb)))
    if rng.integers(0, 3) == 0:
        ov.update(DFParametersFlag=1)
        for k in ("RefISlice", "NRefISlice", "RefPSlice", "NRefPSlice"):
            ov["DFDisable" + k] = int(rng.choice([0, 0, 1, 2])); ov["DFAlpha" + k] = int(rng.integers(-3, 4)); ov["DFBeta" + k] = int(rng.integers(-3, 4))
    if rng.integers(0, 4) == 0:
        for k in ("16x8", "8x16", "8x4", "4x8", "4x4"):
            ov["PSliceSearch" + k] = int(rng.integers(0, 2))
        if t8:
            ov["PSliceSearch8x8"] = 1
    if sm == 3:
        ov.update(EPZSPattern=int(rng.integers(0, 6)), EPZSDualRefinement=int(rng.integers(0, 7)), EPZSFixedPredictors=int(rng.integers(0, 4)), EPZSTemporal=int(rng.integers(0, 2)),
                  EPZSSpatialMem=int(rng.integers(0, 2)), EPZSBlockType=int(rng.integers(0, 2)), EPZSAggressiveWindow=int(rng.integers(0, 2)))
    if yuv == 2 and rng.integers(0, 2):
        ov.update(OffsetMatrixPresentFlag=1)
    if prof >= 100 and rng.integers(0, 3) == 0:               # the High profiles' separate chroma offsets
        ov.update(CbQPOffset=int(rng.integers(-6, 7)), CrQPOffset=int(rng.integers(-6, 7)))
    if rng.integers(0, 4) == 0:                               # smaller levels: tighter vertical vector limits (conformance.c:604-631)
        ov.update(LevelIDC=int(rng.choice([11, 20, 30])))
    if rng.integers(0, 8) == 0:
        ov.update(QPISlice=int(rng.choice([0, 1, 50, 51])), QPPSlice=int(rng.choice([0, 2, 49, 51])))
    if rng.integers(0, 3) == 0:                               # any search range, more references with the smaller ones, skipped source frames (picture order count distances)
        ov.update(SearchRange=int(rng.integers(2, 33)))
    if ov["SearchRange"] <= 16 and rng.integers(0, 4) == 0:
        ov.update(NumberReferenceFrames=int(rng.integers(5, 9)), FramesToBeEncoded=int(rng.integers(6, 10)))
    if rng.integers(0, 5) == 0:
        ov.update(FrameSkip=int(rng.integers(1, 3)))
    if MORE:                                                  # round 3's last set of dimensions (drawn after everything else: the earlier seeds keep their meaning)
        if rng.integers(0, 6) == 0:
            ov.update(DisableIntraInInter=1)
        if rng.integers(0, 8) == 0:
            ov[str(rng.choice(["DisableIntra4x4", "DisableIntra16x16"]))] = 1
        if rng.integers(0, 6) == 0:
            ov.update(PList0References=int(rng.integers(1, int(ov["NumberReferenceFrames"]) + 1)))
        if rng.integers(0, 6) == 0:
            ov.update(UseMVLimits=1, SetMVXLimit=int(rng.choice([8, 16, 40, 512])), SetMVYLimit=int(rng.choice([8, 16, 40, 512])))
        if sm == 3 and rng.integers(0, 3) == 0:
            ov.update(EPZSMinThresScale=int(rng.integers(0, 3)), EPZSMedThresScale=int(rng.integers(0, 3)), EPZSMaxThresScale=int(rng.integers(0, 4)), EPZSSubPelThresScale=int(rng.integers(0, 4)))
        if rng.integers(0, 6) == 0:
            ov.update(UseExplicitLambdaParams=1, LambdaWeightPSlice=float(rng.choice([0.3, 0.68, 1.4])), LambdaWeightISlice=float(rng.choice([0.3, 0.65, 1.2])))
        if rng.integers(0, 8) == 0:
            ov.update(ReferenceReorder=1, PocMemoryManagement=1)
        if rng.integers(0, 8) == 0:
            ov.update(PicOrderCntType=2)
        if rng.integers(0, 8) == 0:
            ov.update(IDRPeriod=int(rng.choice([2, 3])))
        if rng.integers(0, 8) == 0:
            ov.update(ChangeQPFrame=2, ChangeQPI=int(rng.integers(10, 45)), ChangeQPP=int(rng.integers(10, 45)))
    if seed >= 3000000:                                       # sequences with B pictures (round 5): what the pipeline takes (non-reference, spatial direct, full searches) and what it turns away
        nb = int(rng.choice([1, 1, 2, 3]))
        ov.update(NumberBFrames=nb, ProfileIDC=max(prof, 77), DirectModeType=int(rng.choice([1, 1, 1, 1, 0])), DirectInferenceFlag=int(rng.integers(0, 2)), QPBSlice=int(rng.integers(10, 45)),
                  BiPredMotionEstimation=int(rng.integers(0, 3) > 0), FramesToBeEncoded=int(rng.integers(3, 10)), IntraPeriod=int(rng.choice([0, 0, 0, 0, 4])))
        if ov["BiPredMotionEstimation"]:
            ov.update(BiPredMERefinements=int(rng.integers(0, 4)), BiPredMESearchRange=int(rng.choice([r for r in (2, 4, 8, 16, 16, 16, 32) if r <= int(ov["SearchRange"])] or [int(ov["SearchRange"])])),
                      BiPredMESubPel=int(rng.integers(0, 3)), BiPredSearch16x16=int(rng.integers(0, 4) > 0), BiPredSearch16x8=int(rng.integers(0, 2)), BiPredSearch8x16=int(rng.integers(0, 2)),
                      BiPredSearch8x8=int(rng.integers(0, 6) == 0))
        if rng.integers(0, 2):
            ov.update(BList0References=int(rng.integers(0, int(ov["NumberReferenceFrames"]) + 1)), BList1References=int(rng.integers(0, 3)))
        if rng.integers(0, 4) == 0:
            for k in ("16x16", "16x8", "8x16", "8x8", "8x4", "4x8", "4x4"):
                ov["BSliceSearch" + k] = int(rng.integers(0, 4) > 0)
            ov["BSliceDirect"] = int(rng.integers(0, 3) > 0)
            if t8:
                ov["BSliceSearch8x8"] = 1
        if "DFParametersFlag" in ov:
            for k in ("RefBSlice", "NRefBSlice"):
                ov["DFDisable" + k] = int(rng.choice([0, 0, 1, 2])); ov["DFAlpha" + k] = int(rng.integers(-3, 4)); ov["DFBeta" + k] = int(rng.integers(-3, 4))
        if rng.integers(0, 10) == 0:
            ov.update(BReferencePictures=1)
        if rng.integers(0, 10) == 0:
            ov.update(HierarchicalCoding=int(rng.choice([1, 2])))
        if rng.integers(0, 6) == 0:
            ov.update(WeightedBiprediction=int(rng.choice([1, 2])))
        ov.pop("ChangeQPFrame", None); ov.pop("ChangeQPI", None); ov.pop("ChangeQPP", None)
    if os.environ.get("FUZZ_LIST"):                            # (the generator alone: which seed made a configuration)
        print(seed, (sw, sh), ov, flush=True)
        seed += 1; done += 1
        if done >= int(os.environ["FUZZ_LIST"]):
            break
        continue
    tmp = tempfile.mkdtemp(prefix="fz_")
    try:
        nsrc = (int(ov["FramesToBeEncoded"]) - 1) * (1 + int(ov.get("FrameSkip", 0))) + 1
        np.concatenate(synth_motion.motion_clip(sw, sh, nsrc, seed, yuv422=yuv == 2)).tofile(os.path.join(tmp, "clip.yuv"))
        shutil.copyfile(os.path.join(G, "q_offset.cfg"), os.path.join(tmp, "q_offset.cfg"))
        res = []
        for exe, tag in ((CPU, "c"),) if os.environ.get("FUZZ_CPU_ONLY") else ((CPU, "c"), (HIP, "h")):
            args = [exe, "-d", os.path.join(G, "jm_baseline.cfg")]
            for k, v in dict(ov, OutputFile=f"{tag}.264", ReconFile=f"{tag}.yuv", TraceFile="/dev/null").items():
                args += ["-p", f"{k}={v}"]
            env = dict(os.environ)
            if exe == HIP and int(ov.get("NumberBFrames", 0)) and int(ov.get("BiPredMotionEstimation", 0)):
                # a sequence the pipeline turns away falls to the per-call kernels, and the bi-predictive full search then asks for millions of single candidate distortions, a
                # round trip each (part evalp; covered by the fixed cases G3a / G3b / G3w / G3wb of tests/test_lencod_dropin.py): every part but that one
                env["JMHIP_ADAPTER_PARTS"] = "interp,fs,subpel,ffs,deblock,tq4,tq8,tqc,tq16,mcl,mcc,eval,evalbatch,ip4,ip8,i16,ic,interpc,load,mbpipe,nulltrace,readframe"
            r = subprocess.run(args, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
            res.append(r)
        if os.environ.get("FUZZ_CPU_ONLY"):                    # (a dry run of the generator where there is no GPU: does JM take the configuration?)
            print(seed, res[0].returncode, res[0].stderr.decode(errors="replace")[-200:] if res[0].returncode else "", flush=True)
            seed += 1; done += 1
            continue
        c, h = res
        if c.returncode != 0:                                 # a configuration JM itself refuses: both must refuse
            assert h.returncode != 0, ("CPU JM refused, the drop-in did not", ov)
            seed += 1
            continue
        err = h.stderr.decode(errors="replace")
        if h.returncode != 0:
            print("FAILED", seed, ov, "the drop-in exits with", h.returncode, [l for l in err.splitlines() if "jmhip" in l and "on the MI355X" not in l][-3:], flush=True)
            failed += 1; done += 1; seed += 1
            continue
        same = md5(os.path.join(tmp, "c.264")) == md5(os.path.join(tmp, "h.264")) and md5(os.path.join(tmp, "c.yuv")) == md5(os.path.join(tmp, "h.yuv"))
        if not same:
            # where, and how often: the drop-in once more, twenty times, on the same input
            a, b = np.fromfile(os.path.join(tmp, "c.yuv"), np.uint8), np.fromfile(os.path.join(tmp, "h.yuv"), np.uint8)
            fsz = sw * sh * (2 if yuv == 2 else 3) // (1 if yuv == 2 else 2)
            d = np.nonzero(a != b)[0] if len(a) == len(b) else np.array([0])
            f, o = divmod(int(d[0]), fsz) if len(d) else (-1, 0)
            again = 0
            for rep in range(20):
                args = [HIP, "-d", os.path.join(G, "jm_baseline.cfg")]
                for k, v in dict(ov, OutputFile="h2.264", ReconFile="h2.yuv", TraceFile="/dev/null").items():
                    args += ["-p", f"{k}={v}"]
                subprocess.run(args, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
                again += md5(os.path.join(tmp, "h2.264")) != md5(os.path.join(tmp, "c.264"))
            print("FAILED", seed, ov, f"first differing reconstruction byte: frame {f}, offset {o} (luma row {o // sw if o < sw * sh else -1}); {len(d)} bytes differ; {again} of 20 reruns differ too",
                  [l for l in err.splitlines() if "jmhip" in l][-2:], flush=True)
            failed += 1
        piped += "macroblock pipeline:" in err and "not used" not in err
        done += 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    seed += 1
print(f"fuzz_dropin: {done} random RDO-off configurations run in {time.time() - t0:.0f} s, {failed} NOT byte-identical to CPU JM ({piped} through the macroblock pipeline); next seed {seed}")
sys.exit(1 if failed else 0)
