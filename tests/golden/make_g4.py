"""Adds BASELINE.json configs[3] at its own size to tests/golden/md5.json: TEST INFRASTRUCTURE, needs /root/reference (oracle/_ref/lencod.exe).

  G4   synthetic 2160p, 8 slices (SliceMode 1, SliceArgument 4080), FullSearch SR 32, one reference, AdaptiveRounding 0 (933ebd28...)
  G4r  the same with RDOptimization = 0: the macroblock pipeline's configuration (tests/test_lencod_dropin.py runs it through lencod_hip.exe)

  python tests/golden/make_g4.py          # two CPU JM runs of two 2160p pictures each: about three minutes
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

G = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(G))
sys.path.insert(0, G)
import synclip  # noqa: E402

EXE = os.path.join(ROOT, "oracle", "_ref", "lencod.exe")
BASE = dict(InputFile="syn2160p.yuv", SourceWidth=3840, SourceHeight=2160, OutputWidth=3840, OutputHeight=2160, FramesToBeEncoded=2, SearchMode=-1, SearchRange=32,
            NumberReferenceFrames=1, LevelIDC=51, SliceMode=1, SliceArgument=4080, AdaptiveRounding=0)


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def main():
    path = os.path.join(G, "md5.json")
    md5s = json.load(open(path))
    with tempfile.TemporaryDirectory() as tmp:
        synclip.syn2160p(os.path.join(tmp, "syn2160p.yuv"))
        assert md5(os.path.join(tmp, "syn2160p.yuv")) == "72acbcabe33b08e22e5385af5a1fcc73", "the clip generator no longer reproduces SURVEY.md's input"
        for tag, extra in (("G4", {}), ("G4r", dict(RDOptimization=0))):
            o = dict(BASE, **extra)
            args = [EXE, "-d", os.path.join(G, "jm_baseline.cfg")]
            for k, v in dict(o, OutputFile="o.264", ReconFile="o_rec.yuv", TraceFile="/dev/null").items():
                args += ["-p", f"{k}={v}"]
            subprocess.run(args, cwd=tmp, check=True, stdout=subprocess.DEVNULL)
            md5s[tag] = dict(cfg="encoder_baseline.cfg", overrides={k: str(v) for k, v in o.items()}, md5_264=md5(os.path.join(tmp, "o.264")),
                             md5_recon=md5(os.path.join(tmp, "o_rec.yuv")), bytes_264=os.path.getsize(os.path.join(tmp, "o.264")),
                             input="tests/golden/synclip.syn2160p(path, 2): SURVEY.md Appendix A clip at 3840x2160, seed 4321")
            print(tag, md5s[tag]["md5_264"], md5s[tag]["bytes_264"])
    json.dump(md5s, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
