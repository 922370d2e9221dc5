"""Bit-exactness over more pictures than the golden cases hold: lencod_hip.exe (macroblock pipeline) against CPU JM (oracle/_ref/lencod.exe) run side by side
on the GPU box -- the synthetic 1080p clip (configs[1], RDO off) for N pictures, and the QCIF clip (3 pictures) in several configurations
(1 / 5 references, CABAC, slices, search range 32, QP 20 / 36).  Prints the md5 pairs.  usage: python profiles/long_run_md5.py [n_1080p_pictures]"""
import hashlib, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
G = os.path.join(ROOT, "tests", "golden")
HIP, CPU = os.path.join(ROOT, "oracle", "_ref", "lencod_hip.exe"), os.path.join(ROOT, "oracle", "_ref", "lencod.exe")
md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()
RDO_OFF = {"RDOptimization": "0", "AdaptiveRounding": "0", "SearchMode": "-1"}
def run(name, ov, prep=None):
    out = {}
    for tag, exe in (("hip", HIP), ("cpu", CPU)):
        d = tempfile.mkdtemp()
        for f in ("foreman_part_qcif.yuv", "q_offset.cfg"):
            os.symlink(os.path.join(G, f), os.path.join(d, f))
        if prep: prep(d)
        args = [exe, "-d", os.path.join(G, "jm_baseline.cfg")]
        for k, v in dict(RDO_OFF, **ov, OutputFile="o.264", ReconFile="o_rec.yuv", TraceFile="/dev/null").items():
            args += ["-p", f"{k}={v}"]
        t0 = time.time()
        r = subprocess.run(args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, (name, tag, r.stderr.decode(errors="replace")[-600:])
        out[tag] = (md5(os.path.join(d, "o.264")), md5(os.path.join(d, "o_rec.yuv")), time.time() - t0)
        if tag == "hip":
            assert "encode_one_macroblock_low never ran on the host" in r.stderr.decode(errors="replace"), (name, r.stderr.decode(errors="replace")[-400:])
    ok = out["hip"][:2] == out["cpu"][:2]
    print(f"{name}: {'EQUAL' if ok else 'DIFFERENT'}  .264 {out['hip'][0]} / {out['cpu'][0]}  recon {out['hip'][1]} / {out['cpu'][1]}  wall {out['hip'][2]:.1f} s / {out['cpu'][2]:.1f} s", flush=True)
    return ok
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ok = True
q = {"FramesToBeEncoded": "30"}
ok &= run("QCIF clip (3 pictures), 1 reference, SR 16", dict(q, SearchRange="16", NumberReferenceFrames="1"))
ok &= run("QCIF clip (3 pictures), 5 references, SR 32", dict(q, SearchRange="32"))
ok &= run("QCIF clip (3 pictures), CABAC, 2 references, SR 16", dict(q, SearchRange="16", NumberReferenceFrames="2", SymbolMode="1", ProfileIDC="77"))
ok &= run("QCIF clip (3 pictures), slices of 40 macroblocks, 2 references", dict(q, SearchRange="16", NumberReferenceFrames="2", SliceMode="1", SliceArgument="40"))
ok &= run("QCIF clip (3 pictures), QP 20", dict(q, SearchRange="16", NumberReferenceFrames="1", QPISlice="20", QPPSlice="20"))
ok &= run("QCIF clip (3 pictures), QP 36, SR 32", dict(q, SearchRange="32", NumberReferenceFrames="3", QPISlice="36", QPPSlice="36"))
ok &= run("QCIF clip (3 pictures), intra period 5", dict(q, SearchRange="16", NumberReferenceFrames="2", IntraPeriod="5"))
ov = {"InputFile": "syn1080p.yuv", "SourceWidth": "1920", "SourceHeight": "1080", "OutputWidth": "1920", "OutputHeight": "1080", "FramesToBeEncoded": str(n),
      "SearchRange": "32", "NumberReferenceFrames": "1", "LevelIDC": "51"}
ok &= run(f"configs[1] at 1080p, {n} pictures", ov, lambda d: bench.write_yuv(os.path.join(d, "syn1080p.yuv"), n))
print("ALL EQUAL" if ok else "MISMATCH")
