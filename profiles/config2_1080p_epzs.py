"""BASELINE configs[2] at 1080p (encoder_main.cfg values: Main profile, CABAC, EPZS; with ProfileIDC 100 + Transform8x8Mode 1 for the 8x8 path, SURVEY 7 hard part 7):
lencod_hip.exe with EPZS's candidate distortions batched on the device (JM's own walk on the host) beside CPU JM, same clip and flags; md5s compared.
usage: python profiles/config2_1080p_epzs.py [frames]"""
import os, sys, subprocess, tempfile, time, hashlib, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
G = os.path.join(ROOT, "tests", "golden")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ov = dict(InputFile="syn1080p.yuv", SourceWidth=1920, SourceHeight=1080, OutputWidth=1920, OutputHeight=1080, FramesToBeEncoded=frames, SearchMode=3, LevelIDC=51,
          ProfileIDC=100, Transform8x8Mode=1, NumberReferenceFrames=1, OutputFile="o.264", ReconFile="o_rec.yuv", TraceFile="/dev/null")
md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()
res = {}
for which, exe, env in (("cpu", "lencod.exe", {}), ("hip", "lencod_hip.exe", {"JMHIP_ADAPTER_PARTS": "load,interp,interpc,eval,evalp,evalbatch,deblock"})):
    tmp = tempfile.mkdtemp()
    for f in ("q_offset.cfg",):
        if os.path.exists(os.path.join(G, f)):
            import shutil; shutil.copyfile(os.path.join(G, f), os.path.join(tmp, f))
    bench.write_yuv(os.path.join(tmp, "syn1080p.yuv"), frames)
    args = [os.path.join(ROOT, "oracle", "_ref", exe), "-d", os.path.join(G, "jm_main.cfg")]
    for k, v in ov.items():
        args += ["-p", f"{k}={v}"]
    t0 = time.time()
    r = subprocess.run(args, cwd=tmp, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out, err = r.stdout.decode(errors="replace"), r.stderr.decode(errors="replace")
    res[which] = md5(os.path.join(tmp, "o.264")) if r.returncode == 0 else None
    print(which, "rc", r.returncode, "wall %.1f s" % (time.time() - t0), "md5", res[which])
    print("\n".join(l for l in out.splitlines() if re.match(r"^\s*\d+\(", l) or "Total ME time" in l or "Total encoding time" in l))
    for l in err.splitlines():
        if "candidate distortions:" in l or "on the MI355X:" in l:
            print(l[:400])
print("bitstreams", "EQUAL" if res["cpu"] and res["cpu"] == res["hip"] else "DIFFERENT")
