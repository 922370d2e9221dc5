"""End to end at BASELINE configs[1] with RDOptimization = 0: lencod_hip.exe (macroblock pipeline on the MI355X) and CPU JM (oracle/_ref/lencod.exe)
on the same clip and flags; prints lencod's own per-frame times, the adapter's report and the md5 check.  usage: python profiles/e2e_1080p.py [frames] [cpu]"""
import os, sys, re, tempfile, time
os.environ["JMHIP_ADAPTER_TIMELINE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_lencod_dropin as T
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for which in (["hip", "cpu"] if "cpu" in sys.argv else ["hip"]):
    tmp = tempfile.mkdtemp()
    exe = T.EXE if which == "hip" else os.path.join(ROOT, "oracle", "_ref", "lencod.exe")
    t0 = time.time()
    r, z = T.run_rdo_off_case("g2r", tmp, exe=exe, frames=frames)
    wall = time.time() - t0
    out, err = r.stdout.decode(errors="replace"), r.stderr.decode(errors="replace")
    print(which, "rc", r.returncode, "wall %.2f s" % wall, "frame times", T.frame_times(out))
    print("\n".join(l for l in out.splitlines() if re.match(r"^\s*\d+\(|^ Total encoding time|^ Total ME time", l)))
    m = re.search(r"jmhip adapter: macroblock pipeline.*", err)
    print(m.group(0) if m else err[-300:])
    print("\n".join(l for l in err.splitlines() if "adapter: picture" in l))
    print("md5", T.md5(os.path.join(tmp, "o.264")), str(z["md5_264"]) if frames == 2 else "")
