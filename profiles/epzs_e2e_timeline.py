"""Measurement aid: the drop-in encoder on configs[2]'s flags (EPZS, CABAC, 8x8 transform, five references), nine pictures, with the adapter's per-picture timeline.
usage: python profiles/epzs_e2e_timeline.py [pictures] [JMHIP_ADAPTER_FLIGHT]"""
import os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
env = dict(os.environ, JMHIP_ADAPTER_TIMELINE="1")
if len(sys.argv) > 2:
    env["JMHIP_ADAPTER_FLIGHT"] = sys.argv[2]
exe = os.path.join(ROOT, "oracle", "_ref", "lencod_hip.exe")
with tempfile.TemporaryDirectory() as tmp:
    shutil.copyfile(os.path.join(ROOT, "tests", "golden", "q_offset.cfg"), os.path.join(tmp, "q_offset.cfg"))
    bench.write_yuv(os.path.join(tmp, "syn1080p.yuv"), n)
    args = [exe, "-d", os.path.join(ROOT, "tests", "golden", "jm_baseline.cfg")]
    for kv in bench.G3E_FLAGS + (f"FramesToBeEncoded={n}",):
        args += ["-p", kv]
    r = subprocess.run(args, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    out = r.stdout.decode(errors="replace")
    print("\n".join(l for l in out.splitlines() if "(P)" in l or "(IDR)" in l or "( P )" in l or "(I)" in l)[:3000])
    print(r.stderr.decode(errors="replace")[-6000:])
