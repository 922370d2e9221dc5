"""Measurement aid (round 6): fuzz_dropin's seed 700411 -- the drop-in encoder N times on the same input against CPU JM: how often does a run differ, where (frame / plane / macroblock),
and is it the bitstream or only the reconstruction file?   usage: python profiles/r06_race_probe.py <runs> [ENV=VALUE ...]"""
import hashlib, os, shutil, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, G)
import synth_motion
CPU, HIP = os.path.join(ROOT, "oracle", "_ref", "lencod.exe"), os.path.join(ROOT, "oracle", "_ref", "lencod_hip.exe")
ov = {'RDOptimization': 0, 'AdaptiveRounding': 0, 'InputFile': 'clip.yuv', 'SourceWidth': 176, 'SourceHeight': 112, 'OutputWidth': 176, 'OutputHeight': 112, 'FramesToBeEncoded': 4, 'YUVFormat': 2, 'ProfileIDC': 122, 'LevelIDC': 40, 'SymbolMode': 1, 'Transform8x8Mode': 1, 'SearchMode': 0, 'SearchRange': 28, 'NumberReferenceFrames': 4, 'QPISlice': 50, 'QPPSlice': 0, 'ChromaQPOffset': -3, 'DisableSubpelME': 0, 'IntraPeriod': 3, 'NumberBFrames': 0, 'OffsetMatrixPresentFlag': 1, 'CbQPOffset': -4, 'CrQPOffset': 1, 'ReferenceReorder': 1, 'PocMemoryManagement': 1, 'PicOrderCntType': 2, 'IDRPeriod': 3, 'ChangeQPFrame': 2, 'ChangeQPI': 17, 'ChangeQPP': 21}
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 50
env = dict(os.environ, **dict(a.split("=", 1) for a in sys.argv[2:]))
for a in sys.argv[2:]:
    if a.startswith("OV_"):
        k, v = a[3:].split("=", 1); ov[k] = int(v)
sw, sh, nfr = 176, 112, int(ov["FramesToBeEncoded"])
md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()
tmp = tempfile.mkdtemp(prefix="rp_")
np.concatenate(synth_motion.motion_clip(sw, sh, nfr, 700411, yuv422=True)).tofile(os.path.join(tmp, "clip.yuv"))
shutil.copyfile(os.path.join(G, "q_offset.cfg"), os.path.join(tmp, "q_offset.cfg"))
def run(exe, tag):
    args = [exe, "-d", os.path.join(G, "jm_baseline.cfg")]
    for k, v in dict(ov, OutputFile=f"{tag}.264", ReconFile=f"{tag}.yuv", TraceFile="/dev/null").items():
        args += ["-p", f"{k}={v}"]
    return subprocess.run(args, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
assert run(CPU, "c").returncode == 0
a = np.fromfile(os.path.join(tmp, "c.yuv"), np.uint8)
fsz = sw * sh * 2
bad = 0
for k in range(runs):
    r = run(HIP, "h")
    if r.returncode != 0:
        bad += 1; print(f"run {k}: exit {r.returncode}: {r.stderr.decode(errors='replace')[-400:]}", flush=True); continue
    b = np.fromfile(os.path.join(tmp, "h.yuv"), np.uint8)
    same264 = md5(os.path.join(tmp, "c.264")) == md5(os.path.join(tmp, "h.264"))
    if same264 and len(a) == len(b) and (a == b).all():
        continue
    bad += 1
    d = np.nonzero(a != b)[0] if len(a) == len(b) else np.array([], np.int64)
    where = set()
    for o in d:
        f, q = divmod(int(o), fsz)
        if q < sw * sh: where.add((f, "Y", (q % sw) // 16, (q // sw) // 16))
        else:
            q -= sw * sh; pl = "U" if q < sw * sh // 2 else "V"; q %= sw * sh // 2
            where.add((f, pl, (q % (sw // 2)) // 8, (q // (sw // 2)) // 16))
    vals = [(int(a[o]), int(b[o])) for o in d[:6]]
    print(f"run {k}: bitstream {'equal' if same264 else 'DIFFERENT'}; {len(d)} reconstruction bytes differ: (frame, plane, mbx, mby) {sorted(where)}; first (JM, ours): {vals}", flush=True)
print(f"{runs} runs, {bad} different; env {[x for x in sys.argv[2:]]}")
shutil.rmtree(tmp, ignore_errors=True)
