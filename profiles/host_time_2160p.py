"""Where the host's wall time goes at 2160p (configs[3], RDO off, 8 slices): user / system time and page faults of lencod_hip.exe (measurement aid).
usage: python profiles/host_time_2160p.py [frames] [extra env assignments ...]"""
import os, resource, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synclip
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
env = dict(os.environ, JMHIP_ADAPTER_TIMELINE="1", **dict(a.split("=", 1) for a in sys.argv[2:]))
with tempfile.TemporaryDirectory() as t:
    synclip.syn2160p(os.path.join(t, "syn2160p.yuv"), frames)
    args = [os.path.join(ROOT, "oracle", "_ref", "lencod_hip.exe"), "-d", os.path.join(ROOT, "tests", "golden", "jm_baseline.cfg")]
    for kv in ("InputFile=syn2160p.yuv SourceWidth=3840 SourceHeight=2160 OutputWidth=3840 OutputHeight=2160 SearchMode=-1 SearchRange=32 NumberReferenceFrames=1 LevelIDC=51 "
               "RDOptimization=0 AdaptiveRounding=0 SliceMode=1 SliceArgument=4080 OutputFile=o.264 ReconFile=o_rec.yuv TraceFile=/dev/null FramesToBeEncoded=%d" % frames).split():
        args += ["-p", kv]
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.time()
    p = subprocess.run(args, cwd=t, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    wall = time.time() - t0
    import hashlib
    print("md5 of the .264:", hashlib.md5(open(os.path.join(t, "o.264"), "rb").read()).hexdigest())
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    for line in p.stdout.decode(errors="replace").splitlines():
        if line.strip()[:5].isdigit() and "(" in line:
            print(line)
    for line in p.stderr.decode(errors="replace").splitlines():
        if "picture" in line and "slices" in line:
            print(line[:520])
    print(f"wall {wall:.2f} s, user {r1.ru_utime - r0.ru_utime:.2f} s, system {r1.ru_stime - r0.ru_stime:.2f} s, minor faults {r1.ru_minflt - r0.ru_minflt}, major {r1.ru_majflt - r0.ru_majflt}, "
          f"voluntary switches {r1.ru_nvcsw - r0.ru_nvcsw}, max rss {r1.ru_maxrss // 1024} MB")
