import os, sys, subprocess, hashlib, re, tempfile, shutil
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT,"tests","golden"))
import numpy as np, synth_motion
G=os.path.join(ROOT,"tests","golden")
cpu=os.path.join(ROOT,"oracle","_ref","lencod.exe"); hip=os.path.join(ROOT,"oracle","_ref","lencod_hip.exe")
md5=lambda p: hashlib.md5(open(p,"rb").read()).hexdigest()
for name, ov in (("intra4", {"IntraPeriod":"4"}), ("idr4", {"IDRPeriod":"4"}), ("idr5_intra", {"IDRPeriod":"5","IntraPeriod":"3"}), ("qp_change5", {"ChangeQPFrame":"5","ChangeQPP":"5","ChangeQPI":"5"}), ("epzs_idr6", {"SearchMode":"3","IDRPeriod":"6"})):
    outs=[]
    for exe in (cpu, hip):
        d=tempfile.mkdtemp()
        np.concatenate(synth_motion.motion_clip(176,144,12,123)).tofile(os.path.join(d,"motion.yuv"))
        args=[exe,"-d",os.path.join(G,"jm_baseline.cfg")]
        for k,v in dict({"InputFile":"motion.yuv","RDOptimization":"0","AdaptiveRounding":"0","SearchMode":"-1","SearchRange":"16","NumberReferenceFrames":"2","FramesToBeEncoded":"12","FrameSkip":"0","OutputFile":"o.264","ReconFile":"o_rec.yuv","TraceFile":"/dev/null"}, **ov).items():
            args+=["-p",f"{k}={v}"]
        r=subprocess.run(args,cwd=d,stdout=subprocess.PIPE,stderr=subprocess.PIPE,timeout=600)
        outs.append((r.returncode, md5(os.path.join(d,"o.264")) if r.returncode==0 else None, r.stderr.decode(errors="replace")))
        shutil.rmtree(d)
    m=re.search(r"pictures in flight: (\d+) pictures, (\d+) launched ahead of time \(up to (\d+) in flight\), (\d+) of them served as launched, (\d+) voided", outs[1][2])
    print(name, "equal" if outs[0][1]==outs[1][1] else "DIFFERENT", outs[0][0], outs[1][0], m.groups() if m else outs[1][2][-300:])
