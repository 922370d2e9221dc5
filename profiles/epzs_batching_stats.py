import sys, os, tempfile, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_lencod_dropin as T
for tag in ("G3a", "G3b"):
    for env in ({}, {"JMHIP_ADAPTER_PARTS": "load,interp,interpc,fs,subpel,ffs,eval,evalp,tq4,tq8,tq16,tqc,mcl,mcc,ip4,ip8,i16,ic,deblock"}):
        tmp = tempfile.mkdtemp(); t0 = time.time()
        r, o, rec = T.run_lencod(T.EXE, tag, tmp, env)
        err = r.stderr.decode()
        line = [l for l in err.splitlines() if "candidate distortions:" in l]
        print(tag, "batched" if not env else "one by one", "wall %.1f s" % (time.time() - t0), T.md5(o) == T.MD5[tag]["md5_264"], line[0][15:] if line else "")
