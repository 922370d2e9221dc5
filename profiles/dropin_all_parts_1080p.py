"""BASELINE configs[1] (G2: RDO on, 1080p, two pictures) through lencod_hip.exe with EVERY adapter part on -- the per-call path's transform/quant,
prediction and intra kernels produce the bitstream at the headline size.  Minutes (millions of synchronous single-block calls);
not part of the default -m gpu set.  Prints the md5 check and the adapter's report."""
import os, sys, subprocess, tempfile, time, hashlib, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
G = os.path.join(ROOT, "tests", "golden")
e = json.load(open(os.path.join(G, "md5.json")))["G2"]
tmp = tempfile.mkdtemp()
bench.write_yuv(os.path.join(tmp, "syn1080p.yuv"), 2)
args = [os.path.join(ROOT, "oracle", "_ref", "lencod_hip.exe"), "-d", os.path.join(G, "jm_baseline.cfg")]
for k, v in dict(e["overrides"], OutputFile="o.264", ReconFile="o_rec.yuv", TraceFile="/dev/null").items():
    args += ["-p", f"{k}={v}"]
t0 = time.time()
r = subprocess.run(args, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()
print("rc", r.returncode, "wall %.0f s" % (time.time() - t0))
print(".264 md5", md5(os.path.join(tmp, "o.264")), "expected", e["md5_264"], "EQUAL" if md5(os.path.join(tmp, "o.264")) == e["md5_264"] else "DIFFERENT")
print("recon md5", md5(os.path.join(tmp, "o_rec.yuv")), "expected", e["md5_recon"], "EQUAL" if md5(os.path.join(tmp, "o_rec.yuv")) == e["md5_recon"] else "DIFFERENT")
print("\n".join(l for l in r.stdout.decode(errors="replace").splitlines() if l.strip().startswith("0000")))
print(r.stderr.decode(errors="replace")[-2500:])
