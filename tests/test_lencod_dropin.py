"""End-to-end drop-in parity: JM's own encoder with its hot path served by libjmhip must write the SAME bitstream.

oracle/_ref/lencod_hip.exe = the unmodified reference lencod objects + jm_amd/adapter/jm_adapter.c (host C glue, ld --wrap)
+ jm_amd/libjmhip.so (oracle/Makefile.ref, target `hip`; built in the build container, travels to the GPU box).  It is run
on the reference's own sample clips with the reference's own configurations (tests/golden/jm_*.cfg, values only) and the
Annex-B output / reconstruction are compared by md5 with CPU JM's (tests/golden/md5.json, produced by the real lencod).
The adapter's exit report must show that the device really served the calls (no silent pass-through)."""
import hashlib
import json
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "oracle", "_ref", "lencod_hip.exe")
CFG = {"encoder_baseline.cfg": "jm_baseline.cfg", "encoder_main.cfg": "jm_main.cfg", "encoder_yuv422.cfg": "jm_yuv422.cfg"}
MD5 = json.load(open(os.path.join(G, "md5.json")))


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def run_lencod(exe, tag, tmp, env_extra=None):
    e = MD5[tag]
    for f in ("foreman_part_qcif.yuv", "foreman_part_qcif_422.yuv", "q_offset.cfg"):
        shutil.copyfile(os.path.join(G, f), os.path.join(tmp, f))
    args = [exe, "-d", os.path.join(G, CFG[e["cfg"]])]
    for k, v in dict(e["overrides"], OutputFile="o.264", ReconFile="o_rec.yuv", TraceFile="/dev/null").items():
        args += ["-p", f"{k}={v}"]
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run(args, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    return r, os.path.join(tmp, "o.264"), os.path.join(tmp, "o_rec.yuv")


def counters(stderr):
    m = re.search(r"jmhip adapter: on the MI355X: (\d+) getSubImagesLuma, (\d+) full_search_motion_estimation, (\d+) sub_pel_motion_estimation, "
                  r"(\d+) setup_fast_full_search, (\d+) DeblockFrame \((\d+) current pictures uploaded\); passed to JM's own code: (\d+) calls; "
                  r"transform/quant blocks on the MI355X: (\d+) 4x4, (\d+) 8x8, (\d+) chroma planes; prediction blocks on the MI355X: (\d+) luma, (\d+) chroma; "
                  r"Intra16x16 macroblocks on the MI355X: (\d+); candidate distortions \(computeSAD / computeSATD\) on the MI355X: (\d+); "
                  r"intra predictions on the MI355X: (\d+) 4x4 blocks, (\d+) Intra16x16 mode searches; getSubImagesChroma on the MI355X: (\d+); "
                  r"weighted / bi-predictive candidate distortions on the MI355X: (\d+); source pictures padded on the MI355X: (\d+); chroma intra predictions on the MI355X: (\d+) macroblocks; Intra8x8 predictions on the MI355X: (\d+) blocks", stderr)
    assert m, stderr[-2000:]
    return dict(zip(("interp", "fs", "subpel", "ffs", "deblock", "cur", "passed", "tq4", "tq8", "tqc", "mcl", "mcc", "tq16", "eval", "ip4", "i16", "interpc", "evalp", "load", "ic", "ip8"), (int(x) for x in m.groups())))


# tag -> which adapter counters must be non-zero (what that configuration exercises on the device)
CASES = [
    ("G1", ("load", "ic", "interp", "interpc", "fs", "subpel", "tq4", "tqc", "tq16", "mcl", "mcc", "ip4", "i16", "deblock")),       # BASELINE configs[0]: FullSearch SR=16, 5 refs, RDO, CAVLC
    ("G0", ("interp", "ffs", "subpel", "tq4", "tqc", "mcl", "mcc", "deblock")),      # encoder_baseline.cfg as shipped: FastFullSearch SR=32
    ("G1_1ref_2frames", ("interp", "fs", "subpel", "tq4", "tqc", "deblock")),
    ("G4q", ("interp", "fs", "subpel", "tq4", "tqc", "deblock")),      # configs[3] shape: 3 slices, AdaptiveRounding off (quant_4x4_normal)
    ("G3a", ("ic", "interp", "eval", "evalp", "tq4", "tqc", "mcl", "mcc", "deblock")),              # Main, CABAC, B frame; EPZS: JM's own walk (me_epzs*.c), every candidate's distortion on the device
    ("G3b", ("ip8", "interp", "eval", "evalp", "tq4", "tq8", "tqc", "mcl", "mcc", "deblock")),               # + 8x8 transform (High, CABAC): residual_transform_quant_luma_8x8
    # explicit weighted prediction (P and B): JM's searches call compute*WP / computeBiPred*2 with the weights it estimated (33 / -5, 33 + 32 / -2)
    ("G3w", ("interp", "evalp", "tq4", "tqc", "mcl", "mcc", "deblock")),
    ("G3wb", ("interp", "evalp", "tq4", "tq8", "tqc", "mcl", "mcc", "deblock")),           # + 8x8 transform: computeBiPredSATD2's 8x8 path (me_distortion.c:1113-1175)
    ("G5", ("load", "ic", "ip8", "interp", "interpc", "ffs", "subpel", "tq4", "tq8", "tqc", "tq16", "mcl", "mcc", "ip4", "i16", "deblock")),  # configs[4]: High 4:2:2, FFS, 5 refs, 8x8 transform
]


@pytest.mark.gpu
@pytest.mark.parametrize("tag,must_run", CASES)
def test_lencod_with_libjmhip_writes_jm_bitstream(tmp_path, tag, must_run):
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing: run __graft_entry__.build() where /root/reference exists")
    r, out264, rec = run_lencod(EXE, tag, str(tmp_path))
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0, (r.stdout.decode(errors="replace")[-1500:], err[-1500:])
    c = counters(err)
    print(f"{tag}: calls on the device / passed to JM: {c}")
    for k in must_run:
        assert c[k] > 0, (tag, k, c)
    assert md5(out264) == MD5[tag]["md5_264"], (tag, "bitstream differs from CPU JM", c)
    assert md5(rec) == MD5[tag]["md5_recon"], (tag, "reconstruction differs from CPU JM", c)


@pytest.mark.gpu
def test_lencod_configs1_full_size_1080p(tmp_path):
    """BASELINE.json configs[1] end to end at full size: synthetic 1080p, Baseline IPPP, FullSearch SR=32, one reference, I + P.
    334,560 BlockMotionSearch calls of the P frame go to the device one by one (integer search + sub-pel refinement), plus the
    sub-pel planes and the deblocking of both frames; transform/quant stays with JM here (3.3 M synchronous single-block calls
    would only measure PCIe latency -- the QCIF cases above cover it).  The bitstream must equal CPU JM's (md5 of SURVEY.md 8c, G2)."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    import sys
    sys.path.insert(0, ROOT)
    import bench
    tmp = str(tmp_path)
    bench.write_yuv(os.path.join(tmp, "syn1080p.yuv"), 2)
    e = MD5["G2"]
    args = [EXE, "-d", os.path.join(G, "jm_baseline.cfg")]
    for k, v in dict(e["overrides"], OutputFile="o.264", ReconFile="o_rec.yuv", TraceFile="/dev/null").items():
        args += ["-p", f"{k}={v}"]
    env = dict(os.environ, JMHIP_ADAPTER_PARTS="load,interp,fs,subpel,deblock")
    r = subprocess.run(args, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0, (r.stdout.decode(errors="replace")[-1500:], err[-1500:])
    c = counters(err)
    assert c["fs"] > 300000 and c["subpel"] > 300000 and c["interp"] == 2 and c["deblock"] == 2 and c["load"] >= 2, c    # load: 1080 -> 1088 rows on the device
    assert md5(os.path.join(tmp, "o.264")) == e["md5_264"], ("bitstream differs from CPU JM", c)
    assert md5(os.path.join(tmp, "o_rec.yuv")) == e["md5_recon"], ("reconstruction differs from CPU JM", c)
    m = re.search(r"^\s*0*1\(\s*P\s*\)\s+\d+\s+\d+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+(\d+)\s+(\d+)", r.stdout.decode(errors="replace"), re.M)
    if m:
        print(f"P frame with the hot path on the device (per-call offload): {m.group(1)} ms total, {m.group(2)} ms ME")


def run_rdo_off_case(tag, tmp, exe=EXE, env_extra=None, frames=None):
    """lencod with the overrides of tests/golden/mb_low_<tag>.npz (RDOptimization = 0: the macroblock pipeline's configurations)"""
    import numpy as np
    z = np.load(os.path.join(G, f"mb_low_{tag}.npz"))
    ov = dict(s.split("=") for s in z["overrides"])
    if frames:
        ov["FramesToBeEncoded"] = str(frames)
    for f in ("foreman_part_qcif.yuv", "foreman_part_qcif_422.yuv", "q_offset.cfg"):
        shutil.copyfile(os.path.join(G, f), os.path.join(tmp, f))
    clip = str(z["clip"]) if "clip" in z.files else ""
    cfg = str(z["cfg"]) if "cfg" in z.files else "jm_baseline.cfg"
    if clip == "syn422":                                   # BASELINE configs[4]'s input at its own size (tests/golden/synclip.py)
        import sys
        sys.path.insert(0, G)
        import synclip
        synclip.syn1080p422(os.path.join(tmp, "syn1080p422.yuv"), int(ov["FramesToBeEncoded"]))
    elif tag == "g2r" or clip == "True":
        import sys
        sys.path.insert(0, ROOT)
        import bench
        bench.write_yuv(os.path.join(tmp, "syn1080p.yuv"), int(ov["FramesToBeEncoded"]))
    elif clip.startswith("motion"):                        # tests/golden/synth_motion.py at the case's size (the overrides name the file)
        import sys
        sys.path.insert(0, G)
        import synth_motion
        sw, sh = int(z["size"][0]), int(z["size"][1])
        data = np.concatenate(synth_motion.motion_clip(sw, sh, int(ov["FramesToBeEncoded"]), int(clip.split(":")[1]), yuv422=clip.startswith("motion422")))
        assert frames or hashlib.md5(data.tobytes()).hexdigest() == str(z["clip_md5"])
        data.tofile(os.path.join(tmp, "motion.yuv"))
    args = [exe, "-d", os.path.join(G, cfg)]
    for k, v in dict(ov, OutputFile="o.264", ReconFile="o_rec.yuv", TraceFile="/dev/null").items():
        args += ["-p", f"{k}={v}"]
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run(args, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    return r, z


def pipeline_report(stderr):
    m = re.search(r"macroblock pipeline: (\d+) slices, (\d+) macroblocks encoded on the MI355X .*device calls ([\d.]+) s, waiting for records ([\d.]+) s, unpacking them ([\d.]+) s", stderr)
    return None if not m else dict(slices=int(m.group(1)), mbs=int(m.group(2)), t_dev=float(m.group(3)), t_wait=float(m.group(4)), t_fill=float(m.group(5)))


def frame_times(stdout):
    """{frame type: [total ms per frame]} from lencod's per-frame lines"""
    out = {}
    for m in re.finditer(r"^\s*\d+\(\s*(IDR|I|P)\s*\)\s+\d+\s+\d+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+(\d+)\s+(\d+)", stdout, re.M):
        out.setdefault("I" if m.group(1) != "P" else "P", []).append(int(m.group(2)))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("tag,nmb_total", [("q1r", 297), ("q5r", 297), ("q4r", 297), ("q4s", 297), ("q1c", 297), ("q0c", 297), ("q0r", 297),
                                           ("q1e", 297), ("m5e", 780), ("m2c", 396), ("m3p", 396), ("m2t", 384),      # EPZS (SearchMode 3)
                                           ("q1h", 297), ("q2hc", 297), ("m3h", 650), ("m2he", 396), ("m1hq", 297),   # High profile: the 8x8 transform, Intra8x8
                                           ("q5f", 297), ("m5f", 780), ("m3fh", 396),                                  # fast full search (SearchMode 0)
                                           ("q5y", 297), ("q2yv", 297), ("m3y", 650), ("m2yq", 396),                   # 4:2:2 (encoder_yuv422.cfg: BASELINE configs[4] with RDO off, P pictures only)
                                           ("m2pd", 396), ("m3pe", 300), ("q1pd", 297),                               # partitions switched off (PSliceSearch*)
                                           ("m2cq", 396), ("m2yc", 396),                                               # CbQPOffset != CrQPOffset
                                           ("m3fl", 175), ("m3fm", 175), ("m2sl", 192), ("m2el", 192),                               # level 1.1
                                           ("m2es", 396), ("m5es", 192),                                               # EPZS at SearchRange 2
                                           # B pictures (NumberBFrames 1; encoder_main.cfg / encoder_yuv422.cfg with RDO off: q1b, q5yb), without (*b0) and with the bi-predictive search
                                           ("q1b0", 297), ("q1b", 297), ("m3b0", 910), ("m3b", 910), ("m2b4", 495), ("q5yb", 297),
                                           # DirectModeType 0 (temporal direct), one / two B pictures between the references
                                           ("q1bt", 297), ("m3bt", 910)])
def test_lencod_macroblock_pipeline_writes_jm_bitstream(tmp_path, tag, nmb_total):
    """RDOptimization = 0: encode_one_macroblock_low never runs on the host -- every macroblock of every slice is encoded by jmhip_encode_slice,
    JM's own write_macroblock codes the records, DeblockFrame and the sub-pel planes stay on the device.  The Annex-B output and the reconstruction
    must equal CPU JM's (md5s inside tests/golden/mb_low_*.npz, from the unmodified encoder).  One / five references, three slices, slices that
    start mid-row with DFDisableIdc = 2."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing: run __graft_entry__.build() where /root/reference exists")
    r, z = run_rdo_off_case(tag, str(tmp_path))
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0, (r.stdout.decode(errors="replace")[-1500:], err[-1500:])
    rep = pipeline_report(err)
    c = counters(err)
    assert rep and rep["mbs"] == nmb_total, (rep, err[-1500:])
    assert c["passed"] == 0 and c["fs"] == 0 and c["subpel"] == 0 and c["tq4"] == 0 and c["eval"] == 0, c        # nothing per block (no jmhip_me_eval either), nothing on the host
    assert md5(os.path.join(str(tmp_path), "o.264")) == str(z["md5_264"]), (tag, "bitstream differs from CPU JM", rep)
    assert md5(os.path.join(str(tmp_path), "o_rec.yuv")) == str(z["md5_recon"]), (tag, "reconstruction differs from CPU JM", rep)


def wild_clip(path, W, H, nfr, seed):
    """CIF-sized synthetic clip made to leave the comfortable cases: global motion of up to 24 samples per picture in changing directions (long vectors,
    search centres clamped to the range, windows over the picture edge), an object moving against it, a brightness ramp, a scene cut, coarse noise."""
    import numpy as np
    rng = np.random.default_rng(seed)
    def texture(s):
        r = np.random.default_rng(s)
        base = np.kron(r.integers(0, 256, (H // 8 + 16, W // 8 + 16)).astype(np.float64), np.ones((8, 8)))
        k = 3
        pad = np.pad(base, k, mode="edge")
        return sum(pad[i:i + base.shape[0], j:j + base.shape[1]] for i in range(2 * k + 1) for j in range(2 * k + 1)) / (2 * k + 1) ** 2
    tex, ox, oy = texture(seed), 64, 64
    with open(path, "wb") as f:
        for n in range(nfr):
            if n == nfr // 2:
                tex = texture(seed + 1)                                      # scene cut
            ox = int(np.clip(ox + rng.integers(-24, 25), 0, 127)); oy = int(np.clip(oy + rng.integers(-24, 25), 0, 127))
            y = tex[oy:oy + H, ox:ox + W].copy()
            bx, by = (37 * n) % (W - 48), (23 * n) % (H - 48)
            y[by:by + 48, bx:bx + 48] = 255 - y[by:by + 48, bx:bx + 48]       # an object moving against the background
            y = y * (0.7 + 0.05 * n) + rng.normal(0, 3 + (n % 3) * 3, (H, W))  # brightness ramp, noise
            y = np.clip(np.rint(y), 0, 255).astype(np.uint8)
            c = y[::2, ::2].astype(np.float64)
            f.write(y.tobytes()); f.write(np.clip(np.rint(128 + 0.3 * (c - 128)), 0, 255).astype(np.uint8).tobytes())
            f.write(np.clip(np.rint(128 - 0.2 * (c - 128)), 0, 255).astype(np.uint8).tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("ov", [dict(SearchRange="32", NumberReferenceFrames="2"), dict(SearchRange="16", NumberReferenceFrames="1", QPISlice="36", QPPSlice="38"),
                                dict(SearchRange="8", NumberReferenceFrames="3", SliceMode="1", SliceArgument="100", SymbolMode="1", ProfileIDC="77", QPPSlice="22"),
                                dict(SearchMode="3", SearchRange="32", NumberReferenceFrames="4", SymbolMode="1", ProfileIDC="77"),
                                dict(SearchMode="3", SearchRange="16", NumberReferenceFrames="2", SliceMode="1", SliceArgument="150", QPISlice="34", QPPSlice="36", EPZSPattern="5", EPZSDualRefinement="2")],
                         ids=["sr32_2ref", "sr16_qp38", "sr8_3ref_slices_cabac", "epzs_4ref_cabac", "epzs_2ref_slices_pmvfast"])
def test_lencod_macroblock_pipeline_side_by_side_on_a_wild_clip(tmp_path, ov):
    """No golden file: CPU JM (oracle/_ref/lencod.exe) and the drop-in encoder run on the same generated CIF clip here on the GPU box (10 pictures:
    large changing global motion, a moving object, brightness change, a scene cut, noise) and must write the same bitstream and reconstruction."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    ref_exe = os.path.join(os.path.dirname(EXE), "lencod.exe")
    W, H, nfr = 352, 288, 10
    sums = {}
    for name, exe in (("hip", EXE), ("cpu", ref_exe)):
        d = os.path.join(str(tmp_path), name)
        os.makedirs(d)
        shutil.copyfile(os.path.join(G, "q_offset.cfg"), os.path.join(d, "q_offset.cfg"))
        wild_clip(os.path.join(d, "wild.yuv"), W, H, nfr, 77)
        args = [exe, "-d", os.path.join(G, "jm_baseline.cfg")]
        full = dict(RDOptimization="0", AdaptiveRounding="0", SearchMode="-1", InputFile="wild.yuv", SourceWidth=str(W), SourceHeight=str(H), OutputWidth=str(W),
                    OutputHeight=str(H), FramesToBeEncoded=str(nfr), LevelIDC="40", OutputFile="o.264", ReconFile="o_rec.yuv", TraceFile="/dev/null")
        full.update(ov)
        for k, v in full.items():
            args += ["-p", f"{k}={v}"]
        r = subprocess.run(args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        err = r.stderr.decode(errors="replace")
        assert r.returncode == 0, (name, r.stdout.decode(errors="replace")[-800:], err[-800:])
        if name == "hip":
            rep = pipeline_report(err)
            assert rep and rep["mbs"] == nfr * (W // 16) * (H // 16), (rep, err[-800:])
        sums[name] = (md5(os.path.join(d, "o.264")), md5(os.path.join(d, "o_rec.yuv")))
    assert sums["hip"] == sums["cpu"], sums


@pytest.mark.gpu
def test_lencod_macroblock_pipeline_leaves_a_real_trace_file_alone(tmp_path):
    """The adapter drops JM's syntax-element trace inside the process only when the configuration sends it to /dev/null (part nulltrace).  With a
    real TraceFile the drop-in encoder writes the same trace, byte for byte, as CPU JM (and the same bitstream)."""
    import numpy as np
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    ref_exe = os.path.join(os.path.dirname(EXE), "lencod.exe")
    z = np.load(os.path.join(G, "mb_low_q1r.npz"))
    ov = dict(s.split("=") for s in z["overrides"])
    ov["FramesToBeEncoded"] = "2"
    sums = {}
    for name, exe in (("hip", EXE), ("cpu", ref_exe)):
        d = os.path.join(str(tmp_path), name)
        os.makedirs(d)
        for f in ("foreman_part_qcif.yuv", "q_offset.cfg"):
            shutil.copyfile(os.path.join(G, f), os.path.join(d, f))
        args = [exe, "-d", os.path.join(G, "jm_baseline.cfg")]
        for k, v in dict(ov, OutputFile="o.264", ReconFile="o_rec.yuv", TraceFile="trace.txt").items():
            args += ["-p", f"{k}={v}"]
        r = subprocess.run(args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, (name, r.stderr.decode(errors="replace")[-800:])
        if name == "hip":
            rep = pipeline_report(r.stderr.decode(errors="replace"))
            assert rep and rep["mbs"] == 2 * 99, rep
        assert os.path.getsize(os.path.join(d, "trace.txt")) > 100000
        sums[name] = (md5(os.path.join(d, "trace.txt")), md5(os.path.join(d, "o.264")))
    assert sums["hip"] == sums["cpu"], sums


@pytest.mark.gpu
def test_lencod_macroblock_pipeline_configs1_full_size_1080p(tmp_path):
    """BASELINE.json configs[1] with RDOptimization = 0 end to end: .264 md5 04ce4cdee722defe8c3c7c0b249eda7e, zero
    per-block calls, and the P picture's wall time as lencod itself prints it."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    r, z = run_rdo_off_case("g2r", str(tmp_path))
    err, out = r.stderr.decode(errors="replace"), r.stdout.decode(errors="replace")
    assert r.returncode == 0, (out[-1500:], err[-1500:])
    rep = pipeline_report(err)
    c = counters(err)
    assert rep and rep["mbs"] == 2 * 8160 and rep["slices"] == 2, (rep, err[-1500:])
    assert c["passed"] == 0 and c["fs"] == 0 and c["subpel"] == 0, c
    assert str(z["md5_264"]) == "04ce4cdee722defe8c3c7c0b249eda7e"
    assert md5(os.path.join(str(tmp_path), "o.264")) == str(z["md5_264"]), ("bitstream differs from CPU JM", rep)
    assert md5(os.path.join(str(tmp_path), "o_rec.yuv")) == str(z["md5_recon"]), ("reconstruction differs from CPU JM", rep)
    print(f"configs[1], RDO off, macroblock pipeline: frame times (ms) {frame_times(out)}, adapter {rep}")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["g3e", "g3h"])
def test_lencod_macroblock_pipeline_configs2_epzs_1080p(tmp_path, tag):
    """BASELINE.json configs[2] end to end at full size with RDOptimization = 0 and P pictures only: g3e = Main profile (CABAC, EPZS with the shipped switches,
    five references configured, I + 2 P pictures), g3h = the same in High profile with the 8x8 transform on (configs[2] as stated).  Every EPZS search runs
    inside the device's macroblock pipeline -- zero jmhip_me_eval calls, zero calls passed to JM -- and the bitstream and the reconstruction equal CPU JM's."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    r, z = run_rdo_off_case(tag, str(tmp_path))
    err, out = r.stderr.decode(errors="replace"), r.stdout.decode(errors="replace")
    assert r.returncode == 0, (out[-1500:], err[-1500:])
    rep = pipeline_report(err)
    c = counters(err)
    assert rep and rep["mbs"] == 3 * 8160 and rep["slices"] == 3, (rep, err[-1500:])
    assert c["passed"] == 0 and c["eval"] == 0 and c["evalp"] == 0 and c["fs"] == 0 and c["subpel"] == 0, c
    assert md5(os.path.join(str(tmp_path), "o.264")) == str(z["md5_264"]), ("bitstream differs from CPU JM", rep)
    assert md5(os.path.join(str(tmp_path), "o_rec.yuv")) == str(z["md5_recon"]), ("reconstruction differs from CPU JM", rep)
    print(f"configs[2] ({tag}: EPZS, CABAC), RDO off, macroblock pipeline: frame times (ms) {frame_times(out)}, adapter {rep}")


@pytest.mark.gpu
def test_lencod_macroblock_pipeline_fast_full_search_1080p(tmp_path):
    """encoder_baseline.cfg's search as it ships -- SearchMode 0 (fast full search), SearchRange 32, five references configured -- at 1920x1080 with
    RDOptimization = 0 (g5f): I + 3 P pictures searching 1, 2 and 3 references, 32 640 macroblocks through the pipeline, bitstream and reconstruction
    equal to CPU JM's."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    r, z = run_rdo_off_case("g5f", str(tmp_path))
    err, out = r.stderr.decode(errors="replace"), r.stdout.decode(errors="replace")
    assert r.returncode == 0, (out[-1500:], err[-1500:])
    rep = pipeline_report(err)
    c = counters(err)
    assert rep and rep["mbs"] == 4 * 8160 and rep["slices"] == 4, (rep, err[-1500:])
    assert c["passed"] == 0 and c["eval"] == 0 and c["fs"] == 0 and c["subpel"] == 0, c
    assert md5(os.path.join(str(tmp_path), "o.264")) == str(z["md5_264"]), ("bitstream differs from CPU JM", rep)
    assert md5(os.path.join(str(tmp_path), "o_rec.yuv")) == str(z["md5_recon"]), ("reconstruction differs from CPU JM", rep)
    print(f"1080p fast full search, up to three references, RDO off, macroblock pipeline: frame times (ms) {frame_times(out)}, adapter {rep}")


@pytest.mark.gpu
def test_lencod_macroblock_pipeline_configs4_yuv422_1080p(tmp_path):
    """BASELINE.json configs[4] at its own size with RDOptimization = 0 and P pictures only (g4y): encoder_yuv422.cfg -- High 4:2:2 profile, CABAC, 8x8 transform on, fast
    full search SR 32, five references configured, q_offset.cfg's quantiser offsets -- on 1920x1080 4:2:2 synthetic input, I + 2 P pictures: every macroblock through
    the device's pipeline, bitstream and reconstruction equal to CPU JM's."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    r, z = run_rdo_off_case("g4y", str(tmp_path))
    err, out = r.stderr.decode(errors="replace"), r.stdout.decode(errors="replace")
    assert r.returncode == 0, (out[-1500:], err[-1500:])
    rep = pipeline_report(err)
    c = counters(err)
    assert rep and rep["mbs"] == 3 * 8160 and rep["slices"] == 3, (rep, err[-1500:])
    assert c["passed"] == 0 and c["eval"] == 0 and c["fs"] == 0 and c["subpel"] == 0, c
    assert md5(os.path.join(str(tmp_path), "o.264")) == str(z["md5_264"]), ("bitstream differs from CPU JM", rep)
    assert md5(os.path.join(str(tmp_path), "o_rec.yuv")) == str(z["md5_recon"]), ("reconstruction differs from CPU JM", rep)
    print(f"configs[4] (1080p 4:2:2, CABAC, 8x8 transform, fast full search, RDO off), macroblock pipeline: frame times (ms) {frame_times(out)}, adapter {rep}")


def run_2160p(tag, tmp, env_extra=None):
    """lencod_hip.exe on BASELINE configs[3] at its own size: synthetic 2160p, 8 slices of 4080 macroblocks, two pictures (tests/golden/make_g4.py)"""
    import sys
    sys.path.insert(0, G)
    import synclip
    e = MD5[tag]
    synclip.syn2160p(os.path.join(tmp, "syn2160p.yuv"))
    args = [EXE, "-d", os.path.join(G, "jm_baseline.cfg")]
    for k, v in dict(e["overrides"], OutputFile="o.264", ReconFile="o_rec.yuv", TraceFile="/dev/null").items():
        args += ["-p", f"{k}={v}"]
    env = dict(os.environ)
    env.update(env_extra or {})
    return subprocess.run(args, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=2400), e


@pytest.mark.gpu
def test_lencod_macroblock_pipeline_configs3_2160p_8_slices(tmp_path):
    """BASELINE.json configs[3] at 3840x2160 with RDOptimization = 0 (G4r): 8 slices per picture, 64 800 macroblocks through jmhip_encode_slice, the
    bitstream and the reconstruction equal CPU JM's."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    r, e = run_2160p("G4r", str(tmp_path))
    err, out = r.stderr.decode(errors="replace"), r.stdout.decode(errors="replace")
    assert r.returncode == 0, (out[-1500:], err[-1500:])
    rep = pipeline_report(err)
    c = counters(err)
    assert rep and rep["mbs"] == 2 * 32400 and rep["slices"] == 16, (rep, err[-1500:])
    assert c["passed"] == 0 and c["fs"] == 0 and c["subpel"] == 0, c
    assert md5(os.path.join(str(tmp_path), "o.264")) == e["md5_264"], ("bitstream differs from CPU JM", rep)
    assert md5(os.path.join(str(tmp_path), "o_rec.yuv")) == e["md5_recon"], ("reconstruction differs from CPU JM", rep)
    print(f"configs[3] at 2160p, RDO off, macroblock pipeline: frame times (ms) {frame_times(out)}, adapter {rep}")


@pytest.mark.gpu
@pytest.mark.parametrize("tag,devices,nmb_total", [("q4r", "0,0,0", 297), ("m2ed", "0,0,0", 495), ("G4r", "0,0", 64800), ("G4r", "0,0,0,0,0,0,0,0", 64800)])
def test_lencod_slices_dealt_to_several_contexts(tmp_path, tag, devices, nmb_total):
    """JMHIP_DEVICES: the slices of a picture dealt to several contexts of one process (here all on this box's one device; between devices the same calls are peer copies),
    the bands exchanged with jmhip_allgather_bands before DeblockFrame, every context keeping the whole reference: QCIF in three slices on three contexts, BASELINE configs[3]
    (2160p, 8 slices, RDO off) on two and on eight.  Bitstream and reconstruction equal CPU JM's."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    if tag == "G4r":
        r, e = run_2160p(tag, str(tmp_path), {"JMHIP_DEVICES": devices})
        want = (e["md5_264"], e["md5_recon"])
    else:
        r, z = run_rdo_off_case(tag, str(tmp_path), env_extra={"JMHIP_DEVICES": devices})
        want = (str(z["md5_264"]), str(z["md5_recon"]))
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0, (r.stdout.decode(errors="replace")[-1500:], err[-1500:])
    assert f"{len(devices.split(','))} contexts (JMHIP_DEVICES)" in err, err[-1500:]
    rep = pipeline_report(err)
    assert rep and rep["mbs"] == nmb_total, (rep, err[-1500:])
    assert md5(os.path.join(str(tmp_path), "o.264")) == want[0], (tag, devices, "bitstream differs from CPU JM")
    assert md5(os.path.join(str(tmp_path), "o_rec.yuv")) == want[1], (tag, devices, "reconstruction differs from CPU JM")


@pytest.mark.gpu
def test_lencod_pictures_in_flight_are_verified_not_trusted(tmp_path):
    """The adapter launches pictures ahead of time and uses such a launch only if JM's own parameters and source planes equal what it was given.
    (a) the plain case: the third of three pictures is served as launched; (b) a QP that changes at the third picture (ChangeQPFrame): the launch made with the old QP is voided and
    redone; (c) JMHIP_ADAPTER_FLIGHT=0: no launch ahead of time.  All three write CPU JM's bytes."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    import re as _re
    cpu = os.path.join(ROOT, "oracle", "_ref", "lencod.exe")
    for name, ov, env in (("plain", {}, {}), ("qp change", {"ChangeQPFrame": "1", "ChangeQPP": "6", "ChangeQPI": "6"}, {}), ("off", {}, {"JMHIP_ADAPTER_FLIGHT": "0"})):
        outs = []
        for exe in (cpu, EXE):
            d = os.path.join(str(tmp_path), name.replace(" ", "_") + ("_cpu" if exe == cpu else "_hip"))
            os.makedirs(d)
            shutil.copyfile(os.path.join(G, "foreman_part_qcif.yuv"), os.path.join(d, "foreman_part_qcif.yuv"))
            args = [exe, "-d", os.path.join(G, "jm_baseline.cfg")]
            for k, v in dict({"RDOptimization": "0", "AdaptiveRounding": "0", "SearchMode": "-1", "SearchRange": "16", "NumberReferenceFrames": "2", "FramesToBeEncoded": "3", "FrameSkip": "0",
                              "OutputFile": "o.264", "ReconFile": "o_rec.yuv", "TraceFile": "/dev/null"}, **ov).items():
                args += ["-p", f"{k}={v}"]
            r = subprocess.run(args, cwd=d, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            assert r.returncode == 0, (name, r.stderr.decode(errors="replace")[-1500:])
            outs.append((md5(os.path.join(d, "o.264")), md5(os.path.join(d, "o_rec.yuv")), r.stderr.decode(errors="replace")))
        assert outs[0][:2] == outs[1][:2], (name, "the drop-in encoder's bytes differ from CPU JM's")
        m = _re.search(r"pictures in flight: (\d+) pictures, (\d+) launched ahead of time \(up to (\d+) in flight\), (\d+) of them served as launched, (\d+) voided", outs[1][2])
        if name == "off":
            assert m is None, outs[1][2][-800:]
        else:
            assert m, outs[1][2][-1500:]
            pics, ahead, depth, hit, void = (int(x) for x in m.groups())
            assert pics == 3 and ahead >= 1, (name, m.groups())
            if name == "qp change":
                assert void >= 1 and hit == 0, (name, m.groups())      # the third picture was launched with the second one's QP: found out, voided, launched again
            else:
                assert hit >= 1 and void == 0, (name, m.groups())


@pytest.mark.gpu
@pytest.mark.parametrize("name,ov,max_void,min_hit", [
    ("intra period", {"IntraPeriod": "4"}, 6, 6),                                       # an I picture where a P picture was launched ahead of time: at most the pictures in flight (3) per event
    ("IDR period", {"IDRPeriod": "4"}, 6, 6),                                           # ... and the reference list starts again behind it
    ("both", {"IDRPeriod": "5", "IntraPeriod": "3"}, 9, 3),
    ("QP change", {"ChangeQPFrame": "5", "ChangeQPP": "5", "ChangeQPI": "5"}, 3, 6),
    ("EPZS, IDR period", {"SearchMode": "3", "IDRPeriod": "6"}, 15, 5),                 # sixteen entries: one IDR picture voids what is in flight behind it
])
def test_lencod_look_ahead_voids_are_bounded_and_harmless(tmp_path, name, ov, max_void, min_hit):
    """Twelve pictures with events the adapter's look-ahead cannot foresee (intra / IDR periods, a QP change): the launches made ahead of time with the wrong parameters are
    found out byte for byte and redone -- the .264 and the reconstruction are CPU JM's --, and what that costs is bounded: at most the pictures in flight per event are voided,
    the others are served as launched (profiles/r05_void_probe.txt: 3 voided / 9 served for the full searches, 11 / 8 for EPZS with its sixteen entries)."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    import re as _re
    import sys as _sys
    import numpy as np
    _sys.path.insert(0, G)
    import synth_motion
    cpu = os.path.join(ROOT, "oracle", "_ref", "lencod.exe")
    clip = np.concatenate(synth_motion.motion_clip(176, 144, 12, 123))
    outs = []
    for exe in (cpu, EXE):
        d = os.path.join(str(tmp_path), "cpu" if exe == cpu else "hip")
        os.makedirs(d)
        clip.tofile(os.path.join(d, "motion.yuv"))
        args = [exe, "-d", os.path.join(G, "jm_baseline.cfg")]
        for k, v in dict({"InputFile": "motion.yuv", "RDOptimization": "0", "AdaptiveRounding": "0", "SearchMode": "-1", "SearchRange": "16", "NumberReferenceFrames": "2",
                          "FramesToBeEncoded": "12", "FrameSkip": "0", "OutputFile": "o.264", "ReconFile": "o_rec.yuv", "TraceFile": "/dev/null"}, **ov).items():
            args += ["-p", f"{k}={v}"]
        r = subprocess.run(args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, (name, r.stderr.decode(errors="replace")[-1500:])
        outs.append((md5(os.path.join(d, "o.264")), md5(os.path.join(d, "o_rec.yuv")), r.stderr.decode(errors="replace")))
    assert outs[0][:2] == outs[1][:2], (name, "the drop-in encoder's bytes differ from CPU JM's")
    m = _re.search(r"pictures in flight: (\d+) pictures, (\d+) launched ahead of time \(up to (\d+) in flight\), (\d+) of them served as launched, (\d+) voided", outs[1][2])
    assert m, outs[1][2][-1500:]
    pics, ahead, depth, hit, void = (int(x) for x in m.groups())
    assert pics == 12 and void <= max_void and hit >= min_hit, (name, m.groups())
    assert "passed to JM's own code: 0 calls" in outs[1][2], outs[1][2][-800:]


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["m3b", "m2b4", "m3bt"])
def test_lencod_b_pictures_are_launched_ahead_of_time_too(tmp_path, tag):
    """Sequences with B pictures (coding order I P B P B ...): the adapter predicts the order, the frames' places in the file and both reference lists of the pictures to come
    (a sliding window's init_lists_p_slice / init_lists_b_slice) and launches them ahead of time -- the B pictures beside the P pictures that follow them --, every launch
    checked against what JM really has when it gets there.  Pictures must have been served as launched, none voided, the bytes CPU JM's."""
    import re as _re
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    r, z = run_rdo_off_case(tag, str(tmp_path))
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0, err[-1500:]
    assert md5(os.path.join(str(tmp_path), "o.264")) == str(z["md5_264"]) and md5(os.path.join(str(tmp_path), "o_rec.yuv")) == str(z["md5_recon"]), (tag, "differs from CPU JM")
    m = _re.search(r"pictures in flight: (\d+) pictures, (\d+) launched ahead of time \(up to (\d+) in flight\), (\d+) of them served as launched, (\d+) voided", err)
    assert m, err[-1500:]
    pics, ahead, depth, hit, void = (int(x) for x in m.groups())
    npic = len(z["slice_type"])
    # the first B picture is launched when JM gets there (no B picture's parameters are known before); from then on everything is launched ahead
    assert pics == npic and hit >= npic - 3 and void == 0, (tag, m.groups())


@pytest.mark.gpu
@pytest.mark.parametrize("depth", ["0", "2", "3", "10"])
def test_lencod_b_pictures_with_any_number_of_pictures_in_flight(tmp_path, depth):
    """m3b (I P B P B P B, three references, the bi-predictive search) with JMHIP_ADAPTER_FLIGHT = 0 (off: picture after picture), 2, 3 (fewer entries than a group of pictures
    needs ahead) and 10: the same bytes as CPU JM's whatever the depth."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    r, z = run_rdo_off_case("m3b", str(tmp_path), env_extra={"JMHIP_ADAPTER_FLIGHT": depth})
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0, err[-1500:]
    assert md5(os.path.join(str(tmp_path), "o.264")) == str(z["md5_264"]) and md5(os.path.join(str(tmp_path), "o_rec.yuv")) == str(z["md5_recon"]), (depth, "differs from CPU JM", err[-600:])
    assert ("pictures in flight:" in err) == (depth != "0"), err[-800:]


@pytest.mark.gpu
@pytest.mark.parametrize("full", [False, True])
def test_lencod_leaves_with_and_without_the_teardown(tmp_path, full):
    """After a normal end the adapter writes its report and tears its contexts down; with JMHIP_ADAPTER_FAST_EXIT=1 it flushes every stream and leaves with _exit instead
    (jmhip_destroy and the runtime's exit handlers free what the driver reclaims anyway: INTEGRATION.md section 10).  Either way: exit status 0, the whole of stdout (JM's summary
    is its last output), the same bytes as CPU JM."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    r, z = run_rdo_off_case("q1r", str(tmp_path), env_extra=dict({"JMHIP_INIT_PROF": "1"}, **({} if full else {"JMHIP_ADAPTER_FAST_EXIT": "1"})))
    out, err = r.stdout.decode(errors="replace"), r.stderr.decode(errors="replace")
    assert r.returncode == 0, err[-1500:]
    assert md5(os.path.join(str(tmp_path), "o.264")) == str(z["md5_264"]) and md5(os.path.join(str(tmp_path), "o_rec.yuv")) == str(z["md5_recon"]), ("differs from CPU JM", err[-600:])
    assert "Total bits" in out and "Exit JM" in out, out[-600:]                      # stdout complete through a pipe
    assert "jmhip adapter: macroblock pipeline:" in err
    assert ("exit: jmhip_destroy done" in err) == full and ("exit: leaving without the teardown" in err) == (not full), err[-800:]


@pytest.mark.gpu
def test_lencod_configs3_full_size_2160p_per_call(tmp_path):
    """BASELINE.json configs[3] as SURVEY.md 8c states it (G4: RDO on, 2160p, 8 slices, md5 933ebd28...): the per-call path (every BlockMotionSearch of
    the P picture on the device one by one, sub-pel planes, deblocking), as the 1080p test does for configs[1].  Minutes, not seconds."""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    r, e = run_2160p("G4", str(tmp_path), {"JMHIP_ADAPTER_PARTS": "load,interp,fs,subpel,deblock"})
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0, (r.stdout.decode(errors="replace")[-1500:], err[-1500:])
    c = counters(err)
    assert e["md5_264"] == "933ebd28693881fb1a22886d332d751d"
    assert c["fs"] > 1000000 and c["subpel"] > 1000000 and c["interp"] == 2 and c["deblock"] == 2, c
    assert md5(os.path.join(str(tmp_path), "o.264")) == e["md5_264"], ("bitstream differs from CPU JM", c)
    assert md5(os.path.join(str(tmp_path), "o_rec.yuv")) == e["md5_recon"], ("reconstruction differs from CPU JM", c)


def test_adapter_fails_loudly_without_a_device(tmp_path):
    """not gpu: on a box without a HIP device the adapter must stop the encoder, not fall back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    if not os.path.exists(EXE):
        pytest.skip("lencod_hip.exe not built here")
    r, out264, _ = run_lencod(EXE, "G1_1ref_2frames", str(tmp_path))
    assert r.returncode != 0
    assert b"jmhip_create failed" in r.stderr


def test_adapter_off_is_plain_jm(tmp_path):
    """not gpu: JMHIP_ADAPTER=off passes every call to JM's own functions -- the wrapped binary is the reference encoder."""
    if not os.path.exists(EXE):
        pytest.skip("lencod_hip.exe not built here")
    r, out264, rec = run_lencod(EXE, "G1_1ref_2frames", str(tmp_path), {"JMHIP_ADAPTER": "off"})
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-1500:]
    assert md5(out264) == MD5["G1_1ref_2frames"]["md5_264"]


@pytest.mark.gpu
def test_random_rdo_off_configurations_equal_cpu_jm():
    """Forty seconds of tests/fuzz_dropin.py: seeded random RDO-off configurations (search mode and range, references, QPs of I and P slices, chroma QP offset, entropy coder, 8x8
    transform, 4:2:0 / 4:2:2, slices, loop filter parameters, partition switches, intra period, cropped picture sizes, EPZS switches) through lencod_hip.exe and CPU JM: .264 and
    reconstruction byte-identical.  (Twelve minutes of the same script: profiles/r03_fuzz_dropin.txt; it is what found the stale block-type predictors of switched-off partitions.)"""
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/lencod_hip.exe missing")
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_dropin.py"), "40", "777000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and " 0 NOT byte-identical" in out, out[-3000:]


@pytest.mark.gpu
def test_vector_limits_below_the_search_range_go_to_jms_own_function(tmp_path):
    """UseMVLimits with SetMVXLimit 8 under SearchRange 32 (conformance.c:615-630): jmhip_encode_slice refuses limits narrower than the search range, so the adapter must turn the
    sequence away BEFORE the first slice (pipe_config_ok) -- JM's own encode_one_macroblock_low runs, the output equals CPU JM's.  (The drop-in used to exit here: found by
    tests/fuzz_dropin.py's seed 500000.)"""
    import sys
    import numpy as np
    sys.path.insert(0, G)
    import synth_motion
    cpu = os.path.join(ROOT, "oracle", "_ref", "lencod.exe")
    if not os.path.exists(EXE) or not os.path.exists(cpu):
        pytest.fail("oracle/_ref/lencod_hip.exe / lencod.exe missing")
    sw, sh = 96, 80
    np.concatenate(synth_motion.motion_clip(sw, sh, 3, 500000)).tofile(str(tmp_path / "clip.yuv"))
    ov = dict(RDOptimization=0, AdaptiveRounding=0, InputFile="clip.yuv", SourceWidth=sw, SourceHeight=sh, OutputWidth=sw, OutputHeight=sh, FramesToBeEncoded=3, SearchMode=-1, SearchRange=32,
              NumberReferenceFrames=2, NumberBFrames=0, UseMVLimits=1, SetMVXLimit=8, SetMVYLimit=512)
    out = {}
    for exe, tag in ((cpu, "c"), (EXE, "h")):
        args = [exe, "-d", os.path.join(G, "jm_baseline.cfg")]
        for k, v in dict(ov, OutputFile=f"{tag}.264", ReconFile=f"{tag}.yuv", TraceFile="/dev/null").items():
            args += ["-p", f"{k}={v}"]
        r = subprocess.run(args, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
        out[tag] = (md5(str(tmp_path / f"{tag}.264")), md5(str(tmp_path / f"{tag}.yuv")), r.stderr.decode(errors="replace"))
    assert out["c"][:2] == out["h"][:2]
    assert "macroblock pipeline not used (vector limits" in out["h"][2], out["h"][2][-1500:]


@pytest.mark.gpu
def test_lencod_small_picture_whose_workgroups_code_one_macroblock_each_many_times(tmp_path):
    """fuzz_dropin's seed 700411 (round 6): 176 x 112, 4:2:2, 8x8 transform, fast full search with up to four references, IDRPeriod 3, a QP change at picture 2 -- 77 macroblocks
    per picture against 80 workgroups per picture in flight, so every workgroup codes ONE macroblock and runs its post stage behind the loop.  One run in fifty left a macroblock's
    chroma of picture 2 at zero: the post stage read its picture's descriptor (Shared::Vp) before the other waves' threads had copied it (profiles/r06_vp_race.txt).  Thirty runs
    against CPU JM's bytes (the reference binary travels with the repo: oracle/_ref/lencod.exe)."""
    import sys
    import numpy as np
    cpu = os.path.join(ROOT, "oracle", "_ref", "lencod.exe")
    if not os.path.exists(EXE) or not os.path.exists(cpu):
        pytest.fail("oracle/_ref/lencod_hip.exe / lencod.exe missing: run __graft_entry__.build() where /root/reference exists")
    sys.path.insert(0, G)
    import synth_motion
    ov = dict(RDOptimization=0, AdaptiveRounding=0, InputFile="clip.yuv", SourceWidth=176, SourceHeight=112, OutputWidth=176, OutputHeight=112, FramesToBeEncoded=4, YUVFormat=2,
              ProfileIDC=122, LevelIDC=40, SymbolMode=1, Transform8x8Mode=1, SearchMode=0, SearchRange=28, NumberReferenceFrames=4, QPISlice=50, QPPSlice=0, ChromaQPOffset=-3,
              DisableSubpelME=0, IntraPeriod=3, NumberBFrames=0, OffsetMatrixPresentFlag=1, CbQPOffset=-4, CrQPOffset=1, ReferenceReorder=1, PocMemoryManagement=1, PicOrderCntType=2,
              IDRPeriod=3, ChangeQPFrame=2, ChangeQPI=17, ChangeQPP=21)
    tmp = str(tmp_path)
    np.concatenate(synth_motion.motion_clip(176, 112, 4, 700411, yuv422=True)).tofile(os.path.join(tmp, "clip.yuv"))
    shutil.copyfile(os.path.join(G, "q_offset.cfg"), os.path.join(tmp, "q_offset.cfg"))

    def run(exe, tag):
        args = [exe, "-d", os.path.join(G, "jm_baseline.cfg")]
        for k, v in dict(ov, OutputFile=f"{tag}.264", ReconFile=f"{tag}.yuv", TraceFile="/dev/null").items():
            args += ["-p", f"{k}={v}"]
        return subprocess.run(args, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert run(cpu, "c").returncode == 0
    want = md5(os.path.join(tmp, "c.264")), md5(os.path.join(tmp, "c.yuv"))
    for k in range(30):
        r = run(EXE, "h")
        err = r.stderr.decode(errors="replace")
        assert r.returncode == 0, (k, err[-1200:])
        assert "macroblock pipeline:" in err and "pictures in flight:" in err, err[-800:]
        assert (md5(os.path.join(tmp, "h.264")), md5(os.path.join(tmp, "h.yuv"))) == want, (k, "differs from CPU JM")
