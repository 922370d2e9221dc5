#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference encoder.

Runs in the build container only (needs /root/reference and oracle/_ref/, built by
`make -C oracle ref`).  It executes the unmodified JM 19.0 lencod -- plain and with the
--wrap taps of oracle/ref_tap.c -- on the reference's own sample clips and configs, and
stores what the reference's hot-path functions were given and what they returned:

  tests/golden/qcif_fs.npz    FullSearch SR=16, 1 ref, baseline cfg: full_search / sub_pel ME
                              call records, sub-pel plane digests, deblock in/out, 4x4
                              transform / quant / reconstruct records
  tests/golden/qcif_ffs.npz   FastFullSearch SR=16: BlockSAD tables + argmin records
  tests/golden/qcif_422.npz   High 4:2:2, 8x8 transform, CABAC: deblock in/out (intra + inter)
  tests/golden/qcif_main.npz  Main profile with a B frame: deblock in/out (two-list strengths)
  tests/golden/qcif_mc.npz    luma_prediction / chroma_prediction_4x4 records (4:2:0 P, 4:2:2 P, B picture) + the reference planes they read
  tests/golden/qcif_intra.npz get_intrapred_4x4 records and find_sad_16x16_JM (Intra16x16 mode search) records of the same three runs
  tests/golden/qcif_pad.npz   digests of the source picture after read_one_frame + pad_borders for a 168x136 source (coded 176x144)
  tests/golden/md5.json       .264 / recon md5 of the BASELINE.json configurations at QCIF

Fixtures are data only (inputs and expected outputs); no reference source is stored.
"""
import hashlib, json, os, shutil, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/bin"
EXE = os.path.join(ROOT, "oracle/_ref/lencod.exe")
TAP = os.path.join(ROOT, "oracle/_ref/lencod_tap.exe")
DEC = os.path.join(ROOT, "oracle/_ref/ldecod.exe")
OUT = os.path.join(ROOT, "tests/golden")


def run(exe, cfg, overrides, workdir, tap=False, tap_max=6000):
    # run as JM users do: from a directory holding the reference's cfg / yuv files (q_offset.cfg
    # and friends are opened relative to the cwd), here a scratch dir of symlinks to them
    for f in os.listdir(REF):
        if not os.path.exists(os.path.join(workdir, f)):
            os.symlink(os.path.join(REF, f), os.path.join(workdir, f))
    args = [exe, "-d", cfg]
    ov = dict(OutputFile="o.264", ReconFile="o_rec.yuv", TraceFile="/dev/null")
    ov.update(overrides)
    for k, v in ov.items():
        args += ["-p", f"{k}={v}"]
    env = dict(os.environ)
    if tap:
        env["JM_TAP_DIR"] = workdir
        env["JM_TAP_MAX"] = str(tap_max)
    r = subprocess.run(args, cwd=workdir, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return ov


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


class Reader:
    def __init__(self, path):
        self.b = open(path, "rb").read() if os.path.exists(path) else b""
        self.o = 0

    def eof(self):
        return self.o >= len(self.b)

    def i32(self, n=1):
        a = np.frombuffer(self.b, np.int32, n, self.o)
        self.o += 4 * n
        return a if n > 1 else int(a[0])

    def i64(self):
        a = np.frombuffer(self.b, np.int64, 1, self.o)
        self.o += 8
        return int(a[0])

    def plane(self):
        h, w = self.i32(), self.i32()
        a = np.frombuffer(self.b, np.uint16, h * w, self.o).reshape(h, w)
        self.o += 2 * h * w
        return a


def read_me(path, n32a, n32b):
    """records: n32a int32, one int64 (min_mcost in), n32b int32 (mv out), one int64 (cost)."""
    r, rows = Reader(path), []
    while not r.eof():
        a = list(r.i32(n32a)); mc = r.i64(); b = list(r.i32(n32b)); c = r.i64()
        rows.append(a + [mc] + b + [c])
    return np.array(rows, np.int64)


def read_cur(path):
    r, out = Reader(path), {}
    while not r.eof():
        k = r.i32(); out[k] = r.plane().astype(np.uint8)
    return out


def read_subimages(path):
    r, out = Reader(path), []
    while not r.eof():
        idx, w, h, maxv = r.i32(), r.i32(), r.i32(), r.i32()
        src = r.plane()
        planes = [r.plane() for _ in range(16)]
        out.append(dict(idx=idx, w=w, h=h, maxv=maxv, src=src.astype(np.uint8),
                        sha=[hashlib.sha256(p.astype(np.uint8).tobytes()).hexdigest() for p in planes],
                        planes=planes))
    return out


def read_mc(path, luma):
    """mc_luma.bin: 13 ints + 5 weight ints + bsx*bsy samples;  mc_chroma.bin: 44 ints + 5 weight ints + 16 samples (oracle/ref_tap.c)"""
    r, hdr, pix = Reader(path), [], []
    while not r.eof():
        h = r.i32(18 if luma else 49).copy()
        n = int(h[3]) * int(h[4]) if luma else 16
        px = np.frombuffer(r.b, np.uint16, n, r.o).astype(np.uint8); r.o += 2 * n
        full = np.zeros(256 if luma else 16, np.uint8); full[:n] = px
        hdr.append(h); pix.append(full)
    return (np.array(hdr, np.int32).reshape(-1, 18 if luma else 49), np.array(pix, np.uint8).reshape(-1, 256 if luma else 16))


def read_refchroma(path):
    r, out = Reader(path), {}
    while not r.eof():
        idx, fmt = r.i32(), r.i32()
        out[int(idx)] = (r.plane().astype(np.uint8), r.plane().astype(np.uint8))
    return out


def mc_arrays(tag, workdir, d, n_luma=260, n_chroma=420):
    """prediction records of one tapped run: a spread sample, every record whose vector leaves the picture far enough to hit the
    origin clamps, and the reference pictures (luma source + integer chroma planes) they read"""
    lh, lp = read_mc(os.path.join(workdir, "mc_luma.bin"), True)
    ch, cp = read_mc(os.path.join(workdir, "mc_chroma.bin"), False)
    def pick(h, n, far):
        keep = np.zeros(len(h), bool); keep[:: max(1, len(h) // n)] = True; keep |= far
        return np.flatnonzero(keep)[: 2 * n]
    mv = np.abs(lh[:, [8, 9, 11, 12]]).max(1)
    il = pick(lh, n_luma, (mv > 40) | (lh[:, 5] == 2))
    mvc = np.abs(ch[:, 10:26]).max(1)
    ic = pick(ch, n_chroma, (mvc > 40) | (ch[:, 5] == 2))
    d[tag + "_mcl_hdr"], d[tag + "_mcl_pix"] = lh[il], lp[il]
    d[tag + "_mcc_hdr"], d[tag + "_mcc_pix"] = ch[ic], cp[ic]
    used = set(lh[il][:, 7].tolist()) | set(lh[il][:, 10].tolist()) | set(ch[ic][:, 8].tolist()) | set(ch[ic][:, 26].tolist())
    subs = {s_["idx"]: s_ for s_ in read_subimages(os.path.join(workdir, "subimages.bin"))}
    rc = read_refchroma(os.path.join(workdir, "refchroma.bin"))
    for k in sorted(u for u in used if u >= 0):
        d[f"{tag}_ref{k}_y"] = subs[k]["src"]
        d[f"{tag}_ref{k}_u"], d[f"{tag}_ref{k}_v"] = rc[k]
    # getSubImagesChroma of the first reference picture: a digest of every sub-image of both planes, and every 13th row of the U planes
    r = Reader(os.path.join(workdir, "chromasub.bin"))
    idx, fmt, ny, nx, py, px = [r.i32() for _ in range(6)]
    planes = [[r.plane().astype(np.uint8) for _ in range(ny * nx)] for _ in range(2)]
    assert idx == 0, idx
    d[tag + "_csub_hdr"] = np.array([idx, fmt, ny, nx, py, px], np.int32)
    d[tag + "_csub_sha"] = np.array([[hashlib.sha256(p.tobytes()).hexdigest() for p in pl] for pl in planes])
    d[tag + "_csub_u_rows"] = np.stack([p[::13] for p in planes[0]])


def intra_arrays(tag, workdir, d, n4=450, n16=40, nc=120, n8=400):
    """get_intrapred_4x4 records (mode left up max_pel | 13 predictor samples | 16 predicted) and find_sad_16x16_JM records
    (left up upleft mode_mask metric max_pel | 33 predictor samples | source 16x16 | cost lo hi, i16mode | 4 predictions) of one tapped run"""
    a = read_i32_records(os.path.join(workdir, "intra4x4.bin"), 33)
    a = np.unique(a, axis=0)
    d[tag + "_i4"] = a[:: max(1, -(-len(a) // n4))].astype(np.int16)          # sorted by mode: an even stride keeps all nine
    b = read_i32_records(os.path.join(workdir, "intra16_search.bin"), 6 + 33 + 256 + 3 + 1024)
    b = b[:: max(1, -(-len(b) // n16))]
    d[tag + "_i16_hdr"] = b[:, :6].astype(np.int16)
    d[tag + "_i16_edge"] = b[:, 6:39].astype(np.uint8)
    d[tag + "_i16_orig"] = b[:, 39:295].astype(np.uint8)
    d[tag + "_i16_cost"] = (b[:, 295].astype(np.int64) & 0xffffffff) | (b[:, 296].astype(np.int64) << 32)
    d[tag + "_i16_mode"] = b[:, 297].astype(np.int16)
    d[tag + "_i16_pred"] = b[:, 298:].reshape(-1, 4, 256).astype(np.uint8)
    # intra_chroma_prediction records: yuv up left upleft | per plane up[8] left[16] corner | per plane per mode 16 rows x 8 (distinct ones, spread)
    c = np.unique(read_i32_records(os.path.join(workdir, "intra_chroma.bin"), 4 + 2 * 25 + 2 * 4 * 128), axis=0)
    c = c[:: max(1, -(-len(c) // nc))]
    p8 = os.path.join(workdir, "intra8x8.bin")
    if os.path.exists(p8):                                  # get_intrapred_8x8 records (8x8 transform runs): mode left up | 25 samples | 64 predicted
        e = np.unique(read_i32_records(p8, 3 + 25 + 64), axis=0)
        d[tag + "_i8"] = e[:: max(1, -(-len(e) // n8))].astype(np.int16)
    d[tag + "_ic_hdr"] = c[:, :4].astype(np.int16)
    d[tag + "_ic_edge"] = c[:, 4:54].reshape(-1, 2, 25).astype(np.uint8)
    d[tag + "_ic_pred"] = c[:, 54:].reshape(-1, 2, 4, 16, 8).astype(np.uint8)


def read_deblock(path):
    r, out = Reader(path), []
    while not r.eof():
        n, w, h, fmt, maxy, maxc, d8, nmb = [r.i32() for _ in range(8)]
        mbs = r.i32(12 * nmb).reshape(nmb, 12).copy()
        mot = r.i32((h // 4) * (w // 4) * 6).reshape(h // 4, w // 4, 2, 3).copy()
        pre = [r.plane().astype(np.uint8)]
        if fmt != 0:
            pre += [r.plane().astype(np.uint8), r.plane().astype(np.uint8)]
        post = [r.plane().astype(np.uint8)]
        if fmt != 0:
            post += [r.plane().astype(np.uint8), r.plane().astype(np.uint8)]
        out.append(dict(n=n, w=w, h=h, fmt=fmt, maxy=maxy, maxc=maxc, d8=d8, mbs=mbs, mot=mot, pre=pre, post=post))
    return out


def read_i32_records(path, width):
    if not os.path.exists(path):
        return np.zeros((0, width), np.int32)
    a = np.fromfile(path, np.int32)
    return a.reshape(-1, width)


def deblock_arrays(prefix, recs, d):
    for i, r in enumerate(recs):
        p = f"{prefix}{i}_"
        d[p + "hdr"] = np.array([r["w"], r["h"], r["fmt"], r["maxy"], r["maxc"], r["d8"]], np.int32)
        d[p + "mbs"] = r["mbs"].astype(np.int32)
        d[p + "mot"] = r["mot"].astype(np.int32)
        for k, nm in enumerate("yuv"[: len(r["pre"])]):
            d[p + "pre_" + nm] = r["pre"][k]
            d[p + "post_" + nm] = r["post"][k]


def main():
    md5s = {}
    tmp = tempfile.mkdtemp(prefix="jmgold_")
    try:
        # ---- A: full search, 1 reference, 2 frames (I, P)
        wa = os.path.join(tmp, "A"); os.makedirs(wa)
        ov = dict(SearchMode=-1, SearchRange=16, NumberReferenceFrames=1, FramesToBeEncoded=2)
        run(TAP, "encoder_baseline.cfg", ov, wa, tap=True)
        tapped = md5(os.path.join(wa, "o.264"))
        wa2 = os.path.join(tmp, "A2"); os.makedirs(wa2)
        run(EXE, "encoder_baseline.cfg", ov, wa2)
        assert tapped == md5(os.path.join(wa2, "o.264")), "taps changed the bitstream"
        d = {}
        cur = read_cur(os.path.join(wa, "cur_frames.bin"))
        for k, v in cur.items():
            d[f"cur{k}"] = v
        subs = read_subimages(os.path.join(wa, "subimages.bin"))
        for s in subs:
            d[f"ref{s['idx']}_src"] = s["src"]
            d[f"ref{s['idx']}_sha"] = np.array(s["sha"])
        # one full set of planes, sparsely sampled, for quick localisation of a mismatch
        d["ref0_plane_rows"] = np.stack([p[::23].astype(np.uint8) for p in subs[0]["planes"]])
        d["me_fs"] = read_me(os.path.join(wa, "me_fs.bin"), 13, 2)
        d["me_subpel"] = read_me(os.path.join(wa, "me_subpel.bin"), 18, 2)
        deblock_arrays("db", read_deblock(os.path.join(wa, "deblock.bin")), d)
        d["fwd4x4"] = read_i32_records(os.path.join(wa, "fwd4x4.bin"), 32)
        d["inv4x4"] = read_i32_records(os.path.join(wa, "inv4x4.bin"), 32)
        d["quant4x4_around"] = read_i32_records(os.path.join(wa, "quant4x4_around.bin"), 136)
        d["recon4x4"] = read_i32_records(os.path.join(wa, "recon4x4.bin"), 50)
        np.savez_compressed(os.path.join(OUT, "qcif_fs.npz"), **d)

        # ---- A3: same without adaptive rounding -> quant_4x4_normal records
        wa3 = os.path.join(tmp, "A3"); os.makedirs(wa3)
        run(TAP, "encoder_baseline.cfg", dict(ov, AdaptiveRounding=0), wa3, tap=True)
        qn = read_i32_records(os.path.join(wa3, "quant4x4_normal.bin"), 136)

        # ---- B: fast full search
        wb = os.path.join(tmp, "B"); os.makedirs(wb)
        run(TAP, "encoder_baseline.cfg", dict(ov, SearchMode=0), wb, tap=True)
        d = {"quant4x4_normal": qn}
        for k, v in read_cur(os.path.join(wb, "cur_frames.bin")).items():
            d[f"cur{k}"] = v
        for s in read_subimages(os.path.join(wb, "subimages.bin")):
            d[f"ref{s['idx']}_src"] = s["src"]
        r, setups, tables = Reader(os.path.join(wb, "me_ffs_setup.bin")), [], []
        while not r.eof():
            hdr = list(r.i32(8))
            tab = np.frombuffer(r.b, np.uint32, 7 * 16 * hdr[7], r.o).reshape(7, 16, hdr[7]); r.o += 4 * tab.size
            setups.append(hdr + [int(hashlib.sha256(tab.astype(np.uint16).tobytes()).hexdigest()[:12], 16)])
            tables.append(tab.astype(np.uint16))
        d["ffs_setup"] = np.array(setups, np.int64)
        d["ffs_table0"] = tables[0]
        d["ffs_table7"] = tables[7]
        d["me_ffs"] = read_me(os.path.join(wb, "me_ffs.bin"), 15, 2)
        np.savez_compressed(os.path.join(OUT, "qcif_ffs.npz"), **d)

        # ---- C: High 4:2:2, 8x8 transform (encoder_yuv422.cfg) -> deblock
        wc = os.path.join(tmp, "C"); os.makedirs(wc)
        run(TAP, "encoder_yuv422.cfg", dict(NumberBFrames=0, FramesToBeEncoded=2), wc, tap=True)
        d = {}
        deblock_arrays("db", read_deblock(os.path.join(wc, "deblock.bin")), d)
        np.savez_compressed(os.path.join(OUT, "qcif_422.npz"), **d)

        # ---- E: Main profile with B frames -> deblock with two lists
        we = os.path.join(tmp, "E"); os.makedirs(we)
        run(TAP, "encoder_main.cfg", dict(FramesToBeEncoded=3), we, tap=True)
        d = {}
        deblock_arrays("db", read_deblock(os.path.join(we, "deblock.bin")), d)
        np.savez_compressed(os.path.join(OUT, "qcif_main.npz"), **d)

        # ---- motion-compensated prediction (luma_prediction / chroma_prediction_4x4) of runs A (4:2:0 P), C (4:2:2 P), E (B picture)
        d = {}
        mc_arrays("a", wa, d); mc_arrays("c", wc, d); mc_arrays("e", we, d)
        # explicit weighted prediction in P and B pictures (the G3w configuration): weighted_mc_prediction / weighted_bi_prediction
        ww = os.path.join(tmp, "W"); os.makedirs(ww)
        run(TAP, "encoder_main.cfg", dict(SearchMode=3, WeightedPrediction=1, WeightedBiprediction=1), ww, tap=True)
        mc_arrays("w", ww, d)
        np.savez_compressed(os.path.join(OUT, "qcif_mc.npz"), **d)

        # ---- luma intra prediction (get_intrapred_4x4) and the Intra16x16 mode search (find_sad_16x16_JM) of the same runs
        d = {}
        intra_arrays("a", wa, d); intra_arrays("c", wc, d); intra_arrays("e", we, d)
        np.savez_compressed(os.path.join(OUT, "qcif_intra.npz"), **d)

        # ---- D: 8x8 transform / quantisation and the DC transforms (High 4:2:2 CABAC + adaptive rounding; the same with CAVLC
        #         and plain rounding; High 4:2:0 CAVLC for the 2x2 chroma DC transform)
        d, q8, r8, rc, r16 = {}, [], [], [], []
        for name, cfg, o in [("D0", "encoder_baseline.cfg", dict(ov)),
                             ("D", "encoder_yuv422.cfg", dict(NumberBFrames=0, FramesToBeEncoded=2)),
                             ("D2", "encoder_yuv422.cfg", dict(NumberBFrames=0, FramesToBeEncoded=2, SymbolMode=0, AdaptiveRounding=0)),
                             ("D3", "encoder_main.cfg", dict(FramesToBeEncoded=2, Transform8x8Mode=1, ProfileIDC=100, SymbolMode=0)),
                             ("D4", "encoder_yuv422.cfg", dict(NumberBFrames=0, FramesToBeEncoded=2, AdaptiveRounding=0))]:
            w = os.path.join(tmp, name); os.makedirs(w)
            run(TAP, cfg, o, w, tap=True)
            q = read_i32_records(os.path.join(w, "quant8x8.bin"), 718)
            r = read_i32_records(os.path.join(w, "rtq8x8.bin"), 530)
            # keep the records that exercise something: at least one non-zero level, spread over the run
            if len(q):
                q = q[np.abs(q[:, 4 + 192 + 128 + 64 + 64 + 64:4 + 192 + 128 + 64 + 64 + 64 + 68]).sum(1) > 0]
            if len(q):
                q8.append(q[:: max(1, len(q) // 60)][:60]); r8.append(r[:: max(1, len(r) // 60)][:60])
            i16 = read_i32_records(os.path.join(w, "rtq16x16.bin"), 1435)
            if len(i16):                                            # Intra16x16 luma: a spread sample, the few all-zero-AC records included
                quiet = i16[i16[:, 568] == 0]
                r16.append(i16[:: max(1, len(i16) // 18)][:18]); r16.append(quiet[:3])
            c = read_i32_records(os.path.join(w, "rtq_chroma.bin"), 853)
            busy = c[np.abs(c[:, 451:460]).sum(1) + np.abs(c[:, 469:725]).sum(1) > 0]       # some DC or AC level survives
            quiet = c[np.abs(c[:, 451:460]).sum(1) + np.abs(c[:, 469:725]).sum(1) == 0]
            rc.append(busy[:: max(1, len(busy) // 45)][:45]); rc.append(quiet[:: max(1, len(quiet) // 10)][:10])
            for nm, wd in (("fwd8x8", 128), ("inv8x8", 128), ("hadamard4x4", 32), ("ihadamard4x4", 32), ("hadamard4x2", 16),
                           ("ihadamard4x2", 16), ("hadamard2x2", 8), ("ihadamard2x2", 8), ("quant_dc4x4", 73)):
                a = read_i32_records(os.path.join(w, nm + ".bin"), wd)
                if len(a):
                    a = np.unique(a, axis=0)
                    a = a[np.abs(a).sum(1) > 0]
                    d.setdefault(nm, []).append(a[:: max(1, len(a) // 80)][:80])
        d = {k: np.concatenate(v) for k, v in d.items()}
        d["quant8x8"] = np.concatenate(q8)
        d["rtq8x8"] = np.concatenate(r8)
        d["rtq_chroma"] = np.concatenate(rc)
        d["rtq16x16"] = np.concatenate(r16)
        np.savez_compressed(os.path.join(OUT, "qcif_tq8.npz"), **d)

        # ---- P: a source size that is not a multiple of 16 (168x136 read from the QCIF clip's bytes): read_one_frame + pad_borders fill the
        #         right 8 columns and the bottom 8 rows of the coded 176x144 picture; digests of the P picture's planes (clip frame 1)
        wp = os.path.join(tmp, "P"); os.makedirs(wp)
        run(TAP, "encoder_baseline.cfg", dict(ov, SourceWidth=168, SourceHeight=136, OutputWidth=168, OutputHeight=136), wp, tap=True)
        r = Reader(os.path.join(wp, "cur_yuv.bin"))
        idx, fmt = int(r.i32()), int(r.i32())
        planes = [r.plane().astype(np.uint8) for _ in range(3)]
        assert idx == 1 and planes[0].shape == (144, 176) and planes[1].shape == (72, 88)
        np.savez_compressed(os.path.join(OUT, "qcif_pad.npz"), geometry=np.array([168, 136, 176, 144, fmt, idx], np.int32),
                            sha=np.array([hashlib.sha256(p_.tobytes()).hexdigest() for p_ in planes]), y_tail=planes[0][130:, 160:], u_tail=planes[1][64:, 80:])

        # ---- md5 goldens of whole-encoder runs
        runs = {
            "G0": ("encoder_baseline.cfg", {}),
            "G1": ("encoder_baseline.cfg", dict(SearchMode=-1, SearchRange=16)),
            "G1_1ref_2frames": ("encoder_baseline.cfg", ov),
            "G3a": ("encoder_main.cfg", dict(SearchMode=3)),
            "G3b": ("encoder_main.cfg", dict(SearchMode=3, Transform8x8Mode=1, ProfileIDC=100)),
            "G4q": ("encoder_baseline.cfg", dict(SearchMode=-1, SearchRange=16, SliceMode=1, SliceArgument=33, AdaptiveRounding=0)),
            "G5": ("encoder_yuv422.cfg", dict(NumberBFrames=0)),
            # explicit weighted prediction in P and B pictures: compute*WP and computeBiPred*2 with the weights JM estimates (33 / -5, 33+32 / -2);
            # with the 8x8 transform computeBiPredSATD2 takes its 8x8 path (me_distortion.c:1113-1175)
            "G3w": ("encoder_main.cfg", dict(SearchMode=3, WeightedPrediction=1, WeightedBiprediction=1)),
            "G3wb": ("encoder_main.cfg", dict(SearchMode=3, WeightedPrediction=1, WeightedBiprediction=1, Transform8x8Mode=1, ProfileIDC=100)),
        }
        for tag, (cfg, o) in runs.items():
            w = os.path.join(tmp, tag); os.makedirs(w)
            run(EXE, cfg, o, w)
            md5s[tag] = dict(cfg=cfg, overrides={k: str(v) for k, v in o.items()},
                             md5_264=md5(os.path.join(w, "o.264")), md5_recon=md5(os.path.join(w, "o_rec.yuv")),
                             bytes_264=os.path.getsize(os.path.join(w, "o.264")))
        # ---- G2 = BASELINE.json configs[1] at full size: synthetic 1080p (SURVEY.md Appendix A clip as bench.write_yuv makes it),
        #      Baseline IPPP, FullSearch SR=32, 1 reference, two frames (I + P)
        sys.path.insert(0, ROOT)
        import bench
        w = os.path.join(tmp, "G2"); os.makedirs(w)
        bench.write_yuv(os.path.join(w, "syn1080p.yuv"), 2)
        o = dict(InputFile="syn1080p.yuv", SourceWidth=1920, SourceHeight=1080, OutputWidth=1920, OutputHeight=1080, FramesToBeEncoded=2,
                 SearchMode=-1, SearchRange=32, NumberReferenceFrames=1, LevelIDC=51)
        run(EXE, "encoder_baseline.cfg", o, w)
        md5s["G2"] = dict(cfg="encoder_baseline.cfg", overrides={k: str(v) for k, v in o.items()},
                          md5_264=md5(os.path.join(w, "o.264")), md5_recon=md5(os.path.join(w, "o_rec.yuv")),
                          bytes_264=os.path.getsize(os.path.join(w, "o.264")),
                          input="bench.write_yuv(path, 2): SURVEY.md Appendix A clip, first two frames")
        json.dump(md5s, open(os.path.join(OUT, "md5.json"), "w"), indent=1, sort_keys=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    sys.exit(main())
