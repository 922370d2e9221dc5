"""The CPU oracle (oracle/*.c) against golden vectors captured from the REAL reference
encoder (tests/golden/make_golden.py + oracle/ref_tap.c) and against the known-answer
vectors of SURVEY.md Appendix C.  Bit-exact: every comparison is integer equality."""
import hashlib
import os
import numpy as np
import pytest

from oracle import pyjmo as J

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def fs():
    return np.load(os.path.join(G, "qcif_fs.npz"))


@pytest.fixture(scope="module")
def ffs():
    return np.load(os.path.join(G, "qcif_ffs.npz"))


def test_kat_transforms_and_hadamard():
    # SURVEY.md Appendix C (from lcommon/src/transform.c, lencod/src/me_distortion.c linked standalone)
    i4 = [186, -102, 235, -128, -152, -74, -203, -21, -136, -81, 1, 38, 38, 83, 42, -33]
    o4 = [-307, -89, -109, 578, -150, 513, 212, 1619, 949, 1037, -281, 1376, 605, -486, -199, 1757]
    assert J.forward4x4(i4).tolist() == o4
    assert J.inverse4x4(o4).tolist() == [3960, -2772, 4916, -2926, -2715, -860, -3967, -203, -2503, -1526, 641,
                                          1085, 394, 1874, 110, -420]
    d4 = np.array([207, 244, -28, 186, -3, 239, -62, 77, -223, 111, -232, 166, -139, 132, 163, -190], np.int16)
    assert J.L.jmo_hadamard_sad4x4(J._p(d4)) == 4516
    d8 = np.array([-195, 127, -22, 41, -195, -24, 150, 140, -100, -117, 83, 89, -250, -228, 95, -177, 98, -113, 19,
                   65, 99, -54, 247, 230, 69, -17, 73, -53, 178, 194, -151, 153, 200, -182, 115, -123, -55, 198, -35,
                   -231, -177, 16, 109, -175, -60, -52, -222, -202, -192, 162, 156, 49, -87, 211, -225, -204, 252,
                   -80, 77, 196, 146, 114, 70, 125], np.int16)
    assert J.L.jmo_hadamard_sad8x8(J._p(d8)) == 15291
    i8 = [-53, -193, 233, -116, 37, 102, 31, 45, -164, 213, -179, 230, -9, -4, -121, -182, -79, -47, -106, 47, 134,
          -245, -31, 28, -110, 8, 68, 26, -214, -173, -178, -196, -9, 56, 160, 187, -234, -201, -235, -105, 89, 113,
          181, 71, -157, 163, 97, -139, 201, 80, -77, -63, -35, 111, -79, -92, -93, -141, 4, -186, -74, -243, -5, 8]
    o8 = [-1845, 2241, -611, -1432, -569, 625, -24, -457, 213, -960, -1429, -558, 370, -564, -505, -331, 361, -1866,
          273, 1258, -24, -624, 323, 1066, 2485, 856, -157, 809, -2036, 1018, 1818, -717, -1743, -237, 123, -2232,
          -191, 727, 1296, 1531, 251, -624, 588, 1419, -89, -132, 1121, 1357, 542, -1604, 278, -346, -103, 567, 394,
          1850, 188, -273, 257, -392, 1873, -394, 193, 816]
    assert J.forward8x8(i8).tolist() == o8


def test_spiral_closed_form_matches_table():
    for R in (1, 4, 16, 32):
        sp = J.spiral(R)
        n = (2 * R + 1) ** 2
        idx = [J.L.jmo_spiral_index(int(x), int(y)) for x, y in sp[:n]]
        assert idx == list(range(n))


def test_transform_records(fs):
    for rec in fs["fwd4x4"]:
        assert J.forward4x4(rec[:16]).tolist() == rec[16:].tolist()
    for rec in fs["inv4x4"]:
        assert J.inverse4x4(rec[:16]).tolist() == rec[16:].tolist()


def _check_quant(recs, around):
    assert len(recs) > 100
    for r in recs:
        qp, qp_per, cavlc, arw = (int(v) for v in r[:4])
        q = r[4:52].reshape(16, 3)
        tin, tout = r[52:68], r[68:84]
        lev, run = r[84:101], r[101:118]
        dcost, nz = int(r[118]), int(r[119])
        fadj = r[120:136]
        tb, l, rn, c, n, fa = J.quant_4x4(tin, q, qp_per, cavlc, around, arw)
        assert tb.tolist() == tout.tolist()
        k = int(np.argmax(lev == 0)) + 1          # JM writes levels up to and including the 0 terminator
        assert l[:k].tolist() == lev[:k].tolist()
        assert rn[:k - 1].tolist() == run[:k - 1].tolist()
        assert c == dcost and n == nz
        if around:
            assert fa.tolist() == fadj.tolist()
        # the flat-matrix parameter generator must reproduce JM's q_params for inter luma defaults
        assert (J.qparams_4x4(qp, 0, 0)[:, 1:] == q[:, 1:]).all()


def test_quant4x4_around_records(fs):
    _check_quant(fs["quant4x4_around"], True)


def test_quant4x4_normal_records(ffs):
    _check_quant(ffs["quant4x4_normal"], False)


@pytest.fixture(scope="module")
def tq8():
    return np.load(os.path.join(G, "qcif_tq8.npz"))


def _lists_equal(got_l, got_r, want_l, want_r, cavlc):
    """JM writes each list up to and including its 0 terminator; entries behind it are stale."""
    for k in range(4 if cavlc else 1):
        o = 17 * k if cavlc else 0
        n = int(np.argmax(want_l[o:o + (17 if cavlc else 65)] == 0)) + 1
        assert got_l[o:o + n].tolist() == want_l[o:o + n].tolist()
        assert got_r[o:o + n - 1].tolist() == want_r[o:o + n - 1].tolist()


def test_transform8x8_and_dc_transform_records(tq8):
    for rec in tq8["fwd8x8"]:
        assert J.forward8x8(rec[:64]).tolist() == rec[64:].tolist()
    for rec in tq8["inv8x8"]:
        assert J.inverse8x8(rec[:64]).tolist() == rec[64:].tolist()
    for name, fn, n in (("hadamard4x4", J.hadamard4x4, 16), ("ihadamard4x4", J.ihadamard4x4, 16), ("hadamard4x2", J.hadamard4x2, 8),
                        ("ihadamard4x2", J.ihadamard4x2, 8), ("hadamard2x2", J.hadamard2x2, 4), ("ihadamard2x2", J.ihadamard2x2, 4)):
        recs = tq8[name]
        assert len(recs) >= 20, name
        for rec in recs:
            assert fn(rec[:n]).tolist() == rec[n:].tolist(), name


def test_quant8x8_records(tq8):
    """quant_8x8_normal / _around / quant_8x8cavlc_normal / _around as called by the real encoder."""
    recs = tq8["quant8x8"]
    seen = set()
    for r in recs:
        qp, qp_per, arw, variant = (int(v) for v in r[:4])
        q = r[4:196].reshape(64, 3)
        scan, ccost = r[196:324].reshape(64, 2), r[324:388]
        tin, tout = r[388:452], r[452:516]
        lev, run = r[516:584], r[584:652]
        dcost, nz, fadj = int(r[652]), int(r[653]), r[654:718]
        assert scan.tolist() == J.scan8x8(variant >= 2).tolist()            # the scan tables JM hands over are the restated ones
        assert ccost.tolist() == J.coeff_cost8x8(0).tolist()
        tb, l, rn, c, n, fa = J.quant_8x8(tin, q, qp_per, variant, arw)
        assert tb.tolist() == tout.tolist()
        _lists_equal(l, rn, lev, run, variant >= 2)
        assert (c, n) == (dcost, nz)
        if variant & 1:
            assert fa.tolist() == fadj.tolist()
        seen.add(variant)
    assert seen == {0, 1, 2, 3}


def test_quant_dc4x4_records(tq8):
    recs = tq8["quant_dc4x4"]
    assert len(recs) >= 20
    for r in recs:
        qp, qp_per, cavlc = (int(v) for v in r[:3])
        tb, l, rn, nz = J.quant_dc4x4(r[6:22], r[3:6], qp_per, cavlc)
        assert tb.tolist() == r[22:38].tolist()
        n = int(np.argmax(r[38:55] == 0)) + 1
        assert l[:n].tolist() == r[38:38 + n].tolist() and rn[:n - 1].tolist() == r[55:55 + n - 1].tolist()
        assert nz == int(r[72])


def test_residual_transform_quant_luma_8x8_records(tq8):
    """residual_transform_quant_luma_8x8 and _cavlc end to end: prediction + residual in, reconstruction / levels / cost out."""
    recs = tq8["rtq8x8"]
    seen = set()
    for r in recs:
        variant, b8, intra, qp, qp_per, arw, ar_on, maxpel = (int(v) for v in r[:8])
        q = r[8:200].reshape(64, 3)
        pred, ores = r[200:264], r[264:328]
        ret, dcost, rec = int(r[328]), int(r[329]), r[330:394]
        lev, run = r[394:462], r[462:530]
        nz, cost, grec, l, rn, fa, anyr = J.rtq_luma_8x8(pred + ores, pred, q, qp_per, variant, ar_on, arw, maxpel)
        assert (nz, cost) == (ret, dcost)
        assert grec.tolist() == rec.tolist()
        _lists_equal(l, rn, lev, run, variant == 1)
        seen.add((variant, ar_on))
    assert len(seen) >= 3, seen


def unpack_chroma_record(r):
    h = dict(zip(("uv", "cr_cbp", "intra", "yuv", "qp", "qp_per_ac", "qp_per_dc", "cavlc", "arw", "around", "max_pel"), (int(v) for v in r[:11])))
    h["cbp_in"] = (int(r[11]) & 0xffffffff) | (int(r[12]) << 32)                 # signed 64-bit, as JM keeps it
    h["q_ac"], h["q_dc"] = r[13:61].reshape(16, 3), r[61:64]
    h["pred"], h["ores"] = r[64:192], r[192:320]
    h["ret"] = int(r[320]); h["cbp_out"] = (int(r[321]) & 0xffffffff) | (int(r[322]) << 32)
    h["rec"], h["dc_level"], h["dc_run"] = r[323:451], r[451:460], r[460:469]
    ac = r[469:725].reshape(8, 2, 16)
    h["ac_level"], h["ac_run"], h["fadjust"] = ac[:, 0], ac[:, 1], r[725:853]
    return h


def check_chroma_lists(h, dl, dr, al, ar):
    n = int(np.argmax(h["dc_level"] == 0)) + 1
    assert np.asarray(dl)[:n].tolist() == h["dc_level"][:n].tolist() and np.asarray(dr)[:n - 1].tolist() == h["dc_run"][:n - 1].tolist()
    for k in range(4 if h["yuv"] == 1 else 8):
        n = int(np.argmax(h["ac_level"][k] == 0)) + 1
        assert np.asarray(al)[k][:n].tolist() == h["ac_level"][k][:n].tolist(), k
        assert np.asarray(ar)[k][:n - 1].tolist() == h["ac_run"][k][:n - 1].tolist(), k


def test_residual_transform_quant_chroma_records(tq8):
    """residual_transform_quant_chroma_4x4 as the real encoder called it: 4:2:0 and 4:2:2, CAVLC and CABAC, with and without
    adaptive rounding; DC path, AC quantisation, coefficient thresholding, cbp bits, reconstruction."""
    recs = tq8["rtq_chroma"]
    seen = set()
    for r in recs:
        h = unpack_chroma_record(r)
        rows = 64 if h["yuv"] == 1 else 128
        ret, cbp, rec, dl, dr, al, ar, fa = J.rtq_chroma(h["yuv"], h["uv"], h["cr_cbp"], h["cbp_in"], h["q_ac"], h["q_dc"], h["qp_per_ac"], h["qp_per_dc"],
                                                          h["cavlc"], h["around"], h["arw"], h["max_pel"], h["pred"] + h["ores"], h["pred"])
        assert (ret, cbp) == (h["ret"], h["cbp_out"])
        assert rec[:rows].tolist() == h["rec"][:rows].tolist()
        check_chroma_lists(h, dl, dr, al, ar)
        if h["around"]:
            # JM only writes the AC positions; the DC position of every 4x4 keeps whatever an earlier call left there
            mask = np.ones(128, bool); mask[[(4 * (k >> 1)) * 8 + 4 * (k & 1) for k in range(8)]] = False
            assert fa[:rows][mask[:rows]].tolist() == h["fadjust"][:rows][mask[:rows]].tolist()
        seen.add((h["yuv"], h["cavlc"], h["around"], h["ret"]))
    assert len({s[:3] for s in seen}) >= 3 and {s[3] for s in seen} == {0, 1, 2}, seen


def test_reconstruct_records(fs):
    for r in fs["recon4x4"]:
        maxv, dq = int(r[0]), int(r[1])
        pred, rres, out = r[2:18], r[18:34], r[34:50]
        got = np.clip(((rres + (1 << (dq - 1))) >> dq) + pred, 0, maxv)
        assert got.tolist() == out.tolist()


def test_sub_images_luma_digests(fs):
    for k in (0, 1):
        ref = J.RefPic(fs[f"ref{k}_src"])
        planes = ref.planes_u8()
        sha = [hashlib.sha256(planes[i].tobytes()).hexdigest() for i in range(16)]
        assert sha == fs[f"ref{k}_sha"].tolist()
    ref = J.RefPic(fs["ref0_src"])
    assert (ref.planes_u8()[:, ::23] == fs["ref0_plane_rows"]).all()


def test_full_search_records(fs):
    ref = J.RefPic(fs["ref0_src"])
    cur = fs["cur1"]
    recs = fs["me_fs"]
    assert len(recs) == 99 * 41
    for r in recs[::3]:
        (_, refidx, bt, px, py, bsx, bsy, pdx, pdy, cx, cy, R, lam, mc_in, ox, oy, cost) = (int(v) for v in r)
        mv, c, _ = J.full_search(ref, cur, px, py, bsx, bsy, (pdx, pdy), (cx, cy), R, lam, mc_in)
        assert (mv, c) == ((ox, oy), cost)


def test_sub_pel_records(fs):
    ref = J.RefPic(fs["ref0_src"])
    cur = fs["cur1"]
    for r in fs["me_subpel"][::3]:
        (_, refidx, bt, px, py, bsx, bsy, pdx, pdy, mx, my, lh, lq, mh, mq, shp, sqp, t8, mc_in, ox, oy,
         cost) = (int(v) for v in r)
        mv, c = J.sub_pel_search(ref, cur, px, py, bsx, bsy, (pdx, pdy), (mx, my), lh, lq, mh, mq, shp, sqp, t8, mc_in)
        assert (mv, c) == ((ox, oy), cost)


def test_fast_full_search_tables_and_argmin(ffs):
    ref = J.RefPic(ffs["ref0_src"])
    cur = ffs["cur1"]
    setups = ffs["ffs_setup"]
    for si, key in ((0, "ffs_table0"), (7, "ffs_table7")):
        _, refidx, mbx, mby, cx, cy, R, max_pos, _ = (int(v) for v in setups[si])
        tab = J.ffs_setup(ref, cur, mbx, mby, (cx, cy), R)
        assert (tab[1:8].astype(np.uint16) == ffs[key]).all()
    for s in setups:                                   # the rest by digest
        _, refidx, mbx, mby, cx, cy, R, max_pos, dig = (int(v) for v in s)
        tab = J.ffs_setup(ref, cur, mbx, mby, (cx, cy), R)
        assert int(hashlib.sha256(tab[1:8].astype(np.uint16).tobytes()).hexdigest()[:12], 16) == dig
    recs = ffs["me_ffs"]
    cache = {}
    for r in recs[::5]:
        (_, refidx, bt, mbx, mby, bx, by, pdx, pdy, cx, cy, R, Rtab, lam, max_mvd, mc_in, ox, oy, cost) = (int(v) for v in r)
        key = (mbx, mby, cx, cy)
        if key not in cache:
            cache[key] = J.ffs_setup(ref, cur, mbx, mby, (cx, cy), Rtab)
        mv, c = J.ffs_search(cache[key], bt, by * 4 + bx, (cx, cy), (pdx, pdy), R, lam, max_mvd, mc_in)
        assert (mv, c) == ((ox, oy), cost)


@pytest.mark.parametrize("name,frames", [("qcif_fs.npz", (0, 1)), ("qcif_422.npz", (0, 1)), ("qcif_main.npz", (0, 1, 2))])
def test_deblock_frames(name, frames):
    d = np.load(os.path.join(G, name))
    for i in frames:
        p = f"db{i}_"
        w, h, fmt, maxy, maxc, d8 = (int(v) for v in d[p + "hdr"])
        y, u, v = J.deblock_frame(d[p + "pre_y"], d[p + "pre_u"], d[p + "pre_v"], fmt, d[p + "mbs"], d[p + "mot"],
                                  maxy, maxc, d8)
        assert (y == d[p + "post_y"]).all(), f"{name} frame {i} luma"
        assert (u == d[p + "post_u"]).all() and (v == d[p + "post_v"]).all(), f"{name} frame {i} chroma"
        assert not (d[p + "pre_y"] == d[p + "post_y"]).all()   # the filter did something


# ---------------------------------------------------------------- motion-compensated prediction (tests/golden/qcif_mc.npz)
def mc_golden():
    return np.load(os.path.join(G, "qcif_mc.npz"))


def mc_luma_records(g, tag):
    """(x, y, bsx, bsy, p_dir, ref0, mv0, ref1, mv1, weights or None, expected (bsy, bsx)) of the luma_prediction records;
    weights = (w0, w1, offset, round, shift) as the reference handed them to weighted_mc_prediction / weighted_bi_prediction"""
    for h, px in zip(g[tag + "_mcl_hdr"], g[tag + "_mcl_pix"]):
        frame, x, y, bsx, bsy, p_dir, wp, r0, m0x, m0y, r1, m1x, m1y = [int(v) for v in h[:13]]
        yield x, y, bsx, bsy, p_dir, r0, (m0x, m0y), r1, (m1x, m1y), tuple(int(v) for v in h[13:18]) if wp else None, px[: bsx * bsy].reshape(bsy, bsx)


def mc_chroma_records(g, tag):
    for h, px in zip(g[tag + "_mcc_hdr"], g[tag + "_mcc_pix"]):
        frame, yuv, uv, xc, yc, p_dir, wp, buffered = [int(v) for v in h[:8]]
        if not buffered:
            continue
        r0, r1 = int(h[8]), int(h[26])
        mv0, mv1 = h[10:26].reshape(4, 2, 2), h[28:44].reshape(4, 2, 2)
        yield yuv, uv, xc, yc, p_dir, r0, mv0, r1, mv1, tuple(int(v) for v in h[44:49]) if wp else None, px.reshape(4, 4)


@pytest.mark.parametrize("tag", ["a", "c", "e", "w"])
def test_oracle_luma_prediction_matches_the_reference(tag):
    """jmo_luma_pred (+ jmo_weighted_samples) == luma_prediction (mc_prediction.c:144) on the real encoder's calls: 4:2:0 P, 4:2:2 P,
    B picture (bi-prediction), and explicit weighted prediction in P and B pictures (run w)"""
    g = mc_golden()
    refs = {}
    n = nw = 0
    for x, y, bsx, bsy, p_dir, r0, mv0, r1, mv1, wts, want in mc_luma_records(g, tag):
        for r in (r0, r1):
            if r >= 0 and r not in refs:
                refs[r] = J.RefPic(g[f"{tag}_ref{r}_y"])
        if wts is None:
            got = J.luma_pred(refs.get(r0), refs.get(r1), p_dir, x, y, bsx, bsy, mv0, mv1)
        else:
            got = J.luma_pred_wp(refs.get(r0), refs.get(r1), p_dir, x, y, bsx, bsy, mv0, mv1, wts)
            nw += 1
        assert np.array_equal(got, want), (tag, x, y, bsx, bsy, p_dir, mv0, mv1, wts)
        n += 1
    assert n > 300 and (nw > 300) == (tag == "w")


@pytest.mark.parametrize("tag", ["a", "c", "e", "w"])
def test_oracle_chroma_prediction_matches_the_reference(tag):
    """jmo_chroma_pred4x4 (+ jmo_weighted_samples) == chroma_prediction_4x4 (mc_prediction.c:568, buffered chroma sub-images) on the
    real encoder's calls"""
    g = mc_golden()
    n = nw = 0
    for yuv, uv, xc, yc, p_dir, r0, mv0, r1, mv1, wts, want in mc_chroma_records(g, tag):
        pl = "uv"[uv]
        p0 = g[f"{tag}_ref{r0}_{pl}"] if r0 >= 0 else None
        p1 = g[f"{tag}_ref{r1}_{pl}"] if r1 >= 0 else None
        if wts is None:
            got = J.chroma_pred4x4(p0, p1, yuv, p_dir, xc, yc, mv0, mv1)
        else:
            got = J.chroma_pred4x4_wp(p0, p1, yuv, p_dir, xc, yc, mv0, mv1, wts)
            nw += 1
        assert np.array_equal(got, want), (tag, yuv, uv, xc, yc, p_dir, mv0.tolist(), mv1.tolist(), wts, got.tolist(), want.tolist())
        n += 1
    assert n > 100 and (nw > 100) == (tag == "w")


# ---------------------------------------------------------------- Intra16x16 luma: residual_transform_quant_luma_16x16
def unpack_i16_record(r):
    """fields of one rtq16x16 record (oracle/ref_tap.c)"""
    d = dict(qp=int(r[1]), qp_per=int(r[2]), cavlc=int(r[3]), around=int(r[4]), arw=int(r[5]), max_pel=int(r[6]), mode=int(r[7]))
    d["q"], d["orig"], d["pred"] = r[8:56].reshape(16, 3), r[56:312].reshape(16, 16), r[312:568].reshape(16, 16)
    d["ret"], d["dc_level"], d["dc_run"] = int(r[568]), r[569:586], r[586:603]
    ac = r[603:1115].reshape(16, 2, 16)
    d["ac_level"], d["ac_run"] = ac[:, 0], ac[:, 1]
    d["rec"], d["fadjust"] = r[1115:1371].reshape(16, 16), r[1371:1435].reshape(4, 16)
    return d


def check_level_lists(want_level, want_run, got_level, got_run, what):
    """JM's lists end at the first zero level; entries behind it are stale"""
    n = 0
    while want_level[n] != 0:
        n += 1
    assert np.array_equal(np.asarray(got_level)[: n + 1], want_level[: n + 1]), (what, "levels")
    assert np.array_equal(np.asarray(got_run)[:n], want_run[:n]), (what, "runs")


I16_AC_MASK = np.ones((4, 16), bool)
I16_AC_MASK[0, ::4] = False                      # the DC position of each block is never written by quant_ac4x4


def test_oracle_rtq_luma_16x16_matches_the_reference(tq8):
    """jmo_rtq_luma_16x16 == residual_transform_quant_luma_16x16 (block.c:208) on the real encoder's calls: CAVLC / CABAC, with and
    without adaptive rounding (incl. JM's un-offset fadjust rows), macroblocks with and without AC levels"""
    recs = tq8["rtq16x16"]
    assert len(recs) > 80
    for k, r in enumerate(recs):
        d = unpack_i16_record(r)
        ret, dl, dr, al, ar, rec, fadj = J.rtq_luma_16x16(d["orig"], d["pred"], d["q"], d["qp_per"], d["cavlc"], d["around"], d["arw"], d["max_pel"])
        assert ret == d["ret"], k
        check_level_lists(d["dc_level"], d["dc_run"], dl, dr, (k, "dc"))
        for b in range(16):
            check_level_lists(d["ac_level"][b], d["ac_run"][b], al[b], ar[b], (k, "ac", b))
        assert np.array_equal(rec, d["rec"]), k
        if d["around"]:
            assert np.array_equal(fadj[I16_AC_MASK], d["fadjust"][I16_AC_MASK]), k


# ---------------------------------------------------------------- luma intra prediction, Intra16x16 mode search (tests/golden/qcif_intra.npz)
def intra_golden():
    return np.load(os.path.join(G, "qcif_intra.npz"))


@pytest.mark.parametrize("tag", ["a", "c", "e"])
def test_oracle_intra_prediction_matches_the_reference(tag):
    """jmo_intrapred_4x4 == get_intrapred_4x4 (intra4x4.c:521, all nine modes, every availability case) and jmo_intra16_search ==
    find_sad_16x16_JM (intra16x16.c:463: the four predictions, the SATD mode cost, the chosen mode) on the real encoder's calls"""
    g = intra_golden()
    i4 = g[tag + "_i4"]
    assert len(i4) > 300 and set(np.unique(i4[:, 0])) == set(range(9))
    for r in i4:
        assert np.array_equal(J.intrapred_4x4(r[4:17], r[0], r[1], r[2]).reshape(-1), r[17:33]), r.tolist()
    hdr = g[tag + "_i16_hdr"]
    for k in range(len(hdr)):
        left, up, allav, mask, metric, maxp = [int(v) for v in hdr[k]]
        cost, mode, pred = J.intra16_search(g[tag + "_i16_edge"][k], left, up, mask, metric, g[tag + "_i16_orig"][k], maxp)
        assert cost == int(g[tag + "_i16_cost"][k]) and mode == int(g[tag + "_i16_mode"][k]), k
        for m in range(4):
            if (mask >> m) & 1:
                assert np.array_equal(pred[m].reshape(-1), g[tag + "_i16_pred"][k, m]), (k, m)
    assert len(hdr) >= 20


@pytest.mark.parametrize("tag,fmt", [("a", 1), ("c", 2), ("e", 1)])
def test_oracle_chroma_subimages_match_the_reference(tag, fmt):
    """jmo_sub_images_chroma == getSubImagesChroma (img_chroma.c:338): every sub-image of both planes of the first reference picture,
    padding included (digests), and every 13th row of the U sub-images sample by sample"""
    g = mc_golden()
    for pl, name in enumerate("uv"):
        sub = J.sub_images_chroma(g[f"{tag}_ref0_{name}"], fmt)
        sha = [hashlib.sha256(sub[j, i].tobytes()).hexdigest() for j in range(sub.shape[0]) for i in range(8)]
        assert sha == list(g[tag + "_csub_sha"][pl]), (tag, name)
        if pl == 0:
            assert np.array_equal(sub.reshape(-1, *sub.shape[2:])[:, ::13], g[tag + "_csub_u_rows"])


def pred_dist_golden():
    """tests/golden/pred_dist.npz (make_pred_dist.py): the REAL reference's compute{,BiPred}{SAD,SSE,SATD}{,WP,1,2} called on seeded
    blocks / candidates / weights.  Returns (cur, ref1, ref2, records as dict rows)."""
    g = np.load(os.path.join(G, "pred_dist.npz"))
    cols = [str(c) for c in g["columns"]]
    return g["cur"], g["ref1"], g["ref2"], [dict(zip(cols, (int(v) for v in r))) for r in g["records"]]


def pred_dist_weights(r):
    """(w1, w2, offset, round, shift) as the reference's formula uses them: BiPred*2 doubles wp_luma_round and adds 1 to the
    denominator (me_distortion.c:636-637), *WP takes them as they are (:452-455)"""
    if r["pred"] == J.PRED_BI_WP:
        return (r["w1"], r["w2"], r["offset"], 2 * r["wp_round"], r["log_denom"] + 1)
    return (r["w1"], 0, r["offset"], r["wp_round"], r["log_denom"])


def test_oracle_weighted_and_bipred_distortions_match_the_reference():
    cur, r1, r2, recs = pred_dist_golden()
    p1, p2 = J.RefPic(r1), J.RefPic(r2)
    seen = set()
    for k, r in enumerate(recs):
        orig = cur[r["pos_y"]:r["pos_y"] + r["bsy"], r["pos_x"]:r["pos_x"] + r["bsx"]]
        c1 = (4 * r["pos_x"] + r["c1x"], 4 * r["pos_y"] + r["c1y"])
        c2 = (4 * r["pos_x"] + r["c2x"], 4 * r["pos_y"] + r["c2y"])
        for thr, want in ((r["min_mcost"], r["result"]), (J.DIST_MAX, r["full_result"])):
            got = J.pred_dist(p1, p2, orig, r["bsx"], r["bsy"], r["test8x8"], r["metric"], r["pred"], pred_dist_weights(r), thr, c1, c2)
            assert got == want, (k, r, got)
        seen.add((r["pred"], r["metric"], r["test8x8"]))
    assert len(seen) == 16                      # 4 prediction kinds x {SAD, SSE, SATD 4x4, SATD 8x8}
    # the plain kind agrees with the functions the rest of the oracle uses
    for r in recs:
        if r["pred"] == J.PRED_UNI and r["metric"] != 1:
            orig = np.ascontiguousarray(cur[r["pos_y"]:r["pos_y"] + r["bsy"], r["pos_x"]:r["pos_x"] + r["bsx"]], np.uint16)
            c1 = (4 * r["pos_x"] + r["c1x"], 4 * r["pos_y"] + r["c1y"])
            if r["metric"] == 0:
                got = J.L.jmo_compute_sad(p1.ptr(), J._p(orig), r["bsx"], r["bsy"], r["min_mcost"], c1[0], c1[1])
            else:
                got = J.L.jmo_compute_satd(p1.ptr(), J._p(orig), r["bsx"], r["bsy"], r["test8x8"], r["min_mcost"], c1[0], c1[1])
            assert got == r["result"], r


def test_oracle_source_frame_padding_matches_the_reference():
    """jmo_load_frame == read_one_frame + pad_borders (lcommon/src/input.c:792, :880): digests of the planes the reference encoder held for
    a 168x136 source coded as 176x144 (right and bottom padding; tests/golden/qcif_pad.npz) and for the 1080p clip of configs[1] coded as
    1920x1088 (bottom padding; tests/golden/g2_sideinfo.npz, clip regenerated by bench.write_yuv)"""
    g = np.load(os.path.join(G, "qcif_pad.npz"))
    sw, sh, W, H, fmt, idx = (int(v) for v in g["geometry"])
    raw = open(os.path.join(G, "foreman_part_qcif.yuv"), "rb").read()
    fs = sw * sh * 3 // 2
    y, u, v = J.load_frame(raw[idx * fs:(idx + 1) * fs], sw, sh, W, H, fmt)
    assert [hashlib.sha256(p.tobytes()).hexdigest() for p in (y, u, v)] == [str(s_) for s_ in g["sha"]]
    assert np.array_equal(y[130:, 160:], g["y_tail"]) and np.array_equal(u[64:, 80:], g["u_tail"])
    assert (y[:, 168:] == y[:, 167:168]).all() and (y[136:] == y[135]).all()
    import tempfile
    import bench
    with tempfile.TemporaryDirectory() as t:
        bench.write_yuv(os.path.join(t, "c.yuv"), 2)
        raw = open(os.path.join(t, "c.yuv"), "rb").read()
    fs = 1920 * 1080 * 3 // 2
    y, u, v = J.load_frame(raw[fs:2 * fs], 1920, 1080, 1920, 1088, 1)
    want = np.load(os.path.join(G, "g2_sideinfo.npz"))["p_cur_yuv_sha"]
    assert [hashlib.sha256(p.tobytes()).hexdigest() for p in (y, u, v)] == [str(s_) for s_ in want]


def test_oracle_general_frame_reader_matches_the_reference():
    """jmo_load_frame_ex == read_one_frame's buf2img calls + pad_borders on the REAL reference's outputs (tests/golden/load_frame.npz, made by calling buf2img_basic /
    buf2img_bitshift / pad_borders of the unmodified lencod objects: tests/golden/make_load_frame.py): 4:0:0 .. 4:4:4, 8 .. 14 bit in one or two bytes, depth conversion up and
    down, padded sizes, a file frame larger / smaller than the picture -- and the sheared planes JM makes of two-byte pictures whose width is not a multiple of 16."""
    g = np.load(os.path.join(G, "load_frame.npz"))
    assert len(g["cases"]) >= 12
    for k, (yuv, sw, sh, ow, oh, sb, sd, od) in enumerate(g["cases"]):
        y, u, v = J.load_frame_ex(g[f"raw{k}"], yuv, sw, sh, ow, oh, sb, sd, od)
        assert np.array_equal(y, g[f"y{k}"]), k
        if yuv:
            assert np.array_equal(u, g[f"u{k}"]) and np.array_equal(v, g[f"v{k}"]), k
    # the 8-bit reader of the pipeline is the special case
    k = [i for i, c in enumerate(g["cases"]) if tuple(c[:5]) == (1, 176, 144, 176, 144)][0]
    y8, u8, v8 = J.load_frame(g[f"raw{k}"], 176, 144, 176, 144, 1)
    assert np.array_equal(y8, g[f"y{k}"]) and np.array_equal(u8, g[f"u{k}"]) and np.array_equal(v8, g[f"v{k}"])


@pytest.mark.parametrize("tag", ["a", "c", "e"])
def test_oracle_intra_chroma_prediction_matches_the_reference(tag):
    """jmo_intra_chroma_pred == intra_chroma_prediction (intra_chroma.c:530): DC / horizontal / vertical / plane of both planes on the
    real encoder's calls, 4:2:0 (8x8) and 4:2:2 (8x16), every neighbour-availability combination the runs reach"""
    g = np.load(os.path.join(G, "qcif_intra.npz"))
    hdr, edge, pred = g[tag + "_ic_hdr"], g[tag + "_ic_edge"], g[tag + "_ic_pred"]
    combos = set()
    for h, e, p in zip(hdr, edge, pred):
        yuv, up, left, ul = (int(v) for v in h)
        ch = 8 if yuv == 1 else 16
        for uv in range(2):
            mask, got = J.intra_chroma_pred(e[uv][:8], e[uv][8:24], e[uv][24], up, left, ul, ch)
            assert mask == (1 | (2 if left else 0) | (4 if up else 0) | (8 if (up and left and ul) else 0))
            assert np.array_equal(got, p[uv][:, :ch]), (tag, h.tolist(), uv)
        combos.add((up, left, ul))
    assert len(hdr) > 30 and len(combos) >= 4


def test_oracle_intra8x8_prediction_matches_the_reference():
    """jmo_intrapred_8x8 == get_intrapred_8x8 (intra8x8.c:716) on the real encoder's calls (High 4:2:2 run with the 8x8 transform): all nine modes"""
    g = np.load(os.path.join(G, "qcif_intra.npz"))
    rec = g["c_i8"]
    for r in rec:
        mode, left, up = (int(v) for v in r[:3])
        assert np.array_equal(J.intrapred_8x8(r[3:28], mode, left, up), r[28:].reshape(8, 8).astype(np.uint8)), (mode, left, up, r[3:28].tolist())
    assert set(rec[:, 0].tolist()) == set(range(9)) and len(rec) > 300


def test_direct_spatial_equals_the_reference_encoder():
    """B slices, first building block: the spatial direct mode's references, prediction directions and vectors of every macroblock of the B slices of four
    runs of the REAL encoder (tests/golden/direct_b.npz, oracle/ref_tap_mb.c tap_b_slice: foreman QCIF, motion clips, one or two B pictures between the P pictures, up to five
    references, direct_8x8_inference on and off, slices) from the neighbours and the co-located motion the encoder itself read."""
    z = np.load(os.path.join(G, "direct_b.npz"))
    tags = [k for k in z.files if not k.endswith("_overrides")]
    assert len(tags) >= 4
    n = 0
    for tag in tags:
        recs = z[tag]
        assert (recs["weighted_bipred_idc"] == 0).all()
        for q in recs:
            ro, po, mo = J.direct_spatial(q["nb_avail"], q["nb_ref"], q["nb_mv"], q["col_long_term"], q["col_ref"], q["col_mv"])
            assert np.array_equal(ro, q["direct_ref_idx"]) and np.array_equal(po, q["direct_pdir"]), (tag, int(q["frame_no"]), int(q["mb_addr"]))
            used = q["direct_ref_idx"] >= 0                      # a list that is not used keeps whatever all_mv held
            assert np.array_equal(mo[used], q["direct_mv"][used]), (tag, int(q["frame_no"]), int(q["mb_addr"]), "vectors")
            n += 1
    assert n >= 1000
    # both co-located rules and every prediction direction occur
    assert {int(v) for t in tags for v in np.unique(z[t]["direct_8x8_inference"])} == {0, 1}
    assert {int(v) for t in tags for v in np.unique(z[t]["direct_pdir"])} == {0, 1, 2}
