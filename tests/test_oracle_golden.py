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
