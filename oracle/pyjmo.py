"""ctypes bindings for the CPU parity oracle (oracle/libjmo.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by the product package jm_amd.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PAD_X, PAD_Y = 32, 20
DIST_MAX = (2 ** 31 - 1) << 5


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libjmo.so"])


def _load():
    so = os.path.join(_HERE, "libjmo.so")
    if not os.path.exists(so):
        build()
    return C.CDLL(so)


L = _load()


class MV(C.Structure):
    _fields_ = [("x", C.c_int16), ("y", C.c_int16)]


class RefPicS(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("pitch", C.c_int),
                ("sub", (C.POINTER(C.c_uint16) * 4) * 4)]


class FsJob(C.Structure):
    _fields_ = [("pos_x", C.c_int), ("pos_y", C.c_int), ("bsx", C.c_int), ("bsy", C.c_int),
                ("pred", MV), ("center", MV), ("search_range", C.c_int), ("lambda_factor", C.c_int),
                ("min_mcost", C.c_int64)]


class SubpelJob(C.Structure):
    _fields_ = [("pos_x", C.c_int), ("pos_y", C.c_int), ("bsx", C.c_int), ("bsy", C.c_int),
                ("pred", MV), ("mv", MV), ("lambda_h", C.c_int), ("lambda_q", C.c_int),
                ("metric_h", C.c_int), ("metric_q", C.c_int), ("start_hp", C.c_int), ("start_qp", C.c_int),
                ("test8x8", C.c_int), ("min_mcost", C.c_int64)]


class QParam(C.Structure):
    _fields_ = [("OffsetComp", C.c_int), ("ScaleComp", C.c_int), ("InvScaleComp", C.c_int)]


class DbMb(C.Structure):
    _fields_ = [("mb_type", C.c_int16), ("slice_type", C.c_int16), ("qp", C.c_int16), ("qpc", C.c_int16 * 2),
                ("cbp", C.c_int16), ("cbp_blk", C.c_uint32), ("slice_nr", C.c_int16),
                ("df_disable_idc", C.c_int16), ("df_alpha_c0", C.c_int16), ("df_beta", C.c_int16),
                ("transform8x8", C.c_int16), ("pad_", C.c_int16)]


class DbMotion(C.Structure):
    _fields_ = [("mv", (C.c_int16 * 2) * 2), ("ref_id", C.c_int32 * 2)]


class WP(C.Structure):
    _fields_ = [("weight", C.c_int * 2), ("offset", C.c_int), ("round", C.c_int), ("shift", C.c_int)]


L.jmo_compute_pred_dist.restype = C.c_int64
L.jmo_compute_pred_dist.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
L.jmo_weighted_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
L.jmo_load_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
L.jmo_full_search.restype = C.c_int64
L.jmo_compute_sad.restype = C.c_int64
L.jmo_compute_satd.restype = C.c_int64
L.jmo_ffs_search.restype = C.c_int64
L.jmo_sub_pel_search.restype = C.c_int64
L.jmo_compute_sad.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int]
L.jmo_compute_satd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int]
L.jmo_sub_images_luma.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_long]
L.jmo_ffs_search.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, MV, MV, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p]
L.jmo_ffs_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, MV, C.c_int, C.c_void_p]
L.jmo_luma_pred.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, MV, MV, C.c_void_p]
L.jmo_chroma_pred4x4.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RefPic:
    """A reference picture with its 16 quarter-pel planes (getSubImagesLuma)."""

    def __init__(self, luma, max_pel=255):
        luma = np.ascontiguousarray(luma, dtype=np.uint16)
        self.h, self.w = luma.shape
        self.pitch = self.w + 2 * PAD_X
        self.hp = self.h + 2 * PAD_Y
        self.planes = np.zeros((16, self.hp, self.pitch), np.uint16)
        L.jmo_sub_images_luma(_p(luma), self.w, self.w, self.h, max_pel, _p(self.planes), self.pitch,
                              self.hp * self.pitch)
        self.s = RefPicS(self.w, self.h, self.pitch)
        base = self.planes.ctypes.data
        for j in range(4):
            for i in range(4):
                off = ((j * 4 + i) * self.hp * self.pitch + PAD_Y * self.pitch + PAD_X) * 2
                self.s.sub[j][i] = C.cast(base + off, C.POINTER(C.c_uint16))

    def ptr(self):
        return C.byref(self.s)

    def planes_u8(self):
        return self.planes.astype(np.uint8)


def mvbits(d):
    """mvbits LUT of lencod/src/mv_search.c:366-374 (jmo_mvbits)."""
    return int(L.jmo_mvbits(int(d)))


def block_of(cur, x, y, w, h):
    return np.ascontiguousarray(cur[y:y + h, x:x + w], dtype=np.uint16)


def full_search(ref, cur, pos_x, pos_y, bsx, bsy, pred, center, R, lam, min_mcost=DIST_MAX):
    job = FsJob(pos_x, pos_y, bsx, bsy, MV(*pred), MV(*center), R, lam, min_mcost)
    orig = block_of(cur, pos_x, pos_y, bsx, bsy)
    evals = C.c_long(0)
    cost = L.jmo_full_search(ref.ptr(), _p(orig), C.byref(job), C.byref(evals))
    return (job.center.x, job.center.y), cost, evals.value


def sub_pel_search(ref, cur, pos_x, pos_y, bsx, bsy, pred, mv, lam_h, lam_q, metric_h, metric_q,
                   start_hp, start_qp, test8x8, min_mcost):
    job = SubpelJob(pos_x, pos_y, bsx, bsy, MV(*pred), MV(*mv), lam_h, lam_q, metric_h, metric_q,
                    start_hp, start_qp, test8x8, min_mcost)
    orig = block_of(cur, pos_x, pos_y, bsx, bsy)
    cost = L.jmo_sub_pel_search(ref.ptr(), _p(orig), C.byref(job))
    return (job.mv.x, job.mv.y), cost


def ffs_setup(ref, cur, mb_x, mb_y, center, R):
    max_pos = (2 * R + 1) ** 2
    tab = np.zeros((8, 16, max_pos), np.uint32)
    mb = block_of(cur, mb_x, mb_y, 16, 16)
    L.jmo_ffs_setup(ref.ptr(), _p(mb), mb_x, mb_y, MV(*center), R, _p(tab))
    return tab


def ffs_search(tab, blocktype, block_index, center, pred, R, lam, max_mvd, min_mcost=DIST_MAX):
    best = MV(0, 0)
    cost = L.jmo_ffs_search(_p(tab), tab.shape[2], blocktype, block_index, MV(*center), MV(*pred), R, lam,
                            max_mvd, min_mcost, C.byref(best))
    return (best.x, best.y), cost


def spiral(R):
    n = max(9, (2 * R + 1) ** 2)
    a = np.zeros((n, 2), np.int16)
    L.jmo_spiral(R, _p(a))
    return a


def forward4x4(x):
    x = np.ascontiguousarray(x, np.int32); o = np.zeros(16, np.int32); L.jmo_forward4x4(_p(x), _p(o)); return o


def inverse4x4(x):
    x = np.ascontiguousarray(x, np.int32); o = np.zeros(16, np.int32); L.jmo_inverse4x4(_p(x), _p(o)); return o


def forward8x8(x):
    x = np.ascontiguousarray(x, np.int32); o = np.zeros(64, np.int32); L.jmo_forward8x8(_p(x), _p(o)); return o


def inverse8x8(x):
    x = np.ascontiguousarray(x, np.int32); o = np.zeros(64, np.int32); L.jmo_inverse8x8(_p(x), _p(o)); return o


def scan4x4():
    return np.ctypeslib.as_array((C.c_uint8 * 32).in_dll(L, "JMO_SNGL_SCAN")).copy()


def coeff_cost4x4(k=0):
    return np.ctypeslib.as_array((C.c_uint8 * 48).in_dll(L, "JMO_COEFF_COST4x4")).reshape(3, 16)[k].copy()


def quant_4x4(tblock, qparams, qp_per, cavlc, around=False, arw=0):
    """qparams: (16,3) int32 [OffsetComp, ScaleComp, InvScaleComp] at [j*4+i]."""
    tb = np.ascontiguousarray(tblock, np.int32).copy()
    q = np.ascontiguousarray(qparams, np.int32)
    level = np.zeros(17, np.int32); run = np.zeros(17, np.int32); cost = C.c_int(0)
    fadj = np.zeros(16, np.int32)
    sc, cc = scan4x4(), coeff_cost4x4(0)
    if around:
        nz = L.jmo_quant_4x4_around(_p(tb), _p(q), qp_per, cavlc, _p(sc), _p(cc), arw, _p(level), _p(run),
                                    C.byref(cost), _p(fadj))
    else:
        nz = L.jmo_quant_4x4_normal(_p(tb), _p(q), qp_per, cavlc, _p(sc), _p(cc), _p(level), _p(run), C.byref(cost))
    return tb, level, run, cost.value, nz, fadj


def _xf(fn, x, n):
    x = np.ascontiguousarray(x, np.int32); o = np.zeros(n, np.int32); fn(_p(x), _p(o)); return o


def hadamard4x4(x): return _xf(L.jmo_hadamard4x4, x, 16)
def ihadamard4x4(x): return _xf(L.jmo_ihadamard4x4, x, 16)
def hadamard4x2(x): return _xf(L.jmo_hadamard4x2, x, 8)
def ihadamard4x2(x): return _xf(L.jmo_ihadamard4x2, x, 8)
def hadamard2x2(x): return _xf(L.jmo_hadamard2x2, x, 4)
def ihadamard2x2(x): return _xf(L.jmo_ihadamard2x2, x, 4)


def scan8x8(cavlc=False):
    if cavlc:
        o = np.zeros((64, 2), np.uint8); L.jmo_scan8x8_cavlc(_p(o)); return o
    return np.ctypeslib.as_array((C.c_uint8 * 128).in_dll(L, "JMO_SNGL_SCAN8x8")).reshape(64, 2).copy()


def coeff_cost8x8(k=0):
    return np.ctypeslib.as_array((C.c_uint8 * 128).in_dll(L, "JMO_COEFF_COST8x8")).reshape(2, 64)[k].copy()


def quant_8x8(tblock, qparams, qp_per, variant, arw=0):
    """variant 0 normal, 1 around, 2 cavlc_normal, 3 cavlc_around; qparams (64,3) at [j*8+i].
    Returns (dequantised block, level[68], run[68], coeff_cost, nonzero, fadjust[64])."""
    tb = np.ascontiguousarray(tblock, np.int32).copy()
    q = np.ascontiguousarray(qparams, np.int32)
    level = np.zeros(68, np.int32); run = np.zeros(68, np.int32); cost = C.c_int(0); fadj = np.zeros(64, np.int32)
    sc, cc = scan8x8(variant >= 2), coeff_cost8x8(0)
    nz = L.jmo_quant_8x8(_p(tb), _p(q), qp_per, variant, _p(sc), _p(cc), arw, _p(level), _p(run), C.byref(cost), _p(fadj))
    return tb, level, run, cost.value, nz, fadj


def quant_dc4x4(tblock, qparam, qp_per, cavlc):
    tb = np.ascontiguousarray(tblock, np.int32).copy()
    q = np.ascontiguousarray(qparam, np.int32)
    level = np.zeros(17, np.int32); run = np.zeros(17, np.int32)
    nz = L.jmo_quant_dc4x4_normal(_p(tb), _p(q), qp_per, cavlc, _p(level), _p(run))
    return tb, level, run, nz


def rtq_luma_8x8(orig, pred, qparams, qp_per, cavlc, around, arw, max_pel=255):
    """Returns (nonzero, coeff_cost, rec[64], level[68], run[68], fadjust[64], any_residual)."""
    o = np.ascontiguousarray(orig, np.uint16); p_ = np.ascontiguousarray(pred, np.uint16)
    q = np.ascontiguousarray(qparams, np.int32)
    level = np.zeros(68, np.int32); run = np.zeros(68, np.int32); cost = C.c_int(0)
    rec = np.zeros(64, np.uint16); fadj = np.zeros(64, np.int32); anyr = C.c_int(0)
    nz = L.jmo_rtq_luma_8x8(_p(o), _p(p_), _p(q), qp_per, cavlc, around, arw, max_pel, _p(level), _p(run), C.byref(cost),
                            _p(rec), _p(fadj), C.byref(anyr))
    return nz, cost.value, rec, level, run, fadj, anyr.value


def rtq_chroma(yuv, uv, cr_cbp, cbp_blk, q_ac, q_dc, qp_per_ac, qp_per_dc, cavlc, around, arw, max_pel, orig, pred):
    """residual_transform_quant_chroma_4x4 for one plane of one macroblock; orig/pred: 128 samples (rows of 8).
    Returns (cr_cbp, cbp_blk, rec[128], dc_level[9], dc_run[9], ac_level[8,16], ac_run[8,16], fadjust[128])."""
    o = np.ascontiguousarray(orig, np.uint16); p_ = np.ascontiguousarray(pred, np.uint16)
    qa = np.ascontiguousarray(q_ac, np.int32).reshape(16, 3); qd = np.ascontiguousarray(q_dc, np.int32).reshape(3)
    cb = C.c_int64(int(cbp_blk)); rec = np.zeros(128, np.uint16)
    dl = np.zeros(9, np.int32); dr = np.zeros(9, np.int32); al = np.zeros((8, 16), np.int32); ar = np.zeros((8, 16), np.int32)
    fa = np.zeros(128, np.int32)
    r = L.jmo_rtq_chroma(yuv, uv, cr_cbp, C.byref(cb), _p(qa), _p(qd), qp_per_ac, qp_per_dc, cavlc, around, arw, max_pel,
                         _p(o), _p(p_), _p(rec), _p(dl), _p(dr), _p(al), _p(ar), _p(fa))
    return r, cb.value, rec, dl, dr, al, ar, fa


def qparams_4x4(qp, intra, offset):
    """offset: one value for all sixteen positions, or sixteen values (a q_offset.cfg list)"""
    q = np.zeros((16, 3), np.int32)
    if np.ndim(offset):
        o = np.ascontiguousarray(offset, np.int16)
        L.jmo_qparams_4x4_m(qp, _p(o), _p(q))
    else:
        L.jmo_qparams_4x4(qp, intra, offset, _p(q))
    return q


def qparams_8x8(qp, intra, offset):
    q = np.zeros((64, 3), np.int32)
    if np.ndim(offset):
        o = np.ascontiguousarray(offset, np.int16)
        L.jmo_qparams_8x8_m(qp, _p(o), _p(q))
    else:
        L.jmo_qparams_8x8(qp, intra, offset, _p(q))
    return q


def deblock_frame(y, u, v, fmt, mbs, mot, maxy=255, maxc=255, d8=1):
    """y,u,v uint8/uint16 planes (copied); mbs (N,12) int array in ref_tap order; mot (H/4,W/4,2,3)."""
    Y = np.ascontiguousarray(y, np.uint16).copy()
    U = np.ascontiguousarray(u, np.uint16).copy() if u is not None else None
    V = np.ascontiguousarray(v, np.uint16).copy() if v is not None else None
    h, w = Y.shape
    n = mbs.shape[0]
    arr = (DbMb * n)()
    for i in range(n):
        m = mbs[i]
        arr[i] = DbMb(int(m[0]), int(m[1]), int(m[2]), (C.c_int16 * 2)(int(m[3]), int(m[4])), int(m[5]),
                      int(m[6]) & 0xFFFF, int(m[7]), int(m[8]), int(m[9]), int(m[10]), int(m[11]), 0)
    mo = np.zeros((h // 4, w // 4, 4), np.int32)      # packs {int16 mv[2][2]; int32 ref_id[2]} = 16 bytes
    mv = mot[:, :, :, 0:2].astype(np.int16)            # [y][x][list][xy]
    mo_bytes = np.zeros((h // 4, w // 4, 16), np.uint8)
    mo_bytes[:, :, 0:8] = mv.reshape(h // 4, w // 4, 4).view(np.uint8).reshape(h // 4, w // 4, 8)
    mo_bytes[:, :, 8:16] = mot[:, :, :, 2].astype(np.int32).reshape(h // 4, w // 4, 2).view(np.uint8).reshape(h // 4, w // 4, 8)
    mo_bytes = np.ascontiguousarray(mo_bytes)
    L.jmo_deblock_frame(_p(Y), w, _p(U) if U is not None else None, _p(V) if V is not None else None,
                        U.shape[1] if U is not None else 0, w, h, fmt, arr, _p(mo_bytes), maxy, maxc, d8)
    return Y, U, V


PRED_AVG, PRED_BI_WP, PRED_UNI_WP, PRED_UNI = 0, 1, 2, 3


def pred_dist(ref1, ref2, orig, bsx, bsy, test8x8, metric, pred, weights, min_mcost, cand1, cand2, max_pel=255):
    """computeBiPred{SAD,SSE,SATD}1 / 2, compute{SAD,SSE,SATD}WP, compute{SAD,SSE,SATD} (me_distortion.c) for one candidate (pair).
    weights = (w1, w2, offset, round, shift) exactly as the formula uses them; cand* absolute quarter-pel positions."""
    wp = WP((C.c_int * 2)(int(weights[0]), int(weights[1])), int(weights[2]), int(weights[3]), int(weights[4]))
    orig = np.ascontiguousarray(orig, np.uint16)
    return int(L.jmo_compute_pred_dist(ref1.ptr(), (ref2 or ref1).ptr(), _p(orig), int(bsx), int(bsy), int(test8x8), int(metric), int(pred),
                                       C.byref(wp), int(max_pel), int(min_mcost), int(cand1[0]), int(cand1[1]), int(cand2[0]), int(cand2[1])))


def load_frame(raw, src_w, src_h, w, h, yuv):
    """read_one_frame's buf2img + pad_borders: raw planar 8-bit frame bytes -> (y, u, v) uint8 planes of the coded size (u, v None at 4:0:0)"""
    raw = np.ascontiguousarray(np.frombuffer(raw, np.uint8) if not isinstance(raw, np.ndarray) else raw, np.uint8)
    cw, ch = w // 2, (h // 2 if yuv == 1 else h)
    y = np.zeros((h, w), np.uint16); u = np.zeros((ch, cw), np.uint16); v = np.zeros((ch, cw), np.uint16)
    L.jmo_load_frame(_p(raw), int(src_w), int(src_h), int(w), int(h), int(yuv), _p(y), _p(u), _p(v))
    return y.astype(np.uint8), (u.astype(np.uint8) if yuv else None), (v.astype(np.uint8) if yuv else None)


def load_frame_ex(raw, yuv, src_w, src_h, out_w, out_h, symbol_bytes, src_depth, out_depth):
    """the general reader (jmo_load_frame_ex): (y, u, v) uint16 planes of the coded size, chroma by yuv (0: none)"""
    raw = np.ascontiguousarray(np.frombuffer(raw, np.uint8) if not isinstance(raw, np.ndarray) else raw, np.uint8)
    W, H = int(out_w + 15) // 16 * 16, int(out_h + 15) // 16 * 16
    sx, sy = (1 if yuv in (1, 2) else 0), (1 if yuv == 1 else 0)
    y = np.zeros((H, W), np.uint16)
    u = np.zeros((H >> sy, W >> sx) if yuv else (1, 1), np.uint16)
    v = np.zeros_like(u)
    I3 = C.c_int * 3
    L.jmo_load_frame_ex(_p(raw), int(yuv), int(src_w), int(src_h), int(out_w), int(out_h), W, H, int(symbol_bytes), I3(*[int(src_depth)] * 3), I3(*[int(out_depth)] * 3), _p(y), _p(u), _p(v))
    return y, u, v


# ---- motion-compensated prediction (jmo_mc.c)
def luma_pred(r0, r1, p_dir, x, y, bsx, bsy, mv0, mv1):
    """luma_prediction, un-weighted: r0 / r1 RefPic of list 0 / 1 (either may be None when unused); returns (bsy, bsx) uint8"""
    out = np.zeros(bsx * bsy, np.uint16)
    any_ref = r0 if r0 is not None else r1
    L.jmo_luma_pred((r0 or any_ref).ptr(), (r1 or any_ref).ptr(), int(p_dir), int(x), int(y), int(bsx), int(bsy),
                    MV(int(mv0[0]), int(mv0[1])), MV(int(mv1[0]), int(mv1[1])), _p(out))
    return out.reshape(bsy, bsx).astype(np.uint8)


def weighted_samples(p0, p1, p_dir, weights, max_pel=255):
    """weighted_mc_prediction / weighted_bi_prediction on per-list predictions (arrays of equal shape); weights = (w0, w1, offset, round, shift)"""
    a = np.ascontiguousarray(p0 if p0 is not None else p1, np.uint16)
    b = np.ascontiguousarray(p1 if p1 is not None else p0, np.uint16)
    wp = WP((C.c_int * 2)(int(weights[0]), int(weights[1])), int(weights[2]), int(weights[3]), int(weights[4]))
    out = np.zeros(a.size, np.uint16)
    L.jmo_weighted_samples(_p(a), _p(b), a.size, int(p_dir), C.byref(wp), int(max_pel), _p(out))
    return out.reshape(a.shape).astype(np.uint8)


def luma_pred_wp(r0, r1, p_dir, x, y, bsx, bsy, mv0, mv1, weights):
    """luma_prediction with weighted prediction: the per-list predictions, then weighted_samples"""
    a = luma_pred(r0, r1, 0, x, y, bsx, bsy, mv0, mv1) if p_dir != 1 else None
    b = luma_pred(r0, r1, 1, x, y, bsx, bsy, mv0, mv1) if p_dir != 0 else None
    return weighted_samples(a, b, p_dir, weights)


def chroma_pred4x4_wp(p0, p1, yuv, p_dir, xc, yc, mv0, mv1, weights):
    a = chroma_pred4x4(p0, p1, yuv, 0, xc, yc, mv0, mv1) if p_dir != 1 else None
    b = chroma_pred4x4(p0, p1, yuv, 1, xc, yc, mv0, mv1) if p_dir != 0 else None
    return weighted_samples(a, b, p_dir, weights)


def chroma_pred4x4(p0, p1, yuv, p_dir, xc, yc, mv0, mv1):
    """chroma_prediction_4x4 (ChromaMCBuffer = 1), un-weighted: p0 / p1 integer chroma planes (H, W) of list 0 / 1;
    mv0 / mv1: (4, 2, 2) vectors per sample row and sample pair; returns (4, 4) uint8"""
    anyp = p0 if p0 is not None else p1
    a = np.ascontiguousarray(p0 if p0 is not None else anyp, np.uint16)
    b = np.ascontiguousarray(p1 if p1 is not None else anyp, np.uint16)
    h, w = a.shape
    m0 = np.ascontiguousarray(mv0, np.int16).reshape(4, 2, 2)
    m1 = np.ascontiguousarray(mv1, np.int16).reshape(4, 2, 2)
    out = np.zeros(16, np.uint16)
    L.jmo_chroma_pred4x4(_p(a), _p(b), w, w, h, int(yuv), int(p_dir), int(xc), int(yc), _p(m0), _p(m1), _p(out))
    return out.reshape(4, 4).astype(np.uint8)


def rtq_luma_16x16(orig, pred, qparams, qp_per, cavlc, around, arw, max_pel=255):
    """residual_transform_quant_luma_16x16 (block.c:208): returns (ac_coef, dc_level[17], dc_run[17], ac_level[16][16], ac_run[16][16],
    rec (16, 16) uint8, fadjust (4, 16) with the AC positions JM writes)"""
    o = np.ascontiguousarray(orig, np.uint16).reshape(256); p = np.ascontiguousarray(pred, np.uint16).reshape(256)
    q = np.ascontiguousarray(qparams, np.int32).reshape(16, 3)
    dl, dr = np.zeros(17, np.int32), np.zeros(17, np.int32)
    al, ar = np.zeros((16, 16), np.int32), np.zeros((16, 16), np.int32)
    rec, fadj = np.zeros(256, np.uint16), np.zeros(64, np.int32)
    r = L.jmo_rtq_luma_16x16(_p(o), _p(p), _p(q), int(qp_per), int(cavlc), int(around), int(arw), int(max_pel),
                             _p(dl), _p(dr), _p(al), _p(ar), _p(rec), _p(fadj))
    return r, dl, dr, al, ar, rec.reshape(16, 16).astype(np.uint8), fadj.reshape(4, 16)


def hadamard_sad(diff):
    """HadamardSAD4x4 / HadamardSAD8x8 (me_distortion.c:175 / :266) of one int16 difference block (16 or 64 values)"""
    d = np.ascontiguousarray(diff, np.int16).reshape(-1)
    return int(L.jmo_hadamard_sad4x4(_p(d)) if d.size == 16 else L.jmo_hadamard_sad8x8(_p(d)))


# ---- intra prediction (jmo_intra.c)
L.jmo_intra16_search.restype = C.c_int64
L.jmo_dist_i16x16.restype = C.c_int64


def intrapred_4x4(edge, mode, left, up):
    e = np.ascontiguousarray(edge, np.uint16).reshape(13); out = np.zeros(16, np.uint16)
    L.jmo_intrapred_4x4(_p(e), int(mode), int(left), int(up), _p(out))
    return out.reshape(4, 4).astype(np.uint8)


def intrapred_8x8(edge, mode, left, up):
    e = np.ascontiguousarray(edge, np.uint16).reshape(25); out = np.zeros(64, np.uint16)
    L.jmo_intrapred_8x8(_p(e), int(mode), int(left), int(up), _p(out))
    return out.reshape(8, 8).astype(np.uint8)


def intrapred_16x16(edge, mode, left, up, max_pel=255):
    e = np.ascontiguousarray(edge, np.uint16).reshape(33); out = np.zeros(256, np.uint16)
    L.jmo_intrapred_16x16(_p(e), int(mode), int(left), int(up), int(max_pel), _p(out))
    return out.reshape(16, 16).astype(np.uint8)


def intra16_search(edge, left, up, mode_mask, metric, orig, max_pel=255):
    """find_sad_16x16_JM: returns (best cost, best mode, predictions (4, 16, 16) of the evaluated modes)"""
    e = np.ascontiguousarray(edge, np.uint16).reshape(33); o = np.ascontiguousarray(orig, np.uint16).reshape(256)
    pred = np.zeros((4, 256), np.uint16); best = C.c_int(0)
    c = L.jmo_intra16_search(_p(e), int(left), int(up), int(mode_mask), int(metric), int(max_pel), _p(o), _p(pred), C.byref(best))
    return int(c), best.value, pred.reshape(4, 16, 16).astype(np.uint8)


def intra_chroma_pred(up, left, corner, up_avail, left_avail, upleft_avail, ch, max_pel=255):
    """intra_chroma_prediction for one plane: returns (mask of modes written, predictions (4, ch, 8) uint8; modes DC, horizontal, vertical, plane)"""
    u = np.ascontiguousarray(up, np.uint16); l = np.ascontiguousarray(left, np.uint16)
    out = np.zeros((4, 128), np.uint16)
    m = L.jmo_intra_chroma_pred(_p(u), _p(l), int(corner), int(up_avail), int(left_avail), int(upleft_avail), int(ch), int(max_pel), _p(out))
    return int(m), out.reshape(4, 16, 8)[:, :ch].astype(np.uint8)


def sub_images_chroma(plane, yuv):
    """getSubImagesChroma of one plane: (ny, 8, H + 2 pad_y, W + 2 pad_x) uint8, ny = 8 (4:2:0) or 4 (4:2:2)"""
    p = np.ascontiguousarray(plane, np.uint16); h, w = p.shape
    ny, pad_x, pad_y = (4 if yuv == 2 else 8), PAD_X // 2, (PAD_Y if yuv == 2 else PAD_Y // 2)
    out = np.zeros((ny, 8, h + 2 * pad_y, w + 2 * pad_x), np.uint16)
    L.jmo_sub_images_chroma(_p(p), w, w, h, int(yuv), _p(out))
    return out.astype(np.uint8)


# ---- the RDO-off macroblock pipeline (jmo_mbenc.c) ----
MAX_REF = 16


class MbEncCfg(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("slice_type", C.c_int32), ("first_mb", C.c_int32), ("num_mb", C.c_int32),
                ("qp", C.c_int32), ("qpc", C.c_int32), ("search_range", C.c_int32), ("num_ref", C.c_int32), ("lambda_mf", C.c_int32 * 3),
                ("lambda_mdfp", C.c_int32), ("max_mvd", C.c_int32), ("mv_limit", C.c_int32 * 4), ("inter_valid", C.c_int32 * 8),
                ("intra4_valid", C.c_int32), ("intra16_valid", C.c_int32), ("subpel", C.c_int32), ("off4", C.c_int16 * 16 * 2 * 3), ("start_qp", C.c_int32), ("refbits", C.c_int32 * MAX_REF), ("cabac", C.c_int32), ("search_mode", C.c_int32), ("transform8x8", C.c_int32), ("off8", C.c_int16 * 64 * 2), ("intra8_valid", C.c_int32), ("qpc_cr_delta", C.c_int32), ("yuv_format", C.c_int32)]


NO_REF = -(1 << 30)


class EpzsCfg(C.Structure):
    """jmo_epzs_cfg: the configuration's EPZS switches + what EPZSSliceInit reads from the reference list"""
    _fields_ = [("pattern", C.c_int32), ("dual", C.c_int32), ("fixed", C.c_int32), ("aggressive", C.c_int32), ("temporal", C.c_int32), ("spatial_mem", C.c_int32),
                ("blocktype", C.c_int32), ("min_scale", C.c_int32), ("med_scale", C.c_int32), ("max_scale", C.c_int32), ("sub_scale", C.c_int32),
                ("poc_cur", C.c_int32), ("poc_ref", C.c_int32 * MAX_REF), ("col_mv", C.c_void_p * 2), ("col_refpoc", C.c_void_p * 2),
                ("alias_hits", C.c_int64), ("searches", C.c_int64)]


EPZS_DEFAULTS = dict(pattern=2, dual=3, fixed=2, aggressive=0, temporal=1, spatial_mem=1, blocktype=1, min_scale=0, med_scale=1, max_scale=2, sub_scale=2)   # the shipped .cfg files


MB_RECORD = np.dtype([("mb_type", "i1"), ("i16mode", "i1"), ("c_ipred_mode", "i1"), ("transform8x8", "i1"), ("cbp", "<i2"), ("pad1", "<i2"),
                      ("cbp_blk", "<u8"), ("min_rdcost", "<i8"), ("b8mode", "i1", (4,)), ("b8ref", "i1", (4,)), ("ipredmode", "i1", (16,)),
                      ("ipred_syntax", "i1", (16,)), ("mv", "<i2", (16, 2)), ("luma", "<i2", (16, 16)), ("luma_dc", "<i2", (16,)),
                      ("chroma_dc", "<i2", (2, 8)), ("chroma_ac", "<i2", (2, 8, 16)),
                      # B slices (zero otherwise): list-1 vectors and references, prediction direction per 8x8 block, which set of bi-predictive vectors
                      ("mv1", "<i2", (16, 2)), ("b8ref1", "i1", (4,)), ("b8pdir", "i1", (4,)), ("b8bipred", "i1", (4,)), ("pad2", "i1", (4,))])
MB_DEBUG = np.dtype([("motion_cost", "<i8", (8, 4)), ("all_mv", "<i2", (8, 16, 2)), ("best_mode", "<i4"), ("pad", "<i4"), ("motion_cost1", "<i8", (8, 4))])
assert MB_RECORD.itemsize == 1296


class BCfg(C.Structure):
    """jmo_b_cfg: what a B slice has beside MbEncCfg"""
    _fields_ = [("num_ref1", C.c_int32), ("direct_8x8_inference", C.c_int32), ("col_long_term", C.c_int32), ("bipred_me", C.c_int32), ("bipred_search", C.c_int32 * 4),
                ("bipred_refinements", C.c_int32), ("bipred_range", C.c_int32), ("bipred_subpel", C.c_int32), ("col_ref", C.c_void_p), ("col_mv", C.c_void_p),
                ("direct_temporal", C.c_int32), ("poc_cur", C.c_int32), ("poc_l0", C.c_int32 * 16), ("poc_l1_0", C.c_int32), ("col_refpoc", C.c_void_p)]


def mbenc_cfg(width, height, slice_type, first_mb, num_mb, qp, R, num_ref, lambda_mf, lambda_mdfp, level_mv=(-8192, 8191, -2048, 2047),
              subpel=1, cabac=0, search_mode=-1, transform8x8=0, yuv_format=1, offsets=None, inter_valid=None, qpc=None, qpc_cr_delta=0):
    """qpc: currMB->qpc[0] when the chroma QP offsets are not 0 (default: the table value of qp); qpc_cr_delta: qpc[1] - qpc[0].  inter_valid: enc_mb.valid[0..7] of a P slice (PSliceSkip, PSliceSearch16x16 .. 4x4; default all on).  offsets: None = JM's default quantiser offsets, or the lists of a q_offset.cfg (load_q_offsets) when the sequence has OffsetMatrixPresentFlag = 1"""
    c = MbEncCfg()
    o4, o8 = slice_offsets(slice_type, offsets)
    for pl in range(3):
        for intra in range(2):
            for k in range(16):
                c.off4[pl][intra][k] = int(o4[pl][intra][k])
    for intra in range(2):
        for k in range(64):
            c.off8[intra][k] = int(o8[intra][k])
    c.yuv_format = yuv_format
    c.cabac = cabac
    c.search_mode = search_mode
    c.transform8x8 = transform8x8
    c.intra8_valid = 1 if transform8x8 else 0
    c.width, c.height, c.slice_type, c.first_mb, c.num_mb = width, height, slice_type, first_mb, num_mb
    c.qp = qp
    c.qpc = qp if qp < 30 else [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39][qp - 30]
    if qpc is not None:
        c.qpc = int(qpc)
    c.qpc_cr_delta = int(qpc_cr_delta)
    c.search_range, c.num_ref = R, num_ref
    for i in range(3):
        c.lambda_mf[i] = lambda_mf[i]
    c.lambda_mdfp = lambda_mdfp
    max_mv_bits = 3 + 2 * int(np.ceil(np.log2(4 * (2 * R + 3) + 1) + 1e-10))
    c.max_mvd = (1 << (max_mv_bits >> 1)) - 1
    for i in range(4):
        c.mv_limit[i] = level_mv[i]
    for m in range(8):
        c.inter_valid[m] = 1 if inter_valid is None else int(inter_valid[m])
    c.intra4_valid = c.intra16_valid = 1
    c.subpel = subpel
    c.start_qp = 1
    c.refbits[0] = 1
    bits = 3
    while True:
        i_max = (1 << ((bits >> 1) + 1)) - 1
        i_min = i_max >> 1
        for i in range(i_min, min(i_max, MAX_REF)):
            c.refbits[i] = bits
        if i_max >= MAX_REF:
            break
        bits += 2
    return c


def load_q_offsets(path):
    """the lists of a q_offset.cfg (lencod/src/q_offsets.c:300-420 parses `NAME = v, v, ...`) as {name: [values]}"""
    import re
    txt = re.sub(r"#[^\n]*", "", open(path).read())
    out = {}
    for m in re.finditer(r"([A-Z0-9_]+)\s*=\s*([-0-9,\s]+)", txt):
        out[m.group(1)] = [int(v) for v in m.group(2).replace(",", " ").split()]
    return out


def slice_offsets(slice_type, offsets=None):
    """(off4[plane][inter, intra][16], off8[inter, intra][64]) as CalculateOffset4x4Param / 8x8Param (q_offsets.c:633, :720) pick them for an I (2) or a P (0) slice"""
    if offsets is None:
        intra = 682 if slice_type == 2 else 342                      # Offset_intra_default_intra / _inter, Offset_inter_default (q_offsets.c:135-162)
        return [[[342] * 16, [intra] * 16] for _ in range(3)], [[342] * 64, [intra] * 64]
    sfx = "INTRA" if slice_type == 2 else ("INTERB" if slice_type == 1 else "INTERP")       # B slices: the lists 12..14 / 6..8 (q_offsets.c:660-678)
    isfx = "INTERB" if slice_type == 1 else "INTERP"
    o4 = [[offsets[f"INTER4X4_{pl}_{isfx}"], offsets[f"INTRA4X4_{pl}_{sfx}"]] for pl in ("LUMA", "CHROMAU", "CHROMAV")]
    return o4, [offsets[f"INTER8X8_LUMA_{isfx}"], offsets[f"INTRA8X8_LUMA_{sfx}"]]


class Picture:
    """Per-picture state the slices of a picture share: reconstruction, mv_info, ipredmode."""

    def __init__(self, width, height, yuv_format=1):
        self.w, self.h = width, height
        ch = height if yuv_format == 2 else height // 2
        self.rec = [np.zeros((height, width), np.uint16), np.zeros((ch, width // 2), np.uint16), np.zeros((ch, width // 2), np.uint16)]
        self.mv = np.zeros((height // 4, width // 4, 2), np.int16)
        self.ref_idx = np.full((height // 4, width // 4), -1, np.int8)
        self.mv1 = np.zeros((height // 4, width // 4, 2), np.int16)           # list 1 (B slices)
        self.ref_idx1 = np.full((height // 4, width // 4), -1, np.int8)
        self.ipredmode = np.full((height // 4, width // 4), 2, np.int8)


def encode_slice(cfg, cur, refs, refc, pic, debug=False, epzs=None):
    """cur: (y, u, v) uint16 planes at the coded size; refs: list of RefPic; refc: list of (u, v) uint16 chroma planes per reference.
    epzs (SearchMode 3): dict(params=EPZS switches, poc_cur, poc_ref=[..], col=[(mv (h4, w4, 2) int16, refpoc (h4, w4) int32)] of references 0 and 1);
    the EpzsCfg comes back in epzs["out"] (alias_hits, searches)."""
    cy, cu, cv = [np.ascontiguousarray(p, np.uint16) for p in cur]
    n = cfg.num_mb
    out = np.zeros(n, MB_RECORD)
    dbg = np.zeros(n, MB_DEBUG) if debug else None
    RA = RefPicS * max(1, len(refs))
    ra = RA(*[r.s for r in refs]) if refs else RA()
    keep = [np.ascontiguousarray(p, np.uint16) for pair in refc for p in pair]
    PA = C.c_void_p * max(1, len(keep))
    pa = PA(*[k.ctypes.data for k in keep]) if keep else PA()
    ez, keep2 = None, []
    if epzs is not None:
        ez = EpzsCfg()
        for k, v in dict(EPZS_DEFAULTS, **epzs.get("params", {})).items():
            setattr(ez, k, v)
        ez.poc_cur = epzs["poc_cur"]
        for i, pc in enumerate(epzs["poc_ref"]):
            ez.poc_ref[i] = pc
        for i, (cmv, crp) in enumerate(epzs.get("col", [])[:2]):
            a, b = np.ascontiguousarray(cmv, np.int16), np.ascontiguousarray(crp, np.int32)
            keep2 += [a, b]
            ez.col_mv[i], ez.col_refpoc[i] = a.ctypes.data, b.ctypes.data
        epzs["out"] = ez
    r = L.jmo_encode_slice_ex(C.byref(cfg), C.byref(ez) if ez is not None else None, _p(cy), _p(cu), _p(cv), ra, pa, _p(pic.rec[0]), _p(pic.rec[1]), _p(pic.rec[2]),
                              _p(pic.mv), _p(pic.ref_idx), _p(pic.ipredmode), _p(out), _p(dbg) if debug else None)
    assert r == 0, r
    return (out, dbg) if debug else out


def encode_slice_b(cfg, b, cur, refs0, refc0, refs1, refc1, pic, debug=False):
    """A B slice (jmo_encode_slice_b): cfg.slice_type 1, cfg.num_ref = size of list 0; b: dict(num_ref1, direct_8x8_inference, col_ref (h4, w4, 2) int8, col_mv (h4, w4, 2, 2) int16 =
    the motion of listX[LIST_1][0], col_long_term, bipred_me, bipred_search[4], bipred_refinements, bipred_range, bipred_subpel; with direct_temporal: poc_cur, poc_l0[], poc_l1_0,
    col_refpoc (h4, w4, 2) int32 = the picture order counts of the co-located blocks' reference pictures); refs0 / refc0, refs1 / refc1: the lists' RefPic / chroma planes"""
    cy, cu, cv = [np.ascontiguousarray(p, np.uint16) for p in cur]
    n = cfg.num_mb
    out = np.zeros(n, MB_RECORD)
    dbg = np.zeros(n, MB_DEBUG) if debug else None
    keepers = []

    def lists(refs, refc):
        RA = RefPicS * max(1, len(refs))
        ra = RA(*[r.s for r in refs])
        keep = [np.ascontiguousarray(p, np.uint16) for pair in refc for p in pair]
        PA = C.c_void_p * max(1, len(keep))
        keepers.append((ra, keep))
        return ra, PA(*[k.ctypes.data for k in keep])
    ra0, pa0 = lists(refs0, refc0)
    ra1, pa1 = lists(refs1, refc1)
    bc = BCfg()
    bc.num_ref1 = len(refs1)
    bc.direct_8x8_inference = int(b.get("direct_8x8_inference", 1))
    bc.col_long_term = int(b.get("col_long_term", 0))
    bc.bipred_me = int(b.get("bipred_me", 0))
    for i, v in enumerate(b.get("bipred_search", (1, 1, 1, 0))):
        bc.bipred_search[i] = int(v)
    bc.bipred_refinements, bc.bipred_range, bc.bipred_subpel = int(b.get("bipred_refinements", 3)), int(b.get("bipred_range", 16)), int(b.get("bipred_subpel", 2))
    cr, cm = np.ascontiguousarray(b["col_ref"], np.int8), np.ascontiguousarray(b["col_mv"], np.int16)
    bc.col_ref, bc.col_mv = cr.ctypes.data, cm.ctypes.data
    crp = None
    if b.get("direct_temporal", 0):                          # DirectModeType 0: the picture order counts of the picture, of list 0 and of list 1's first picture; the co-located blocks' reference pictures'
        bc.direct_temporal, bc.poc_cur, bc.poc_l1_0 = 1, int(b["poc_cur"]), int(b["poc_l1_0"])
        for i, v in enumerate(b["poc_l0"]):
            bc.poc_l0[i] = int(v)
        crp = np.ascontiguousarray(b["col_refpoc"], np.int32)
        bc.col_refpoc = crp.ctypes.data
    r = L.jmo_encode_slice_b(C.byref(cfg), C.byref(bc), _p(cy), _p(cu), _p(cv), ra0, pa0, ra1, pa1, _p(pic.rec[0]), _p(pic.rec[1]), _p(pic.rec[2]),
                             _p(pic.mv), _p(pic.ref_idx), _p(pic.mv1), _p(pic.ref_idx1), _p(pic.ipredmode), _p(out), _p(dbg) if debug else None)
    assert r == 0, r
    return (out, dbg) if debug else out


def direct_spatial(nb_avail, nb_ref, nb_mv, col_long_term, col_ref, col_mv):
    """jmo_direct_spatial (oracle/jmo_direct.c; Get_Direct_MV_Spatial_Normal lencod/src/mv_direct.c:522): the neighbours A, B, C and the co-located picture's motion of one
    macroblock of a B slice -> (direct_ref_idx [16][2], direct_pdir [16], vectors [16][2][2])"""
    av = np.ascontiguousarray(np.asarray(nb_avail)[:3], np.int8)
    nr, nm = np.ascontiguousarray(nb_ref, np.int8), np.ascontiguousarray(nb_mv, np.int16)
    cr, cm = np.ascontiguousarray(col_ref, np.int8), np.ascontiguousarray(col_mv, np.int16)
    ro, po, mo = np.zeros((16, 2), np.int8), np.zeros(16, np.int8), np.zeros((16, 2, 2), np.int16)
    L.jmo_direct_spatial(_p(av), _p(nr), _p(nm), int(col_long_term), _p(cr), _p(cm), _p(ro), _p(po), _p(mo))
    return ro, po, mo
