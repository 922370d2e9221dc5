"""TEST INFRASTRUCTURE: the oracle's sequence encoder -- oracle/jmo_mbenc.c per slice, oracle deblocking, oracle sub-pel planes -- chained
the way lencod chains pictures (IPPP, frame pictures, sliding-window references), plus the conversion of macroblock records into the
loop filter's side information.  Used by tests/test_oracle_mbenc.py (against the real encoder's dumps) and by the GPU parity tests
(as the checker of jmhip_encode_slice).  Nothing here is on the product path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyjmo  # noqa: E402


def slices_of(nmb, slice_mbs):
    """[(first_mb, num_mb)] for SliceMode 0 (slice_mbs = 0) or 1 (fixed number of macroblocks)."""
    if not slice_mbs:
        return [(0, nmb)]
    return [(f, min(slice_mbs, nmb - f)) for f in range(0, nmb, slice_mbs)]


def db_side_info(records, slice_nr, slice_type, qp, qpc, W, H, ref_ids, disable_idc=0, ref_ids1=None):
    """records (nmb) + per-macroblock slice numbers -> (mbs (nmb, 12) in pyjmo.deblock_frame's order, motion (H/4, W/4, 2, 3)).
    ref_ids[r] = picture identity of reference index r of this picture's list 0 (ref_ids1: list 1, B slices)."""
    wmb = W // 16
    nmb = len(records)
    mbs = np.zeros((nmb, 12), np.int64)
    mot = np.zeros((H // 4, W // 4, 2, 3), np.int32)
    mot[:, :, :, 2] = -1
    for k in range(nmb):
        r = records[k]
        qu, qv = qpc if isinstance(qpc, tuple) else (qpc, qpc)
        mbs[k] = [int(r["mb_type"]), slice_type, qp, qu, qv, int(r["cbp"]), int(r["cbp_blk"]) & 0xFFFF, int(slice_nr[k]), disable_idc, 0, 0, int(r["transform8x8"])]
        mbx, mby = k % wmb, k // wmb
        if int(r["mb_type"]) < 9:
            mv = np.array(r["mv"]).reshape(4, 4, 2)
            mot[mby * 4:mby * 4 + 4, mbx * 4:mbx * 4 + 4, 0, 0:2] = mv
            for b8 in range(4):
                rid = ref_ids[int(r["b8ref"][b8])]
                mot[mby * 4 + (b8 >> 1) * 2:mby * 4 + (b8 >> 1) * 2 + 2, mbx * 4 + (b8 & 1) * 2:mbx * 4 + (b8 & 1) * 2 + 2, 0, 2] = rid
            if slice_type == 1:                  # B: a list that a block does not use has reference index -1 (no picture), list 1 beside list 0
                mot[mby * 4:mby * 4 + 4, mbx * 4:mbx * 4 + 4, 1, 0:2] = np.array(r["mv1"]).reshape(4, 4, 2)
                for b8 in range(4):
                    ys, xs = slice(mby * 4 + (b8 >> 1) * 2, mby * 4 + (b8 >> 1) * 2 + 2), slice(mbx * 4 + (b8 & 1) * 2, mbx * 4 + (b8 & 1) * 2 + 2)
                    r0, r1 = int(r["b8ref"][b8]), int(r["b8ref1"][b8])
                    mot[ys, xs, 0, 2] = ref_ids[r0] if r0 >= 0 else -1
                    mot[ys, xs, 1, 2] = ref_ids1[r1] if r1 >= 0 else -1
    return mbs, mot


class SeqEncoder:
    """IPPP with num_ref sliding-window references, RDOptimization = 0, AdaptiveRounding = 0 (the scope of jmo_mbenc.c)."""

    def __init__(self, W, H, qp, R, num_ref, lambdas, slice_mbs=0, level_mv=(-8192, 8191, -2048, 2047), disable_idc=0, cabac=0, search_mode=-1, epzs=None, transform8x8=0, yuv_format=1, offsets=None, inter_valid=None, qpc=None, qpc_cr_delta=0, qp_p=None, qpc_p=None, qpc_cr_delta_p=None):
        """lambdas[slice_type] = (lambda_mf[3], lambda_mdfp): JM's own tables (double arithmetic, never recomputed).
        search_mode 3 = EPZS with the switches in `epzs` (defaults: the shipped .cfg files', pyjmo.EPZS_DEFAULTS)."""
        self.W, self.H, self.qp, self.R, self.num_ref, self.lambdas = W, H, qp, R, num_ref, lambdas
        self.slice_mbs, self.level_mv, self.disable_idc = slice_mbs, level_mv, disable_idc
        self.cabac = cabac     # SymbolMode: the quantiser clamps levels for CAVLC only
        self.search_mode, self.epzs = search_mode, dict(epzs or {})
        self.transform8x8 = transform8x8     # Transform8x8Mode (0 / 1)
        self.yuv_format = yuv_format         # 1: 4:2:0, 2: 4:2:2
        self.qp_p = qp if qp_p is None else qp_p             # QPPSlice when it differs from QPISlice
        self.qpc, self.qpc_cr_delta = qpc, qpc_cr_delta     # chroma QPs when CbQPOffset / CrQPOffset are not 0
        self.qpc_p, self.qpc_cr_delta_p = (qpc, qpc_cr_delta) if qpc_p is None else (qpc_p, qpc_cr_delta_p)      # ... of the P pictures, when QPPSlice differs from QPISlice as well
        self.inter_valid = inter_valid       # enc_mb.valid[0..7] of P slices (None: all on)
        self.offsets = offsets               # None, or pyjmo.load_q_offsets(q_offset.cfg) with OffsetMatrixPresentFlag = 1
        self.refs = []         # most recent first: (RefPic, (u, v), picture id, (mv, refpoc) per 4x4 block of the stored picture)
        self.npic = 0
        self.keep = 0          # stored reference pictures when that is more than P's list (B pictures: a list 1 longer than any list 0 of the sequence)
        self.epzs_stats = []   # per P slice: (searches, alias_hits) of the oracle's EPZS

    def encode_b(self, cur, poc, l0_pocs, l1_pocs, lambdas_b, qp_b, b=None, debug=False, qpc_b=None, qpc_cr_delta_b=None, inter_valid_b=None):
        """A non-reference B picture (slice type 1) with picture order count poc: its lists are the stored reference pictures with the given picture order counts (as the real
        encoder ordered them), lambdas_b = (lambda_mf[3], lambda_mdfp) of B slices, qp_b = QPBSlice; b: the switches of pyjmo.encode_slice_b (bipred_me ...).
        Nothing is stored: the next P picture refers to the reference pictures only."""
        W, H = self.W, self.H
        nmb = (W // 16) * (H // 16)
        by_poc = {r[4]: r for r in self.refs}
        L0, L1 = [by_poc[p] for p in l0_pocs], [by_poc[p] for p in l1_pocs]
        pic = pyjmo.Picture(W, H, self.yuv_format)
        recs = np.zeros(nmb, pyjmo.MB_RECORD)
        dbg = np.zeros(nmb, pyjmo.MB_DEBUG) if debug else None
        slice_nr = np.zeros(nmb, np.int32)
        cur16 = [np.ascontiguousarray(p, np.uint16) for p in cur]
        col = L1[0][5]                           # (mv0, ref_idx0, mv1, ref_idx1) of the stored picture
        bb = dict(b or {}, col_ref=np.stack([col[1], col[3]], axis=-1), col_mv=np.stack([col[0], col[2]], axis=2))
        if bb.get("direct_temporal", 0):                     # DirectModeType 0: picture order counts for compute_colocated's scales and the co-located blocks' reference pictures (list 0 only: stored pictures are I / P)
            rp = L1[0][3][1]
            bb.update(poc_cur=poc, poc_l0=list(l0_pocs), poc_l1_0=l1_pocs[0], col_refpoc=np.stack([rp, np.full(rp.shape, pyjmo.NO_REF, np.int32)], axis=-1))
        qpc = None
        for sn, (first, num) in enumerate(slices_of(nmb, self.slice_mbs)):
            cfg = pyjmo.mbenc_cfg(W, H, 1, first, num, qp_b, self.R, len(L0), lambdas_b[0], lambdas_b[1], level_mv=self.level_mv, cabac=self.cabac, search_mode=self.search_mode, transform8x8=self.transform8x8,
                                  yuv_format=self.yuv_format, offsets=self.offsets, inter_valid=inter_valid_b, qpc=qpc_b, qpc_cr_delta=self.qpc_cr_delta_p if qpc_cr_delta_b is None else qpc_cr_delta_b)
            qpc = cfg.qpc
            res = pyjmo.encode_slice_b(cfg, bb, cur16, [r[0] for r in L0], [r[1] for r in L0], [r[0] for r in L1], [r[1] for r in L1], pic, debug=debug)
            if debug:
                recs[first:first + num], dbg[first:first + num] = res
            else:
                recs[first:first + num] = res
            slice_nr[first:first + num] = sn
        pre = [p.copy() for p in pic.rec]
        mbs, mot = db_side_info(recs, slice_nr, 1, qp_b, (qpc, qpc + (self.qpc_cr_delta_p if qpc_cr_delta_b is None else qpc_cr_delta_b)), W, H, [r[2] for r in L0], self.disable_idc, [r[2] for r in L1])
        y, u, v = pyjmo.deblock_frame(pic.rec[0], pic.rec[1], pic.rec[2], self.yuv_format, mbs, mot, d8=int((b or {}).get("direct_8x8_inference", 1)))     # (loopFilter.c:180: a direct macroblock's inner edges)
        self.npic += 1
        return recs, dbg, pre, (y, u, v)

    def encode(self, cur, debug=False, poc=None):
        """cur = (y, u, v) at the coded size.  Returns (records, debug records or None, reconstruction before the loop filter, after it).
        poc: the picture's order count when the sequence has B pictures (default: 2 per coded picture)."""
        W, H = self.W, self.H
        nmb = (W // 16) * (H // 16)
        st = 2 if self.npic == 0 else 0
        nref = min(self.num_ref, len(self.refs)) if st == 0 else 0
        pic = pyjmo.Picture(W, H, self.yuv_format)
        recs = np.zeros(nmb, pyjmo.MB_RECORD)
        dbg = np.zeros(nmb, pyjmo.MB_DEBUG) if debug else None
        slice_nr = np.zeros(nmb, np.int32)
        cur16 = [np.ascontiguousarray(p, np.uint16) for p in cur]
        qpc = None
        for sn, (first, num) in enumerate(slices_of(nmb, self.slice_mbs)):
            lam_mf, lam_md = self.lambdas[st]
            cfg = pyjmo.mbenc_cfg(W, H, st, first, num, self.qp if st == 2 else self.qp_p, self.R, nref, lam_mf, lam_md, level_mv=self.level_mv, cabac=self.cabac, search_mode=self.search_mode, transform8x8=self.transform8x8, yuv_format=self.yuv_format, offsets=self.offsets, inter_valid=self.inter_valid, qpc=self.qpc if st == 2 else self.qpc_p, qpc_cr_delta=self.qpc_cr_delta if st == 2 else self.qpc_cr_delta_p)
            qpc = cfg.qpc
            ez = None
            if self.search_mode == 3 and st == 0:       # picture order counts: 2 per frame (IPPP, PicOrderCntType 0)
                ez = dict(params=self.epzs, poc_cur=2 * self.npic, poc_ref=[2 * r[2] for r in self.refs[:nref]], col=[r[3] for r in self.refs[:min(nref, 2)]])
            res = pyjmo.encode_slice(cfg, cur16, [r[0] for r in self.refs[:nref]], [r[1] for r in self.refs[:nref]], pic, debug=debug, epzs=ez)
            if ez is not None:
                self.epzs_stats.append((int(ez["out"].searches), int(ez["out"].alias_hits)))
            if debug:
                recs[first:first + num], dbg[first:first + num] = res
            else:
                recs[first:first + num] = res
            slice_nr[first:first + num] = sn
        pre = [p.copy() for p in pic.rec]
        mbs, mot = db_side_info(recs, slice_nr, st, self.qp if st == 2 else self.qp_p, (qpc, qpc + (self.qpc_cr_delta if st == 2 else self.qpc_cr_delta_p)), W, H, [r[2] for r in self.refs[:nref]] or [0], self.disable_idc)
        y, u, v = pyjmo.deblock_frame(pic.rec[0], pic.rec[1], pic.rec[2], self.yuv_format, mbs, mot)
        refpoc = np.full(pic.ref_idx.shape, pyjmo.NO_REF, np.int32)     # the stored picture's motion, as EPZSSliceInit of later pictures reads it
        for k, r in enumerate(self.refs[:nref]):
            refpoc[pic.ref_idx == k] = r[4]                  # (2 x the picture number in an IPPP sequence)
        self.refs.insert(0, (pyjmo.RefPic(y), (u, v), self.npic, (pic.mv.copy(), refpoc), 2 * self.npic if poc is None else poc, (pic.mv.copy(), pic.ref_idx.copy(), pic.mv1.copy(), pic.ref_idx1.copy())))
        self.refs = self.refs[:max(self.num_ref, self.keep)]
        self.npic += 1
        return recs, dbg, pre, (y, u, v)


def lambdas_from_tap(tap):
    """{slice_type: (lambda_mf[3], lambda_mdfp)} as the real encoder used them (oracle/ref_tap_mb.c records)."""
    out = {}
    for t in tap:
        st = int(t["slice_type"])
        if st not in out:
            out[st] = ([int(x) for x in t["lambda_mf"]], int(t["lambda_mdfp"]))
    return out
