"""Reader of oracle/ref_tap_mb.c's dump (mb_low.bin) and helpers shared by make_mb_golden.py and the tests: TEST INFRASTRUCTURE.

The dump is what the REAL reference encoder's encode_one_macroblock_low left behind per macroblock; `to_records` turns it into the
record layout of the product ABI / the oracle (jmhip_mb_record = jmo_mb_record, 1216 bytes) so the three can be compared field by field."""
import numpy as np

TAP = np.dtype([("frame_no", "<i4"), ("mb_addr", "<i4"), ("slice_type", "<i4"), ("slice_nr", "<i4"),
                ("best_mode", "<i4"), ("mb_type", "<i4"), ("cbp", "<i4"), ("c_ipred_mode", "<i4"), ("i16mode", "<i4"), ("i16offset", "<i4"),
                ("transform8x8", "<i4"), ("qp", "<i4"),
                ("lambda_mf", "<i4", (3,)), ("lambda_mdfp", "<i4"), ("num_ref", "<i4"), ("max_mvd", "<i4"), ("mv_limit", "<i4", (4,)),
                ("qpc", "<i4"), ("search_range", "<i4"),
                ("cbp_blk", "<i8"), ("min_rdcost", "<i8"), ("b8mode", "i1", (4,)), ("b8pdir", "i1", (4,)),
                ("ipred_syntax", "i1", (16,)), ("ipredmode", "i1", (16,)), ("mv", "<i2", (16, 2)), ("ref_idx", "i1", (16,)),
                ("motion_cost", "<i8", (8, 4)), ("all_mv", "<i2", (8, 16, 2)),
                ("luma_level", "<i4", (16, 17)), ("luma_run", "<i4", (16, 17)), ("dc_level", "<i4", (3, 18)), ("dc_run", "<i4", (3, 18)),
                ("chroma_level", "<i4", (8, 17)), ("chroma_run", "<i4", (8, 17)),
                ("rec_y", "u1", (256,)), ("rec_u", "u1", (64,)), ("rec_v", "u1", (64,)),
                ("poc", "<i4"), ("ref_poc", "<i4", (16,)), ("motion_cost_ref", "<i8", (8, 4, 4)),
                ("luma8_level", "<i4", (4, 65)), ("luma8_run", "<i4", (4, 65)),
                ("chroma2_level", "<i4", (8, 17)), ("chroma2_run", "<i4", (8, 17)), ("rec_u2", "u1", (64,)), ("rec_v2", "u1", (64,)), ("yuv_format", "<i4"), ("qpc_v", "<i4"),
                # B slices
                ("mv1", "<i2", (16, 2)), ("ref_idx1", "i1", (16,)), ("b8bipred", "i1", (4,)), ("num_ref1", "<i4"), ("motion_cost1", "<i8", (8, 2, 4)), ("poc_l1", "<i4", (4,)),
                ("direct_8x8_inference", "<i4"), ("pad_", "<i4")])


def read(path):
    return np.fromfile(path, TAP)


def dense(level, run, start, n=16):
    """(level, run) list of JM (zero terminated) -> levels at their scan positions."""
    out = np.zeros(n, np.int16)
    pos = start
    for lv, rn in zip(level, run):
        if lv == 0:
            break
        pos += int(rn)
        out[pos] = lv
        pos += 1
    return out


def expected_coeffs(t, cabac=0):
    """The coefficient arrays write_macroblock would read for tap record t, in the record layout; blocks the coded block pattern hides are zero.
    An 8x8 transform block's 64 levels lie in the 8x8 scan's zig-zag order at luma[4 * b8 + (s >> 4)][s & 15]: JM keeps them as one list (CABAC) or as four
    lists of every fourth position (CAVLC: list s & 3, place s >> 2)."""
    luma = np.zeros((16, 16), np.int16)
    luma_dc = np.zeros(16, np.int16)
    cdc = np.zeros((2, 8), np.int16)
    cac = np.zeros((2, 8, 16), np.int16)
    mbt, cbp = int(t["mb_type"]), int(t["cbp"])
    y422 = int(t["yuv_format"]) == 2
    t8 = int(t["transform8x8"]) and mbt not in (9, 10)
    if mbt == 10:
        luma_dc = dense(t["dc_level"][0], t["dc_run"][0], 0)
    for b8 in range(4):
        if cbp & (1 << b8):
            if t8 and cabac:
                luma[4 * b8:4 * b8 + 4] = dense(t["luma8_level"][b8], t["luma8_run"][b8], 0, 64).reshape(4, 16)
            elif t8:
                z = np.zeros(64, np.int16)
                for l in range(4):
                    z[l::4] = dense(t["luma_level"][4 * b8 + l], t["luma_run"][4 * b8 + l], 0)
                luma[4 * b8:4 * b8 + 4] = z.reshape(4, 16)
            else:
              for b4 in range(4):
                luma[4 * b8 + b4] = dense(t["luma_level"][4 * b8 + b4], t["luma_run"][4 * b8 + b4], 1 if mbt == 10 else 0)
    if cbp > 15:
        for uv in range(2):
            n = 8 if y422 else 4
            cdc[uv][:n] = dense(t["dc_level"][1 + uv], t["dc_run"][1 + uv], 0, n)
    if cbp >> 4 == 2:
        for uv in range(2):
            for b4 in range(4):
                cac[uv][b4] = dense(t["chroma_level"][4 * uv + b4], t["chroma_run"][4 * uv + b4], 1)
                if y422:                         # the plane's blocks 4..7: cofAC[5 + 2 uv][b4]
                    cac[uv][4 + b4] = dense(t["chroma2_level"][4 * uv + b4], t["chroma2_run"][4 * uv + b4], 1)
    return luma, luma_dc, cdc, cac


def visible_coeffs(r):
    """The same view of an oracle / device record: coefficients masked by the record's own coded block pattern."""
    luma = np.array(r["luma"]).copy()
    luma_dc = np.array(r["luma_dc"]).copy()
    cdc = np.array(r["chroma_dc"]).copy()
    cac = np.array(r["chroma_ac"]).copy()
    mbt, cbp = int(r["mb_type"]), int(r["cbp"])
    for b8 in range(4):
        if not cbp & (1 << b8):
            luma[4 * b8:4 * b8 + 4] = 0
    if mbt != 10:
        luma_dc[:] = 0
    if cbp <= 15:
        cdc[:] = 0
    if cbp >> 4 != 2:
        cac[:] = 0
    return luma, luma_dc, cdc, cac


def compare(t, r, rec_mb=None):
    """List of differences between tap record t (the reference) and record r; empty when they agree on everything write_macroblock reads."""
    d = []
    mbt = int(t["mb_type"])
    for k in ("mb_type", "cbp"):
        if int(t[k]) != int(r[k]):
            d.append((k, int(t[k]), int(r[k])))
    if mbt in (1, 2, 3, 8):
        if not np.array_equal(t["mv"], r["mv"]):
            d.append(("mv", t["mv"].tolist(), np.array(r["mv"]).tolist()))
        ref8 = [int(t["ref_idx"][j * 8 + i * 2]) for j in range(2) for i in range(2)]
        if ref8 != [int(x) for x in r["b8ref"]]:
            d.append(("ref", ref8, [int(x) for x in r["b8ref"]]))
    if mbt == 8 and not np.array_equal(t["b8mode"], r["b8mode"]):
        d.append(("b8mode", t["b8mode"].tolist(), np.array(r["b8mode"]).tolist()))
    if mbt == 9 and not np.array_equal(t["ipred_syntax"], r["ipred_syntax"]):
        d.append(("ipred", t["ipred_syntax"].tolist(), np.array(r["ipred_syntax"]).tolist()))
    if mbt == 10 and int(t["i16mode"]) != int(r["i16mode"]):
        d.append(("i16mode", int(t["i16mode"]), int(r["i16mode"])))
    if mbt >= 9 and int(t["c_ipred_mode"]) != int(r["c_ipred_mode"]):
        d.append(("c_ipred_mode", int(t["c_ipred_mode"]), int(r["c_ipred_mode"])))
    if int(t["mb_type"]) == int(r["mb_type"]) and int(t["cbp"]) == int(r["cbp"]):
        for name, a, b in zip(("luma", "luma_dc", "chroma_dc", "chroma_ac"), expected_coeffs(t), visible_coeffs(r)):
            if not np.array_equal(a, b):
                d.append((name, a.tolist(), b.tolist()))
    if rec_mb is not None:
        ru, rv = t["rec_u"].reshape(8, 8), t["rec_v"].reshape(8, 8)
        if int(t["yuv_format"]) == 2:
            ru, rv = np.concatenate([ru, t["rec_u2"].reshape(8, 8)]), np.concatenate([rv, t["rec_v2"].reshape(8, 8)])
        for name, a, b in zip(("rec_y", "rec_u", "rec_v"), (t["rec_y"].reshape(16, 16), ru, rv), rec_mb):
            if not np.array_equal(a, b):
                d.append((name, int(np.abs(a.astype(int) - b.astype(int)).max())))
    return d


# ---- canonical records: what write_macroblock / DeblockFrame can observe, in the jmhip_mb_record layout -------------------------------
def _record_dtype():
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import pyjmo
    return pyjmo.MB_RECORD


def canonical(recs, bslice=False):
    """Normalise an array of macroblock records (oracle or device output): fields the bitstream writer and the loop filter never read for the
    record's macroblock type are set to fixed values, coefficient arrays are masked by the coded block pattern.  bslice: the records of a B slice
    (mb_type 0 = direct, list 1 / prediction directions kept); otherwise the B fields are zero."""
    out = np.array(recs, copy=True)
    for r in out:
        mbt, cbp = int(r["mb_type"]), int(r["cbp"])
        r["pad1"] = 0
        r["pad2"] = 0
        if not bslice:
            r["mv1"], r["b8ref1"], r["b8pdir"], r["b8bipred"] = 0, 0, 0, 0
        elif mbt >= 9:
            r["mv1"], r["b8ref1"], r["b8pdir"], r["b8bipred"] = 0, -1, -1, 0
        luma, luma_dc, cdc, cac = visible_coeffs(r)
        r["luma"], r["luma_dc"], r["chroma_dc"], r["chroma_ac"] = luma, luma_dc, cdc, cac
        if mbt >= 9:
            r["mv"] = 0
            r["b8ref"] = -1
            r["cbp_blk"] = 0
        else:
            r["c_ipred_mode"] = 0
            r["cbp_blk"] = int(r["cbp_blk"]) & 0xFFFF
        if mbt != 10:
            r["i16mode"] = 0
        if mbt == 13:                            # Intra8x8: one syntax element per 8x8 block, at [4 * b8]
            syn = np.array(r["ipred_syntax"]).copy()
            keep = syn[0::4].copy()
            syn[:] = 2
            syn[0::4] = keep
            r["ipred_syntax"] = syn
        elif mbt != 9:
            r["ipred_syntax"] = 2
            r["ipredmode"] = 2
        if mbt != 8:
            r["b8mode"] = 11 if mbt == 9 else (13 if mbt == 13 else (0 if mbt in (0, 10) else mbt))
        if mbt == 0 and not bslice:
            r["b8ref"] = 0
        (cbp)
    return out


def tap_to_records(tap, cabac=0):
    """The reference encoder's dump as canonical records."""
    out = np.zeros(len(tap), _record_dtype())
    for t, r in zip(tap, out):
        mbt = int(t["mb_type"])
        r["transform8x8"] = int(t["transform8x8"])
        r["mb_type"], r["cbp"], r["min_rdcost"] = mbt, int(t["cbp"]), int(t["min_rdcost"])
        r["i16mode"], r["c_ipred_mode"] = int(t["i16mode"]), int(t["c_ipred_mode"])
        r["cbp_blk"] = int(t["cbp_blk"]) & 0xFFFFFFFFFFFFFFFF
        r["b8mode"] = t["b8mode"]
        r["b8ref"] = [int(t["ref_idx"][j * 8 + i * 2]) for j in range(2) for i in range(2)]
        r["ipredmode"], r["ipred_syntax"], r["mv"] = t["ipredmode"], t["ipred_syntax"], t["mv"]
        r["luma"], r["luma_dc"], r["chroma_dc"], r["chroma_ac"] = expected_coeffs(t, cabac)
        if int(t["slice_type"]) == 1:
            r["mv1"], r["b8pdir"], r["b8bipred"] = t["mv1"], t["b8pdir"], t["b8bipred"]
            r["b8ref1"] = [int(t["ref_idx1"][j * 8 + i * 2]) for j in range(2) for i in range(2)]
    bs = np.array([int(t["slice_type"]) == 1 for t in tap])
    res = canonical(out)
    if bs.any():
        res[bs] = canonical(out[bs], bslice=True)
    return res


def widen(recs):
    """Records stored with an earlier, narrower layout (the 944-byte record of the 4:2:0-only pipeline: chroma_dc[2][4], chroma_ac[2][4][16]) in today's layout."""
    dt = _record_dtype()
    if recs.dtype == dt:
        return recs
    out = np.zeros(recs.shape, dt)
    for old, new in zip(recs.dtype.names, dt.names):          # same fields in the same order (one was renamed when it got a meaning)
        a = recs[old]
        out[new][tuple([slice(None)] + [slice(0, k) for k in a.shape[1:]])] = a
    return out


def diff_fields(a, b):
    """Names of the fields in which two canonical records differ."""
    return [n for n in a.dtype.names if not np.array_equal(a[n], b[n])]
