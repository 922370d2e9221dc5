"""Regenerates tests/golden/load_frame.npz: the REAL reference's source-picture reader on seeded frames -- read_one_frame's buf2img calls (lcommon/src/input.c:822-853:
buf2img_basic :552, buf2img_bitshift :440, chosen as initInput :41 does) followed by pad_borders (:880) -- called directly through oracle/ref_call.c
(oracle/_ref/libjmrefcall.so, the unmodified lencod objects).  TEST INFRASTRUCTURE; needs /root/reference.

  python tests/golden/make_load_frame.py

Cases: 4:0:0 / 4:2:0 / 4:2:2 / 4:4:4, 8 .. 14 bit samples in one or two bytes (little endian), source depth == / > / < output depth, picture sizes that are not multiples
of 16 (right / bottom padding), source size != output size (centred copy into a larger picture, crop into a smaller one)."""
import ctypes as C
import os
import subprocess

import numpy as np

G = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(G))
SO = os.path.join(ROOT, "oracle", "_ref", "libjmrefcall.so")

# (yuv_format, src_w, src_h, out_w, out_h, symbol_bytes, source depth, output depth)
CASES = [
    (3, 48, 32, 48, 32, 1, 8, 8),       # 4:4:4, 8 bit, whole macroblocks
    (3, 50, 34, 50, 34, 1, 8, 8),       # 4:4:4, padded right and below
    (1, 36, 20, 36, 20, 2, 10, 10),     # 4:2:0, 10 bit in two bytes
    (2, 40, 24, 40, 24, 2, 12, 8),      # 4:2:2, 12 bit source rounded down to 8 (rshift_rnd)
    (1, 44, 28, 44, 28, 1, 8, 10),      # 4:2:0, 8 bit source scaled up to 10
    (0, 34, 18, 34, 18, 2, 14, 14),     # 4:0:0, 14 bit
    (3, 32, 16, 32, 16, 2, 14, 14),     # 4:4:4, 14 bit
    (1, 40, 24, 48, 32, 1, 8, 8),       # output larger than the source: centred
    (1, 52, 40, 48, 32, 1, 8, 8),       # output smaller: cropped
    (2, 38, 22, 38, 22, 2, 9, 9),       # 4:2:2, 9 bit
    (1, 176, 144, 176, 144, 1, 8, 8),   # QCIF, the common case
    (3, 20, 20, 24, 20, 1, 8, 12),      # 4:4:4, wider output, 8 -> 12 bit (two-byte samples may not be scaled up: input.c:440-443)
    # two-byte samples at equal depth with a size mismatch: buf2img_basic takes row i at BYTE offset i * size_x (input.c:586-588), i.e. half way into the sample row
    (1, 40, 24, 48, 32, 2, 10, 10),     # centred
    (1, 52, 40, 48, 32, 2, 10, 10),     # cropped
    (3, 20, 20, 24, 20, 2, 12, 12),     # 4:4:4, wider output
]


def planes_shape(yuv, w, h):
    sx, sy = (1 if yuv in (1, 2) else 0), (1 if yuv == 1 else 0)
    return (h, w), ((h >> sy, w >> sx) if yuv else (0, 0))


def main():
    if not os.path.exists(SO):
        subprocess.check_call(["make", "-s", "-f", os.path.join(ROOT, "oracle", "Makefile.ref"), "call"])
    L = C.CDLL(SO)
    rng = np.random.default_rng(20260930)
    out = {"cases": np.array(CASES, np.int32)}
    for k, (yuv, sw, sh, ow, oh, sb, sd, od) in enumerate(CASES):
        (hy, wy), (hc, wc) = planes_shape(yuv, sw, sh)
        n = hy * wy + 2 * hc * wc
        samples = rng.integers(0, 1 << sd, n).astype(np.uint16)
        samples[:4] = [(1 << sd) - 1, 0, (1 << sd) - 1, 1]
        raw = samples.astype(np.uint8).tobytes() if sb == 1 else samples.astype("<u2").tobytes()
        W, H = (ow + 15) // 16 * 16, (oh + 15) // 16 * 16
        (_, _), (ch, cw) = planes_shape(yuv, W, H)
        y = np.zeros((H, W), np.uint16)
        u = np.zeros((max(ch, 1), max(cw, 1)), np.uint16)
        v = np.zeros((max(ch, 1), max(cw, 1)), np.uint16)
        buf = C.create_string_buffer(raw, len(raw))
        I3 = C.c_int * 3
        r = L.refcall_load_frame(buf, yuv, sw, sh, ow, oh, W, H, sb, I3(sd, sd, sd), I3(od, od, od), y.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p))
        assert r == 0
        out[f"raw{k}"] = np.frombuffer(raw, np.uint8)
        out[f"y{k}"], out[f"u{k}"], out[f"v{k}"] = y, u, v
    np.savez_compressed(os.path.join(G, "load_frame.npz"), **out)
    print("load_frame.npz:", len(CASES), "cases")


if __name__ == "__main__":
    main()
