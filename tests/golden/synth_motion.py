"""TEST INFRASTRUCTURE: a small synthetic clip with a non-trivial motion field (own generator, seed-pinned): a textured background that drifts by a
fractional number of samples per picture, rectangles with their own textures and velocities on top of it, one of them appearing late, sensor noise.
EPZS (predictor sets, early exits, pattern walks, several references) is pinned on this clip; the reference's own sample clip has three pictures only.
planar 4:2:0 (or, on request, 4:2:2) bytes per frame."""
import numpy as np


def _texture(rng, h, w, cell, k):
    base = rng.integers(0, 256, (h // cell + 2, w // cell + 2)).astype(np.float64)
    big = np.kron(base, np.ones((cell, cell)))[:h + k, :w + k]
    c = np.cumsum(np.cumsum(np.pad(big, ((1, 0), (1, 0))), 0), 1)
    return (c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k])[:h, :w] / (k * k)


def _shifted(tex, y, x, h, w):
    """tex sampled at (y, x) + integer grid, bilinear"""
    iy, ix = int(np.floor(y)), int(np.floor(x))
    fy, fx = y - iy, x - ix
    a = tex[iy:iy + h + 1, ix:ix + w + 1]
    return (1 - fy) * ((1 - fx) * a[:h, :w] + fx * a[:h, 1:w + 1]) + fy * ((1 - fx) * a[1:h + 1, :w] + fx * a[1:h + 1, 1:w + 1])


def motion_clip(W, H, nfr, seed, yuv422=False):
    rng = np.random.default_rng(seed)
    bg = _texture(rng, H + 96, W + 96, 8, 5)
    objs = []
    for k in range(5):
        w, h = int(rng.integers(12, W // 3)), int(rng.integers(12, H // 3))
        objs.append(dict(x=float(rng.integers(0, W - w)), y=float(rng.integers(0, H - h)), w=w, h=h, vx=float(rng.integers(-9, 10)) / 2, vy=float(rng.integers(-7, 8)) / 2,
                         tex=_texture(rng, h + 40, w + 40, 4 + 2 * (k % 3), 3), born=0 if k < 4 else 2, tx=float(rng.integers(-3, 4)) / 4))
    frames = []
    for n in range(nfr):
        m = 24 - abs(n % 48 - 24)                            # the background's drift and the textures' slide turn round every 24 pictures (m = n up to there): the margins are finite
        y = _shifted(bg, 40 + 0.5 * m, 40 + 1.75 * m, H, W)
        for o in objs:
            if n < o["born"]:
                continue
            x0, y0 = int(round(o["x"] + o["vx"] * n)), int(round(o["y"] + o["vy"] * n))
            xa, ya, xb, yb = max(0, x0), max(0, y0), min(W, x0 + o["w"]), min(H, y0 + o["h"])
            if xa >= xb or ya >= yb:
                continue
            patch = _shifted(o["tex"], 20 + (ya - y0), 20 + (xa - x0) + o["tx"] * m, yb - ya, xb - xa)
            y[ya:yb, xa:xb] = patch
        y = np.clip(np.rint(y + rng.normal(0, 1.5, (H, W))), 0, 255).astype(np.uint8)
        yd = y.reshape(H, W // 2, 2).mean(axis=2) if yuv422 else y.reshape(H // 2, 2, W // 2, 2).mean(axis=(1, 3))     # 4:2:2: chroma planes W / 2 x H
        u = np.clip(np.rint(128 + 0.3 * (yd - 128)), 0, 255).astype(np.uint8)
        v = np.clip(np.rint(128 - 0.2 * (yd - 128)), 0, 255).astype(np.uint8)
        frames.append(np.concatenate([y.ravel(), u.ravel(), v.ravel()]))
    return frames
