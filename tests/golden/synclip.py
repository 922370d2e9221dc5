"""The synthetic clips of SURVEY.md Appendix A (own generator): seed-pinned 4:2:0 YUV, a blurred random block field translating by (3, 2) samples per
picture plus noise.  TEST INFRASTRUCTURE / bench input; bench.write_yuv is the 1080p instance of the same generator."""
import numpy as np


def write_clip(path, W, H, n_frames, seed):
    rng = np.random.default_rng(seed)
    extra = 8 if W == 1920 else 16
    base = rng.integers(0, 256, size=(H // 8 + extra, W // 8 + extra)).astype(np.float32)
    base = np.kron(base, np.ones((8, 8), np.float32))
    k = 5
    b = np.cumsum(np.cumsum(np.pad(base, ((k, k), (k, k)), mode="edge"), 0), 1)
    sm = (b[2 * k:, 2 * k:] - b[:-2 * k, 2 * k:] - b[2 * k:, :-2 * k] + b[:-2 * k, :-2 * k]) / (4 * k * k)
    if W == 1920:
        sm = sm[:H + 64, :W + 64]
    with open(path, "wb") as f:
        for n in range(n_frames):
            dx, dy = 3 * n, 2 * n
            y = sm[dy:dy + H, dx:dx + W] + rng.normal(0, 2, size=(H, W))
            y = np.clip(np.rint(y), 0, 255).astype(np.uint8)
            u = np.clip(np.rint(128 + 0.25 * (y[::2, ::2].astype(np.float32) - 128)), 0, 255).astype(np.uint8)
            v = np.clip(np.rint(128 - 0.25 * (y[::2, ::2].astype(np.float32) - 128)), 0, 255).astype(np.uint8)
            f.write(y.tobytes()); f.write(u.tobytes()); f.write(v.tobytes())


def syn2160p(path, n_frames=2):
    """configs[3]'s input: md5 72acbcabe33b08e22e5385af5a1fcc73 for two pictures"""
    write_clip(path, 3840, 2160, n_frames, 4321)


def syn1080p422(path, n_frames=3):
    """configs[4]'s input at its own size: the 1080p clip of bench.write_yuv (same luma, seed 1234) as planar 4:2:2 -- chroma planes 960 x 1080, taken from every
    other luma column of every row"""
    W, H, seed = 1920, 1080, 1234
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(H // 8 + 8, W // 8 + 8)).astype(np.float32)
    base = np.kron(base, np.ones((8, 8), np.float32))
    k = 5
    b = np.cumsum(np.cumsum(np.pad(base, ((k, k), (k, k)), mode="edge"), 0), 1)
    sm = (b[2 * k:, 2 * k:] - b[:-2 * k, 2 * k:] - b[2 * k:, :-2 * k] + b[:-2 * k, :-2 * k]) / (4 * k * k)
    sm = sm[:H + 64, :W + 64]
    with open(path, "wb") as f:
        for n in range(n_frames):
            dx, dy = 3 * n, 2 * n
            y = sm[dy:dy + H, dx:dx + W] + rng.normal(0, 2, size=(H, W))
            y = np.clip(np.rint(y), 0, 255).astype(np.uint8)
            u = np.clip(np.rint(128 + 0.25 * (y[:, ::2].astype(np.float32) - 128)), 0, 255).astype(np.uint8)
            v = np.clip(np.rint(128 - 0.25 * (y[:, 1::2].astype(np.float32) - 128)), 0, 255).astype(np.uint8)
            f.write(y.tobytes()); f.write(u.tobytes()); f.write(v.tobytes())
