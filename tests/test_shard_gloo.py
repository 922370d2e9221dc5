"""Multi-GPU path on CPU: world_size 2 and 3 over gloo.  Each rank owns one slice band of a picture; after the one collective (jm_amd.shard.BandGather: all-gather of
the un-deblocked bands and their loop-filter side information) every rank must hold the whole picture; and the closed-GOP split of a sequence gives every picture one owner."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jm_amd import shard


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _full_picture(h, w, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (h, w)).astype(np.uint8)


def test_slice_argument_matches_jm_config():
    # configs[3]: 2160p = 240 x 135 macroblocks, 8 slices -> SliceArgument 4080 (17 rows each, the last band 16)
    assert shard.slice_argument(135, 240, 8) == 4080
    bands = [shard.band_of(r, 8, 135) for r in range(8)]
    assert [b.mb_rows for b in bands] == [17] * 7 + [16]


def test_closed_gops_are_dealt_whole():
    # bench.py --gpus N: every picture has one owner, a GOP is never split, GOPs go to the ranks in turn
    for world, n, period in ((8, 200, 24), (2, 7, 3), (4, 4, 10), (3, 30, 10)):
        owner = {}
        for r in range(world):
            for first, cnt in shard.gop_of(r, world, n, period):
                assert first % period == 0 and 0 < cnt <= period
                for k in range(first, first + cnt):
                    assert k not in owner
                    owner[k] = r
        assert sorted(owner) == list(range(n))
        assert all(owner[k] == (k // period) % world for k in range(n))


def _uneven_worker(rank, world, port, mb_rows, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # configs[3]'s exchange in small: bands of k macroblock rows with a shorter last band; luma, two chroma planes, one row of loop filter
        # side information per macroblock row (28 bytes per macroblock) and four rows of per-4x4 motion per macroblock row
        wmb, pitch, cw = 5, 96, 40
        k = -(-mb_rows // world)
        shapes = [(16 * mb_rows, pitch, 16 * k), (8 * mb_rows, cw, 8 * k), (8 * mb_rows, cw, 8 * k), (mb_rows, wmb * 28, k), (4 * mb_rows, wmb * 4 * 16, 4 * k)]
        full = [_full_picture(r, p, 20 + i) for i, (r, p, _) in enumerate(shapes)]
        mine = [torch.full(p.shape, 0xEE, dtype=torch.uint8) for p in full]
        for p, m, (_, _, kk) in zip(full, mine, shapes):
            m[rank * kk:(rank + 1) * kk] = torch.from_numpy(p[rank * kk:(rank + 1) * kk])
        g = shard.BandGather([(m, kk) for m, (_, _, kk) in zip(mine, shapes)], world, rank)
        ok = True
        for rep in range(2):
            g()
            ok &= all(bool((m.numpy() == p).all()) for p, m in zip(full, mine))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mb_rows", [(2, 7), (3, 8), (2, 8)])
def test_band_gather_uneven_gloo(world, mb_rows):
    """The N > 1 leg of bench.py on configs[3]: the last band is shorter (17 ... 17, 16 macroblock rows); every rank must end up with every
    plane complete -- the un-deblocked reconstruction and the loop filter's side information -- after ONE collective."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_uneven_worker, args=(r, world, port, mb_rows, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res
