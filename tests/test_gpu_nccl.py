"""gpu, needs two devices: the N > 1 exchange of DESIGN.md section 5 over RCCL itself (backend "nccl"), one process per GPU -- so that the first run on an 8-GPU node does not
meet RCCL for the first time.  Each rank owns one band of a picture on ITS device; after jm_amd.shard.BandGather's one all-gather every rank must hold the whole picture.
(The same worker runs over gloo on CPU tensors in tests/test_shard_gloo.py; on a one-GPU box this test is skipped.)"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, mb_rows, q):
    import torch.distributed as dist
    from jm_amd import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", rank)
        wmb, pitch, cw = 5, 96, 40
        k = -(-mb_rows // world)
        shapes = [(16 * mb_rows, pitch, 16 * k), (8 * mb_rows, cw, 8 * k), (8 * mb_rows, cw, 8 * k), (mb_rows, wmb * 28, k), (4 * mb_rows, wmb * 4 * 16, 4 * k)]
        full = [np.random.default_rng(20 + i).integers(0, 256, (r, p)).astype(np.uint8) for i, (r, p, _) in enumerate(shapes)]
        mine = [torch.full(p.shape, 0xEE, dtype=torch.uint8, device=dev) for p in full]
        for p, m, (_, _, kk) in zip(full, mine, shapes):
            m[rank * kk:(rank + 1) * kk] = torch.from_numpy(p[rank * kk:(rank + 1) * kk]).to(dev)
        g = shard.BandGather([(m, kk) for m, (_, _, kk) in zip(mine, shapes)], world, rank)
        g()
        torch.cuda.synchronize()
        ok = all(np.array_equal(m.cpu().numpy(), p) for m, p in zip(mine, full))
        t = torch.tensor([1.0 + rank], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                # bench.py's one collective: the slowest rank's time
        q.put((rank, ok and float(t.item()) == float(world)))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU node); tests/test_shard_gloo.py covers the same exchange on CPU")
def test_band_gather_over_rccl_two_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, 9, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == [(0, True), (1, True)], res
