"""N > 1 path on CPU: two `gloo` ranks shard a frame by the product's tile layout, exchange their
tile buffers with ONE all_gather_into_tensor (the path's only collective) and rebuild
the image.  The device kernels that produce / consume the same layout are covered by
tests/test_gpu_multi.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, W, H, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
    import mq_tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(42)
    full = rng.random((H, W, 4), dtype=np.float32)  # what a 1-rank render would produce
    local = torch.from_numpy(mq_tiles.tile_image(full, rank, world).reshape(-1).copy())
    gathered = torch.empty(world * local.numel(), dtype=torch.float32)
    dist.all_gather_into_tensor(gathered, local)
    out = mq_tiles.untile(gathered.numpy(), W, H, world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's max-over-ranks timing reduction
    q.put((rank, bool(np.array_equal(out, full)), float(t.item())))
    dist.destroy_process_group()


@pytest.mark.parametrize("W,H", [(64, 48), (70, 44)])
def test_two_rank_tile_exchange_rebuilds_the_frame(W, H):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, W, H, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res) and all(r[2] == 2.0 for r in res)


def test_partition_covers_every_tile_once():
    sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
    import mq_tiles
    for W, H in ((1920, 1080), (320, 240), (70, 44)):
        tx, ty = mq_tiles.grid(W, H)
        for world in (1, 2, 4, 8):
            allt = np.concatenate([mq_tiles.local_tiles(W, H, r, world) for r in range(world)])
            assert sorted(allt) == list(range(tx * ty))
            sizes = [len(mq_tiles.local_tiles(W, H, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == mq_tiles.tiles_per_rank(W, H, world)
    assert mq_tiles.tiles_per_rank(1920, 1080, 8) * 64 * 16 == 4147200  # 4.15 MB per rank at 1080p / 8 GPUs
