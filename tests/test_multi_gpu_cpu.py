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


def test_bench_gpus_n_launches_n_ranks_itself(monkeypatch, capsys):
    """`python bench.py --gpus N` without a launcher around it starts N ranks with torch.distributed.run (rendezvous on
    127.0.0.1), relays rank 0's result line and insists that the line reports N ranks."""
    import importlib
    import json
    import subprocess
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None, timeout=None):
        seen["cmd"], seen["env"], seen["timeout"] = cmd, env, timeout
        if seen.get("hang"):
            raise subprocess.TimeoutExpired(cmd, timeout)
        class R:
            returncode = 0
        R.stdout = "noise from a rank\n" + json.dumps({"metric": "m", "n_gpus": seen["n"]}) + "\n"
        return R
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    seen["n"] = 4
    bench.main()
    out = capsys.readouterr()
    assert json.loads(out.out.strip())["n_gpus"] == 4 and "noise" in out.err
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert seen["timeout"] and seen["timeout"] > 0  # the child run is bounded ...
    seen["n"] = 1  # a run that silently rendered on one rank must not pass for a 4-rank result
    with pytest.raises(SystemExit):
        bench.main()
    seen["n"], seen["hang"] = 4, True  # ... and a run that does not come back is a failure, not a hang
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "did not finish" in str(e.value)


def test_bench_launcher_fails_when_a_rank_fails():
    """On this GPU-less box every rank stops with "needs a HIP device": the launcher must turn that into its own
    failure (and must get that far without touching a GPU itself)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run")
    assert r.returncode != 0 and '{"metric"' not in r.stdout
    assert "2-rank run failed" in r.stderr


def _covered(spans, lo, hi):
    rows = np.zeros(hi + 1, bool)
    for _, r0, r1 in spans:
        assert not rows[r0:r1].any(), "rows planned twice"
        rows[r0:r1] = True
    return rows


def test_band_plan_moves_exactly_the_rows_every_rank_needs():
    """The row partition of the ReSTIR node / post chain (include/mq.h "row partition"): mq_band_layout on a host-only context
    and merian-quake_amd/mq_bands.py's exchange plan.  Bands tile the image in whole tile rows; what a rank receives is exactly
    [need_begin, need_end) minus its own rows, each row from its owner; sends mirror the receives of the peers."""
    sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
    import mq_bands
    import mqhip
    ctx = mqhip.Context(-1)
    for radius, iters, halo in ((30, 1, 32), (100, 1, 64), (12, 0, 8), (30, 2, 0)):
        ctx.set_property("restir: spatital radius", radius); ctx.set_property("restir: spatial reuse iterations", iters); ctx.set_property("band: reprojection halo", halo)
        for W, H in ((3840, 2160), (1920, 1080), (328, 200), (70, 44), (64, 8)):
            for world in (1, 2, 3, 4, 8):
                bands = mq_bands.bands_of(ctx, W, H, world)
                assert bands[0].row_begin == 0 and bands[-1].row_end == H
                for r, b in enumerate(bands):
                    assert b.row_begin % 8 == 0 and (b.row_end % 8 == 0 or b.row_end == H)
                    assert r == 0 or b.row_begin == bands[r - 1].row_end
                    if b.row_end > b.row_begin:
                        rs = radius if iters > 0 and world > 1 else 0
                        assert b.reuse_begin <= max(0, b.row_begin - rs) and b.reuse_end >= min(H, b.row_end + rs)
                        assert b.need_begin <= max(0, b.reuse_begin - (halo if world > 1 else 0)) and b.need_end >= min(H, b.reuse_end + (halo if world > 1 else 0))
                    sends, recvs = mq_bands.plan(bands, r)
                    got = _covered(recvs, 0, H)
                    want = np.zeros(H + 1, bool); want[b.need_begin:b.need_end] = True; want[b.row_begin:b.row_end] = False
                    assert np.array_equal(got, want), (W, H, world, r)
                    for peer, r0, r1 in recvs:
                        assert bands[peer].row_begin <= r0 < r1 <= bands[peer].row_end
                        assert (r, r0, r1) in mq_bands.plan(bands, peer)[0]
                    for peer, r0, r1 in sends:
                        assert (r, r0, r1) in mq_bands.plan(bands, peer)[1]
    # the bytes DESIGN.md section 7 quotes: 8 ranks at 3840x2160, radius 30 (4 tile rows), halo 32: 64 rows on either side of an inner band
    ctx.set_property("restir: spatital radius", 30); ctx.set_property("restir: spatial reuse iterations", 1); ctx.set_property("band: reprojection halo", 32)
    bands = mq_bands.bands_of(ctx, 3840, 2160, 8)
    assert mq_bands.halo_bytes(bands, 3, [3840 * 64]) == 2 * 64 * 3840 * 64
    ctx.close()


def _band_worker(rank, world, port, W, H, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
    import mq_bands
    import mqhip
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = mqhip.Context(-1)
    ctx.set_property("restir: spatial reuse iterations", 1); ctx.set_property("restir: spatital radius", 12); ctx.set_property("band: reprojection halo", 16)
    bands = mq_bands.bands_of(ctx, W, H, world)
    sends, recvs = mq_bands.plan(bands, rank)
    pairs = []
    for row_bytes in (W * 64, W * 16, W * 4):  # reservoirs, accumulated image, history
        rows = torch.arange(H, dtype=torch.int64).view(H, 1)
        cols = torch.arange(row_bytes, dtype=torch.int64).view(1, row_bytes)
        truth = ((rows * 131 + cols * 7 + row_bytes) % 251).to(torch.uint8)  # what the one-rank node would hold
        send = torch.zeros(H, row_bytes, dtype=torch.uint8); recv = torch.full((H, row_bytes), 255, dtype=torch.uint8)
        b = bands[rank]
        send[b.row_begin:b.row_end] = truth[b.row_begin:b.row_end]  # a rank's outputs are valid on its own rows only
        recv[b.row_begin:b.row_end] = truth[b.row_begin:b.row_end]  # (its own rows of the previous-frame buffer it copies itself)
        pairs.append((send, recv, truth))
    mq_bands.exchange(dist, [(s, r) for s, r, _ in pairs], sends, recvs)
    b = bands[rank]
    ok = all(torch.equal(r[b.need_begin:b.need_end], t[b.need_begin:b.need_end]) for _, r, t in pairs)
    untouched = all(bool((r[:b.need_begin] == 255).all()) and bool((r[b.need_end:] == 255).all()) for _, r, _ in pairs)
    # the final image: every rank's rows to everybody with one all-gather of row blocks
    img = pairs[1][0].clone()
    mq_bands.gather_rows(dist, img, bands, rank)
    q.put((rank, ok, untouched, bool(torch.equal(img, pairs[1][2])), len(sends), len(recvs)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(2, 64, 48), (2, 70, 100), (3, 40, 120)])
def test_halo_exchange_and_row_gather_on_gloo_ranks(world, W, H):
    """merian-quake_amd/mq_bands.py on real torch.distributed ranks (gloo, CPU tensors standing for the device images): after
    ONE batch of point-to-point sends every rank holds, in its "previous frame" buffers, the rows of the other ranks it needs
    and nothing else; one all-gather of row blocks rebuilds the final image everywhere."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_band_worker, args=(r, world, port, W, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] and r[2] and r[3] for r in res), res
    assert all(r[4] > 0 and r[5] > 0 for r in res)
