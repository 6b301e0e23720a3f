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
