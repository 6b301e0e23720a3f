"""Multi-GPU sharding on ONE GPU: two contexts play rank 0 and rank 1 of a world of 2; their tile
buffers are concatenated as an all-gather would and scattered by mq_untile.  In reference mode the
result is bit-identical to the 1-GPU image."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scene,W,H,world", [("synth_tiny", 96, 64, 2), ("synth_tiny", 70, 44, 3),
                                             ("synth_sepulcher", 1920, 1080, 8)])  # the bench frame as 8 ranks render it
def test_sharded_frame_equals_single_gpu_frame(mqlib, scene, W, H, world):
    import torch
    import mqhip
    sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
    import mq_tiles
    props = {"reference mode": 1, "randomize seed": 0, "seed": 0x5EED, "spp": 2, "adaptive grid buf size": 1 << 16,
             "static grid buf size": 1 << 12, "LC buf size": 1 << 14}

    def make(rank, nranks):
        c = mqhip.Context(0)
        c.synth_scene(scene, 3)
        for k, v in props.items():
            c.set_property(k, v)
        c.commit(); c.set_partition(rank, nranks); c.connect(W, H)
        return c
    single = make(0, 1)
    u = single.synth_camera(2)
    single.process(u)
    full = single.irradiance()
    assert full[..., :3].sum() > 0
    ranks = [make(r, world) for r in range(world)]
    bufs = []
    for c in ranks:
        c.process(u)
        tiles, nbytes = c.tiles_per_rank()
        assert tiles == mq_tiles.tiles_per_rank(W, H, world)
        buf = c.read_output(mqhip.OUT_TILES).view(np.float32)
        assert buf.size * 4 == nbytes
        bufs.append(buf)
    gathered = np.concatenate(bufs)
    # host mirror of the layout agrees with the device layout
    assert np.array_equal(mq_tiles.untile(gathered, W, H, world), full)
    for r, c in enumerate(ranks):
        assert np.array_equal(bufs[r].reshape(-1, 64, 4), mq_tiles.tile_image(full, r, world))
    # device untile (what bench.py runs after the RCCL all-gather)
    g = torch.from_numpy(gathered).cuda()
    ranks[0].untile(g.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(ranks[0].irradiance(), full)
    for c in ranks + [single]:
        c.close()


def test_volume_tiles_reassemble_the_single_gpu_volume_image(mqlib):
    """Config-4 style frame (fog + volume samples, learning inputs off so the frame is deterministic): the
    per-rank MQ_OUT_VOLUME_TILES buffers, gathered and untiled, equal the 1-GPU `volume` image."""
    import torch
    import mqhip
    sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
    import mq_tiles
    W, H, world = 100, 52, 3  # ragged: partial tiles on both edges, tiles not divisible by the world size
    props = {"reference mode": 1, "randomize seed": 0, "seed": 0x5EED, "spp": 1, "mc samples": 0, "dist mc samples": 0, "volume spp": 2, "particle size": 7.0,
             "volume: use LC": 1, "dist guide p": 0.9, "Phase Prob": 0.1, "volume forward project": 0}
    def make(rank, nranks):
        c = mqhip.Context(0)
        c.header_defaults()
        c.synth_scene("synth_tiny_fog", 3)
        for k, v in props.items():
            c.set_property(k, v)
        c.commit(); c.set_partition(rank, nranks); c.connect(W, H)
        return c
    single = make(0, 1)
    u = single.synth_camera(2)
    single.process(u)
    full = single.volume()
    assert full[..., :3].sum() > 0
    ranks = [make(r, world) for r in range(world)]
    bufs = []
    for c in ranks:
        c.process(u)
        bufs.append(c.read_output(mqhip.OUT_VOLUME_TILES).view(np.float32))
    gathered = np.concatenate(bufs)
    assert np.array_equal(mq_tiles.untile(gathered, W, H, world), full)
    g = torch.from_numpy(gathered).cuda()
    ranks[0].untile_volume(g.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(ranks[0].volume(), full)


def test_timing_interval_and_images_do_not_depend_on_it(mqlib):
    """Per-launch timing events on every k-th frame only (mq_timing_set_interval): the frame count and the totals cover
    all frames, the per-kernel split the sampled ones, and the rendered images are the same either way."""
    import mqhip
    imgs = []
    for every in (1, 3):
        ctx = mqhip.Context(0)
        ctx.header_defaults()
        ctx.synth_scene("synth_tiny", 2)
        for k, v in {"randomize seed": 0, "seed": 0x5EED, "reference mode": 1, "adaptive grid buf size": 1 << 18, "static grid buf size": 1 << 14, "LC buf size": 1 << 16}.items():
            ctx.set_property(k, v)
        ctx.commit(); ctx.connect(96, 64)
        ctx.timing_set_interval(every); ctx.timing_reset()
        for f in range(7):
            ctx.process(ctx.synth_camera(f))
        n, render, update = ctx.timing_get()
        det = ctx.timing_detail()
        assert n == 7 and render > 0 and update >= 0
        assert ctx.timing_detail_frames() == (7 if every == 1 else 3)  # frames 0, 3, 6
        assert 0 < det["primary_ms"] + det["trace_ms"] + det["bounce_ms"] <= render * 1.001
        imgs.append(ctx.irradiance().copy())
        ctx.close()
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))


def test_bench_exchange_path_on_real_rccl_with_one_rank(mqlib):
    """`MQ_BENCH_RCCL_SELFTEST=1`: bench.py's exchange code (RCCL process group, all_gather_into_tensor of the tile buffer on
    the side stream overlapped with the next frame, device un-tiling, the MAX all-reduce of the time) with the largest world
    a one-GPU box can give RCCL: one rank.  bench.py itself asserts that the assembled image equals the rendered tiles."""
    import json
    import subprocess
    env = dict(os.environ, MQ_BENCH_RCCL_SELFTEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                        "--width", "640", "--height", "360", "--scene", "synth_start", "--scene-seed", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["backend"] == "nccl" and "all_gather" in line["config"]["collective"]
    assert line["value"] > 0


def test_bench_line_carries_the_contract_fields(mqlib):
    """One JSON line with the fields the driver and the judge read (small frame, short run, CPU baseline included)."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--width", "320", "--height", "200",
                        "--scene", "synth_start", "--scene-seed", "1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["unit"] == "Msamples/s" and b["n_gpus"] == 1 and b["steps"] == 6 and b["higher_is_better"] is True and b["vs_baseline"] is None
    assert b["learning_frames"] >= 64 and "workload" in b["config"] and "model" not in b["config"]
    assert abs(b["value"] - 320 * 200 * 6 / (b["ms_per_step"] * 6e-3) / 1e6) < 0.02 * b["value"]
    rf = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_per_launch"):
        assert k in rf, k
    assert rf["bound"] == "l1-gather+valu" and rf["contract_bound"] == "hbm" and rf["frac_algorithmic"] == rf["frac"] and rf["peak"] == 8000.0  # the measured limiter; the algorithmic figure stays priced against the HBM peak and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None  # not the workload the committed counter summary was collected on
    cb = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "threads_busy"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1


def test_cpp_node_adapter_renders_on_the_device(mqlib, tmp_path):
    """include/mq_node.hpp -- the five-method classes a merian maintainer subclasses -- drive three frames of the fused
    GBuffer + MCPG node, the ReSTIR node and the post chain on the GPU (tests/node_adapter_gpu_test.cpp); the images they
    leave equal those of the same frames rendered through the ctypes binding.  Built with hipcc (the box may lack g++)."""
    import shutil
    import subprocess
    import mqhip
    cc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(cc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "node_adapter_gpu_test")
    libdir = os.path.join(ROOT, "merian-quake_amd", "lib")
    b = subprocess.run([cc, "-std=c++17", "-O1", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                        os.path.join(ROOT, "tests", "node_adapter_gpu_test.cpp"), "-L" + libdir, "-lmqhip", "-L/opt/rocm/lib", "-lamdhip64",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert b.returncode == 0, b.stdout[-3000:]
    r = subprocess.run([exe, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "node adapter gpu ok" in r.stdout, r.stdout[-3000:]
    W, H = 64, 48
    ctx = mqhip.Context(0)
    ctx.header_defaults()
    ctx.synth_scene("synth_tiny", 3)
    for k, v in {"reference mode": 1, "randomize seed": 0, "seed": 0x5EED, "spp": 2, "max path length": 3, "adaptive grid buf size": 1 << 16,
                 "static grid buf size": 1 << 12, "LC buf size": 1 << 14, "restir: randomize seed": 0, "restir: seed": 77, "restir: spp": 2,
                 "restir: enable temporal reuse": 1, "restir: spatial reuse iterations": 2}.items():
        ctx.set_property(k, v)
    ctx.commit(); ctx.connect(W, H)
    for f in range(3):
        u = ctx.synth_camera(10 + f)
        ctx.process(u); ctx.restir_process(u); ctx.post_process()
    for name, which in (("irradiance", mqhip.OUT_IRRADIANCE), ("hits", mqhip.OUT_HITS), ("restir_irradiance", mqhip.OUT_RESTIR_IRRADIANCE), ("final", mqhip.OUT_FINAL)):
        got = np.fromfile(str(tmp_path / (name + ".bin")), np.uint8)
        want = ctx.read_output(which).view(np.uint8).reshape(-1)
        assert np.array_equal(got, want), name
    assert ctx.irradiance()[..., :3].sum() > 0 and ctx.image(mqhip.OUT_FINAL)[..., :3].sum() > 0
    ctx.close()


def test_bench_restir_line_on_one_rank(mqlib):
    """`bench.py --restir`: config 5's frame (MCPG + ReSTIR DI node + accumulate / compose) as the timed step, one rank."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--restir", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                        "--width", "640", "--height", "360", "--scene", "synth_start", "--scene-seed", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and "ReSTIR DI" in line["config"]["workload"] and line["value"] > 0 and line["roofline"]["traffic"] is None


def test_bench_restir_two_ranks_rehearsed_on_one_gpu(mqlib):
    """`bench.py --gpus 2 --restir` end to end with both ranks on this box's one GPU (MQ_BENCH_REHEARSAL_ONE_GPU: gloo-staged
    exchange, never a measurement): tile all-gather, row bands for the ReSTIR node and the post chain, the point-to-point halo
    exchange of merian-quake_amd/mq_bands.py, the row gather of the final image -- and no overflow flag (bench.py fails on one)."""
    import json
    import subprocess
    env = dict(os.environ, MQ_BENCH_REHEARSAL_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--restir", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                        "--width", "328", "--height", "200", "--scene", "synth_start", "--scene-seed", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stderr[-3000:], r.stdout[-500:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["backend"] == "gloo" and "halo rows" in line["config"]["collective"] and line["value"] > 0


@pytest.mark.parametrize("mode", ["1", "2"])
def test_bench_two_ranks_rehearsed_on_one_gpu(mqlib, mode):
    """`python bench.py --gpus 2` end to end -- self-launch of the ranks, tile partition, exchange (mode 1: in line; mode 2: on the
    side stream, overlapped with the next frame, with bench.py's own check that the assembled image holds the rank's tiles),
    MAX all-reduce of the time, rank 0's line with the CPU baseline (every N carries it) -- with both ranks on this box's one GPU
    and the exchange staged through gloo (`MQ_BENCH_REHEARSAL_ONE_GPU`; never a measurement)."""
    import json
    import subprocess
    env = dict(os.environ, MQ_BENCH_REHEARSAL_ONE_GPU=mode, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--width", "320", "--height", "200",
                        "--scene", "synth_start", "--scene-seed", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stderr[-3000:], r.stdout[-500:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["backend"] == "gloo" and line["value"] > 0
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
    assert ("overlapped" in line["config"]["collective"]) == (mode == "2")


def test_partitioned_frame_forward_projects_from_every_pixel(mqlib):
    """Config 4 on N ranks with the JSON default `"volume forward project": true`: the projection is a scatter from EVERY pixel of
    last frame's `volume_depth` (render_mcpg.cpp:296-311), and a rank renders only its tiles.  With the third exchange buffer
    (MQ_OUT_VOLUME_DEPTH_TILES, gathered and scattered back by mq_untile_volume_depth) every rank projects from the whole image:
    guided volume frames from a given state -- their distance lookups follow `volume_mv` -- are bit-identical to the one-rank
    frames in the `volume` tiles and in `volume_mv` at the rank's pixels."""
    import torch
    import mqhip
    sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
    import mq_tiles
    W, H, world = 200, 120, 3
    props = {"reference mode": 0, "randomize seed": 0, "seed": 0x5EED, "spp": 1, "max path length": 3, "volume spp": 2, "particle size": 7.0, "volume: use LC": 1,
             "dist guide p": 0.9, "Phase Prob": 0.1, "volume forward project": 1, "adaptive grid buf size": 1 << 16, "static grid buf size": 1 << 12, "LC buf size": 1 << 14}

    def make(rank, nranks):
        c = mqhip.Context(0)
        c.header_defaults()
        c.synth_scene("synth_start_fog", 3)
        for k, v in props.items():
            c.set_property(k, v)
        c.commit(); c.set_partition(rank, nranks); c.connect(W, H)
        return c
    teacher = make(0, 1)
    for f in range(8):
        teacher.process(teacher.synth_camera(f))
    n_mc, n_lc = props["adaptive grid buf size"] + props["static grid buf size"], props["LC buf size"]
    state = (teacher.state_read(0, n_mc), teacher.state_read(1, n_lc), teacher.state_read(2, (W // 25 + 2) * (H // 25 + 2) * 10))  # render_mcpg.cpp:80-82
    assert (state[2]["N"] > 0).sum() > 50
    teacher.close()
    single, ranks = make(0, 1), [make(r, world) for r in range(world)]
    def exchange_depth():
        depth = torch.cat([torch.from_numpy(c.read_output(mqhip.OUT_VOLUME_DEPTH_TILES).view(np.float32).copy()) for c in ranks]).cuda()
        for c in ranks:
            c.untile_volume_depth(depth.data_ptr())
        torch.cuda.synchronize()
    for c in [single] + ranks:
        c.set_property("debug: freeze learning", 1)  # (from the first frame on: a free-running frame's distance chains depend on the order its pixels ran in)
        c.process(c.synth_camera(8))  # the first frame zeroes the tables
        for which, st in enumerate(state):
            c.state_write(which, st)
    exchange_depth()
    moved = 0
    for f in (10, 12, 14, 16):
        u = single.synth_camera(f)
        single.process(u)
        for c in ranks:
            c.process(u)
        exchange_depth()
        full, full_mv = single.volume(), single.read_output(mqhip.OUT_VOLUME_MV).view(np.uint32).reshape(H, W)
        assert np.array_equal(ranks[0].read_output(mqhip.OUT_VOLUME_DEPTH), single.read_output(mqhip.OUT_VOLUME_DEPTH)), f
        for r, c in enumerate(ranks):
            got = c.read_output(mqhip.OUT_VOLUME_TILES).view(np.float32).reshape(-1, 64, 4)
            want = mq_tiles.tile_image(full, r, world)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "frame %d rank %d: volume tiles differ in %d pixels" % (f, r, (got != want).any(-1).sum())
            mv = c.read_output(mqhip.OUT_VOLUME_MV).view(np.uint32).reshape(H, W)
            mine = mq_tiles.tile_image(np.repeat(mv[..., None], 4, -1), r, world)[..., 0], mq_tiles.tile_image(np.repeat(full_mv[..., None], 4, -1), r, world)[..., 0]
            assert np.array_equal(mine[0], mine[1]), (f, r)
        moved += int((full_mv != single.read_output(mqhip.OUT_GB_MV).view(np.uint32).reshape(H, W)).sum())
        assert full[..., :3].sum() > 0
    assert moved > 500, moved  # the projection rewrote vectors
    for c in [single] + ranks:
        c.close()
