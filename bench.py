#!/usr/bin/env python3
"""bench.py -- headline benchmark of the HIP path: Msamples/s at 1920x1080 1 spp, MCPG guiding on.

A "step" is one frame of the hot path (primary ray + guided surface estimator + Markov-chain
update application) over the whole 1920x1080 framebuffer.  With --gpus N the framebuffer's 8x8
tiles are dealt round-robin to the N ranks (one process per GPU), each rank renders its tiles with
its own scene/BVH replica and its own learning state, and one RCCL all-gather of the RGBA32F
radiance tiles rebuilds the full image on every rank: total work is fixed, so the
scaling is "strong".

Workload: the reference's `ad_sepulcher` map is not available offline (no Quake data in the image),
so the seeded synthetic stand-in `synth_sepulcher(seed=2)` is rendered with the
shipped JSON renderer defaults (res/default_config.json:599-638) except spp=1, fixed seed.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes of the render megakernel
per launch (counted by an instrumented launch of the same kernel on the same frame)
divided by its average launch duration measured with HIP events on the launch stream.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


LEARN_FRAMES = 64  # the tables learn for >= 64 untimed frames before the first timed one


def pmc_traffic(kernel, default_workload, first_frame, steps):
    """HBM bytes per launch of `kernel` over the frames [first_frame, first_frame + steps) of the fly-through, from the
    newest committed counter summary (profiles/*_pmc_summary.json: separate rocprofv3 --pmc passes over this same
    command, 1024 * (2 * FETCH_SIZE + WRITE_SIZE) per dispatch -- the gfx950 correction of MI355X_MICROARCH.md -- kept
    PER FRAME of the run).  Counters cannot be read from inside the process, so the figure is reported only when the
    summary was collected on this workload AND covers exactly this run's timed frames; None otherwise."""
    if not default_workload:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))  # named per round: the last name is the newest
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        run = d.get("_run")
        if not run or run.get("workload") != "default":
            continue
        for name, v in d.items():
            if name.startswith(kernel) and "true>" not in name.split(",")[-1] and "hbm_bytes_per_frame" in v:
                per_frame, k = v["hbm_bytes_per_frame"], v["launches_per_frame"]
                if first_frame + steps > len(per_frame):
                    return None, None
                mean = sum(per_frame[first_frame:first_frame + steps]) / steps / k
                return int(mean), "%s frames %d..%d" % (os.path.relpath(f, ROOT), first_frame, first_frame + steps - 1)
    return None, None


def algorithmic_bytes(c, pixels):
    """B = P*(40+16) + sum_rays(80 n_nodes + 48 n_tris + 120) + guided segs*(K*64) + 24*lc + U*192."""
    return (pixels * 56 + 80 * c["nodes"] + 48 * c["tris"] + 120 * c["rays"] + 64 * c["mc_state_reads"]
            + 24 * c["lc_touches"] + 192 * c["mc_updates_accepted"])


class _DevArray:
    """Zero-copy view of a device buffer for torch.as_tensor (via __cuda_array_interface__)."""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def host_cores():
    """Cores this process may really use: its affinity mask, cut to the container's CPU quota (a GPU box shows 256 hardware threads
    to a job that is given 16 cores' worth of time: 256 worker threads would only take turns)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for quota_file, period_file in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if period_file is None:
                q, p = open(quota_file).read().split()[:2]
            else:
                q, p = open(quota_file).read().strip(), open(period_file).read().strip()
            if q != "max" and float(q) > 0:
                n = min(n, max(1, int(float(q) / float(p) + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return max(1, min(256, n))


def cpu_baseline(ctx, scene, seed, props, W, H, warm, timed):
    """Naive CPU path tracer (the oracle: plain binary BVH, scalar code) on a bounded sample: a persistent pool of one worker
    thread per host core the process may run on, rows claimed dynamically, the update pass spread over the workers too."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    cores = host_cores()
    p = orc.params_from_ctx(ctx, ctx.get_constants())
    o = orc.Oracle(p)
    orc.mirror_scene(ctx, o)
    o.commit(1)
    o.connect(W, H)
    for f in range(warm):
        w0, p0 = time.perf_counter(), time.process_time()
        o.process(ctx.synth_camera(f), threads=cores, parallel_update=True)
        got = (time.process_time() - p0) / max(1e-9, time.perf_counter() - w0)
        if f == 0 and got < 0.75 * cores:  # a CPU quota no file told us about: as many workers as the job really gets to run at once
            cores = max(1, int(got + 0.5))
    t0 = time.perf_counter()
    c0 = time.process_time()
    for f in range(warm, warm + timed):
        o.process(ctx.synth_camera(f), threads=cores, parallel_update=True)
    dt = time.perf_counter() - t0
    busy = (time.process_time() - c0) / dt  # threads actually busy on average (CPU seconds per wall second)
    spp = int(ctx.get_property("spp"))
    val = W * H * spp * timed / dt / 1e6
    o.close()
    return {"value": round(val, 4), "unit": "Msamples/s", "cores": cores, "threads_busy": round(busy, 1), "kind": "port",
            "sample": "%dx%d (same scene, camera, parameters), %d warm-up + %d timed guided frames, plain binary BVH, %d pooled pthreads (%.0f busy on average), "
                      "rows and update slots claimed dynamically" % (W, H, warm, timed, cores, busy)}


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks of this script (one process per
    GPU) with torch.distributed.run BEFORE this process has touched the GPU, relay rank 0's JSON line, fail if a rank
    fails.  Nothing is re-exec'd: the ranks are fresh child processes."""
    port = free_port()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    limit = float(os.environ.get("MQ_BENCH_RANKS_TIMEOUT", "900"))  # seconds; the ranks are fresh children: killing them re-execs nothing
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=limit)
    except subprocess.TimeoutExpired:
        raise SystemExit("bench.py: the %d-rank run did not finish within %.0f s (MQ_BENCH_RANKS_TIMEOUT)" % (n, limit))
    line = None
    for l in r.stdout.splitlines():
        if l.startswith('{"metric"'):
            line = l
        else:
            print(l, file=sys.stderr)
    if r.returncode != 0 or line is None:
        raise SystemExit("bench.py: the %d-rank run failed (exit code %d%s)" % (n, r.returncode, "" if line else ", no result line"))
    got = json.loads(line)
    if got.get("n_gpus") != n:
        raise SystemExit("bench.py: asked for %d ranks, the run reports %r" % (n, got.get("n_gpus")))
    print(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="synth_sepulcher")
    ap.add_argument("--scene-seed", type=int, default=2)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--reference-mode", type=int, default=0)
    ap.add_argument("--volume-spp", type=int, default=0)
    ap.add_argument("--restir", action="store_true", help="BASELINE config 5's frame: the ReSTIR DI node (temporal + spatial reuse) and the post chain (accumulate, compose "
                    "with the ReSTIR irradiance) behind the MCPG node; on N > 1 ranks they run on row bands with a point-to-point halo exchange per frame")
    ap.add_argument("--config5", action="store_true", help="--restir on the config-5 workload: synth_azad(seed=4), 3840x2160, 1 spp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prop", action="append", default=[], metavar="KEY=VALUE", help="set a renderer property away from the workload's (A/B experiments: the line then names it)")
    ap.add_argument("--bsp", default=None, help="a user-supplied BSP29 / BSP2 map instead of the synthetic stand-in (camera at the player start); "
                    "default: $MQ_QUAKE_DIR/id1/maps/<--map>.bsp if that file exists")
    ap.add_argument("--map", default="ad_sepulcher")
    ap.add_argument("--palette", default=None, help="gfx/palette.lmp for --bsp (default: the loader's built-in grey ramp)")
    args = ap.parse_args()
    if args.config5:
        args.restir, args.scene, args.scene_seed, args.width, args.height = True, "synth_azad", 4, 3840, 2160
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)

    import torch
    import mqhip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # Rehearsal switch (tests only): all ranks share GPU 0 and the exchange is staged through gloo,
    # so the N > 1 code path can be exercised on a one-GPU box.  Never set for a measurement.
    rehearsal = os.environ.get("MQ_BENCH_REHEARSAL_ONE_GPU") in ("1", "2")  # "2": rehearse the overlapped exchange too
    if rehearsal:
        local_rank = 0
    # Second rehearsal switch (tests only): the REAL exchange path -- RCCL process group, all_gather_into_tensor on the side
    # stream, un-tiling -- with a world of ONE rank, which is all a one-GPU box can give RCCL (it refuses two ranks on one
    # device).  The frame is not partitioned; the assembled image must equal the rendered one.
    selftest = world == 1 and os.environ.get("MQ_BENCH_RCCL_SELFTEST") == "1"
    exchange = world > 1 or selftest

    def nccl_options():
        # The collectives' own stream at HIGH priority = on a hardware queue of its own: streams of one priority share a small pool of queues,
        # and an all-gather queued in line with the frame's kernels holds that queue until every peer has arrived (the same effect,
        # measured, on the upload stream of per-frame geometry: profiles/r03_z_per_frame_geometry.txt).  MQ_BENCH_NCCL_PRIORITY=0: default streams.
        if os.environ.get("MQ_BENCH_NCCL_PRIORITY") == "0":
            return None
        try:
            opts = torch.distributed.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = True
            return opts
        except Exception:
            return None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=nccl_options())
            except TypeError:  # (a torch without that keyword: default streams)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, "the process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus)
    elif selftest:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % free_port(), rank=0, world_size=1, device_id=torch.device("cuda", local_rank), pg_options=nccl_options())
    else:
        torch.cuda.set_device(local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")

    W, H = args.width, args.height
    ctx = mqhip.Context(local_rank)
    ctx.json_defaults()
    props = {"randomize seed": 0, "seed": 0x5EED, "spp": args.spp, "max path length": 3, "reference mode": args.reference_mode,
             "volume spp": args.volume_spp}  # config 3 has no volumes; config 4 (synth_tears, fog) renders them
    if args.restir:  # config 5: "ReSTIR DI + MCPG GI combined" -- both nodes on one g-buffer, their radiance summed by the `add` node
        props.update({"restir: randomize seed": 0, "restir: seed": 77, "restir: spp": 1, "restir: enable temporal reuse": 1, "restir: spatial reuse iterations": 1,
                      "add: restir irradiance": 1, "band: reprojection halo": max(64, args.height // 16)})  # the fly-through moves pixels by up to H / 18 rows per frame
    extra = {}
    for kv in args.prop:
        k, _, v = kv.partition("=")
        try:
            extra[k] = float(v)
        except ValueError:
            extra[k] = v
    props.update(extra)
    for k, v in props.items():
        ctx.set_property(k, v)
    bsp = args.bsp
    if bsp is None and os.environ.get("MQ_QUAKE_DIR"):  # real maps only when the user supplies them; nothing ships with the repository
        cand = os.path.join(os.environ["MQ_QUAKE_DIR"], "id1", "maps", args.map + ".bsp")
        bsp = cand if os.path.exists(cand) else None
    if bsp:
        ctx.load_bsp(bsp, args.palette)
        args.scene, args.scene_seed = os.path.basename(bsp), 0
    else:
        ctx.synth_scene(args.scene, args.scene_seed)
    ctx.commit()
    ctx.set_partition(rank, world)
    ctx.connect(W, H)
    stats = ctx.scene_stats()
    stream = torch.cuda.current_stream().cuda_stream
    tiles, tile_bytes = ctx.tiles_per_rank()
    tiles_ptr, _ = ctx.map_output(mqhip.OUT_TILES)
    local = gathered = vlocal = vgathered = dlocal = dgathered = None
    side = staging = vstaging = image = vimage = None
    # The exchange of frame N overlaps the rendering of frame N + 1: tiles are copied to a staging buffer on the
    # render stream, the RCCL all-gather and the un-tiling into a bench-owned image run on a side stream.
    # MQ_BENCH_SYNC_EXCHANGE=1 keeps everything on the render stream (and un-tiles into MQ_OUT_IRRADIANCE).
    # (with --restir the nodes behind the MCPG node consume the gathered image of the SAME frame: no overlap with the next one)
    overlap = exchange and not args.restir and os.environ.get("MQ_BENCH_SYNC_EXCHANGE") != "1" and (not rehearsal or os.environ.get("MQ_BENCH_REHEARSAL_ONE_GPU") == "2")
    if exchange:
        local = torch.as_tensor(_DevArray(tiles_ptr, tile_bytes // 4), device="cuda")
        gathered = torch.empty(world * (tile_bytes // 4), dtype=torch.float32, device="cuda")
        if args.volume_spp > 0:  # configs with volumes exchange the "volume" image too
            vptr, _ = ctx.map_output(mqhip.OUT_VOLUME_TILES)
            vlocal = torch.as_tensor(_DevArray(vptr, tile_bytes // 4), device="cuda")
            vgathered = torch.empty(world * (tile_bytes // 4), dtype=torch.float32, device="cuda")
            if ctx.get_property("volume forward project"):  # the projection scatters from every pixel of last frame's volume_depth: a third, small exchange (2 B/pixel; passed around as float32 words)
                dptr, dbytes = ctx.map_output(mqhip.OUT_VOLUME_DEPTH_TILES)
                dlocal = torch.as_tensor(_DevArray(dptr, dbytes // 4), device="cuda")
                dgathered = torch.empty(world * (dbytes // 4), dtype=torch.float32, device="cuda")
        if overlap:
            side = torch.cuda.Stream(priority=-1)  # (staging wait + un-tiling: beside the next frame, not in line with it)
            staging = torch.empty_like(local)
            image = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
            if vlocal is not None:
                vstaging = torch.empty_like(vlocal)
                vimage = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")

    def gather_sync(dst, src, untile):
        if rehearsal:
            torch.cuda.synchronize()
            g_cpu = torch.empty(dst.numel(), dtype=torch.float32)
            dist.all_gather_into_tensor(g_cpu, src.cpu())
            dst.copy_(g_cpu)
        else:
            dist.all_gather_into_tensor(dst, src)  # the exchange step: RCCL over xGMI
        untile(dst.data_ptr(), stream)

    bands = halo_pairs = plan = final_img = halo_side = halo_group = None
    halo_recv_bytes = 0
    if args.restir and exchange:
        import mq_bands
        # The halo rows of frame n are first read by the temporal pass of frame n + 1: the exchange (and the row gather of the final
        # image) runs on a side stream and its own communicator beside the MCPG pass of frame n + 1.  MQ_BENCH_SYNC_EXCHANGE=1: in line.
        if not rehearsal and os.environ.get("MQ_BENCH_SYNC_EXCHANGE") != "1":
            halo_side = torch.cuda.Stream(priority=-1)
            try:
                halo_group = dist.new_group(backend="nccl", pg_options=nccl_options()) if world > 1 else None
            except TypeError:
                halo_group = dist.new_group(backend="nccl") if world > 1 else None
        bands = mq_bands.bands_of(ctx, W, H, world)
        which = [mqhip.HALO_RESTIR_RESERVOIRS, mqhip.HALO_ACCUM, mqhip.HALO_ACCUM_HISTORY] + ([mqhip.HALO_VOLUME_ACCUM, mqhip.HALO_VOLUME_ACCUM_HISTORY] if args.volume_spp > 0 else [])
        halo_pairs = mq_bands.halo_tensors(ctx, H, which)
        plan = mq_bands.plan(bands, rank)
        halo_recv_bytes = mq_bands.halo_bytes(bands, rank, [p[0].shape[1] for p in halo_pairs])
        final_img = torch.as_tensor(mq_bands._DevRows(ctx.map_output(mqhip.OUT_FINAL)[0], H, W * 16), device="cuda")

    def restir_and_post(u):
        if halo_side is not None:
            torch.cuda.current_stream().wait_stream(halo_side)  # last frame's halo rows have landed (they moved beside this frame's MCPG pass)
        ctx.restir_process(u, True, stream)
        ctx.post_process(stream)
        if bands is None:
            return
        to_host = (lambda t: t.cpu()) if rehearsal else None  # one shared GPU: stage through gloo
        if rehearsal:
            torch.cuda.synchronize()
        if halo_side is not None:
            halo_side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(halo_side):
                mq_bands.exchange(dist, halo_pairs, plan[0], plan[1], group=halo_group)
                mq_bands.gather_rows(dist, final_img, bands, rank, group=halo_group)
            return
        mq_bands.exchange(dist, halo_pairs, plan[0], plan[1], stage=to_host)            # neighbours' rows of the new reservoirs / accumulated images
        mq_bands.gather_rows(dist, final_img, bands, rank, stage=to_host)               # the composed image, whole, on every rank

    def step(frame):
        u = ctx.synth_camera(frame)
        ctx.process(u, True, stream)
        if not exchange:
            if args.restir:
                restir_and_post(u)
            return
        if not overlap:
            gather_sync(gathered, local, ctx.untile)
            if vlocal is not None:
                gather_sync(vgathered, vlocal, ctx.untile_volume)
            if dlocal is not None:
                gather_sync(dgathered, dlocal, ctx.untile_volume_depth)
            if args.restir:
                restir_and_post(u)
            return
        main = torch.cuda.current_stream()
        main.wait_stream(side)          # the previous frame's exchange (it ran beside this frame's kernels) is done
        staging.copy_(local)
        if vlocal is not None:
            vstaging.copy_(vlocal)
        if dlocal is not None:  # (needed by the NEXT frame's projection: in line, it is 1 / 8 of the radiance exchange)
            gather_sync(dgathered, dlocal, ctx.untile_volume_depth)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for dst, src, img in ((gathered, staging, image), (vgathered, vstaging, vimage)):
                if src is None:
                    continue
                if rehearsal:  # one shared GPU: stage the gather through gloo
                    side.synchronize()
                    g_cpu = torch.empty(dst.numel(), dtype=torch.float32)
                    dist.all_gather_into_tensor(g_cpu, src.cpu())
                    dst.copy_(g_cpu)
                else:
                    dist.all_gather_into_tensor(dst, src)
                ctx.untile_to(dst.data_ptr(), img.data_ptr(), side.cuda_stream)

    def sync_all():
        if exchange:
            dist.barrier()
        torch.cuda.synchronize()

    # The tables always learn over the same >= LEARN_FRAMES untimed frames of the fly-through before the first timed one,
    # whatever --warmup says (a short warm-up would time barely learned tables on a lighter stretch of the path).
    learn = max(args.warmup, LEARN_FRAMES)
    frame = 0
    for _ in range(learn):
        step(frame); frame += 1
    sync_all()
    ctx.timing_set_interval(4)  # per-launch events (the kernel times below) on every 4th timed frame: an event between two launches delays the second
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(frame); frame += 1
    sync_all()
    dt = time.perf_counter() - t0
    if ctx.counters()["queue_overflow"] != 0:
        raise SystemExit("bench.py: the run raised overflow flags %d (a queue ran out of room, or -- bit 3 -- a reprojected pixel left the rows a rank holds: "
                         "raise \"band: reprojection halo\"): the frames are not valid" % ctx.counters()["queue_overflow"])
    if overlap and (rehearsal or selftest):  # rehearsals only: the assembled image must contain this rank's tiles of the last frame
        import mq_tiles
        got = mq_tiles.tile_image(image.cpu().numpy().reshape(H, W, 4), rank, world)
        assert np.array_equal(got, local.cpu().numpy().reshape(-1, 64, 4)), "overlapped exchange lost tiles"
    # device time of the render megakernel, averaged over exactly the timed launches (hipEvents
    # recorded on the launch stream inside mq_process)
    n_timed, render_sum, update_sum = ctx.timing_get()
    assert n_timed == args.steps
    det = ctx.timing_detail()
    per_round = ctx.timing_rounds()
    n_detail = ctx.timing_detail_frames()
    ctx.timing_set_interval(1)
    if exchange:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = W * H * args.spp * args.steps / dt / 1e6

    # algorithmic bytes: instrumented launches of the same kernels on the cameras of four timed frames (the middles of the
    # quarters of the timed region), averaged -- the fly-through gets heavier along its path, so the one frame after the
    # timed region would overstate the bytes of the average timed launch
    ctx.enable_counters(True)
    c = None
    n_counted = 4
    for k in range(n_counted):
        step(learn + (2 * k + 1) * args.steps // (2 * n_counted))
        torch.cuda.synchronize()
        ck = ctx.counters()
        c = ck if c is None else {key: c[key] + v for key, v in ck.items()}
    c = {key: (v if key == "queue_overflow" else v // n_counted) for key, v in c.items()}
    ctx.enable_counters(False)
    local_pixels = c["pixels"]
    # the interval after the surface pass: link + apply, and (configs with volume samples) the volume passes
    upd_name = "mq_apply_kernel" if args.volume_spp <= 0 else "mq_apply_kernel + volume passes"
    kms = {"mq_primary_kernel": det["primary_ms"] / n_detail, "mq_trace_queue_kernel": det["trace_ms"] / n_detail,
           "mq_bounce_kernel": det["bounce_ms"] / n_detail, upd_name: update_sum / n_timed}
    rounds = args.spp * 2  # spp * (max path length - 1) launches of trace + bounce per frame
    # algorithmic bytes per kernel class (SURVEY.md 8d prices)
    prim_rays = c["rays"] - c["queue_rays"]
    bounce_rays = c["queue_rays"]
    kbytes = {
        "mq_primary_kernel": local_pixels * 56 + 80 * (c["nodes"] - c["queue_nodes"]) + 48 * (c["tris"] - c["queue_tris"]) + 120 * prim_rays,
        "mq_trace_queue_kernel": 80 * c["queue_nodes"] + 48 * c["queue_tris"] + 48 * c["queue_rays"],   # + 32 B ray in, 16 B hit out
        "mq_bounce_kernel": bounce_rays * (120 + 16 + 4 + 320) + 64 * c["mc_state_reads"] + 24 * c["lc_touches"] + 64 * c["mc_updates_accepted"],
        upd_name: 128 * c["mc_updates_accepted"],
    }
    B = algorithmic_bytes(c, local_pixels)
    launches = {"mq_primary_kernel": 1, "mq_trace_queue_kernel": rounds, "mq_bounce_kernel": rounds, upd_name: 1}
    dom = max(("mq_trace_queue_kernel", "mq_bounce_kernel", "mq_primary_kernel"), key=lambda k: kms[k])  # most device time per frame
    dom_ms_per_launch = kms[dom] / launches[dom]
    dom_bytes_per_launch = kbytes[dom] / launches[dom]
    achieved = dom_bytes_per_launch / (dom_ms_per_launch * 1e-3) / 1e9
    pipeline_ms = render_sum / n_timed
    # device time of a whole frame: surface pass + update pass (+ volume passes).  The camera rays of the NEXT frame run beside
    # the update pass ("overlap camera rays"), so the surface-pass interval alone no longer holds all of a frame's work.
    frame_ms = (render_sum + update_sum) / n_timed
    default_workload = (world == 1 and not selftest and (W, H) == (1920, 1080) and args.scene == "synth_sepulcher" and args.scene_seed == 2
                        and args.spp == 1 and not args.reference_mode and args.volume_spp == 0 and not extra and not args.restir)
    traffic, traffic_src = pmc_traffic(dom, default_workload, learn, args.steps)
    # `achieved` / `frac`: ALGORITHMIC bytes of the dominant kernel per launch over its measured launch time (the contract's
    # definition).  `traffic` is what the HBM actually moved per launch (PMC); `hbm_measured` prices that against the peak:
    # the kernel is bound by the L1 gather path and VALU issue, not by HBM -- most of its node and
    # triangle reads are served by L2 / Infinity Cache.
    hbm_measured = None if traffic is None else {"GB/s": round(traffic / (dom_ms_per_launch * 1e-3) / 1e9, 1), "frac": round(traffic / (dom_ms_per_launch * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    # "bound": what the counters say limits the dominant kernel (profiles/*_unit_busy.json: texture-data unit busy 0.79, vector ALU
    # active 0.74 of the CU cycles; HBM at 5 % of its peak).  `achieved` / `frac` stay the contract's figure: ALGORITHMIC bytes over
    # the kernel time against the HBM peak (`contract_bound`), the same number under the name that says what it is (`frac_algorithmic`).
    roofline = {"bound": "l1-gather+valu", "contract_bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_algorithmic": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src, "hbm_measured": hbm_measured,
                "limiter": "L1 gather path (TA/TD busy) and VALU issue; measured HBM traffic is far below the algorithmic bytes (L2 / Infinity Cache hits)", "kernel": dom,
                "overlap": ("none" if ctx.get_property("overlap camera rays") == 0 else "the camera rays of frame n+1 run on a low-priority stream beside the terminal bounce kernel and the update pass of frame n (a full frame) or beside its whole last round (a rank of a "
                            "partitioned frame): property \"overlap camera rays\"; the intervals below are those of the launch stream"),
                "achievable_peak": round(ctx.measure_stream_read(), 1),  # streaming read of 2 GiB on this GPU, GB/s (the 8 TB/s above is the spec figure)
                "kernel_ms_per_launch": round(dom_ms_per_launch, 4), "launches_per_frame": launches[dom], "kernel_timed_frames": n_detail,
                "algorithmic_bytes_per_launch": int(dom_bytes_per_launch),
                "frame": {"algorithmic_bytes": int(B), "frame_ms": round(frame_ms, 4), "render_ms": round(pipeline_ms, 4), "achieved": round(B / (frame_ms * 1e-3) / 1e9, 1),
                          "frac": round(B / (frame_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_sample": round(B / max(1, local_pixels * args.spp), 1)},
                "kernels_ms_per_frame": {k: round(v, 4) for k, v in kms.items()},
                "launches_ms": [[round(a / n_detail, 4), round(b / n_detail, 4)] for a, b in per_round[:rounds + 1]],  # [trace, shade] per round; entry 0 = primary
                "kernels_algorithmic_bytes_per_frame": {k: int(v) for k, v in kbytes.items()}, "counters": c}

    out = {"metric": "Msamples/s at 1920x1080 1spp (ad_sepulcher); per-pixel L2 vs reference", "value": round(value, 3),
           "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "learning_frames": learn, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "real" if bsp else "synthetic",
           "config": {"workload": "%s, %dx%d %dspp, MCPG guiding %s, max path length 3, JSON renderer defaults%s"
                                  % ("user-supplied map %s, camera at the player start" % args.scene if bsp else "%s(seed=%d) stand-in for ad_sepulcher" % (args.scene, args.scene_seed),
                                     W, H, args.spp, "off (reference mode)" if args.reference_mode else "on",
                                     " + ReSTIR DI node (1 candidate, temporal + spatial reuse) + accumulate / compose (config 5)" if args.restir else ""),
                      **({"properties_changed": extra} if extra else {}), "triangles": stats["n_tris"], "bvh_nodes": stats["n_nodes"], "parallelism": "tiles%d" % world, "ranks": world, "backend": (dist.get_backend() if exchange else "none"),
                      "collective": "none" if not exchange else "%dx RCCL all_gather of %d B/rank per frame%s%s" % ((3 if dlocal is not None else 2) if args.volume_spp > 0 else 1, tile_bytes, ", overlapped with the next frame" if overlap else "",
                                    "" if not args.restir else "; row bands for the ReSTIR node / post chain: point-to-point halo rows, %d B received per frame by rank 0, + all_gather of the final image's rows (%d B/rank)"
                                    % (halo_recv_bytes, max(b.row_end - b.row_begin for b in bands) * W * 16))},
           "roofline": roofline}
    if exchange:
        dist.barrier()  # every rank is done with the GPU: the CPU leg below has the host to itself
    if rank == 0 and not args.no_cpu_baseline:  # (N > 1 too: north_star wants the CPU figure in the same run; the other ranks wait in the barrier below)
        out["cpu_baseline"] = cpu_baseline(ctx, args.scene, args.scene_seed, props, W, H, 3, 8)  # the headline frame size: ~25 s of host time
    if rank == 0:
        print(json.dumps(out))
    if exchange:
        if rehearsal:
            dist.barrier()
        else:
            torch.cuda.synchronize()
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
