"""Kernel timeline of a few frames (run under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/timeline.py`),
then `python3 tools/timeline.py --read DIR` prints start offsets / durations / queues of the last frames' kernels:
shows whether launches on different streams really run side by side.
Environment: MQ_WORLD (tile partition emulated on one GPU, default 8), MQ_PIPELINES, MQ_OVERLAP, MQ_FRAMES (default 40), MQ_PARTICLES
(that many fresh particles produced and committed before every frame: the per-frame geometry path; add --memory-copy-trace to see its uploads)."""
import os, sys, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    rows = []
    for fn in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        with open(fn) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    for fn in glob.glob(os.path.join(sys.argv[2], "**", "*memory_copy_trace.csv"), recursive=True):
        with open(fn) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?").replace("MEMORY_COPY_", ""), "-", r.get("Stream_Id", "?")))
    rows.sort()
    last = int(os.environ.get("MQ_SHOW", "40"))
    rows = rows[-last:]
    t0 = rows[0][0]
    prev_end = t0
    for s, e, name, q, st in rows:
        print("%9.1f us  +%7.1f us  gap %6.1f  q%s s%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, st, name))
        prev_end = max(prev_end, e)
    sys.exit(0)

import mqhip
ctx = mqhip.Context(0)
ctx.json_defaults()
for k, v in {"randomize seed": 0, "seed": 0x5EED, "spp": 1, "max path length": 3}.items():
    ctx.set_property(k, v)
if os.environ.get("MQ_PIPELINES"):
    ctx.set_property("pipelines", int(os.environ["MQ_PIPELINES"]))
if os.environ.get("MQ_OVERLAP"):
    ctx.set_property("overlap camera rays", int(os.environ["MQ_OVERLAP"]))
ctx.synth_scene("synth_sepulcher", 2); ctx.commit()
ctx.set_partition(0, int(os.environ.get("MQ_WORLD", "8"))); ctx.connect(1920, 1080)
P = int(os.environ.get("MQ_PARTICLES", "0"))
if P:
    import numpy as np
    rng = np.random.default_rng(5)
    u0 = ctx.synth_camera(0)
    parts = np.zeros(P, mqhip.PARTICLE_DTYPE)
    parts["org"] = np.array([u0.cam_x[0], u0.cam_x[1], u0.cam_x[2]]) + rng.uniform(-200, 200, (P, 3)); parts["vel"] = rng.uniform(-30, 30, (P, 3))
    parts["seed"] = rng.integers(1, 2 ** 32, P); parts["color_rgba"] = 0x00ffffff
    view = mqhip.View()
    for k in range(3):
        view.origin[k] = u0.cam_x[k]; view.forward[k] = u0.cam_w[k]; view.up[k] = u0.cam_u[k]
    view.right[1] = -1.0
for f in range(int(os.environ.get("MQ_FRAMES", "40"))):
    if P:
        parts["prev_org"] = parts["org"]; parts["org"] = parts["org"] + parts["vel"] / 60.0
        ctx.dyn_begin(); ctx.dyn_add_particles(parts, view, 1, 2, f / 60.0, (f - 1) / 60.0); ctx.dyn_end(2)
        ctx.commit()
    ctx.process(ctx.synth_camera(f))
ctx.sync()
ctx.close()
