"""The update pass alone on the queue contents of ONE real bench frame (1920x1080, guided, after 64 learning frames):
prints how the frame's updates spread over the slots, then runs mq_link_kernel + mq_apply_kernel on exactly these
records N times (mq_debug_apply_updates), so that `rocprofv3 --kernel-trace --stats -- python3 tools/apply_probe.py`
times the two kernels on the same input whatever library build MQHIP_LIB names.
Usage: python tools/apply_probe.py [repeats]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
import mqhip

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = mqhip.Context(0)
ctx.json_defaults()
for k, v in {"randomize seed": 0, "seed": 0x5EED, "spp": 1, "max path length": 3}.items():
    ctx.set_property(k, v)
ctx.synth_scene("synth_sepulcher", 2); ctx.commit(); ctx.connect(1920, 1080)
for f in range(64):
    ctx.process(ctx.synth_camera(f))
ctx.set_property("debug: log learning writes", 1)
u = ctx.synth_camera(64)
ctx.process(u)
log = ctx.learn_log()
ctx.set_property("debug: log learning writes", 0)
upd = log[log[:, 15] == 1]
ranks = upd[:, 13] >> 16
kept = upd[ranks < 10].copy()
kept[:, 15] = 0
per_slot = np.bincount(np.unique(upd[:, 14], return_counts=True)[1])
print("update records of the frame: %d proposed, %d kept (rank < 10), %d slots" % (len(upd), len(kept), len(np.unique(upd[:, 14]))))
print("slots by number of arrivals:", {int(k): int(v) for k, v in enumerate(per_slot) if v})
n_static = int(ctx.get_property("static grid buf size")); n_adaptive = int(ctx.get_property("adaptive grid buf size"))
print("slots in the static part of the table: %d of the kept records" % int((kept[:, 14] >= n_adaptive).sum()), "(table: %d adaptive + %d static states)" % (n_adaptive, n_static))
for _ in range(reps):
    ctx.apply_updates(kept, u)
ctx.sync()
print("done: %d update passes" % reps)
