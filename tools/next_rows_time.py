"""Wall time per frame of the "next" rows on one GPU (ReSTIR DI node, accumulate + compose) beside the MCPG pass they
follow: python tools/next_rows_time.py [W H scene seed]   (default: 3840 2160 synth_azad 4 -- BASELINE config 5's frame)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
import mqhip

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
scene, seed = (sys.argv[3], int(sys.argv[4])) if len(sys.argv) > 4 else ("synth_azad", 4)
ctx = mqhip.Context(0)
ctx.json_defaults()
for k, v in {"randomize seed": 0, "seed": 0x5EED, "spp": 1, "max path length": 3, "restir: spp": 1, "restir: enable temporal reuse": 1,
             "restir: spatial reuse iterations": 1, "restir: randomize seed": 0}.items():
    ctx.set_property(k, v)
ctx.synth_scene(scene, seed); ctx.commit(); ctx.connect(W, H)

def run(frames, what):
    for f in frames:
        u = ctx.synth_camera(f)
        ctx.process(u)
        if "restir" in what: ctx.restir_process(u)
        if "post" in what: ctx.post_process()
    ctx.sync()

run(range(0, 40), ("restir", "post"))
for what in ((), ("restir",), ("post",), ("restir", "post")):
    t0 = time.perf_counter(); run(range(40, 90), what); dt = (time.perf_counter() - t0) / 50 * 1e3
    print("%dx%d %s: MCPG%s%s  %.3f ms per frame" % (W, H, scene, " + ReSTIR DI" if "restir" in what else "", " + accumulate/compose" if "post" in what else "", dt), flush=True)
ctx.close()
