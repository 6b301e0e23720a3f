import sys
sys.path.insert(0,'merian-quake_amd')
import mqhip
def run(abl, label, frames=30, warm=64):
    ctx = mqhip.Context(0); ctx.json_defaults()
    for k,v in {"randomize seed":0,"seed":0x5EED,"spp":1,"max path length":3}.items(): ctx.set_property(k,v)
    ctx.synth_scene("synth_sepulcher",2); ctx.commit(); ctx.connect(1920,1080)
    for f in range(warm): ctx.process(ctx.synth_camera(f))
    ctx.sync()
    if abl: ctx.set_property("debug output", abl)
    ctx.process(ctx.synth_camera(warm)); ctx.sync(); ctx.timing_reset()
    for f in range(warm+1,warm+1+frames): ctx.process(ctx.synth_camera(warm+1))
    ctx.sync()
    n,r,u=ctx.timing_get(); d=ctx.timing_detail()
    print(label, {k:round(v/n,3) for k,v in d.items()}); sys.stdout.flush()
    ctx.close()
run(0,'full')



run(106,'full no enqueue')

run(108,'no fast recovery store')
run(109,'no count atomic')
