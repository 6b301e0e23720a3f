import sys
sys.path.insert(0,'merian-quake_amd')
import mqhip
def run(props, label, frames=30, warm=64):
    ctx = mqhip.Context(0); ctx.json_defaults()
    base={"randomize seed":0,"seed":0x5EED,"spp":1,"max path length":3,"volume spp":0}; base.update(props)
    for k,v in base.items(): ctx.set_property(k,v)
    ctx.synth_scene("synth_sepulcher",2); ctx.commit(); ctx.connect(1920,1080)
    for f in range(warm): ctx.process(ctx.synth_camera(f))
    ctx.sync(); ctx.timing_reset()
    for f in range(warm,warm+frames): ctx.process(ctx.synth_camera(f))
    ctx.sync()
    n,r,u=ctx.timing_get(); d=ctx.timing_detail()
    print(label, {k:round(v/n,3) for k,v in d.items()}, 'apply %.3f'%(u/n)); sys.stdout.flush()
    ctx.close()
run({}, 'default tables')
run({"adaptive grid buf size":1<<20,"static grid buf size":1<<16,"LC buf size":1<<18}, 'small tables (64 MB MC, 4 MB LC)')
run({"adaptive grid buf size":1<<16,"static grid buf size":1<<12,"LC buf size":1<<14}, 'tiny tables')
