import sys
sys.path.insert(0,'merian-quake_amd'); sys.path.insert(0,'tests')
import numpy as np, mqhip, orc
from test_gpu_parity import make_pair
ctx = mqhip.Context(0)
W,H,N=96,64,256
for mode in (1,0):
    make_pair(ctx,"synth_start",11,{"reference mode":mode,"spp":2},W,H)
    u=ctx.synth_camera(0)
    s=np.zeros((H,W,3)); s2=np.zeros((H,W,3)); sc=np.zeros((H,W,3))
    for f in range(N):
        u.frame=f; ctx.process(u); im=ctx.irradiance()[...,:3].astype(np.float64)
        s+=im; sc+=np.minimum(im,50); 
        if f in (0,1,3,7,15,31,63,127,255): print(mode,f,'mean so far',(s/(f+1)).mean(),'clamped',(sc/(f+1)).mean(),'nz frac last',(im.sum(-1)>0).mean(), 'max', im.max())
    print('mode',mode,'mean',(s/N).mean())
