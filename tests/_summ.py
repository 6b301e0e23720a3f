import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], 'Msamples/s', d['ms_per_step'],'ms', r['kernels_ms_per_frame'])
