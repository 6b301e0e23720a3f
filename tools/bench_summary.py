import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; c=r['counters']
        print(d['value'], 'Msamples/s', d['ms_per_step'],'ms', r['kernels_ms_per_frame'], r.get('launches_ms'), 'nodes/ray %.1f tris/ray %.1f' % (c['queue_nodes']/max(1,c['queue_rays']), c['queue_tris']/max(1,c['queue_rays'])), d['config'].get('bvh_nodes'))
