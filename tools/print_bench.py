import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    r=d['roofline']
    print('ms_per_step',d['ms_per_step'],'value',d['value']/1e9,'kernel_us',r['avg_kernel_us'],'frac',r['frac'],'var_dt',r.get('variable_dt',{}).get('avg_kernel_us'),'hbm',r.get('hbm_resident',{}).get('avg_kernel_us'),'hbm_ring',(r.get('hbm_resident_ring') or {}).get('avg_kernel_us'),(r.get('hbm_resident_ring') or {}).get('frac'))
