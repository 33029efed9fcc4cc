import json,glob,sys
for f in sorted(glob.glob('gpurun_out/variant_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['roofline']['kernels_ms']
        print(f.split('variant_')[1][:-5].ljust(12), '%.4f'%d['ms_per_step'], ' '.join('%s=%.4f'%(a.replace('k_filter_','').replace('k_',''),b) for a,b in k.items()), d['config']['interior_waypoint_items'], d['config']['undecided_items_last_step'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
