#!/bin/bash
# round 6, job 24: the default bench line after the late changes (sanity + refreshed profiles/r06_bench_line.json)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
tail -3 gpurun_out/r06_bench_final.err
python -c "
import json
for l in open('gpurun_out/r06_bench_final.json'):
    if l.startswith('{'):
        r=json.loads(l); print(r['metric'], r['value'], r['unit'], r['ms_per_step'], r['roofline']['frac'], r['cpu_baseline'])
        for k,v in (r.get('secondary') or {}).items():
            print('  ', k, {kk: vv for kk,vv in v.items() if kk in ('ms','frac','us','value','vs_reference')} if isinstance(v, dict) else v)
"
