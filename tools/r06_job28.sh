#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py ) > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
tail -3 gpurun_out/r06_bench_final.err
python -c "
import json
for l in open('gpurun_out/r06_bench_final.json'):
    if l.startswith('{'):
        r=json.loads(l); print(r['value'], r['unit'], r['ms_per_step'], r['roofline']['frac'], r['vs_reference']['ok'])
        for row in r['secondary']: print('  ', row['config'][:70], row.get('ms_per_eval'))
"
