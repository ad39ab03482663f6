#!/bin/bash
# round 5, GPU call E: whole suite + the default bench line of the tree with the pair-form backward twins
R=$PWD; T=r05_e; O=$R/gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; cp gpurun_out/parity_errors.json $O/parity_errors.json; tail -4 $O/suite.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); t=d['train_step']
print('fwd', d['value'], d['ms_per_step'], 'roofline', {k:d['roofline'][k] for k in ('frac','frac_executed','ms_per_launch')})
print('train', t['value'], t['ms_per_step'], 'readback', t['value_with_per_step_readback'], t['split_ms_per_step'])
print('  at_2_cpus', {k:v for k,v in t.get('at_2_cpus',{}).items() if k!='host_ms_each_step'})
print('  host cpu', t['host_cpu_ms_median'], 'unblocked', t['host_ms_unblocked'])
print('   ', {k:v for k,v in t['hip_kernels_ms_per_step'].items() if v>0.3})
print('arbiter', d['fp64_arbiter_at_benched_shape'])
print('cpu', d.get('cpu_baseline'))
PY
