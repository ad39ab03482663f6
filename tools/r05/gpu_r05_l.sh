#!/bin/bash
# round 5, GPU call L: new agreement test, host synchronisations of a train step with the N > 1 exchange (torch's sync debug mode)
R=$PWD; T=r05_l; O=$R/gpurun_out/$T; mkdir -p $O
timeout 300 python -m pytest tests/test_bwd_twins.py -m gpu -x -q -k "partner" > $O/tests.log 2>&1; tail -5 $O/tests.log
cat > /tmp/syncdbg.py <<'PY'
import os, sys, warnings
sys.argv = ['x', '--steps', '2', '--warmup', '6', '--flat-exchange-steps', '1']
sys.path.insert(0, '/root/repo/tools')
import runpy, torch
# run the bench tool up to its end, then one more step of each kind under sync debug
g = runpy.run_path('/root/repo/tools/train_step_bench.py', run_name='__main__')
tr, scenes = g['tr'], g['scenes']
import torch.distributed as dist
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29641', rank=0, world_size=1, device_id=torch.device('cuda', 0))
tr.flat_exchange = True
tr.step(scenes); torch.cuda.synchronize()
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode(1)
print('=== sync-debug step (flat exchange on, log_every', tr.log_every, ') ===', file=sys.stderr)
tr.step(scenes)
torch.cuda.set_sync_debug_mode(0)
torch.cuda.synchronize()
print('=== end ===', file=sys.stderr)
dist.destroy_process_group()
PY
timeout 400 python /tmp/syncdbg.py > $O/syncdbg.out 2> $O/syncdbg.err
sed -n '/=== sync-debug step/,/=== end ===/p' $O/syncdbg.err | grep -v "^$" | cut -c1-220 | head -40
