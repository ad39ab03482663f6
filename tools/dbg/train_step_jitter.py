"""Where do the slow training steps spend their time?  Host timestamps around the phases of Trainer.step (no extra syncs) and
HIP events on the stream, 40 steps.  python tools/dbg/train_step_jitter.py  (GPU)"""
import os, sys, time, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ['train_step_bench.py', '--steps', '1', '--warmup', '5']
g = runpy.run_path(os.path.join(ROOT, 'tools', 'train_step_bench.py'), run_name='__main__')
import torch
from graspnerf_amd import trainer as T, losses as LS
tr, scenes, net = g['tr'], g['scenes'], g['net']
marks = []
def stamp(name):
    ev = torch.cuda.Event(enable_timing=True); ev.record()
    marks.append((name, time.perf_counter(), ev))
orig_fs, orig_tl, orig_total, orig_ar, orig_opt = net.forward_scenes, T.train_losses_stacked, LS.total_loss, tr._allreduce_grads, tr.optimizer.step
def fs(*a, **k):
    stamp('fwd0'); out = orig_fs(*a, **k); stamp('fwd1'); return out
def tl(*a, **k):
    out = orig_tl(*a, **k); stamp('loss1'); return out
def ar(*a, **k):
    stamp('bwd1'); out = orig_ar(*a, **k); stamp('ar1'); return out
def opt(*a, **k):
    out = orig_opt(*a, **k); stamp('opt1'); return out
net.forward_scenes, T.train_losses_stacked, tr._allreduce_grads, tr.optimizer.step = fs, tl, ar, opt
rows = []
for i in range(80):
    marks.clear()
    torch.cuda.synchronize(); tp = time.perf_counter()
    tr.net.train(); tp1 = time.perf_counter()
    tr.optimizer.zero_grad(set_to_none=True); tp2 = time.perf_counter()
    pre = ((tp1 - tp) * 1e3, (tp2 - tp1) * 1e3)
    t = time.perf_counter()
    tr.step(scenes)
    t_ret = time.perf_counter()
    torch.cuda.synchronize(); t_end = time.perf_counter()
    host = {marks[j + 1][0]: (marks[j + 1][1] - marks[j][1]) * 1e3 for j in range(len(marks) - 1)}
    gpu = {marks[j + 1][0]: marks[j][2].elapsed_time(marks[j + 1][2]) for j in range(len(marks) - 1)}
    host['pre_train()'], host['pre_zero_grad'], host['step_start_to_fwd0'] = pre[0], pre[1], (marks[0][1] - t) * 1e3
    rows.append(((t_end - t) * 1e3, (t_ret - t) * 1e3, host, gpu))
med = sorted(r[0] for r in rows)[40]
print('median step %.1f ms' % med)
for i, (tot, ret, host, gpu) in enumerate(rows):
    if tot > med + 8 or i == 3 or max(host['pre_train()'], host['pre_zero_grad']) > 5:
        print(f'step {i}: total {tot:.1f} (step() returned after {ret:.1f})  host ms {dict((k, round(v, 1)) for k, v in host.items())}  gpu ms {dict((k, round(v, 1)) for k, v in gpu.items())}', flush=True)
