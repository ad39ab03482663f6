"""Host-side profile of a training step (cProfile, sorted by own time): where the Python / launch time of Trainer.step goes.
python tools/dbg/train_step_hostprof.py  (GPU)"""
import cProfile, pstats, os, sys, io, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ['train_step_bench.py', '--steps', '1', '--warmup', '3']
g = runpy.run_path(os.path.join(ROOT, 'tools', 'train_step_bench.py'), run_name='__main__')
import torch, time
tr, scenes = g['tr'], g['scenes']
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): tr.step(scenes)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'3 steps: host issue {t_issue * 1e3 / 3:.1f} ms/step, with sync {t_all * 1e3 / 3:.1f} ms/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(3): tr.step(scenes)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
print(s.getvalue()[:9000])
