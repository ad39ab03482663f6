"""Where a train step's host time and GPU time go, phase by phase (forward backbones+hot path / losses / backward /
grad reduce / Adam / log).  Two passes: queue-only host times (no sync between phases), then with a device sync after each."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from graspnerf_amd import losses
from graspnerf_amd.trainer import Trainer, train_losses, train_losses_stacked, exp_decay_lr

dev = torch.device('cuda', 0)
net = bench.build_model(dev)
tr = Trainer(net)
scenes = bench.train_scenes(8, 0, dev)
for _ in range(6):
    tr.step(scenes)
torch.cuda.synchronize()


def step(sync):
    T = {}
    def mark(name, t0):
        if sync:
            torch.cuda.synchronize()
        T[name] = (time.perf_counter() - t0) * 1e3
        return time.perf_counter()
    t = time.perf_counter()
    tr.net.train()
    tr.optimizer.zero_grad(set_to_none=True)
    datas = [dict(d, step=tr.step_id) for d in scenes]
    st = tr.net.forward_scenes(datas, stacked=True)
    t = mark('forward', t)
    terms = train_losses_stacked(st, datas)
    tot = losses.total_loss(terms, scenes=len(datas))
    t = mark('losses', t)
    tot.backward()
    t = mark('backward', t)
    tr._allreduce_grads(len(scenes))
    t = mark('reduce', t)
    tr.optimizer.step()
    t = mark('adam', t)
    means = torch.stack([terms[k].detach().float().mean() for k in terms])
    t = mark('log_queue', t)
    means.tolist()
    T['log_wait'] = (time.perf_counter() - t) * 1e3
    tr.step_id += 1
    return T

res = {}
for sync in (False, True):
    rows = [step(sync) for _ in range(5)]
    res['sync' if sync else 'queue'] = {k: round(float(np.median([r[k] for r in rows])), 2) for k in rows[0]}
    res[('sync' if sync else 'queue') + '_total'] = round(float(np.median([sum(r.values()) for r in rows])), 2)
print(json.dumps(res))
