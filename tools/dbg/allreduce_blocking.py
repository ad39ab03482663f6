"""Does the gradient exchange of Trainer._allreduce_grads block the HOST?  A long kernel queue is put in front of it and the host time of
each piece is measured (a non-blocking call returns in microseconds whatever the queue holds).  python tools/dbg/allreduce_blocking.py"""
import os, sys, time, json
import torch, torch.distributed as dist
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29577', rank=0, world_size=1, device_id=dev)
n = 4656258
flat = torch.zeros(n + 1, device=dev)
sizes = [n // 346] * 345 + [n - (n // 346) * 345]
views = list(torch.split(flat[:-1], sizes))
grads = [torch.randn(s, device=dev) for s in sizes]
a = torch.randn(8192, 8192, device=dev)


def busy(ms=40):
    for _ in range(max(1, int(ms / 7.5))):           # one 8192^3 fp32 product takes ~7.5 ms
        torch.mm(a, a)


def host_time(fn):
    torch.cuda.synchronize()
    busy()
    t0 = time.perf_counter()
    fn()
    t = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    return round(t, 3)


for _ in range(3):
    dist.all_reduce(flat)
torch.cuda.synchronize()
res = {
    'foreach_copy_in': host_time(lambda: torch._foreach_copy_(views, grads)),
    'scalar_store_setitem': host_time(lambda: flat.__setitem__(-1, 8.0)),
    'scalar_store_fill': host_time(lambda: flat[-1:].fill_(8.0)),
    'all_reduce': host_time(lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM)),
    'all_reduce_async_op': host_time(lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)),
    'div': host_time(lambda: flat[:-1].div_(flat[-1].clamp(min=1.0))),
    'foreach_copy_out': host_time(lambda: torch._foreach_copy_(grads, views)),
    'queue_ms_in_front': 40,
}
print(json.dumps(res))
dist.destroy_process_group()
