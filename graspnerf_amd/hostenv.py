"""Host-side CPU budget of a rank.  A GPU node shows every hardware thread to every container (os.cpu_count() = 256 on the MI355X
boxes) while the container's cgroup grants a fraction of them (cpu.max = 16 CPUs there).  PyTorch sizes its intra-op pool from the
visible count (128 threads): one small CPU op of the training step (the reference's torch.rand / randperm draws, a torch.cat on
host tensors) wakes the whole pool, the pool burns the container's quota of the 100 ms scheduling period, and the kernel throttles
every thread of the process -- including the one that queues GPU work -- for the rest of the period: training steps of 62 ms
turned into 130-160 ms one time in five (tools/dbg/train_step_series.py; 0 collections of the Python GC, 0 allocator retries, the
stall sits in the host time of the forward / loss queueing).  limit_host_threads() sizes the pools to the budget instead."""
import os


def cpu_budget():
    """CPUs this process may actually use: min(scheduler affinity, cgroup quota) (cgroup v2 cpu.max, then v1 cfs quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def limit_host_threads(local_world=1, cap=8):
    """Size torch's CPU pools for one of `local_world` ranks sharing this host: at most `cap` intra-op threads and no more than the
    rank's share of the CPU budget minus the two threads that queue GPU work (main + autograd).  -> threads set."""
    import torch
    share = max(1, cpu_budget() // max(1, local_world))
    n = max(1, min(cap, share - 2))
    torch.set_num_threads(n)
    try:
        torch.set_num_interop_threads(max(1, min(4, n)))     # only possible before the first parallel region
    except RuntimeError:
        pass
    return n
