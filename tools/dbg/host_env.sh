# host-side facts that explain step-time jitter: CPU quota of the container (CFS throttling), visible CPUs, OpenMP / torch thread defaults
echo "nproc: $(nproc)  cpu_count: $(python -c 'import os; print(os.cpu_count())')"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cpu.stat:"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null
echo "cfs_quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) period: $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
echo "affinity: $(python -c 'import os; print(len(os.sched_getaffinity(0)))')"
python -c "import torch; print('torch threads', torch.get_num_threads(), 'interop', torch.get_num_interop_threads())"
echo "OMP_NUM_THREADS=$OMP_NUM_THREADS MKL_NUM_THREADS=$MKL_NUM_THREADS"
uptime
