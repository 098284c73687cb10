"""Host CPU share of this process -- the smaller of the scheduler affinity and the cgroup's CPU quota.

A one-GPU box of the pool shows 256 logical CPUs but its cgroup allows 16 (cpu.max "1600000 100000"); torch then picks
128 intra-op threads and a CPU-side fp32 pass runs 2.7x SLOWER than with 16 (17.8 s vs 6.55 s for one bs=1 full-size
micro-batch, gpurun_out/cpu_threads_probe.log).  Used by the checkers (tests, smoke, bench's cpu_baseline); nothing on
the GPU path depends on it."""
import os


def cpu_share(cgroup_root="/sys/fs/cgroup"):
    """CPUs this process may keep busy: scheduler affinity, cut by the cgroup's quota (v2 ``cpu.max``, v1 ``cpu/cpu.cfs_*``)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open(os.path.join(cgroup_root, "cpu.max")) as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open(os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")) as f:
                q = int(f.read())
            with open(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us")) as f:
                p = int(f.read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def limit_torch_threads(cap=None):
    """torch's intra-op pool at the CPU share (never above what torch chose itself); returns the count in force."""
    import torch
    n = min(torch.get_num_threads(), cpu_share())
    if cap:
        n = min(n, cap)
    torch.set_num_threads(max(1, n))
    return torch.get_num_threads()
