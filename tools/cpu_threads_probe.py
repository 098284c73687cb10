"""How many host threads should the CPU oracle use on a GPU box?  Times oracle.ldm_oracle.recon_step (bs=1, full SD-1.5
size, forward + backward) at torch's default intra-op thread count and at fixed counts.  Test infrastructure only."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch default threads",
      torch.get_num_threads(), flush=True)
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try:
        print(p, open(p).read().strip(), flush=True)
    except OSError:
        pass
for th in [torch.get_num_threads()] + [int(a) for a in sys.argv[1:]]:
    t0 = time.time()
    r = bench.cpu_baseline(th, batch=1, timed=1)
    print("threads", th, "s/micro-batch", r["seconds_per_micro_batch"], "wall", round(time.time() - t0, 1), flush=True)
