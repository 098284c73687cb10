"""Two ranks of the product's training step on the GPU: ``LatentDiffusion.training_step`` +
``GradReducer`` + fused Prodigy, one process per rank, against the same step recomputed in one process with hand-averaged
gradients.  With two GPUs on the box the exchange runs over RCCL (backend "nccl"), one GPU per rank; with one GPU both
ranks share it and exchange over gloo -- the training-step / reducer code under test is the same."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_training_step_two_ranks_matches_hand_averaged_step():
    ndev = torch.cuda.device_count()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   ADAP_DIST_BACKEND="nccl" if ndev >= 2 else "gloo")
        if ndev < 2:          # two processes on one device: two-pass GroupNorm (two single-launch grids cannot both be resident)
            env["ADAP_GN_TWO_PASS"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dp_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    res = [json.loads(l.split("DPRESULT ", 1)[1]) for so, _ in outs for l in so.splitlines() if l.startswith("DPRESULT ")]
    assert len(res) == 2
    for r in res:
        assert r["backend"] == ("nccl" if ndev >= 2 else "gloo")
        assert r["replicas_identical"] and r["grad_buffer_zeroed"] and r["moved"] > 0
        # same gradients, summed in a different order (ranks in parallel vs one after the other): f32 round-off only
        assert r["rel_err_vs_hand_averaged"] < 1e-4, r


@pytest.mark.gpu
def test_allreduce_bucket_c_entry_single_rank():
    """``adap_allreduce_bucket`` on a communicator the library creates itself (unique id -> ncclCommInitRank -> ncclAllReduce,
    RCCL resolved with dlopen): with one rank the mean and the sum of a buffer are the buffer, f32 and bf16, on a side stream.
    (More ranks need more GPUs than a test box has: the N > 1 path of this entry is exercised by the driver's scaling run
    only when ADAP_REDUCER_BACKEND=c_abi is set; the default exchange goes through torch.distributed.)"""
    import torch
    from adaprompt_amd.parallel import CAbiComm
    uid = CAbiComm.new_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = CAbiComm(uid, 1, 0)
    x = torch.randn(1 << 20, device="cuda")
    ref = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        comm.allreduce_(x, average=True)
        comm.allreduce_(x, average=False)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    y = torch.randn(4096, device="cuda").to(torch.bfloat16)
    ry = y.clone()
    comm.allreduce_(y)
    torch.cuda.synchronize()
    assert torch.equal(y, ry)
    comm.destroy()
