"""The data-parallel gradient exchange on CPU: world_size 2, gloo backend (the GPU path uses the same code
over RCCL).  Checks the DDP semantics of the reference: gradients are averaged over the ranks after EVERY
micro-batch backward and accumulated over 2 micro-batches before the step."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    from adaprompt_amd.parallel import GradReducer, init_distributed
    r, w, _ = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 4)
    extra = torch.nn.Parameter(torch.zeros(3))            # a trainable parameter that gets no gradient (unused-param tolerance)
    params = list(lin.parameters()) + [extra]
    red = GradReducer(params, bucket_bytes=64)            # tiny buckets: several collectives
    assert lin.weight.grad.data_ptr() == red.flat.data_ptr()
    xs = [torch.full((2, 8), float(rank + 1 + 10 * i)) for i in range(2)]
    local = []
    for i in range(2):
        red.wait()                                    # the previous exchange must land before backward writes
        before = red.flat.clone()
        lin(xs[i]).sum().backward()
        local.append((red.flat - before).clone())
        red.reduce()
    red.wait()
    # expected: mean over ranks of each micro-batch gradient, summed over the 2 micro-batches
    gathered = [[torch.zeros_like(local[i]) for _ in range(world)] for i in range(2)]
    for i in range(2):
        dist.all_gather(gathered[i], local[i])
    expect = sum(torch.stack(gathered[i]).mean(0) for i in range(2))
    ok = torch.allclose(red.flat, expect, rtol=1e-6, atol=1e-6)
    # replicas stay identical after the step
    opt = torch.optim.SGD(params, lr=0.1)
    opt.step()
    wl = [torch.zeros_like(lin.weight) for _ in range(world)]
    dist.all_gather(wl, lin.weight.detach())
    same = torch.equal(wl[0], wl[1])
    red.zero()
    q.put((rank, bool(ok), bool(same), float(red.flat.abs().sum())))
    dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, same, z in res:
        assert ok and same and z == 0.0, (rank, ok, same, z)


def test_single_process_reducer_is_a_noop():
    from adaprompt_amd.parallel import GradReducer
    lin = torch.nn.Linear(4, 2)
    red = GradReducer(lin.parameters())
    lin(torch.ones(1, 4)).sum().backward()
    g = red.flat.clone()
    red.reduce()
    red.wait()
    assert torch.equal(red.flat, g) and red.world == 1


def _worker_shared_flat(rank, world, port, q):
    """the optimiser-owned buffer form (GradReducer(flat=...)): gradients are views of a padded flat buffer somebody
    else allocated (adaprompt_amd.ldm.prodigy.Prodigy.grad_buffer on the GPU); padding stays zero through the exchange."""
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    from adaprompt_amd.parallel import GradReducer, init_distributed
    init_distributed(backend="gloo")
    torch.manual_seed(0)
    lin = torch.nn.Linear(5, 3)                          # 15 + 3 values
    flat = torch.zeros(15 + 1 + 3 + 1)                   # groups padded to multiples of 4, as the optimiser lays them out
    lin.weight.grad = flat[0:15].view(3, 5)
    lin.bias.grad = flat[16:19]
    red = GradReducer(list(lin.parameters()), flat=flat, bucket_bytes=32)
    assert red.flat.data_ptr() == flat.data_ptr()
    lin(torch.full((2, 5), float(rank + 1))).sum().backward()
    mine = flat.clone()
    red.reduce()
    red.wait()
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    ok = torch.allclose(flat, torch.stack(both).mean(0), rtol=1e-6, atol=1e-7)
    q.put((rank, bool(ok), float(flat[15]), float(flat[19])))
    dist.destroy_process_group()


def test_grad_reducer_on_a_shared_flat_buffer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_shared_flat, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, pad0, pad1 in res:
        assert ok and pad0 == 0.0 and pad1 == 0.0, (rank, ok, pad0, pad1)


def _worker_training_step(rank, world, port, q):
    """the PRODUCT's ``LatentDiffusion.training_step`` under a reducer, two ranks: the all-reduce issued after micro-batch i
    must have landed before micro-batch i+1's backward adds into the same flat buffer, and the step sees
    mean_r(g_1) + mean_r(g_2).  The UNet itself needs the GPU, so ``shared_step`` is replaced by a small differentiable
    model; everything else (the wait / backward / reduce / step / zero sequence) is the shipped code."""
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaprompt_amd.parallel import GradReducer, init_distributed
    init_distributed(backend="gloo")
    torch.manual_seed(0)

    class Toy(LatentDiffusion):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.lin = torch.nn.Linear(6, 3)
            self.manual_accumulate_grad_batches, self.grad_clip, self.batch_idx = 2, 0.0, 0
            self.cond_fn = None

        def shared_step(self, batch, **kw):
            out = self.lin(batch["x"])
            return out.detach().sum(), torch.ones_like(out), out, {}

    events = []

    class Spy(GradReducer):
        def reduce(self):
            events.append("reduce")
            super().reduce()

        def wait(self):
            events.append("wait:pending" if self.pending else "wait")
            super().wait()

    m = Toy()
    params = list(m.lin.parameters())
    red = Spy(params, bucket_bytes=32)
    hooks = [p.register_hook(lambda g: events.append("backward") or g) for p in params[:1]]
    opt = torch.optim.SGD(params, lr=0.5)
    w0 = m.lin.weight.detach().clone()
    xs = [torch.full((2, 6), float(rank + 1 + 3 * i)) for i in range(2)]
    for i in range(2):
        m.training_step({"x": xs[i]}, optimizer=opt, reducer=red)
    # every backward is preceded by a wait, and the 2nd one found a pending exchange and awaited it
    order_ok = all(events[j - 1].startswith("wait") for j, e in enumerate(events) if e == "backward") \
        and "wait:pending" in events[events.index("reduce"):]
    # expected update: -lr * (mean_r g_1 + mean_r g_2); d sum(lin(x)) / dW = ones(3,1) * sum_rows(x)
    g = sum(torch.stack([torch.full((2, 6), float(r + 1 + 3 * i)).sum(0) for r in range(world)]).mean(0) for i in range(2))
    want = w0 - 0.5 * g.expand(3, 6)
    ok = torch.allclose(m.lin.weight.detach(), want, rtol=1e-6, atol=1e-6)
    zeroed = float(red.flat.abs().sum()) == 0.0 and m.batch_idx == 2
    for h in hooks:
        h.remove()
    q.put((rank, bool(order_ok), bool(ok), bool(zeroed), events))
    dist.destroy_process_group()


def test_training_step_waits_for_the_previous_exchange_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_training_step, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, order_ok, ok, zeroed, events in res:
        assert order_ok and ok and zeroed, (rank, order_ok, ok, zeroed, events)


def _bucketed_worker(rank, world, port, q):
    """the bucketed exchange (chunks go out during the backward, as their gradients complete) against the single-shot one"""
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    from adaprompt_amd import functional
    from adaprompt_amd.parallel import GradReducer, init_distributed
    init_distributed(backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                              torch.nn.Linear(32, 8))
    direct = torch.nn.Parameter(torch.zeros(40))          # written "through a raw pointer", like the block Functions' weight gradients
    unused = torch.nn.Parameter(torch.zeros(5))           # gets no gradient: its chunk only goes out with reduce()
    params = list(net.parameters()) + [direct, unused]
    red = GradReducer(params, bucket_bytes=256)           # 64 floats per chunk: ~30 chunks, parameters straddle chunk borders
    assert len(red.chunks) > 10
    x = torch.randn(4, 16, generator=torch.Generator().manual_seed(10 + rank))
    issued_early = []
    totals = []
    for mb in range(2):                                   # two accumulated micro-batches, exchanged after each (DDP semantics)
        red.wait()
        red.begin_backward()
        before = red.flat.clone()
        loss = net(x * (mb + 1)).pow(2).sum()
        loss.backward()
        # a gradient the host code wrote itself: final once the "block" says so (functional.GRAD_DONE is the reducer's hook)
        direct.grad.add_(torch.arange(40.0) * (rank + 1) * (mb + 1))
        assert functional.GRAD_DONE == red.grad_ready
        functional._grads_done({"w": (direct, None)})
        issued_early.append(sum(red._issued))
        local = (red.flat - before).clone()               # NB: chunks already exchanged hold the SUM by now -> recompute below
        red.reduce()
        totals.append(None)
    red.wait()
    bucketed = red.flat.clone()
    # the same two micro-batches with the single-shot exchange on a twin
    torch.manual_seed(0)
    net2 = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                               torch.nn.Linear(32, 8))
    direct2, unused2 = torch.nn.Parameter(torch.zeros(40)), torch.nn.Parameter(torch.zeros(5))
    red2 = GradReducer(list(net2.parameters()) + [direct2, unused2], bucket_bytes=256)
    for mb in range(2):
        red2.wait()
        net2(x * (mb + 1)).pow(2).sum().backward()
        direct2.grad.add_(torch.arange(40.0) * (rank + 1) * (mb + 1))
        red2.reduce()
    red2.wait()
    same = torch.equal(bucketed, red2.flat)
    nonzero = float(bucketed.abs().sum()) > 0
    q.put((rank, bool(same), bool(nonzero), issued_early, len(red.chunks)))
    dist.destroy_process_group()


def test_bucketed_exchange_equals_single_shot_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucketed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, nonzero, early, nchunks in res:
        assert same and nonzero, (rank, same, nonzero)
        # most chunks left during the backward; the chunk of the parameter without a gradient only with reduce()
        assert all(0 < e < nchunks for e in early), (early, nchunks)


def _two_pass_worker(rank, world, port, q):
    """a block Function that runs TWICE in one graph writes its weight gradient through a raw pointer once
    per invocation; the bucketed exchange must not send the chunk after the first report."""
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    from adaprompt_amd import functional
    from adaprompt_amd.parallel import GradReducer, init_distributed
    init_distributed(backend="gloo")

    class RawLin(torch.autograd.Function):               # the block Functions' pattern (functional.ResBlockFn / SpatialTransformerFn)
        @staticmethod
        def forward(ctx, x, T):
            functional._note_forward(ctx, T)
            ctx.T = T
            ctx.save_for_backward(x)
            return x @ T["w"][0].detach().t()

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            w = ctx.T["w"][0]
            w.grad.add_(g.t() @ x)                        # "raw pointer" write into the flat buffer
            functional._grads_done(ctx.T)
            return g @ w.detach(), None

    def run(bucketed):
        torch.manual_seed(0)
        w = torch.nn.Parameter(torch.randn(24, 24) * 0.2)
        tail = torch.nn.Parameter(torch.randn(24) * 0.1)  # an autograd-managed parameter behind the block
        red = GradReducer([w, tail], bucket_bytes=512)    # 128 floats per chunk: w spans 5 chunks
        T = {"w": (w, None)}
        x = torch.randn(6, 24, generator=torch.Generator().manual_seed(20 + rank)).requires_grad_(True)
        sent_after_first = []
        for mb in range(2):
            red.wait()
            h = RawLin.apply(RawLin.apply(x * (mb + 1), T), T)          # two invocations, one backward
            loss = (h * tail).pow(2).sum()
            if bucketed:
                red.begin_backward()
                reports = []
                orig = red.grad_ready

                def spy(p, _orig=orig, _reports=reports):
                    _orig(p)
                    if p is w:
                        _reports.append(sum(red._issued[c] for c in red._chunks_of[id(w)]))
                functional.GRAD_DONE = spy
            loss.backward()
            if bucketed:
                sent_after_first.append(reports)
                functional.GRAD_DONE = red.grad_ready
            red.reduce()
        red.wait()
        return red.flat.clone(), sent_after_first

    a, reports = run(True)
    b, _ = run(False)
    # per micro-batch: two reports for w; none of w's chunks may be out after the first, all after the second
    order_ok = all(len(r) == 2 and r[0] == 0 and r[1] > 0 for r in reports)
    q.put((rank, bool(torch.equal(a, b)), bool(order_ok), reports, float(a.abs().sum())))
    dist.destroy_process_group()


def test_bucketed_exchange_with_a_block_invoked_twice_in_one_backward_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_pass_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, order_ok, reports, mass in res:
        assert same and order_ok and mass > 0, (rank, same, order_ok, reports)
