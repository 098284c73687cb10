"""One rank of the 2-rank data-parallel GPU test (tests/test_parallel_gpu.py starts two of these as child processes).
Runs two micro-batches of the product's ``LatentDiffusion.training_step`` (narrow UNet, hook stand-in, Prodigy on its flat
buffer, ``GradReducer`` on that buffer) and, on every rank, the same optimiser step recomputed WITHOUT a process group from
both ranks' micro-batches with the gradients averaged by hand -- the DDP semantics of the reference (main.py:829,
ddpm.py:591-608).  Backend: "nccl" (= RCCL) with one GPU per rank when the box has two, otherwise both ranks share cuda:0
and exchange over gloo (RCCL refuses two ranks on one device)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def build(dev, seed):
    from adaprompt_amd import synth
    from adaprompt_amd.hook_standin import SyntheticSubjBasisGenerator, make_cond_fn
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaprompt_amd.ldm.prodigy import Prodigy
    ucfg = dict(synth.SD15_UNET, model_channels=64, context_dim=128)        # 8 heads x dim_head 8 (the kernels need a multiple of 8)
    vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=64)
    torch.manual_seed(seed)
    hook = SyntheticSubjBasisGenerator(n_params=3 * 16 * 77 * 128, tokens=77, dim=128, id_dim=32)
    with torch.no_grad():
        hook.bases.mul_(20.0)
    hook = hook.to(dev)
    ld = LatentDiffusion.hot_path({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}},
                                  {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg},
                                  cond_fn=make_cond_fn(hook, capture=False))
    ld.load_state_dict(synth.synthetic_unet_state_dict(ucfg), strict=False)
    ld = ld.to(dev)
    ld.freeze_unet()
    params = list(hook.parameters())
    opt = Prodigy(params, lr=1.0, betas=(0.9, 0.999), d_coef=2.0, use_bias_correction=True, weight_decay=0.0)
    return ld, hook, params, opt


def micro_batch(rank, mb, dev):
    from adaprompt_amd import synth
    B = 2
    tag = f"dp.{rank}.{mb}"
    x0 = synth.synthetic_input(tag + ".x0", (B, 4, 64, 64))
    noise = synth.synthetic_input(tag + ".noise", (B, 4, 64, 64))
    ids = synth.synthetic_input(tag + ".ids", (B, 32))
    t = torch.tensor([(100 + 300 * mb + 50 * rank) % 1000, (850 - 200 * mb - 30 * rank) % 1000])       # (timesteps 0 .. 999)
    m = torch.ones(B, 64, 64)
    batch = {"zs_id_embs": ids.to(dev), "fg_mask": m.to(dev), "aug_mask": m.to(dev)}
    return batch, dict(t=t.to(dev), noise=noise.to(dev), x_start=x0.to(dev))


def lanes_main(rank, world, dev):
    """ADAP_DP_MODE=lanes: the SHIPPED mode under data parallelism -- two accumulation windows through
    ``training_window`` on two micro-batch lanes with the exchange inside the lanes' gate (``MicroBatchLanes(reducer=...)``) --
    against the hand-averaged sequential loop: per window, the gradients of BOTH ranks' micro-batches at the window's weights,
    mean over ranks, summed over the window, clip + Prodigy step (main.py:829; ddpm.py:591-633)."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import MicroBatchLanes
    from adaprompt_amd.parallel import GradReducer
    ld, hook, params, opt = build(dev, seed=3)
    red = GradReducer(params, flat=opt.grad_buffer)
    assert red.world == world
    lanes = MicroBatchLanes(params, n=2, reducer=red)
    losses = []
    for w in range(2):
        mbs = [micro_batch(rank, 2 * w + k, dev) for k in range(2)]
        out = ld.training_window([b for b, _ in mbs], opt, red, None, lanes, step_kwargs=[kw for _, kw in mbs])
        losses += [float(o[0]) for o in out]
    red.wait()
    torch.cuda.synchronize()
    lanes.remove()
    dp = torch.cat([p.detach().flatten() for p in params]).clone()
    d_dp = opt.device_state()["d"]
    ld2, hook2, params2, opt2 = build(dev, seed=3)
    ref_losses = []
    for w in range(2):
        acc = torch.zeros_like(opt2.grad_buffer)
        for k in range(2):
            for r in range(world):
                batch, kw = micro_batch(r, 2 * w + k, dev)
                opt2.grad_buffer.zero_()
                loss, grad, out, aux = ld2.shared_step(batch, **kw)
                ld2.manual_backward(out, grad, aux)
                acc += opt2.grad_buffer / world
                if r == rank:
                    ref_losses.append(float(loss))
        opt2.grad_buffer.copy_(acc)
        opt2.step(clip_norm=ld2.grad_clip)
        opt2.grad_buffer.zero_()
    torch.cuda.synchronize()
    ref = torch.cat([p.detach().flatten() for p in params2])
    init = torch.cat([p.detach().flatten() for p in build(dev, seed=3)[2]])
    both = [torch.zeros_like(dp) for _ in range(world)]
    dist.all_gather(both, dp)
    print("DPRESULT " + json.dumps({
        "rank": rank, "backend": dist.get_backend(), "device": str(dev), "mode": "lanes", "moved": float((ref - init).norm()),
        "rel_err_vs_hand_averaged": float((dp - ref).norm() / (ref - init).norm()),
        "replicas_identical": bool(torch.equal(both[0], both[1])),
        "grad_buffer_zeroed": float(opt.grad_buffer.abs().max()) == 0.0,
        "losses": losses, "ref_losses": ref_losses, "d": d_dp, "ref_d": opt2.device_state()["d"],
        "optimizer_steps": opt.device_state()["k"], "bytes_per_reduce": red.bytes_per_reduce}), flush=True)
    dist.destroy_process_group()


def main():
    from adaprompt_amd.parallel import GradReducer, init_distributed
    rank, world, local = init_distributed()
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if os.environ.get("ADAP_DP_MODE") == "lanes":
        return lanes_main(rank, world, dev)
    # ---- data parallel: each rank its own micro-batches, exchange after every backward
    ld, hook, params, opt = build(dev, seed=3)
    red = GradReducer(params, flat=opt.grad_buffer)
    assert red.world == world
    for mb in range(2):
        batch, kw = micro_batch(rank, mb, dev)
        ld.training_step(batch, optimizer=opt, reducer=red, **kw)
    red.wait()
    torch.cuda.synchronize()
    dp = torch.cat([p.detach().flatten() for p in params]).clone()
    # ---- the same step by hand, no process group: gradients of BOTH ranks' micro-batches, mean over ranks, summed over
    # the two micro-batches
    ld2, hook2, params2, opt2 = build(dev, seed=3)
    acc = torch.zeros_like(opt2.grad_buffer)
    for mb in range(2):
        for r in range(world):
            batch, kw = micro_batch(r, mb, dev)
            opt2.grad_buffer.zero_()
            loss, grad, out, aux = ld2.shared_step(batch, **kw)
            ld2.manual_backward(out, grad, aux)
            acc += opt2.grad_buffer / world
    opt2.grad_buffer.copy_(acc)
    opt2.step(clip_norm=ld2.grad_clip)
    torch.cuda.synchronize()
    ref = torch.cat([p.detach().flatten() for p in params2])
    init = torch.cat([p.detach().flatten() for p in build(dev, seed=3)[2]])
    moved = float((ref - init).norm())
    err = float((dp - ref).norm() / (ref - init).norm())
    # replicas identical after the step
    both = [torch.zeros_like(dp) for _ in range(world)]
    dist.all_gather(both, dp)
    same = bool(torch.equal(both[0], both[1]))
    print("DPRESULT " + json.dumps({"rank": rank, "backend": dist.get_backend(), "device": str(dev), "moved": moved,
                                    "rel_err_vs_hand_averaged": err, "replicas_identical": same,
                                    "grad_buffer_zeroed": float(opt.grad_buffer.abs().max()) == 0.0,
                                    "bytes_per_reduce": red.bytes_per_reduce}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
