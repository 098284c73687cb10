#!/usr/bin/env python3
"""Headline benchmark: SD-1.5 UNet training images/sec @512px, bs=4/GPU (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torchrun, one rank per GPU, RCCL gradient all-reduce over xGMI)

A "step" is one pure-recon distillation micro-batch of Stage-1 AdaFace training at the full
SD-1.5 sizes (BASELINE.json configs[1], the stable sub-case of SURVEY.md 8d config 2):
  VAE encode of 4 synthetic 512x512 face-shaped images with fg/aug masks (no grad)
  -> posterior sample x 0.18215 -> q_sample -> UNet eps-prediction forward with the 16-way
  layerwise context, img_mask on self-attention and capture of the 12 distillation layers
  -> masked fg/bg-weighted MSE -> backward through the frozen UNet to the context
  -> backward through the embedding hook -> data-parallel mean of the trainable gradients
  (every micro-batch, as DDP does) -> every 2nd micro-batch: clip 0.5 + optimizer step + zero_grad.
Weights are random (no checkpoints offline), data synthetic.  The embedding hook (CLIP text
encoder + SubjBasisGenerator, HF weights unavailable offline) is replaced by a stand-in with the
same trainable byte size (149 M fp32 parameters, adaprompt_amd/hook_standin.py).

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel family,
HIP-event timed) and `cpu_baseline` (the oracle on the host cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

BF16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)
HBM_PEAK_GBS = 8000.0

# algorithmic work per image of the step (SURVEY.md 8d, 2*MAC, forward): UNet 803.7 GFLOP, bwd (dX only) ~1x,
# VAE encoder 1116.7 GFLOP
GFLOP_PER_IMAGE = 1116.7 + 2 * 803.7


def synthetic_batch(B, device, seed):
    """image = clamp(randn*0.5,-1,1) HWC, centred-ellipse fg mask (~35 % area), aug mask with a
    zero border of U{0..76} px, L2-normalised id embedding."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    img = (torch.randn(B, 512, 512, 3, generator=g) * 0.5).clamp(-1, 1)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 512), torch.linspace(-1, 1, 512), indexing="ij")
    fg = ((xx / 0.62) ** 2 + (yy / 0.72) ** 2 <= 1.0).float()[None].repeat(B, 1, 1)
    aug = torch.zeros(B, 512, 512)
    for b in range(B):
        bd = int(torch.randint(0, 77, (1,), generator=g))
        aug[b, bd:512 - bd, bd:512 - bd] = 1
    ids = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1)
    return {"image": img.to(device), "fg_mask": fg.to(device), "aug_mask": aug.to(device), "zs_id_embs": ids.to(device)}


def device_state_dict(shapes, prefix, device, seed):
    """random-init weights of the named architecture, generated on the device (fan-in scaled, norms ~1)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in shapes:
        if len(shape) >= 2:
            fan = 1
            for s in shape[1:]:
                fan *= s
            t = torch.randn(shape, device=device, generator=g) * (0.8 / fan ** 0.5)
        elif name.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shape, device=device, generator=g)
        else:
            t = 0.05 * torch.randn(shape, device=device, generator=g)
        sd[prefix + name] = t
    return sd


def build_model(device, seed=0):
    from adaprompt_amd import synth
    from adaprompt_amd.hook_standin import SyntheticSubjBasisGenerator, make_cond_fn
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    with torch.device(device):        # parameters are created (and default-initialised) directly in HBM
        ld = LatentDiffusion.hot_path(
            {"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": dict(synth.SD15_VAE_DD), "embed_dim": 4}},
            {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(synth.SD15_UNET)})
    ld = ld.to(device)
    sd = device_state_dict(synth.unet_param_shapes(**synth.SD15_UNET), "model.diffusion_model.", device, seed)
    sd.update(device_state_dict(synth.vae_encoder_param_shapes(**synth.SD15_VAE_DD), "first_stage_model.", device, seed + 1))
    missing, unexpected = ld.load_state_dict(sd, strict=False)
    assert not unexpected
    del sd
    ld.freeze_unet()
    with torch.device(device):
        hook = SyntheticSubjBasisGenerator()
    # regs=True: the conditioning side also hands over what the recon iteration's two regularisers read (ddpm.py:3207-3270)
    ld.cond_fn = make_cond_fn(hook, capture=True, regs=True)
    # attribution aid: ADAP_OFF_<weight name>=1 switches one auxiliary loss off
    for _k in ("fg_bg_complementary_loss_weight", "fg_bg_xlayer_consist_loss_weight", "prompt_emb_delta_reg_weight"):
        if os.environ.get("ADAP_OFF_" + _k):
            setattr(ld, _k, 0.0)
    return ld, hook


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(threads, batch=1, timed=3):
    """the oracle (a CPU port of the reference path, pinned by the reference's golden vectors) on the
    host cores -- one pure-recon micro-batch (VAE encode -> q_sample -> UNet forward -> masked MSE -> backward to the
    context) at full SD-1.5 size; 1 warm-up + ``timed`` timed iterations, median.  The default sample is bs=1 (about 25 s of CPU
    work in all); ``--cpu-batch 4`` times the section's bs=4 unit."""
    from adaprompt_amd import synth
    from oracle import ldm_oracle as O
    torch.set_num_threads(threads)
    usd = synth.synthetic_unet_state_dict()
    vsd = synth.synthetic_vae_state_dict()
    B = batch
    g = torch.Generator().manual_seed(7)
    img = (torch.randn(B, 3, 512, 512, generator=g) * 0.5).clamp(-1, 1)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 512), torch.linspace(-1, 1, 512), indexing="ij")
    fg = ((xx / 0.62) ** 2 + (yy / 0.72) ** 2 <= 1.0).float()[None, None].repeat(B, 1, 1, 1)
    aug = torch.zeros(B, 1, 512, 512)
    aug[:, :, 20:492, 20:492] = 1
    fg64 = torch.nn.functional.interpolate(fg, size=(64, 64), mode="nearest")
    im64 = torch.nn.functional.interpolate(aug, size=(64, 64), mode="nearest")
    ctx = torch.randn(16 * B, 77, 768, generator=g) * 0.05
    times = []
    for i in range(1 + timed):
        pn, nz = torch.randn(B, 4, 64, 64, generator=g), torch.randn(B, 4, 64, 64, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        t0 = time.time()
        O.recon_step(usd, vsd, dict(synth.SD15_UNET), dict(synth.SD15_VAE_DD), img, {"fg_mask": fg, "aug_mask": aug},
                     pn, t, nz, ctx, im64, fg64, 0.1, need_grad=True)
        if i > 0:
            times.append(time.time() - t0)
    times.sort()
    dt = times[len(times) // 2]
    return {"value": round(B / dt, 5), "unit": "images/sec", "cores": threads, "kind": "port",
            "cpu_model": cpu_model_name(), "host_logical_cpus": os.cpu_count(),
            "sample": f"bs={B} pure-recon micro-batch at full SD-1.5 size (VAE encode + q_sample + UNet fwd + masked MSE + "
                      f"bwd to the context; MSE only -- the GPU step additionally computes the iteration's auxiliary losses and "
                      f"the optimiser step), oracle/ldm_oracle.py (torch fp32), 1 warm-up + {timed} timed, median",
            "seconds_per_micro_batch": round(dt, 2), "timed_seconds": [round(x, 2) for x in times],
            "achieved_gflops": round(GFLOP_PER_IMAGE * B / dt, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU per micro-batch (reference: 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="encode each micro-batch inline on the main stream instead of prefetching the next one's VAE encode "
                         "on a second stream")
    ap.add_argument("--graph", action="store_true",
                    help="replay the micro-batch as two hipGraphs (fwd, bwd) instead of launching eagerly; measured "
                         "45.1 vs 44.3 ms/step on MI355X -- the step is GPU-bound, not launch-bound, so eager is the default")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-batch", type=int, default=1, help="batch of the cpu_baseline sample (BASELINE.md section 3 unit: 4)")
    ap.add_argument("--no-cpu-k8", action="store_true", help="skip the second, 8-thread cpu_baseline sample")
    ap.add_argument("--no-aggregates", action="store_true",
                    help="skip the north star's isolated aggregates (tools/measure_round.sh: the rocprofv3 statistics of the "
                         "dominant kernel then cover the training step's launches only, like the roofline's HIP events)")
    ap.add_argument("--no-ddim", action="store_true",
                    help="skip the extra leg that times config 5 (50-step DDIM at bs=8 with guidance + VAE decode)")
    ap.add_argument("--no-clock-probe", action="store_true",
                    help="skip the ~2 s in-kernel clock measurement of the roofline leg (its 8000 launches of one shape "
                         "would dominate a rocprofv3 --stats average of the dominant kernel)")
    ap.add_argument("--no-unfrozen", action="store_true",
                    help="skip the `unfreeze_model: True` leg (weight gradients + 4.5 GB optimiser / all-reduce payload)")
    ap.add_argument("--no-zs-frontend", action="store_true", help="skip the zero-shot front end leg (CLIP ViT-L/14 image encoder)")
    ap.add_argument("--no-compos", action="store_true", help="skip the config-4 leg (Stage-2 compositional micro-batches)")
    ap.add_argument("--fuse", action="store_true",
                    help="run an accumulation window's two recon micro-batches as ONE batched UNet pass instead of on two lanes "
                         "(measured slower here: 26.3 vs 24.9 ms; it is the default only where lanes cannot be used)")
    ap.add_argument("--no-lanes", action="store_true",
                    help="one stream for both micro-batches of an accumulation window (round 3's loop) instead of one each")
    ap.add_argument("--no-distill-mix", action="store_true",
                    help="skip the extra (untimed-for-`value`) leg that runs config 2's Arc2Face-distillation iteration mix")
    ap.add_argument("--entry", choices=["window", "lightning"], default="window",
                    help="how the timed loop drives the model: 'window' = training_window per accumulation window (the default); "
                         "'lightning' = training_step(batch, batch_idx) once per micro-batch, as Lightning calls it (ddpm.py:515), "
                         "with adaprompt_amd.trainer.Trainer attached (lanes + two windows of latents prefetched ahead).  The default "
                         "run times the other entry as an extra leg (`entry_lightning`)")
    ap.add_argument("--no-entry-leg", action="store_true", help="skip the extra leg that times the other entry")
    ap.add_argument("--no-rehearse-exchange", action="store_true",
                    help="skip the one-GPU REHEARSAL of the data-parallel exchange (extra leg `exchange_rehearsal`: a resident kernel "
                         "of a collective's footprint stands in for RCCL's all-reduce of the gradient payload after every backward)")
    ap.add_argument("--rehearse-ranks", type=int, default=8, help="the node size the rehearsal prices the ring all-reduce for")
    ap.add_argument("--rehearse-blocks", type=int, default=64, help="workgroups (x 512 threads) of the stand-in collective kernel")
    ap.add_argument("--emulate-node-share", type=int, default=0, metavar="RANKS",
                    help="rehearse this rank's HOST side as one of RANKS ranks sharing the node's CPUs: before any HIP call the "
                         "process is pinned to cpu_share / RANKS CPUs (sched_setaffinity) and torch's pool sized to match; the "
                         "line reports it (`emulated_node_share`).  8 ranks of an 8-GPU node share its CPU quota (main.py:829)")
    args = ap.parse_args()

    # a hung run should say where: after ADAP_BENCH_WATCHDOG seconds (default 900; 0 = off) every thread's Python stack goes to
    # stderr and the process exits
    import faulthandler
    wd = int(os.environ.get("ADAP_BENCH_WATCHDOG", "900"))
    if wd > 0:
        faulthandler.dump_traceback_later(wd, exit=True)
    if os.environ.get("ADAP_BENCH_NATIVE_STACKS"):           # diagnostic: native stacks of a hung run, a few seconds before the watchdog
        import ctypes
        import subprocess
        import threading
        try:
            ctypes.CDLL("libc.so.6").prctl(0x59616d61, ctypes.c_ulong(-1), 0, 0, 0)      # PR_SET_PTRACER, PR_SET_PTRACER_ANY
        except OSError:
            pass

        def _native(pid=os.getpid(), after=max(5, wd - 25)):
            time.sleep(after)
            gdb = "/opt/rocm/bin/rocgdb" if os.path.exists("/opt/rocm/bin/rocgdb") else "gdb"
            subprocess.run([gdb, "-p", str(pid), "-batch", "-ex", "thread apply all bt 14"], stdout=sys.stderr, stderr=sys.stderr, timeout=20)
        threading.Thread(target=_native, daemon=True).start()

    emulated = None
    if args.emulate_node_share > 1:
        from adaprompt_amd import hostinfo
        share = hostinfo.cpu_share()
        ncpu = max(1, share // args.emulate_node_share)
        allowed = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, set(allowed[:ncpu]))          # (no HIP call has been made yet; child threads inherit the mask)
        torch.set_num_threads(ncpu)
        emulated = {"ranks": args.emulate_node_share, "cpus": ncpu, "of_cpu_share": share}

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks here, one process per GPU (the reference's launcher does the
        # same through Lightning, main.py:829 strategy="ddp").  This parent has made no HIP call and never will.
        raise SystemExit(spawn_ranks(args.gpus))

    from adaprompt_amd import _lib, ops
    from adaprompt_amd.parallel import GradReducer, init_distributed

    rank, world, local = init_distributed()
    if os.environ.get("ADAP_BENCH_KEEP_TORCH_THREADS") != "1":
        # torch's CPU pool at this rank's share of the host cores (N ranks share the node's; a one-GPU box shows 256 CPUs and
        # allows 16).  The step's host side is one thread, but an oversubscribed pool costs: two ranks on one box measured
        # 281 vs 786 ms/step with / without the cap
        from adaprompt_amd import hostinfo
        hostinfo.limit_torch_threads(cap=max(1, hostinfo.cpu_share() // world))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    _lib.load()

    ld, hook = build_model(device)
    params = list(hook.parameters())
    # the optimiser of the shipped config (v1-finetune-ada.yaml:59,74-84; ddpm.py:5207-5247): Prodigy lr=1, zero-shot
    # betas (0.9, 0.999), d_coef 2, bias correction, weight decay 0, warm-up 500 then one linear cycle to 60000 steps
    from adaprompt_amd.ldm.prodigy import Prodigy
    from adaprompt_amd.ldm.util import prodigy_linear_schedule
    opt = Prodigy(params, lr=1.0, betas=(0.9, 0.999), d_coef=2.0, use_bias_correction=True, weight_decay=0.0)
    reducer = GradReducer(params, flat=opt.grad_buffer)      # the exchange runs on the optimiser's flat buffer
    allreduce_bytes = reducer.bytes_per_reduce if world > 1 else 0
    dist_backend = dist.get_backend() if world > 1 else None
    sched = prodigy_linear_schedule(opt, max_steps=60000, warm_up_steps=500, scheduler_cycles=1)
    B = args.batch
    batches = [synthetic_batch(B, device, 1234 + rank * 100 + i) for i in range(2)]
    gen = torch.Generator(device=device).manual_seed(99 + rank)
    # the host RNG the step draws from (timestep annealing: one random.random() + np.random.randint per element, util.py:372-395):
    # seeded, so that a run's losses are a function of the command line alone (tools/determinism_check.sh repeats them digit for digit)
    import random as _py_random
    import numpy as _np
    _py_random.seed(4321 + rank)
    _np.random.seed(4321 + rank)
    torch.manual_seed(777 + rank)

    # ---- hipGraph capture of the launch-bound part (~1400 kernel launches per micro-batch): forward and backward
    # are two graphs, so the previous micro-batch's gradient all-reduce can be awaited between them.  Inputs live in
    # static buffers; RNG draws, the collective, clip and the optimizer step stay outside the graphs.
    static = {k: v.clone() for k, v in batches[0].items()}
    st_t = torch.zeros(B, device=device, dtype=torch.int64)
    st_noise = torch.zeros(B, 4, 64, 64, device=device)
    st_pn = torch.zeros(B, 4, 64, 64, device=device)
    st_x = torch.zeros(B, 4, 64, 64, device=device)          # --graph with the prefetch stream: the encoded latent of this step
    graphs = None

    def draw(i):
        batch = batches[i % 2]
        for k in static:
            static[k].copy_(batch[k])
        st_t.copy_(torch.randint(0, 1000, (B,), device=device, generator=gen))
        st_noise.copy_(torch.randn(B, 4, 64, 64, device=device, generator=gen))
        st_pn.copy_(torch.randn(B, 4, 64, 64, device=device, generator=gen))

    def fwd(latent_from_prefetch=False):
        # the all-reduce issued by the previous micro-batch overlaps this VAE encode + UNet forward and is
        # awaited just before this micro-batch's backward writes into the gradient buffer
        if latent_from_prefetch:        # (--graph: the VAE encode stays eager on the prefetch stream and hands its latent over here)
            return ld.shared_step(static, t=st_t, noise=st_noise, x_start=st_x)
        return ld.shared_step(static, t=st_t, noise=st_noise, post_noise=st_pn)

    # ---- default mode: eager launches, the next micro-batch's VAE encode prefetched on a second stream ----------
    prefetch = None if args.no_prefetch else ld.make_prefetcher()
    pf_state = {"next": 0}

    def pf_submit(inline=False):
        i = pf_state["next"]
        pn = torch.randn(B, 4, 64, 64, device=device, generator=gen)
        prefetch.submit(batches[i % 2], pn, inline=inline)
        pf_state["next"] = i + 1

    # latents encoded ahead of their use: 4 = two accumulation windows, submitted behind the window's backwards, so that the
    # encodes run in the window's tail, under the optimiser step and in the next window's head -- where one lane alone leaves
    # most of the chip idle (25.5 vs 25.75 ms; 2 = one window ahead, submitted behind the forwards)
    pf_depth = int(os.environ.get("ADAP_BENCH_PF_DEPTH", "4"))
    if prefetch is not None:
        pf_submit()
        if not (args.graph or args.no_lanes):
            pf_submit()            # two micro-batches ahead: both latents of a window are encoded while the previous one runs
            for _ in range(pf_depth - 2):
                pf_submit()

    def capture():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(2):                    # warm-up on the capture stream (weight packs, workspaces, BLAS handles)
                draw(i)
                loss, grad, out, aux = fwd(prefetch is not None)
                ld.manual_backward(out, grad, aux)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        reducer.zero()
        g_f, g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # the capture stream owns the single-launch GroupNorm path while the graphs are recorded (the decision is taken on the
        # host at capture time; the replayed kernels use the same exchange buffer on whatever stream replays them)
        ops.set_gn_single_launch_stream(device, side.cuda_stream)
        try:
            with torch.cuda.graph(g_f, stream=side):
                loss, grad, out, aux = fwd(prefetch is not None)
            with torch.cuda.graph(g_b, pool=g_f.pool(), stream=side):
                ld.manual_backward(out, grad, aux)
        finally:
            ops.set_gn_single_launch_stream(device, torch.cuda.default_stream(device).cuda_stream)
        torch.cuda.synchronize()
        reducer.zero()
        return g_f, g_b, loss

    if args.graph:
        try:
            graphs = capture()
        except Exception as e:          # noqa: BLE001 -- report and run eagerly; the result line says which mode ran
            if rank == 0:
                print(f"warning: hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphs = None
            torch.cuda.synchronize()
            reducer.zero()

    def step_prefetch(i):
        batch = batches[i % 2]
        t = torch.randint(0, 1000, (B,), device=device, generator=gen)
        noise = torch.randn(B, 4, 64, 64, device=device, generator=gen)
        x_start = prefetch.get()                       # encoded while the previous step's UNet pass ran
        loss, grad, out, aux = ld.shared_step(batch, t=t, noise=noise, x_start=x_start, anneal_t=True)
        # the next micro-batch's encode goes to the side stream once forward + losses are queued: it then fills the
        # CUs during the regularisers' many small kernels and the backward (measured +1 % over submitting it first)
        pf_submit()
        reducer.wait()
        ld.manual_backward(out, grad, aux)
        reducer.reduce()
        ld.batch_idx += 1
        if ld.batch_idx % ld.manual_accumulate_grad_batches == 0:
            reducer.wait()
            opt.step(clip_norm=ld.grad_clip)               # clip 0.5 fused into the step (ddpm.py:606-633)
            reducer.zero()
            sched.step()
        return loss

    # ---- the two micro-batches of an accumulation window on two streams (LatentDiffusion.training_window) -------------
    from adaprompt_amd.ldm.models.diffusion.ddpm import MicroBatchLanes
    use_lanes = (prefetch is not None and not args.graph and not args.no_lanes and ld.manual_accumulate_grad_batches == 2)
    lanes = MicroBatchLanes(params, n=2, reducer=reducer if world > 1 else None) if use_lanes else None

    fuse = use_lanes and (args.fuse or os.environ.get("ADAP_WINDOW_FUSE", "0") == "1")          # the window as one batched UNet pass
    diag_no_vae = os.environ.get("ADAP_DIAG_NO_VAE", "0") == "1"      # DIAGNOSTIC (invalid as a result): latents not encoded
    diag_x = torch.randn(B, 4, 64, 64, device=device) if diag_no_vae else None

    def window_prefetch(i):
        def draws(k):
            t = torch.randint(0, 1000, (B,), device=device, generator=gen)
            noise = torch.randn(B, 4, 64, 64, device=device, generator=gen)
            if diag_no_vae:
                return dict(t=t, noise=noise, x_start=diag_x, anneal_t=True)
            return dict(t=t, noise=noise, x_start=prefetch.get(), anneal_t=True)
        if diag_no_vae:
            return ld.training_window([batches[(i + k) % 2] for k in range(2)], opt, reducer, sched, lanes, step_kwargs=draws, fuse=fuse)[-1][0]
        # (measured alternatives, each 26.7 vs 25.6 ms: encoding the next latents on the micro-batch's own lane behind its forward or
        # its backward, or submitting them to the prefetch stream only once the backward is issued)
        if pf_depth == 4:        # tuning: two windows ahead, submitted behind the backwards (the encodes then run in the window's tail)
            out = ld.training_window([batches[(i + k) % 2] for k in range(2)], opt, reducer, sched, lanes, step_kwargs=draws,
                                     after_backward=lambda k: pf_submit(), fuse=fuse)
        else:
            out = ld.training_window([batches[(i + k) % 2] for k in range(2)], opt, reducer, sched, lanes, step_kwargs=draws,
                                     after_forward=lambda k: pf_submit(), fuse=fuse)
        return out[-1][0]

    def step(i):
        if prefetch is not None and ops.TIMER is None and graphs is None:
            return step_prefetch(i)
        draw(i)
        if graphs is not None:
            g_f, g_b, loss = graphs
            if prefetch is not None:
                st_x.copy_(prefetch.get())              # encoded on the prefetch stream while the previous replay ran
            g_f.replay()
            if prefetch is not None:
                pf_submit()
            reducer.wait()
            g_b.replay()
        else:
            loss, grad, out, aux = fwd()
            reducer.wait()
            ld.manual_backward(out, grad, aux)
        reducer.reduce()
        ld.batch_idx += 1
        if ld.batch_idx % ld.manual_accumulate_grad_batches == 0:
            reducer.wait()
            opt.step(clip_norm=ld.grad_clip)               # clip 0.5 fused into the step (ddpm.py:606-633)
            reducer.zero()
            sched.step()
        return loss

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def run_steps(first, n):
        """n micro-batches; with lanes, whole accumulation windows go through ``window_prefetch`` (two micro-batches on two
        streams, one optimiser step), a micro-batch off a window border through ``step``."""
        loss, i = None, first
        while i < first + n:
            if lanes is not None and ld.batch_idx % 2 == 0 and i + 2 <= first + n:
                loss = window_prefetch(i)
                i += 2
            else:
                loss = step(i)
                i += 1
        return loss

    def lightning_runner():
        from adaprompt_amd.trainer import Trainer
        tr = Trainer(max_steps=60000, every_n_train_steps=0, micro_batch_lanes=True,
                     prefetch_windows=int(os.environ.get("ADAP_ENTRY_PREFETCH_WINDOWS", "2")))
        tr.optimizer, tr.scheduler, tr.reducer = opt, sched, reducer
        tr.lanes = lanes                                   # (the lanes this process already has: streams are few and shared)
        object.__setattr__(ld, "trainer", tr)
        was = ld.composition_regs_iter_gap
        ld.composition_regs_iter_gap = 0                   # stage 1 as the bench runs it: recon iterations only
        st = {"i": 0, "loss": None}

        def run(n):
            for _ in range(n):
                loss_, _aux = ld.training_step(batches[st["i"] % 2], st["i"])
                st["i"] += 1
                if loss_ is not None:
                    st["loss"] = loss_
            return st["loss"]

        def close():
            ld.flush_window(run=False)
            ld.composition_regs_iter_gap = was
            object.__setattr__(ld, "trainer", None)
        return run, close

    window_run_steps = run_steps
    _close_l = None
    if args.entry == "lightning":
        if lanes is None:
            raise SystemExit("--entry lightning needs the lanes (no --no-lanes / --graph / --no-prefetch)")
        _run_l, _close_l = lightning_runner()
        _run_l(6)          # untimed: the entry keeps two windows of batches buffered ahead, so its first calls run nothing

        def run_steps(first, n):                            # noqa: F811 -- the timed loop through the per-batch entry
            return _run_l(n)

    if os.environ.get("ADAP_DIAG_WARM_ONE_STREAM") == "1" and lanes is not None:
        # DIAGNOSTIC (tools/lanes_two_process_soak.sh): one accumulation window on ONE stream first, so that every lazily built
        # cache is filled and complete before the lanes start -- separates first-use races from steady-state ones
        for i_ in range(2):
            step_prefetch(i_)
        torch.cuda.synchronize()
    run_steps(0, args.warmup)
    if lanes is not None and ld.batch_idx % 2 == 1:
        # an odd warm-up ends in the middle of an accumulation window: close it here, untimed (clip + optimiser step on the one
        # micro-batch it holds), so that the timed region is whole windows -- K micro-batches and K / 2 optimiser steps either way
        reducer.wait()
        opt.step(clip_norm=ld.grad_clip)
        reducer.zero()
        sched.step()
        ld.batch_idx += 1
    sync()
    t0 = time.perf_counter()
    loss = run_steps(0, args.steps)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    loss_val = float(loss)
    if _close_l is not None:
        _close_l()
    # diagnostic, outside the timed region: the HOST's own work per step -- one step issued into an EMPTY launch queue (a
    # synchronisation first), so the host never waits for the GPU; median of three.  (Timing the issue of the K timed steps
    # says nothing: the queue holds ~55 ms of this workload, and once it is full the host advances at the GPU's pace.)
    host_ms = []
    for i in range(3):
        sync()
        th = time.perf_counter()
        step(args.steps + i)
        host_ms.append(1e3 * (time.perf_counter() - th))
    sync()
    host_work_ms = sorted(host_ms)[1]

    def timed_leg(run, n):
        sync()
        t1 = time.perf_counter()
        run(n)
        sync()
        return time.perf_counter() - t1

    if lanes is not None and ld.batch_idx % 2 == 1:
        step(0)                 # (the host-work probe's three steps left an accumulation window open: close it for the legs below)
    n_leg = max(2, args.steps // 2 * 2)                      # the legs time whole windows

    # ---- extra leg: the data-parallel exchange REHEARSED on one GPU (VERDICT r4 #1c).  8-GPU nodes are the driver's to run; what
    # one GPU can show is what the step costs with a collective's kernel resident beside it: after every micro-batch backward a
    # kernel of RCCL's footprint (--rehearse-blocks workgroups x 512 threads, ~100 registers) occupies its CUs on the exchange
    # stream for the time a ring all-reduce of the gradient payload takes over xGMI -- 2 (N-1)/N x bytes through 7 links x 153 GB/s
    # (MI355X_MICROARCH.md) -- and the lanes' gate / the optimiser step wait for it exactly as they wait for the real one.
    rehearsal = None
    if rank == 0 and world == 1 and lanes is not None and not args.no_rehearse_exchange and ld.batch_idx % 2 == 0:
        class RehearsedExchange:
            """``GradReducer``'s interface with a resident stand-in kernel where the collective would run."""
            world = 1

            def __init__(self, flat, nbytes, ranks, blocks):
                self.flat, self.bytes_per_reduce = flat, nbytes
                self.usec = int(round(1e6 * nbytes * 2.0 * (ranks - 1) / ranks / (7 * 153e9)))
                self.blocks, self.ranks = blocks, ranks
                self.side = torch.cuda.Stream()
                self.sink = torch.zeros(1, device=flat.device)
                self._done = None
                self.launches = 0

            def begin_backward(self):
                pass

            def reduce(self):
                ev = torch.cuda.Event()
                ev.record()
                self.side.wait_event(ev)              # the gradients the collective would read
                with torch.cuda.stream(self.side):
                    _lib.call("adap_debug_occupy", self.blocks, 512, self.usec, self.sink.data_ptr(), _lib.current_stream())
                    self._done = torch.cuda.Event()
                    self._done.record()
                self.launches += 1

            def wait(self):
                if self._done is not None:
                    torch.cuda.current_stream().wait_event(self._done)
                    self._done = None

            @property
            def pending(self):
                return self._done is not None

            def zero(self):
                self.wait()
                self.flat.zero_()

        real_reducer = reducer
        reh = RehearsedExchange(opt.grad_buffer, real_reducer.bytes_per_reduce, args.rehearse_ranks, args.rehearse_blocks)
        reducer, lanes.reducer = reh, reh
        window_run_steps(0, 4)
        d_reh = timed_leg(lambda n: window_run_steps(0, n), n_leg)
        reducer, lanes.reducer = real_reducer, None
        d_ref = timed_leg(lambda n: window_run_steps(0, n), n_leg)    # the same loop again without it, back to back
        rehearsal = {"what": "REHEARSAL on one GPU, not a measurement of N GPUs: a resident kernel stands in for RCCL's ring all-reduce "
                             "of the gradient payload after every micro-batch backward, on the exchange stream, awaited by the lanes' gate "
                             "and the optimiser step like the real collective",
                     "ranks_priced": reh.ranks, "payload_bytes": int(reh.bytes_per_reduce), "collective_us": reh.usec,
                     "standin_kernel": f"{reh.blocks} workgroups x 512 threads x ~100 VGPRs", "collectives_launched": reh.launches,
                     "ms_per_step_with_exchange": round(1e3 * d_reh / n_leg, 3),
                     "ms_per_step_without": round(1e3 * d_ref / n_leg, 3),
                     "predicted_scaling_efficiency": round(d_ref / d_reh, 4),
                     "predicted_speedup_at_ranks": round(reh.ranks * d_ref / d_reh, 2),
                     "unfrozen_payload_collective_us": int(round(1e6 * 4.5e9 * 2.0 * (reh.ranks - 1) / reh.ranks / (7 * 153e9)))}

    # ---- extra leg: the other ENTRY.  'lightning': training_step(batch, batch_idx) per micro-batch with a Trainer
    # attached -- the call a drop-in user's Lightning loop makes; the micro-batches of a window are buffered, the window runs on the
    # lanes when its last one arrives, one window of latents is prefetched ahead.
    entry_leg = None

    if rank == 0 and world == 1 and lanes is not None and not args.no_entry_leg and ld.batch_idx % 2 == 0 and args.entry == "window":
        run_l, close_l = lightning_runner()
        run_l(8)                                            # fills the look-ahead buffer, warms the global generator's path
        d_l = timed_leg(run_l, n_leg)
        close_l()
        d_w = timed_leg(lambda n: window_run_steps(0, n), n_leg)
        entry_leg = {"entry": "training_step(batch, batch_idx) per micro-batch, Trainer attached (lanes, 2 windows of latents ahead)",
                     "ms_per_step": round(1e3 * d_l / n_leg, 3), "images_per_sec": round(B * n_leg / d_l, 2),
                     "window_entry_ms_per_step_back_to_back": round(1e3 * d_w / n_leg, 3),
                     "ratio_to_window_entry": round(d_l / d_w, 4)}

    roofline = None
    if not args.no_roofline:
        # live HIP-event timing of the dominant kernel family on the same workload (2 extra, separately run steps)
        ops.TIMER = ops.KernelTimer()
        saved_graphs, graphs = graphs, None          # per-launch events need eager launches
        for i in range(2):
            step(i)
        summ = ops.TIMER.summary()
        ops.TIMER = None
        graphs = saved_graphs
        # the dominant kernel = the contraction kernel (by symbol name, as rocprofv3 --stats lists it) with the most time
        convs = {k: v for k, v in summ.items() if k.startswith("conv")}
        dom = max(convs, key=lambda k: convs[k]["ms"])
        c = convs[dom]
        tf = c["work"] / (c["ms"] * 1e-3) / 1e12
        call = {"launches_per_step": 0, "ms": 0.0, "work": 0.0}
        for v in convs.values():
            call["launches_per_step"] += v["launches"] // 2
            call["ms"] += v["ms"] / 2
            call["work"] += v["work"] / 2

        def fam(v, unit):
            return {"launches_per_step": v["launches"] // 2, "ms_per_step": round(v["ms"] / 2, 3),
                    "avg_launch_us": round(1e3 * v["ms"] / v["launches"], 1),
                    "achieved": round(v["work"] / (v["ms"] * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9), 1), "unit": unit}
        traffic, traffic_src = pmc_traffic(dom)
        clock = in_kernel_clock(device) if (rank == 0 and not args.no_clock_probe) else None
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(tf, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                    "traffic_source": traffic_src,
                    "launches_per_step": c["launches"] // 2, "avg_launch_us": round(1e3 * c["ms"] / c["launches"], 1),
                    "ms_per_step_in_kernel": round(c["ms"] / 2, 3), "in_kernel_clock": clock,
                    "note": "algorithmic 2*MAC FLOPs of the launches dispatched to this kernel / their HIP-event time on the "
                            "launch stream; per-launch events add ~10 us to short launches, rocprofv3 durations in profiles/",
                    "all_contraction_kernels": {"launches_per_step": call["launches_per_step"],
                                                "ms_per_step": round(call["ms"], 3),
                                                "achieved": round(call["work"] / (call["ms"] * 1e-3) / 1e12, 1), "unit": "TFLOP/s"},
                    "others": {**{k: fam(v, "TFLOP/s") for k, v in summ.items() if k != dom and (k.startswith("conv") or "attention" in k)},
                               **{k: fam(v, "GB/s") for k, v in summ.items() if k.startswith("groupnorm") or k.startswith("hbm:")}}}

    # ---- the north star's two named aggregates, measured in isolation on the step's own modules (HIP events, 20
    # launches each): self-attention at the 64x64 level against the bf16 MFMA peak, and one 320->320 ResBlock at
    # 64x64 against the HBM peak over its COMPULSORY bytes (read x, write out, both 3x3 weight sets, the
    # embedding row; everything between stays on chip in the ideal).  The second number is reported because the
    # north star asks for it; with 60 GFLOP per 49 MB the block sits far on the compute side of the ridge, so its
    # HBM fraction is structurally tiny (DESIGN.md section 3 / SURVEY 7 "roofline honesty").
    aggregates = None
    if rank == 0 and not args.no_roofline and not args.no_aggregates:
        def timed(fn, n=50):
            for _ in range(20):          # past the first launches' clock ramp: the rate the kernel holds inside the step
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e-3
        C, N = 320, 4096
        qkv = (torch.randn(B, N, 3 * C, device=device, generator=gen)).to(torch.bfloat16)
        q_, k_, v_ = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        # the form the training step calls: with PRESCALE_Q the query projection's pack carries d^-1/2 * log2(e) and the kernels
        # are asked for scale = 0 (functional.py); the operands here are random either way
        from adaprompt_amd import functional as HF
        att_scale = 0.0 if HF.PRESCALE_Q else None
        o_, lse_ = ops.attention_fwd(q_, k_, v_, 8, scale=att_scale)
        do_ = torch.randn(B, N, C, device=device, generator=gen).to(torch.bfloat16)
        t_f = timed(lambda: ops.attention_fwd(q_, k_, v_, 8, scale=att_scale))
        t_b = timed(lambda: ops.attention_bwd(q_, k_, v_, o_, do_, lse_, 8, scale=att_scale))
        fl = 4.0 * B * 8 * N * N * 40
        rb = ld.model.diffusion_model.input_blocks[1][0]
        xr = torch.randn(B, 64, 64, C, device=device, generator=gen)
        er = torch.randn(B, 4 * C, device=device, generator=gen)
        with torch.no_grad():
            t_rf = timed(lambda: rb(xr, er))
        xg = xr.clone().requires_grad_(True)
        go = torch.randn(B, 64, 64, C, device=device, generator=gen)

        def rb_fb():
            xg.grad = None
            rb(xg, er).backward(go)
        t_rfb = timed(rb_fb)
        rb_bytes = 4.0 * (2 * B * 64 * 64 * C + 2 * 9 * C * C + B * 4 * C)
        rb_flops = 2.0 * 2 * B * 64 * 64 * C * C * 9
        # the HBM-bound part of that block, where the north star's HBM target applies: GroupNorm32 + SiLU (f32 in -> bf16 out)
        gw, gb_ = torch.ones(C, device=device), torch.zeros(C, device=device)
        def graph_timed(fn, n=20):
            """GPU time per call of a ~10 us kernel: ``n`` calls captured into one hipGraph and replayed (issued eagerly from
            Python the loop measures the host -- torch.empty x 5 + ctypes is ~15 us per call; inside the training step these
            launches are queued well ahead of the GPU).  The capture stream is made the owner of the single-launch path for the
            measurement and the default stream gets it back afterwards."""
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            ops.set_gn_single_launch_stream(device, side.cuda_stream)
            try:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        for _ in range(n):
                            fn()
                    for _ in range(5):
                        g.replay()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        g.replay()
                    e1.record()
                torch.cuda.synchronize()
            finally:
                ops.set_gn_single_launch_stream(device, torch.cuda.default_stream(device).cuda_stream)
            return e0.elapsed_time(e1) / (5 * n) * 1e-3
        dy16 = go.to(torch.bfloat16)
        _, _, gm_, gr_ = ops.groupnorm_fwd(xr, gw, gb_, 1e-5, 1)
        gn_bytes_f, gn_bytes_b = xr.numel() * (4 + 2), xr.numel() * (4 + 2 + 2)       # algorithmic: x once + y / x, dy once + dx

        def gn_aggregate(xr=xr, dy16=dy16, gm_=gm_, gr_=gr_):
            """The two graph-timed GroupNorm numbers.  Run AFTER every other leg: a hipGraph capture in the process (torch empties its
            allocator cache and re-registers the generator for it) left config 2's mix 15 % slower when it came first (64 vs 75
            img/s, bisected leg by leg)."""
            t_gn = graph_timed(lambda: ops.groupnorm_fwd(xr, gw, gb_, 1e-5, 1))
            t_gnb = graph_timed(lambda: ops.groupnorm_bwd(dy16, xr, gw, gb_, gm_, gr_, 1, out_f32=False, out_bf16=True))
            return {"fwd_us": round(t_gn * 1e6, 1), "bwd_us": round(t_gnb * 1e6, 1),
                    "algorithmic_bytes_fwd": int(gn_bytes_f), "algorithmic_bytes_bwd": int(gn_bytes_b),
                    "fwd_GBps": round(gn_bytes_f / t_gn / 1e9, 1), "bwd_GBps": round(gn_bytes_b / t_gnb / 1e9, 1),
                    "fwd_frac_of_hbm_peak": round(gn_bytes_f / t_gn / 8e12, 4),
                    "bwd_frac_of_hbm_peak": round(gn_bytes_b / t_gnb / 8e12, 4),
                    "note": "ONE launch each (norms.hip gn_fused_*: x read once, the sample's workgroups exchange their "
                            "group partials through tagged 8-byte records inside the launch); in-kernel stamps: ~3 us load, "
                            "~4 us hand-off (publish + sweep = memory-side round trips), ~1.2 us finish, ~1.2 us store issue"}
        aggregates = {
            "attention_self_64x64": {"shape": f"B{B} h8 N{N} d40", "form": "pre-scaled queries (scale = 0), as the step calls it"
                                     if att_scale == 0.0 else "scale applied in the kernel",
                                     "fwd_us": round(t_f * 1e6, 1), "bwd_us": round(t_b * 1e6, 1),
                                     "fwd_tflops": round(fl / t_f / 1e12, 1), "bwd_tflops": round(2.5 * fl / t_b / 1e12, 1),
                                     "fwd_frac_of_bf16_mfma_peak": round(fl / t_f / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                                     "bwd_frac_of_bf16_mfma_peak": round(2.5 * fl / t_b / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)},
            "resblock_320_64x64": {"fwd_us": round(t_rf * 1e6, 1), "fwd_bwd_us": round(t_rfb * 1e6, 1),
                                   "compulsory_bytes_fwd": int(rb_bytes),
                                   "fwd_frac_of_hbm_peak": round(rb_bytes / t_rf / 8e12, 4),
                                   "fwd_tflops": round(rb_flops / t_rf / 1e12, 1),
                                   "fwd_frac_of_bf16_mfma_peak": round(rb_flops / t_rf / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)},
            "groupnorm_silu_320_64x64": gn_aggregate,          # (measured last, see gn_aggregate)
        }
        del qkv, o_, lse_, do_, xr, er, xg, go, dy16

    # ---- extra leg, reported beside `value`: config 2's actual iteration mix -- VAE encode of 4, then the
    # Arc2Face teacher rolls ND in {1,3,5,7} steps out on HALF_BS instances and the student is distilled on them
    distill = None
    if world == 1 and not args.no_distill_mix:
        import numpy as np
        from adaprompt_amd import synth
        from adaprompt_amd.ldm.models.diffusion.ddpm import Arc2FaceWrapper, LatentDiffusion
        with torch.device(device):
            teacher = Arc2FaceWrapper(unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel",
                                                   "params": dict(synth.SD15_UNET)})
        teacher.unet.load_state_dict(device_state_dict(synth.unet_param_shapes(**synth.SD15_UNET), "", device, 5))
        ld.set_arc2face_teacher(teacher.to(device).eval())
        rs = np.random.RandomState(0)
        nd_seq = [LatentDiffusion.draw_num_denoising_steps(7, rs) for _ in range(12)]
        tctx = torch.randn(B, 21, 768, device=device, generator=gen) * 0.05

        # the no-grad front of the NEXT micro-batch (VAE encode + teacher rollout on 1-2 instances) runs on a side stream
        # under the current micro-batch's student pass
        dpf = None if args.no_prefetch else ld.make_distill_prefetcher()
        seq_all = [1, 3, 5, 7] + nd_seq

        def dsubmit(j):
            if dpf is None or j >= len(seq_all):
                return
            b = dict(batches[j % 2])
            b["arc2face_prompt_emb"] = tctx
            dpf.submit(b, torch.randn(B, 4, 64, 64, device=device, generator=gen),
                       torch.randint(0, 1000, (B,), device=device, generator=gen),
                       torch.randn(B, 4, 64, 64, device=device, generator=gen), seq_all[j])

        dstate = {"j": 0}
        dsubmit(0)

        def dstep(i, nd):
            batch = dict(batches[i % 2])
            batch["arc2face_prompt_emb"] = tctx
            j = dstate["j"]
            dstate["j"] = j + 1
            if dpf is not None:
                assert seq_all[j] == nd
                x_start, t, noise, teacher_out, hb = dpf.get()
                dsubmit(j + 1)
                batch = {k: (v[:hb] if torch.is_tensor(v) and v.dim() > 0 else v) for k, v in batch.items()}
                loss, grads, outs, aux = ld.shared_step(batch, t=t, noise=noise, num_denoising_steps=nd,
                                                        use_arc2face_as_target=True, x_start=x_start,
                                                        trim_to_half_batch=False, teacher_out=teacher_out)
            else:
                t = torch.randint(0, 1000, (B,), device=device, generator=gen)
                noise = torch.randn(B, 4, 64, 64, device=device, generator=gen)
                loss, grads, outs, aux = ld.shared_step(batch, t=t, noise=noise, num_denoising_steps=nd,
                                                        use_arc2face_as_target=True)
            reducer.wait()
            torch.autograd.backward(outs, grads)
            reducer.reduce()
            ld.batch_idx += 1
            if ld.batch_idx % ld.manual_accumulate_grad_batches == 0:
                reducer.wait()
                opt.step(clip_norm=ld.grad_clip)
                reducer.zero()
                sched.step()
            return len(aux["model_outputs_per_step"])

        for i, nd in enumerate((1, 3, 5, 7)):
            dstep(i, nd)
        sync()
        t1 = time.perf_counter()
        student_passes = sum(dstep(i, nd) for i, nd in enumerate(nd_seq))
        sync()
        dd = time.perf_counter() - t1
        distill = {"workload": "config 2 iteration mix: VAE encode of 4 + Arc2Face teacher rollout (full SD-1.5 UNet on the "
                               "same kernels, no grad) of ND steps on HALF_BS instances + student fwd/bwd on the teacher's "
                               "predictions (the student's ND passes batched into one; the next micro-batch's VAE encode + teacher rollout prefetched on a side stream); ND drawn from {1,3,5,7} p=(.4,.3,.2,.1), seed 0",
                   "nd_sequence": nd_seq, "micro_batches": len(nd_seq),
                   "images_per_sec": round(B * len(nd_seq) / dd, 2), "ms_per_micro_batch": round(1e3 * dd / len(nd_seq), 2),
                   "teacher_unet_passes": int(sum(nd_seq)), "student_unet_fwd_bwd_passes": int(student_passes)}
        ld.set_arc2face_teacher(None)
        del teacher

    def leg_reset():
        """between the extra legs (never inside a timed region): hand the caching allocator's blocks back.  The legs build
        further models (teacher UNet, VAE decoder, optimiser state for 1.13 B values) with allocation patterns of their own;
        carried over, the earlier legs' cached blocks left the last leg 15 % slower (50 vs 59 img/s unfrozen, and only when every
        leg had run before it)."""
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    # ---- extra leg: config 4 -- Stage-2 compositional distillation micro-batches at bs=3
    compos = None
    if world == 1 and not args.no_compos:
        leg_reset()
        compos = compos_leg(device, gen)

    # ---- extra leg: the zero-shot feature front end's image encoder, which runs in every zero-shot iteration
    zs_front = None
    if world == 1 and not args.no_zs_frontend:
        leg_reset()
        zs_front = zs_frontend_leg(device, gen)

    # ---- extra leg: config 5 -- 50 DDIM steps at bs=8 (UNet batch 16 under classifier-free guidance,
    # context [256,77,768]) and the VAE decode of the 8 latents
    ddim = None
    if world == 1 and not args.no_ddim:
        leg_reset()
        from adaprompt_amd import synth
        from adaprompt_amd.ldm.models.diffusion.ddim import DDIMSampler
        dec = ld.first_stage_model.build_decoder()
        dsd = device_state_dict(synth.vae_decoder_param_shapes(**synth.SD15_VAE_DD), "", device, 9)
        ld.first_stage_model.load_state_dict(dsd, strict=False)
        del dsd
        nb, S = 8, 50
        ex = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
              "is_training": False, "capture_distill_attn": False, "img_mask": None}
        c = (torch.randn(16 * nb, 77, 768, device=device, generator=gen) * 0.05, ["a photo of z"] * nb, dict(ex))
        uc = (torch.randn(16 * nb, 77, 768, device=device, generator=gen) * 0.05, [""] * nb, dict(ex))
        sampler = DDIMSampler(ld)

        def gen_images(steps):
            xT = torch.randn(nb, 4, 64, 64, device=device, generator=gen)
            with torch.no_grad():
                z, _ = sampler.sample(S=steps, batch_size=nb, shape=[4, 64, 64], conditioning=c, verbose=False,
                                      guidance_scale=[10, 4], unconditional_conditioning=uc, eta=0.0, x_T=xT)
                return ld.decode_first_stage(z)

        gen_images(2)                     # warm-up: weight packs of the decoder, batch-16 workspaces
        sync()
        t1 = time.perf_counter()
        img = gen_images(S)
        sync()
        dd_ = time.perf_counter() - t1
        ddim = {"workload": "config 5: 50-step DDIM (eta 0, guidance annealed 10 -> 4, UNet batch 16 = 8 cond + 8 uncond, "
                            "context [256,77,768]) + VAE decode of 8 latents to 512x512",
                "images": nb, "steps": S, "seconds": round(dd_, 3), "sec_per_image": round(dd_ / nb, 4),
                "images_per_sec": round(nb / dd_, 3), "finite": bool(torch.isfinite(img).all())}
        del dec, sampler, c, uc, img

    # ---- config 3's second payload: `unfreeze_model: True` -- the UNet's 859.5 M parameters train too.
    # Every block backward also produces weight gradients (csrc/wgrad.hip), Prodigy steps over hook + UNet + a
    # CLIP-text-sized stand-in bucket (123 M, the third member of the reference's model group, ddpm.py:5179), and the
    # per-micro-batch all-reduce carries all of it (~4.5 GB).  Runs last: it moves the UNet's weights.
    unfrozen = None
    if not args.no_unfrozen:
        leg_reset()
        for p_ in ld.model.parameters():
            p_.requires_grad_(True)
        clip_standin = torch.nn.Parameter(torch.zeros(123_060_480, device=device))
        groups = [{"params": list(hook.parameters())}, {"params": list(ld.model.parameters()) + [clip_standin]}]
        del opt, reducer, sched
        torch.cuda.empty_cache()
        opt_u = Prodigy(groups, lr=1.0, betas=(0.9, 0.999), d_coef=2.0, use_bias_correction=True, weight_decay=0.0)
        all_u = [q for g_ in groups for q in g_["params"]]
        red_u = GradReducer(all_u, flat=opt_u.grad_buffer)
        sched_u = prodigy_linear_schedule(opt_u, max_steps=60000, warm_up_steps=500, scheduler_cycles=1)
        pf_u = ld.make_prefetcher()

        def usubmit(j):
            pf_u.submit(batches[j % 2], torch.randn(B, 4, 64, 64, device=device, generator=gen))

        def ustep(i):
            t = torch.randint(0, 1000, (B,), device=device, generator=gen)
            noise = torch.randn(B, 4, 64, 64, device=device, generator=gen)
            x_start = pf_u.get()
            loss, grad, out, aux = ld.shared_step(batches[i % 2], t=t, noise=noise, x_start=x_start, anneal_t=True)
            usubmit(i + 1)
            red_u.wait()
            red_u.begin_backward()          # UNet chunks are exchanged while the backward still runs
            ld.manual_backward(out, grad, aux)
            red_u.reduce()
            if (i + 1) % ld.manual_accumulate_grad_batches == 0:
                red_u.wait()
                opt_u.step(clip_norm=ld.grad_clip)
                red_u.zero()
                sched_u.step()
            return loss

        def uwindow(i):
            # the window's two micro-batches through the UNet as ONE batched pass (training_window's default where lanes cannot
            # be used -- weight gradients are written from the start of a backward): one weight-gradient pass per window
            def draws(k):
                return dict(t=torch.randint(0, 1000, (B,), device=device, generator=gen),
                            noise=torch.randn(B, 4, 64, 64, device=device, generator=gen), x_start=pf_u.get(), anneal_t=True)
            out = ld.training_window([batches[(i + k) % 2] for k in range(2)], opt_u, red_u, sched_u, None, step_kwargs=draws,
                                     after_backward=lambda k: usubmit(i + 2 + k))
            return out[-1][0]

        u_fused = ld.manual_accumulate_grad_batches == 2 and not args.no_lanes and os.environ.get("ADAP_WINDOW_FUSE", "1") != "0"
        UW, UK = 2, 6
        if u_fused:
            ld.batch_idx = 0
            usubmit(0)
            usubmit(1)
            for i in range(0, UW, 2):
                uwindow(i)
        else:
            usubmit(0)
            for i in range(UW):
                ustep(i)
        sync()
        t1 = time.perf_counter()
        if u_fused:
            for i in range(UW, UW + UK, 2):
                lu = uwindow(i)
        else:
            for i in range(UW, UW + UK):
                lu = ustep(i)
        red_u.wait()
        sync()
        du = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([du], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            du = float(tt.item())
        nparam = sum(q.numel() for q in all_u)
        unfrozen = {"workload": "config 3, second payload: the same micro-batch with `unfreeze_model: True` -- weight gradients of "
                                "all 686 UNet tensors, Prodigy over hook + UNet + a 123 M CLIP-text stand-in, all of it all-reduced",
                    "trainable_params": nparam, "grad_allreduce_bytes": 4 * nparam if world > 1 else 0,
                    "window_fused": bool(u_fused),
                    "steps": UK, "ms_per_step": round(1e3 * du / UK, 2), "images_per_sec": round(world * B * UK / du, 2),
                    "final_loss": round(float(lu), 5), "finite": bool(torch.isfinite(opt_u.param_buffer).all())}
        del opt_u, red_u, sched_u, pf_u, clip_standin, groups, all_u
        opt = reducer = sched = None

    cpu = None
    # a single-launch GroupNorm that ever gave up waiting for its sample's other workgroups would have produced garbage: every
    # rank checks its device's flag, and a set flag fails the run instead of reporting a number
    if ops.gn_sync_poisoned():
        raise SystemExit(f"rank {rank}: the single-launch GroupNorm's exchange timed out (ops.gn_sync_poisoned): results invalid")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the GPU box gives one GPU a 16-core CPU share; more torch threads than that only oversubscribe
        from adaprompt_amd import hostinfo
        threads = args.cpu_threads or min(16, hostinfo.cpu_share())
        del ld, hook, reducer, opt, sched
        torch.cuda.empty_cache()
        cpu = cpu_baseline(threads, batch=args.cpu_batch)
        if threads > 8 and not args.no_cpu_k8:
            # the survey container's figure (0.048 img/s for the reference's modules) was taken on 8
            # cores: one more bounded sample on 8 threads for comparability (1 warm-up + 1 timed, ~25 s)
            k8 = cpu_baseline(8, batch=1, timed=1)
            cpu["k8_threads_sample"] = {"value": k8["value"], "unit": k8["unit"], "cores": 8,
                                        "seconds_per_micro_batch": k8["seconds_per_micro_batch"],
                                        "achieved_gflops": k8["achieved_gflops"], "sample": "bs=1, 1 warm-up + 1 timed"}

    if rank == 0:
        imgs = world * B * args.steps
        ms = 1e3 * dt / args.steps
        res = {
            "metric": "SD-1.5 UNet training images/sec @512px bs=4/GPU",
            "value": round(imgs / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic", "hipgraph": graphs is not None, "vae_prefetch_stream": prefetch is not None,
            "micro_batch_lanes": lanes is not None, "window_fused": bool(fuse), **({"DIAGNOSTIC_INVALID": "no VAE encode"} if diag_no_vae else {}),
            **({"emulated_node_share": emulated} if emulated else {}),
            "config": {"workload": "Stage-1 AdaFace recon distillation micro-batch, full SD-1.5 UNet (859.5M, frozen) + VAE "
                                   "encoder, 512x512, 16-layer layerwise context [64,77,768], img_mask + distill-attn capture, masked MSE + "
                                   "fg/bg complementary loss with its mask hinges + cross-layer attention consistency (both with the "
                                   "gradient through the captured attnscore) + prompt-delta loss, "
                                   "hook stand-in with 149M trainable fp32 params, clip 0.5 + Prodigy step + LR schedule every 2nd micro-batch",
                       "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "grad_allreduce_bytes": allreduce_bytes, "dist_backend": dist_backend},
            "model_tflops_per_step": round(GFLOP_PER_IMAGE * B / 1e3, 2),
            "achieved_model_tflops_per_gpu": round(GFLOP_PER_IMAGE * B / 1e3 / (ms * 1e-3), 1),
            "final_loss": round(loss_val, 5),
            # diagnostic: the host's own work per step (see above).  Well below ms_per_step: the GPU is the bottleneck; close to
            # it: this run was bound by the host's single-thread speed
            "host_work_ms_per_step": round(host_work_ms, 3),
        }
        if roofline is not None:
            res["roofline"] = roofline
        if cpu is not None:
            res["cpu_baseline"] = cpu
        if aggregates is not None:
            aggregates["groupnorm_silu_320_64x64"] = aggregates["groupnorm_silu_320_64x64"]()
            res["north_star_aggregates"] = aggregates
        res["entry"] = args.entry
        if rehearsal is not None:
            res["exchange_rehearsal"] = rehearsal
        if entry_leg is not None:
            res["entry_lightning"] = entry_leg
        if distill is not None:
            res["config2_distill_mix"] = distill
        if compos is not None:
            res["config4_compos"] = compos
        if zs_front is not None:
            res["f4_zero_shot_frontend"] = zs_front
        if ddim is not None:
            res["config5_ddim"] = ddim
        if unfrozen is not None:
            res["config3_unfrozen"] = unfrozen
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def compos_leg(device, gen, micro_batches=8):
    """BASELINE config 4 on one GPU: Stage-2 compositional (prompt-mix) distillation micro-batches, bs = 3, full SD-1.5 UNet
    (frozen) + VAE encoder and decoder.  A fresh iteration is: the no-grad teacher-filter pass of 2 candidates x (subject comp,
    mix comp) under classifier-free guidance (UNet batch 8, 154-token split K/V context), VAE decode of the 4 images, scoring,
    then the with-grad pass of the selected candidate under the four contexts (UNet batch 4), the stage-2 losses on the
    captured outfeat / attnscore / q of 12 layers and the backward through all of it into the embedding manager; a reuse
    iteration starts from the cached prediction and skips the filter.  The text encoder, the embedding manager and the
    CLIP scorer are third-party / boundary callees: stand-ins from adaprompt_amd/standins.py and a scripted score."""
    import random as _random
    from adaprompt_amd import synth
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    B = 3
    with torch.device(device):
        ld = LatentDiffusion(
            first_stage_config={"target": "ldm.models.autoencoder.AutoencoderKL",
                                "params": {"ddconfig": dict(synth.SD15_VAE_DD), "embed_dim": 4, "with_decoder": True}},
            cond_stage_config={"target": "adaprompt_amd.standins.StubTextEncoder", "params": {"dim": 768}},
            personalization_config={"target": "adaprompt_amd.standins.StubEmbeddingManager", "params": {"dim": 768, "num_vectors_per_subj_token": 16}},
            unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(synth.SD15_UNET)},
            scale_factor=0.18215, linear_start=0.00085, linear_end=0.012, conditioning_key="crossattn", cond_stage_trainable=True,
            use_layerwise_embedding=True, do_zero_shot=True, mix_prompt_distill_weight=1e-4, comp_fg_bg_preserve_loss_weight=1e-3,
            prompt_emb_delta_reg_weight=2e-4, normalize_ca_q_and_outfeat=True, num_candidate_teachers=2, composition_regs_iter_gap=3)
    ld = ld.to(device)
    sd = device_state_dict(synth.unet_param_shapes(**synth.SD15_UNET), "model.diffusion_model.", device, 0)
    sd.update(device_state_dict(synth.vae_encoder_param_shapes(**synth.SD15_VAE_DD), "first_stage_model.", device, 1))
    sd.update(device_state_dict(synth.vae_decoder_param_shapes(**synth.SD15_VAE_DD), "first_stage_model.", device, 9))
    missing, unexpected = ld.load_state_dict(sd, strict=False)
    assert not unexpected
    del sd
    params = [p for p in ld.embedding_manager.parameters() if p.requires_grad]
    for p in params:
        p.data.mul_(0.05)

    def score(prompts, images):          # scripted CLIP similarity: the mixed-prompt image always beats the subject one
        n = images.shape[0]
        base = torch.tensor([0.30, 0.31, 0.25, 0.22] if n == 4 else [0.30, 0.22], device=images.device)
        return 0.5 - (base + 0.02 * torch.tanh(images.float()[:, :, ::8, ::8].mean(dim=(1, 2, 3))))

    ld.clip_score_fn = score
    ld.training_percent = 0.3
    batch = synthetic_batch(B, device, 4321)
    ss, sc = "a photo of z", "a photo of z dancing in a park"
    cs, cc = "a photo of person", "a photo of person dancing in a park"
    bg = " with background y"
    batch.update(subject_name=["alice"] * B, is_in_mix_subj_folder=[False] * B, has_fg_mask=torch.ones(B, dtype=torch.bool, device=device),
                 has_wds_comp=torch.zeros(B, dtype=torch.bool, device=device),
                 zs_clip_features=torch.randn(B, 514, 768, device=device, generator=gen) * 0.1,
                 caption=[ss] * B, caption_bg=[ss + bg] * B, subj_prompt_single=[ss] * B, subj_prompt_comp=[sc] * B,
                 cls_prompt_single=[cs] * B, cls_prompt_comp=[cc] * B, subj_prompt_single_bg=[ss + bg] * B,
                 subj_prompt_comp_bg=[sc + bg] * B, cls_prompt_single_bg=[cs + bg] * B, cls_prompt_comp_bg=[cc + bg] * B)
    kinds = []

    def cstep():
        ld.init_iteration_flags()
        ld.iter_flags.update(do_mix_prompt_distillation=True, do_ada_prompt_delta_reg=True, is_compos_iter=True, calc_clip_loss=True,
                             do_normal_recon=False)
        loss, grads, outs, aux = ld.shared_step(batch)
        ld.manual_backward(outs, grads, aux)
        kinds.append(("reuse" if ld.iter_flags["reuse_init_conds"] else "fresh") + ("" if ld.iter_flags["is_teachable"] else "-unteachable"))
        for p in params:
            p.grad = None
        return loss

    _random.seed(0)
    for _ in range(2):                   # one fresh + one reuse iteration untimed
        cstep()
    torch.cuda.synchronize()
    del kinds[:]
    t0 = time.perf_counter()
    for _ in range(micro_batches):
        loss = cstep()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": "config 4: Stage-2 compositional distillation micro-batches, bs=3 (one subject x 4 prompt types; fresh "
                        "iterations: no-grad teacher filter of 2 candidates under CFG at UNet batch 8 with the 154-token split K/V "
                        "context + VAE decode of 4 images + scripted CLIP score, then the with-grad pass at UNet batch 4 + stage-2 "
                        "losses on 12 layers' captured outfeat/attnscore/q + backward into the embedding manager; reuse iterations "
                        "start from the cached prediction); text encoder / embedding manager / CLIP are stand-ins",
            "micro_batches": micro_batches, "iteration_kinds": kinds, "ms_per_micro_batch": round(1e3 * dt / micro_batches, 2),
            "images_per_sec": round(B * micro_batches / dt, 2), "final_loss": round(float(loss), 6)}


def zs_frontend_leg(device, gen, iters=10):
    """ddpm.py:2411-2431 for one bs=4 batch: the CLIP ViT-L/14 image encoder (openai/clip-vit-large-patch14 topology, random
    init) twice under no_grad -- foreground mask, background mask -- up to the penultimate hidden states, [4,257,1024] each.
    Algorithmic FLOPs per image and pass: 23 layers x (8 H^2 N + 4 N^2 H + 4 H I N) + the patch embedding."""
    from adaprompt_amd import synth
    from adaprompt_amd.clip_vision import CLIPVisionModelWithMask
    cfg = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224, patch_size=14,
               hidden_act="quick_gelu")
    with torch.device(device):
        enc = CLIPVisionModelWithMask(**cfg)
    enc.load_hf_state_dict(device_state_dict(synth.clip_vision_param_shapes(**cfg), "", device, 31))
    B, N, H, I = 4, 257, 1024, 4096
    pv = torch.randn(B, 3, 224, 224, device=device, generator=gen).half()
    mask = (torch.rand(B, 224, 224, device=device, generator=gen) > 0.5).half()
    n = enc.stop_before_last_layer

    def once():          # as conditioning.encode_zero_shot_image_features drives it: both passes as one batch of 8
        h = enc(torch.cat([pv, pv]), attn_mask=torch.cat([mask, 1 - mask]), layers_needed=n).hidden_states[-1]
        return h.chunk(2)

    for _ in range(3):
        a, b = once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        a, b = once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    flop = 2 * B * (n * (8 * H * H * N + 4 * N * N * H + 4 * H * I * N) + 2 * 256 * 588 * H)
    return {"workload": "zero-shot front end: CLIP ViT-L/14 image encoder (24 layers, width 1024, 16 heads, 257 tokens), bs=4, the "
                        "reference's two passes with its additive foreground-pair attention bias (mask, 1 - mask) run as one batch "
                        "of 8, stopped before the last layer; preprocessing and ArcFace stay third-party",
            "ms_per_batch": round(1e3 * dt, 3), "images_per_sec": round(B / dt, 1), "algorithmic_tflop": round(flop / 1e12, 3),
            "achieved_tflops": round(flop / dt / 1e12, 1), "finite": bool(torch.isfinite(a).all() and torch.isfinite(b).all())}


def spawn_ranks(n):
    """Start ``n`` copies of this script as child processes, rank r on GPU r (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    their env, which is what torchrun would have set), forward rank 0's stdout, and return a non-zero exit code if any
    rank fails.  The parent touches no GPU: children are started with ``subprocess`` (never ``exec``) before any HIP
    call.  With fewer than ``n`` GPUs on the box (a rehearsal) the ranks share devices over gloo -- RCCL refuses two
    ranks on one device -- and the line says so (``config.dist_backend``)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ndev = torch.cuda.device_count()            # counts devices without initialising the runtime on this image
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if ndev < n:
            env.setdefault("ADAP_DIST_BACKEND", "gloo")
            # two processes on one device: the single-launch GroupNorm needs ALL of its workgroups resident at once, which two
            # such grids from two processes cannot both have -- the rehearsal runs the two-pass kernels
            env.setdefault("ADAP_GN_TWO_PASS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    if ndev < n:
        print(f"warning: --gpus {n} on a box with {ndev} GPU(s): ranks share devices, gradient exchange over gloo",
              file=sys.stderr)
    rc = 0
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
        if bad:                                 # a rank died: the others would wait in a collective for ever
            rc = bad[0]
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
    for p in procs:
        rc = rc or p.wait()
    return rc


def in_kernel_clock(device):
    """clock the chip holds under the dominant kernel's load (MI355X_MICROARCH.md 'DVFS give-back', item 6): ~2 s of
    back-to-back launches of its best-fed shape (512 -> 512 @128^2, bs 4) on random data, then one launch stamped with
    s_memtime / s_memrealtime around every workgroup's K loop (adap_conv2d_set_clock_probe)."""
    from adaprompt_amd import _lib, ops
    B, C, H = 4, 512, 128
    x = torch.randn(B, H, H, C, device=device).to(torch.bfloat16)
    pk = ops.PackedConv(torch.randn(C, C, 3, 3, device=device) * 0.02, torch.zeros(C, device=device))
    flops = 2.0 * B * H * H * C * C * 9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(8000):                               # ~2 s at ~0.25 ms per launch
        ops.conv2d(x, pk.fwd, C, 3, 1, 1, bias=pk.bias)
    nwg = B * (H * H // 256) * (C // 128)
    buf = torch.zeros(2 * nwg, device=device, dtype=torch.int64)
    _lib.call("adap_conv2d_set_clock_probe", buf.data_ptr())
    e0.record()
    ops.conv2d(x, pk.fwd, C, 3, 1, 1, bias=pk.bias)
    e1.record()
    _lib.call("adap_conv2d_set_clock_probe", 0)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    v = buf.view(-1, 2).cpu().double()
    v = v[v[:, 1] > 0]
    ghz = float((v[:, 0] / v[:, 1] * 0.1).median())
    peak = BF16_MFMA_PEAK_TFLOPS * ghz / 2.4
    tf = flops / us / 1e6
    return {"shape": "conv3x3 512->512 @128x128 bs4, conv3x3_halo_kernel<128, true, true>", "ghz": round(ghz, 3),
            "nominal_ghz": 2.4, "launch_us": round(us, 1), "achieved": round(tf, 1), "peak_at_clock": round(peak, 1),
            "unit": "TFLOP/s", "frac_at_clock": round(tf / peak, 4),
            "note": "d(s_memtime)/d(s_memrealtime) x 100 MHz, median over the launch's workgroups after ~2 s of sustained load"}


def pmc_traffic(kernel):
    """HBM bytes per launch of ``kernel`` from the committed PMC summary (profiles/rNN_pmc_traffic.json, written by
    tools/pmc_summary.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same
    command; counters cannot be collected from inside the timed run).  None when no summary holds the kernel."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        e = d.get("kernels", {}).get(kernel)
        if e:
            return e["traffic_bytes_per_launch"], os.path.relpath(path, here)
    return None, None


if __name__ == "__main__":
    main()
