"""The accumulation window's scheduling on the MI355X (``LatentDiffusion.training_window`` / ``MicroBatchLanes`` / the
per-batch ``training_step(batch, batch_idx)`` entry): everything here compares two ORDERS OF ISSUE of the same arithmetic, so
with the kernels held to their bit-reproducible forms (two-pass GroupNorm on every stream, lone-stream split-K plans) the
results must be equal bit for bit -- losses, Prodigy's d, every parameter.  Reference semantics: ddpm.py:515-638 (iteration
draw per micro-batch, manual accumulation over 2 micro-batches, one optimiser step per window)."""
import math
import os
import random

import numpy as np
import pytest
import torch

from adaprompt_amd import ops, synth
from conftest import ellipse_mask, border_mask

pytestmark = pytest.mark.gpu

NARROW = dict(synth.SD15_UNET, model_channels=64, context_dim=128)


def dev():
    return torch.device("cuda:0")


def build(B, teacher=False, seed=3, max_steps=8):
    """narrow UNet + stand-in hook (conditioning through ``cond_fn``, regularisers on) + Prodigy on its flat buffer."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion, Arc2FaceWrapper
    from adaprompt_amd.ldm.prodigy import Prodigy
    from adaprompt_amd.ldm.util import prodigy_linear_schedule
    from adaprompt_amd.parallel import GradReducer
    from adaprompt_amd.hook_standin import SyntheticSubjBasisGenerator, make_cond_fn
    ucfg = dict(NARROW)
    vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=64)
    torch.manual_seed(seed)
    hook = SyntheticSubjBasisGenerator(n_params=3 * 16 * 77 * 128, tokens=77, dim=128, id_dim=32)
    with torch.no_grad():
        hook.bases.mul_(20.0)
    hook = hook.to(dev())
    ld = LatentDiffusion.hot_path({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}},
                                  {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg},
                                  cond_fn=make_cond_fn(hook, capture=True, regs=True))
    ld.load_state_dict(synth.synthetic_unet_state_dict(ucfg), strict=False)
    ld = ld.to(dev())
    ld.freeze_unet()
    # stage 1 as this model can run it: the compositional iteration (every 3rd global step in the yaml) needs the reference's
    # conditioning side and is refused by the preamble without it
    ld.composition_regs_iter_gap = 0
    if teacher:
        TP = "arc2face.unet."
        tsd = synth.synthetic_unet_state_dict(ucfg, prefix=TP)
        tw = Arc2FaceWrapper(unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg})
        tw.unet.load_state_dict({k[len(TP):]: v for k, v in tsd.items()}, strict=True)
        ld.set_arc2face_teacher(tw.to(dev()).eval())
    params = list(hook.parameters())
    opt = Prodigy(params, lr=1.0, betas=(0.9, 0.999), d_coef=2.0, use_bias_correction=True, weight_decay=0.0)
    red = GradReducer(params, flat=opt.grad_buffer)
    sched = prodigy_linear_schedule(opt, max_steps=max_steps, warm_up_steps=1, scheduler_cycles=1)
    return ld, hook, params, opt, red, sched


def image_batches(n, B, with_teacher_ctx=False):
    """batches as the dataloader hands them over: 512 x 512 HWC images in [-1, 1] (the first stage encodes them to 64 x 64
    latents), image-resolution masks, id embeddings."""
    fg, im = ellipse_mask(B, 512, 512), border_mask(B, 512, 512, 40)
    out = []
    for mb in range(n):
        b = {"image": synth.synthetic_input(f"win.img.{mb}", (B, 512, 512, 3)).tanh().to(dev()),
             "zs_id_embs": synth.synthetic_input(f"win.ids.{mb}", (B, 32)).to(dev()),
             "fg_mask": fg[:, 0].to(dev()), "aug_mask": im[:, 0].to(dev())}
        if with_teacher_ctx:
            b["arc2face_prompt_emb"] = synth.synthetic_input(f"win.tctx.{mb}", (B, 21, NARROW["context_dim"])).to(dev())
        out.append(b)
    return out


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


class reproducible_kernels:
    """two-pass GroupNorm on every stream + the lone-stream split-K plans under lanes: every kernel bit-reproducible."""

    def __enter__(self):
        ops.gn_two_pass(True)
        os.environ["ADAP_LANES_KSPLIT_SCALE"] = "100"

    def __exit__(self, *a):
        ops.gn_two_pass(False)
        os.environ.pop("ADAP_LANES_KSPLIT_SCALE", None)


def final_state(ld, opt, params, losses):
    torch.cuda.synchronize()
    st = opt.device_state()
    return ([None if x is None else float(x) for x in losses], st["d"], st["k"], [p.detach().cpu().clone() for p in params])


def assert_same(a, b):
    la, da, ka, pa = a
    lb, db, kb, pb = b
    assert la == lb and da == db and ka == kb, (la, lb, da, db, ka, kb)
    assert all(math.isfinite(v) for v in la if v is not None)
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)


def test_window_with_distillation_micro_batches_equals_sequential_steps():
    """a window whose micro-batches are DRAWN (``auto_iteration``: the reference's preamble, ddpm.py:516-572,
    1839-1859) -- plain recon iterations and Arc2Face-distillation iterations with ND in {1, 3}, the latter trimmed to HALF_BS
    instances (ddpm.py:1857) BEFORE their conditioning is computed -- on two lanes against ``training_step`` on the same
    batches one after the other.  Same host RNG (np.random / random) and device RNG consumption, micro-batch by micro-batch:
    the conditioning is a request the window serves on lane 0 at the point where the sequential step computes it."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import MicroBatchLanes
    B, n_mb = 4, 8
    auto = {"max_steps": 8, "arc2face_distill_iter_prob": 0.5, "max_num_denoising_steps": 3}
    kinds = {}

    def run(mode):
        ld, hook, params, opt, red, sched = build(B, teacher=True)
        batches = image_batches(n_mb, B, with_teacher_ctx=True)
        seen = []
        orig = ld._shared_step_gen

        def spy(batch, **kw):
            seen.append((bool(kw.get("use_arc2face_as_target")), int(kw.get("num_denoising_steps", 1)), batch["zs_id_embs"].shape[0]))
            return orig(batch, **kw)
        ld._shared_step_gen = spy
        seed_all(10)          # draws (recon, distill x3), (recon, distill x1), (distill x3, distill x1), (recon, recon)
        losses = []
        with reproducible_kernels():
            if mode == "steps":
                for b in batches:
                    losses.append(ld.training_step(b, optimizer=opt, reducer=red, scheduler=sched, auto_iteration=dict(auto))[0])
            else:
                lanes = MicroBatchLanes(params, n=2)
                for w in range(n_mb // 2):
                    out = ld.training_window(batches[2 * w:2 * w + 2], opt, red, sched, lanes, auto_iteration=dict(auto))
                    losses += [o[0] for o in out]
                lanes.remove()
        kinds[mode] = seen
        assert ld.batch_idx == n_mb
        return final_state(ld, opt, params, losses)

    a = run("steps")
    b = run("window")
    assert kinds["steps"] == kinds["window"]
    ks = kinds["window"]
    # the draw must have exercised what the test is for: a multi-step distillation micro-batch as the SECOND of a window (its
    # conditioning is made on lane 0 for the trimmed batch) and as the first, beside plain recon ones
    assert any(d and nd > 1 for i, (d, nd, _) in enumerate(ks) if i % 2 == 1), ks
    assert any(d and nd > 1 for i, (d, nd, _) in enumerate(ks) if i % 2 == 0), ks
    assert any(not d for d, _, _ in ks), ks
    assert_same(a, b)


def test_training_step_per_batch_equals_training_window():
    """``training_step(batch, batch_idx)`` as Lightning calls it -- once per micro-batch, trainer attached --
    buffers a window's micro-batches and runs the window on the trainer's lanes when its last one arrives; against
    ``training_window`` called with the same batches: bit-equal.  The deferred call returns (None, {'deferred': True}), the
    completing call the window's losses; an odd tail is run by ``flush_window`` on one stream and leaves the window open."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import MicroBatchLanes
    from adaprompt_amd.trainer import Trainer
    B, n_mb = 2, 5

    def attach(ld, opt, red, sched, params, lanes=True):
        tr = Trainer(max_steps=8, every_n_train_steps=0, micro_batch_lanes=lanes)
        tr.optimizer, tr.scheduler, tr.reducer = opt, sched, red
        object.__setattr__(ld, "trainer", tr)
        tr._make_lanes(ld)
        return tr

    def run(mode):
        ld, hook, params, opt, red, sched = build(B)
        batches = image_batches(n_mb, B)
        seed_all(5)
        losses = []
        auto = {"max_steps": 8, "composition_regs_iter_gap": ld.composition_regs_iter_gap,
                "arc2face_distill_iter_prob": ld.arc2face_distill_iter_prob, "mix_prompt_distill_weight": ld.mix_prompt_distill_weight,
                "max_num_denoising_steps": ld.max_num_denoising_steps}
        with reproducible_kernels():
            if mode == "entry":
                tr = attach(ld, opt, red, sched, params)
                assert tr.lanes is not None
                for i, b in enumerate(batches):
                    loss, aux = ld.training_step(b, i)
                    if i % 2 == 0:
                        assert loss is None and aux == {"deferred": True}
                    else:
                        assert len(aux["window"]) == 2 and aux["window"][1][0] is loss
                        losses += [l for l, _ in aux["window"]]
                tail = ld.flush_window()
                assert len(tail) == 1
                losses.append(tail[0][0])
                tr.detach()
            elif mode == "window":
                lanes = MicroBatchLanes(params, n=2)
                for w in range(n_mb // 2):
                    losses += [o[0] for o in ld.training_window(batches[2 * w:2 * w + 2], opt, red, sched, lanes, auto_iteration=dict(auto))]
                losses.append(ld.training_step(batches[-1], optimizer=opt, reducer=red, scheduler=sched, auto_iteration=dict(auto))[0])
                lanes.remove()
            else:                                   # the one-stream loop: the same entry without lanes
                tr = attach(ld, opt, red, sched, params, lanes=False)
                assert tr.lanes is None
                for i, b in enumerate(batches):
                    losses.append(ld.training_step(b, i)[0])
        assert ld.batch_idx == n_mb and opt.device_state()["k"] == 2
        # the open window's gradient (micro-batch 5's, not yet stepped) belongs to the state too
        return final_state(ld, opt, params + [opt.grad_buffer], losses)

    a = run("entry")
    assert_same(a, run("window"))
    assert_same(a, run("steps"))


def test_trainer_fit_runs_windows_through_the_per_batch_entry():
    """``Trainer.fit``: Lightning's loop over ``training_step(batch, batch_idx)`` with lanes and one window of latents
    prefetched ahead (``prefetch_windows=1``: the VAE encodes of window w+1 go to the prefetch stream behind window w's
    backwards): max_steps honoured, every micro-batch logged once and in order, the gates removed afterwards."""
    from adaprompt_amd.trainer import Trainer
    B = 2
    ld, hook, params, opt, red, sched = build(B)
    seed_all(7)
    tr = Trainer(max_steps=3, every_n_train_steps=0, micro_batch_lanes=True, prefetch_windows=1)
    tr.optimizer, tr.scheduler, tr.reducer = opt, sched, red
    object.__setattr__(ld, "trainer", tr)
    logged = tr.fit(ld, image_batches(9, B))
    torch.cuda.synchronize()
    assert ld.global_step == 3 and ld.batch_idx == 6 and opt.device_state()["k"] == 3
    assert len(logged) == 6 and all(math.isfinite(float(x)) for x in logged)
    assert tr.lanes is None and not ld._win_entry["buf"] and not ld._win_entry["pf"]._queue
    assert all(not p._backward_hooks for p in params)
    assert not ops.gn_sync_poisoned()


def test_cached_device_data_is_complete_before_another_lane_can_read_it():
    """Weight packs (and the other cached device tensors) are built by kernels on whatever stream first needs them and read
    afterwards from every stream.  While micro-batch lanes exist a fill drains its stream before the object is handed out --
    found as a NaN loss in two processes sharing one card (lane 1's first convolutions read packs lane 0's pack kernels had not
    written yet).  Outside that mode nothing is drained (``unfreeze_model`` rebuilds 686 packs per step on one stream)."""
    from adaprompt_amd import _lib, ops
    dev = torch.device("cuda:0")
    w = torch.randn(320, 320, 3, 3, device=dev)
    sink = torch.zeros(1, device=dev)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    fills = ops.CACHE_FILLS
    ops.multi_stream(+1)
    try:
        with torch.cuda.stream(side):
            _lib.call("adap_debug_occupy", 64, 512, 20000, sink.data_ptr(), _lib.current_stream())     # 20 ms ahead of the pack kernel
            pk = ops.PackedConv(w)
            assert side.query(), "the pack kernel is still queued behind the resident kernel: another stream could read garbage"
            _lib.call("adap_debug_occupy", 64, 512, 20000, sink.data_ptr(), _lib.current_stream())
            _ = pk.bwd
            assert side.query()
    finally:
        ops.multi_stream(-1)
    assert ops.CACHE_FILLS == fills + 2
    with torch.cuda.stream(side):
        _lib.call("adap_debug_occupy", 64, 512, 20000, sink.data_ptr(), _lib.current_stream())
        pk2 = ops.PackedConv(w)
        assert not side.query(), "a cache fill must not drain its stream outside the multi-stream mode"
    torch.cuda.synchronize()
    assert torch.equal(pk.fwd, pk2.fwd)
    assert ops._MULTI_STREAM == 0           # (training_window announces the mode for the duration of a window on lanes only)
