"""The thin training driver this package ships in place of the reference's Lightning launcher (main.py:1149-1206
``Trainer.from_argparse_args(strategy="ddp") ... trainer.fit(model, data)``; SURVEY.md section 2 marks the launcher itself out
of scope): it gives ``LatentDiffusion`` exactly the trainer-side objects its ``training_step(batch, batch_idx)`` /
``configure_optimizers()`` / ``on_save_checkpoint(checkpoint)`` hooks read from Lightning -- ``trainer.max_steps``, the
optimiser and LR scheduler, ``trainer.checkpoint_callback.dirpath`` -- plus the data-parallel gradient exchange
(``parallel.GradReducer`` on the optimiser's flat buffer, one process per GPU over RCCL).  No callbacks, loggers or
dataloader management."""
import os
from types import SimpleNamespace


class Trainer:
    def __init__(self, max_steps, ckpt_dir=None, every_n_train_steps=500, process_group=None, micro_batch_lanes=None,
                 prefetch_windows=0):
        """``max_steps`` / ``every_n_train_steps``: yaml:178-180, 190 (``lightning.trainer.max_steps``,
        ``modelcheckpoint.params.every_n_train_steps``), counted in optimiser steps (Lightning's ``global_step``).
        ``micro_batch_lanes``: issue the micro-batches of an accumulation window on separate HIP streams
        (``LatentDiffusion.training_window`` / ``MicroBatchLanes``); default: on a GPU with a frozen UNet, unless
        ``ADAP_MB_LANES=0``.  ``training_step(batch, batch_idx)`` -- the call Lightning makes per micro-batch -- then buffers a
        window's micro-batches and runs the window when its last one arrives.  ``prefetch_windows``: keep that many windows of
        batches buffered ahead and encode their latents on the prefetch stream under the running window (0: each micro-batch
        encodes its own latent on its lane, RNG order of the sequential loop)."""
        self.max_steps = int(max_steps)
        self.micro_batch_lanes = (os.environ.get("ADAP_MB_LANES", "1") != "0") if micro_batch_lanes is None else micro_batch_lanes
        self.prefetch_windows = int(prefetch_windows)
        self.lanes = None
        self.checkpoint_callback = SimpleNamespace(dirpath=ckpt_dir)
        self.every_n_train_steps = every_n_train_steps
        self.process_group = process_group
        self.optimizer = self.scheduler = self.reducer = None
        self.logged = []

    def attach(self, model):
        """``configure_optimizers()`` (Lightning's return shape) -> optimiser, scheduler, reducer on the flat gradient buffer;
        the micro-batch lanes where they apply (an accumulation window of >= 2 micro-batches, parameters on the GPU, the UNet
        frozen -- its own weight gradients are written through raw pointers during the whole backward: one stream)."""
        from .parallel import GradReducer
        object.__setattr__(model, "trainer", self)
        conf = model.configure_optimizers()[0]
        self.optimizer = conf["optimizer"]
        self.scheduler = conf["lr_scheduler"]["scheduler"]
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        self.reducer = GradReducer(params, process_group=self.process_group, flat=getattr(self.optimizer, "grad_buffer", None))
        self._make_lanes(model)
        return self

    def _make_lanes(self, model):
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        n = int(model.manual_accumulate_grad_batches)
        unet = getattr(model, "model", None)
        frozen = unet is None or not any(p.requires_grad for p in unet.parameters())
        if self.micro_batch_lanes and self.lanes is None and n > 1 and params and params[0].is_cuda and frozen:
            from .ldm.models.diffusion.ddpm import MicroBatchLanes
            self.lanes = MicroBatchLanes(params, n=n, reducer=self.reducer if self.reducer.world > 1 else None)

    def detach(self):
        """take the lanes' gates off the parameters (and restore autograd's stream-mismatch warning)."""
        if self.lanes is not None:
            self.lanes.remove()
            self.lanes = None

    def fit(self, model, batches):
        """one epoch over ``batches`` (an iterable of batch dicts), stopping at ``max_steps`` optimiser steps: the loop
        Lightning runs -- ``training_step(batch, batch_idx)`` per micro-batch (which, on lanes, defers a window's micro-batches
        until its last one is there), checkpoints every ``every_n_train_steps`` optimiser steps, and at the end what the model
        still holds buffered (a partial window runs on one stream and stays open, as in the reference's loop)."""
        if self.optimizer is None:
            self.attach(model)
        self._make_lanes(model)
        last_saved = -1
        try:
            for batch_idx, batch in enumerate(batches):
                if model.global_step >= self.max_steps:
                    break
                hook = getattr(model, "on_train_batch_start", None)
                if hook is not None:
                    hook(batch, batch_idx)
                loss, aux = model.training_step(batch, batch_idx)
                self._log(loss, aux)
                gs = model.global_step
                if self.every_n_train_steps and gs > 0 and gs % self.every_n_train_steps == 0 and gs != last_saved:
                    self.save_checkpoint(model)
                    last_saved = gs
            flush = getattr(model, "flush_window", None)
            if flush is not None:
                for loss, aux in flush(run=model.global_step < self.max_steps):
                    self._log(loss, aux)
                gs = model.global_step
                if self.every_n_train_steps and gs > 0 and gs % self.every_n_train_steps == 0 and gs != last_saved:
                    self.save_checkpoint(model)
            if self.reducer is not None:
                self.reducer.wait()
        finally:
            self.detach()
        return self.logged

    def _log(self, loss, aux):
        if isinstance(aux, dict) and aux.get("window") is not None:
            self.logged.extend(l for l, _a in aux["window"])
        elif loss is not None:
            self.logged.append(loss)

    def save_checkpoint(self, model):
        from . import ops
        if ops.gn_sync_poisoned():          # (one device -> host read per checkpoint; never set in any run so far)
            raise RuntimeError("a single-launch GroupNorm's in-launch exchange timed out since the last checkpoint: the "
                               "parameters may be wrong; rerun with ADAP_GN_TWO_PASS=1")
        ckpt = {"state_dict": {}, "global_step": model.global_step}
        d = self.checkpoint_callback.dirpath
        if d is not None:
            os.makedirs(d, exist_ok=True)
        model.on_save_checkpoint(ckpt)
        return ckpt
