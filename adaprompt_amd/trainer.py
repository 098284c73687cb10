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
    def __init__(self, max_steps, ckpt_dir=None, every_n_train_steps=500, process_group=None, micro_batch_lanes=None):
        """``max_steps`` / ``every_n_train_steps``: yaml:178-180, 190 (``lightning.trainer.max_steps``,
        ``modelcheckpoint.params.every_n_train_steps``), counted in optimiser steps (Lightning's ``global_step``).
        ``micro_batch_lanes``: issue the micro-batches of an accumulation window on separate HIP streams
        (``LatentDiffusion.training_window`` / ``MicroBatchLanes``); default: on a GPU, unless ``ADAP_MB_LANES=0``."""
        self.max_steps = int(max_steps)
        self.micro_batch_lanes = (os.environ.get("ADAP_MB_LANES", "1") != "0") if micro_batch_lanes is None else micro_batch_lanes
        self.lanes = None
        self.checkpoint_callback = SimpleNamespace(dirpath=ckpt_dir)
        self.every_n_train_steps = every_n_train_steps
        self.process_group = process_group
        self.optimizer = self.scheduler = self.reducer = None
        self.logged = []

    def attach(self, model):
        """``configure_optimizers()`` (Lightning's return shape) -> optimiser, scheduler, reducer on the flat gradient buffer."""
        from .parallel import GradReducer
        object.__setattr__(model, "trainer", self)
        conf = model.configure_optimizers()[0]
        self.optimizer = conf["optimizer"]
        self.scheduler = conf["lr_scheduler"]["scheduler"]
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        self.reducer = GradReducer(params, process_group=self.process_group, flat=getattr(self.optimizer, "grad_buffer", None))
        return self

    def fit(self, model, batches):
        """one epoch over ``batches`` (an iterable of batch dicts), stopping at ``max_steps`` optimiser steps."""
        if self.optimizer is None:
            self.attach(model)
        last_saved = -1
        n = int(model.manual_accumulate_grad_batches)
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        windowed = (self.micro_batch_lanes and n > 1 and params and params[0].is_cuda and model.batch_idx % n == 0)
        if windowed and self.lanes is None:
            from .ldm.models.diffusion.ddpm import MicroBatchLanes
            self.lanes = MicroBatchLanes(params, n=n, reducer=self.reducer)
        auto = {"max_steps": self.max_steps, "composition_regs_iter_gap": model.composition_regs_iter_gap,
                "arc2face_distill_iter_prob": model.arc2face_distill_iter_prob,
                "mix_prompt_distill_weight": model.mix_prompt_distill_weight,
                "max_num_denoising_steps": model.max_num_denoising_steps}
        it = iter(enumerate(batches))
        while model.global_step < self.max_steps:
            # a whole window's batches are taken BEFORE any of its work is issued: whatever produced them on the current
            # stream is then ahead of the window's first kernel, and the side lanes need not wait for lane 0's micro-batch
            window = []
            for batch_idx, batch in it:
                window.append((batch_idx, batch))
                if len(window) == (n if windowed else 1):
                    break
            if not window:
                break
            if windowed and len(window) == n:
                self.lanes.window_start()
                out = model.training_window([b for _, b in window], self.optimizer, self.reducer, self.scheduler, self.lanes,
                                            auto_iteration=auto)
                self.logged.extend(loss for loss, _aux in out)
            else:
                for batch_idx, batch in window:
                    loss, _aux = model.training_step(batch, batch_idx)
                    self.logged.append(loss)
            gs = model.global_step
            if self.every_n_train_steps and gs > 0 and gs % self.every_n_train_steps == 0 and gs != last_saved:
                self.save_checkpoint(model)
                last_saved = gs
        if self.reducer is not None:
            self.reducer.wait()
        return self.logged

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
