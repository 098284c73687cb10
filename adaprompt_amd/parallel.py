"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference trains with Lightning ``strategy="ddp"`` (main.py:829): torch DDP averages every
trainable gradient over the ranks after EVERY micro-batch backward (``manual_backward``
ddpm.py:595; there is no ``no_sync``), then clips and steps every 2nd micro-batch on the summed
gradients (ddpm.py:606-633).  Only the subject-basis generators (+ CLIP text encoder inside them)
are trainable in the shipped config, ~149 M fp32 values = 0.6 GB per exchange.

MI355X design: all trainable gradients live in ONE flat fp32 buffer (``p.grad`` are views of it),
so the exchange is a few large collectives instead of DDP's many 25 MB buckets -- xGMI is
point-to-point (7 links x ~153 GB/s), large messages keep every link busy.  The collective is
issued asynchronously right after backward and only awaited immediately before the next backward
writes into the buffer (or before the optimizer step), so it overlaps the next micro-batch's
no-grad VAE encode and UNet forward.  Averaging an already-averaged accumulation is exact:
mean_r(mean(g1) + g2_r) = mean(g1) + mean(g2), the same sum DDP accumulates into ``.grad``."""
import os

import torch
import torch.distributed as dist

_HOLD_GN = os.environ.get("ADAP_GN_HOLD_DURING_EXCHANGE", "0") == "1"


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    -> (rank, world_size, local_rank).  Single-process runs need no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("ADAP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():           # "nccl" IS RCCL on ROCm
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class CAbiComm:
    """An RCCL communicator owned by libadaprompt_hip.so (``adap_comm_*`` / ``adap_allreduce_bucket``, include/adaprompt_hip.h)."""

    def __init__(self, unique_id, nranks, rank):
        import ctypes
        from . import _lib
        self._lib, self.nranks, self.rank = _lib, nranks, rank
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), len(unique_id))
        _lib.call("adap_comm_init", ctypes.byref(h), ctypes.addressof(buf), nranks, rank)
        self.handle = h

    @staticmethod
    def new_unique_id():
        import ctypes
        from . import _lib
        n = _lib.call_long("adap_comm_unique_id_bytes")
        buf = ctypes.create_string_buffer(n)
        _lib.call("adap_comm_unique_id", ctypes.addressof(buf))
        return buf.raw

    @classmethod
    def from_process_group(cls, group=None):
        """rank 0 draws the id, ``torch.distributed`` (any backend) carries its 128 bytes to the others, once."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.new_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], world, rank)

    def allreduce_(self, t, average=True):
        """in place on the current stream; f32 or bf16, contiguous."""
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
        self._lib.call("adap_allreduce_bucket", self.handle, t.data_ptr(), t.numel(), 0 if t.dtype == torch.float32 else 1,
                       1 if average else 0, self._lib.current_stream())
        return t

    def destroy(self):
        if self.handle:
            self._lib.call("adap_comm_destroy", self.handle)
            self.handle = None


class GradReducer:
    """Flat-buffer gradient all-reduce (mean over ranks), asynchronous.

    reduce()  -- issue the collective(s) for the current contents of the gradient buffer
    wait()    -- make the current stream wait for them (call before the next backward / the step)
    """

    def __init__(self, params, process_group=None, bucket_bytes=256 << 20, flat=None, backend=None):
        """``flat``: an existing flat gradient buffer that every ``p.grad`` already views (e.g.
        ``adaprompt_amd.ldm.prodigy.Prodigy.grad_buffer``) -- the exchange then runs on the optimiser's own buffer.
        ``backend="c_abi"`` (or ``ADAP_REDUCER_BACKEND=c_abi``): the collectives go through the library's own
        ``adap_allreduce_bucket`` on an RCCL communicator it creates (``CAbiComm``) instead of ``torch.distributed``'s; the
        process group is then only used once, to hand the communicator's unique id to the other ranks."""
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._cabi = None
        if (backend or os.environ.get("ADAP_REDUCER_BACKEND")) == "c_abi" and self.world > 1:
            self._cabi = CAbiComm.from_process_group(process_group)
            self._side = torch.cuda.Stream()           # the exchange's own stream (RCCL kernels run under the next forward)
            self._done = None
        dev = self.params[0].device if self.params else torch.device("cpu")
        if flat is not None:
            lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
            for p in self.params:
                assert p.grad is not None and lo <= p.grad.data_ptr() and p.grad.data_ptr() + p.numel() * 4 <= hi, \
                    "flat= must hold every p.grad entirely"
            self.flat = flat
            n = flat.numel()
        else:
            n = sum(p.numel() for p in self.params)
            self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
            off = 0
            for p in self.params:
                assert p.dtype == torch.float32, "trainable parameters are fp32 in the reference (Trainer precision 32)"
                p.grad = self.flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        per = max(1, bucket_bytes // 4)
        self.chunks = [self.flat[i:i + per] for i in range(0, n, per)]
        self._works = []
        self.bytes_per_reduce = n * 4
        # RCCL averages inside the collective (ReduceOp.AVG: the 1/world factor rides on the reduction, no extra pass over
        # the 0.6-4.5 GB buffer); gloo has no AVG, so there the sum is scaled once it has landed (wait()).
        backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self._avg_in_collective = backend == "nccl"
        self._scale_pending = False
        self._gn_held = False
        # bucketed mode (begin_backward): which chunks a parameter's gradient lies in, and how many parameters a chunk waits for
        self._chunk_elems = per
        self._chunks_of, self._need = {}, [0] * len(self.chunks)
        base = self.flat.data_ptr()
        for p_ in self.params:
            lo = (p_.grad.data_ptr() - base) // 4
            cs = list(range(lo // per, (lo + p_.numel() - 1) // per + 1))
            self._chunks_of[id(p_)] = cs
            for c in cs:
                self._need[c] += 1
        self._left = self._issued = self._ready = None
        self._expect = {}
        self._hooks = []

    # ---- one chunk's collective ----------------------------------------------------------------------------------------
    def _issue(self, ci):
        c = self.chunks[ci]
        if os.environ.get("ADAP_DIAG_NO_EXCHANGE") == "1":           # DIAGNOSTIC (tools/lanes_two_process_soak.sh): the ranks stop
            return                                                   # talking -- two processes share a card at full speed
        if self._cabi is not None:
            # the chunk's gradients were written on the current (compute) stream: the exchange stream waits for exactly that
            ev = torch.cuda.Event()
            ev.record()
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                self._cabi.allreduce_(c, average=True)
                self._done = torch.cuda.Event()
                self._done.record()
            return
        op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
        self._works.append(dist.all_reduce(c, op=op, group=self.group, async_op=True))
        self._scale_pending = not self._avg_in_collective

    def _hold_groupnorm(self):
        # Optional (ADAP_GN_HOLD_DURING_EXCHANGE=1): while collective kernels are resident the GroupNorms take the two-launch
        # form (ops.gn_two_pass).  Off by default: beside a resident kernel of a collective's footprint (48-128 workgroups x
        # 512 threads x ~100 registers for 4 ms) the single-launch kernel's workgroups still all find room and a call costs
        # 20 us instead of 17 (two launches: 31 us) -- tests/test_parallel_gpu.py::test_groupnorm_beside_a_resident_collective_kernel.
        # If a communicator ever did crowd them out, the GroupNorm would last as long as the collective (bounded wait; a
        # time-out turns its output into NaN and sets the poison word ops.gn_poison_poll reads), and this switch is the remedy.
        if not _HOLD_GN:
            return
        if not self._gn_held and self.flat.is_cuda:
            from . import ops
            ops.gn_two_pass(True)
            self._gn_held = True

    # ---- bucketed exchange: a chunk goes out as soon as the backward has finished every gradient in it -------------------
    def begin_backward(self):
        """Arm the bucketed exchange for the backward that follows (call after ``wait()``): with ``unfreeze_model`` the UNet's
        4.5 GB of gradients are complete block by block, output blocks first, long before the hook's -- each 256 MB chunk's
        all-reduce starts when the last gradient in it is final (``grad_ready``: autograd's post-accumulate hook for
        parameters autograd manages, ``functional.GRAD_DONE`` for the ones the block Functions write through raw pointers)
        and runs under the rest of the backward; ``reduce()`` afterwards sends what is left.  The numbers are those of the
        single-shot exchange: the same collectives on the same data, only earlier."""
        if self.world == 1:
            return
        assert not self.pending, "begin_backward() with an exchange still in flight: call wait() first"
        if not self._hooks:
            from . import functional
            for p_ in self.params:
                self._hooks.append(p_.register_post_accumulate_grad_hook(self.grad_ready))
            functional.GRAD_DONE = self.grad_ready
        self._left = list(self._need)
        self._issued = [False] * len(self.chunks)
        self._ready = set()
        # a block Function that ran k times in the graph (k UNet passes under one autograd.backward) reports its parameters
        # k times; only the last report makes them final (functional.FWD_PASSES counts the forwards that recorded a
        # backward node since the previous begin_backward; an over-count only delays a chunk until reduce())
        from . import functional
        self._expect = functional.take_forward_passes()

    def grad_ready(self, param):
        """``param.grad`` is final for this backward once every expected report has come (autograd's post-accumulate hook
        fires once per backward; a block Function reports once per invocation, see ``begin_backward``)."""
        pid = id(param)
        if self._left is None or pid in self._ready or pid not in self._chunks_of:
            return
        n = self._expect.get(pid, 1) - 1
        if n > 0:
            self._expect[pid] = n
            return
        self._ready.add(pid)
        for c in self._chunks_of[id(param)]:
            self._left[c] -= 1
            if self._left[c] == 0 and not self._issued[c]:
                self._issued[c] = True
                self._hold_groupnorm()
                self._issue(c)

    def reduce(self):
        if self.world == 1:
            return
        issued, self._left, self._issued, self._ready = self._issued, None, None, None
        if issued is None:
            self.wait()                       # single-shot mode: nothing of this backward is in flight yet
        self._hold_groupnorm()
        for ci in range(len(self.chunks)):
            if issued is None or not issued[ci]:
                self._issue(ci)

    def wait(self):
        """the current stream waits for the outstanding collectives.  MUST run before anything writes the gradient
        buffer again (the next ``manual_backward``) and before the optimiser reads it."""
        if self._cabi is not None and self._done is not None:
            torch.cuda.current_stream().wait_event(self._done)
            self._done = None
        for w in self._works:
            w.wait()
        self._works = []
        if self._scale_pending:
            self.flat.mul_(1.0 / self.world)
            self._scale_pending = False
        if self._gn_held:
            from . import ops
            ops.gn_two_pass(False)
            self._gn_held = False

    @property
    def pending(self):
        return bool(self._works) or (self._cabi is not None and self._done is not None)

    def zero(self):
        self.wait()
        self.flat.zero_()
