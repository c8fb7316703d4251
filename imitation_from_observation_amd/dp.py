"""Data-parallel training of the translator: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI) for the single exchange step of the path.

The reference has no multi-GPU code; this is new functionality defined by
SURVEY.md 8e: the global batch is split evenly, every rank holds a full replica + Adam state, and per
step   local fwd/bwd  ->  SUM all-reduce of the flat f32 gradient arena  ->  identical local Adam.
recon1/recon2 are sums over the batch (arm_shaping.py:1352-1353) so their gradients add across
shards; simloss is a mean over (global batch x featsize) (arm_shaping.py:1345), so each shard divides
its simloss gradient by the GLOBAL batch (the `sim_batch` argument of ctx_dev_forward_backward).

`DataParallelTrainer` only sequences an *engine* and the collective, so the same code runs on the
HIP engine (cuda tensors, RCCL) and -- in the CPU test-suite -- on a gloo group with a stand-in engine.

`RcclTrainer` is the same step with the collective INSIDE libctxtrans (ctx_dp_*: its own RCCL communicator, the
two-bucket schedule on a second stream): torch.distributed is then needed only to ship the 128-byte rendezvous blob,
and a host without torch can do that by any other means.
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.distributed as dist

from .translator import Translator


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class HipEngine:
    """The HIP translator on torch-owned device memory: torch allocates the [params|grads|m|v] arena
    and owns the stream, libctxtrans enqueues its kernels on that stream, so RCCL calls issued through
    torch.distributed are stream-ordered with them."""

    def __init__(self, H, W, df_dim, featsize, max_batch, device, seed, precision=None):
        self.dev = torch.device("cuda", device)
        n = Translator.arena_floats(H, W, df_dim, featsize)
        self.arena = torch.zeros(n, device=self.dev, dtype=torch.float32)
        self.stride = n // 4
        torch.cuda.synchronize(self.dev)      # the zero-fill ran on torch's default stream
        # a dedicated torch stream (the legacy default stream has handle 0 = "make your own" in the C ABI)
        self.stream = torch.cuda.Stream(self.dev)
        self.translator = Translator(H, W, df_dim, featsize, max_batch, device=device,
                                     stream=self.stream.cuda_stream, arena_ptr=self.arena.data_ptr(), precision=precision)
        self.translator.init_params(seed)
        self.n_params = self.translator.n_params
        self.params = self.arena[: self.stride]
        self.grads = self.arena[self.stride: 2 * self.stride]
        self._scal_view = None

    @staticmethod
    def _ptr(t):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("frames must be contiguous float32 cuda tensors")
        return t.data_ptr()

    def forward_backward(self, src, ctx, tgt, sim_batch, bucket_cb=None):
        """bucket_cb(first, count): called from inside the backward pass once grads[first:first+count] are final."""
        B = src.shape[0]
        self.translator.set_grad_bucket_callback(bucket_cb)
        try:
            self.translator.dev_forward_backward(self._ptr(src), self._ptr(ctx), self._ptr(tgt), B, sim_batch)
        finally:
            if bucket_cb is not None:
                self.translator.set_grad_bucket_callback(None)

    def adam(self, lr):
        self.translator.dev_adam(lr)

    def train_step(self, src, ctx, tgt, lr):
        """The whole single-replica step in one ABI call (ctx_dev_train_step): same result as forward_backward + adam, with
        Adam's slices enqueued beside the remaining backward."""
        self.translator.dev_train_step(self._ptr(src), self._ptr(ctx), self._ptr(tgt), src.shape[0], lr)

    def scalars_tensor(self):
        """{loss, simloss, recon1, recon2} of the last forward as a device tensor: a view of the f32[4] the loss kernel wrote
        (ctx_dev_scalar_buf) -- stream-ordered with the step, no host round trip."""
        if self._scal_view is None:
            # zero-copy wrap of the device pointer through the CUDA array interface
            holder = type("_DevArr", (), {"__cuda_array_interface__": {"shape": (4,), "typestr": "<f4", "version": 2,
                                                                         "data": (self.translator.scalars_ptr, False)}})()
            self._scal_view = torch.as_tensor(holder, device=self.dev)
            self._scal_holder = holder
        return self._scal_view


class DataParallelTrainer:
    def __init__(self, H=64, W=64, df_dim=64, featsize=1024, max_batch=256, device=0, seed=1234, engine=None, precision=None):
        self.engine = engine or HipEngine(H, W, df_dim, featsize, max_batch, device, seed, precision)
        self.world = _world()
        # CTX_DP_OVERLAP=1: bucketed all-reduce overlapped with the encoders' backward (opt-in until it has run on a
        # multi-GPU node; the default is one all-reduce after backward).  CTX_DP_FORCE=1 runs the collectives at world 1 too.
        self.overlap = os.environ.get("CTX_DP_OVERLAP", "0") == "1"
        self.force_collectives = os.environ.get("CTX_DP_FORCE", "0") == "1"
        self.n_params = self.engine.n_params
        if self.world > 1:
            # replicas start identical.  Issued on the engine's stream (the collective orders itself after the
            # current stream) and drained, so the first forward cannot race the incoming parameters.
            stream = getattr(self.engine, "stream", None)
            with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                dist.broadcast(self.engine.params, src=0)
            if stream is not None:
                stream.synchronize()
        self.translator = getattr(self.engine, "translator", None)

    def step(self, src, ctx, tgt, lr=1e-4):
        """One data-parallel train step on this rank's shard (all ranks pass equal batch sizes)."""
        B = src.shape[0]
        stream = getattr(self.engine, "stream", None)
        with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
            if self.overlap and (self.world > 1 or self.force_collectives):
                # two buckets: the tail of the arena (translate/*, deconv/*: final after the decoder's backward) is reduced
                # while the encoders' backward runs; the head after it.  Collectives order themselves after the current stream.
                works, split = [], [self.engine.grads.numel()]

                def tail_ready(first, count):
                    split[0] = first
                    works.append(dist.all_reduce(self.engine.grads[first:first + count], op=dist.ReduceOp.SUM, async_op=True))

                self.engine.forward_backward(src, ctx, tgt, sim_batch=B * self.world, bucket_cb=tail_ready)
                works.append(dist.all_reduce(self.engine.grads[:split[0]], op=dist.ReduceOp.SUM, async_op=True))
                for w in works:
                    w.wait()
            elif self.world == 1 and not self.force_collectives and hasattr(self.engine, "train_step"):
                self.engine.train_step(src, ctx, tgt, lr)      # nothing to reduce: the fused step
                return
            else:
                self.engine.forward_backward(src, ctx, tgt, sim_batch=B * self.world)
                if self.world > 1 or self.force_collectives:
                    # RCCL orders itself after the current stream = the stream the kernels run on
                    dist.all_reduce(self.engine.grads, op=dist.ReduceOp.SUM)
            self.engine.adam(lr)

    def scalars(self):
        """Global {loss, simloss, recon1, recon2} of the last step's forward pass."""
        stream = getattr(self.engine, "stream", None)
        with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
            return self._scalars()

    def _scalars(self):
        t = self.engine.scalars_tensor().clone()
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t[1] /= self.world                               # simloss: mean over equal shards
            # `loss` = the terms Adam minimises (ctx_config.loss_terms; ablations_code/ablations.py:175-182)
            terms = int(getattr(getattr(self.translator, "cfg", None), "loss_terms", 0) or 0) or 7
            t[0] = (t[2] if terms & 1 else 0) + (t[3] if terms & 2 else 0) + (t[1] if terms & 4 else 0)
        v = [float(x) for x in t.cpu()]
        return dict(loss=v[0], simloss=v[1], recon1=v[2], recon2=v[3])


class RcclTrainer:
    """Data-parallel training with the exchange step inside libctxtrans (include/ctxtrans.h: ctx_dp_*).  One instance per
    process / GPU.  `unique_id`: the blob from `Translator.dp_unique_id()` made on rank 0 -- when omitted and a
    torch.distributed group is initialised it is broadcast through that group (any backend)."""

    def __init__(self, H=64, W=64, df_dim=64, featsize=1024, max_batch=256, device=0, seed=1234, rank=None, world=None,
                 unique_id=None, precision=None):
        if rank is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if world is None:
            world = _world()
        self.rank, self.world = rank, world
        self.translator = Translator(H, W, df_dim, featsize, max_batch, device=device, precision=precision)
        self.translator.init_params(seed)
        self.n_params = self.translator.n_params
        if unique_id is None:
            unique_id = Translator.dp_unique_id() if rank == 0 else bytes(128)
            if world > 1:
                box = [unique_id]
                dist.broadcast_object_list(box, src=0)
                unique_id = box[0]
        self.translator.dp_init(unique_id, rank, world)

    def step(self, src, ctx, tgt, lr=1e-4, scalars=False):
        """src / ctx / tgt: contiguous float32 cuda tensors [B,H,W,3] of this rank's shard (ready on the device: the library's
        stream does not wait for torch's)."""
        return self.translator.dp_train_step(HipEngine._ptr(src), HipEngine._ptr(ctx), HipEngine._ptr(tgt), src.shape[0], lr, scalars)

    def scalars(self):
        return self.translator.dp_scalars()
