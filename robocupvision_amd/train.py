"""Counterpart of the reference's training-step body (train.py:43-74) for the HIP path, single GPU
or data parallel (one process per GPU, RCCL all-reduce of the flat gradient buffer over xGMI).

    trainer = Trainer(model.cuda(), class_weights=[1,10,30,10,2], lr=1e-3, decay=1e-6)
    for imgs, targets in loader:
        trainer.step(imgs.cuda(), targets.cuda())
    print(trainer.pop_metrics())

The reference's three per-step ``.item()`` host syncs (train.py:70-73) are replaced by on-device
accumulation of (loss, reg, #correct pixels); ``pop_metrics()`` reads them once per epoch.

Data-parallel semantics (the reference is single device): BatchNorm statistics are per
rank, the weighted-CE normaliser is per rank, gradients are averaged over ranks, every rank applies the
identical optimizer step, running BN buffers stay per rank (rank 0 is authoritative for checkpoints).
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import torch

from . import _lib as L
from .model import CrossEntropyLoss2d, DiceLoss
from .optim import AdamL1


class _HipStreams:
    """The stream operations GradExchange needs, on the real device (tests/test_dp_gloo.py injects a recording stand-in)."""

    def __init__(self, device, overlap: bool):
        self.device = device
        self.comm = torch.cuda.Stream(device=device) if overlap else None

    def current(self):
        return torch.cuda.current_stream(self.device)

    def wait(self, waiter, waited):
        waiter.wait_stream(waited)

    def ptr(self, stream) -> int:
        return stream.cuda_stream

    def on(self, stream):
        return torch.cuda.stream(stream)


class GradExchange:
    """Sums finished ranges of the flat gradient buffer over the ranks while backward is still running.

    ``grad_ready(lo, hi)`` is the engine's callback (engine.run_bucketed): flat.grad[lo:hi] is final on the compute stream,
    except for the filter gradients still in flight on the library's side stream.  With a communication stream, that stream
    (not the compute stream) waits for both and runs the all-reduce, so the remaining backward kernels keep going;
    ``finish()`` makes the compute stream wait for the last all-reduce before the optimizer reads the buffer."""

    def __init__(self, engine, streams, all_reduce, join_side=None):
        self.engine = engine
        self.streams = streams
        self.all_reduce = all_reduce
        self.join_side = join_side if join_side is not None else L.join_side
        self.pending = False
        self.ranges = []                  # (lo, hi) of this backward pass, in the order they were exchanged

    def begin(self):
        self.pending = False
        self.ranges = []

    def grad_ready(self, lo: int, hi: int):
        eng, st = self.engine, self.streams
        bucket = eng.flat.grad[lo:hi]
        self.ranges.append((lo, hi))
        cur = st.current()
        if st.comm is None:
            self.join_side(eng.handle, st.ptr(cur))        # filter gradients are produced on the library's side stream
            self.all_reduce(bucket)
            return
        st.wait(st.comm, cur)
        self.join_side(eng.handle, st.ptr(st.comm))
        with st.on(st.comm):
            self.all_reduce(bucket)
        self.pending = True

    def finish(self):
        if self.pending:
            self.streams.wait(self.streams.current(), self.streams.comm)
            self.pending = False


class Trainer:
    def __init__(self, model, class_weights: Optional[Sequence[float]] = (1, 10, 30, 10, 2), lr: float = 1e-3,
                 decay: float = 1e-6, transfer: int = 0, distributed: bool = False, overlap: bool = True,
                 use_dice: bool = False, optimizer=None, fuse_loss: bool = True, prune_indices=None):
        self.model = model
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise L.RcvError("Trainer needs the model on the HIP device (model.cuda())")
        self.device = dev
        w = None if class_weights is None else torch.tensor(list(class_weights), dtype=torch.float32, device=dev)
        if use_dice:       # train.py:315 (--useDice); class_weights are then the Dice weights of train.py:309
            if w is None:
                raise ValueError("DiceLoss needs class weights")
            self.criterion = DiceLoss(w).to(dev)
        else:
            self.criterion = CrossEntropyLoss2d(w).to(dev)
        # default: the train.py optimizer (Adam over 5 groups + L1 term); trainer.py's PB_FCN path passes optim.SGD(model, ...)
        self.optimizer = optimizer if optimizer is not None else AdamL1(model, lr=lr, decay=decay, transfer=transfer)
        self.metrics = torch.zeros(4, dtype=torch.float64, device=dev)     # loss, reg, correct, steps
        self.distributed = distributed
        self.fuse_loss = fuse_loss and not os.environ.get("RCV_NO_FUSED_LOSS")
        self.world = 1
        self.exchange: Optional[GradExchange] = None
        self.force_collectives = bool(int(os.environ.get("RCV_FORCE_COLLECTIVES", "0")))   # exercise the path at world size 1
        # train.py:59-65: gradients of pruned weights are zeroed after backward; prune_indices = pruneModelNew(model.parameters())
        # (one boolean mask per parameter with dim() > 1, in parameters() order)
        self.prune_indices = None
        if prune_indices is not None:
            self.set_prune_indices(prune_indices)
        if distributed:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise L.RcvError("distributed=True needs torch.distributed.init_process_group('nccl') first")
            self.world = dist.get_world_size()
            self.optimizer.grad_scale = 1.0 / self.world
            if os.environ.get("RCV_GRAD_BUCKETS"):       # number of gradient buckets (default 3)
                model._get_engine().grad_buckets = max(1, int(os.environ["RCV_GRAD_BUCKETS"]))
            self.exchange = GradExchange(model._get_engine(), _HipStreams(dev, overlap), dist.all_reduce)
            # identical parameters on every rank before the first step
            for p in model.parameters():
                dist.broadcast(p.data, 0)
            model._get_engine().invalidate()        # (.data writes bump no version counter: a cached eval head would be stale)

    def set_prune_indices(self, prune_indices):
        """The list pruneModelNew(model.parameters()) returns (train.py:345-347): masks aligned with the parameters of dim() > 1."""
        big = [p for p in self.model.parameters() if p.dim() > 1]
        prune_indices = list(prune_indices)
        if len(prune_indices) != len(big):
            raise ValueError("prune_indices has %d masks, the model has %d parameters with dim() > 1" % (len(prune_indices), len(big)))
        for m, p in zip(prune_indices, big):
            if tuple(m.shape) != tuple(p.shape):
                raise ValueError("prune mask %s does not match parameter %s" % (tuple(m.shape), tuple(p.shape)))
        self.prune_indices = [m.to(device=self.device, dtype=torch.bool) for m in prune_indices]
        if hasattr(self.optimizer, "set_prune_mask"):
            self.optimizer.set_prune_mask(self.prune_indices)

    def step(self, imgs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        """One train.py:43-74 iteration; returns the logits tensor (engine-owned, valid until the next forward)."""
        model, opt, crit = self.model, self.optimizer, self.criterion
        model.train()
        opt.zero_grad(set_to_none=True)
        eng = model._get_engine()
        exchanging = self.exchange is not None and (self.world > 1 or self.force_collectives)
        eng.grad_ready_cb = self.exchange.grad_ready if exchanging else None
        if exchanging:
            self.exchange.begin()
        fused = None
        if self.fuse_loss and type(crit) is CrossEntropyLoss2d and imgs.dtype == torch.float32 and imgs.is_contiguous():
            # fast path: the loss is evaluated inside the classifier op and its gradient inside the classifier's backward op
            # (no logits-gradient tensor, no separate loss kernels, no autograd graph); bit-identical to the path below
            w = crit.weight
            if w is not None and w.device != imgs.device:
                w = w.to(imgs.device)
                crit.weight = w
            fused = eng.forward_ce([imgs], targets.to(torch.int64).contiguous(), w)
        if fused is not None:
            pred, out, argmax = fused
            crit.last_stats, crit.last_argmax = out, argmax
            ce = out[0]
            eng.backward_ce()
            fl = eng.flat
            for k, p in enumerate(fl.params):
                if eng.param_used[k]:
                    p.grad = fl.grad_view(k)
        else:
            pred = model(imgs)
            ce = crit(pred, targets)
            ce.backward()
        if exchanging:
            self.exchange.finish()
        if self.prune_indices is not None and not hasattr(opt, "set_prune_mask"):
            # stock optimizers: the literal train.py:59-65 loop on the gradient views (they alias the flat buffer)
            k = 0
            for p in model.parameters():
                if p.dim() > 1:
                    if p.grad is not None:
                        p.grad[self.prune_indices[k]] = 0
                    k += 1
        if isinstance(opt, AdamL1):
            # loss + decay*sum|p|, reg, #correct and the step count are booked by the optimizer launch itself (the parameters
            # it reads are the pre-update ones the reference's l1reg(model) sees): no per-step torch reductions
            opt.step(metrics=self.metrics, loss_stats=crit.last_stats)
        else:
            with torch.no_grad():
                reg = opt.l1_term() if hasattr(opt, "l1_term") else torch.zeros((), device=self.device)
                self.metrics += torch.stack([ce.detach().double() + reg.double(), reg.double(),
                                             crit.last_stats[2].double(), torch.ones((), dtype=torch.float64, device=self.device)])
            opt.step()
        return pred

    def capture(self, imgs: torch.Tensor, targets: torch.Tensor):
        """Capture ONE whole training step (forward, loss, backward on both streams, optimizer + metrics) as a hipGraph and return
        ``step_fn(imgs, targets)`` that replays it: one graph launch per step instead of ~10 host calls enqueuing ~150 kernels --
        what the short steps (160x120: 2 ms of GPU work) need to stay GPU bound.  Same kernels in the same order on the same
        buffers, so results are those of ``step``.  Call after a few eager steps (plans, optimizer state and the measured backward
        schedule exist then).  The learning rate is baked in: capture again after a scheduler changed it.  Single GPU only."""
        if self.exchange is not None:
            raise L.RcvError("Trainer.capture: data-parallel steps run eagerly (the RCCL all-reduces are not captured)")
        opt = self.optimizer
        if not isinstance(opt, AdamL1):
            raise L.RcvError("Trainer.capture needs the fused AdamL1 optimizer")
        eng = self.model._get_engine()
        if eng.flat is None:
            self.step(imgs, targets)
        static_x, static_t = imgs.detach().clone(), targets.detach().to(torch.int64).clone()
        opt.use_device_step()
        self.step(static_x, static_t)                   # everything the step touches is allocated before the capture
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            pred = self.step(static_x, static_t)
        opt._t -= 1                                     # the capture recorded the step, it did not run it
        # the graph replays raw pointers into this plan's buffers: the plan must outlive it (LRU eviction skips pinned plans)
        plan = eng._last[0]
        plan.pinned += 1

        class _Pin:
            def __del__(self, plan=plan):
                plan.pinned -= 1

        def step_fn(x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
            if x.data_ptr() != static_x.data_ptr():
                static_x.copy_(x)
            if t.data_ptr() != static_t.data_ptr():
                static_t.copy_(t)
            graph.replay()
            opt._t += 1
            # the replayed kernels wrote parameters and BatchNorm buffers through raw pointers: no tensor version moved, so the
            # engine is told explicitly (an eval forward after this must not reuse packed filters / BN constants of older weights)
            eng.invalidate()
            return pred
        step_fn.graph = graph
        step_fn._pin = _Pin()                           # released together with step_fn (and its graph)
        return step_fn

    @torch.no_grad()
    def evaluate(self, imgs: torch.Tensor, targets: torch.Tensor):
        """valid() forward (train.py:102-131): eval-mode BN, CE, arg-max mask."""
        if self.model.training:
            self.model.eval()
        pred = self.model(imgs)
        loss = self.criterion(pred, targets)
        return pred, loss, self.criterion.last_argmax

    def pop_metrics(self) -> dict:
        m = self.metrics.cpu().tolist()
        self.metrics.zero_()
        n = max(m[3], 1.0)
        return {"loss": m[0] / n, "reg": m[1] / n, "correct_pixels": m[2], "steps": int(m[3])}
