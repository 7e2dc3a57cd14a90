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


class Trainer:
    def __init__(self, model, class_weights: Optional[Sequence[float]] = (1, 10, 30, 10, 2), lr: float = 1e-3,
                 decay: float = 1e-6, transfer: int = 0, distributed: bool = False, overlap: bool = True,
                 use_dice: bool = False, optimizer=None, fuse_loss: bool = True):
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
        self.comm_stream = None
        self.force_collectives = bool(int(os.environ.get("RCV_FORCE_COLLECTIVES", "0")))   # exercise the path at world size 1
        if distributed:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise L.RcvError("distributed=True needs torch.distributed.init_process_group('nccl') first")
            self.world = dist.get_world_size()
            self.optimizer.grad_scale = 1.0 / self.world
            self.comm_stream = torch.cuda.Stream(device=dev) if overlap else None
            if os.environ.get("RCV_GRAD_BUCKETS"):       # experiment knob: number of gradient buckets (default 3)
                model._get_engine().grad_buckets = max(1, int(os.environ["RCV_GRAD_BUCKETS"]))
            # identical parameters on every rank before the first step
            for p in model.parameters():
                dist.broadcast(p.data, 0)

    def _grad_ready(self, lo: int, hi: int):
        """flat.grad[lo:hi] is final (called from inside backward, reverse layer order): sum it over the ranks on the
        side stream while the remaining backward kernels keep the compute stream busy."""
        import torch.distributed as dist
        eng = self.model._get_engine()
        fl = eng.flat
        bucket = fl.grad[lo:hi]
        cur = torch.cuda.current_stream(self.device)
        if self.comm_stream is None:
            L.join_side(eng.handle, cur.cuda_stream)        # filter gradients are produced on the library's side stream
            dist.all_reduce(bucket)
            return
        self.comm_stream.wait_stream(cur)
        L.join_side(eng.handle, self.comm_stream.cuda_stream)
        with torch.cuda.stream(self.comm_stream):
            dist.all_reduce(bucket)
        self._pending = self.comm_stream

    def step(self, imgs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        """One train.py:43-74 iteration; returns the logits tensor (engine-owned, valid until the next forward)."""
        model, opt, crit = self.model, self.optimizer, self.criterion
        model.train()
        opt.zero_grad(set_to_none=True)
        self._pending = None
        eng = model._get_engine()
        eng.grad_ready_cb = self._grad_ready if (self.distributed and (self.world > 1 or self.force_collectives)) else None
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
        if self._pending is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._pending)
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

    @torch.no_grad()
    def evaluate(self, imgs: torch.Tensor, targets: torch.Tensor):
        """valid() forward (train.py:102-131): eval-mode BN, CE, arg-max mask."""
        self.model.eval()
        pred = self.model(imgs)
        loss = self.criterion(pred, targets)
        return pred, loss, self.criterion.last_argmax

    def pop_metrics(self) -> dict:
        m = self.metrics.cpu().tolist()
        self.metrics.zero_()
        n = max(m[3], 1.0)
        return {"loss": m[0] / n, "reg": m[1] / n, "correct_pixels": m[2], "steps": int(m[3])}
