"""Fused caller-side step pieces: Adam + the L1 term's gradient in one launch.

The reference adds ``decay * sum|p|`` to the loss (train.py:23-27,52-55) and lets autograd produce
``decay*sign(p)`` for every one of its 62-74 parameter tensors, then steps ``torch.optim.Adam`` over 5
parameter groups (train.py:357-363).  Here the parameters and their gradients already live in two flat
HBM buffers (engine.FlatParams), so the same update is ONE kernel (RCV_OP_ADAM_L1) over ~1.1 M floats.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L


def reference_param_groups(model, lr: float, transfer: int = 0):
    """The five groups of train.py:357-363 (group 0 = downPart[0:transfer] at 10x lr)."""
    return [{"params": list(model.downPart[0:transfer].parameters()), "lr": lr * 10},
            {"params": list(model.downPart[transfer:].parameters())},
            {"params": list(model.PB.parameters())},
            {"params": list(model.upPart.parameters())},
            {"params": list(model.segmenter.parameters())}]


class AdamL1(torch.optim.Optimizer):
    """torch.optim.Adam(betas=(.9,.999), eps=1e-8) + gradient of ``decay*sum|p|``, fused.

    It is a ``torch.optim.Optimizer`` (param_groups with per-group ``lr``), so the stock schedulers the
    reference uses (CosineAnnealingLR, train.py:366) drive it unchanged.  ``grad_scale`` multiplies the
    stored gradient first (1/world_size after a summing all-reduce)."""

    def __init__(self, model, lr: float = 1e-3, decay: float = 0.0, transfer: int = 0, betas=(0.9, 0.999), eps: float = 1e-8):
        self.model = model
        self.decay = float(decay)
        self.grad_scale = 1.0
        self._m: Optional[torch.Tensor] = None
        self._v: Optional[torch.Tensor] = None
        self._t = 0
        self._lr_elem = None
        self._lr_key = None
        self._met_ws: Optional[torch.Tensor] = None
        self._step_dev: Optional[torch.Tensor] = None     # device step counter (graph replay), see use_device_step()
        self._prune_src = None            # list of boolean masks (parameters with dim() > 1, parameters() order) or None
        self._prune_flat: Optional[torch.Tensor] = None
        super().__init__(reference_param_groups(model, lr, transfer), dict(lr=lr, betas=betas, eps=eps))

    def use_device_step(self):
        """Keep the Adam step number in a device int32 that every step() advances with a stream-ordered add and the launch reads
        (RCV_OP_ADAM_L1 p[IN_AUX]): the step can then be captured once and replayed as a hipGraph (Trainer.capture)."""
        if self._step_dev is None:
            _, fl = self._flat()
            self._step_dev = torch.full((1,), self._t, dtype=torch.int32, device=fl.data.device)

    def set_prune_mask(self, masks):
        """train.py:50-65 inside the fused step: ``masks`` = pruneModelNew(model.parameters()) (True = pruned weight).  As in
        the reference, a pruning run has NO L1 term (train.py:52-55: ``if indices is None: loss += decay*l1reg``): the launch
        uses decay 0, and zeroes the gradient of the masked elements before the moment update -- what
        ``param.grad[indices] = 0`` after ``loss.backward()`` leaves for torch.optim.Adam.  None switches it off."""
        self._prune_src = None if masks is None else list(masks)
        self._prune_flat = None

    def _flat(self):
        eng = self.model._get_engine()
        if eng.flat is None:
            raise L.RcvError("AdamL1.step() before the first forward: the engine has not laid out the parameters yet")
        return eng, eng.flat

    @torch.no_grad()
    def step(self, closure=None, metrics: Optional[torch.Tensor] = None, loss_stats: Optional[torch.Tensor] = None):
        """``metrics`` (float64[4], device) and ``loss_stats`` (the float row the loss wrote: [0] loss, [2] #correct pixels):
        the same launch also adds {loss + decay*sum|p|, decay*sum|p|, #correct, 1} to ``metrics`` (train.py:52-53,69-73)."""
        eng, fl = self._flat()
        if self._m is None or self._m.numel() != fl.numel or self._m.device != fl.data.device:
            self._m = torch.zeros_like(fl.data)
            self._v = torch.zeros_like(fl.data)
        # gradients normally ARE views of fl.grad; copy in the ones autograd had to clone (accumulation)
        for k, p in enumerate(fl.params):
            if p.grad is not None:
                view = fl.grad_view(k)
                if p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
        lrs = [(g["lr"], g["params"]) for g in self.param_groups if len(g["params"])]
        uniform = all(abs(lr - lrs[0][0]) == 0.0 for lr, _ in lrs)
        # a flat parameter outside the five groups, or one the graph never reads (grad None under the reference's autograd: the
        # pooled head of PB_FCN_2), must not be stepped at all: torch.optim.Adam skips it.  Per-element lr 0 = skip.
        grouped = set()
        for _, params in lrs:
            grouped.update(id(p) for p in params)
        stepped = [id(p) in grouped and eng.param_used[k] for k, p in enumerate(fl.params)]
        lr_elem_ptr = 0
        if not uniform or not all(stepped):
            key = (tuple(lr for lr, _ in lrs), tuple(stepped))
            if key != self._lr_key:
                t = torch.zeros_like(fl.data)
                for lr, params in lrs:
                    for p in params:
                        k = fl.index(p)
                        if stepped[k]:
                            t[fl.offsets[k]:fl.offsets[k] + p.numel()] = lr
                self._lr_elem, self._lr_key = t, key
            lr_elem_ptr = self._lr_elem.data_ptr()
        prune_ptr = 0
        if self._prune_src is not None:
            if self._prune_flat is None or self._prune_flat.numel() != fl.numel or self._prune_flat.device != fl.data.device:
                big = [k for k, p in enumerate(fl.params) if p.dim() > 1]
                if len(big) != len(self._prune_src):
                    raise L.RcvError("prune mask list has %d entries, the model has %d parameters with dim() > 1" % (len(self._prune_src), len(big)))
                t = torch.zeros(fl.numel, dtype=torch.uint8, device=fl.data.device)
                for k, m in zip(big, self._prune_src):
                    p = fl.params[k]
                    t[fl.offsets[k]:fl.offsets[k] + p.numel()] = m.to(device=fl.data.device, dtype=torch.uint8).reshape(-1)
                self._prune_flat = t
            prune_ptr = self._prune_flat.data_ptr()
        self._t += 1
        if self._step_dev is not None:
            self._step_dev.add_(1)
        b1, b2 = self.defaults["betas"]
        op = L.make_op(L.OP_ADAM_L1, 0, count=fl.numel, aux0=self._t, f0=lrs[0][0], f1=b1, f2=b2, f3=self.defaults["eps"],
                       f4=(0.0 if self._prune_src is not None else self.decay), f5=self.grad_scale, p_in=fl.data.data_ptr(), p_in2=fl.grad.data_ptr(),
                       p_x0=self._m.data_ptr(), p_x1=self._v.data_ptr(), p_x2=lr_elem_ptr, p_x5=prune_ptr,
                       p_in_aux=(self._step_dev.data_ptr() if self._step_dev is not None else 0))
        if metrics is not None:
            if loss_stats is None or metrics.dtype != torch.float64 or metrics.numel() < 4 or loss_stats.dtype != torch.float32 \
                    or loss_stats.numel() < 3 or not metrics.is_cuda or not loss_stats.is_cuda:
                raise L.RcvError("AdamL1.step(metrics=...): needs a float64[4] device tensor and the loss op's float stats row")
            nbytes = L.op_workspace(eng.handle, op)                  # fills op.i[NPART]
            if self._met_ws is None or self._met_ws.numel() * 8 < nbytes or self._met_ws.device != fl.data.device:
                self._met_ws = torch.zeros((nbytes + 7) // 8, dtype=torch.float64, device=fl.data.device)
            op.p[L.RCV_P_X3], op.p[L.RCV_P_X4], op.p[L.RCV_P_PART] = metrics.data_ptr(), loss_stats.data_ptr(), self._met_ws.data_ptr()
        L.OpList([op]).run(eng.handle, torch.cuda.current_stream(fl.data.device).cuda_stream)
        eng.params_dirty = True

    def l1_term(self) -> torch.Tensor:
        """decay * sum|p| (what train.py:53 logs as `reg`), one reduction over the flat buffer."""
        _, fl = self._flat()
        return self.decay * fl.data.abs().sum()


class SGD(torch.optim.Optimizer):
    """torch.optim.SGD(lr, momentum, weight_decay) as trainer.py:176-178 configures it (dampening 0, no nesterov), fused into
    ONE launch (RCV_OP_SGD) over the engine's flat parameter / gradient buffers.  Parameters the network graph never reads
    (PB_FCN's pooled classification head) are left untouched, like parameters whose ``grad`` is None under torch.optim.SGD.
    A ``torch.optim.Optimizer`` with one param group, so ReduceLROnPlateau (trainer.py:186) drives ``lr`` unchanged."""

    def __init__(self, model, lr: float, momentum: float = 0.0, weight_decay: float = 0.0):
        self.model = model
        self.grad_scale = 1.0
        self._buf: Optional[torch.Tensor] = None
        self._lr_elem: Optional[torch.Tensor] = None
        self._lr_key = None
        self._t = 0
        super().__init__([{"params": list(model.parameters())}], dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        eng = self.model._get_engine()
        fl = eng.flat
        if fl is None:
            raise L.RcvError("SGD.step() before the first forward: the engine has not laid out the parameters yet")
        if self._buf is None or self._buf.numel() != fl.numel or self._buf.device != fl.data.device:
            self._buf = torch.zeros_like(fl.data)
            self._t = 0
        for k, p in enumerate(fl.params):
            if p.grad is not None:
                view = fl.grad_view(k)
                if p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
        g = self.param_groups[0]
        lr = float(g["lr"])
        lr_elem_ptr = 0
        if not all(eng.param_used):
            if self._lr_key != lr:
                t = torch.zeros_like(fl.data)
                for k, p in enumerate(fl.params):
                    if eng.param_used[k]:
                        t[fl.offsets[k]:fl.offsets[k] + p.numel()] = lr
                self._lr_elem, self._lr_key = t, lr
            lr_elem_ptr = self._lr_elem.data_ptr()
        self._t += 1
        op = L.make_op(L.OP_SGD, 0, count=fl.numel, aux0=self._t, f0=lr, f1=g["momentum"], f2=g["weight_decay"], f5=self.grad_scale,
                       p_in=fl.data.data_ptr(), p_in2=fl.grad.data_ptr(), p_x0=self._buf.data_ptr(), p_x2=lr_elem_ptr)
        L.OpList([op]).run(eng.handle, torch.cuda.current_stream(fl.data.device).cuda_stream)
        eng.params_dirty = True
