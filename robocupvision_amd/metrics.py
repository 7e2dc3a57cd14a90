"""Device-side evaluation metrics: the pixel accuracy / mean class accuracy / mean IoU that
the reference's valid() builds with O(B*C^2) Python mask loops and `.item()` syncs (train.py:136-178).  One kernel
(RCV_OP_CONFUSION) counts (pred, label) pairs per image; everything else is arithmetic on B*C*C integers."""
from __future__ import annotations

import torch

from . import _lib as L


class SegmentationMetrics:
    def __init__(self, num_class: int, device="cuda"):
        self.C = num_class
        self.device = torch.device(device)
        self.conf = torch.zeros(num_class, num_class, dtype=torch.float64, device=self.device)   # [pred][label]
        self.iou_sum = torch.zeros(num_class, dtype=torch.float64, device=self.device)
        self.n_img = 0

    def update(self, argmax_u8: torch.Tensor, targets: torch.Tensor):
        """argmax_u8: uint8 [B,H,W] (CrossEntropyLoss2d.last_argmax); targets: int64 [B,H,W]."""
        if argmax_u8.dtype != torch.uint8 or argmax_u8.device.type != "cuda":
            raise L.RcvError("SegmentationMetrics.update needs the uint8 arg-max mask on the HIP device")
        B, H, W = argmax_u8.shape
        targets = targets.to(torch.int64).contiguous()
        counts = torch.zeros(B, self.C, self.C, dtype=torch.int32, device=argmax_u8.device)
        h = L.handle(argmax_u8.device.index if argmax_u8.device.index is not None else torch.cuda.current_device())
        op = L.make_op(L.OP_CONFUSION, 0, n=B, h=H, w=W, cout=self.C, p_in=argmax_u8.contiguous().data_ptr(),
                       p_in2=targets.data_ptr(), p_out=counts.data_ptr())
        L.OpList([op]).run(h, torch.cuda.current_stream(argmax_u8.device).cuda_stream)
        c = counts.to(torch.float64)                       # [B][pred][label]
        inter = torch.diagonal(c, dim1=1, dim2=2)          # [B][C]
        union = c.sum(2) + c.sum(1) - inter                # |pred==c| + |label==c| - inter
        self.iou_sum += torch.where(union == 0, torch.ones_like(inter), inter / union.clamp(min=1)).sum(0)   # train.py:148-153
        self.conf += c.sum(0)
        self.n_img += B

    def compute(self) -> dict:
        lab_cnt = self.conf.sum(0)                         # pixels per label (train.py:142)
        total = float(self.conf.sum())
        class_acc = torch.diagonal(self.conf) / (lab_cnt / 100.0)        # conf[(j,j)] of train.py:157-163
        return {"pixel_acc": float(torch.diagonal(self.conf).sum()) / max(total, 1.0) * 100.0,
                "mean_class_acc": float(class_acc.sum()) / self.C,
                "mean_iou": float((self.iou_sum / max(self.n_img, 1)).sum()) / self.C * 100.0,
                "confusion_percent": (self.conf / (lab_cnt / 100.0)).cpu()}
