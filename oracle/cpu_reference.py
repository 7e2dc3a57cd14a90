"""CPU oracle for the ROBO-UNet / U-Net training hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The shipped package
(``robocupvision_amd``) never imports anything under ``oracle/`` and fails loudly when the
HIP library is missing.

What it is: a functional restatement of the reference's arithmetic for the hot path, written
against a plain ``state_dict`` (name -> tensor) instead of ``nn.Module`` objects.  The
reference's arithmetic lives in stock PyTorch CPU operators, so the
restatement issues the same operator sequence on CPU tensors:

    reference item                       file:line               here
    Conv block  bn(relu(conv(x)))        model.py:105-116        conv_block()
    Pool        MaxPool2d(2,2)           model.py:92-100         level_down() (pool branch)
    LevelDown                            model.py:379-401        level_down()
    upSampleTransposeConv relu(bn(ct))   model.py:178-194        up_block()
    UltClassifier (1x1 conv)             model.py:403-414        classifier()
    ROBO_UNet.forward                    model.py:495-511        robo_unet_forward()
    CrossEntropyLoss2d                   model.py:76-82          cross_entropy_2d()
    l1reg                                train.py:23-27          l1reg()
    train step body                      train.py:43-74          train_step()
    LabelProp.forward / ConvPoolSimple   model.py:538-567,166-176 labelprop_forward()
    labelToPred                          transform.py:172-183    label_to_pred()
    valid() mask loops                   train.py:127,136-163    valid_metrics()

Parity pinning: ``tests/test_oracle_golden.py`` checks this file bit-for-bit (8 threads)
against golden vectors produced by importing the real reference in the build container
(``tests/golden/make_golden.py``).  The reference's own tests pin nothing at this boundary
(SURVEY.md F8), so those goldens are the pin.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


@dataclass(frozen=True)
class NetConfig:
    """Constructor arguments of ROBO_UNet (model.py:462) that shape the graph."""
    noScale: bool = False
    planes: int = 8
    nClass: int = 5
    depth: int = 4
    levels: int = 2
    bellySize: int = 5
    bellyPlanes: int = 128
    pool: bool = False
    v2: bool = False
    classSize: int = 1

    @property
    def eff_depth(self) -> int:          # model.py:469-470
        return self.depth + 1 if self.noScale else self.depth


def _bn(x: Tensor, sd: Dict[str, Tensor], prefix: str, training: bool) -> Tensor:
    # F.batch_norm updates running stats in place when training (what nn.BatchNorm2d does).
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    y = F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"],
                     training, BN_MOMENTUM, BN_EPS)
    if training and (prefix + ".num_batches_tracked") in sd:
        sd[prefix + ".num_batches_tracked"] += 1
    return y


def conv_block(x: Tensor, sd: Dict[str, Tensor], prefix: str, stride: int, training: bool) -> Tensor:
    """model.py:115-116  bn(relu(conv3x3(x)))  -- ReLU BEFORE BatchNorm."""
    z = F.conv2d(x, sd[prefix + ".conv.weight"], sd[prefix + ".conv.bias"], stride=stride, padding=1)
    return _bn(F.relu(z), sd, prefix + ".bn", training)


def up_block(x: Tensor, sd: Dict[str, Tensor], prefix: str, training: bool) -> Tensor:
    """model.py:190-194  relu(bn(convT(k3,s2,p1,op1)(x)))."""
    z = F.conv_transpose2d(x, sd[prefix + ".conv.weight"], sd[prefix + ".conv.bias"],
                           stride=2, padding=1, output_padding=1)
    return F.relu(_bn(z, sd, prefix + ".bn", training))


def level_layout(levels: int, do_pool: bool, pool: bool) -> Tuple[bool, List[int]]:
    """(has_maxpool, [stride of Conv0, Conv1, ...]) for one LevelDown (model.py:379-398)."""
    if pool:
        has_pool = bool(do_pool)
        if do_pool:
            levels -= 1
        return has_pool, [1] + [1] * max(levels - 1, 0)
    return False, [2 if do_pool else 1] + [1] * max(levels - 1, 0)


def level_down(x: Tensor, sd, prefix: str, levels: int, do_pool: bool, pool: bool, training: bool) -> Tensor:
    has_pool, strides = level_layout(levels, do_pool, pool)
    if has_pool:
        x = F.max_pool2d(x, 2, 2)
    for i, s in enumerate(strides):
        x = conv_block(x, sd, "%s.layers.Conv%d" % (prefix, i), s, training)
    return x


def classifier(x: Tensor, sd, prefix: str = "segmenter.layers.Class", size: int = 1) -> Tensor:
    return F.conv2d(x, sd[prefix + ".weight"], sd[prefix + ".bias"], padding=size // 2)


def robo_unet_forward(sd: Dict[str, Tensor], x: Tensor, cfg: NetConfig, training: bool) -> Tensor:
    """model.py:495-511."""
    depth = cfg.eff_depth
    downs = [x]
    downs.append(level_down(downs[-1], sd, "downPart.Level0", cfg.levels - 1, False, cfg.pool, training))
    for i in range(depth - 1):
        downs.append(level_down(downs[-1], sd, "downPart.Level%d" % (i + 1), cfg.levels, True, cfg.pool, training))
    if cfg.bellySize > 0:
        b = level_down(downs[-1], sd, "PB.PB_1", cfg.bellySize - 1, False, False, training)
        downs[-1] = level_down(b, sd, "PB.PB_2", 1, False, False, training)
    up = downs[-1]
    for i in range(depth - 1):
        t = up_block(up, sd, "upPart.Up%d" % i, training)
        skip = downs[-(i + 2)]
        up = torch.cat([t, skip], 1) if cfg.v2 else t + skip
    return classifier(up, sd, size=cfg.classSize)


def cross_entropy_2d(logits: Tensor, target: Tensor, weight: Optional[Tensor]) -> Tensor:
    """model.py:76-82  NLLLoss(weight, mean)(log_softmax(x, 1), t)."""
    return F.nll_loss(F.log_softmax(logits, dim=1), target, weight, reduction="mean")


def dice_weights(weights: Tensor) -> Tensor:
    """model.py:6-9: class weights rescaled to mean 1."""
    return weights / weights.sum().item() * weights.shape[0]


def dice_loss(logits: Tensor, target: Tensor, weights: Tensor, eps: float = 1e-7) -> Tensor:
    """model.py:11-43 (`weights` already rescaled by dice_weights):
    1 - mean_c( 2 w_c sum(P_c [t==c]) / (sum(P_c) + #[t==c] + eps) ), sums over (B,H,W); multi-class branch (model.py:34-37): P = softmax
    over channels; single-class branch (model.py:25-33): P = (sigmoid(z), 1 - sigmoid(z)) against the classes (t == 1, t == 0)."""
    C = logits.shape[1]
    if C == 1:
        hot = F.one_hot(target.long(), 2).permute(0, 3, 1, 2).to(logits.dtype)
        one_hot = torch.cat([hot[:, 1:2], hot[:, 0:1]], dim=1)
        pos = torch.sigmoid(logits)
        probas = torch.cat([pos, 1 - pos], dim=1)
    else:
        one_hot = F.one_hot(target.long(), C).permute(0, 3, 1, 2).to(logits.dtype)
        probas = F.softmax(logits, dim=1)
    dims = (0, 2, 3)
    intersection = torch.sum(probas * one_hot, dims)
    cardinality = torch.sum(probas + one_hot, dims)
    return 1 - (2.0 * weights * intersection / (cardinality + eps)).mean()


def l1reg(params: Sequence[Tensor]) -> Tensor:
    """train.py:23-27 (sum of |p| over every parameter, BN affine and biases included)."""
    reg = 0
    for p in params:
        reg = reg + torch.sum(torch.abs(p))
    return reg


def param_names(sd: Dict[str, Tensor]) -> List[str]:
    """Parameter (not buffer) names in state_dict == model.parameters() order."""
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


def param_groups(sd: Dict[str, Tensor], transfer: int = 0) -> List[List[str]]:
    """The five Adam groups of train.py:357-363 as lists of parameter names."""
    names = param_names(sd)
    def lvl(n):
        return int(n.split(".")[1][len("Level"):])
    g0 = [n for n in names if n.startswith("downPart.") and lvl(n) < transfer]
    g1 = [n for n in names if n.startswith("downPart.") and lvl(n) >= transfer]
    return [g0, g1, [n for n in names if n.startswith("PB.")],
            [n for n in names if n.startswith("upPart.")],
            [n for n in names if n.startswith("segmenter.")]]


class TrainState:
    """Parameters as autograd leaves + stock Adam, mirroring train.py:337-366."""

    def __init__(self, sd: Dict[str, Tensor], cfg: NetConfig, ce_weight: Sequence[float] = (1, 10, 30, 10, 2),
                 lr: float = 1e-3, decay: float = 1e-6, transfer: int = 0, use_dice: bool = False):
        self.cfg = cfg
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.names = param_names(self.sd)
        for n in self.names:
            self.sd[n].requires_grad_(True)
        self.ce_weight = torch.tensor(list(ce_weight), dtype=torch.float32)
        self.use_dice = use_dice          # train.py:315 (--useDice); ce_weight then holds the Dice class weights
        self.decay = decay
        groups = param_groups(self.sd, transfer)
        self.opt = torch.optim.Adam(
            [{"params": [self.sd[n] for n in groups[0]], "lr": lr * 10}]
            + [{"params": [self.sd[n] for n in g]} for g in groups[1:]], lr=lr)

    def params(self) -> List[Tensor]:
        return [self.sd[n] for n in self.names]


def prune_model_new(params: Sequence[Tensor], ratio: float = 0.01) -> List[Tensor]:
    """model.py:45-57: per parameter with dim() > 1, zero the weights below ratio*max|w| and return the masks."""
    indices = []
    with torch.no_grad():
        for param in params:
            if param.dim() > 1:
                thresh = torch.max(torch.abs(param)) * ratio
                param[torch.abs(param) < thresh] = 0
                indices.append(torch.abs(param) < thresh)
    return indices


def train_step(st: TrainState, imgs: Tensor, targets: Tensor, do_step: bool = True, indices=None) -> Dict[str, object]:
    """train.py:43-74: zero_grad, fwd, CE (+ decay*L1 unless pruning), backward, (gradients of pruned weights zeroed), Adam, argmax."""
    st.opt.zero_grad()
    pred = robo_unet_forward(st.sd, imgs, st.cfg, training=True)
    if st.use_dice:
        ce = dice_loss(pred, targets, dice_weights(st.ce_weight))
    else:
        ce = cross_entropy_2d(pred, targets, st.ce_weight)
    reg = torch.zeros(())
    loss = ce
    if indices is None:                      # train.py:52-55
        reg = st.decay * l1reg(st.params())
        loss = ce + reg
    loss.backward()
    if indices is not None:                  # train.py:59-65
        pIdx = 0
        for param in st.params():
            if param.dim() > 1:
                if param.grad is not None:
                    param.grad[indices[pIdx]] = 0
                pIdx += 1
    if do_step:
        st.opt.step()
    _, pred_class = torch.max(pred, 1)
    correct = int(torch.sum(pred_class == targets).item())
    return {"pred": pred.detach(), "ce": float(ce.item()), "reg": float(reg.item()),
            "loss": float(loss.item()), "pred_class": pred_class, "correct": correct}


# ----------------------------------------------------------------------------------------
# LabelProp (inference only; model.py:538-567).  ConvPoolSimple = relu(bn(conv(bias=False))).
# ----------------------------------------------------------------------------------------
def _cps(x, sd, prefix, stride, padding, dilation, training=False):
    z = F.conv2d(x, sd[prefix + ".conv.weight"], None, stride=stride, padding=padding, dilation=dilation)
    return F.relu(_bn(z, sd, prefix + ".bn", training))


def labelprop_forward(sd: Dict[str, Tensor], x: Tensor, training: bool = False) -> Tensor:
    top = _cps(x, sd, "pre", 1, 1, 1, training)
    middle = _cps(top, sd, "down1", 2, 1, 1, training)
    bottom = _cps(middle, sd, "down2", 2, 1, 1, training)
    x = _cps(bottom, sd, "down3", 2, 1, 1, training)
    x = _cps(x, sd, "conv1", 1, 2, 2, training)
    x = _cps(x, sd, "conv2", 1, 2, 2, training)
    x = _cps(x, sd, "conv3", 1, 2, 2, training)
    x = bottom + up_block(x, sd, "upConv1", training)
    x = middle + up_block(x, sd, "upConv2", training)
    x = up_block(x, sd, "upConv3", training)
    x = torch.cat([x[:, 0:8] + top, x[:, 8:]], 1)       # model.py:565 (in-place there)
    return F.conv2d(x, sd["classifier.weight"], sd["classifier.bias"])


# ----------------------------------------------------------------------------------------
# PB_FCN (model.py:126-142, 201-229, 269-309) and the trainer.py:205-221 step
# ----------------------------------------------------------------------------------------
def conv_pool(x, sd, prefix, training):
    """model.py:126-142: relu(bn(pool(relu(conv1(x))))), conv1 dilated by 2, 'pool' a stride-2 3x3 conv."""
    x = F.relu(F.conv2d(x, sd[prefix + ".conv1.weight"], None, padding=2, dilation=2))
    x = F.conv2d(x, sd[prefix + ".pool.weight"], None, stride=2, padding=1)
    return F.relu(_bn(x, sd, prefix + ".bn", training))


def pb_fcn_forward(sd: Dict[str, Tensor], x: Tensor, noScale: bool, training: bool) -> Tensor:
    """model.py:221-229 (DownSampler) + model.py:291-309 (PB_FCN, classify == 0)."""
    def belly(v):
        v = conv_pool(v, sd, "FCN.conv3", training)
        for name in ("conv4", "conv5", "conv6", "conv7", "conv8"):
            v = _cps(v, sd, "FCN." + name, 1, 2, 2, training)
        return v

    x0 = _cps(x, sd, "FCN.conv0", 1, 2, 2, training)
    x1 = _cps(x0, sd, "FCN.conv1", 2, 1, 1, training)
    x2 = conv_pool(x1, sd, "FCN.conv2", training)
    if noScale:
        x3 = conv_pool(x2, sd, "FCN.conv_ext", training)
        x4 = belly(x3)
        y = up_block(x4, sd, "up1", training) + x3
        y = up_block(y, sd, "up2", training) + x2
        y = up_block(y, sd, "up3", training) + x1
        y = up_block(y, sd, "up4", training) + x0
    else:
        x3 = belly(x2)
        y = up_block(x3, sd, "up1", training) + x2
        y = up_block(y, sd, "up2", training) + x1
        y = up_block(y, sd, "up3", training) + x0
    return F.conv2d(y, sd["segmenter.classifier.weight"], sd["segmenter.classifier.bias"])


def pb_fcn_2_forward(sd: Dict[str, Tensor], x: Tensor, training: bool, depth: int = 4, levels: int = 2, bellySize: int = 5) -> Tensor:
    """model.py:416-458 (PB_FCN_2, classify False): the ROBO-UNet graph with a single-conv Level0 (model.py:426)."""
    downs = [x]
    downs.append(level_down(downs[-1], sd, "downPart.Level0", 1, False, False, training))
    for i in range(depth - 1):
        downs.append(level_down(downs[-1], sd, "downPart.Level%d" % (i + 1), levels, True, False, training))
    b = level_down(downs[-1], sd, "PB.PB_1", bellySize - 1, False, False, training)
    downs[-1] = level_down(b, sd, "PB.PB_2", 1, False, False, training)
    up = downs[-1]
    for i in range(depth - 1):
        up = up_block(up, sd, "upPart.Up%d" % i, training) + downs[-(i + 2)]
    return classifier(up, sd)


class PBTrainState:
    """trainer.py:135-178: CrossEntropyLoss2d([1,6,1.5,3,3]) and SGD(lr .1, momentum .5, weight_decay 1e-3) over all parameters
    (those of the unused pooled classification head keep grad None and are skipped by SGD)."""

    def __init__(self, sd: Dict[str, Tensor], noScale: bool, ce_weight: Sequence[float] = (1, 6, 1.5, 3, 3), lr: float = 1e-1,
                 momentum: float = 0.5, weight_decay: float = 1e-3, v2: bool = False):
        self.noScale = noScale
        self.v2 = v2                      # trainer.py:126-127: PB_FCN_2 instead of PB_FCN
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.names = param_names(self.sd)
        for n in self.names:
            self.sd[n].requires_grad_(True)
        self.ce_weight = torch.tensor(list(ce_weight), dtype=torch.float32)
        self.opt = torch.optim.SGD([{"params": [self.sd[n] for n in self.names]}], lr=lr, momentum=momentum, weight_decay=weight_decay)


def pb_train_step(st: PBTrainState, imgs: Tensor, targets: Tensor, do_step: bool = True) -> Dict[str, object]:
    """trainer.py:205-224."""
    st.opt.zero_grad()
    pred = pb_fcn_2_forward(st.sd, imgs, True) if st.v2 else pb_fcn_forward(st.sd, imgs, st.noScale, training=True)
    loss = cross_entropy_2d(pred, targets, st.ce_weight)
    loss.backward()
    if do_step:
        st.opt.step()
    _, pred_class = torch.max(pred, 1)
    return {"pred": pred.detach(), "loss": float(loss.item()), "pred_class": pred_class,
            "correct": int(torch.sum(pred_class == targets).item())}


def valid_metrics(pred_class: Tensor, targets: Tensor, num_class: int) -> Dict[str, object]:
    """The mask loops of valid() (train.py:136-163; test.py:148-169 has the same shape): per image and (pred, label) pair the
    intersection counts -> confusion matrix in per cent of the label's pixels, per-image IoU (1 where the class is absent from both
    masks), pixel accuracy.  Restated with the reference's own fp32 accumulators; pure-Python loops: small cases only.
    (No reference fixture exists for these numbers -- valid() needs the dataset, cv2 and progressbar: "parity unpinned" -- the
    restatement follows the source lines one for one.)"""
    B = pred_class.shape[0]
    conf, iou, lab_cnts = torch.zeros(num_class, num_class), torch.zeros(num_class), torch.zeros(num_class)
    running_acc = 0.0
    out_size = 1.0 / float(pred_class[0].numel())                         # train.py:286 outSize = 1 / (H*W)
    running_acc += torch.sum(pred_class == targets).item() * out_size * 100     # train.py:127
    mask_pred = torch.stack([pred_class == c for c in range(num_class)])        # train.py:136-140
    mask_tgt = torch.stack([targets == c for c in range(num_class)])
    for img in range(B):                                                  # train.py:142-153
        for lab in range(num_class):
            lab_cnts[lab] += torch.sum(mask_tgt[lab, img]).item()
            for prd in range(num_class):
                inter = torch.sum(mask_pred[prd, img] & mask_tgt[lab, img]).item()
                conf[(prd, lab)] += inter
                if lab == prd:
                    union = torch.sum(mask_pred[prd, img] | mask_tgt[lab, img]).item()
                    iou[lab] += 1 if union == 0 else inter / union
    for lab in range(num_class):                                          # train.py:157-159
        for prd in range(num_class):
            conf[(prd, lab)] /= (lab_cnts[lab] / 100.0)
    mean_iou = torch.sum(iou / B).item() / num_class * 100                # train.py:161
    mean_class_acc = sum(float(conf[(j, j)]) for j in range(num_class)) / num_class      # train.py:162-163
    return {"pixel_acc": running_acc / B, "mean_class_acc": mean_class_acc, "mean_iou": mean_iou, "confusion_percent": conf}


def label_to_pred(label: Tensor, num_class: int) -> Tensor:
    """transform.py:172-183: +1 at the labelled class, -1 elsewhere, [B,C,H,W] float32."""
    B, H, W = label.shape
    out = torch.ones(B * H * W, num_class).scatter_(1, label.reshape(-1, 1), -1.0) * (-1)
    return out.view(B, H, W, num_class).permute(0, 3, 1, 2)


def labelprop_inputs(y_t: Tensor, y_n: Tensor, lab_t: Tensor, lab_n: Tensor, num_class: int = 5) -> Tensor:
    """labelPropTrain.py:178-182: one frame pair -> two 8-channel inputs (both directions)."""
    preds = label_to_pred(torch.stack([lab_t, lab_n]), num_class)
    a = torch.cat([y_t[None], y_n[None], (y_t - y_n)[None], preds[1]])
    b = torch.cat([y_n[None], y_t[None], (y_n - y_t)[None], preds[0]])
    return torch.stack([a, b])


# ----------------------------------------------------------------------------------------
# Synthetic inputs of SURVEY.md 8(d): identical on every machine (CPU generator).
# ----------------------------------------------------------------------------------------
def synthetic_batch(B: int, H: int, W: int, n_class: int = 5, seed: int = 1) -> Tuple[Tensor, Tensor]:
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, H, W, generator=g)
    t = torch.randint(0, n_class, (B, H, W), generator=g)
    return x, t


def conv_macs(cfg: NetConfig, H: int, W: int) -> Tuple[int, int]:
    """(forward MACs per image, MACs of the first conv) -- basis of the roofline FLOP count."""
    depth = cfg.eff_depth
    total, first = 0, None
    ch_in, h, w = 3, H, W
    def add(cin, cout, ho, wo, k=9):
        nonlocal total, first
        m = k * cin * cout * ho * wo
        if first is None:
            first = m
        total += m
    plan = [(cfg.planes, cfg.levels - 1, False)] + [(cfg.planes * 2 ** (i + 1), cfg.levels, True) for i in range(depth - 1)]
    for planes, levels, do_pool in plan:
        has_pool, strides = level_layout(levels, do_pool, cfg.pool)
        if has_pool:
            h, w = h // 2, w // 2
        for s in strides:
            h, w = h // s, w // s
            add(ch_in, planes, h, w)
            ch_in = planes
    if cfg.bellySize > 0:
        for _ in range(cfg.bellySize - 1):
            add(ch_in, cfg.bellyPlanes, h, w)
            ch_in = cfg.bellyPlanes
        add(ch_in, cfg.planes * 2 ** (depth - 1), h, w)
        ch_in = cfg.planes * 2 ** (depth - 1)
    for i in range(depth - 1):
        n_ch = cfg.planes * 2 ** (depth - 1 - i)
        o_ch = n_ch // 2
        add(n_ch * (2 if (i > 0 and cfg.v2) else 1), o_ch, h, w)   # convT: 9*Cin*Cout per INPUT pixel
        h, w = h * 2, w * 2
    add(cfg.planes * (2 if cfg.v2 else 1), cfg.nClass, h, w, k=cfg.classSize ** 2)
    return total, first
