"""Drop-in counterpart of the reference's ``model`` module for the ROBO-UNet / U-Net hot path.

Same public names, constructor arguments, sub-module names, ``state_dict`` keys and parameter
order as the reference (model.py:76-124, 166-199, 379-414, 461-567), so a caller written against
``from model import *`` (train.py:3,13; test.py:14; detect.py:13) keeps working -- but ``forward``
and ``backward`` run as hand-written HIP kernels on gfx950 through librcv.so (see engine.py).

The ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.ConvTranspose2d`` children exist only as parameter
containers (identical construction order => identical random init for a given seed, identical
state_dict); their own ``forward`` is never used.  There is no CPU path: calling a module on a
CPU tensor raises.
"""
from __future__ import annotations

from typing import List, Optional

import os

import torch
import torch.nn as nn

from . import _lib as L
from .engine import Engine

__all__ = ["ROBO_UNet", "CrossEntropyLoss2d", "DiceLoss", "PB_FCN", "PB_FCN_2", "LabelProp", "Conv", "Pool", "LevelDown",
           "upSampleTransposeConv", "UltClassifier", "ConvPoolSimple", "ConvPool", "DownSampler", "Classifier", "pruneModelNew",
           "count_zero_weights", "getParamSize"]


# ------------------------------------------------------------------------------------------
# autograd glue: one node for the whole network
# ------------------------------------------------------------------------------------------
class _EngineFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine: Engine, training: bool, n_inputs: int, *tensors):
        inputs = [t.detach() for t in tensors[:n_inputs]]
        out = engine.forward(inputs, training, fresh_out=not training and not ALIAS_OUTPUTS)
        ctx.engine = engine
        ctx.generation = engine._generation
        ctx.n_inputs = n_inputs
        ctx.n_params = len(tensors) - n_inputs
        return out

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.engine
        fl = eng.flat
        # a parameter whose .grad already aliases the engine's gradient buffer is about to be
        # overwritten: detach it first so autograd's accumulation keeps its meaning
        base, end = fl.grad.data_ptr(), fl.grad.data_ptr() + 4 * fl.numel
        for p in fl.params:
            if p.grad is not None and base <= p.grad.data_ptr() < end:
                p.grad = p.grad.clone()
        plan = eng.backward(grad_out, ctx.generation)
        in_grads = []
        for k in range(ctx.n_inputs):
            g = plan.input_grads[k] if ctx.needs_input_grad[3 + k] else None
            in_grads.append(g)
        # parameters the graph never reads (PB_FCN's pooled classification head) keep grad None, as under the reference's autograd
        p_grads = [fl.grad_view(k) if eng.param_used[k] else None for k in range(ctx.n_params)]
        return (None, None, None, *in_grads, *p_grads)


# The engine writes its result into a buffer it owns and overwrites at the next forward of the same shape.  The reference's modules
# return a fresh tensor every call (a caller may keep predictions across iterations), so by default the module surface hands out a
# copy (one extra pass over the logits: 0.08 ms at 32 x 5 x 480 x 640).  ALIAS_OUTPUTS = True (or RCV_ALIAS_OUTPUTS=1) returns the
# engine's buffer itself: valid until the next forward of the same shape.  (Trainer.step's fused path never goes through here.)
ALIAS_OUTPUTS = bool(os.environ.get("RCV_ALIAS_OUTPUTS"))


def _run_engine(engine: Engine, training: bool, inputs: List[torch.Tensor]) -> torch.Tensor:
    out = _EngineFunction.apply(engine, training, len(inputs), *inputs, *engine.param_list)
    # (an inference pass wrote its result into a fresh tensor already: Engine.forward(fresh_out=True))
    return out if (ALIAS_OUTPUTS or not training) else out.clone()


def _bn_modules(m: nn.Module):
    return [x for x in m.modules() if isinstance(x, nn.BatchNorm2d)]


class _EngineOwner:
    """Mixed into every module that can own an Engine.  The engine caches, per eval plan, everything that depends on the parameters
    only (packed filters, BatchNorm constants) and re-derives it when it can SEE a change: its own kernels wrote parameters
    (``params_dirty``) or a torch in-place op bumped a tensor's version counter.  Writes through ``.data`` (``p.data.mul_()``, the
    reference's ``pruneModel``: ``param = param.data; param[...] = 0``, model.py:626-640) bump no counter, so the cache is ALSO dropped
    at every ``.train()`` / ``.eval()`` call and after ``load_state_dict`` -- the reference's scripts call ``model.eval()`` at the top
    of every validation pass (train.py:104, trainer.py:162-171).  A caller that edits ``.data`` BETWEEN two eval-mode forwards without
    calling ``.eval()`` again must call ``invalidate()`` itself."""

    def invalidate(self):
        eng = self.__dict__.get("_engine")
        if eng is not None:
            eng.invalidate()

    def train(self, mode: bool = True):
        self.invalidate()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate()
        return out


class _BlockMixin(_EngineOwner):
    """Standalone call of a single block (NCHW in, NCHW out like the reference's blocks): the block
    becomes a one-node graph with an NHWC input and a materialised output."""

    def _block_graph(self):
        raise NotImplementedError

    def _block_forward(self, *xs: torch.Tensor) -> torch.Tensor:
        eng = self.__dict__.get("_engine")
        if eng is None:
            graph = self._block_graph()
            eng = Engine(graph, list(self.parameters()), _bn_modules(self))
            self.__dict__["_engine"] = eng
        ins = [x.permute(0, 2, 3, 1).contiguous() for x in xs]
        y = _run_engine(eng, self.training, ins)
        return y.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------
# building blocks (reference: model.py:92-124, 166-199, 379-414)
# ------------------------------------------------------------------------------------------
class Pool(_BlockMixin, nn.Module):
    """MaxPool2d(2,2) (model.py:92-103).  Parameter-free; inside a network it is fused with the
    producer's BatchNorm apply; called on its own (NCHW in, NCHW out, like the reference's module) it is a one-node graph."""

    def __init__(self, ch, stride=2):
        super().__init__()
        if stride != 2:
            raise NotImplementedError("only the 2x2/2 max-pool of the reference networks is built")
        self.ch = ch
        self.stride = stride
        self.pool = nn.MaxPool2d(stride, stride)     # container for repr/state parity only

    def _block_graph(self):
        return {"inputs": [{"layout": "nhwc", "requires_grad": True}], "nodes": [{"op": "pool", "src": ("in", 0)}]}

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] % 4 or x.shape[2] % 2 or x.shape[3] % 2:
            raise ValueError("Pool expects float32 [B,C,H,W] with C a multiple of 4 and even H, W (got %s)" % (tuple(x.shape),))
        return self._block_forward(x.to(torch.float32))

    def getComp(self, W, H, pruned):
        return W * H * self.ch, W // self.stride, H // self.stride


class Conv(_BlockMixin, nn.Module):
    """bn(relu(conv3x3(x))) -- ReLU before BatchNorm (model.py:105-124)."""

    def __init__(self, inplanes, planes, size, stride=1):
        super().__init__()
        if size != 3:
            raise NotImplementedError("only 3x3 Conv blocks are built (reference networks use size=3)")
        self.stride = stride
        self.size = size
        self.inch = inplanes
        self.ch = planes
        self.conv = nn.Conv2d(inplanes, planes, kernel_size=size, padding=size // 2, stride=stride)
        self.bn = nn.BatchNorm2d(planes)

    def _node(self, src):
        return {"op": "conv", "src": src, "weight": self.conv.weight, "bias": self.conv.bias, "bn": self.bn,
                "stride": self.stride, "dil": 1, "order": "relu_bn"}

    def _block_graph(self):
        return {"inputs": [{"layout": "nhwc", "requires_grad": True}],
                "nodes": [self._node(("in", 0)), {"op": "mat", "src": ("node", 0)}]}

    def forward(self, x):
        return self._block_forward(x)

    def getComp(self, W, H, pruned):
        W = W // self.stride
        H = H // self.stride
        ratio = float(self.conv.weight.nonzero().size(0)) / float(self.conv.weight.numel()) if pruned else 1
        return self.size * self.size * W * H * self.inch * self.ch * 2 * ratio + W * H * self.ch * 4, W, H


class ConvPoolSimple(_BlockMixin, nn.Module):
    """relu(bn(conv(x))) with dilation (model.py:166-176): LabelProp inference and the PB_FCN encoder."""

    def __init__(self, inplanes, planes, size, stride, padding, dilation, bias, *_ignored):
        super().__init__()
        if size != 3 or padding != dilation:
            raise NotImplementedError("ConvPoolSimple is built for 3x3 kernels with padding == dilation")
        self.stride, self.dilation = stride, dilation
        self.conv = nn.Conv2d(inplanes, planes, size, stride=stride, padding=padding, dilation=dilation, bias=bias)
        self.bn = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU()

    def _node(self, src):
        return {"op": "conv", "src": src, "weight": self.conv.weight, "bias": self.conv.bias, "bn": self.bn,
                "stride": self.stride, "dil": self.dilation, "order": "bn_relu"}

    def _block_graph(self):
        return {"inputs": [{"layout": "nhwc", "requires_grad": True}],
                "nodes": [self._node(("in", 0)), {"op": "mat", "src": ("node", 0)}]}

    def forward(self, x):
        return self._block_forward(x)


class ConvPool(_BlockMixin, nn.Module):
    """relu(bn(pool(relu(conv1(x))))): a dilated 3x3 conv, then a stride-2 3x3 conv as the 'pool' (model.py:126-142)."""

    def __init__(self, inplanes, planes):
        super().__init__()
        self.relu = nn.ReLU()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, dilation=2, padding=2, bias=False)
        self.pool = nn.Conv2d(planes, planes, kernel_size=3, padding=1, stride=2, bias=False)
        self.bn = nn.BatchNorm2d(planes)

    def _nodes(self, nodes: list, src):
        nodes.append({"op": "conv", "src": src, "weight": self.conv1.weight, "bias": None, "bn": None, "stride": 1, "dil": 2,
                      "order": "relu"})
        nodes.append({"op": "conv", "src": ("node", len(nodes) - 1), "weight": self.pool.weight, "bias": None, "bn": self.bn,
                      "stride": 2, "dil": 1, "order": "bn_relu"})
        return ("node", len(nodes) - 1)

    def _block_graph(self):
        nodes: list = []
        out = self._nodes(nodes, ("in", 0))
        nodes.append({"op": "mat", "src": out})
        return {"inputs": [{"layout": "nhwc", "requires_grad": True}], "nodes": nodes}

    def forward(self, x):
        return self._block_forward(x)


class upSampleTransposeConv(_BlockMixin, nn.Module):
    """relu(bn(ConvTranspose2d(k3,s2,p1,op1)(x))) (model.py:178-199)."""

    def __init__(self, inplanes, planes):
        super().__init__()
        self.stride = 2
        self.size = 3
        self.inch = inplanes
        self.ch = planes
        self.relu = nn.ReLU()
        self.conv = nn.ConvTranspose2d(inplanes, planes, kernel_size=3, padding=1, stride=2, output_padding=1, bias=True)
        self.bn = nn.BatchNorm2d(planes)

    def _node(self, src, skip, concat=False):
        return {"op": "up", "src": src, "skip": skip, "concat": concat, "weight": self.conv.weight, "bias": self.conv.bias,
                "bn": self.bn}

    def _block_graph(self):
        return {"inputs": [{"layout": "nhwc", "requires_grad": True}],
                "nodes": [self._node(("in", 0), None), {"op": "mat", "src": ("node", 0)}]}

    def forward(self, x):
        return self._block_forward(x)

    def getComp(self, W, H, pruned):
        ratio = float(self.conv.weight.nonzero().size(0)) / float(self.conv.weight.numel()) if pruned else 1
        return self.size * self.size * W * H * self.inch * self.ch * 2 * ratio + W * H * self.ch * 4, W * self.stride, H * self.stride


class LevelDown(nn.Module):
    """One encoder level (model.py:379-401): [Pool], Conv0 (stride 2 when downsampling without
    pooling), Conv1.. ; children are named exactly like the reference (state_dict keys)."""

    def __init__(self, inplanes, planes, levels, doPool, pool=False):
        super().__init__()
        self.layers = nn.Sequential()
        first_stride = 1
        if pool:
            if doPool:
                self.layers.add_module("Pool", Pool(inplanes, 2))
                levels -= 1
        elif doPool:
            first_stride = 2
        self.layers.add_module("Conv0", Conv(inplanes, planes, 3, stride=first_stride))
        for i in range(levels - 1):
            self.layers.add_module("Conv%d" % (i + 1), Conv(planes, planes, 3))

    def _nodes(self, nodes: list, src):
        """Append this level's nodes; returns the reference of its output."""
        for m in self.layers:
            if isinstance(m, Pool):
                nodes.append({"op": "pool", "src": src})
            else:
                nodes.append(m._node(src))
            src = ("node", len(nodes) - 1)
        return src

    def forward(self, x):
        for m in self.layers:
            x = m(x)
        return x


class UltClassifier(nn.Module):
    """1x1 (or, for the v2 net, 3x3) classifier (model.py:403-414); the pooled/dropout variant belongs to the
    patch classification scripts and is out of scope."""

    def __init__(self, inplanes, nClass, pool, dropout=0.5, size=1):
        super().__init__()
        if size not in (1, 3):
            raise NotImplementedError("classifier kernel sizes 1 and 3 are built (got %d)" % size)
        self.pooled = bool(pool)
        self.layers = nn.Sequential()
        if pool:        # patch-classification head (PB_FCN_2.classifier): holds its parameters (state_dict / init parity), never executed
            self.layers.add_module("Pool", nn.AdaptiveAvgPool2d(1))
            self.layers.add_module("DO", nn.Dropout2d(dropout))
        self.layers.add_module("Class", nn.Conv2d(inplanes, nClass, size, padding=size // 2))

    def _node(self, src):
        if self.pooled:
            raise L.RcvError("the pooled classification head (classify=True) is outside the segmentation path")
        c = self.layers.Class
        return {"op": "cls", "src": src, "weight": c.weight, "bias": c.bias}

    def forward(self, x):
        raise L.RcvError("UltClassifier is executed as part of its network graph; standalone use is not built")


# ------------------------------------------------------------------------------------------
# ROBO_UNet (model.py:461-536)
# ------------------------------------------------------------------------------------------
class ROBO_UNet(_EngineOwner, nn.Module):
    def __init__(self, noScale=False, planes=8, nClass=5, depth=4, levels=2, bellySize=5, bellyPlanes=128, pool=False, v2=False,
                 classSize=1):
        super().__init__()
        self.numClass = nClass
        self.planes = planes
        self.v2 = v2
        self.img_shape = (240, 320) if noScale else (120, 160)
        if noScale:
            depth += 1
        maxDepth = planes * pow(2, depth - 1)

        self.downPart = nn.ModuleList()
        self.downPart.add_module("Level0", LevelDown(3, planes, levels - 1, False, pool))
        for i in range(depth - 1):
            nCh = planes * pow(2, i)
            self.downPart.add_module("Level%d" % (i + 1), LevelDown(nCh, nCh * 2, levels, True, pool))

        self.PB = nn.Sequential()
        if bellySize > 0:
            self.PB.add_module("PB_1", LevelDown(maxDepth, bellyPlanes, bellySize - 1, False))
            self.PB.add_module("PB_2", LevelDown(bellyPlanes, maxDepth, 1, False))

        self.upPart = nn.ModuleList()
        for i in range(depth - 1):
            nCh = planes * pow(2, depth - 1 - i)
            oCh = nCh // 2
            if i > 0 and v2:      # v2: the previous level's output is concatenated with its skip tensor (model.py:487-488,507)
                nCh *= 2
            self.upPart.add_module("Up%d" % i, upSampleTransposeConv(nCh, oCh))

        self.segmenter = UltClassifier(planes * 2 if v2 else planes, nClass, False, size=classSize)

    # graph of model.py:495-511
    def _graph(self):
        nodes: list = []
        downs = [("in", 0)]
        for level in self.downPart:
            downs.append(level._nodes(nodes, downs[-1]))
        for level in self.PB:
            downs[-1] = level._nodes(nodes, downs[-1])
        up = downs[-1]
        for i, layer in enumerate(self.upPart):
            nodes.append(layer._node(up, downs[-(i + 2)], concat=self.v2))
            up = ("node", len(nodes) - 1)
        nodes.append(self.segmenter._node(up))
        return {"inputs": [{"layout": "nchw"}], "nodes": nodes}

    def _get_engine(self) -> Engine:
        eng = self.__dict__.get("_engine")
        if eng is None:
            eng = Engine(self._graph(), list(self.parameters()), _bn_modules(self))
            self.__dict__["_engine"] = eng
        return eng

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("ROBO_UNet expects float32 [B,3,H,W], got %s" % (tuple(x.shape),))
        down = 2 ** (len(self.downPart) - 1)
        if x.shape[2] % down or x.shape[3] % down:
            raise ValueError("H and W must be multiples of %d (got %dx%d)" % (down, x.shape[2], x.shape[3]))
        x = x.to(torch.float32).contiguous()
        return _run_engine(self._get_engine(), self.training, [x])

    def get_computations(self, pruned=False):
        """Analytic per-layer operation counts (model.py:513-536); host arithmetic only."""
        H, W = self.img_shape
        computations = []
        for part in self.downPart:
            for module in part.layers:
                comp, W, H = module.getComp(W, H, pruned)
                computations.append(comp)
        for part in self.PB:
            for module in part.layers:
                comp, W, H = module.getComp(W, H, pruned)
                computations.append(comp)
        for module in self.upPart:
            comp, W, H = module.getComp(W, H, pruned)
            computations.append(comp)
        computations.append(self.img_shape[0] * self.img_shape[1] * self.numClass * self.planes * 2)
        return computations


# ------------------------------------------------------------------------------------------
# CrossEntropyLoss2d (model.py:76-82) + fused arg-max / accuracy (train.py:70-71)
# ------------------------------------------------------------------------------------------
class _CEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, weight, module):
        if logits.device.type != "cuda":
            raise L.RcvError("CrossEntropyLoss2d runs on the HIP device only (got %s)" % logits.device)
        h = L.handle(logits.device.index if logits.device.index is not None else torch.cuda.current_device())
        logits_c = logits.detach().to(torch.float32).contiguous()
        targets_c = targets.detach().to(torch.int64).contiguous()
        N, Cc, H, W = logits_c.shape
        if targets_c.shape != (N, H, W):
            raise ValueError("targets must be [B,H,W] matching logits %s, got %s" % (tuple(logits_c.shape), tuple(targets_c.shape)))
        out = torch.empty(4, dtype=torch.float32, device=logits.device)
        argmax = torch.empty(N, H, W, dtype=torch.uint8, device=logits.device)
        op = L.make_op(L.OP_CE_FWD, L.F_ARGMAX, n=N, h=H, w=W, cout=Cc, p_in=logits_c.data_ptr(), p_in2=targets_c.data_ptr(),
                       p_w=(weight.data_ptr() if weight is not None else 0), p_out=out.data_ptr(), p_x0=argmax.data_ptr())
        nbytes = L.op_workspace(h, op)
        part = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=logits.device)
        op.p[L.RCV_P_PART] = part.data_ptr()
        L.OpList([op]).run(h, torch.cuda.current_stream(logits.device).cuda_stream)
        ctx.save_for_backward(logits_c, targets_c, out)
        ctx.weight = weight
        ctx.handle = h
        module.last_argmax = argmax
        module.last_stats = out          # [loss, sum_w, #correct, sum_w*nll]
        return out[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        logits, targets, out = ctx.saved_tensors
        N, Cc, H, W = logits.shape
        go = grad_out.detach().to(torch.float32).reshape(1).contiguous()
        dl = torch.empty_like(logits)
        w = ctx.weight
        op = L.make_op(L.OP_CE_BWD, 0, n=N, h=H, w=W, cout=Cc, p_in=logits.data_ptr(), p_in2=targets.data_ptr(),
                       p_w=(w.data_ptr() if w is not None else 0), p_x0=out.data_ptr(), p_x1=go.data_ptr(), p_out=dl.data_ptr())
        L.OpList([op]).run(ctx.handle, torch.cuda.current_stream(logits.device).cuda_stream)
        return dl, None, None, None


class CrossEntropyLoss2d(nn.Module):
    """NLLLoss(weight, mean)(log_softmax(inputs, 1), targets) (model.py:76-82).  After a call,
    ``last_argmax`` (uint8 [B,H,W], first maximum on ties) and ``last_stats[2]`` (#pixels whose
    arg-max equals the target) hold what train.py:70-71 computes with torch.max."""

    def __init__(self, weight=None):
        super().__init__()
        self.register_buffer("weight", None if weight is None else torch.as_tensor(weight, dtype=torch.float32).clone())
        self.last_argmax: Optional[torch.Tensor] = None
        self.last_stats: Optional[torch.Tensor] = None

    def forward(self, inputs, targets):
        w = self.weight
        if w is not None and w.device != inputs.device:
            w = w.to(inputs.device)
            self.weight = w
        return _CEFunction.apply(inputs, targets, w, self)


# ------------------------------------------------------------------------------------------
# DiceLoss (model.py:5-43; train.py:315 --useDice) + the same fused arg-max / accuracy
# ------------------------------------------------------------------------------------------
class _DiceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, weight, eps, module):
        if logits.device.type != "cuda":
            raise L.RcvError("DiceLoss runs on the HIP device only (got %s)" % logits.device)
        h = L.handle(logits.device.index if logits.device.index is not None else torch.cuda.current_device())
        logits_c = logits.detach().to(torch.float32).contiguous()
        N, Cc, H, W = logits_c.shape
        targets_c = targets.detach().to(torch.int64).reshape(N, H, W).contiguous()     # [B,H,W] or [B,1,H,W] (model.py:36)
        out = torch.empty(4 + 16, dtype=torch.float32, device=logits.device)
        argmax = torch.empty(N, H, W, dtype=torch.uint8, device=logits.device)
        op = L.make_op(L.OP_DICE_FWD, L.F_ARGMAX, n=N, h=H, w=W, cout=Cc, f1=eps, p_in=logits_c.data_ptr(), p_in2=targets_c.data_ptr(),
                       p_w=weight.data_ptr(), p_out=out.data_ptr(), p_x0=argmax.data_ptr())
        nbytes = L.op_workspace(h, op)
        part = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=logits.device)
        op.p[L.RCV_P_PART] = part.data_ptr()
        L.OpList([op]).run(h, torch.cuda.current_stream(logits.device).cuda_stream)
        ctx.save_for_backward(logits_c, targets_c, out)
        ctx.handle = h
        module.last_argmax = argmax
        module.last_stats = out          # [loss, -, #correct, -, backward coefficients...]
        return out[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        logits, targets, out = ctx.saved_tensors
        N, Cc, H, W = logits.shape
        go = grad_out.detach().to(torch.float32).reshape(1).contiguous()
        dl = torch.empty_like(logits)
        op = L.make_op(L.OP_DICE_BWD, 0, n=N, h=H, w=W, cout=Cc, p_in=logits.data_ptr(), p_in2=targets.data_ptr(),
                       p_x0=out.data_ptr(), p_x1=go.data_ptr(), p_out=dl.data_ptr())
        L.OpList([op]).run(ctx.handle, torch.cuda.current_stream(logits.device).cuda_stream)
        return dl, None, None, None, None


class DiceLoss(nn.Module):
    """1 - mean_c(2 w_c I_c / (S_c + N_c + eps)) over softmax probabilities (model.py:5-43); the class weights are rescaled to
    mean 1 exactly as model.py:8.  ``last_argmax`` / ``last_stats[2]`` as CrossEntropyLoss2d.  The single-class branch (model.py:25-33: one
    logit channel z, probabilities (sigmoid z, 1 - sigmoid z) against the classes (t == 1, t == 0)) runs on the same kernels: those
    probabilities are the softmax of (z, 0), so the logits are widened by a zero channel and the target is flipped."""

    def __init__(self, weights, eps=1e-7):
        super().__init__()
        weights = torch.as_tensor(weights, dtype=torch.float32)
        self.register_buffer("weights", (weights / weights.sum().item() * weights.shape[0]).clone())
        self.eps = eps
        self.last_argmax: Optional[torch.Tensor] = None
        self.last_stats: Optional[torch.Tensor] = None

    def forward(self, logits, true):
        if logits.shape[1] == 1:
            if self.weights.numel() not in (1, 2):
                raise ValueError("single-class DiceLoss takes 1 or 2 class weights (got %d)" % self.weights.numel())
            if true.dim() == 4:
                true = true[:, 0]
            logits = torch.cat([logits, torch.zeros_like(logits)], dim=1)        # softmax((z, 0)) = (sigmoid z, 1 - sigmoid z)
            true = 1 - true.long()                                               # class 0 <-> target == 1 (model.py:28-30)
            if self.weights.numel() == 1:
                self.weights = self.weights.expand(2).clone()
        if logits.shape[1] != self.weights.shape[0]:
            raise ValueError("DiceLoss has %d class weights but the logits have %d channels" % (self.weights.shape[0], logits.shape[1]))
        w = self.weights
        if w.device != logits.device:
            w = w.to(logits.device)
            self.weights = w
        return _DiceFunction.apply(logits, true, w, float(self.eps), self)


# ------------------------------------------------------------------------------------------
# LabelProp (model.py:538-567), inference only
# ------------------------------------------------------------------------------------------
class LabelProp(_EngineOwner, nn.Module):
    def __init__(self, numClass, numPlanes, dropout=0.0):
        super().__init__()
        self.pre = ConvPoolSimple(8, numPlanes // 4, 3, 1, 1, 1, False, dropout)
        self.down1 = ConvPoolSimple(numPlanes // 4, numPlanes // 2, 3, 2, 1, 1, False, dropout)
        self.down2 = ConvPoolSimple(numPlanes // 2, numPlanes // 2, 3, 2, 1, 1, False, dropout)
        self.down3 = ConvPoolSimple(numPlanes // 2, numPlanes, 3, 2, 1, 1, False, dropout)
        self.conv1 = ConvPoolSimple(numPlanes, numPlanes * 2, 3, 1, 2, 2, False, dropout)
        self.conv2 = ConvPoolSimple(numPlanes * 2, numPlanes * 2, 3, 1, 2, 2, False, dropout)
        self.conv3 = ConvPoolSimple(numPlanes * 2, numPlanes, 3, 1, 2, 2, False, dropout)
        self.upConv1 = upSampleTransposeConv(numPlanes, numPlanes // 2)
        self.upConv2 = upSampleTransposeConv(numPlanes // 2, numPlanes // 2)
        self.upConv3 = upSampleTransposeConv(numPlanes // 2, numPlanes // 2)
        self.classifier = nn.Conv2d(numPlanes // 2, numClass, 1, padding=0)

    # graph of model.py:556-567
    def _graph(self):
        nodes = []

        def add(node):
            nodes.append(node)
            return ("node", len(nodes) - 1)

        top = add(self.pre._node(("in", 0)))
        middle = add(self.down1._node(top))
        bottom = add(self.down2._node(middle))
        x = add(self.down3._node(bottom))
        x = add(self.conv1._node(x))
        x = add(self.conv2._node(x))
        x = add(self.conv3._node(x))
        x = add(self.upConv1._node(x, bottom))          # bottom + upConv1(x)
        x = add(self.upConv2._node(x, middle))          # middle + upConv2(x)
        x = add(self.upConv3._node(x, None))
        x = add({"op": "add_slice", "src": x, "add": top})      # x[:, 0:8] += top
        add({"op": "cls", "src": x, "weight": self.classifier.weight, "bias": self.classifier.bias})
        return {"inputs": [{"layout": "nhwc", "requires_grad": False}], "nodes": nodes}

    def forward(self, x):
        """Inference only (BASELINE config 5): x float32 [B,8,H,W] -> logits [B,numClass,H,W]."""
        if self.training:
            raise L.RcvError("LabelProp is built for inference only; call .eval()")
        if x.dim() != 4 or x.shape[1] != 8 or x.shape[2] % 8 or x.shape[3] % 8:
            raise ValueError("LabelProp expects float32 [B,8,H,W] with H,W multiples of 8, got %s" % (tuple(x.shape),))
        eng = self.__dict__.get("_engine")
        if eng is None:
            eng = Engine(self._graph(), list(self.parameters()), _bn_modules(self))
            self.__dict__["_engine"] = eng
        return _run_engine(eng, False, [x.to(torch.float32).permute(0, 2, 3, 1).contiguous()])


# ------------------------------------------------------------------------------------------
# host-side helpers of the module surface (model.py:45-74) -- scalar bookkeeping, not hot path
# ------------------------------------------------------------------------------------------
def pruneModelNew(params, ratio=0.01):
    """Magnitude pruning masks (model.py:45-57): zero weights below ratio*max|w|, return the masks."""
    indices = []
    for param in params:
        if param.dim() > 1:
            with torch.no_grad():
                thresh = torch.max(torch.abs(param)) * ratio
                mask = torch.abs(param) < thresh
                print("Pruned %f%% of the weights" % (float(torch.sum(mask)) / float(torch.sum(param != 0)) * 100))
                param[mask] = 0
                indices.append(torch.abs(param) < thresh)
    return indices


def count_zero_weights(model):
    """Fraction of weights below 1% of their tensor's maximum (model.py:59-66)."""
    small = 0.0
    total = 0
    for param in model.parameters():
        mx = torch.max(torch.abs(param))
        small += float((torch.abs(param) < mx * 0.01).sum())
        total += param.numel()
    return float(small / total)


def getParamSize(x):
    n = 1
    for s in x.size():
        n *= s
    return n


class DownSampler(nn.Module):
    """PB_FCN encoder (model.py:201-237): children and construction order as the reference (state_dict / init parity)."""

    def __init__(self, planes, noScale):
        super().__init__()
        self.noScale = noScale
        outPlanes = planes // 4
        self.conv0 = ConvPoolSimple(3, outPlanes, 3, 1, 2, 2, False)
        self.conv1 = ConvPoolSimple(outPlanes, planes // 2, 3, 2, 1, 1, False)
        self.conv2 = ConvPool(planes // 2, planes)
        self.conv_ext = ConvPool(planes, planes) if noScale else None
        self.conv3 = ConvPool(planes, planes * 2)
        self.conv4 = ConvPoolSimple(planes * 2, planes * 4, 3, 1, 2, 2, False)
        self.conv5 = ConvPoolSimple(planes * 4, planes * 4, 3, 1, 2, 2, False)
        self.conv6 = ConvPoolSimple(planes * 4, planes * 4, 3, 1, 2, 2, False)
        self.conv7 = ConvPoolSimple(planes * 4, planes * 4, 3, 1, 2, 2, False)
        self.conv8 = ConvPoolSimple(planes * 4, planes * 2, 3, 1, 2, 2, False)

    def _nodes(self, nodes: list, src):
        """model.py:221-229; returns the references of (x4, x3, x2, x1, x0) (x4 is None without noScale)."""
        def add(node):
            nodes.append(node)
            return ("node", len(nodes) - 1)

        def belly(x):
            x = self.conv3._nodes(nodes, x)
            for m in (self.conv4, self.conv5, self.conv6, self.conv7, self.conv8):
                x = add(m._node(x))
            return x

        x0 = add(self.conv0._node(src))
        x1 = add(self.conv1._node(x0))
        x2 = self.conv2._nodes(nodes, x1)
        if self.noScale:
            x3 = self.conv_ext._nodes(nodes, x2)
            return belly(x3), x3, x2, x1, x0
        return None, belly(x2), x2, x1, x0

    def forward(self, x):
        raise L.RcvError("DownSampler is executed as part of PB_FCN's graph; standalone use is not built")


class Classifier(nn.Module):
    """1x1 (kernelSize) classifier conv, optionally behind a max-pool (model.py:255-266).  Only the un-pooled segmentation
    head runs here; the pooled patch-classification head only holds its parameters."""

    def __init__(self, inplanes, num_classes, poolSize=0, kernelSize=1):
        super().__init__()
        self.classifier = nn.Conv2d(inplanes, num_classes, kernel_size=kernelSize, padding=kernelSize // 2)
        self.pool = None
        if poolSize > 1:
            self.pool = nn.MaxPool2d(poolSize)

    def _node(self, src):
        if self.pool is not None:
            raise L.RcvError("the pooled classification head of PB_FCN (classify=1) is outside the segmentation path")
        return {"op": "cls", "src": src, "weight": self.classifier.weight, "bias": self.classifier.bias}

    def forward(self, x):
        raise L.RcvError("Classifier is executed as part of PB_FCN's graph; standalone use is not built")


class PB_FCN(_EngineOwner, nn.Module):
    """The older PB-FCN segmentation net of trainer.py (model.py:269-309): dilated conv->BN->ReLU encoder, three (four with
    noScale) transposed-conv decoder blocks with skip adds, 1x1 segmenter.  classify=1 (patch classification through the
    pooled head) is not built."""

    def __init__(self, planes, num_classes, kernelSize, noScale, classify):
        super().__init__()
        if classify:
            raise NotImplementedError("PB_FCN(classify=1) (patch classification) is outside the segmentation hot path")
        self.noScale = noScale
        self.classify = classify
        self.img_shape = (240, 320) if self.noScale else (120, 160)
        muliplier = 2 if noScale else 1
        outPlanes = planes // 4
        self.FCN = DownSampler(planes, noScale)
        self.up1 = upSampleTransposeConv(planes * 2, planes)
        self.up2 = upSampleTransposeConv(planes, planes // 2 * muliplier)
        self.up3 = upSampleTransposeConv(planes // 2 * muliplier, outPlanes * muliplier)
        self.up4 = upSampleTransposeConv(planes // 2, outPlanes) if noScale else None
        self.classifier = Classifier(planes * 2, num_classes, poolSize=(2 if noScale else 4), kernelSize=kernelSize)
        self.segmenter = Classifier(outPlanes, num_classes, kernelSize=kernelSize)

    # graph of model.py:291-309 (classify == 0)
    def _graph(self):
        nodes: list = []
        f4, f3, f2, f1, f0 = self.FCN._nodes(nodes, ("in", 0))

        def up(layer, x, skip):
            nodes.append(layer._node(x, skip))
            return ("node", len(nodes) - 1)

        if self.noScale:
            x = up(self.up1, f4, f3)
            x = up(self.up2, x, f2)
            x = up(self.up3, x, f1)
            x = up(self.up4, x, f0)
        else:
            x = up(self.up1, f3, f2)
            x = up(self.up2, x, f1)
            x = up(self.up3, x, f0)
        nodes.append(self.segmenter._node(x))
        return {"inputs": [{"layout": "nchw"}], "nodes": nodes}

    def _get_engine(self) -> Engine:
        eng = self.__dict__.get("_engine")
        if eng is None:
            eng = Engine(self._graph(), list(self.parameters()), _bn_modules(self))
            self.__dict__["_engine"] = eng
        return eng

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("PB_FCN expects float32 [B,3,H,W], got %s" % (tuple(x.shape),))
        down = 16 if self.noScale else 8
        if x.shape[2] % down or x.shape[3] % down:
            raise ValueError("H and W must be multiples of %d (got %dx%d)" % (down, x.shape[2], x.shape[3]))
        x = x.to(torch.float32).contiguous()
        return _run_engine(self._get_engine(), self.training, [x])


class PB_FCN_2(ROBO_UNet):
    """trainer.py's v2 net (model.py:416-458, built at trainer.py:126-127): the ROBO-UNet graph with a one-conv Level0 plus a
    pooled patch-classification head whose parameters stay outside the segmentation graph (grad None).  classify=True is not built."""

    def __init__(self, classify, nClass=5, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128):
        nn.Module.__init__(self)
        if classify:
            raise NotImplementedError("PB_FCN_2(classify=True) (patch classification) is outside the segmentation hot path")
        self.classify = classify
        self.numClass = nClass
        self.planes = planes
        self.v2 = False
        self.img_shape = (120, 160)
        maxDepth = planes * pow(2, depth - 1)
        self.downPart = nn.ModuleList()
        self.downPart.add_module("Level0", LevelDown(3, planes, 1, False))
        for i in range(depth - 1):
            nCh = planes * pow(2, i)
            self.downPart.add_module("Level%d" % (i + 1), LevelDown(nCh, nCh * 2, levels, True))
        self.PB = nn.Sequential()
        self.PB.add_module("PB_1", LevelDown(maxDepth, bellyPlanes, bellySize - 1, False))
        self.PB.add_module("PB_2", LevelDown(bellyPlanes, maxDepth, 1, False))
        self.upPart = nn.ModuleList()
        for i in range(depth - 1):
            nCh = planes * pow(2, depth - 1 - i)
            self.upPart.add_module("Up%d" % i, upSampleTransposeConv(nCh, nCh // 2))
        self.classifier = UltClassifier(maxDepth, nClass, True)
        self.segmenter = UltClassifier(planes, nClass, False)
