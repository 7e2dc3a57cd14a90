"""Plan/execute engine behind the nn.Module surface.

A network is described once as a small graph (conv / pool / up / cls nodes, built by model.py from
the module tree).  For every (input shape, train|eval) the engine lowers that graph to two cached
arrays of `rcv_op` records -- one forward, one backward -- whose operands are persistent HBM
buffers owned by the engine.  Running a pass is then ONE call into librcv.so (rcv_run), which
enqueues every kernel of the pass on the current HIP stream: no per-layer Python, no allocation,
no host synchronisation, graph-capturable.

Data layout in HBM:
  * activations fp32 NHWC; the image and the logits stay NCHW (reference callers' layout);
  * per conv block only r = relu(conv+b) is stored; BatchNorm is carried as per-channel
    (scale, shift) constants and applied by whichever kernel loads r next;
  * parameters live in ONE flat fp32 buffer (the nn.Parameters are views into it) and their
    gradients in a second flat buffer of the same layout -- one fused optimizer launch, one
    contiguous all-reduce payload.
"""
from __future__ import annotations

import collections
import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
SIDE_STREAM_WGRAD = not os.environ.get("RCV_NO_SIDE_STREAM")
# "auto" (default): the first backward pass of a plan times itself under three schedules (3 runs each) -- filter gradients, their
# reductions and the bias memsets on the library's second stream ("all"), only the reductions and memsets there ("reduce"), everything on
# the caller's stream ("off") -- and keeps the fastest; "1" / "all" / "reduce" / "off": that schedule, no measurement.
SIDE_STREAM_MODE = os.environ.get("RCV_SIDE_STREAM", "auto")
FUSE_UP_INTO_CLS = not os.environ.get("RCV_NO_FUSED_UP")
# Filter-gradient reductions per batched launch (RCV_OP_WGRAD_REDUCE_BATCH): the reductions of up to this many consecutive layers of the
# backward list run as ONE launch at the position of the last of them (18 launches at the ~5 us floor of a dependent launch -> 3).
# 0 / 1: one launch per layer.
# Data-parallel runs keep 6 (a bucket's gradients are final only behind the launch that reduces them: three launches = three buckets to
# overlap with backward); a single-GPU step on large planes folds ALL reductions into one launch at the end of the list (24 covers every
# network here): measured 6.36 against 6.39 ms on the headline step.  RCV_REDUCE_BATCH overrides both.
REDUCE_BATCH = min(int(os.environ.get("RCV_REDUCE_BATCH", "6")), 24)      # (the job table of one launch holds at most 64 rows)
REDUCE_BATCH_SINGLE = min(int(os.environ.get("RCV_REDUCE_BATCH", "24")), 24)
# Inference on graphs whose BatchNorms all follow their conv directly (relu(bn(conv(x))): ConvPoolSimple model.py:175, ConvPool model.py:140-142,
# upSampleTransposeConv model.py:190-194 -- LabelProp and PB_FCN): the eval-mode BatchNorm is folded into the filter while it is packed
# (w'[co] = w[co]*scale[co]) and into the bias (b*scale + shift), the ReLU runs in the conv's epilogue and the skip add of a decoder
# block rides in its transposed conv's epilogue as a residual: every stored tensor is a final activation, no load transform, no
# RCV_OP_COMBINE launches.  (ROBO_UNet's Conv is bn(relu(conv)), model.py:115-116: the ReLU sits between conv and BatchNorm, nothing folds.)
EVAL_FOLD_BN = not os.environ.get("RCV_NO_EVAL_FOLD")
CLS3_PAD = 8                    # class channels of the 3x3 classifier (v2) are padded to this many NHWC channels
# Backward list order inside a layer.  Default: the layer's filter gradient and its reduction (side stream) are enqueued AHEAD of its
# data-gradient op.  RCV_DGRAD_FIRST=1 enqueues the data gradient (the critical d loss / d activation chain) first so that its kernel is
# dispatched first and the filter gradient fills what it leaves -- measured 0.5 % SLOWER on the headline step (6.36 -> 6.39 ms, three
# interleaved rounds): kept as an experiment switch only.
DGRAD_FIRST = bool(os.environ.get("RCV_DGRAD_FIRST"))
MERGED_TCONV_MAX_COUT = 16      # transposed convs with at most this many output channels use the merged-parity kernel
# Winograd F(2x2,3x3) for the wide stride-1 convs (conv_wino.hip): "auto" = where the library asks for it (>= 64 output and >= 32 input channels and a grid that
# covers the chip), "force" = wherever the kernel can run (tests), "off" = never
WINOGRAD = os.environ.get("RCV_WINOGRAD", "auto")


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def _conv_order(d: dict) -> str:
    order = d.get("order", "relu_bn")
    if d.get("bn") is None:
        if order != "relu":
            raise L.RcvError("a conv node without BatchNorm must be order='relu'")
    elif order not in ("relu_bn", "bn_relu"):
        raise L.RcvError("conv node order '%s' unknown" % order)
    return order


def run_bucketed(marks, numel: int, n_buckets: int, n_ops: int, run_slice, grad_ready, join):
    """The data-parallel backward schedule: run the op list in slices and report finished gradient ranges.

    ``marks`` = [(ops executed, lowest final flat-gradient offset)] (Plan.bwd_marks); ``run_slice(start, end)`` enqueues
    ops [start, end) without joining the filter-gradient stream; ``grad_ready(lo, hi)`` is told that flat.grad[lo:hi] is final
    (descending, contiguous ranges that tile [0, numel)); ``join()`` makes the compute stream wait for the filter-gradient
    stream once at the end.  Engine._run_backward drives the real kernels through this; tests/test_dp_gloo.py drives a
    simulated executor through the very same function."""
    done, hi = 0, numel
    for (end, lo) in select_buckets(marks, numel, n_buckets):
        run_slice(done, end)
        done = end
        if lo < hi:
            grad_ready(lo, hi)
            hi = lo
    if done < n_ops:
        run_slice(done, n_ops)
    if hi > 0:
        grad_ready(0, hi)
    join()


def select_buckets(marks, numel: int, n_buckets: int):
    """Pick <= n_buckets (ops_end, lo_offset) marks so that each bucket carries about numel/n_buckets gradients."""
    if not marks:
        return []
    out, target, hi = [], max(numel // max(n_buckets, 1), 1), numel
    for k, (end, lo) in enumerate(marks):
        last = k == len(marks) - 1
        if (hi - lo >= target and len(out) < n_buckets - 1) or last:
            out.append((end, lo))
            hi = lo
    return out


class Value:
    """A tensor of the graph together with how consumers must read it."""

    def __init__(self, kind: str, buf: Optional[torch.Tensor], C_: int, H: int, W: int, consts=None, producer=None):
        self.kind = kind          # 'plain' | 'affine' | 'affine_relu' | 'nchw'
        self.buf = buf
        self.C, self.H, self.W = C_, H, W
        self.consts = consts      # [5][C] forward constants (scale, shift, -, -, -)
        self.producer = producer  # _Node or None (graph input)
        self.grad: Optional[torch.Tensor] = None       # d loss / d value, NHWC
        self.skip_grad: Optional[torch.Tensor] = None  # extra gradient arriving through a skip connection
        self.needs_grad = False
        self.input_index: Optional[int] = None

    @property
    def load_mode(self) -> int:
        return {"plain": L.LOAD_PLAIN, "affine": L.LOAD_AFFINE, "affine_relu": L.LOAD_AFFINE_RELU, "nchw": L.LOAD_NCHW}[self.kind]      # 'fused_up' values are read by the classifier ops only


class _Node:
    def __init__(self, d: dict, idx: int):
        self.d = d
        self.idx = idx
        self.op = d["op"]
        self.out: Optional[Value] = None
        # per-plan tensors
        self.t: Dict[str, torch.Tensor] = {}


class FlatParams:
    """All parameters in one flat fp32 device buffer; nn.Parameters become views of it."""

    def __init__(self, params: Sequence[torch.nn.Parameter], device: Optional[torch.device] = None):
        self.params = list(params)
        dev = self.params[0].device if self.params else device      # (a parameter-free graph -- a lone max-pool -- lives where its input does)
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += _round_up(p.numel(), 4)          # 16-byte aligned slices
        self.numel = n
        self.data = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                sl = self.data[off:off + p.numel()].view(p.shape)
                sl.copy_(p.data)
                p.data = sl
        self._sig = self.signature()

    def signature(self):
        return tuple(p.data_ptr() for p in self.params)

    def intact(self) -> bool:
        return self.signature() == self._sig

    def index(self, p) -> int:
        for k, q in enumerate(self.params):
            if q is p:
                return k
        raise KeyError("parameter not registered with the engine")

    def grad_view(self, k: int) -> torch.Tensor:
        p, off = self.params[k], self.offsets[k]
        return self.grad[off:off + p.numel()].view(p.shape)

    def grad_ptr(self, p) -> int:
        return self.grad.data_ptr() + 4 * self.offsets[self.index(p)]


class Plan:
    def __init__(self):
        self.fwd: Optional[L.OpList] = None
        self.bwd: Optional[L.OpList] = None
        self.keep: List[object] = []          # tensors referenced by raw pointer
        self.input_slots: List[List[tuple]] = []   # per graph input: [(oplist, op index, slot)]
        self.dlogits_slots: List[tuple] = []
        self.logits: Optional[torch.Tensor] = None
        self.logits_slots: List[tuple] = []   # (fwd op index, slot) of the records that write the logits (inference: a fresh tensor per call)
        self.input_grads: List[Optional[torch.Tensor]] = []
        self.n_head = 0                      # leading ops of fwd that depend on the parameters only (filter repack, eval-mode BN constants)
        self.head_key = None                 # parameter-state key the head was last run for (eval plans)
        self.bwd_marks: List = []            # [(ops executed, lowest final flat-gradient offset)]
        self.reduce_outputs: Dict = {}       # index of a batched reduction launch -> gradient pointers it produces
        self.ce = None                       # lazily built op lists with the cross entropy fused into the classifier ops
        self.side_decided = False            # filter gradients on the second stream: measured on the first backward pass
        self.side_on = True
        self.side_mode = "all"              # one of Engine.SIDE_MODES
        self.side_ms = None                  # backward ms under each of Engine.SIDE_MODES, from that measurement
        self.bytes = 0
        self.pinned = 0                      # > 0: referenced by a captured hipGraph -- never evicted


PLAN_BYTES_BUDGET = 96 << 30      # cached plans beyond this many bytes of engine buffers are dropped, least recently used first


class _Lowering:
    """Lowers one graph for one (input shapes, train|eval) into the two op lists of a Plan: one method per node kind and direction
    (``fwd_conv`` ... ``bwd_conv``), the shared pieces (filter packing table, BatchNorm bookkeeping records, gradient targets) as small
    helpers, and ``finish`` for what spans the lists (batched filter-gradient reductions, gradient-ready marks, the pack / eval head).
    ``eng`` is the Engine that owns the buffers; nothing here runs a kernel."""

    def __init__(self, eng: "Engine", shapes: Sequence[tuple], training: bool):
        self.eng = eng
        self.training = training
        g = eng.graph
        self.plan = Plan()
        self.fl = eng.flat
        self.fwd: List[L.RcvOp] = []
        self.bwd: List[L.RcvOp] = []
        self.N = shapes[0][0]
        # graph inputs
        self.in_vals: List[Value] = []
        self.plan.input_slots = [[] for _ in g["inputs"]]
        for k, (spec, shp) in enumerate(zip(g["inputs"], shapes)):
            if spec["layout"] == "nchw":
                _, c, h, w = shp
                v = Value("nchw", None, c, h, w)
            else:
                _, h, w, c = shp
                v = Value("plain", None, c, h, w)
                v.needs_grad = bool(spec.get("requires_grad")) and training
            if shp[0] != self.N:
                raise L.RcvError("all graph inputs must share the batch size")
            v.input_index = k
            self.in_vals.append(v)
        self.nodes = [_Node(d, i) for i, d in enumerate(g["nodes"])]

        self.jobs: List[L.RcvPackJob] = []          # weight packing table (all layers, one launch per forward)
        self.pre: List[L.RcvOp] = []                # eval mode: running statistics -> constants, ahead of everything else
        self.bn_finalize_flags = L.F_TRAINING if training else 0
        self.batch_at: Dict[int, int] = {}          # index of a folded reduction -> index of the launch that now carries it
        self._late: List[tuple] = []                # side-stream ops of the node being lowered, emitted behind its data-gradient op
        bn_nodes = [d for d in g["nodes"] if d.get("bn") is not None]
        self.fold = (EVAL_FOLD_BN and not training and bool(bn_nodes) and not any(d.get("concat") for d in g["nodes"])
                     and all(d["op"] == "up" or _conv_order(d) == "bn_relu" for d in bn_nodes))

    def ref(self, r) -> Value:
        return self.in_vals[r[1]] if r[0] == "in" else self.nodes[r[1]].out

    def bind_in(self, op_list: List[L.RcvOp], v: Value, slot: int):
        """Operand `slot` of the op about to be appended reads value v (patched per call for inputs)."""
        if v.input_index is not None:
            self.plan.input_slots[v.input_index].append((op_list is self.bwd, len(op_list), slot))
            return 0
        return v.buf.data_ptr()

    def add_pack(self, param, D0, D1, rows_from_d1, flip, merged=False, wino=0, scale: Optional[torch.Tensor] = None):
        """wino: the layout rcv_op_filter_layout asked for (0 plain, 2 Winograd, 3 split-bf16: three bf16 per value, rows padded to 32)."""
        wino = int(wino)
        merged = bool(merged) or wino == 4
        rows = D1 if rows_from_d1 else D0
        cols = D0 if rows_from_d1 else D1
        split = wino in (3, 4, 5)   # three bf16 per value, k = tap * rows + row in steps of 32 (zero beyond the last tap); 5: 16-row chunks, 5 steps each
        rp, cp = _round_up(rows, (16 if wino == 5 else (32 if rows > 32 else 8)) if split else 4), _round_up(cols * (4 if merged else 1), 16)
        taps = 16 if wino == 2 else (4 if merged else 9)
        ksteps = (rp // 16) * 5 if wino == 5 else (taps * rp + 31) // 32
        n_floats = 3 * ksteps * cp * 16 if split else taps * rp * cp
        dst = self.eng._zeros(self.plan, n_floats)
        j = L.RcvPackJob()
        j.src, j.dst, j.D0, j.D1 = param.data_ptr(), dst.data_ptr(), D0, D1
        j.rows_from_d1, j.flip, j.rows_pad, j.cols_pad = int(rows_from_d1), int(flip), rp, cp
        j.merged = wino if wino else int(merged)
        j.scale = scale.data_ptr() if scale is not None else None      # per output channel (inference BatchNorm folding)
        self.jobs.append(j)
        return dst

    def wants_winograd(self, op: L.RcvOp) -> int:
        """Ask the library which filter layout this conv record should get (0 plain, 2 Winograd, 3 split-bf16) and mark the record."""
        layout = L.op_filter_layout(self.eng.handle, op, WINOGRAD == "force")
        if layout == 2 and WINOGRAD == "off":
            layout = 0
        if layout:
            op.i[L.RCV_I_AUX0] = layout
        return layout

    def tconv_layout(self, op: L.RcvOp) -> int:
        """The same question for a transposed-conv record whose filter would be packed in the merged layout (i[AUX0] = 1): 4 = the library
        runs it on the split-bf16 narrow kernel and wants the merged layout split into bf16 (the record is marked)."""
        if op.i[L.RCV_I_AUX0] == 1 and L.op_filter_layout(self.eng.handle, op, False) == 4:
            op.i[L.RCV_I_AUX0] = 4
            return 4
        return 0

    def side(self, op: L.RcvOp, input_slot: Optional[tuple] = None):
        """A filter-gradient-side op of the node being lowered (filter gradient, its reduction, a bias memset).  ``input_slot`` =
        (graph input index, operand slot) when the op reads a graph input: the patch index is the op's FINAL position in the list."""
        self._late.append((op, input_slot))
        if not DGRAD_FIRST:
            self.flush_side()

    def flush_side(self):
        for op, slot in self._late:
            if slot is not None:
                self.plan.input_slots[slot[0]].append((True, len(self.bwd), slot[1]))
            self.bwd.append(op)
        self._late = []

    def use_merged(self, cout: int) -> bool:
        # narrow transposed convs are HBM bound: one pass writing whole output rows beats four parity passes
        return cout <= MERGED_TCONV_MAX_COUT

    def bn_tensors(self, node: _Node, Cc: int):
        node.t["consts"] = self.eng._zeros(self.plan, 5, Cc)
        node.t["mean"] = self.eng._zeros(self.plan, Cc)
        node.t["istd"] = self.eng._zeros(self.plan, Cc)

    def emit_bn_forward(self, node: _Node, bn, conv_op: L.RcvOp, Cc: int, Ho: int, Wo: int, conv_bias=None):
        """Statistics partials of conv_op -> constants (training) or running stats -> constants (eval)."""
        if self.training:
            if self.N * Ho * Wo <= 1:      # same refusal (and text) as torch.nn.functional.batch_norm, which the reference runs
                raise ValueError("Expected more than 1 value per channel when training, got input size %s" % ((self.N, Cc, Ho, Wo),))
            conv_op.i[L.RCV_I_STATS] = L.STATS_FWD
            self.eng._workspace(self.plan, conv_op)
            self.fwd.append(conv_op)
            self.fwd.append(L.make_op(L.OP_BN_FINALIZE, self.bn_finalize_flags, n=self.N, ho=Ho, wo=Wo, cout=Cc,
                                 npart=conv_op.i[L.RCV_I_NPART], f0=BN_MOMENTUM, f1=BN_EPS,
                                 p_part=conv_op.p[L.RCV_P_PART], p_out=node.t["consts"].data_ptr(),
                                 p_x0=bn.weight.data_ptr(), p_x1=bn.bias.data_ptr(),
                                 p_x2=bn.running_mean.data_ptr(), p_x3=bn.running_var.data_ptr(),
                                 p_x4=node.t["mean"].data_ptr(), p_x5=node.t["istd"].data_ptr()))
        else:
            self.pre.append(L.make_op(L.OP_BN_EVAL, 0, cout=Cc, f1=BN_EPS, p_out=node.t["consts"].data_ptr(),
                                 p_x0=bn.weight.data_ptr(), p_x1=bn.bias.data_ptr(),
                                 p_x2=bn.running_mean.data_ptr(), p_x3=bn.running_var.data_ptr(), p_x4=_ptr(conv_bias)))
            self.fwd.append(conv_op)

    def only_consumer_is_cls1x1(self, idx: int) -> bool:
        users = [nd for nd in self.nodes if any(nd.d.get(k) == ("node", idx) for k in ("src", "skip", "add"))]
        return (len(users) == 1 and users[0].op == "cls" and tuple(users[0].d["weight"].shape[2:]) == (1, 1)
                and 1 <= users[0].d["weight"].shape[0] <= 8)

    def fwd_conv(self, node: _Node):
        d = node.d
        src = self.ref(d["src"])
        w, b, bn = d["weight"], d.get("bias"), d.get("bn")
        Cout, Cin = w.shape[0], w.shape[1]
        s, dil = d["stride"], d["dil"]
        if Cin != src.C:
            raise L.RcvError("conv node %d: input has %d channels, weight expects %d" % (node.idx, src.C, Cin))
        Ho, Wo = (src.H - 1) // s + 1, (src.W - 1) // s + 1
        r = self.eng._alloc(self.plan, self.N, Ho, Wo, Cout)
        # order: 'relu_bn' = bn(relu(conv)) (Conv, model.py:115-116); 'bn_relu' = relu(bn(conv)) (ConvPoolSimple,
        # model.py:175; the strided half of ConvPool, model.py:140-142); 'relu' = relu(conv), no BatchNorm (model.py:138-139)
        order = _conv_order(d)
        if self.fold and bn is not None:
            # inference, relu(bn(conv(x))): filter and bias carry the BatchNorm, the epilogue the ReLU; r is the block's final output
            self.bn_tensors(node, Cout)
            cst = node.t["consts"]
            op = L.make_op(L.OP_CONV, L.F_BIAS | L.F_RELU, n=self.N, h=src.H, w=src.W, cin=Cin, cout=Cout, ho=Ho, wo=Wo, stride=s, dil=dil,
                           inmode=src.load_mode, p_in_c=_ptr(src.consts), p_bias=cst.data_ptr() + 4 * 3 * Cout, p_out=r.data_ptr())
            node.t["wp"] = self.add_pack(w, Cout, Cin, True, False, wino=self.wants_winograd(op), scale=cst)      # row 0 = scale
            op.p[L.RCV_P_W] = node.t["wp"].data_ptr()
            op.p[L.RCV_P_IN] = self.bind_in(self.fwd, src, L.RCV_P_IN) or None
            self.emit_bn_forward(node, bn, op, Cout, Ho, Wo, conv_bias=b)
            node.out = Value("plain", r, Cout, Ho, Wo, None, node)
            return
        flags = (L.F_BIAS if b is not None else 0) | (L.F_RELU if order != "bn_relu" else 0)
        op = L.make_op(L.OP_CONV, flags, n=self.N, h=src.H, w=src.W, cin=Cin, cout=Cout, ho=Ho, wo=Wo, stride=s, dil=dil,
                       inmode=src.load_mode, p_in_c=_ptr(src.consts),
                       p_bias=_ptr(b), p_out=r.data_ptr())
        node.t["wp"] = self.add_pack(w, Cout, Cin, True, False, wino=self.wants_winograd(op))
        op.p[L.RCV_P_W] = node.t["wp"].data_ptr()
        op.p[L.RCV_P_IN] = self.bind_in(self.fwd, src, L.RCV_P_IN) or None
        if bn is not None:
            self.bn_tensors(node, Cout)
            self.emit_bn_forward(node, bn, op, Cout, Ho, Wo)
            node.out = Value("affine" if order == "relu_bn" else "affine_relu", r, Cout, Ho, Wo, node.t["consts"], node)
        else:
            self.fwd.append(op)
            node.out = Value("plain", r, Cout, Ho, Wo, None, node)

    def fwd_pool(self, node: _Node):
        d = node.d
        src = self.ref(d["src"])
        if src.H % 2 or src.W % 2:
            raise L.RcvError("max-pool needs even spatial dims, got %dx%d" % (src.H, src.W))
        out = self.eng._alloc(self.plan, self.N, src.H // 2, src.W // 2, src.C)
        op = L.make_op(L.OP_POOL_FWD, 0, n=self.N, h=src.H, w=src.W, cout=src.C, inmode=src.load_mode,
                       p_in_c=_ptr(src.consts), p_out=out.data_ptr())
        op.p[L.RCV_P_IN] = self.bind_in(self.fwd, src, L.RCV_P_IN) or None
        self.fwd.append(op)
        node.out = Value("plain", out, src.C, src.H // 2, src.W // 2, None, node)
        if node.idx == len(self.nodes) - 1:      # a graph that ENDS in the pool (standalone Pool module, model.py:92-100): its output is the result
            self.plan.logits = out
            self.plan.logits_slots = [(len(self.fwd) - 1, L.RCV_P_OUT)]

    def fwd_up(self, node: _Node):
        d = node.d
        src = self.ref(d["src"])
        w, b, bn = d["weight"], d.get("bias"), d["bn"]
        Cin, Cout = w.shape[0], w.shape[1]
        if Cin != src.C:
            raise L.RcvError("up node %d: input has %d channels, weight expects %d" % (node.idx, src.C, Cin))
        Ho, Wo = 2 * src.H, 2 * src.W
        skip = self.ref(d["skip"]) if d.get("skip") is not None else None
        if (self.fold and skip is not None and (skip.C, skip.H, skip.W) == (Cout, Ho, Wo) and skip.kind == "plain" and skip.input_index is None):
            # inference: relu(bn(convT(x) + b)) + skip in ONE launch -- BatchNorm in filter and bias, ReLU, then the skip tensor (a final
            # activation itself: its block folded too) as the epilogue's residual.  (Without a skip there is no launch to save: the block's
            # consumer applies relu(bn(.)) as its load transform, and LabelProp's tail fuses exactly that form into the classifier.)
            self.bn_tensors(node, Cout)
            cst = node.t["consts"]
            t = self.eng._alloc(self.plan, self.N, Ho, Wo, Cout)
            op = L.make_op(L.OP_TCONV, L.F_BIAS | L.F_RELU | L.F_RESID, n=self.N, h=src.H, w=src.W, cin=Cin,
                           cout=Cout, ho=Ho, wo=Wo, stride=2, dil=1, aux0=int(self.use_merged(Cout)), inmode=src.load_mode,
                           p_in_c=_ptr(src.consts), p_bias=cst.data_ptr() + 4 * 3 * Cout, p_out=t.data_ptr(),
                           p_resid=skip.buf.data_ptr())
            node.t["wp"] = self.add_pack(w, Cin, Cout, False, False, merged=self.use_merged(Cout), wino=self.tconv_layout(op), scale=cst)
            op.p[L.RCV_P_W] = node.t["wp"].data_ptr()
            op.p[L.RCV_P_IN] = self.bind_in(self.fwd, src, L.RCV_P_IN) or None
            self.emit_bn_forward(node, bn, op, Cout, Ho, Wo, conv_bias=b)
            node.t["t"] = t
            node.out = Value("plain", t, Cout, Ho, Wo, None, node)
            return
        t = self.eng._alloc(self.plan, self.N, Ho, Wo, Cout)
        self.bn_tensors(node, Cout)
        op = L.make_op(L.OP_TCONV, (L.F_BIAS if b is not None else 0), n=self.N, h=src.H, w=src.W, cin=Cin, cout=Cout,
                       ho=Ho, wo=Wo, stride=2, dil=1, aux0=int(self.use_merged(Cout)), inmode=src.load_mode, p_in_c=_ptr(src.consts),
                       p_bias=_ptr(b), p_out=t.data_ptr())
        node.t["wp"] = self.add_pack(w, Cin, Cout, False, False, merged=self.use_merged(Cout), wino=self.tconv_layout(op))
        op.p[L.RCV_P_W] = node.t["wp"].data_ptr()
        op.p[L.RCV_P_IN] = self.bind_in(self.fwd, src, L.RCV_P_IN) or None
        self.emit_bn_forward(node, bn, op, Cout, Ho, Wo)
        node.t["t"] = t
        if d.get("skip") is not None:
            skip = self.ref(d["skip"])
            if (skip.C, skip.H, skip.W) != (Cout, Ho, Wo):
                raise L.RcvError("up node %d: skip tensor %s does not match output %s" %
                                 (node.idx, (skip.C, skip.H, skip.W), (Cout, Ho, Wo)))
            concat = bool(d.get("concat"))      # v2: torch.cat([layer(up), skip], 1) instead of the add (model.py:507)
            if not concat and FUSE_UP_INTO_CLS and Cout == 8 and skip.input_index is None and self.only_consumer_is_cls1x1(node.idx):
                # the 1x1 classifier forms relu(bn(t)) + bn(skip) itself (RCV_F_FUSED_UP): no RCV_OP_COMBINE, `up` never exists
                node.out = Value("fused_up", None, Cout, Ho, Wo, node.t["consts"], node)
                node.out.fused = (t, skip)
                return
            Cup = 2 * Cout if concat else Cout
            up = self.eng._alloc(self.plan, self.N, Ho, Wo, Cup)
            cop = L.make_op(L.OP_COMBINE, L.F_CONCAT if concat else 0, n=self.N, h=Ho, w=Wo, cout=Cout, inmode2=skip.load_mode,
                            p_in=t.data_ptr(), p_in_c=node.t["consts"].data_ptr(), p_in2_c=_ptr(skip.consts), p_out=up.data_ptr())
            cop.p[L.RCV_P_IN2] = self.bind_in(self.fwd, skip, L.RCV_P_IN2) or None
            self.fwd.append(cop)
            node.out = Value("plain", up, Cup, Ho, Wo, None, node)
        else:
            node.out = Value("affine_relu", t, Cout, Ho, Wo, node.t["consts"], node)

    def fwd_cls(self, node: _Node):
        d = node.d
        src = self.ref(d["src"])
        w, b = d["weight"], d.get("bias")
        Cout, Cin = w.shape[0], w.shape[1]
        if src.kind not in ("plain", "fused_up"):
            raise L.RcvError("classifier input must be a materialised tensor")
        if Cin != src.C:
            raise L.RcvError("classifier: input has %d channels, weight expects %d" % (src.C, Cin))
        logits = self.eng._alloc(self.plan, self.N, Cout, src.H, src.W)
        if tuple(w.shape[2:]) == (1, 1) and src.kind == "fused_up":
            tt, skip = src.fused
            self.fwd.append(L.make_op(L.OP_CLS_FWD, L.F_FUSED_UP, n=self.N, h=src.H, w=src.W, cin=Cin, cout=Cout, aux0=skip.load_mode,
                                 aux1=getattr(src, "fused_rch", 0),
                                 p_in=tt.data_ptr(), p_in_c=src.consts.data_ptr(), p_x3=skip.buf.data_ptr(), p_x4=_ptr(skip.consts),
                                 p_w=w.data_ptr(), p_bias=_ptr(b), p_out=logits.data_ptr()))
        elif tuple(w.shape[2:]) == (1, 1):
            op = L.make_op(L.OP_CLS_FWD, 0, n=self.N, h=src.H, w=src.W, cin=Cin, cout=Cout, p_w=w.data_ptr(), p_bias=_ptr(b),
                           p_out=logits.data_ptr())
            op.p[L.RCV_P_IN] = self.bind_in(self.fwd, src, L.RCV_P_IN) or None
            self.fwd.append(op)
        elif tuple(w.shape[2:]) == (3, 3) and Cout <= CLS3_PAD:
            # v2 (classSize=3): the MFMA conv with the class channels padded to 8, NHWC; then bias + NHWC -> NCHW
            node.t["wp"] = self.add_pack(w, Cout, Cin, True, False)
            node.t["z"] = self.eng._alloc(self.plan, self.N, src.H, src.W, CLS3_PAD)
            op = L.make_op(L.OP_CONV, 0, n=self.N, h=src.H, w=src.W, cin=Cin, cout=CLS3_PAD, ho=src.H, wo=src.W, stride=1, dil=1,
                           inmode=src.load_mode, p_w=node.t["wp"].data_ptr(), p_out=node.t["z"].data_ptr())
            op.p[L.RCV_P_IN] = self.bind_in(self.fwd, src, L.RCV_P_IN) or None
            self.fwd.append(op)
            self.fwd.append(L.make_op(L.OP_NHWC_TO_NCHW, 0, n=self.N, h=src.H, w=src.W, cin=CLS3_PAD, cout=Cout, p_in=node.t["z"].data_ptr(),
                                 p_bias=_ptr(b), p_out=logits.data_ptr()))
        else:
            raise L.RcvError("classifier kernels %s with %d classes are not built (1x1, or 3x3 with <= %d classes)"
                             % (tuple(w.shape[2:]), Cout, CLS3_PAD))
        node.out = Value("plain", logits, Cout, src.H, src.W, None, node)
        self.plan.logits = logits
        assert self.fwd[-1].p[L.RCV_P_OUT] == logits.data_ptr()
        self.plan.logits_slots = [(len(self.fwd) - 1, L.RCV_P_OUT)]

    def fwd_add_slice(self, node: _Node):
        d = node.d
        # out = value(src); out[..., 0:Ca] += value(add)      (LabelProp tail, model.py:565)
        src, add = self.ref(d["src"]), self.ref(d["add"])
        if (add.H, add.W) != (src.H, src.W) or add.C > src.C:
            raise L.RcvError("add_slice: operand shapes do not match")
        if (FUSE_UP_INTO_CLS and not self.training and src.kind == "affine_relu" and src.C == 16 and add.C % 4 == 0 and add.buf is not None
                and add.input_index is None and src.input_index is None and add.kind in ("plain", "affine", "affine_relu")
                and self.only_consumer_is_cls1x1(node.idx)):
            # LabelProp's tail (model.py:563-567): the 1x1 classifier forms relu(bn(t)) and adds the skip to its first add.C input
            # channels itself (RCV_F_FUSED_UP with i[RCV_I_AUX1] = add.C): no RCV_OP_MATERIALIZE / RCV_OP_ADD_SLICE passes
            node.out = Value("fused_up", None, src.C, src.H, src.W, src.consts, node)
            node.out.fused = (src.buf, add)
            node.out.fused_rch = add.C
            return
        out = self.eng._alloc(self.plan, self.N, src.H, src.W, src.C)
        op = L.make_op(L.OP_MATERIALIZE, 0, n=self.N, h=src.H, w=src.W, cout=src.C, inmode=src.load_mode,
                       p_in_c=_ptr(src.consts), p_out=out.data_ptr())
        op.p[L.RCV_P_IN] = self.bind_in(self.fwd, src, L.RCV_P_IN) or None
        self.fwd.append(op)
        op2 = L.make_op(L.OP_ADD_SLICE, 0, n=self.N, h=src.H, w=src.W, cin=add.C, cout=src.C, inmode=add.load_mode,
                        p_in_c=_ptr(add.consts), p_out=out.data_ptr())
        op2.p[L.RCV_P_IN] = self.bind_in(self.fwd, add, L.RCV_P_IN) or None
        self.fwd.append(op2)
        node.out = Value("plain", out, src.C, src.H, src.W, None, node)

    def fwd_mat(self, node: _Node):
        d = node.d
        src = self.ref(d["src"])
        out = self.eng._alloc(self.plan, self.N, src.H, src.W, src.C)
        op = L.make_op(L.OP_MATERIALIZE, 0, n=self.N, h=src.H, w=src.W, cout=src.C, inmode=src.load_mode,
                       p_in_c=_ptr(src.consts), p_out=out.data_ptr())
        op.p[L.RCV_P_IN] = self.bind_in(self.fwd, src, L.RCV_P_IN) or None
        self.fwd.append(op)
        node.out = Value("plain", out, src.C, src.H, src.W, None, node)
        self.plan.logits = out
        self.plan.logits_slots = [(len(self.fwd) - 1, L.RCV_P_OUT)]

    def grad_target(self, v: Value, writer_op: L.RcvOp, Ho: int, Wo: int):
        """Configure writer_op (a dgrad-like op) to produce d loss / d v with the epilogue v's producer needs."""
        v.grad = self.eng._alloc(self.plan, self.N, v.H, v.W, v.C)
        writer_op.p[L.RCV_P_OUT] = v.grad.data_ptr()
        if v.skip_grad is not None:
            writer_op.flags |= L.F_RESID
            writer_op.p[L.RCV_P_RESID] = v.skip_grad.data_ptr()
        prod = v.producer
        if prod is None:
            self.plan.input_grads[v.input_index] = v.grad
            writer_op.i[L.RCV_I_STATS] = L.STATS_NONE
        elif prod.op == "conv" and prod.d.get("bn") is not None:
            writer_op.i[L.RCV_I_STATS] = L.STATS_BWD_ENC if _conv_order(prod.d) == "relu_bn" else L.STATS_BWD_DEC
            writer_op.p[L.RCV_P_EPI_AUX] = v.buf.data_ptr()
            writer_op.p[L.RCV_P_EPI_C] = prod.t["consts"].data_ptr()     # row 2 = batch mean
        elif prod.op == "up" and not prod.d.get("concat"):
            writer_op.i[L.RCV_I_STATS] = L.STATS_BWD_DEC
            writer_op.p[L.RCV_P_EPI_AUX] = prod.t["t"].data_ptr()
            writer_op.p[L.RCV_P_EPI_C] = prod.t["consts"].data_ptr()
        else:
            writer_op.i[L.RCV_I_STATS] = L.STATS_NONE
        self.eng._workspace(self.plan, writer_op)
        if writer_op.i[L.RCV_I_STATS] != L.STATS_NONE:
            prod.t["bwd_part"] = (writer_op.p[L.RCV_P_PART], writer_op.i[L.RCV_I_NPART])

    def emit_bn_backward(self, node: _Node, bn, Cc: int, Ho: int, Wo: int):
        part_ptr, n_part = node.t["bwd_part"]
        node.t["bconsts"] = self.eng._zeros(self.plan, 5, Cc)
        self.bwd.append(L.make_op(L.OP_BN_BWD, 0, n=self.N, ho=Ho, wo=Wo, cout=Cc, npart=n_part, p_part=part_ptr,
                             p_out=node.t["bconsts"].data_ptr(), p_in_c=node.t["consts"].data_ptr(),
                             p_x0=bn.weight.data_ptr(), p_x1=self.fl.grad_ptr(bn.weight), p_x2=self.fl.grad_ptr(bn.bias),
                             p_x4=node.t["mean"].data_ptr(), p_x5=node.t["istd"].data_ptr()))

    def node_params(self, nd):
        ps = [nd.d.get("weight"), nd.d.get("bias")]
        bn_ = nd.d.get("bn")
        if bn_ is not None:
            ps += [bn_.weight, bn_.bias]
        return [q for q in ps if q is not None]

    def bwd_cls(self, node: _Node):
        d = node.d
        src = self.ref(d["src"])
        w, b = d["weight"], d.get("bias")
        Cout, Cin = w.shape[0], w.shape[1]
        if "z" in node.t:      # 3x3 classifier: NCHW dlogits -> padded NHWC, then the ordinary filter / data gradients
            if src.input_index is not None:
                raise L.RcvError("the 3x3 classifier cannot read a graph input directly")
            g8 = self.eng._alloc(self.plan, self.N, src.H, src.W, CLS3_PAD)
            self.plan.dlogits_slots.append((len(self.bwd), L.RCV_P_IN))
            self.bwd.append(L.make_op(L.OP_NCHW_TO_NHWC, 0, n=self.N, h=src.H, w=src.W, cin=Cout, cout=CLS3_PAD, p_out=g8.data_ptr()))
            wop = L.make_op(L.OP_WGRAD, (L.F_BIAS if b is not None else 0), n=self.N, h=src.H, w=src.W, cin=Cin, ho=src.H, wo=src.W,
                            cout=CLS3_PAD, stride=1, dil=1, inmode=src.load_mode, inmode2=L.LOAD_PLAIN,
                            p_in=src.buf.data_ptr(), p_in_c=_ptr(src.consts), p_in2=g8.data_ptr())
            self.eng._workspace(self.plan, wop)
            self.bwd.append(wop)
            self.bwd.append(L.make_op(L.OP_WGRAD_REDUCE, 0, cin=Cin, cout=Cout, nsplit=wop.i[L.RCV_I_NSPLIT],
                                 p_part=wop.p[L.RCV_P_PART], p_out=self.fl.grad_ptr(w),
                                 p_bias=(self.fl.grad_ptr(b) if b is not None else 0)))
            if src.needs_grad:
                node.t["wd"] = self.add_pack(w, Cout, Cin, False, True)
                dop = L.make_op(L.OP_CONV, 0, n=self.N, h=src.H, w=src.W, cin=CLS3_PAD, cout=Cin, ho=src.H, wo=src.W, stride=1, dil=1,
                                inmode=L.LOAD_PLAIN, p_in=g8.data_ptr(), p_w=node.t["wd"].data_ptr())
                self.grad_target(src, dop, src.H, src.W)
                self.bwd.append(dop)
            return
        op = L.make_op(L.OP_CLS_BWD, 0, n=self.N, h=src.H, w=src.W, cin=Cin, cout=Cout, p_in=src.buf.data_ptr() if src.buf is not None else 0,
                       p_w=w.data_ptr(), p_x1=self.fl.grad_ptr(w), p_x2=(self.fl.grad_ptr(b) if b is not None else 0))
        if src.kind == "fused_up":
            _tt, skip = src.fused
            op.flags |= L.F_FUSED_UP
            op.i[L.RCV_I_AUX0] = skip.load_mode
            op.p[L.RCV_P_X3] = skip.buf.data_ptr()
            op.p[L.RCV_P_X4] = _ptr(skip.consts) or None
        if src.input_index is not None:
            self.plan.input_slots[src.input_index].append((True, len(self.bwd), L.RCV_P_IN))
        self.plan.dlogits_slots.append((len(self.bwd), L.RCV_P_IN2))
        self.grad_target(src, op, src.H, src.W)
        self.bwd.append(op)

    def bwd_mat(self, node: _Node):
        d = node.d
        # the gradient of the materialised output arrives from outside: copy + BN-backward sums
        src = self.ref(d["src"])
        op = L.make_op(L.OP_BWD_STATS, 0, n=self.N, h=src.H, w=src.W, cout=src.C)
        self.plan.dlogits_slots.append((len(self.bwd), L.RCV_P_IN))
        self.grad_target(src, op, src.H, src.W)
        if op.i[L.RCV_I_STATS] == L.STATS_NONE:
            raise L.RcvError("a materialised output must follow a conv or up block")
        self.bwd.append(op)

    def bwd_up(self, node: _Node):
        d = node.d
        out, src = node.out, self.ref(d["src"])
        w, b, bn = d["weight"], d.get("bias"), d["bn"]
        Cin, Cout = w.shape[0], w.shape[1]
        if out.grad is None:
            raise L.RcvError("up node %d has no consumer that produces its gradient" % node.idx)
        gout = out.grad
        if d.get("skip") is not None and d.get("concat"):
            # gradient of the concatenation: channels [0,C) belong to this block (copied out together with its
            # BatchNorm-backward sums), channels [C,2C) to the skip tensor
            gout = self.eng._alloc(self.plan, self.N, out.H, out.W, Cout)
            sop = L.make_op(L.OP_BWD_STATS, 0, n=self.N, h=out.H, w=out.W, cin=2 * Cout, cout=Cout, aux0=0, stats=L.STATS_BWD_DEC,
                            p_in=out.grad.data_ptr(), p_epi_aux=node.t["t"].data_ptr(), p_epi_c=node.t["consts"].data_ptr(),
                            p_out=gout.data_ptr())
            self.eng._workspace(self.plan, sop)
            node.t["bwd_part"] = (sop.p[L.RCV_P_PART], sop.i[L.RCV_I_NPART])
            self.bwd.append(sop)
            gskip = self.eng._alloc(self.plan, self.N, out.H, out.W, Cout)
            self.bwd.append(L.make_op(L.OP_BWD_STATS, 0, n=self.N, h=out.H, w=out.W, cin=2 * Cout, cout=Cout, aux0=Cout, stats=L.STATS_NONE,
                                 p_in=out.grad.data_ptr(), p_out=gskip.data_ptr()))
            self.ref(d["skip"]).skip_grad = gskip
        elif d.get("skip") is not None:
            self.ref(d["skip"]).skip_grad = out.grad        # d up / d skip = identity
        self.emit_bn_backward(node, bn, Cout, out.H, out.W)
        # filter gradient: G = dt (2x plane), P = layer input
        wop = L.make_op(L.OP_WGRAD, 0, n=self.N, h=out.H, w=out.W, cin=Cout, ho=src.H, wo=src.W, cout=Cin, stride=2, dil=1,
                        inmode=L.LOAD_GRAD_DEC, inmode2=src.load_mode, p_in=gout.data_ptr(),
                        p_in_aux=node.t["t"].data_ptr(), p_in_c=node.t["bconsts"].data_ptr(),
                        p_in2_c=_ptr(src.consts))
        wop.p[L.RCV_P_IN2] = (src.buf.data_ptr() if src.buf is not None else None)
        self.eng._workspace(self.plan, wop)
        self.side(wop, (src.input_index, L.RCV_P_IN2) if src.input_index is not None else None)
        self.side(L.make_op(L.OP_WGRAD_REDUCE, 0, cin=Cout, cout=Cin, nsplit=wop.i[L.RCV_I_NSPLIT],
                            p_part=wop.p[L.RCV_P_PART], p_out=self.fl.grad_ptr(w)))
        if b is not None:   # bias ahead of a BatchNorm: gradient is identically zero
            self.side(L.make_op(L.OP_MEMSET, 0, count=b.numel(), p_out=self.fl.grad_ptr(b)))
        if src.needs_grad:
            dop = L.make_op(L.OP_CONV, 0, n=self.N, h=out.H, w=out.W, cin=Cout, cout=Cin, ho=src.H, wo=src.W, stride=2, dil=1,
                            inmode=L.LOAD_GRAD_DEC, p_in=gout.data_ptr(), p_in_aux=node.t["t"].data_ptr(),
                            p_in_c=node.t["bconsts"].data_ptr())
            lay = self.wants_winograd(dop)
            node.t["wd"] = self.add_pack(w, Cin, Cout, True, False, wino=(lay if lay in (3, 4, 5) else 0))
            if lay == 2:
                dop.i[L.RCV_I_AUX0] = 0
            dop.p[L.RCV_P_W] = node.t["wd"].data_ptr()
            self.grad_target(src, dop, src.H, src.W)
            self.bwd.append(dop)
        self.flush_side()

    def bwd_pool(self, node: _Node):
        d = node.d
        out, src = node.out, self.ref(d["src"])
        last = node.idx == len(self.nodes) - 1      # the graph's result: its gradient arrives from outside (patched per call)
        if out.grad is None and not last:
            raise L.RcvError("pool node %d has no gradient producer" % node.idx)
        if src.needs_grad:
            pop = L.make_op(L.OP_POOL_BWD, 0, n=self.N, h=src.H, w=src.W, cout=src.C, inmode=src.load_mode,
                            p_in=(None if last else out.grad.data_ptr()), p_in_c=_ptr(src.consts))
            if last:
                self.plan.dlogits_slots.append((len(self.bwd), L.RCV_P_IN))
            self.grad_target(src, pop, src.H, src.W)
            # pool backward recomputes the arg-max from the producer's r; for non-BN producers use the tensor itself
            pop.p[L.RCV_P_EPI_AUX] = src.buf.data_ptr() if src.buf is not None else None
            if src.input_index is not None:
                self.plan.input_slots[src.input_index].append((True, len(self.bwd), L.RCV_P_EPI_AUX))
            if src.producer is not None and src.producer.op == "up":
                raise L.RcvError("max-pool directly after a decoder block is not supported")
            self.bwd.append(pop)

    def bwd_conv(self, node: _Node):
        d = node.d
        out, src = node.out, self.ref(d["src"])
        w, b, bn = d["weight"], d.get("bias"), d.get("bn")
        Cout, Cin = w.shape[0], w.shape[1]
        s, dil = d["stride"], d["dil"]
        order = _conv_order(d)
        if out.grad is None:
            raise L.RcvError("conv node %d has no gradient producer" % node.idx)
        if bn is not None:
            self.emit_bn_backward(node, bn, Cout, out.H, out.W)
        else:       # relu(conv): the gradient load is the ReLU mask alone, v = r > 0 ? 1*g + 0 + 0*r : 0
            ones = self.eng._zeros(self.plan, 5, Cout)
            ones[0].fill_(1.0)
            node.t["bconsts"] = ones
        gmode = L.LOAD_GRAD_DEC if order == "bn_relu" else L.LOAD_GRAD_ENC
        bias_grad = b is not None and order != "bn_relu"
        wop = L.make_op(L.OP_WGRAD, (L.F_BIAS if bias_grad else 0), n=self.N, h=src.H, w=src.W, cin=Cin, ho=out.H, wo=out.W,
                        cout=Cout, stride=s, dil=dil, inmode=src.load_mode, inmode2=gmode,
                        p_in_c=_ptr(src.consts), p_in2=out.grad.data_ptr(), p_in2_aux=out.buf.data_ptr(),
                        p_in2_c=node.t["bconsts"].data_ptr())
        wop.p[L.RCV_P_IN] = (src.buf.data_ptr() if src.buf is not None else None)
        self.eng._workspace(self.plan, wop)
        self.side(wop, (src.input_index, L.RCV_P_IN) if src.input_index is not None else None)
        self.side(L.make_op(L.OP_WGRAD_REDUCE, 0, cin=Cin, cout=Cout, nsplit=wop.i[L.RCV_I_NSPLIT],
                            p_part=wop.p[L.RCV_P_PART], p_out=self.fl.grad_ptr(w),
                            p_bias=(self.fl.grad_ptr(b) if bias_grad else 0)))
        if b is not None and not bias_grad:   # bias ahead of a BatchNorm: gradient is identically zero
            self.side(L.make_op(L.OP_MEMSET, 0, count=b.numel(), p_out=self.fl.grad_ptr(b)))
        if src.needs_grad:
            if s == 1:
                dop = L.make_op(L.OP_CONV, 0, n=self.N, h=out.H, w=out.W, cin=Cout, cout=Cin, ho=src.H, wo=src.W, stride=1, dil=dil,
                                inmode=gmode, p_in=out.grad.data_ptr(), p_in_aux=out.buf.data_ptr(),
                                p_in_c=node.t["bconsts"].data_ptr())
                node.t["wd"] = self.add_pack(w, Cout, Cin, False, True, wino=self.wants_winograd(dop))
                dop.p[L.RCV_P_W] = node.t["wd"].data_ptr()
            else:
                if src.H != 2 * out.H or src.W != 2 * out.W:
                    raise L.RcvError("stride-2 conv backward needs even input dims (got %dx%d)" % (src.H, src.W))
                dop = L.make_op(L.OP_TCONV, 0, n=self.N, h=out.H, w=out.W, cin=Cout, cout=Cin, ho=src.H, wo=src.W, stride=2, dil=1,
                                aux0=int(self.use_merged(Cin)),
                                inmode=gmode, p_in=out.grad.data_ptr(), p_in_aux=out.buf.data_ptr(),
                                p_in_c=node.t["bconsts"].data_ptr())
                node.t["wd"] = self.add_pack(w, Cout, Cin, False, False, merged=self.use_merged(Cin), wino=self.tconv_layout(dop))
                dop.p[L.RCV_P_W] = node.t["wd"].data_ptr()
            self.grad_target(src, dop, src.H, src.W)
            self.bwd.append(dop)
        self.flush_side()

    def run(self) -> Plan:
        for node in self.nodes:
            lower = getattr(self, "fwd_" + node.op, None)
            if lower is None:
                raise L.RcvError("unknown graph node '%s'" % node.op)
            lower(node)
        self.plan.input_grads = [None] * len(self.in_vals)
        if self.training:
            # which values need a gradient: everything produced by a node, plus flagged inputs
            for node in self.nodes:
                if node.op == "add_slice":
                    raise L.RcvError("this graph is inference only (add_slice has no backward); call .eval()")
                if node.op not in ("cls", "mat"):
                    node.out.needs_grad = True

            for node in reversed(self.nodes):
                self.plan.bwd_marks.append([len(self.bwd), node])      # patched to (op count after this node, min flat offset) in finish()
                getattr(self, "bwd_" + node.op)(node)
        return self.finish()

    def finish(self) -> Plan:
        plan, fwd, bwd, fl, training = self.plan, self.fwd, self.bwd, self.fl, self.training
        # ---- batched filter-gradient reductions: the records keep their positions (every index into the list stays valid); all but
        # the last reduction of a group become RCV_OP_NOP, the last one becomes the table-driven launch of the whole group ----
        plan.reduce_outputs = {}               # index of a batched launch -> [(gradient pointer it writes, zero-fill?)] (schedule tests)
        # (single-GPU: one launch for everything on the large planes -- 640x480: headline -0.3 %, U-Net -0.8 %, v2 -0.6 %; three launches
        # where the step is short and the lone launch at the end of the list is a tail nothing overlaps -- 320x240: +1.1 %, 160x120: +0.3 %)
        big = self.N * max(v.H * v.W for v in self.in_vals) >= 5_000_000
        n_batch = REDUCE_BATCH if (self.eng.grad_ready_cb is not None or self.eng.dry_run or not big) else REDUCE_BATCH_SINGLE
        if training and n_batch > 1:
            idxs = [k for k, op in enumerate(bwd) if op.kind == L.OP_WGRAD_REDUCE]
            for g0 in range(0, len(idxs), n_batch):
                grp = idxs[g0:g0 + n_batch]
                if len(grp) < 2:
                    continue
                # the bias-gradient memsets between the group's first and last reduction ride along as zero-fill jobs
                grp = sorted(grp + [k for k in range(grp[0], grp[-1]) if bwd[k].kind == L.OP_MEMSET])
                rjobs, first, kib, outs = [], 0, 0.0, []
                for k in grp:
                    op = bwd[k]
                    if op.kind == L.OP_MEMSET:
                        cnt = op.i[L.RCV_I_COUNT]
                        rjobs.append(L.RcvReduceJob(part=None, dw=None, db=op.p[L.RCV_P_OUT], nsplit=0, CB=cnt, CA=0, first_block=first))
                        first += -(-cnt // 256)
                        outs.append((op.p[L.RCV_P_OUT], True))
                        continue
                    ca, cb, ns = op.i[L.RCV_I_CIN], op.i[L.RCV_I_COUT], op.i[L.RCV_I_NSPLIT]
                    cap, cbp = (4 if ca <= 4 else _round_up(ca, 16)), _round_up(cb, 16)
                    db = op.p[L.RCV_P_BIAS] or None
                    rjobs.append(L.RcvReduceJob(part=op.p[L.RCV_P_PART], dw=op.p[L.RCV_P_OUT], db=db, nsplit=ns, CB=cb, CA=ca, first_block=first))
                    first += -(-(9 * cbp * cap + (cbp if db else 0)) // 64)
                    kib += self.eng.op_work(op)[1] / 1024.0
                    outs += [(q, False) for q in (op.p[L.RCV_P_OUT], db) if q]
                table = (L.RcvReduceJob * len(rjobs))(*rjobs)
                dev_table = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(self.eng.device)
                plan.keep.append(dev_table)
                for k in grp[:-1]:
                    bwd[k] = L.make_op(L.OP_NOP, 0)
                    self.batch_at[k] = grp[-1]
                bwd[grp[-1]] = L.make_op(L.OP_WGRAD_REDUCE_BATCH, 0, count=len(rjobs), npart=first, aux0=int(kib), p_in=dev_table.data_ptr())
                self.batch_at[grp[-1]] = grp[-1]
                plan.reduce_outputs[grp[-1]] = outs

        # gradient-ready marks: after bwd ops [0:end) every parameter at flat offset >= lo is final (parameters are laid
        # out in forward order and backward visits the nodes in reverse, so the finished region is a growing suffix)
        marks, lo, prev_end = [], fl.numel, 0
        for k, (start, node) in enumerate(plan.bwd_marks):
            end = plan.bwd_marks[k + 1][0] if k + 1 < len(plan.bwd_marks) else len(bwd)
            for q in range(start, end):            # a node whose reduction was folded into a later launch is final only behind that launch
                if q in self.batch_at:
                    end = max(end, self.batch_at[q] + 1)
            end = prev_end = max(end, prev_end)
            offs = [fl.offsets[fl.index(q)] for q in self.node_params(node)] if training else []
            if offs:
                lo = min(lo, min(offs))
            if end > 0 and (not marks or marks[-1][0] != end):
                marks.append((end, lo))
            elif marks:
                marks[-1] = (end, min(marks[-1][1], lo))
        plan.bwd_marks = marks if training else []

        # ---- head of the forward list: running statistics -> constants (eval), then the pack launch (whose jobs may scale by them) ----
        head = list(self.pre)
        if self.jobs:        # (a graph without 3x3 filters -- a lone 1x1 classifier -- packs nothing)
            table = (L.RcvPackJob * len(self.jobs))(*self.jobs)
            host = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8)
            dev_table = host.to(self.eng.device)
            plan.keep.append(dev_table)
            assert C.sizeof(table) == dev_table.numel()
            max_elems = max((4 if j.merged in (1, 4) else 9) * j.rows_pad * j.cols_pad for j in self.jobs)
            head += [L.make_op(L.OP_PACK, 0, count=len(self.jobs), aux0=max_elems, p_in=dev_table.data_ptr())]
        for slots in plan.input_slots:
            for i, (is_bwd, k, sl) in enumerate(slots):
                if not is_bwd:
                    slots[i] = (is_bwd, k + len(head), sl)
        fwd = head + fwd
        plan.n_head = len(head)
        plan.logits_slots = [(k + len(head), sl) for (k, sl) in plan.logits_slots]
        # backward: the filter gradients (and their reductions) are off the critical path d(loss)/d(activation) chain ->
        # second HIP stream inside rcv_run (measured -4 % step time: their latency-bound phases fill the other kernels' gaps)
        if SIDE_STREAM_WGRAD:
            for op in bwd:
                if op.kind in (L.OP_WGRAD, L.OP_WGRAD_REDUCE, L.OP_WGRAD_REDUCE_BATCH, L.OP_MEMSET):
                    op.flags |= L.F_SIDE_STREAM
        plan.fwd = L.OpList(fwd)
        plan.bwd = L.OpList(bwd)
        # plan-time validation: the library answers a query for a record exactly when it would launch it (every shape refusal sits in
        # front of the query return), so an unsupported layer raises HERE with the library's message, not at the first backward
        plan.fwd.labels(self.eng.handle)
        plan.bwd.labels(self.eng.handle)
        return plan


class Engine:
    def __init__(self, graph: dict, params: Sequence[torch.nn.Parameter], bn_modules: Sequence[torch.nn.Module], dry_run: bool = False):
        # dry_run: lower graphs to op lists on whatever device the parameters are on (CPU included) through a planning-only
        # library handle; nothing can be executed.  Used by the CPU tests of the data-parallel schedule.
        self.dry_run = dry_run
        self.graph = graph
        self.param_list = list(params)
        self.bn_modules = list(bn_modules)
        used = set()
        for d in graph["nodes"]:
            for q in (d.get("weight"), d.get("bias")):
                if q is not None:
                    used.add(id(q))
            if d.get("bn") is not None:
                used.add(id(d["bn"].weight))
                used.add(id(d["bn"].bias))
        self.param_used = [id(p) in used for p in self.param_list]
        self.flat: Optional[FlatParams] = None
        self.plans: "collections.OrderedDict[tuple, Plan]" = collections.OrderedDict()
        self.plan_bytes_budget = PLAN_BYTES_BUDGET
        self._generation = 0                 # one forward in flight: backward refuses a stale forward
        self.params_dirty = True             # set by everything that writes parameters / BN buffers through raw pointers (kernels)
        self._last = None
        self.device: Optional[torch.device] = None
        self.handle = None
        # data parallel: called as cb(lo, hi) during backward whenever flat.grad[lo:hi] is final (reverse layer order)
        self.grad_ready_cb = None
        self.grad_buckets = 3

    # ------------------------------------------------------------------ device / parameter state
    def _ensure_device(self, dev: torch.device):
        if self.dry_run:
            if self.flat is None or not self.flat.intact() or self.device != dev:
                self.device = dev
                self.handle = L.planner_handle(256)
                self.flat = FlatParams(self.param_list, dev)
                self.plans.clear()
            return
        if dev.type != "cuda":
            raise L.RcvError("robocupvision_amd computes on an MI355X (HIP) device only; got a tensor on '%s'. "
                             "There is no CPU path: move the model and inputs to cuda." % dev)
        if self.param_list and self.param_list[0].device != dev:
            raise L.RcvError("model parameters are on %s but the input is on %s" % (self.param_list[0].device, dev))
        if self.flat is None or not self.flat.intact() or self.device != dev:
            self.device = dev
            self.handle = L.handle(dev.index if dev.index is not None else torch.cuda.current_device())
            self.flat = FlatParams(self.param_list, dev)
            self.plans.clear()

    def _alloc(self, plan: Plan, *shape, dtype=torch.float32) -> torch.Tensor:
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        plan.keep.append(t)
        plan.bytes += t.numel() * t.element_size()
        return t

    def _zeros(self, plan: Plan, *shape, dtype=torch.float32) -> torch.Tensor:
        t = torch.zeros(*shape, dtype=dtype, device=self.device)
        plan.keep.append(t)
        plan.bytes += t.numel() * t.element_size()
        return t

    def _workspace(self, plan: Plan, op: L.RcvOp) -> Optional[torch.Tensor]:
        nbytes = L.op_workspace(self.handle, op)
        if nbytes == 0:
            return None
        t = self._alloc(plan, (nbytes + 3) // 4)
        op.p[L.RCV_P_PART] = t.data_ptr()
        return t

    # ------------------------------------------------------------------ plan construction
    def _build(self, shapes: Sequence[tuple], training: bool) -> Plan:
        return _Lowering(self, shapes, training).run()

    # ------------------------------------------------------------------ execution
    def _param_state_key(self):
        ts = list(self.param_list)
        for m in self.bn_modules:
            ts += [m.running_mean, m.running_var]
        return tuple(t._version for t in ts if t is not None) + (self.flat._sig if self.flat is not None else ())

    def _plan_for(self, inputs: Sequence[torch.Tensor], training: bool) -> Plan:
        self._ensure_device(inputs[0].device)
        key = (tuple(tuple(t.shape) for t in inputs), training)
        plan = self.plans.get(key)
        if plan is None:
            plan = self._build([tuple(t.shape) for t in inputs], training)
            self.plans[key] = plan
            # every distinct (shape, mode) owns its activations (4.9 GB at 32x640x480): drop the least recently used ones beyond
            # the budget (never the plan just built, nor the one an un-finished forward/backward pair is using)
            total = sum(pl.bytes for pl in self.plans.values())
            for k in list(self.plans):
                if total <= self.plan_bytes_budget:
                    break
                pl = self.plans[k]
                if pl is plan or pl.pinned or (self._last is not None and pl is self._last[0]):
                    continue          # (pinned: a live hipGraph replays raw pointers into this plan's buffers, Trainer.capture)
                total -= pl.bytes
                del self.plans[k]
        else:
            self.plans.move_to_end(key)
        return plan

    def invalidate(self):
        """Parameters or BatchNorm buffers were written behind the engine's back (``.data`` edits, raw-pointer kernels, a graph replay):
        the next eval-mode forward re-derives the packed filters and BatchNorm constants instead of reusing its cached head."""
        self.params_dirty = True

    def forward(self, inputs: Sequence[torch.Tensor], training: bool, fresh_out: bool = False) -> torch.Tensor:
        """Runs the forward list.  Returns the engine-owned logits buffer (overwritten by the next forward of this shape) -- or, with
        ``fresh_out`` on an inference pass, a NEW tensor the last kernel wrote directly (no copy of the result: the reference's modules
        return fresh tensors, and for one LabelProp frame pair the copy launch was 4 % of the call)."""
        for t in inputs:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise L.RcvError("engine inputs must be contiguous float32 tensors")
        plan = self._plan_for(inputs, training)
        out = plan.logits
        if fresh_out and not training and plan.logits_slots:
            out = torch.empty_like(plan.logits)
            for (idx, slot) in plan.logits_slots:
                plan.fwd.arr[idx].p[slot] = out.data_ptr()
        for k, t in enumerate(inputs):
            for (is_bwd, idx, slot) in plan.input_slots[k]:
                (plan.bwd if is_bwd else plan.fwd).arr[idx].p[slot] = t.data_ptr()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if training:
            plan.fwd.run(self.handle, stream)
            self.params_dirty = True          # train-mode BatchNorm updates the running statistics
            nbt = [m.num_batches_tracked for m in self.bn_modules if m.num_batches_tracked is not None]
            if nbt:
                torch._foreach_add_(nbt, 1)
        else:
            # Inference: the packed filters and the BatchNorm constants depend on the parameters only.  They are recomputed when a
            # kernel of this package wrote parameters / buffers (params_dirty) or a torch in-place op did (tensor version counters);
            # otherwise the forward starts behind them (LabelProp frame pairs: 11 of 27 launches of a 0.19 ms call).
            key = self._param_state_key()
            if self.params_dirty or plan.head_key != key:
                plan.fwd.run(self.handle, stream)
                plan.head_key = self._param_state_key()
                self.params_dirty = False
                for pl in self.plans.values():
                    if pl is not plan:
                        pl.head_key = None
            else:
                plan.fwd.run_slice(self.handle, stream, plan.n_head, plan.fwd.n)
            if out is not plan.logits:       # the records are copied at enqueue: point them back at the engine's own buffer (profile_last re-runs the list)
                for (idx, slot) in plan.logits_slots:
                    plan.fwd.arr[idx].p[slot] = plan.logits.data_ptr()
        self._last = (plan, [t for t in inputs])
        self._generation += 1
        return out

    @staticmethod
    def op_work(op) -> tuple:
        """(algorithmic FLOPs, algorithmic bytes) of one op record: FLOPs = 2*MAC of the contraction (the memory-bound
        arithmetic of BN / ReLU / loss is not counted); bytes = every operand tensor read once + every output written once (fp32;
        int64 labels 8 B, arg-max 1 B per pixel; per-channel constants and the <= 4.5 MB of filters excluded)."""
        i = op.i
        n, h, w, cin, cout, ho, wo = (i[L.RCV_I_N], i[L.RCV_I_H], i[L.RCV_I_W], i[L.RCV_I_CIN], i[L.RCV_I_COUT], i[L.RCV_I_HO], i[L.RCV_I_WO])
        two = lambda mode: 2 if mode in (L.LOAD_GRAD_ENC, L.LOAD_GRAD_DEC) else 1
        bstats = 1 if i[L.RCV_I_STATS] in (L.STATS_BWD_ENC, L.STATS_BWD_DEC) else 0
        resid = 1 if op.flags & L.F_RESID else 0
        k, px = op.kind, n * h * w
        flops = nbytes = 0.0
        if k in (L.OP_CONV, L.OP_TCONV):
            flops = 2.0 * 9 * cin * cout * (n * ho * wo if k == L.OP_CONV else px)
            nbytes = 4.0 * (px * cin * two(i[L.RCV_I_INMODE]) + n * ho * wo * cout * (1 + resid + bstats))
        elif k == L.OP_WGRAD:
            flops = 2.0 * 9 * cin * cout * n * ho * wo
            nbytes = 4.0 * (px * cin * two(i[L.RCV_I_INMODE]) + n * ho * wo * cout * two(i[L.RCV_I_INMODE2]))
        elif k == L.OP_WGRAD_REDUCE:
            nbytes = 4.0 * i[L.RCV_I_NSPLIT] * (9 * _round_up(cout, 16) * max(_round_up(cin, 4), 4)) + 4.0 * 9 * cin * cout
        elif k == L.OP_WGRAD_REDUCE_BATCH:
            nbytes = 1024.0 * i[L.RCV_I_AUX0]          # the folded reductions' bytes (set when the group was built)
        elif k in (L.OP_BN_FINALIZE, L.OP_BN_BWD):
            nbytes = 4.0 * i[L.RCV_I_NPART] * 2 * cout
        elif k == L.OP_CLS_FWD:
            flops = 2.0 * cin * cout * px
            skip_c = (i[L.RCV_I_AUX1] or cin) if op.flags & L.F_FUSED_UP else 0
            nbytes = 4.0 * px * (cin + skip_c + cout) + (px * 9.0 if op.flags & L.F_FUSED_CE else 0.0)
        elif k == L.OP_CLS_BWD:
            flops = 2.0 * 2 * cin * cout * px        # data gradient + filter gradient
            src = cin * (2 if op.flags & L.F_FUSED_UP else 1)
            nbytes = 4.0 * px * (src + cin + (0 if op.flags & L.F_FUSED_CE else cout)) + (px * 8.0 if op.flags & L.F_FUSED_CE else 0.0)
        elif k in (L.OP_CE_FWD, L.OP_DICE_FWD):
            nbytes = px * (4.0 * cout + 8 + 1)
        elif k in (L.OP_CE_BWD, L.OP_DICE_BWD):
            nbytes = px * (8.0 * cout + 8)
        elif k == L.OP_POOL_FWD:
            nbytes = 4.0 * px * cout * 1.25
        elif k == L.OP_POOL_BWD:
            nbytes = 4.0 * px * cout * (0.25 + 1 + 1 + resid)
        elif k == L.OP_COMBINE:
            nbytes = 4.0 * px * cout * (2 + (2 if op.flags & L.F_CONCAT else 1))
        elif k == L.OP_MATERIALIZE:
            nbytes = 4.0 * px * cout * 2
        elif k == L.OP_ADD_SLICE:
            nbytes = 4.0 * px * cin * 3
        elif k == L.OP_BWD_STATS:
            nbytes = 4.0 * px * ((cin or cout) + cout * (1 + bstats))
        elif k in (L.OP_NHWC_TO_NCHW, L.OP_NCHW_TO_NHWC):
            nbytes = 4.0 * px * (cin + cout)
        elif k in (L.OP_ADAM_L1, L.OP_SGD):
            nbytes = 4.0 * (i[L.RCV_I_COUNT] & 0xFFFFFFFF) * (7 if k == L.OP_ADAM_L1 else 5)
        elif k == L.OP_MEMSET:
            nbytes = 4.0 * i[L.RCV_I_COUNT]
        elif k == L.OP_PACK:
            nbytes = 8.0 * i[L.RCV_I_AUX0] * 0      # (filter repack: parameter bytes, excluded like the filters themselves)
        return flops, nbytes

    def profile_last(self, reps: int = 3, with_loss: bool = True):
        """Profiling aid for bench.py: re-runs the last forward (+ backward) plan with a HIP event pair around
        every op (rcv_run_timed) and returns rows of (label, kind, avg ms, algorithmic FLOPs, algorithmic bytes).  When the
        Trainer's fast path is in use (cross entropy fused into the classifier ops) those are the lists that are timed."""
        plan, _inputs = self._last
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ce = plan.ce if (with_loss and plan.ce) else None
        lists = [ce["fwd"] if ce else plan.fwd]
        bwd = ce["bwd"] if ce else plan.bwd
        if bwd is not None and bwd.n:
            if not ce:
                dl = torch.full_like(plan.logits, 1e-4)
                for (idx, slot) in plan.dlogits_slots:
                    plan.bwd.arr[idx].p[slot] = dl.data_ptr()
            lists.append(bwd)
        rows = []
        for lst in lists:
            labels = lst.labels(self.handle)
            acc = [0.0] * lst.n
            for _ in range(reps):
                for k, v in enumerate(lst.run_timed(self.handle, stream)):
                    acc[k] += v
            for k in range(lst.n):
                op = lst.arr[k]
                i = op.i
                flops, nbytes = self.op_work(op)
                if lst is lists[0] and len(lists) == 1 and k < plan.n_head:
                    continue          # inference: the parameter-only head (filter repack, BN constants) is not part of a steady-state call
                rows.append({"label": labels[k], "kind": int(op.kind), "ms": acc[k] / reps, "flops": flops, "bytes": nbytes,
                             "shape": "%dx%dx%d %d->%d s%d" % (i[L.RCV_I_N], i[L.RCV_I_H], i[L.RCV_I_W], i[L.RCV_I_CIN], i[L.RCV_I_COUT],
                                                              i[L.RCV_I_STRIDE]), "bwd": lst is not lists[0]})
        return rows

    def backward(self, dlogits: torch.Tensor, generation: Optional[int] = None):
        """Backward of the LAST forward.  The activations, the logits and the op lists are engine-owned and overwritten by the next
        forward of this model (any shape, train or eval), so only one forward may be in flight: ``generation`` (the value of
        ``_generation`` right after the forward being differentiated) is checked against the current one."""
        if generation is not None and generation != self._generation:
            raise L.RcvError("backward of a stale forward: this model ran another forward (generation %d -> %d) before "
                             "loss.backward(); the engine keeps ONE forward in flight (its activations are overwritten by the "
                             "next call) -- call backward before the next forward, or clone what you need" % (generation, self._generation))
        plan, _inputs = self._last
        if plan.bwd is None or plan.bwd.n == 0:
            raise L.RcvError("backward called on an eval-mode forward; call model.train() first")
        if dlogits.dtype != torch.float32 or not dlogits.is_contiguous() or dlogits.shape != plan.logits.shape:
            dlogits = dlogits.to(torch.float32).contiguous()
        for (idx, slot) in plan.dlogits_slots:
            plan.bwd.arr[idx].p[slot] = dlogits.data_ptr()
        self._run_backward(plan, plan.bwd)
        return plan

    SIDE_MODES = ("all", "reduce", "off")      # which backward ops run on the library's second stream

    @staticmethod
    def _set_side(ops: L.OpList, mode):
        """mode: "all" (True) = filter gradients, their reductions and the bias memsets; "reduce" = only the reductions and memsets (small
        bandwidth-bound launches that fit beside the next data-gradient kernel, while the filter-gradient kernels themselves keep
        the whole chip); "off" (False) = everything on the caller's stream."""
        mode = {True: "all", False: "off"}.get(mode, mode)
        for k in range(ops.n):
            op = ops.arr[k]
            if op.kind in (L.OP_WGRAD, L.OP_WGRAD_REDUCE, L.OP_WGRAD_REDUCE_BATCH, L.OP_MEMSET):
                on = mode == "all" or (mode == "reduce" and op.kind != L.OP_WGRAD)
                op.flags = (op.flags | L.F_SIDE_STREAM) if on else (op.flags & ~L.F_SIDE_STREAM)

    def _decide_side_stream(self, plan: Plan, ops: L.OpList):
        """Times the backward list under each schedule (it only overwrites engine buffers: re-running it is harmless) and keeps the fastest."""
        plan.side_decided = True
        capturing = torch.cuda.is_current_stream_capturing()          # (no event synchronisation inside a graph capture)
        # (data-parallel runs measure too -- each rank for itself, with plain un-bucketed passes before its first exchange: the three
        # schedules are bit-identical, so ranks may even settle on different ones)
        if not SIDE_STREAM_WGRAD or SIDE_STREAM_MODE != "auto" or capturing:
            plan.side_mode = ("all" if SIDE_STREAM_MODE in ("auto", "1") else SIDE_STREAM_MODE) if SIDE_STREAM_WGRAD else "off"
            if plan.side_mode not in self.SIDE_MODES:
                plan.side_mode = "all"
        else:
            stream = torch.cuda.current_stream(self.device)
            ms = []
            for mode in self.SIDE_MODES:
                self._set_side(ops, mode)
                ops.run(self.handle, stream.cuda_stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(3):
                    ops.run(self.handle, stream.cuda_stream)
                e1.record(stream)
                e1.synchronize()
                ms.append(e0.elapsed_time(e1) / 3)
            plan.side_ms = tuple(ms)
            # "all" unless another schedule is CLEARLY faster here (> 3 %: the U-Net configuration, whose one-stream backward measures
            # 5 % faster).  Inside a real step the two-stream schedule gains more than this back-to-back measurement shows -- measured on
            # the headline step: schedules within 1 % of each other here (4.61 / 4.59 ms), but 6.37 ms per step with "all" against 6.58
            # with "off" -- so a near-tie must not flip the choice.
            best = min(range(len(ms)), key=lambda k: ms[k])
            plan.side_mode = self.SIDE_MODES[best if ms[best] < 0.97 * ms[0] else 0]
        plan.side_on = plan.side_mode != "off"
        for lst in (plan.bwd, plan.ce["bwd"] if plan.ce else None):
            if lst is not None:
                self._set_side(lst, plan.side_mode)

    def _run_backward(self, plan: Plan, ops):
        if self.dry_run:
            plan.side_decided = True
            stream = 0
        else:
            if not plan.side_decided:
                self._decide_side_stream(plan, ops)
            stream = torch.cuda.current_stream(self.device).cuda_stream
        if self.grad_ready_cb is None or not plan.bwd_marks:
            ops.run(self.handle, stream)
            return
        # bucketed: run the op list in slices and hand finished gradient ranges to the caller (all-reduce on a side stream).  The
        # slices do not join the filter-gradient stream back (that would stall the d(activation) chain at every bucket border): the
        # callback makes ITS stream wait for it (join_side), the compute stream joins once at the end.
        run_bucketed(plan.bwd_marks, self.flat.numel, self.grad_buckets, ops.n,
                     lambda a, b: ops.run_slice(self.handle, stream, a, b, join=False),
                     self.grad_ready_cb,
                     (lambda: None) if self.dry_run else (lambda: L.join_side(self.handle, stream)))

    # ------------------------------------------------------------------ loss fused into the classifier (Trainer fast path)
    def _ce_variant(self, plan: Plan):
        """Copies of the plan's op lists in which the fused 1x1 classifier also evaluates CrossEntropyLoss2d (forward) and forms
        d loss / d logits itself (backward): RCV_F_FUSED_CE.  None when the graph does not end in that classifier."""
        if plan.ce is not None:
            return plan.ce or None
        plan.ce = False
        if plan.bwd is None or plan.bwd.n == 0:
            return None
        kf = [k for k in range(plan.fwd.n) if plan.fwd.arr[k].kind == L.OP_CLS_FWD and plan.fwd.arr[k].flags & L.F_FUSED_UP]
        kb = [k for k in range(plan.bwd.n) if plan.bwd.arr[k].kind == L.OP_CLS_BWD and plan.bwd.arr[k].flags & L.F_FUSED_UP]
        if len(kf) != 1 or len(kb) != 1 or kf[0] != plan.fwd.n - 1 or kb[0] != 0:
            return None
        fops = [L.RcvOp.from_buffer_copy(plan.fwd.arr[k]) for k in range(plan.fwd.n)]
        bops = [L.RcvOp.from_buffer_copy(plan.bwd.arr[k]) for k in range(plan.bwd.n)]
        f, b = fops[kf[0]], bops[kb[0]]
        N, H, W = f.i[L.RCV_I_N], f.i[L.RCV_I_H], f.i[L.RCV_I_W]
        loss_out = self._zeros(plan, 4)
        argmax = self._alloc(plan, N, H, W, dtype=torch.uint8)
        gone = torch.ones(1, dtype=torch.float32, device=self.device)
        plan.keep.append(gone)
        f.flags |= L.F_FUSED_CE
        f.p[L.RCV_P_X1] = loss_out.data_ptr()
        f.p[L.RCV_P_X2] = argmax.data_ptr()
        self._workspace(plan, f)
        b.flags |= L.F_FUSED_CE
        b.p[L.RCV_P_BIAS] = f.p[L.RCV_P_BIAS]
        b.p[L.RCV_P_X5] = loss_out.data_ptr()
        b.p[L.RCV_P_IN2_AUX] = gone.data_ptr()
        plan.ce = {"fwd": L.OpList(fops), "bwd": L.OpList(bops), "kf": kf[0], "kb": kb[0], "loss_out": loss_out, "argmax": argmax}
        plan.ce["fwd"].labels(self.handle)          # (plan-time validation, as in _build)
        plan.ce["bwd"].labels(self.handle)
        return plan.ce

    def forward_ce(self, inputs: Sequence[torch.Tensor], targets: torch.Tensor, weight: Optional[torch.Tensor]):
        """Training forward with CrossEntropyLoss2d(weight) fused into the classifier op.  Returns (logits, loss_out[4], argmax) --
        loss_out as RCV_OP_CE_FWD: [loss, sum_w, #correct, sum_w*nll] -- or None when this graph has no such fast path."""
        for t in inputs:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise L.RcvError("engine inputs must be contiguous float32 tensors")
        plan = self._plan_for(inputs, True)
        ce = self._ce_variant(plan)
        if ce is None:
            return None
        N, H, W = ce["argmax"].shape
        if targets.dtype != torch.int64 or not targets.is_contiguous() or tuple(targets.shape) != (N, H, W) or targets.device != self.device:
            raise L.RcvError("targets must be a contiguous int64 [B,H,W] tensor on the model's device")
        for k, t in enumerate(inputs):
            for (is_bwd, idx, slot) in plan.input_slots[k]:
                (ce["bwd"] if is_bwd else ce["fwd"]).arr[idx].p[slot] = t.data_ptr()
                (plan.bwd if is_bwd else plan.fwd).arr[idx].p[slot] = t.data_ptr()      # keeps the plain lists usable (profile_last)
        wptr = None if weight is None else weight.data_ptr()
        for op in (ce["fwd"].arr[ce["kf"]], ce["bwd"].arr[ce["kb"]]):
            op.p[L.RCV_P_IN2] = targets.data_ptr()
            op.p[L.RCV_P_X0] = wptr
        ce["fwd"].run(self.handle, torch.cuda.current_stream(self.device).cuda_stream)
        self.params_dirty = True
        nbt = [m.num_batches_tracked for m in self.bn_modules if m.num_batches_tracked is not None]
        if nbt:
            torch._foreach_add_(nbt, 1)
        self._last = (plan, [t for t in inputs])
        self._generation += 1
        self._last_ce = (targets, weight)        # keep alive until backward_ce
        return plan.logits, ce["loss_out"], ce["argmax"]

    def backward_ce(self):
        """Backward of the loss produced by the last forward_ce (d loss = 1): fills the flat gradient buffer."""
        plan, _inputs = self._last
        self._run_backward(plan, plan.ce["bwd"])
        return plan
