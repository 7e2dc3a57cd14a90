"""GPU: every block of the hot path through the C ABI against golden vectors of the imported reference
(layer KATs, tests/golden/layer_kats.npz) -- forward (train + eval), data / filter / bias / BatchNorm
gradients, running statistics.  Tolerance: 1e-3 relative (BASELINE.json north_star), checked as
|a-b| <= 1e-3*|b| + 1e-3*max|b|*1e-2 so that values near zero are judged against the tensor scale."""
import numpy as np
import pytest
import torch

from oracle import cpu_reference as O
import robocupvision_amd.model as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    a = np.asarray(a)
    return torch.from_numpy(np.ascontiguousarray(a)).reshape(a.shape)


def close(a, b, what, rtol=1e-3, floor=1e-2):
    """|a-b| <= rtol*|b| + rtol*floor*max|b|.  floor=1e-2 for activations; gradients (sums with heavy
    cancellation, where fp32 summation order alone moves small entries) use floor=1: 1e-3 of the tensor scale."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    scale = float(b.abs().max()) + 1e-30
    err = (a - b).abs()
    bound = rtol * b.abs() + rtol * floor * scale
    bad = err > bound
    assert not bool(bad.any()), "%s: %d/%d elements off, max err %.3e (scale %.3e), worst rel-to-scale %.3e" % (
        what, int(bad.sum()), bad.numel(), float(err.max()), scale, float(err.max()) / scale)


def _load_block(kats, name, mod):
    pre = name + "/p/"
    sd = {k[len(pre):]: _t(kats[k]) for k in kats.files if k.startswith(pre)}
    mod.load_state_dict(sd)
    return mod.to(DEV)


CONV_CASES = [("conv_3_8_s1", 3, 8, 1), ("conv_8_16_s2", 8, 16, 2), ("conv_16_16_s1", 16, 16, 1), ("conv_32_64_s2", 32, 64, 2),
              ("conv_64_64_s1", 64, 64, 1), ("conv_128_128_s1", 128, 128, 1), ("conv_8_8_s1_odd", 8, 8, 1)]


def _run_block(layer_kats, name, mod, check_gx=True):
    x = _t(layer_kats[name + "/x"]).to(DEV).requires_grad_(True)
    mod.train()
    y = mod(x)
    close(y, _t(layer_kats[name + "/y_train"]), name + " y_train")
    y.backward(_t(layer_kats[name + "/gy"]).to(DEV))
    torch.cuda.synchronize()
    for k, p in mod.named_parameters():
        ref = _t(layer_kats["%s/g/%s" % (name, k)])
        if k == "conv.bias" and isinstance(mod, M.upSampleTransposeConv):
            # bias ahead of BatchNorm: exact gradient is 0; the reference returns rounding noise
            assert float(p.grad.abs().max()) == 0.0 and float(ref.abs().max()) < 1e-4
            continue
        close(p.grad, ref, "%s grad %s" % (name, k), floor=1.0)
    if check_gx:
        close(x.grad, _t(layer_kats[name + "/gx"]), name + " gx", floor=1.0)
    close(mod.bn.running_mean, _t(layer_kats[name + "/after/bn.running_mean"]), name + " running_mean")
    close(mod.bn.running_var, _t(layer_kats[name + "/after/bn.running_var"]), name + " running_var")
    assert int(mod.bn.num_batches_tracked) == 1
    mod.eval()
    with torch.no_grad():
        ye = mod(x.detach())
    close(ye, _t(layer_kats[name + "/y_eval"]), name + " y_eval")


def _graph_engine(graph, mod):
    from robocupvision_amd.engine import Engine
    return Engine(graph, list(mod.parameters()), [m for m in mod.modules() if isinstance(m, torch.nn.BatchNorm2d)])


@pytest.mark.parametrize("name,cin,cout,stride", CONV_CASES)
def test_conv_block(layer_kats, name, cin, cout, stride):
    mod = _load_block(layer_kats, name, M.Conv(cin, cout, 3, stride))
    if cin % 4 == 0:
        _run_block(layer_kats, name, mod)
        return
    # The 3-channel block is the network's first layer: it reads the NCHW image directly (conv_first.hip / wgrad_first.hip).
    # One-node graph with an NCHW input: forward (train, eval), filter / bias / BatchNorm gradients and running statistics against
    # the reference's values (the image has no gradient in the network, so there is no gx to compare).
    eng = _graph_engine({"inputs": [{"layout": "nchw"}], "nodes": [mod._node(("in", 0)), {"op": "mat", "src": ("node", 0)}]}, mod)
    x = _t(layer_kats[name + "/x"]).to(DEV)
    mod.train()
    y = M._run_engine(eng, True, [x]).permute(0, 3, 1, 2)
    close(y, _t(layer_kats[name + "/y_train"]), name + " y_train")
    labels = eng._last[0].fwd.labels(eng.handle)
    assert any(l.startswith("conv_first") for l in labels), labels
    y.backward(_t(layer_kats[name + "/gy"]).to(DEV))
    torch.cuda.synchronize()
    assert any(l.startswith("wgrad_first") for l in eng._last[0].bwd.labels(eng.handle))
    for k, p in mod.named_parameters():
        close(p.grad, _t(layer_kats["%s/g/%s" % (name, k)]), "%s grad %s" % (name, k), floor=1.0)
    close(mod.bn.running_mean, _t(layer_kats[name + "/after/bn.running_mean"]), name + " running_mean")
    close(mod.bn.running_var, _t(layer_kats[name + "/after/bn.running_var"]), name + " running_var")
    mod.eval()
    with torch.no_grad():
        ye = M._run_engine(eng, False, [x]).permute(0, 3, 1, 2)
    close(ye, _t(layer_kats[name + "/y_eval"]), name + " y_eval")


def test_classifier_1x1_kat(layer_kats):
    """UltClassifier 1x1 (model.py:403-414) as a one-node graph: logits, data / filter / bias gradients against the reference."""
    cls = M.UltClassifier(8, 5, False)
    cls.load_state_dict({k[len("cls/p/"):]: _t(layer_kats[k]) for k in layer_kats.files if k.startswith("cls/p/")})
    cls = cls.to(DEV)
    eng = _graph_engine({"inputs": [{"layout": "nhwc", "requires_grad": True}], "nodes": [cls._node(("in", 0))]}, cls)
    x = _t(layer_kats["cls/x"]).to(DEV).permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    y = M._run_engine(eng, True, [x])
    close(y, _t(layer_kats["cls/y"]), "cls y")
    y.backward(_t(layer_kats["cls/gy"]).to(DEV))
    torch.cuda.synchronize()
    close(x.grad.permute(0, 3, 1, 2), _t(layer_kats["cls/gx"]), "cls gx", floor=1.0)
    for k, p in cls.named_parameters():
        close(p.grad, _t(layer_kats["cls/g/" + k]), "cls grad " + k, floor=1.0)


@pytest.mark.parametrize("first", [0, 3])
def test_maxpool_kat(layer_kats, first):
    """MaxPool2d(2,2) forward / backward kernels (model.py:92-103) against the reference's pool/* vectors.  The pool runs inside a
    graph, so it is followed by a 1x1 classifier whose weights SELECT five of the eight pooled channels (exact 1.0 / 0.0): the
    logits are then the pooled values themselves and the data gradient is the pool's own backward."""
    cls = M.UltClassifier(8, 5, False)
    with torch.no_grad():
        cls.layers.Class.weight.zero_()
        cls.layers.Class.bias.zero_()
        for c in range(5):
            cls.layers.Class.weight[c, first + c, 0, 0] = 1.0
    cls = cls.to(DEV)
    eng = _graph_engine({"inputs": [{"layout": "nhwc", "requires_grad": True}],
                         "nodes": [{"op": "pool", "src": ("in", 0)}, cls._node(("node", 0))]}, cls)
    x = _t(layer_kats["pool/x"]).to(DEV).permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    y = M._run_engine(eng, True, [x])
    ref_y = _t(layer_kats["pool/y"])[:, first:first + 5]
    assert torch.equal(y.cpu(), ref_y)                              # a max and a multiplication by 1.0: bit exact
    gy = _t(layer_kats["pool/gy"])[:, first:first + 5].contiguous()
    y.backward(gy.to(DEV))
    torch.cuda.synchronize()
    gx = x.grad.permute(0, 3, 1, 2).cpu()
    assert torch.equal(gx[:, first:first + 5], _t(layer_kats["pool/gx"])[:, first:first + 5])
    rest = [c for c in range(8) if not first <= c < first + 5]
    assert float(gx[:, rest].abs().max()) == 0.0


def test_standalone_pool_module_kat(layer_kats):
    """The reference's Pool module called on its own (model.py:92-100: NCHW in, NCHW out, autograd): forward and backward bit-exact against
    its pool/* vectors (a max and a routing of the gradient: integer-like work), in training mode (copy of the engine's buffer) and in
    eval mode (fresh output tensor)."""
    pool = M.Pool(8).to(DEV)
    x = _t(layer_kats["pool/x"]).to(DEV).requires_grad_(True)
    y = pool(x)
    assert torch.equal(y.cpu(), _t(layer_kats["pool/y"]))
    y.backward(_t(layer_kats["pool/gy"]).to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(x.grad.cpu(), _t(layer_kats["pool/gx"]))
    pool.eval()
    with torch.no_grad():
        y2 = pool(x.detach())
        y3 = pool(x.detach() * 2)
    assert torch.equal(y2.cpu(), _t(layer_kats["pool/y"])) and torch.equal(y3, 2 * y2) and y3.data_ptr() != y2.data_ptr()
    with pytest.raises(ValueError):
        pool(torch.zeros(1, 8, 5, 6, device=DEV))


@pytest.mark.parametrize("name,cin,cout", [("up_16_8", 16, 8), ("up_64_32", 64, 32), ("up_128_64", 128, 64)])
def test_up_block(layer_kats, name, cin, cout):
    mod = _load_block(layer_kats, name, M.upSampleTransposeConv(cin, cout))
    _run_block(layer_kats, name, mod)


@pytest.mark.parametrize("name,cin,cout,s,d", [("cps_32_64_d2", 32, 64, 1, 2), ("cps_8_16_s2", 8, 16, 2, 1)])
def test_dilated_block_eval(layer_kats, name, cin, cout, s, d):
    mod = _load_block(layer_kats, name, M.ConvPoolSimple(cin, cout, 3, s, d, d, False))
    # the golden's eval output uses the running statistics AFTER its one training step
    mod.bn.running_mean.copy_(_t(layer_kats[name + "/after/bn.running_mean"]))
    mod.bn.running_var.copy_(_t(layer_kats[name + "/after/bn.running_var"]))
    mod.eval()
    with torch.no_grad():
        y = mod(_t(layer_kats[name + "/x"]).to(DEV))
    close(y, _t(layer_kats[name + "/y_eval"]), name + " y_eval")


@pytest.mark.parametrize("name", ["ce_w", "ce_now"])
def test_cross_entropy(layer_kats, name):
    w = _t(layer_kats[name + "/w"]) if (name + "/w") in layer_kats.files else None
    crit = M.CrossEntropyLoss2d(w).to(DEV)
    lg = _t(layer_kats[name + "/logits"]).to(DEV).requires_grad_(True)
    t = _t(layer_kats[name + "/t"]).to(DEV)
    loss = crit(lg, t)
    ref = float(layer_kats[name + "/loss"])
    assert abs(float(loss) - ref) <= 1e-5 * abs(ref)
    loss.backward()
    close(lg.grad, _t(layer_kats[name + "/glogits"]), name + " dlogits", rtol=1e-4)
    # mask: bit exact (integer work)
    assert np.array_equal(crit.last_argmax.cpu().numpy(), layer_kats[name + "/argmax"].astype(np.uint8))
    assert int(crit.last_stats[2]) == int((layer_kats[name + "/argmax"] == layer_kats[name + "/t"]).sum())


def test_argmax_tie_rule():
    lg = torch.tensor([1.0, 3.0, 3.0, 2.0], device=DEV).view(1, 4, 1, 1)
    crit = M.CrossEntropyLoss2d().to(DEV)
    crit(lg, torch.zeros(1, 1, 1, dtype=torch.long, device=DEV))
    assert int(crit.last_argmax.view(-1)[0]) == 1     # first maximum wins (train.py:70 torch.max)


def test_empty_and_bad_shapes_raise():
    m = M.ROBO_UNet().to(DEV)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 20, 24, device=DEV))       # not a multiple of 8
    with pytest.raises(ValueError):
        m(torch.zeros(1, 4, 16, 16, device=DEV))


def test_device_metrics_match_reference_loops():
    """RCV_OP_CONFUSION + SegmentationMetrics against the reference's Python mask loops (train.py:136-163)."""
    from robocupvision_amd.metrics import SegmentationMetrics
    g = torch.Generator().manual_seed(3)
    C, B, H, W = 5, 3, 24, 40
    pred = torch.randint(0, C, (B, H, W), generator=g)
    tgt = torch.randint(0, C, (B, H, W), generator=g)
    tgt[1][tgt[1] == 3] = 0                     # a class absent from one image (union == 0 branch)
    pred[1][pred[1] == 3] = 1
    ref = O.valid_metrics(pred, tgt, C)                                          # the oracle's restatement of train.py:127,136-163
    ref_cls, ref_iou = ref["mean_class_acc"], ref["mean_iou"]
    m = SegmentationMetrics(C, DEV)
    m.update(pred.to(torch.uint8).to(DEV), tgt.to(DEV))
    out = m.compute()
    assert abs(out["mean_class_acc"] - ref_cls) < 1e-4 and abs(out["mean_iou"] - ref_iou) < 1e-4      # the loop reference is fp32
    assert abs(out["pixel_acc"] - float((pred == tgt).sum()) / pred.numel() * 100) < 1e-9 and abs(out["pixel_acc"] - ref["pixel_acc"]) < 1e-4
    assert float((out["confusion_percent"].float() - ref["confusion_percent"]).abs().max()) < 1e-3


@pytest.mark.parametrize("N,H,W,dil,mode2", [(2, 37, 70, 1, "grad_enc"), (1, 19, 131, 2, "grad_dec"), (3, 8, 64, 1, "plain"),
                                              (2, 40, 129, 2, "grad_enc")])
def test_first_layer_filter_gradient_kernel(N, H, W, dil, mode2):
    """RCV_OP_WGRAD on the NCHW image (3 -> 8 channels: model.py:475 Level0.Conv0, PB_FCN's dilated conv0) through the C ABI: the
    vector-ALU kernel (wgrad_first.hip) against the float64 filter gradient of torch's conv2d for ragged planes; tolerance 2e-5 of
    the largest entry (fp32 sums over N*H*W pixels)."""
    from robocupvision_amd import _lib as L
    dev = torch.device("cuda:0")
    h = L.handle(0)
    g = torch.Generator().manual_seed(7)
    img = torch.randn(N, 3, H, W, generator=g)
    gy = torch.randn(N, H, W, 8, generator=g)                     # gradient at the block output (NHWC)
    r = torch.randn(N, H, W, 8, generator=g)                      # the stored relu(conv) / raw conv output
    k = torch.rand(5, 8, generator=g) + 0.5
    k64, gy64, r64 = k.double(), gy.double(), r.double()
    if mode2 == "plain":
        dz = gy64
    elif mode2 == "grad_enc":                                     # dz = r > 0 ? k0*g + k2*r + k1 : 0      (include/rcv.h RCV_LOAD_GRAD_ENC)
        dz = torch.where(r64 > 0, k64[0] * gy64 + k64[2] * r64 + k64[1], torch.zeros_like(r64))
    else:                                                         # dz = k0*(k3*r + k4 > 0 ? g : 0) + k2*r + k1  (RCV_LOAD_GRAD_DEC)
        dz = k64[0] * torch.where(k64[3] * r64 + k64[4] > 0, gy64, torch.zeros_like(gy64)) + k64[2] * r64 + k64[1]
    w = torch.zeros(8, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(8, dtype=torch.float64, requires_grad=True)
    out = torch.nn.functional.conv2d(img.double(), w, b, stride=1, padding=dil, dilation=dil)
    out.backward(dz.permute(0, 3, 1, 2))
    modes = {"plain": L.LOAD_PLAIN, "grad_enc": L.LOAD_GRAD_ENC, "grad_dec": L.LOAD_GRAD_DEC}
    img_d, gy_d, r_d, k_d = img.to(dev), gy.to(dev), r.to(dev), k.to(dev)
    dw = torch.full((8, 3, 3, 3), float("nan"), device=dev)
    db = torch.full((8,), float("nan"), device=dev)
    op = L.make_op(L.OP_WGRAD, L.F_BIAS, n=N, h=H, w=W, cin=3, ho=H, wo=W, cout=8, stride=1, dil=dil, inmode=L.LOAD_NCHW,
                   inmode2=modes[mode2], p_in=img_d.data_ptr(), p_in2=gy_d.data_ptr(), p_in2_aux=r_d.data_ptr(), p_in2_c=k_d.data_ptr())
    nb = L.op_workspace(h, op)
    part = torch.full((max(nb // 4, 4),), float("nan"), device=dev)   # rows no workgroup writes must not be read into results
    op.p[L.RCV_P_PART] = part.data_ptr()
    red = L.make_op(L.OP_WGRAD_REDUCE, 0, cin=3, cout=8, nsplit=op.i[L.RCV_I_NSPLIT], p_part=part.data_ptr(), p_out=dw.data_ptr(),
                    p_bias=db.data_ptr())
    lst = L.OpList([op, red])
    label = lst.labels(h)[0]
    assert label.startswith("wgrad_first"), label
    lst.run(h, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for got, ref, what in ((dw, w.grad, "dW"), (db, b.grad, "db")):
        err = float((got.double().cpu() - ref).abs().max())
        assert err <= 2e-5 * float(ref.abs().max()) + 1e-6, (label, what, err, float(ref.abs().max()))


def test_adam_l1_named_entry_points_vs_torch():
    """rcv_adam_l1_step / rcv_adam_l1_step_metrics called through ctypes as a reference-side binding would:
    three steps against torch.optim.Adam on `grad + decay*sign(p)` (train.py:23-27,52-55,67), and the metrics row against the
    sums train.py:52-53,69-73 keep (loss + decay*sum|p| with the PRE-update parameters, the L1 term, correct pixels, steps)."""
    import ctypes as C
    from robocupvision_amd import _lib as L
    lib, h = L.load(), L.handle(0)
    dev = torch.device("cuda:0")
    n, decay, lr = 100003, 1e-3, 2e-3
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.Adam([ref], lr=lr)
    p = p0.clone().to(dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    metrics = torch.zeros(4, dtype=torch.float64, device=dev)
    op = L.make_op(L.OP_ADAM_L1, 0, count=n)
    ws = torch.zeros((L.op_workspace(h, op) + 7) // 8, dtype=torch.float64, device=dev)
    rows = op.i[L.RCV_I_NPART]
    fn = lib.rcv_adam_l1_step_metrics
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 6 + [C.c_int64] + [C.c_float] * 5 + [C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    exp = [0.0, 0.0, 0.0, 0.0]
    st = torch.cuda.current_stream().cuda_stream
    for step in range(1, 4):
        grad = torch.randn(n, generator=g)
        stats = torch.tensor([0.5 * step, 0.0, 7.0 * step, 0.0], dtype=torch.float32)
        reg = decay * float(ref.detach().abs().sum())
        exp = [exp[0] + 0.5 * step + reg, exp[1] + reg, exp[2] + 7.0 * step, exp[3] + 1]
        ref.grad = grad.double() + decay * torch.sign(ref.detach())
        opt.step()
        gd, sd = grad.to(dev), stats.to(dev)
        rc = fn(h, p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), None, n, lr, 0.9, 0.999, 1e-8, decay, step, 1.0,
                metrics.data_ptr(), sd.data_ptr(), ws.data_ptr(), rows, st)
        assert rc == 0, lib.rcv_last_error()
        torch.cuda.synchronize()
        assert float((p.double().cpu() - ref.detach()).abs().max()) <= 2e-6
    got = metrics.cpu().tolist()
    for a, b in zip(got, exp):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (got, exp)
    # buffers that are not 16-byte aligned (views one float into an allocation) take the element-by-element path: same bits
    step_fn = lib.rcv_adam_l1_step
    step_fn.restype = C.c_int
    step_fn.argtypes = [C.c_void_p] * 6 + [C.c_int64] + [C.c_float] * 5 + [C.c_int, C.c_float, C.c_void_p]
    grad = torch.randn(n, generator=g).to(dev)
    outs = []
    for shift in (0, 1):
        bufs = [torch.zeros(n + 4, device=dev) for _ in range(4)]
        pp, gg, mm, vv = (b[shift:shift + n] for b in bufs)
        pp.copy_(p); gg.copy_(grad); mm.copy_(m); vv.copy_(v)
        assert (pp.data_ptr() % 16 == 0) == (shift == 0)
        rc = step_fn(h, pp.data_ptr(), gg.data_ptr(), mm.data_ptr(), vv.data_ptr(), None, n, lr, 0.9, 0.999, 1e-8, decay, 4, 1.0, st)
        assert rc == 0, lib.rcv_last_error()
        torch.cuda.synchronize()
        outs.append((pp.clone(), mm.clone(), vv.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _abi_conv(x_nhwc, w, mode_wino, relu_bias=None):
    """One RCV_OP_PACK + RCV_OP_CONV through the C ABI: plain 3x3 stride-1 conv of an NHWC tensor (no load transform)."""
    from robocupvision_amd import _lib as L
    import ctypes as C
    dev = x_nhwc.device
    h = L.handle(0)
    N, H, W, Cin = x_nhwc.shape
    Cout = w.shape[0]
    rp, cp = (Cin + 3) // 4 * 4, (Cout + 15) // 16 * 16
    wp = torch.zeros((16 if mode_wino else 9) * rp * cp, device=dev)
    job = L.RcvPackJob()
    job.src, job.dst, job.D0, job.D1 = w.data_ptr(), wp.data_ptr(), Cout, Cin
    job.rows_from_d1, job.flip, job.rows_pad, job.cols_pad, job.merged = 1, 0, rp, cp, 2 if mode_wino else 0
    table = torch.frombuffer(bytearray(bytes((L.RcvPackJob * 1)(job))), dtype=torch.uint8).to(dev)
    out = torch.full((N, H, W, Cout), float("nan"), device=dev)
    pack = L.make_op(L.OP_PACK, 0, count=1, aux0=16 * rp * cp, p_in=table.data_ptr())
    conv = L.make_op(L.OP_CONV, 0, n=N, h=H, w=W, cin=Cin, cout=Cout, ho=H, wo=W, stride=1, dil=1, inmode=L.LOAD_PLAIN,
                     aux0=2 if mode_wino else 0, p_in=x_nhwc.data_ptr(), p_w=wp.data_ptr(), p_out=out.data_ptr())
    lst = L.OpList([pack, conv])
    label = lst.labels(h)[1]
    lst.run(h, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out, label


@pytest.mark.parametrize("N,H,W,C", [(4, 30, 40, 128), (2, 60, 80, 64), (3, 5, 7, 128), (1, 9, 3, 64), (2, 16, 34, 128)])
def test_winograd_error_budget(N, H, W, C):
    """fp32 error budget of the Winograd F(2x2,3x3) kernel (conv_wino.hip) on the layers it serves (the 128 -> 128 and 64 -> 64
    stride-1 convs), against an fp64 convolution and against the direct MFMA kernel on the same data: the Winograd result must be
    within 1e-5 of the output scale of the exact result (the parity bar for logits is 1e-3) and within 4x the direct kernel's own
    error.  Ragged planes (odd sizes, one Winograd tile wide) included."""
    g = torch.Generator().manual_seed(11)
    x = (torch.relu(torch.randn(N, H, W, C, generator=g)) * 1.3 - 0.4)          # what a BatchNorm-of-ReLU activation looks like
    w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    xd, wd = x.to(DEV), w.to(DEV)
    direct, ld = _abi_conv(xd, wd, False)
    wino, lw = _abi_conv(xd, wd, True)
    assert lw.startswith("conv_wino") and not ld.startswith("conv_wino"), (ld, lw)
    scale = float(ref.abs().max())
    e_d = float((direct.double().cpu() - ref).abs().max())
    e_w = float((wino.double().cpu() - ref).abs().max())
    print("winograd %s: scale %.3f, max error direct %.3e, winograd %.3e" % ((N, H, W, C), scale, e_d, e_w))
    assert e_w <= 1e-5 * scale, (e_w, scale)
    assert e_w <= 4 * e_d + 1e-7 * scale, (e_w, e_d)


@pytest.mark.parametrize("name,c", [("conv_64_64_s1", 64), ("conv_128_128_s1", 128)])
def test_conv_block_winograd(layer_kats, name, c, monkeypatch):
    """The Conv block KATs of the wide layers with the Winograd kernel forced for forward AND data gradient (the reference's tiny,
    odd planes: every tile ragged)."""
    import robocupvision_amd.engine as E
    monkeypatch.setattr(E, "WINOGRAD", "force")
    mod = _load_block(layer_kats, name, M.Conv(c, c, 3, 1))
    _run_block(layer_kats, name, mod)
    eng = mod.__dict__["_engine"]
    plan = [pl for (shape, training), pl in eng.plans.items() if training][0]
    assert any(l.startswith("conv_wino") for l in plan.fwd.labels(eng.handle))
    assert any(l.startswith("conv_wino") for l in plan.bwd.labels(eng.handle))
