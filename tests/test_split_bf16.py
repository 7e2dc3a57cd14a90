"""The split-bf16 arithmetic of conv_bf3 / convn_bf3 / wgrad_bf3 / wgradn_bf3 (csrc/wgrad_bf3.hip has the statement).

CPU: the decomposition itself, restated in numpy -- an fp32 value is EXACTLY the sum of three bf16 values obtained by three
round-to-nearest-even steps, and the three products the kernels leave out of the nine are below 2^-23 of |a b|.
GPU: the same operator through the fp32 matrix instruction (RCV_F_MFMA_FP32) and through the split-bf16 kernels, both against fp64: the
split path may not be less accurate than the fp32 MFMA chain (it measured equal or better on every shape)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _bf16_rne(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 does for finite inputs)."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def _split3(x):
    h = _bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = _bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    return h, m, _bf16_rne(r2), r1, r2


def test_three_bf16_values_hold_an_fp32_value_exactly():
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, 200000).astype(np.float32),
                        np.array([0.0, 1.0, -1.0, 3.0e-30, 1.0e30, 0.1, 1.0 + 2.0 ** -23, 2.0 ** -100], dtype=np.float32)])
    h, m, l, r1, r2 = _split3(x)
    # every remainder is exact in fp32, and the last one IS a bf16 value: h + m + l == x in exact arithmetic
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - m.astype(np.float64))
    assert np.array_equal(l, r2)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    # magnitudes: |m| <= 2^-8 |x|, |l| <= 2^-16 |x| (round to nearest: half an ulp of an 8-bit significand, and so on)
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8) and np.all(np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_the_three_products_left_out_are_below_one_fp32_rounding():
    rng = np.random.default_rng(8)
    a, b = rng.standard_normal(100000).astype(np.float32), rng.standard_normal(100000).astype(np.float32)
    ah, am, al, _, _ = _split3(a)
    bh, bm, bl, _, _ = _split3(b)
    d = lambda v: v.astype(np.float64)
    six = d(ah) * d(bh) + d(ah) * d(bm) + d(am) * d(bh) + d(ah) * d(bl) + d(al) * d(bh) + d(am) * d(bm)     # each product exact in fp32
    exact = d(a) * d(b)
    assert np.all(np.abs(exact - six) <= np.abs(exact) * 2.0 ** -23)


gpu = pytest.mark.gpu
# Bars of the GPU comparisons, relative to the largest entry of the fp64 result: the split path within 1.5 x the error of the fp32 matrix
# instruction plus ONE fp32 ulp (max) / a quarter ulp (rms).  (Where the fp32 kernel's chains are short -- a filter gradient split over a
# thousand partial sums that are then added in double -- its error is a fraction of an ulp, and so is the split path's: 1.3e-7 against
# 1.0e-7 max, 5e-8 against 3e-8 rms on the 16 -> 16 layer; on the long chains of the wide layers the split path measured the smaller error.)
ULP = 2.0 ** -23


def _conv_err(L, h, N, H, W, Cin, Cout, s, flags, layout, gen_seed):
    import torch.nn.functional as F
    from test_gpu_kernels import _pack_dims, _rand
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(gen_seed)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x, w = _rand(gen, N, H, W, Cin), _rand(gen, Cout, Cin, 3, 3, scale=0.1)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), stride=s, padding=1).permute(0, 2, 3, 1)
    xd, wd = x.to(dev), w.to(dev)
    rp, cp, nfl = _pack_dims(Cin, Cout, layout)
    wp = torch.zeros(nfl, device=dev)
    job = L.RcvPackJob()
    job.src, job.dst, job.D0, job.D1 = wd.data_ptr(), wp.data_ptr(), Cout, Cin
    job.rows_from_d1, job.flip, job.rows_pad, job.cols_pad, job.merged = 1, 0, rp, cp, layout
    table = torch.frombuffer(bytearray(bytes((L.RcvPackJob * 1)(job))), dtype=torch.uint8).to(dev)
    out = torch.full((N, Ho, Wo, Cout), float("nan"), device=dev)
    pack = L.make_op(L.OP_PACK, 0, count=1, aux0=16 * rp * cp, p_in=table.data_ptr())
    conv = L.make_op(L.OP_CONV, flags, n=N, h=H, w=W, cin=Cin, cout=Cout, ho=Ho, wo=Wo, stride=s, dil=1, inmode=L.LOAD_PLAIN, aux0=layout,
                     p_in=xd.data_ptr(), p_w=wp.data_ptr(), p_out=out.data_ptr())
    lst = L.OpList([pack, conv])
    label = lst.labels(h)[1]
    lst.run(h, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    d = (out.double().cpu() - ref).abs()
    return label, float(d.max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.abs().max())


@gpu
@pytest.mark.parametrize("N,H,W,Cin,Cout,s", [(8, 30, 40, 128, 128, 1), (8, 60, 80, 64, 64, 1), (4, 120, 160, 32, 32, 1), (4, 240, 320, 16, 16, 1),
                                              (4, 240, 320, 16, 32, 2)])
def test_split_bf16_conv_is_as_accurate_as_the_fp32_matrix_instruction(N, H, W, Cin, Cout, s):
    from robocupvision_amd import _lib as L
    h = L.handle(0)
    l32, e32, r32 = _conv_err(L, h, N, H, W, Cin, Cout, s, L.F_MFMA_FP32, 0, 4242)
    l3, e3, r3 = _conv_err(L, h, N, H, W, Cin, Cout, s, 0, 3, 4242)
    print(".%s: max %.3e rms %.3e | %s: max %.3e rms %.3e" % (l32, e32, r32, l3, e3, r3))
    assert "_bf3" in l3 and "_bf3" not in l32
    assert e3 <= 1.5 * e32 + ULP and r3 <= 1.5 * r32 + ULP / 4, (l3, e3, r3, l32, e32, r32)


@gpu
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(8, 30, 40, 128, 128), (8, 60, 80, 64, 64), (8, 64, 128, 32, 32), (8, 64, 128, 16, 16)])
def test_split_bf16_filter_gradient_is_as_accurate_as_the_fp32_matrix_instruction(N, H, W, Cin, Cout):
    from robocupvision_amd import _lib as L
    from test_gpu_kernels import _rand
    h = L.handle(0)
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(777)
    G, P = _rand(gen, N, H, W, Cin), _rand(gen, N, H, W, Cout)
    ref = torch.nn.grad.conv2d_weight(G.double().permute(0, 3, 1, 2), (Cout, Cin, 3, 3), P.double().permute(0, 3, 1, 2), stride=1, padding=1)
    Gd, Pd = G.to(dev), P.to(dev)
    res = {}
    for name, flags in (("fp32", L.F_MFMA_FP32), ("split", 0)):
        dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=dev)
        op = L.make_op(L.OP_WGRAD, flags, n=N, h=H, w=W, cin=Cin, ho=H, wo=W, cout=Cout, stride=1, dil=1, inmode=L.LOAD_PLAIN, inmode2=L.LOAD_PLAIN,
                       p_in=Gd.data_ptr(), p_in2=Pd.data_ptr())
        part = torch.zeros(max(L.op_workspace(h, op) // 4, 4), device=dev)
        op.p[L.RCV_P_PART] = part.data_ptr()
        red = L.make_op(L.OP_WGRAD_REDUCE, 0, cin=Cin, cout=Cout, nsplit=op.i[L.RCV_I_NSPLIT], p_part=part.data_ptr(), p_out=dw.data_ptr())
        lst = L.OpList([op, red])
        lst.run(h, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        d = (dw.double().cpu() - ref).abs()
        res[name] = (lst.labels(h)[0], float(d.max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.abs().max()))
    print(".%s" % (res,))
    assert "_bf3" in res["split"][0] and "_bf3" not in res["fp32"][0]
    assert res["split"][1] <= 1.5 * res["fp32"][1] + ULP and res["split"][2] <= 1.5 * res["fp32"][2] + ULP / 4, res


def test_the_planner_routes_the_layers_of_the_headline_step_and_the_flag_switches_it_off():
    """CPU, planning handle: which records get the split-bf16 kernels at the shapes of the 640 x 480 step (256 CUs), and that
    RCV_F_MFMA_FP32 on a record keeps it on the fp32 matrix instruction (what RCV_MFMA_FP32=1 sets on every record)."""
    from robocupvision_amd import _lib as L
    h = L.planner_handle(256)

    def conv(cin, cout, hh, ww, s, flags=0, kind=None, mode=None):
        kind = L.OP_CONV if kind is None else kind
        ho, wo = (2 * hh, 2 * ww) if kind == L.OP_TCONV else ((hh - 1) // s + 1, (ww - 1) // s + 1)
        return L.make_op(kind, flags, n=32, h=hh, w=ww, cin=cin, cout=cout, ho=ho, wo=wo, stride=s, dil=1,
                         inmode=L.LOAD_AFFINE if mode is None else mode, aux0=1 if kind == L.OP_TCONV else 0)

    assert L.op_filter_layout(h, conv(128, 128, 30, 40, 1)) == 3 and L.op_filter_layout(h, conv(64, 64, 60, 80, 1)) == 3          # conv_bf3
    assert L.op_filter_layout(h, conv(32, 32, 120, 160, 1)) == 3 and L.op_filter_layout(h, conv(16, 16, 240, 320, 1)) == 3      # convn_bf3
    assert L.op_filter_layout(h, conv(32, 16, 120, 160, 2, kind=L.OP_TCONV, mode=L.LOAD_GRAD_ENC)) == 4                         # tconvn_bf3
    assert L.op_filter_layout(h, conv(8, 16, 480, 640, 2)) == 0                                      # HBM bound: stays on convs_mfma
    assert L.op_filter_layout(h, conv(128, 128, 30, 40, 1, flags=L.F_MFMA_FP32)) == 2                # Winograd on the fp32 instruction
    assert L.op_filter_layout(h, conv(32, 32, 120, 160, 1, flags=L.F_MFMA_FP32)) == 0
    for cin, cout, hh, ww, s, want in ((128, 128, 30, 40, 1, "wgrad_bf3"), (64, 64, 60, 80, 1, "wgrad_bf3"), (32, 32, 120, 160, 1, "wgradn_bf3"),
                                       (16, 32, 240, 320, 2, "wgradn_bf3"), (8, 16, 480, 640, 2, "wgrad_mfma")):
        for flags in (0, L.F_MFMA_FP32):
            op = L.make_op(L.OP_WGRAD, flags, n=32, h=hh, w=ww, cin=cin, ho=(hh - 1) // s + 1, wo=(ww - 1) // s + 1, cout=cout, stride=s, dil=1,
                           inmode=L.LOAD_AFFINE, inmode2=L.LOAD_GRAD_ENC)
            L.op_workspace(h, op)
            label = L.OpList([op]).labels(h)[0]
            assert label.startswith(want if not flags else "wgrad_mfma"), (label, want, flags)


@gpu
def test_whole_step_both_matrix_paths_against_the_fp64_oracle(monkeypatch):
    """The headline step (ROBO-UNet, 32 x 640 x 480: the only size at which every split-bf16 kernel is in the plan) lowered twice from
    the same weights and inputs -- as bench.py runs it, and with every contraction on v_mfma_f32_16x16x4_f32 (what RCV_MFMA_FP32=1 sets
    on every record) -- and the same step through the CPU oracle in float64.  Logits and loss of the two paths agree to 1e-5; their
    gradients are judged against fp64, per parameter tensor, relative to its largest entry: the split path may not be further from
    fp64 than the fp32-instruction path (the filter gradients of the BatchNorm'ed wide layers cancel to ~1e-2 of their terms, so BOTH
    paths sit at 1e-2 .. 3e-2 relative there, 7e-7 absolute: measured worst tensor 2.73e-2 split, 2.70e-2 fp32 instruction)."""
    import robocupvision_amd.model as M
    from oracle import cpu_reference as O
    from robocupvision_amd import _lib as L
    dev = "cuda:0"
    g = torch.Generator().manual_seed(5)
    x = torch.randn(32, 3, 480, 640, generator=g)
    t = torch.randint(0, 5, (32, 480, 640), generator=g)
    res = {}
    sd0 = None
    for name, flags in (("split", 0), ("fp32", L.F_MFMA_FP32)):
        monkeypatch.setattr(L, "MATRIX_FLAGS", flags)
        torch.manual_seed(12345678)
        net = M.ROBO_UNet(noScale=True)
        if sd0 is None:
            sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        net = net.to(dev)
        crit = M.CrossEntropyLoss2d(torch.tensor([1, 10, 30, 10, 2], dtype=torch.float32)).to(dev)
        net.train()
        pred = net(x.to(dev))
        loss = crit(pred, t.to(dev))
        loss.backward()
        eng = net._get_engine()
        plan = [pl for (shape, training), pl in eng.plans.items() if training][0]
        labels = plan.fwd.labels(eng.handle) + plan.bwd.labels(eng.handle)
        res[name] = (pred.detach().cpu(), float(loss.detach()), {k: p.grad.detach().cpu() for k, p in net.named_parameters()}, labels)
        del net, pred, loss
    kinds = {l.split("<")[0] for l in res["split"][3] if "_bf3" in l}
    assert {"conv_bf3", "conv2_bf3", "convn_bf3", "tconvn_bf3", "wgrad_bf3", "wgradn_bf3"} <= kinds, kinds
    assert not any("_bf3" in l for l in res["fp32"][3])
    p3, p32 = res["split"][0], res["fp32"][0]
    assert float((p3 - p32).abs().max() / p32.abs().max()) <= 1e-5
    assert abs(res["split"][1] - res["fp32"][1]) <= 2e-6 * abs(res["fp32"][1])
    # fp64: the oracle's forward on double tensors, autograd
    old_threads = torch.get_num_threads()
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    try:
        sd64 = {k: v.double().clone() for k, v in sd0.items()}
        names = O.param_names(sd64)
        for n in names:
            sd64[n].requires_grad_(True)
        cfg = O.NetConfig(noScale=True)
        pred64 = O.robo_unet_forward(sd64, x.double(), cfg, training=True)
        w = torch.tensor([1, 10, 30, 10, 2], dtype=torch.float64)
        loss64 = torch.nn.functional.cross_entropy(pred64, t, weight=w)
        loss64.backward()
    finally:
        torch.set_num_threads(old_threads)
    assert abs(res["split"][1] - float(loss64)) <= 1e-5 * abs(float(loss64))
    rows, worst3, worst32 = [], 0.0, 0.0
    for n in names:
        if n.startswith("upPart") and n.endswith("conv.bias"):
            continue            # exactly zero (bias ahead of a BatchNorm)
        g64 = sd64[n].grad
        scale = float(g64.abs().max()) + 1e-30
        e3 = float((res["split"][2][n].double() - g64).abs().max()) / scale
        e32 = float((res["fp32"][2][n].double() - g64).abs().max()) / scale
        rows.append((e3, e32, n, scale))
        worst3, worst32 = max(worst3, e3), max(worst32, e32)
    for e3, e32, n, scale in sorted(rows, reverse=True)[:6]:
        print("   %-40s against fp64: split %.3e  fp32 instruction %.3e  (largest entry %.3e)" % (n, e3, e32, scale))
    print(".whole step: loss %.9f (split) %.9f (fp32 instruction) %.9f (fp64); worst gradient error split %.3e, fp32 instruction %.3e" %
          (res["split"][1], res["fp32"][1], float(loss64), worst3, worst32))
    # Both paths carry the rounding noise of everything upstream of a gradient (fp32 activations, BatchNorm sums): per tensor the two
    # errors are two draws of the same noise (Up0.conv.weight 1.5e-2 split against 2.4e-2, PB_1.Conv2.conv.weight 1.6e-2 against
    # 0.8e-2), so the comparison is statistical -- the worst tensor, the geometric mean of the per-tensor ratios, and a loose per-tensor cap.
    import math
    ratios = [math.log((e3 + 1e-9) / (e32 + 1e-9)) for e3, e32, n, scale in rows]
    gmean = math.exp(sum(ratios) / len(ratios))
    print(".geometric mean of (split error / fp32-instruction error) over %d parameter tensors: %.3f" % (len(rows), gmean))
    assert worst3 <= 1.25 * worst32 + 1e-5, (worst3, worst32)
    assert gmean <= 1.15, gmean
    for e3, e32, n, scale in rows:
        assert e3 <= 3.0 * e32 + 1e-4, (n, e3, e32)
