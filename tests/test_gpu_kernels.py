"""Kernel-level accuracy through the C ABI against PyTorch CPU in float64: one op record (+ its pack / reduce) per case, plain operands
and the load transforms of the step (BatchNorm apply, BatchNorm + ReLU backward), on planes that are large, tiny and ragged.

Bars (max error relative to the largest entry of the fp64 result): convolutions 3e-6 (direct) / 2e-6 (Winograd), filter gradients 5e-7,
bias gradients 5e-7.  Measured (scripts/experiments/conv_check.py, wgrad_check.py): <= 1.1e-6, <= 4.9e-7, <= 1.7e-7, <= 1.5e-7."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _rand(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


def _load_fp64(mode, x, aux, c, L):
    X, A, C = x.double(), aux.double(), c.double()
    if mode == L.LOAD_GRAD_ENC:        # rcv.h: v = aux > 0 ? c0*x + c1 + c2*aux : 0
        return torch.where(A > 0, C[0] * X + C[1] + C[2] * A, torch.zeros((), dtype=torch.float64))
    if mode == L.LOAD_GRAD_DEC:        # v = c0*(aux*c3 + c4 > 0 ? x : 0) + c1 + c2*aux
        return C[0] * torch.where(A * C[3] + C[4] > 0, X, torch.zeros((), dtype=torch.float64)) + C[1] + C[2] * A
    if mode == L.LOAD_AFFINE:          # v = x*c0 + c1
        return X * C[0] + C[1]
    return X


def _pack_dims(rows, cols, layout, merged=False):
    """rows_pad, cols_pad, floats of the packed filter for a layout of rcv_pack_job.merged (as engine._Lowering.add_pack sizes it)."""
    split = layout in (3, 4, 5)
    rp = (rows + 31) // 32 * 32 if (split and rows > 32) else ((rows + 7) // 8 * 8 if split else (rows + 3) // 4 * 4)
    if layout == 5:
        rp = (rows + 15) // 16 * 16
    cp = (cols * (4 if merged else 1) + 15) // 16 * 16
    taps = 16 if layout == 2 else (4 if merged else 9)
    ksteps = (rp // 16) * 5 if layout == 5 else (taps * rp + 31) // 32
    return rp, cp, (3 * ksteps * cp * 16 if split else taps * rp * cp)


CONV_SHAPES = [(4, 15, 20, 64, 64, 1), (4, 15, 20, 128, 64, 1), (2, 30, 40, 128, 128, 1), (4, 30, 40, 64, 32, 1), (4, 15, 20, 64, 128, 1),
               (4, 30, 40, 32, 64, 2), (4, 30, 40, 32, 32, 1), (4, 60, 80, 16, 16, 1), (3, 5, 7, 64, 64, 1), (2, 37, 53, 32, 64, 1),
               (2, 120, 160, 8, 16, 2), (5, 9, 11, 128, 128, 1), (1, 60, 80, 64, 128, 2), (2, 48, 64, 16, 32, 2),
               (32, 30, 40, 128, 128, 1), (16, 60, 80, 64, 64, 1)]      # (the planes of the 640 x 480 step: the 320-pixel tile of conv_bf3 = its 32x32x16 form)


@pytest.mark.parametrize("N,H,W,Cin,Cout,s", CONV_SHAPES)
@pytest.mark.parametrize("mode_name", ["grad_enc", "affine"])
def test_conv_kernels_vs_fp64(N, H, W, Cin, Cout, s, mode_name):
    from robocupvision_amd import _lib as L
    h = L.handle(0)
    mode = {"grad_enc": L.LOAD_GRAD_ENC, "affine": L.LOAD_AFFINE}[mode_name]
    gen = torch.Generator().manual_seed(1000 + H * W + Cin)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x, xa, c = _rand(gen, N, H, W, Cin), _rand(gen, N, H, W, Cin), _rand(gen, 5, Cin, scale=0.5)
    w, resid = _rand(gen, Cout, Cin, 3, 3, scale=0.1), _rand(gen, N, Ho, Wo, Cout)
    ref = F.conv2d(_load_fp64(mode, x, xa, c, L).permute(0, 3, 1, 2), w.double(), stride=s, padding=1).permute(0, 2, 3, 1) + resid.double()
    xd, xad, cd, wd, rd = (v.to(DEV) for v in (x, xa, c, w, resid))
    # filter layouts (rcv_op_filter_layout): 0 plain, 2 Winograd, 3 split-bf16 (conv_bf3.hip: fp32 products as six bf16 MFMA products)
    # 3 on the wide stride-1 layers: conv_bf3.hip; on the 8 / 16 / 32 -> <= 32 channel layers (stride 1 | 2): convn_bf3.hip
    winos = [0] + ([2] if (s == 1 and Cin % 16 == 0 and Cin >= 32 and Cout >= 64) else []) + \
            ([3] if (s == 1 and Cin % 32 == 0 and Cin >= 64 and Cout >= 64) or (Cin in (8, 16, 32) and Cout <= 32) else []) + \
            ([5] if (s == 2 and Cin % 16 == 0 and Cin >= 32 and Cout >= 64) else [])       # stride-2 wide: conv2_bf3_kernel
    for wino in winos:
        rp, cp, nfl = _pack_dims(Cin, Cout, wino)
        wp = torch.zeros(nfl, device=DEV)
        job = L.RcvPackJob()
        job.src, job.dst, job.D0, job.D1 = wd.data_ptr(), wp.data_ptr(), Cout, Cin
        job.rows_from_d1, job.flip, job.rows_pad, job.cols_pad, job.merged = 1, 0, rp, cp, wino
        table = torch.frombuffer(bytearray(bytes((L.RcvPackJob * 1)(job))), dtype=torch.uint8).to(DEV)
        out = torch.full((N, Ho, Wo, Cout), float("nan"), device=DEV)
        pack = L.make_op(L.OP_PACK, 0, count=1, aux0=16 * rp * cp, p_in=table.data_ptr())
        conv = L.make_op(L.OP_CONV, L.F_RESID, n=N, h=H, w=W, cin=Cin, cout=Cout, ho=Ho, wo=Wo, stride=s, dil=1, inmode=mode,
                         aux0=wino, p_in=xd.data_ptr(), p_in_aux=xad.data_ptr(), p_in_c=cd.data_ptr(), p_w=wp.data_ptr(),
                         p_out=out.data_ptr(), p_resid=rd.data_ptr())
        lst = L.OpList([pack, conv])
        label = lst.labels(h)[1]
        assert label.startswith("conv_wino") == (wino == 2) and ("_bf3" in label) == (wino in (3, 5)), label
        lst.run(h, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
        print(".%s %s: max error %.3e" % (label, (N, H, W, Cin, Cout), err))
        assert err <= (2e-6 if wino == 2 else 3e-6), (label, err)


@pytest.mark.parametrize("N,H,W,Cin,Cout,s,d", [(2, 48, 64, 3, 8, 1, 1), (2, 120, 160, 3, 16, 1, 1), (3, 37, 53, 4, 16, 1, 1), (2, 48, 64, 3, 16, 2, 1),
                                               (2, 40, 56, 3, 8, 1, 2), (1, 33, 47, 3, 16, 2, 1), (2, 24, 40, 2, 32, 1, 2)])
def test_conv_nchw_image_input_vs_fp64(N, H, W, Cin, Cout, s, d):
    """RCV_LOAD_NCHW: the graph input in the reference's own layout (the 3-channel image) read by the first conv without an NHWC copy --
    on the first-layer kernel (8 output channels, stride 1) and on the narrow-layer kernel (the other shapes)."""
    from robocupvision_amd import _lib as L
    h = L.handle(0)
    gen = torch.Generator().manual_seed(77 + H * W + Cin)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x, w, b = _rand(gen, N, Cin, H, W), _rand(gen, Cout, Cin, 3, 3, scale=0.2), _rand(gen, Cout)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=d, dilation=d)).permute(0, 2, 3, 1)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    rp, cp = (Cin + 3) // 4 * 4, (Cout + 15) // 16 * 16
    wp = torch.zeros(9 * rp * cp, device=DEV)
    job = L.RcvPackJob()
    job.src, job.dst, job.D0, job.D1 = wd.data_ptr(), wp.data_ptr(), Cout, Cin
    job.rows_from_d1, job.flip, job.rows_pad, job.cols_pad, job.merged = 1, 0, rp, cp, 0
    table = torch.frombuffer(bytearray(bytes((L.RcvPackJob * 1)(job))), dtype=torch.uint8).to(DEV)
    out = torch.full((N, Ho, Wo, Cout), float("nan"), device=DEV)
    pack = L.make_op(L.OP_PACK, 0, count=1, aux0=9 * rp * cp, p_in=table.data_ptr())
    conv = L.make_op(L.OP_CONV, L.F_BIAS | L.F_RELU, n=N, h=H, w=W, cin=Cin, cout=Cout, ho=Ho, wo=Wo, stride=s, dil=d, inmode=L.LOAD_NCHW,
                     p_in=xd.data_ptr(), p_w=wp.data_ptr(), p_bias=bd.data_ptr(), p_out=out.data_ptr())
    lst = L.OpList([pack, conv])
    label = lst.labels(h)[1]
    lst.run(h, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= 3e-6, (label, err)


WGRAD_SHAPES = [(4, 15, 20, 64, 64, 1), (4, 30, 40, 32, 64, 2), (4, 15, 20, 64, 128, 1), (4, 15, 20, 128, 128, 1), (4, 30, 40, 32, 32, 1),
                (4, 60, 80, 16, 16, 1), (4, 120, 160, 8, 16, 2), (2, 15, 20, 64, 64, 1), (4, 16, 20, 64, 64, 1), (4, 15, 24, 64, 64, 1),
                (3, 5, 7, 64, 64, 1), (4, 10, 14, 128, 64, 1), (64, 15, 20, 64, 64, 1), (1, 30, 40, 64, 64, 1), (4, 7, 10, 128, 128, 1),
                (2, 37, 53, 16, 32, 1), (2, 48, 64, 8, 8, 1), (3, 33, 47, 32, 16, 1), (2, 60, 80, 32, 64, 2), (2, 120, 160, 16, 32, 2),
                (3, 14, 18, 64, 128, 2), (1, 96, 128, 8, 16, 2),
                # planes with at least one 8 x 16 tile per CU: the narrow layers' split-bf16 kernel (wgradn_bf3.hip), every channel-tile shape
                (8, 64, 128, 16, 16, 1), (8, 64, 128, 32, 32, 1), (8, 128, 128, 16, 16, 2), (8, 128, 128, 16, 32, 2), (6, 60, 100, 32, 16, 1),
                (5, 72, 112, 16, 32, 1), (8, 128, 128, 32, 32, 2), (6, 62, 100, 16, 24, 1), (8, 128, 128, 32, 16, 2),
                # eight gathered channels, stride 2: the pixel-pair view of wgradn_bf3
                (8, 128, 128, 8, 16, 2), (8, 128, 128, 8, 32, 2), (6, 126, 140, 8, 16, 2), (7, 100, 132, 8, 24, 2)]


@pytest.mark.parametrize("N,H,W,Cin,Cout,s", WGRAD_SHAPES)
@pytest.mark.parametrize("modes", [False, True, "dec"])
def test_filter_gradient_kernels_vs_fp64(N, H, W, Cin, Cout, s, modes):
    from robocupvision_amd import _lib as L
    h = L.handle(0)
    gen = torch.Generator().manual_seed(2000 + H * W + Cout)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    G, P = _rand(gen, N, H, W, Cin), _rand(gen, N, Ho, Wo, Cout)
    gc, pc, Pa = torch.rand(5, Cin, generator=gen) + 0.5, _rand(gen, 5, Cout, scale=0.5), _rand(gen, N, Ho, Wo, Cout)
    Gd, Pd, gcd, pcd, Pad = (v.to(DEV) for v in (G, P, gc, pc, Pa))
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=DEV)
    db = torch.full((Cout,), float("nan"), device=DEV)
    if modes == "dec":    # transposed-conv layer: gathered operand = the two-tensor gradient at the 2x plane, pointwise = the layer input
        if s != 2:
            pytest.skip("the transposed-conv form is the stride-2 one")
        Ga = _rand(gen, N, H, W, Cin)
        gc = _rand(gen, 5, Cin, scale=0.5)
        Gad, gcd = Ga.to(DEV), gc.to(DEV)
        pc = torch.rand(5, Cout, generator=gen) + 0.5
        pcd = pc.to(DEV)
        op = L.make_op(L.OP_WGRAD, 0, n=N, h=H, w=W, cin=Cin, ho=Ho, wo=Wo, cout=Cout, stride=2, dil=1, inmode=L.LOAD_GRAD_DEC,
                       inmode2=L.LOAD_AFFINE, p_in=Gd.data_ptr(), p_in_aux=Gad.data_ptr(), p_in_c=gcd.data_ptr(), p_in2=Pd.data_ptr(),
                       p_in2_c=pcd.data_ptr())
        x64, gy64 = _load_fp64(L.LOAD_GRAD_DEC, G, Ga, gc, L), _load_fp64(L.LOAD_AFFINE, P, P, pc, L)
    elif modes:      # gathered operand: BatchNorm apply of the producer; pointwise operand: BatchNorm + ReLU backward of (g, r)
        op = L.make_op(L.OP_WGRAD, L.F_BIAS, n=N, h=H, w=W, cin=Cin, ho=Ho, wo=Wo, cout=Cout, stride=s, dil=1, inmode=L.LOAD_AFFINE,
                       inmode2=L.LOAD_GRAD_ENC, p_in=Gd.data_ptr(), p_in_c=gcd.data_ptr(), p_in2=Pd.data_ptr(), p_in2_aux=Pad.data_ptr(),
                       p_in2_c=pcd.data_ptr())
        x64, gy64 = _load_fp64(L.LOAD_AFFINE, G, G, gc, L), _load_fp64(L.LOAD_GRAD_ENC, P, Pa, pc, L)
    else:
        op = L.make_op(L.OP_WGRAD, L.F_BIAS, n=N, h=H, w=W, cin=Cin, ho=Ho, wo=Wo, cout=Cout, stride=s, dil=1, inmode=L.LOAD_PLAIN,
                       inmode2=L.LOAD_PLAIN, p_in=Gd.data_ptr(), p_in2=Pd.data_ptr())
        x64, gy64 = G.double(), P.double()
    nb = L.op_workspace(h, op)
    part = torch.zeros(max(nb // 4, 4), device=DEV)
    op.p[L.RCV_P_PART] = part.data_ptr()
    has_bias = bool(op.flags & L.F_BIAS)
    red = L.make_op(L.OP_WGRAD_REDUCE, 0, cin=Cin, cout=Cout, nsplit=op.i[L.RCV_I_NSPLIT], p_part=part.data_ptr(), p_out=dw.data_ptr(),
                    p_bias=db.data_ptr() if has_bias else 0)
    lst = L.OpList([op, red])
    lst.run(h, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x64.permute(0, 3, 1, 2), (Cout, Cin, 3, 3), gy64.permute(0, 3, 1, 2), stride=s, padding=1)
    refb = gy64.sum((0, 1, 2))
    e = float((dw.double().cpu() - ref).abs().max() / ref.abs().max())
    eb = float((db.double().cpu() - refb).abs().max() / refb.abs().max()) if has_bias else 0.0
    print(".%s %s %s: max error %.3e (bias %.3e)" % (lst.labels(h)[0], (N, H, W, Cin, Cout, s), modes, e, eb))
    assert e <= 5e-7 and eb <= 5e-7, (lst.labels(h)[0], e, eb)


@pytest.mark.parametrize("N,H,W,Cin,Cout,s,d", [(2, 48, 64, 3, 8, 1, 1), (2, 40, 56, 3, 8, 1, 2), (2, 48, 64, 3, 16, 1, 1), (2, 37, 53, 4, 16, 1, 1),
                                               (2, 48, 64, 3, 32, 1, 1), (2, 48, 64, 4, 8, 1, 1), (1, 96, 128, 3, 16, 2, 1), (3, 24, 40, 2, 8, 1, 1)])
@pytest.mark.parametrize("two", [False, True])
def test_filter_gradient_nchw_image_vs_fp64(N, H, W, Cin, Cout, s, d, two):
    """Filter gradient of a first layer: the gathered operand is the NCHW image itself (RCV_LOAD_NCHW, <= 4 channels) -- the vector-ALU
    first-layer kernel (<= 3 channels into 8), the 2-block folded MFMA tile, and the 16-wide gathered tiles (4 channels, or more than 16
    output channels); the pointwise operand plain or the two-tensor BatchNorm + ReLU backward load."""
    from robocupvision_amd import _lib as L
    h = L.handle(0)
    gen = torch.Generator().manual_seed(3000 + H * W + Cout + Cin)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    G, P = _rand(gen, N, Cin, H, W), _rand(gen, N, Ho, Wo, Cout)
    pc, Pa = _rand(gen, 5, Cout, scale=0.5), _rand(gen, N, Ho, Wo, Cout)
    Gd, Pd, pcd, Pad = (v.to(DEV) for v in (G, P, pc, Pa))
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=DEV)
    db = torch.full((Cout,), float("nan"), device=DEV)
    mode2 = L.LOAD_GRAD_ENC if two else L.LOAD_PLAIN
    op = L.make_op(L.OP_WGRAD, L.F_BIAS, n=N, h=H, w=W, cin=Cin, ho=Ho, wo=Wo, cout=Cout, stride=s, dil=d, inmode=L.LOAD_NCHW, inmode2=mode2,
                   p_in=Gd.data_ptr(), p_in2=Pd.data_ptr(), p_in2_aux=Pad.data_ptr(), p_in2_c=pcd.data_ptr())
    gy64 = _load_fp64(mode2, P, Pa, pc, L)
    nb = L.op_workspace(h, op)
    part = torch.zeros(max(nb // 4, 4), device=DEV)
    op.p[L.RCV_P_PART] = part.data_ptr()
    red = L.make_op(L.OP_WGRAD_REDUCE, 0, cin=Cin, cout=Cout, nsplit=op.i[L.RCV_I_NSPLIT], p_part=part.data_ptr(), p_out=dw.data_ptr(), p_bias=db.data_ptr())
    lst = L.OpList([op, red])
    lst.run(h, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(G.double(), (Cout, Cin, 3, 3), gy64.permute(0, 3, 1, 2), stride=s, padding=d, dilation=d)
    refb = gy64.sum((0, 1, 2))
    e = float((dw.double().cpu() - ref).abs().max() / ref.abs().max())
    eb = float((db.double().cpu() - refb).abs().max() / refb.abs().max())
    print(".%s %s: max error %.3e (bias %.3e)" % (lst.labels(h)[0], (N, H, W, Cin, Cout, s, d, two), e, eb))
    assert e <= 5e-7 and eb <= 5e-7, (lst.labels(h)[0], e, eb)


TCONV_SHAPES = [(2, 15, 20, 128, 64), (2, 30, 40, 64, 32), (2, 60, 80, 32, 16), (2, 120, 160, 16, 8), (3, 7, 9, 128, 64), (1, 33, 21, 64, 32),
                (2, 24, 32, 32, 16), (4, 8, 10, 64, 64), (2, 20, 28, 16, 16), (1, 5, 6, 128, 128)]


@pytest.mark.parametrize("N,H,W,Cin,Cout", TCONV_SHAPES)
@pytest.mark.parametrize("mode_name", ["affine", "grad_enc"])
def test_transposed_conv_kernels_vs_fp64(N, H, W, Cin, Cout, mode_name):
    """ConvTranspose2d(k3, s2, p1, output_padding 1) in its three kernel forms (all-parity LDS-DMA tile, merged-parity narrow tile,
    phase-split fallback) -- the forward of the decoder blocks and the data gradient of the stride-2 convs: 3e-6 of the fp64 result."""
    from robocupvision_amd import _lib as L
    from robocupvision_amd.engine import MERGED_TCONV_MAX_COUT
    h = L.handle(0)
    mode = {"grad_enc": L.LOAD_GRAD_ENC, "affine": L.LOAD_AFFINE}[mode_name]
    gen = torch.Generator().manual_seed(3000 + H * W + Cin)
    x, xa, c = _rand(gen, N, H, W, Cin), _rand(gen, N, H, W, Cin), _rand(gen, 5, Cin, scale=0.5)
    w, bias = _rand(gen, Cin, Cout, 3, 3, scale=0.1), _rand(gen, Cout)
    ref = F.conv_transpose2d(_load_fp64(mode, x, xa, c, L).permute(0, 3, 1, 2), w.double(), bias.double(), stride=2, padding=1,
                             output_padding=1).permute(0, 2, 3, 1)
    xd, xad, cd, wd, bd = (v.to(DEV) for v in (x, xa, c, w, bias))
    merged = 1 if Cout <= MERGED_TCONV_MAX_COUT else 0
    # merged layout as it is (fp32 kernels) and, where the split-bf16 narrow kernel is built, split into bf16 (layout 4: tconvn_bf3)
    layouts = [merged] + ([4] if merged and (Cin, (4 * Cout + 15) // 16 * 16) in ((16, 32), (32, 64)) else [])
    for layout in layouts:
        rp, cp, nfl = _pack_dims(Cin, Cout, layout, merged=bool(merged))
        wp = torch.zeros(nfl, device=DEV)
        job = L.RcvPackJob()
        job.src, job.dst, job.D0, job.D1 = wd.data_ptr(), wp.data_ptr(), Cin, Cout
        job.rows_from_d1, job.flip, job.rows_pad, job.cols_pad, job.merged = 0, 0, rp, cp, layout
        table = torch.frombuffer(bytearray(bytes((L.RcvPackJob * 1)(job))), dtype=torch.uint8).to(DEV)
        out = torch.full((N, 2 * H, 2 * W, Cout), float("nan"), device=DEV)
        pack = L.make_op(L.OP_PACK, 0, count=1, aux0=9 * rp * cp, p_in=table.data_ptr())
        tconv = L.make_op(L.OP_TCONV, L.F_BIAS, n=N, h=H, w=W, cin=Cin, cout=Cout, ho=2 * H, wo=2 * W, stride=2, dil=1, inmode=mode, aux0=layout,
                          p_in=xd.data_ptr(), p_in_aux=xad.data_ptr(), p_in_c=cd.data_ptr(), p_w=wp.data_ptr(), p_bias=bd.data_ptr(),
                          p_out=out.data_ptr())
        lst = L.OpList([pack, tconv])
        label = lst.labels(h)[1]
        assert ("_bf3" in label) == (layout == 4), label
        lst.run(h, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
        print(".%s %s: max error %.3e" % (label, (N, H, W, Cin, Cout), err))
        assert err <= 3e-6, (label, err)


@pytest.mark.parametrize("N,H,W,C", [(2, 8, 64, 8), (2, 6, 32, 16), (1, 4, 16, 32), (3, 10, 128, 8), (2, 6, 24, 8), (1, 4, 8, 64), (2, 4, 6, 16)])
@pytest.mark.parametrize("affine", [False, True])
def test_maxpool_backward_first_maximum_exact(N, H, W, C, affine):
    """RCV_OP_POOL_BWD: the pooled gradient goes to the FIRST maximum of each 2x2 window (window order (0,0), (0,1), (1,0), (1,1), as
    aten::max_pool2d_with_indices), bit for bit, on inputs full of ties -- through the row-mapped kernel (C/4 a power of two, W*C/4 a
    multiple of 64) and through the pixel-mapped one (the other shapes); with the producer's BatchNorm folded into the comparison
    (positive and negative scales) and a residual added to the result."""
    from robocupvision_amd import _lib as L
    h = L.handle(0)
    gen = torch.Generator().manual_seed(5 + H * W + C)
    x = torch.randint(-3, 4, (N, H, W, C), generator=gen).float()          # few distinct values: most windows hold ties
    g = _rand(gen, N, H // 2, W // 2, C)
    resid = _rand(gen, N, H, W, C)
    c = torch.zeros(5, C)
    c[0] = torch.where(torch.arange(C) % 3 == 0, -0.5, 2.0) if affine else 1.0
    c[1] = _rand(gen, C) if affine else 0.0
    xv = (x * c[0] + c[1]).permute(0, 3, 1, 2).double().requires_grad_(True)
    y = F.max_pool2d(xv, 2, 2)
    y.backward(g.permute(0, 3, 1, 2).double())
    ref = (xv.grad.permute(0, 2, 3, 1) + resid.double()).float()
    xd, gd, rd, cd = x.to(DEV), g.to(DEV), resid.to(DEV), c.to(DEV)
    out = torch.full((N, H, W, C), float("nan"), device=DEV)
    op = L.make_op(L.OP_POOL_BWD, L.F_RESID, n=N, h=H, w=W, cout=C, inmode=L.LOAD_AFFINE if affine else L.LOAD_PLAIN, stats=L.STATS_NONE,
                   p_in=gd.data_ptr(), p_epi_aux=xd.data_ptr(), p_in_c=cd.data_ptr(), p_resid=rd.data_ptr(), p_out=out.data_ptr())
    L.OpList([op]).run(h, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("N,H,W,Cin,Cout,s,d", [(2, 15, 20, 32, 64, 1, 2), (2, 15, 20, 64, 64, 1, 2), (2, 15, 20, 64, 32, 1, 2), (1, 7, 9, 128, 64, 1, 1),
                                               (3, 9, 11, 16, 48, 2, 1), (1, 30, 40, 64, 128, 1, 1), (2, 5, 3, 20, 36, 1, 2)])
@pytest.mark.parametrize("mode_name,flags", [("affine_relu", 0), ("affine", 3), ("plain", 1)])
def test_tiny_plane_conv_vs_fp64(N, H, W, Cin, Cout, s, d, mode_name, flags):
    """conv_small.hip (inference on planes of a few hundred pixels: LabelProp's dilated convs for one frame pair, model.py:546-548): one
    MFMA block per workgroup, K split over its four waves, operands straight from global memory.  Zero padding applies AFTER the load
    transform (a padded tap contributes 0, not the BatchNorm shift)."""
    from robocupvision_amd import _lib as L
    h = L.handle(0)
    mode = {"plain": L.LOAD_PLAIN, "affine": L.LOAD_AFFINE, "affine_relu": L.LOAD_AFFINE_RELU}[mode_name]
    gen = torch.Generator().manual_seed(31 + H * W + Cin + Cout)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x, c = _rand(gen, N, H, W, Cin), _rand(gen, 5, Cin, scale=0.7)
    w, b = _rand(gen, Cout, Cin, 3, 3, scale=0.1), _rand(gen, Cout)
    X = x.double()
    if mode != L.LOAD_PLAIN:
        X = X * c[0].double() + c[1].double()
    if mode == L.LOAD_AFFINE_RELU:
        X = F.relu(X)
    use_bias, use_relu = bool(flags & 1), bool(flags & 2)
    ref = F.conv2d(X.permute(0, 3, 1, 2), w.double(), b.double() if use_bias else None, stride=s, padding=d, dilation=d).permute(0, 2, 3, 1)
    if use_relu:
        ref = F.relu(ref)
    xd, cd, wd, bd = (v.to(DEV) for v in (x, c, w, b))
    rp, cp = (Cin + 3) // 4 * 4, (Cout + 15) // 16 * 16
    wp = torch.zeros(9 * rp * cp, device=DEV)
    job = L.RcvPackJob()
    job.src, job.dst, job.D0, job.D1 = wd.data_ptr(), wp.data_ptr(), Cout, Cin
    job.rows_from_d1, job.flip, job.rows_pad, job.cols_pad, job.merged = 1, 0, rp, cp, 0
    table = torch.frombuffer(bytearray(bytes((L.RcvPackJob * 1)(job))), dtype=torch.uint8).to(DEV)
    out = torch.full((N, Ho, Wo, Cout), float("nan"), device=DEV)
    pack = L.make_op(L.OP_PACK, 0, count=1, aux0=9 * rp * cp, p_in=table.data_ptr())
    conv = L.make_op(L.OP_CONV, (L.F_BIAS if use_bias else 0) | (L.F_RELU if use_relu else 0), n=N, h=H, w=W, cin=Cin, cout=Cout, ho=Ho, wo=Wo,
                     stride=s, dil=d, inmode=mode, p_in=xd.data_ptr(), p_in_c=cd.data_ptr(), p_w=wp.data_ptr(), p_bias=bd.data_ptr(),
                     p_out=out.data_ptr())
    lst = L.OpList([pack, conv])
    label = lst.labels(h)[1]
    assert label.startswith("conv_small"), label
    lst.run(h, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= 3e-6, (label, err)
