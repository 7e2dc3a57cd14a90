"""One RCV_OP_WGRAD + RCV_OP_WGRAD_REDUCE through the C ABI on plain operands against torch's conv2d weight gradient in fp64."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robocupvision_amd import _lib as L
dev = torch.device("cuda:0"); h = L.handle(0)
g = torch.Generator().manual_seed(0)
def run(N, H, W, Cin, Cout, s, modes=False):
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    G = torch.randn(N, H, W, Cin, generator=g); P = torch.randn(N, Ho, Wo, Cout, generator=g)
    Gd, Pd = G.to(dev), P.to(dev)
    if modes:      # G: BatchNorm apply of the producer (affine), P: BN + ReLU backward (grad_enc) from (g, r) and constants
        gc = torch.rand(5, Cin, generator=g) + 0.5; pc = torch.randn(5, Cout, generator=g) * 0.5
        Pa = torch.randn(N, Ho, Wo, Cout, generator=g)
        gcd, pcd, Pad = gc.to(dev), pc.to(dev), Pa.to(dev)
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=dev); db = torch.full((Cout,), float("nan"), device=dev)
    if modes:
        op = L.make_op(L.OP_WGRAD, L.F_BIAS, n=N, h=H, w=W, cin=Cin, ho=Ho, wo=Wo, cout=Cout, stride=s, dil=1, inmode=L.LOAD_AFFINE, inmode2=L.LOAD_GRAD_ENC,
                       p_in=Gd.data_ptr(), p_in_c=gcd.data_ptr(), p_in2=Pd.data_ptr(), p_in2_aux=Pad.data_ptr(), p_in2_c=pcd.data_ptr())
        G = G.double() * gc[0].double() + gc[1].double()
        P = torch.where(Pa.double() > 0, pc[0].double() * P.double() + pc[1].double() + pc[2].double() * Pa.double(), torch.zeros((), dtype=torch.float64))
    else:
        op = L.make_op(L.OP_WGRAD, L.F_BIAS, n=N, h=H, w=W, cin=Cin, ho=Ho, wo=Wo, cout=Cout, stride=s, dil=1, inmode=L.LOAD_PLAIN, inmode2=L.LOAD_PLAIN,
                       p_in=Gd.data_ptr(), p_in2=Pd.data_ptr())
    nb = L.op_workspace(h, op)
    part = torch.zeros(max(nb // 4, 4), device=dev)
    op.p[L.RCV_P_PART] = part.data_ptr()
    red = L.make_op(L.OP_WGRAD_REDUCE, 0, cin=Cin, cout=Cout, nsplit=op.i[L.RCV_I_NSPLIT], p_part=part.data_ptr(), p_out=dw.data_ptr(), p_bias=db.data_ptr())
    lst = L.OpList([op, red]); label = lst.labels(h)[0]
    lst.run(h, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    x = G.permute(0, 3, 1, 2).double(); gy = P.permute(0, 3, 1, 2).double()
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), gy, stride=s, padding=1)
    refb = gy.sum((0, 2, 3))
    e = (dw.double().cpu() - ref).abs().max() / ref.abs().max()
    eb = (db.double().cpu() - refb).abs().max() / refb.abs().max()
    print("%-28s %s N%d %dx%d %d->%d s%d nsplit %d: dW max err / max %.2e   db %.2e %s" % (label, "modes" if modes else "plain", N, H, W, Cin, Cout, s, op.i[L.RCV_I_NSPLIT], float(e), float(eb),
          "  <-- OFF" if e > 1e-4 or eb > 1e-4 else ""))
for (N, H, W, Cin, Cout, s) in [(4, 15, 20, 64, 64, 1), (4, 30, 40, 32, 64, 2), (4, 15, 20, 64, 128, 1), (4, 15, 20, 128, 128, 1), (4, 30, 40, 32, 32, 1),
                                (4, 60, 80, 16, 16, 1), (4, 120, 160, 8, 16, 2), (2, 15, 20, 64, 64, 1), (4, 16, 20, 64, 64, 1), (4, 15, 24, 64, 64, 1),
                                (3, 5, 7, 64, 64, 1), (4, 10, 14, 128, 64, 1), (64, 15, 20, 64, 64, 1), (1, 30, 40, 64, 64, 1), (4, 7, 10, 128, 128, 1)]:
    run(N, H, W, Cin, Cout, s)
    run(N, H, W, Cin, Cout, s, True)
