"""One RCV_OP_PACK + RCV_OP_CONV through the C ABI (two-tensor BN/ReLU-backward load, residual add) against torch conv2d in fp64."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robocupvision_amd import _lib as L
dev = torch.device("cuda:0"); h = L.handle(0)
g = torch.Generator().manual_seed(0)
def run(N, H, W, Cin, Cout, s, wino, mode):
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(N, H, W, Cin, generator=g); xa = torch.randn(N, H, W, Cin, generator=g); c = torch.randn(5, Cin, generator=g) * 0.5
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1; resid = torch.randn(N, Ho, Wo, Cout, generator=g)
    xd, xad, cd, wd, rd = (v.to(dev) for v in (x, xa, c, w, resid))
    rp, cp = (Cin + 3) // 4 * 4, (Cout + 15) // 16 * 16
    wp = torch.zeros((16 if wino else 9) * rp * cp, device=dev)
    job = L.RcvPackJob(); job.src, job.dst, job.D0, job.D1 = wd.data_ptr(), wp.data_ptr(), Cout, Cin
    job.rows_from_d1, job.flip, job.rows_pad, job.cols_pad, job.merged = 1, 0, rp, cp, 2 if wino else 0
    table = torch.frombuffer(bytearray(bytes((L.RcvPackJob * 1)(job))), dtype=torch.uint8).to(dev)
    out = torch.full((N, Ho, Wo, Cout), float("nan"), device=dev)
    pack = L.make_op(L.OP_PACK, 0, count=1, aux0=16 * rp * cp, p_in=table.data_ptr())
    conv = L.make_op(L.OP_CONV, L.F_RESID, n=N, h=H, w=W, cin=Cin, cout=Cout, ho=Ho, wo=Wo, stride=s, dil=1, inmode=mode, aux0=2 if wino else 0,
                     p_in=xd.data_ptr(), p_in_aux=xad.data_ptr(), p_in_c=cd.data_ptr(), p_w=wp.data_ptr(), p_out=out.data_ptr(), p_resid=rd.data_ptr())
    lst = L.OpList([pack, conv]); label = lst.labels(h)[1]
    lst.run(h, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    X, A, C = x.double(), xa.double(), c.double()
    if mode == L.LOAD_GRAD_ENC: v = torch.where(A > 0, C[0] * X + C[1] + C[2] * A, torch.zeros((), dtype=torch.float64))
    elif mode == L.LOAD_AFFINE: v = X * C[0] + C[1]
    else: v = X
    ref = F.conv2d(v.permute(0, 3, 1, 2), w.double(), stride=s, padding=1).permute(0, 2, 3, 1) + resid.double()
    e = (out.double().cpu() - ref).abs().max() / ref.abs().max()
    print("%-24s N%d %dx%d %d->%d s%d mode %d: max err / max %.2e %s" % (label, N, H, W, Cin, Cout, s, mode, float(e), "  <-- OFF" if e > 1e-4 else ""))
for (N, H, W, Cin, Cout, s) in [(4, 15, 20, 64, 64, 1), (4, 15, 20, 128, 64, 1), (4, 15, 20, 128, 128, 1), (4, 30, 40, 64, 32, 1), (4, 15, 20, 64, 128, 1),
                                (4, 30, 40, 32, 64, 2), (4, 30, 40, 32, 32, 1), (4, 60, 80, 16, 16, 1), (3, 5, 7, 64, 64, 1)]:
    for mode in (L.LOAD_GRAD_ENC, L.LOAD_AFFINE):
        run(N, H, W, Cin, Cout, s, 0, mode)
        if s == 1 and Cin % 16 == 0 and Cin >= 32 and Cout >= 64: run(N, H, W, Cin, Cout, s, 1, mode)
