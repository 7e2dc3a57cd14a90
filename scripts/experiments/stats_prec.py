"""Precision of the BatchNorm partial sums a conv kernel's epilogue produces: rows summed in double vs the fp64 sum of the tensor the
same launch wrote (isolates the accumulation from everything upstream)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robocupvision_amd import _lib as L
dev = torch.device("cuda:0"); h = L.handle(0)
g = torch.Generator().manual_seed(0)
def run(kind, N, H, W, Cin, Cout, s, merged, mean):
    Ho, Wo = (2 * H, 2 * W) if kind == "tconv" else ((H - 1) // s + 1, (W - 1) // s + 1)
    x = (torch.randn(N, H, W, Cin, generator=g) + mean).to(dev)
    taps = 4 if merged else 9
    r4 = lambda v: (v + 3) // 4 * 4
    r16 = lambda v: (v + 15) // 16 * 16
    wp = (torch.randn(taps * r4(Cin) * r16(Cout * (4 if merged else 1)), generator=g) * 0.1).to(dev)
    out = torch.empty(N, Ho, Wo, Cout, device=dev)
    op = L.make_op(L.OP_TCONV if kind == "tconv" else L.OP_CONV, 0, n=N, h=H, w=W, cin=Cin, cout=Cout, ho=Ho, wo=Wo,
                   stride=(2 if kind == "tconv" else s), dil=1, inmode=L.LOAD_PLAIN, stats=L.STATS_FWD, aux0=merged,
                   p_in=x.data_ptr(), p_w=wp.data_ptr(), p_out=out.data_ptr())
    nb = L.op_workspace(h, op)
    part = torch.zeros(max(nb // 4, 4), device=dev)
    op.p[L.RCV_P_PART] = part.data_ptr()
    lst = L.OpList([op]); label = lst.labels(h)[0]
    lst.run(h, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    rows = part[: op.i[L.RCV_I_NPART] * 2 * Cout].view(-1, 2, Cout).double()
    s1 = rows[:, 0].sum(0).cpu()
    o = out.double().reshape(-1, Cout)
    ref = o.sum(0).cpu(); absum = o.abs().sum(0).cpu()
    f32 = out.reshape(-1, Cout).sum(0).double().cpu()          # torch's own fp32 reduction of the same tensor
    print("%-26s %s %dx%dx%d %d->%d: rows %d  max |s1-ref|/sum|v| %.2e   (torch fp32 sum: %.2e)   cancellation sum|v|/|sum v| median %.1f" %
          (label, kind, N, H, W, Cin, Cout, rows.shape[0], float(((s1 - ref).abs() / absum).max()), float(((f32 - ref).abs() / absum).max()),
           float((absum / ref.abs()).median())))
for mean in (0.0, 0.5):
    run("conv", 4, 120, 160, 8, 16, 2, 0, mean)
    run("conv", 4, 60, 80, 16, 16, 1, 0, mean)
    run("tconv", 4, 30, 40, 32, 16, 2, 1, mean)
    run("tconv", 2, 240, 320, 16, 8, 2, 1, mean)
    run("conv", 2, 60, 80, 64, 128, 2, 0, mean)
