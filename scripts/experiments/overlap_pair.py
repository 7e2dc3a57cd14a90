#!/usr/bin/env python3
"""Round 3: do two kernels of the backward pass overlap when one runs on the library's side stream?  Times op A alone, op B alone and
rcv_run([A on the side stream, B]) -- the forked / joined form the step uses -- for pairs (wide filter gradient, narrow data gradient).
   python scripts/experiments/overlap_pair.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robocupvision_amd import _lib as L

dev = torch.device("cuda:0"); h = L.handle(0)
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
keep = []
r4 = lambda x: (x + 3) // 4 * 4
r16 = lambda x: (x + 15) // 16 * 16


def wgrad(N, H, W, Cin, Cout, s=1, mode=L.LOAD_AFFINE, mode2=L.LOAD_GRAD_ENC):
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    G, Ga, Gc = rnd(N, H, W, Cin), rnd(N, H, W, Cin).abs(), torch.rand(5, Cin, generator=g).to(dev)
    P, Pa, Pc = rnd(N, Ho, Wo, Cout), rnd(N, Ho, Wo, Cout).abs(), torch.rand(5, Cout, generator=g).to(dev)
    op = L.make_op(L.OP_WGRAD, L.F_BIAS, n=N, h=H, w=W, cin=Cin, ho=Ho, wo=Wo, cout=Cout, stride=s, dil=1, inmode=mode, inmode2=mode2,
                   p_in=G.data_ptr(), p_in_aux=Ga.data_ptr(), p_in_c=Gc.data_ptr(), p_in2=P.data_ptr(), p_in2_aux=Pa.data_ptr(), p_in2_c=Pc.data_ptr())
    part = torch.empty(max(L.op_workspace(h, op) // 4, 4), device=dev)
    op.p[L.RCV_P_PART] = part.data_ptr()
    keep.extend([G, Ga, Gc, P, Pa, Pc, part])
    return op


def conv(kind, N, H, W, Cin, Cout, s=1, mode=L.LOAD_GRAD_ENC, stats=L.STATS_BWD_ENC, merged=0, resid=0, wino=0):
    Ho, Wo = (2 * H, 2 * W) if kind == L.OP_TCONV else ((H - 1) // s + 1, (W - 1) // s + 1)
    x, aux, c = rnd(N, H, W, Cin), rnd(N, H, W, Cin).abs(), (torch.rand(5, Cin, generator=g) + 0.5).to(dev)
    wp = rnd((16 if wino else (4 if merged else 9)) * r4(Cin) * r16(Cout * (4 if merged else 1)))
    out, ea, rs, ec, b = torch.empty(N, Ho, Wo, Cout, device=dev), rnd(N, Ho, Wo, Cout), rnd(N, Ho, Wo, Cout), torch.rand(5, Cout, generator=g).to(dev), rnd(Cout)
    op = L.make_op(kind, L.F_RESID if resid else 0, n=N, h=H, w=W, cin=Cin, cout=Cout, ho=Ho, wo=Wo, stride=(2 if kind == L.OP_TCONV else s), dil=1,
                   inmode=mode, stats=stats, aux0=(2 if wino else merged), p_in=x.data_ptr(), p_in_aux=aux.data_ptr(), p_in_c=c.data_ptr(), p_w=wp.data_ptr(),
                   p_bias=b.data_ptr(), p_out=out.data_ptr(), p_resid=rs.data_ptr(), p_epi_aux=ea.data_ptr(), p_epi_c=ec.data_ptr())
    nb = L.op_workspace(h, op)
    part = torch.empty(max(nb // 4, 4), device=dev)
    op.p[L.RCV_P_PART] = part.data_ptr()
    keep.extend([x, aux, c, wp, out, ea, rs, ec, b, part])
    return op


def timed(ops, reps=20):
    lst = L.OpList(ops)
    st = torch.cuda.current_stream()
    for _ in range(3):
        lst.run(h, st.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(st)
    for _ in range(reps):
        lst.run(h, st.cuda_stream)
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) / reps, lst.labels(h)


def side(op):
    c = L.RcvOp.from_buffer_copy(op)
    c.flags |= L.F_SIDE_STREAM
    return c


wide = {"wgrad 128->128 @30x40": lambda: wgrad(32, 30, 40, 128, 128), "wgrad 64->64 @60x80": lambda: wgrad(32, 60, 80, 64, 64)}
narrow = {"tconv 16->8 merged dgrad @240x320": lambda: conv(L.OP_TCONV, 32, 240, 320, 16, 8, merged=1, resid=0),
          "conv 16->16 dgrad @240x320": lambda: conv(L.OP_CONV, 32, 240, 320, 16, 16),
          "tconv 32->16 merged dgrad @120x160": lambda: conv(L.OP_TCONV, 32, 120, 160, 32, 16, merged=1),
          "conv 8->16 s2 dgrad(up) @480x640": lambda: conv(L.OP_CONV, 32, 480, 640, 8, 16, s=2, mode=L.LOAD_GRAD_DEC, stats=L.STATS_BWD_DEC),
          "wino 128->128 dgrad @30x40": lambda: conv(L.OP_CONV, 32, 30, 40, 128, 128, wino=1, resid=1)}
for wn, wf in wide.items():
    A = wf()
    ta, la = timed([A])
    for nn, nf in narrow.items():
        B = nf()
        tb, lb = timed([B])
        tab, _ = timed([side(A), B])
        t2, _ = timed([A, B])
        print("%-24s %.4f ms | %-36s %-22s %.4f ms | serial %.4f | A on side stream %.4f  (hidden %.0f %% of the shorter one)" %
              (wn, ta, nn, lb[0], tb, t2, tab, 100 * (t2 - tab) / min(ta, tb)))
