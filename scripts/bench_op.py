#!/usr/bin/env python3
"""Micro-benchmark of ONE librcv op on synthetic buffers (GPU box).  Examples:
   python scripts/bench_op.py conv 32 30 40 128 128 --mode affine --stats fwd
   python scripts/bench_op.py conv 32 240 320 16 16 --mode grad_enc --stats bwd_enc
   python scripts/bench_op.py tconv 32 240 320 16 8 --merged 1
   python scripts/bench_op.py wgrad 32 30 40 128 128
Prints avg ms, TF/s, algorithmic GB/s.  Flags nostage/nomfma give the ablation timings."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robocupvision_amd import _lib as L

MODES = {"plain": L.LOAD_PLAIN, "affine": L.LOAD_AFFINE, "grad_enc": L.LOAD_GRAD_ENC, "grad_dec": L.LOAD_GRAD_DEC, "nchw": L.LOAD_NCHW}
STATS = {"none": L.STATS_NONE, "fwd": L.STATS_FWD, "bwd_enc": L.STATS_BWD_ENC, "bwd_dec": L.STATS_BWD_DEC}

ap = argparse.ArgumentParser()
ap.add_argument("kind", choices=["conv", "tconv", "wgrad"])
ap.add_argument("N", type=int); ap.add_argument("H", type=int); ap.add_argument("W", type=int)
ap.add_argument("Cin", type=int); ap.add_argument("Cout", type=int)
ap.add_argument("--stride", type=int, default=1); ap.add_argument("--dil", type=int, default=1)
ap.add_argument("--mode", default="affine"); ap.add_argument("--mode2", default="grad_enc")
ap.add_argument("--stats", default="none"); ap.add_argument("--merged", type=int, default=0)
ap.add_argument("--resid", type=int, default=0); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--wino", type=int, default=0, help="1 or 2: Winograd kernel (filter buffer in the transformed layout, i[AUX0] = 2); 3: split-bf16 kernel (i[AUX0] = 3)")
ap.add_argument("--stamps", type=int, default=0, help="diagnostic library build (make STAMPS=1): print the per-segment cycle shares of conv_dma_kernel")
a = ap.parse_args()
dev = torch.device("cuda:0"); h = L.handle(0)
N, H, W, Cin, Cout, s, d = a.N, a.H, a.W, a.Cin, a.Cout, a.stride, a.dil
r4 = lambda x: (x + 3) // 4 * 4
r16 = lambda x: (x + 15) // 16 * 16
g = torch.Generator(device="cpu").manual_seed(0)
rnd = lambda *shape: torch.randn(*shape, generator=g).to(dev)
if a.kind == "tconv":
    Ho, Wo = 2 * H, 2 * W
else:
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
flags = a.flags
if a.kind in ("conv", "tconv"):
    x = rnd(N, Cin, H, W) if a.mode == "nchw" else rnd(N, H, W, Cin)
    aux = rnd(N, H, W, Cin).abs()
    consts = torch.rand(5, Cin, generator=g).to(dev) + 0.5
    taps = 16 if a.wino else (4 if a.merged else 9)
    wp = rnd(taps * r4(Cin) * r16(Cout * (4 if a.merged else 1))) * 0.1
    if a.wino in (3, 5) or a.merged == 4:      # three bf16 planes [plane][k / 32][col][32]: any finite bf16 pattern will do for a timing
        ksteps = ((4 if a.merged else 9) * ((Cin + 7) // 8 * 8) + 31) // 32
        if a.wino == 5:
            ksteps = (Cin + 15) // 16 * 5
        wp = (torch.randn(3 * ksteps * r16(Cout * (4 if a.merged else 1)) * 32, generator=g) * 0.1).to(torch.bfloat16).to(dev).view(torch.float32)
    out = torch.empty(N, Ho, Wo, Cout, device=dev)
    ea = rnd(N, Ho, Wo, Cout); resid = rnd(N, Ho, Wo, Cout); ec = torch.rand(5, Cout, generator=g).to(dev)
    bias = rnd(Cout)
    op = L.make_op(L.OP_TCONV if a.kind == "tconv" else L.OP_CONV, flags | L.F_BIAS | (L.F_RESID if a.resid else 0), n=N, h=H, w=W,
                   cin=Cin, cout=Cout, ho=Ho, wo=Wo, stride=(2 if a.kind == "tconv" else s), dil=d, inmode=MODES[a.mode], stats=STATS[a.stats],
                   aux0=((2 if a.wino == 1 else a.wino) if a.wino else a.merged), p_in=x.data_ptr(), p_in_aux=aux.data_ptr(), p_in_c=consts.data_ptr(), p_w=wp.data_ptr(), p_bias=bias.data_ptr(),
                   p_out=out.data_ptr(), p_resid=resid.data_ptr(), p_epi_aux=ea.data_ptr(), p_epi_c=ec.data_ptr())
    two = a.mode in ("grad_enc", "grad_dec")
    npix_out = N * (H * W if a.kind == "tconv" else Ho * Wo)
    flops = 2.0 * 9 * Cin * Cout * npix_out
    nbytes = 4.0 * (N * H * W * Cin * (2 if two else 1) + N * Ho * Wo * Cout * (1 + a.resid + (1 if a.stats.startswith("bwd") else 0)))
else:
    # wgrad: G = gathered [N,H,W,Cin], P = pointwise [N,Ho,Wo,Cout]
    G = rnd(N, Cin, H, W) if a.mode == "nchw" else rnd(N, H, W, Cin); Ga = rnd(N, H, W, max(Cin, 4)).abs(); Gc = torch.rand(5, max(Cin, 4), generator=g).to(dev)
    P = rnd(N, Ho, Wo, Cout); Pa = rnd(N, Ho, Wo, Cout).abs(); Pc = torch.rand(5, Cout, generator=g).to(dev)
    op = L.make_op(L.OP_WGRAD, flags | L.F_BIAS, n=N, h=H, w=W, cin=Cin, ho=Ho, wo=Wo, cout=Cout, stride=s, dil=d, inmode=MODES[a.mode],
                   inmode2=MODES[a.mode2], p_in=G.data_ptr(), p_in_aux=Ga.data_ptr(), p_in_c=Gc.data_ptr(), p_in2=P.data_ptr(),
                   p_in2_aux=Pa.data_ptr(), p_in2_c=Pc.data_ptr())
    flops = 2.0 * 9 * Cin * Cout * N * Ho * Wo
    nbytes = 4.0 * (N * H * W * Cin * (2 if a.mode.startswith("grad") else 1) + N * Ho * Wo * Cout * (2 if a.mode2.startswith("grad") else 1))
stamps = None
if a.stamps:
    stamps = torch.zeros(8192 * 8 * 12, dtype=torch.int64, device=dev)
    op.p[L.RCV_P_X5] = stamps.data_ptr()
nb = L.op_workspace(h, op)
part = torch.empty(max(nb // 4, 4), device=dev)
op.p[L.RCV_P_PART] = part.data_ptr()
lst = L.OpList([op])
label = lst.labels(h)[0]
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    lst.run(h, st)
torch.cuda.synchronize()
ms = sum(lst.run_timed(h, st)[0] for _ in range(a.reps)) / a.reps
print("%-26s %s N%d %dx%d %d->%d s%d mode=%s stats=%s merged=%d tile=%s flags=%x : %.4f ms  %.2f TF/s  %.1f GB/s" %
      (label, a.kind, N, H, W, Cin, Cout, s, a.mode, a.stats, a.merged, os.environ.get("RCV_CONV_TILE", "-"), flags, ms, flops / ms / 1e9, nbytes / ms / 1e6))

if stamps is not None:
    torch.cuda.synchronize()
    st = stamps.view(-1, 12).cpu().double()
    st = st[st[:, 10] > 0]
    names = ["prologue", "loop", "epilogue", "wait_vm_lgkm", "barrier", "dma_issue", "taps0-5", "write_x+load_x", "taps5-9", "realtime(10ns)", "kernel_cycles"]
    if label.startswith("convs_") or label.startswith("tconvms_"):     # narrow kernel: phases of the tile loop
        names = ["prologue", "tile loop", "tail", "barrier A", "wait+write_x", "decode+prefetch", "barrier B", "contraction", "epilogue", "realtime(10ns)", "kernel_cycles"]
    tot = st[:, 10].mean()
    print("waves %d; mean shader cycles per wave %.0f; shader clock %.2f GHz" % (st.shape[0], tot, (st[:, 10] / (st[:, 9] * 10.0)).mean()))
    for k, nm in enumerate(names[:9]):
        print("  %-16s mean %9.0f (%5.1f%%)  min %9.0f  max %9.0f" % (nm, st[:, k].mean(), 100 * st[:, k].mean() / tot, st[:, k].min(), st[:, k].max()))
    t0 = st[:, 11]
    print("  start skew (10 ns ticks): min %.0f max %.0f" % (0, (t0.max() - t0.min())))
