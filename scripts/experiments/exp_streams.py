#!/usr/bin/env python3
"""Experiment (GPU box): does running the filter-gradient ops (wgrad + wgrad_reduce + memset) of the backward pass on a SECOND
HIP stream, overlapped with the data-gradient chain, shorten the step?  Emulated from Python with run_slice per run of ops."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import robocupvision_amd.model as M
from robocupvision_amd import _lib as L
from robocupvision_amd.train import Trainer

dev = torch.device("cuda:0")
torch.manual_seed(12345678)
model = M.ROBO_UNet(noScale=True, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128).to(dev)
B, H, W = 32, 480, 640
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 3, H, W, generator=g).to(dev); t = torch.randint(0, 5, (B, H, W), generator=g).to(dev)
tr = Trainer(model, class_weights=[1, 10, 30, 10, 2])

def bench(n=20):
    for _ in range(5): tr.step(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step(x, t)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

base = bench()
eng = model._get_engine()
side = torch.cuda.Stream(device=dev)
SIDE_KINDS = {L.OP_WGRAD, L.OP_WGRAD_REDUCE, L.OP_MEMSET}

def backward_two_streams(dlogits):
    plan, _ = eng._last
    for (idx, slot) in plan.dlogits_slots:
        plan.bwd.arr[idx].p[slot] = dlogits.data_ptr()
    main = torch.cuda.current_stream(dev)
    n = plan.bwd.n
    k = 0
    while k < n:
        is_side = plan.bwd.arr[k].kind in SIDE_KINDS
        e = k
        while e < n and (plan.bwd.arr[e].kind in SIDE_KINDS) == is_side: e += 1
        if is_side:
            side.wait_stream(main)
            plan.bwd.run_slice(eng.handle, side.cuda_stream, k, e)
        else:
            plan.bwd.run_slice(eng.handle, main.cuda_stream, k, e)
        k = e
    main.wait_stream(side)
    return plan

eng.backward = backward_two_streams
two = bench()
loss = tr.pop_metrics()["loss"]
print("one stream %.3f ms/step, wgrad on a side stream %.3f ms/step (loss %.6f)" % (base, two, loss))
