#!/usr/bin/env python3
"""Round 3: LabelProp frame-pair latency -- how much of the call is host enqueue time?  (eager vs a captured hipGraph of the same call)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import robocupvision_amd.model as M

dev = torch.device("cuda:0")
torch.manual_seed(12345678)
net = M.LabelProp(5, 32, 0.0).to(dev).eval()
x = torch.randn(2, 8, 120, 160, generator=torch.Generator().manual_seed(1)).to(dev)
with torch.no_grad():
    for _ in range(30):
        net(x)
    torch.cuda.synchronize()
    n = 500
    t0 = time.perf_counter()
    for _ in range(n):
        net(x)
    t_enq = time.perf_counter() - t0          # host time to enqueue n calls (the GPU may still be running)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("eager: host enqueue %.1f us per call, wall %.1f us per call" % (1e6 * t_enq / n, 1e6 * t_all / n))
    # the same call as one graph launch
    M.ALIAS_OUTPUTS = True                    # (a graph replays into fixed buffers)
    xs = x.clone()
    for _ in range(3):
        net(xs)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = net(xs)
    torch.cuda.synchronize()
    for _ in range(30):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    print("graph replay: wall %.1f us per call" % (1e6 * (time.perf_counter() - t0) / n))
    ref = net(xs).clone()
    g.replay(); torch.cuda.synchronize()
    print("graph output equals eager:", bool(torch.equal(y, ref)))
