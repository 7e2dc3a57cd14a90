#!/usr/bin/env python3
"""Where does the +0.7 ms of the world-size-1 collective path go?  (GPU box, launched under torch.distributed.run)"""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import robocupvision_amd.model as M
from robocupvision_amd.train import Trainer
dist.init_process_group("nccl", device_id=torch.device("cuda", 0)) if os.environ.get("DEVID") else dist.init_process_group("nccl")
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
torch.manual_seed(12345678)
model = M.ROBO_UNet(noScale=True).to(dev)
g = torch.Generator().manual_seed(1)
x = torch.randn(32, 3, 480, 640, generator=g).to(dev); t = torch.randint(0, 5, (32, 480, 640), generator=g).to(dev)
def bench(tr, n=20):
    for _ in range(5): tr.step(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step(x, t)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
tr = Trainer(model, distributed=True)
print("plain (world 1, no collectives): %.3f ms" % bench(tr))
tr.force_collectives = True
print("forced collectives:              %.3f ms" % bench(tr))
real = dist.all_reduce
dist.all_reduce = lambda *a, **k: None
print("forced, all_reduce stubbed out:  %.3f ms" % bench(tr))
dist.all_reduce = real
tr.comm_stream = None
print("forced, same-stream all_reduce:  %.3f ms" % bench(tr))
dist.destroy_process_group()
