"""Child process of tests/test_a_dist_gpu.py: runs the Trainer for a few steps on one rank's shard of a golden batch and saves
what the parent compares (losses, the exchanged gradient, the state dict).  Started before anything in it touches the GPU:

    python tests/dp_child.py --mode single --out F                                         (plain Trainer)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           tests/dp_child.py --mode dist --out F                                           (Trainer(distributed=True) over RCCL)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["single", "dist"], required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--tag", default="robo_s_2x48x64")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--shard", type=int, default=0, help="1: rank r trains on sample r of the golden batch (world > 1)")
    ap.add_argument("--no-overlap", action="store_true")
    a = ap.parse_args()

    import json
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("RCV_DIST_BACKEND", "nccl")      # "gloo": rehearsal with several ranks sharing this box's one GPU
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if a.mode == "dist":
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend)
    import robocupvision_amd.model as M
    from robocupvision_amd.train import Trainer

    kats = np.load(os.path.join(ROOT, "tests", "golden", "whole_net.npz"))
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "whole_net.json")))[a.tag]
    x = torch.from_numpy(kats[a.tag + "/x"])
    t = torch.from_numpy(kats[a.tag + "/t"])
    if a.shard:
        x, t = x[rank:rank + 1].contiguous(), t[rank:rank + 1].contiguous()
    x, t = x.to(dev), t.to(dev)
    torch.manual_seed(12345678)
    model = M.ROBO_UNet(**meta["ctor"]).to(dev)
    trainer = Trainer(model, class_weights=[1, 10, 30, 10, 2], lr=1e-3, decay=1e-6, distributed=a.mode == "dist",
                      overlap=not a.no_overlap)
    eng = model._get_engine()
    losses, grads, ranges = [], [], []
    for _ in range(a.steps):
        trainer.step(x, t)
        torch.cuda.synchronize()
        grads.append((eng.flat.grad * (1.0 / world)).cpu().clone())        # what the optimizer launch consumed (grad_scale = 1/world)
        m = trainer.pop_metrics()
        losses.append(m["loss"])
        if trainer.exchange is not None:
            ranges.append(list(trainer.exchange.ranges))
    out = {"losses": torch.tensor(losses, dtype=torch.float64), "grad_step0": grads[0], "grad_last": grads[-1],
           "offsets": torch.tensor(eng.flat.offsets), "names": [n for n, _ in model.named_parameters()],
           "ranges": ranges, "world": world,
           "sd": {k: v.detach().cpu() for k, v in model.state_dict().items()}}
    if rank == 0:
        torch.save(out, a.out)
    if a.mode == "dist":
        import torch.distributed as dist
        dist.barrier(device_ids=[local_rank]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
