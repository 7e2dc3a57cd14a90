#!/usr/bin/env python3
"""Benchmark of the hot path: training images/sec of ROBO-UNet 640x480, bs=32 per GPU (BASELINE.json).

    python bench.py                                   # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W     # one rank per GPU, RCCL gradient all-reduce

A "step" is the body of the reference's train.py:43-74 on one synthetic batch already resident in HBM:
zero_grad, forward (train-mode BatchNorm), weighted cross-entropy + arg-max/accuracy, backward,
decay*L1 + Adam update.  Rank 0 prints ONE JSON line.  `roofline` describes the kernel with the largest
share of the step (algorithmic FLOPs of its launches / their HIP-event time, rcv_run_timed, taken right
after the timed region on the same inputs); `cpu_baseline` is the CPU oracle (oracle/cpu_reference.py:
the reference's own operator sequence on stock PyTorch CPU kernels) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (ctor kwargs, B per GPU, H, W)
    "robo_unet_640x480_bs32": (dict(noScale=True, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128), 32, 480, 640),
    "robo_unet_160x120_bs64": (dict(noScale=False, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128), 64, 120, 160),
    "unet_640x480_bs32": (dict(noScale=True, planes=8, depth=4, levels=3, bellySize=0, bellyPlanes=128, pool=True), 32, 480, 640),
    "robo_unet_320x240_bs32": (dict(noScale=True, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128), 32, 240, 320),
    # --v2 (train.py:302-307): concatenated skips, 3x3 classifier, 9-conv belly of 64 planes
    "robo_unet_v2_640x480_bs32": (dict(noScale=True, planes=8, depth=4, levels=1, bellySize=9, bellyPlanes=64, v2=True, classSize=3),
                                  32, 480, 640),
}


def synthetic(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, H, W, generator=g)
    t = torch.randint(0, 5, (B, H, W), generator=g)
    return x, t


def pmc_traffic(label):
    """HBM bytes per launch of the kernel family `label` from the newest committed PMC pass (profiles/*_hbm_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this same command, corrected as MI355X_MICROARCH.md
    prescribes); None when no pass is on disk."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
    if not files:
        return None, None
    kern = json.load(open(files[-1]))["kernels"]
    m = re.match(r"(t?conv)(m?)(s?)_(mfma|dma)<([0-9,]+)>|(wgrad)_mfma<([0-9,]+),f(\d+)>", label)
    if not m:
        return None, None
    if m.group(6):
        base, nums = "wgrad_mfma_kernel", m.group(7).split(",") + [m.group(8)]
        want = lambda k: k.startswith(base + "<") and [x.strip() for x in k[len(base) + 1:-1].split(",")][:6] == nums
    else:
        nums = m.group(5).split(",")
        base = "convs_mfma_kernel" if m.group(3) else ("conv_dma_kernel" if m.group(4) == "dma" else "conv_mfma_kernel")
        kind = "2" if m.group(2) else ("1" if m.group(1) == "tconv" else "0")
        if m.group(3):      # convs_mfma_kernel<WM, WN, CK, KIND, XMAX, TWO>
            want = lambda k: k.startswith(base + "<") and [x.strip() for x in k[len(base) + 1:-1].split(",")][:4] == nums[:3] + [kind]
        elif m.group(4) == "dma":   # conv_dma_kernel<WM, WN, WAVES_M, WAVES_N, KIND, TWO>
            want = lambda k: k.startswith(base + "<") and [x.strip() for x in k[len(base) + 1:-1].split(",")][:5] == nums[:4] + [kind]
        else:               # conv_mfma_kernel<WM, WN, WAVES_M, WAVES_N, CK, KIND>
            want = lambda k: k.startswith(base + "<") and [x.strip() for x in k[len(base) + 1:-1].split(",")][:6] == nums[:5] + [kind]
    tot, n = 0.0, 0
    for k, v in kern.items():
        if want(k):
            tot += v["hbm_bytes_per_launch"] * v["launches"]
            n += v["launches"]
    return (tot / n if n else None), os.path.basename(files[-1])


def cpu_baseline(ctor, H, W, budget_s=20.0, dice=False):
    """The CPU oracle on this host: same step body, all host threads, bounded sample."""
    from oracle import cpu_reference as O
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except Exception:
        pass
    threads = min(threads, 16)      # the share of host cores a 1-GPU slot of the box owns (oversubscribing 256 throttles)
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(12345678)
        import robocupvision_amd.model as M
        sd = M.ROBO_UNet(**ctor).state_dict()
        cfg = O.NetConfig(**ctor)
        st = O.TrainState(sd, cfg, ce_weight=(1, 2, 6, 3, 2), use_dice=True) if dice else O.TrainState(sd, cfg)
        B = 2 if H * W >= 480 * 640 else 8
        x, t = O.synthetic_batch(B, H, W)
        O.train_step(st, x, t)                     # warm-up
        t0 = time.perf_counter()
        n = 0
        while True:
            O.train_step(st, x, t)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 20:
                break
        return {"value": round(B * n / el, 3), "unit": "img/s", "cores": threads, "kind": "port",
                "sample": "%d steps of batch %d at %dx%d, torch %s CPU, %d threads" % (n, B, W, H, torch.__version__, threads)}
    finally:
        torch.set_num_threads(old)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="robo_unet_640x480_bs32", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch size")
    ap.add_argument("--dice", action="store_true", help="train with DiceLoss (train.py --useDice) instead of the cross entropy")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed.run launcher (WORLD_SIZE=%d)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("RCV_FORCE_COLLECTIVES"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # no device_id: eager communicator binding slows EVERY kernel of the process by ~7 % on this stack (measured: 10.27 vs 9.58
        # ms/step with the collectives stubbed out); the device is fixed by torch.cuda.set_device above, barriers name it explicitly
        dist.init_process_group("nccl")

    import robocupvision_amd.model as M
    from robocupvision_amd.train import Trainer

    ctor, B, H, W = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    torch.manual_seed(12345678)
    model = M.ROBO_UNet(**ctor).to(dev)
    x, t = synthetic(B, H, W, seed=1 + rank)
    x, t = x.to(dev), t.to(dev)
    trainer = Trainer(model, class_weights=[1, 2, 6, 3, 2] if args.dice else [1, 10, 30, 10, 2], lr=1e-3, decay=1e-6,
                      distributed=dist is not None, use_dice=args.dice)
    if dist is not None:      # the first barrier builds the communicator: keep that out of the timed region
        dist.barrier(device_ids=[local_rank])
        dist.barrier(device_ids=[local_rank])

    for _ in range(args.warmup):
        trainer.step(x, t)
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(x, t)
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    metrics = trainer.pop_metrics()

    out = {
        "metric": "training images/sec, ROBO-UNet 640x480 bs=32/GPU" if args.workload == "robo_unet_640x480_bs32"
        else "training images/sec, " + args.workload,
        "value": round(B * world * args.steps / elapsed, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload, "per_gpu_batch": B, "global_batch": B * world, "height": H, "width": W,
                   "step": "fwd+CE/argmax+bwd+L1+Adam (train.py:43-74)", "parallelism": "dp%d" % world,
                   "loss_after": round(metrics["loss"], 6)},
    }

    eng = model._get_engine()
    plans = [pl for (shape, training), pl in eng.plans.items() if training and pl.side_decided]
    if plans:      # schedule of the backward list the engine measured and kept on its first pass (engine.SIDE_STREAM_MODE)
        out["config"]["filter_gradients_on_second_stream"] = bool(plans[0].side_on)
        if plans[0].side_ms:
            out["config"]["backward_ms_two_streams_vs_one"] = [round(v, 3) for v in plans[0].side_ms]
    if rank == 0 and not args.no_roofline:
        rows = model._get_engine().profile_last(reps=3)
        by = {}
        for r in rows:
            a = by.setdefault(r["label"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
            a["ms"] += r["ms"]; a["flops"] += r["flops"]; a["bytes"] += r["bytes"]; a["launches"] += 1
        total_ms = sum(a["ms"] for a in by.values())
        dom = max(by, key=lambda k: by[k]["ms"])
        d = by[dom]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                           "launches_per_step": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                           "share_of_kernel_time": round(d["ms"] / total_ms, 4),
                           "algorithmic_gbs": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1) if d["ms"] > 0 else 0.0}
        # the committed PMC pass is of the default workload: per-launch traffic of other shapes is not on disk
        tr, src = pmc_traffic(dom) if args.workload == "robo_unet_640x480_bs32" and not args.batch else (None, None)
        if tr is not None:
            out["roofline"]["traffic"] = int(tr)
            out["roofline"]["traffic_source"] = "profiles/" + src
            out["roofline"]["algorithmic_bytes_per_launch"] = int(d["bytes"] / d["launches"])
        mf = sum(a["flops"] for a in by.values())
        out["roofline"]["step_tflops"] = round(mf / (total_ms * 1e-3) / 1e12, 3)
        out["roofline"]["sum_kernel_ms"] = round(total_ms, 4)
        if args.breakdown:
            for k in sorted(by, key=lambda k: -by[k]["ms"]):
                a = by[k]
                tf = a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else 0
                gb = a["bytes"] / (a["ms"] * 1e-3) / 1e9 if a["ms"] > 0 else 0
                print("%-28s launches %3d  ms %8.3f  %5.1f%%  %7.2f TF/s  %8.1f GB/s" %
                      (k, a["launches"], a["ms"], 100 * a["ms"] / total_ms, tf, gb), file=sys.stderr)
            if os.environ.get("RCV_BENCH_ROWS"):
                # per op: measured vs the per-op roofline bound max(FLOP/157.3T, bytes/8T); sorted by the gap
                tot_gap = 0.0
                for r in sorted(rows, key=lambda r: -(r["ms"] - max(r["flops"] / 157.3e9, r["bytes"] / 8e9))):
                    lb = max(r["flops"] / 157.3e9, r["bytes"] / 8e9)
                    tot_gap += r["ms"] - lb
                    print("%s %-26s %-24s ms %7.4f  bound %7.4f  gap %7.4f  %6.2f TF/s %7.1f GB/s" %
                          ("B" if r["bwd"] else "F", r["label"], r["shape"], r["ms"], lb, r["ms"] - lb, r["flops"] / max(r["ms"], 1e-9) / 1e9,
                           r["bytes"] / max(r["ms"], 1e-9) / 1e6), file=sys.stderr)
                print("sum of per-op bounds %.3f ms, sum of gaps %.3f ms" % (sum(r["ms"] for r in rows) - tot_gap, tot_gap), file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(ctor, H, W, dice=args.dice)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
