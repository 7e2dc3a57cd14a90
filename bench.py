#!/usr/bin/env python3
"""Benchmark of the hot path: training images/sec of ROBO-UNet 640x480, bs=32 per GPU (BASELINE.json).

    python bench.py                                   # 1 GPU
    python bench.py --gpus N --steps K --warmup W     # N GPUs of this node: starts the N ranks below itself (fresh child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W     # one rank per GPU, RCCL gradient all-reduce

A "step" is the body of the reference's train.py:43-74 on one synthetic batch already resident in HBM:
zero_grad, forward (train-mode BatchNorm), weighted cross-entropy + arg-max/accuracy, backward,
decay*L1 + Adam update.  Rank 0 prints ONE JSON line.

`roofline` describes the kernel family with the largest share of the step: its algorithmic FLOPs and bytes (engine.op_work: SURVEY
8(d)'s per-layer figures) over the HIP-event time of its launches (rcv_run_timed on the stream the kernels run on, taken right after
the timed region on the same buffers).  The roof is chosen per kernel from its arithmetic intensity: fp32-MFMA peak when
FLOPs/157.3T >= bytes/8T, HBM otherwise.  `t_roof_ms` = sum over ALL ops of the step of max(FLOPs/peak, bytes/HBM peak) -- the
per-layer roofline of SURVEY 8(d) -- and `step_frac` = t_roof_ms / ms_per_step.  `cpu_baseline` is the CPU oracle
(oracle/cpu_reference.py: the reference's own operator sequence on stock PyTorch CPU kernels) timed on this host on a bounded sample.

Other workloads (--workload): BASELINE.json configs 2 (robo_unet_160x120_bs64), 3 (unet_640x480_bs32) and 5 (labelprop_160x120_b2 /
_b64: LabelProp frame-pair inference, validLabelProp.py:132-135; a step = one forward call, latency = ms_per_step).
"""
import argparse
import glob
import json
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0
# Split-bf16 kernels (conv_bf3 / wgrad_bf3: every fp32 multiply-add = six v_mfma_f32_16x16x32_bf16 products of operands split exactly into
# three bf16 values): their matrix-pipe ceiling in ALGORITHMIC fp32 FLOPs is the dense bf16 peak / 6
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_SPLIT_BF16_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def mfma_peak(label: str) -> float:
    return PEAK_SPLIT_BF16_TFLOPS if "_bf3" in label else PEAK_FP32_MFMA_TFLOPS

WORKLOADS = {
    # name: (ctor kwargs, B per GPU, H, W)
    "robo_unet_640x480_bs32": (dict(noScale=True, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128), 32, 480, 640),
    "robo_unet_160x120_bs64": (dict(noScale=False, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128), 64, 120, 160),
    "unet_640x480_bs32": (dict(noScale=True, planes=8, depth=4, levels=3, bellySize=0, bellyPlanes=128, pool=True), 32, 480, 640),
    "robo_unet_320x240_bs32": (dict(noScale=True, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128), 32, 240, 320),
    # --v2 (train.py:302-307): concatenated skips, 3x3 classifier, 9-conv belly of 64 planes
    "robo_unet_v2_640x480_bs32": (dict(noScale=True, planes=8, depth=4, levels=1, bellySize=9, bellyPlanes=64, v2=True, classSize=3),
                                  32, 480, 640),
    # BASELINE config 5: LabelProp(5, 32) inference on 8-channel frame-pair inputs (labelPropTrain.py:178-182); B = 2 is one pair
    "labelprop_160x120_b2": (None, 2, 120, 160),
    "labelprop_160x120_b64": (None, 64, 120, 160),
}


def synthetic(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, H, W, generator=g)
    t = torch.randint(0, 5, (B, H, W), generator=g)
    return x, t


def pmc_traffic(label, workload):
    """HBM bytes per launch of the kernel family `label` from the newest committed PMC pass of this workload
    (profiles/*<workload>*_hbm_traffic.json, or profiles/*_hbm_traffic.json for the headline: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    in separate passes of this same command, corrected as MI355X_MICROARCH.md prescribes); None when no pass is on disk."""
    if workload == "robo_unet_640x480_bs32":
        files = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
                 if not any(w in os.path.basename(f) for w in WORKLOADS if w != workload)]
    else:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*%s*_hbm_traffic.json" % workload)))
    if not files:
        return None, None
    kern = json.load(open(files[-1]))["kernels"]
    m = re.match(r"(t?conv)([ma]?)(s?)_(mfma|dma)<([0-9,]+)>|(wgrad)_mfma<([0-9,]+),f(\d+)>", label)
    if not m:
        # small kernels: the family name is a prefix of the kernel symbol (cls_fwd -> cls_fwd_kernel<...>)
        want = lambda k: k.startswith(label + "_kernel") or k.startswith(label.split("<")[0] + "_kernel")
    elif m.group(6):
        base, nums = "wgrad_mfma_kernel", m.group(7).split(",") + [m.group(8)]
        want = lambda k: k.startswith(base + "<") and [x.strip() for x in k[len(base) + 1:-1].split(",")][:6] == nums
    else:
        nums = m.group(5).split(",")
        base = "convs_mfma_kernel" if m.group(3) else ("conv_dma_kernel" if m.group(4) == "dma" else "conv_mfma_kernel")
        kind = {"": "1" if m.group(1) == "tconv" else "0", "m": "2", "a": "3"}[m.group(2)]
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


def _host_threads():
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return min(threads, 16)      # the share of host cores a 1-GPU slot of the box owns (oversubscribing 256 throttles)


def cpu_baseline(ctor, H, W, budget_s=15.0, dice=False):
    """The CPU oracle on this host: same step body, all host threads, bounded sample."""
    from oracle import cpu_reference as O
    threads = _host_threads()
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
            if el > budget_s or (n >= 20 and el >= 10.0):      # 10-15 s of CPU work
                break
        return {"value": round(B * n / el, 3), "unit": "img/s", "cores": threads, "kind": "port",
                "sample": "%d steps of batch %d at %dx%d, torch %s CPU, %d threads" % (n, B, W, H, torch.__version__, threads)}
    finally:
        torch.set_num_threads(old)


def cpu_baseline_labelprop(sd, B, H, W, budget_s=15.0):
    from oracle import cpu_reference as O
    threads = _host_threads()
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        x = torch.randn(B, 8, H, W, generator=torch.Generator().manual_seed(1))
        with torch.no_grad():
            O.labelprop_forward(sd, x)
            t0 = time.perf_counter()
            n = 0
            while True:
                O.labelprop_forward(sd, x)
                n += 1
                el = time.perf_counter() - t0
                if el > budget_s or n >= 200:
                    break
        return {"value": round(B * n / el, 3), "unit": "img/s", "cores": threads, "kind": "port",
                "sample": "%d eval-mode forwards of batch %d at %dx%d, torch %s CPU, %d threads" % (n, B, W, H, torch.__version__, threads)}
    finally:
        torch.set_num_threads(old)


def kernel_family(label: str) -> str:
    """Source kernel (__global__ template) behind an op label: conv_dma<2,5,4,1,4> and tconva_dma<...> are both conv_dma_kernel."""
    m = re.match(r"t?conv[ma]?(s?)_(mfma|dma)<", label)
    if m:
        return "convs_mfma_kernel" if m.group(1) else "conv_%s_kernel" % m.group(2)
    return label.split("<")[0] + "_kernel"


def roofline_block(rows, ms_per_step, workload, batch_override, breakdown):
    """Per-kernel-family table from engine.profile_last rows -> the `roofline` object (dominant family) + per-layer roofline sums."""
    by = {}
    for r in rows:
        a = by.setdefault(r["label"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
        a["ms"] += r["ms"]; a["flops"] += r["flops"]; a["bytes"] += r["bytes"]; a["launches"] += 1
    total_ms = sum(a["ms"] for a in by.values())
    bound_ms = lambda fl, by_, label="": max(fl / (mfma_peak(label) * 1e9), by_ / (PEAK_HBM_GBS * 1e6))
    t_roof = sum(bound_ms(r["flops"], r["bytes"], r["label"]) for r in rows)
    t_roof_fp32 = sum(bound_ms(r["flops"], r["bytes"]) for r in rows)      # every contraction priced on the fp32 matrix instruction (rounds 1-2)
    dom = max(by, key=lambda k: by[k]["ms"])
    d = by[dom]
    dom_peak = mfma_peak(dom)
    mfma_bound = d["flops"] / (dom_peak * 1e12) >= d["bytes"] / (PEAK_HBM_GBS * 1e9)
    tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
    gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
    out = {"bound": "mfma" if mfma_bound else "hbm", "kernel": dom,
           "achieved": round(tf if mfma_bound else gbs, 3), "peak": round(dom_peak, 1) if mfma_bound else PEAK_HBM_GBS,
           "unit": "TFLOP/s" if mfma_bound else "GB/s",
           "frac": round((tf / dom_peak) if mfma_bound else (gbs / PEAK_HBM_GBS), 4), "traffic": None,
           "launches_per_step": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4),
           "share_of_kernel_time": round(d["ms"] / total_ms, 4),
           "algorithmic_tflops": round(tf, 3), "algorithmic_gbs": round(gbs, 1),
           "arithmetic_intensity_flop_per_byte": round(d["flops"] / d["bytes"], 2) if d["bytes"] else None,
           "algorithmic_bytes_per_launch": int(d["bytes"] / d["launches"])}
    # Pricing: FLOPs are those of the DIRECT convolution (the roofline is algorithmic).  A Winograd F(2x2,3x3) kernel executes
    # 16/36 of those multiplications (plus tile padding), so for that family `frac` is the fraction of the roofline TIME, not matrix-pipe
    # utilisation: `executed_tflops` / `mfma_pipe_frac` carry the latter.
    out["flop_pricing"] = "direct-conv 2*MAC (algorithmic)"
    if dom.startswith("conv_wino"):
        out["flop_pricing"] += "; conv_wino executes 16/36 of them on the matrix pipe"
        out["executed_tflops"] = round(tf * 16.0 / 36.0, 3)
        out["mfma_pipe_frac"] = round(tf * 16.0 / 36.0 / PEAK_FP32_MFMA_TFLOPS, 4)
    elif "_bf3" in dom:
        out["flop_pricing"] += ("; split-bf16 kernel: six bf16 MFMA products per fp32 multiply-add, peak = %.0f / 6 TFLOP/s of the dense bf16 pipe; "
                                "%.3f of the fp32 matrix-instruction peak" % (PEAK_BF16_MFMA_TFLOPS, tf / PEAK_FP32_MFMA_TFLOPS))
        out["executed_tflops"] = round(tf * 6.0, 3)
        out["mfma_pipe_frac"] = round(tf * 6.0 / PEAK_BF16_MFMA_TFLOPS, 4)
    elif mfma_bound:
        out["executed_tflops"] = round(tf, 3)
        out["mfma_pipe_frac"] = out["frac"]
    # the same figures per SOURCE kernel (all instantiations of one __global__ template together), largest first
    fam = {}
    for r in rows:
        a = fam.setdefault(kernel_family(r["label"]), {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "roof_ms": 0.0})
        a["ms"] += r["ms"]; a["flops"] += r["flops"]; a["bytes"] += r["bytes"]; a["launches"] += 1
        a["roof_ms"] += bound_ms(r["flops"], r["bytes"], r["label"])
    top = []
    for k in sorted(fam, key=lambda k: -fam[k]["ms"])[:8]:
        a = fam[k]
        if a["ms"] <= 0:
            continue
        fb = a["flops"] / (mfma_peak(k) * 1e12) >= a["bytes"] / (PEAK_HBM_GBS * 1e9)
        top.append({"kernel": k, "launches_per_step": a["launches"], "ms_per_step": round(a["ms"], 4), "share": round(a["ms"] / total_ms, 4),
                    "bound": "mfma" if fb else "hbm", "frac_of_own_roof": round(a["roof_ms"] / a["ms"], 4),
                    "algorithmic_tflops": round(a["flops"] / (a["ms"] * 1e-3) / 1e12, 2),
                    "algorithmic_gbs": round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1)})
    out["top_families"] = top
    tr, src = pmc_traffic(dom, workload) if not batch_override else (None, None)
    if tr is not None:
        out["traffic"] = int(tr)
        out["traffic_source"] = "profiles/" + src
    out["t_roof_ms"] = round(t_roof, 4)
    out["step_frac"] = round(t_roof / ms_per_step, 4) if ms_per_step > 0 else None
    # the same sum with EVERY contraction priced at the fp32 matrix-instruction peak (157.3 TFLOP/s): the figure rounds 1-2 reported
    out["t_roof_fp32_mfma_ms"] = round(t_roof_fp32, 4)
    out["step_frac_fp32_mfma"] = round(t_roof_fp32 / ms_per_step, 4) if ms_per_step > 0 else None
    out["step_tflops"] = round(sum(a["flops"] for a in by.values()) / (ms_per_step * 1e-3) / 1e12, 3)
    out["step_hbm_gbs_algorithmic"] = round(sum(a["bytes"] for a in by.values()) / (ms_per_step * 1e-3) / 1e9, 1)
    out["sum_kernel_ms"] = round(total_ms, 4)
    if breakdown:
        for k in sorted(by, key=lambda k: -by[k]["ms"]):
            a = by[k]
            tfk = a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else 0
            gb = a["bytes"] / (a["ms"] * 1e-3) / 1e9 if a["ms"] > 0 else 0
            lb = sum(bound_ms(r["flops"], r["bytes"], r["label"]) for r in rows if r["label"] == k)
            print("%-28s launches %3d  ms %8.3f  %5.1f%%  %7.2f TF/s  %8.1f GB/s  of its roof %5.2f" %
                  (k, a["launches"], a["ms"], 100 * a["ms"] / total_ms, tfk, gb, lb / a["ms"] if a["ms"] > 0 else 0), file=sys.stderr)
        if os.environ.get("RCV_BENCH_ROWS"):
            # per op: measured vs the per-op roofline bound max(FLOP/157.3T, bytes/8T); sorted by the gap
            for r in sorted(rows, key=lambda r: -(r["ms"] - bound_ms(r["flops"], r["bytes"], r["label"]))):
                lb = bound_ms(r["flops"], r["bytes"], r["label"])
                print("%s %-26s %-24s ms %7.4f  bound %7.4f  gap %7.4f  %6.2f TF/s %7.1f GB/s" %
                      ("B" if r["bwd"] else "F", r["label"], r["shape"], r["ms"], lb, r["ms"] - lb, r["flops"] / max(r["ms"], 1e-9) / 1e9,
                       r["bytes"] / max(r["ms"], 1e-9) / 1e6), file=sys.stderr)
            print("sum of per-op bounds %.3f ms, sum of kernel times %.3f ms" % (t_roof, total_ms), file=sys.stderr)
    return out


def run_labelprop(args, dev, rank, world):
    import robocupvision_amd.model as M
    _, B, H, W = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    torch.manual_seed(12345678)
    net = M.LabelProp(5, 32, 0.0)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(dev).eval()
    x = torch.randn(B, 8, H, W, generator=torch.Generator().manual_seed(1 + rank)).to(dev)
    with torch.no_grad():
        for _ in range(args.warmup):
            net(x)
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for k in range(args.steps):
            net(x)
            marks[k + 1].record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    ms = 1e3 * elapsed / args.steps
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    out = {"metric": "inference images/sec, LabelProp 160x120 8-channel frame-pair inputs, B=%d" % B,
           "value": round(B * world * args.steps / elapsed, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms, 4), "ms_per_step_median": round(per_step[len(per_step) // 2], 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": args.workload, "per_gpu_batch": B, "height": H, "width": W,
                      "step": "one eval-mode forward call (validLabelProp.py:132-135), NCHW in -> logits", "latency_ms": round(ms, 4),
                      "parallelism": "replicas%d" % world, "weights": "seeded random init (the shipped .pth does not travel)"}}
    if rank == 0 and not args.no_roofline:
        rows = net.__dict__["_engine"].profile_last(reps=5)
        out["roofline"] = roofline_block(rows, ms, args.workload, args.batch, args.breakdown)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_labelprop(sd, B, H, W)
    return out


# Collective backend of the N > 1 runs: "nccl" (= RCCL over xGMI) is the product path.  RCV_DIST_BACKEND=gloo is a REHEARSAL knob: it lets
# two ranks share the one GPU of a development box (RCCL refuses two ranks on one device), so that the whole N > 1 code path of this
# file and of Trainer(distributed=True) runs before an 8-GPU node is available; its timings mean nothing.
BACKEND = os.environ.get("RCV_DIST_BACKEND", "nccl")


def _barrier(dist, local_rank):
    if BACKEND == "nccl":
        dist.barrier(device_ids=[local_rank])
    else:
        dist.barrier()


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child process, relay rank 0's JSON line and the exit
    code.  RCV_BENCH_LAUNCHER (tests) replaces the `python -m torch.distributed.run` prefix."""
    import shlex
    import subprocess
    have = int(os.environ["RCV_BENCH_ASSUME_DEVICES"]) if os.environ.get("RCV_BENCH_ASSUME_DEVICES") else torch.cuda.device_count()
    if have < n and BACKEND == "nccl":
        print("bench.py: --gpus %d asked for, %d HIP device(s) visible" % (n, have), file=sys.stderr)
        return 2
    launcher = shlex.split(os.environ["RCV_BENCH_LAUNCHER"]) if os.environ.get("RCV_BENCH_LAUNCHER") \
        else [sys.executable, "-m", "torch.distributed.run"]
    cmd = launcher + ["--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                      os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():           # rank 0 prints ONE JSON line; anything else on stdout (launcher chatter) goes to stderr
        try:
            if ln.lstrip().startswith("{") and isinstance(json.loads(ln), dict):
                line = ln
                continue
        except ValueError:
            pass
        print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    elif r.returncode == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 1
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # >= 50 timed steps, >= 10 warm-up; the line carries mean AND median
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="robo_unet_640x480_bs32", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch size")
    ap.add_argument("--dice", action="store_true", help="train with DiceLoss (train.py --useDice) instead of the cross entropy")
    ap.add_argument("--graph", type=int, default=0, help="1: replay the whole training step as ONE captured hipGraph (Trainer.capture); "
                    "0 (default): eager.  Measured on robo_unet_160x120_bs64: 2.41 ms/step replayed vs 2.14 eager -- the step is bound by "
                    "its chain of dependent kernels, not by the host, and the replayed graph schedules its two streams worse")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It has made NO GPU call yet (device_count() does not
        # initialise the device on this stack) and never re-executes itself: the ranks are fresh child processes.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d)" % (args.gpus, world, args.gpus))
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py: no HIP device visible (the hot path has no CPU fallback)")
    if BACKEND == "nccl" and local_rank >= n_dev:
        raise SystemExit("bench.py: rank %d has no GPU (%d visible); RCCL needs one device per rank" % (local_rank, n_dev))
    local_rank %= n_dev          # (only the gloo rehearsal shares a device between ranks)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.workload.startswith("labelprop"):
        # inference shards into independent replicas: no collective on the path, N ranks run N copies (whole-job img/s = sum)
        out = run_labelprop(args, dev, rank, world)
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group(BACKEND)
            tt = torch.tensor([out["ms_per_step"]], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            out["ms_per_step"] = round(float(tt.item()), 4)
            out["value"] = round(out["config"]["per_gpu_batch"] * world / (out["ms_per_step"] * 1e-3), 2)
            _barrier(dist, local_rank)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out))
        return
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
        dist.init_process_group(BACKEND)

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
        _barrier(dist, local_rank)
        _barrier(dist, local_rank)

    use_graph = args.graph == 1
    step = trainer.step
    if use_graph:
        for _ in range(3):
            trainer.step(x, t)                  # plans, optimizer state and the backward schedule exist before the capture
        step = trainer.capture(x, t)            # one hipGraph launch per step from here on (same kernels, same order)
    for _ in range(args.warmup):
        step(x, t)
    if dist is not None:
        _barrier(dist, local_rank)
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # per-step device times (median), no host sync
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        step(x, t)
        marks[k + 1].record()
    if dist is not None:
        _barrier(dist, local_rank)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    metrics = trainer.pop_metrics()
    ms_per_step = 1e3 * elapsed / args.steps

    out = {
        "metric": "training images/sec, ROBO-UNet 640x480 bs=32/GPU" if args.workload == "robo_unet_640x480_bs32"
        else "training images/sec, " + args.workload,
        "value": round(B * world * args.steps / elapsed, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "ms_per_step_median": round(median_ms, 4),
        "ms_per_step_min_max": [round(per_step[0], 4), round(per_step[-1], 4)], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload, "per_gpu_batch": B, "global_batch": B * world, "height": H, "width": W,
                   "step": "fwd+CE/argmax+bwd+L1+Adam (train.py:43-74)", "parallelism": "dp%d" % world,
                   "launch": "one captured hipGraph per step" if use_graph else "eager (one rcv_run per pass)",
                   "loss_after": round(metrics["loss"], 6),
                   # f32 tensors, f32 accumulation everywhere.  The wide (>= 64-channel) stride-1 layers form their fp32 products from
                   # operands split EXACTLY into three bf16 values, six bf16 MFMA products per multiply-add (error against fp64 <= the fp32
                   # MFMA chain's: scripts/micro/split_mfma.hip, tests/test_gpu_kernels.py); RCV_MFMA_FP32=1 runs every contraction on
                   # v_mfma_f32_16x16x4_f32 instead
                   "matrix_arithmetic": "fp32 matrix instructions only (RCV_MFMA_FP32=1)" if os.environ.get("RCV_MFMA_FP32")
                   else "fp32; wide stride-1 layers as exact 3 x bf16 operand splits, 6 bf16 MFMA products per multiply-add, fp32 accumulate"},
    }

    eng = model._get_engine()
    plans = [pl for (shape, training), pl in eng.plans.items() if training and pl.side_decided]
    if plans:      # schedule of the backward list the engine measured and kept on its first pass (engine.SIDE_STREAM_MODE)
        out["config"]["second_stream"] = plans[0].side_mode      # "all": filter gradients + their reductions, "reduce": reductions only, "off"
        if plans[0].side_ms:
            out["config"]["backward_ms_all_reduce_off"] = [round(v, 3) for v in plans[0].side_ms]
    if rank == 0 and not args.no_roofline:
        rows = eng.profile_last(reps=3)
        # the optimizer launch belongs to the step as well (one RCV_OP_ADAM_L1 over the flat buffers: 28 B per parameter)
        n_par = eng.flat.numel
        rows.append({"label": "adam_l1", "kind": 16, "ms": 0.0, "flops": 0.0, "bytes": 28.0 * n_par, "shape": "%d parameters" % n_par, "bwd": True})
        out["roofline"] = roofline_block(rows, ms_per_step, args.workload, args.batch, args.breakdown)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(ctor, H, W, dice=args.dice)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        _barrier(dist, local_rank)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
