#!/usr/bin/env python3
"""Round 3: brute-force check of the planners' tile choices on the op records of a real training step.

Runs one Trainer step of a BASELINE workload, then for every conv / transposed-conv / filter-gradient record of its forward and
backward lists times the library's own tile and a sweep of alternatives forced through the experiment knobs of an EXPERIMENTS build
(RCV_CONV_TILE=tile,R,Wt  RCV_CONVS_TILE=R,Wt,WN  RCV_WGRAD_TILE=R,Wt), each through a FRESH handle (the plan cache is per handle).
   RCV_LIBRARY=$PWD/robocupvision_amd/librcv_X.so python scripts/experiments/sweep_tiles.py [workload]"""
import ctypes as C, os, re, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import robocupvision_amd.model as M
from robocupvision_amd import _lib as L
from robocupvision_amd.train import Trainer

wl = sys.argv[1] if len(sys.argv) > 1 else "robo_unet_640x480_bs32"
ctor, B, H, W = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
torch.manual_seed(12345678)
model = M.ROBO_UNet(**ctor).to(dev)
x, t = bench.synthetic(B, H, W, 1)
x, t = x.to(dev), t.to(dev)
tr = Trainer(model, class_weights=[1, 10, 30, 10, 2])
for _ in range(2):
    tr.step(x, t)
torch.cuda.synchronize()
eng = model._get_engine()
plan = eng._last[0]
lib = L.load()
stream = torch.cuda.current_stream().cuda_stream
KNOBS = ("RCV_CONV_TILE", "RCV_CONVS_TILE", "RCV_WGRAD_TILE")


def fresh_handle():
    h = C.c_void_p()
    L.check(lib.rcv_create(0, C.byref(h)), "rcv_create")
    return h


def time_op(op, env, reps=8):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    h = fresh_handle()
    try:
        o = L.RcvOp.from_buffer_copy(op)
        try:
            nb = L.op_workspace(h, o)
        except L.RcvError:
            return None, None
        part = torch.empty(max(nb // 4, 4), device=dev) if nb else None
        if part is not None:
            o.p[L.RCV_P_PART] = part.data_ptr()
        lst = L.OpList([o])
        try:
            label = lst.labels(h)[0]
            for _ in range(2):
                lst.run(h, stream)
            ms = sorted(lst.run_timed(h, stream)[0] for _ in range(reps))
        except L.RcvError:
            return None, None
        return ms[len(ms) // 2], label
    finally:
        torch.cuda.synchronize()
        lib.rcv_destroy(h)
        for k in KNOBS:
            os.environ.pop(k, None)


TILES = {(2, 5, 4, 1): 0, (2, 5, 2, 2): 1, (2, 5, 1, 4): 2, (1, 5, 1, 4): 3, (1, 5, 4, 1): 4, (1, 5, 2, 2): 5, (1, 5, 1, 2): 6}
PIX = {0: 80, 1: 160, 2: 320, 3: 320, 4: 80, 5: 160, 6: 160}


def widths(TW, lo=4):
    out = []
    for nx in range(1, TW + 1):
        wt = -(-TW // nx)
        if wt < lo:
            break
        if wt not in out:
            out.append(wt)
    return out


total_def = total_best = 0.0
for name, lst in (("F", plan.ce["fwd"]), ("B", plan.ce["bwd"])):
    for k in range(lst.n):
        op = lst.arr[k]
        if op.kind not in (L.OP_CONV, L.OP_TCONV, L.OP_WGRAD):
            continue
        i = op.i
        t0, label = time_op(op, {})
        if t0 is None:
            continue
        TH, TW = (i[L.RCV_I_H], i[L.RCV_I_W]) if op.kind == L.OP_TCONV else (i[L.RCV_I_HO], i[L.RCV_I_WO])
        cands = []
        m = re.match(r"t?conv[ma]?_(?:dma|mfma)<(\d+),(\d+),(\d+),(\d+),", label)
        if op.kind == L.OP_WGRAD and label.startswith("wgrad_mfma"):
            for wt in widths(TW)[:10]:
                wt4 = (wt + 3) // 4 * 4
                for R in sorted({r for r in (1, 2, 3, 4, 5, 6, 8, 10, 11, 12, 15, 16, 19, 20, 24, 30) if r <= TH and r * wt4 <= 1024}):
                    cands.append({"RCV_WGRAD_TILE": "%d,%d" % (R, wt)})
        elif "s_mfma<" in label:
            for WN in (2, 3, 5):
                for wt in widths(TW, 8)[:8]:
                    rmax = 64 * WN // wt
                    for R in sorted({r for r in (rmax, rmax - 1, max(rmax // 2, 1), 1, 2, 3, 4) if 1 <= r <= min(rmax, TH)}):
                        cands.append({"RCV_CONVS_TILE": "%d,%d,%d" % (R, wt, WN)})
        elif m:
            ti = TILES.get(tuple(int(v) for v in m.groups()))
            if ti is not None:
                for wt in widths(TW, 8)[:8]:
                    rmax = PIX[ti] // wt
                    for R in sorted({r for r in (rmax, max(rmax // 2, 1), 1, 2, 3, 4, 5) if 1 <= r <= min(rmax, TH)}):
                        cands.append({"RCV_CONV_TILE": "%d,%d,%d" % (ti, R, wt)})
        best, best_env = t0, {}
        for env in cands:
            tc, lab = time_op(op, env, reps=5)
            if tc is not None and tc < best:
                best, best_env = tc, env
        if best_env:        # confirm with the full repetitions
            tb, _ = time_op(op, best_env)
            t0b, _ = time_op(op, {})
            best, t0 = tb, t0b
        total_def += t0; total_best += min(best, t0)
        print("%s %-28s %dx%dx%d %d->%d s%d  default %.4f ms  best %.4f ms (%+.1f %%) %s  [%d candidates]" %
              (name, label, i[L.RCV_I_N], i[L.RCV_I_H], i[L.RCV_I_W], i[L.RCV_I_CIN], i[L.RCV_I_COUT], i[L.RCV_I_STRIDE], t0, best,
               100 * (best - t0) / t0, best_env, len(cands)), flush=True)
print("sum of the swept ops: default %.3f ms, best tiles %.3f ms" % (total_def, total_best))
