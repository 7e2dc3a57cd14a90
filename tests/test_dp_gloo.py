"""CPU, world_size 2 over gloo: the data-parallel schedule, driven through the PRODUCT code.

The engine lowers the real ROBO-UNet graph to its real backward op list and gradient-ready marks (Engine(dry_run=True): a
planning-only library handle, no GPU); ``Engine._run_backward`` -> ``engine.run_bucketed`` slices that list and calls
``train.GradExchange.grad_ready`` exactly as on the GPU.  Only two things are stand-ins: the executor of a slice (instead of
launching kernels it writes, into the poisoned flat gradient buffer, the CPU oracle's gradient of every parameter whose gradient
one of the slice's ops produces) and the stream operations (recorded).  The all-reduce is a real gloo all-reduce.  A mark that
declares a range final before its producing op ran, a missing range, or a wrong order leaves NaNs / wrong sums behind.

Semantics checked: mean over ranks of the per-shard gradients, per-rank BatchNorm statistics."""
import contextlib
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class SimulatedOps:
    """Stands in for _lib.OpList: 'running' ops [a, b) makes the gradients those ops produce appear in flat.grad."""

    def __init__(self, oplist, flat, grads, L, batch_outputs=None):
        self.arr, self.n, self.flat, self.grads, self.L = oplist.arr, oplist.n, flat, grads, L
        self.batch_outputs = batch_outputs or {}        # Plan.reduce_outputs: index of a batched reduction -> [(gradient pointer, zero-fill?)]
        base = flat.grad.data_ptr()
        self.addr2k = {base + 4 * off: k for k, off in enumerate(flat.offsets)}
        self.produced = set()
        self.slices = []

    def run_slice(self, h, stream, a, b, join=True):
        L = self.L
        slots = {L.OP_WGRAD_REDUCE: (L.RCV_P_OUT, L.RCV_P_BIAS), L.OP_BN_BWD: (L.RCV_P_X1, L.RCV_P_X2),
                 L.OP_CLS_BWD: (L.RCV_P_X1, L.RCV_P_X2), L.OP_MEMSET: (L.RCV_P_OUT,)}
        self.slices.append((a, b, join))
        for i in range(a, b):
            op = self.arr[i]
            outs = [(op.p[slot] or 0, op.kind == L.OP_MEMSET) for slot in slots.get(op.kind, ())]
            if op.kind == L.OP_WGRAD_REDUCE_BATCH:
                outs = list(self.batch_outputs[i])
            for ptr, zero in outs:
                k = self.addr2k.get(ptr)
                if k is None:
                    continue
                view = self.flat.grad_view(k)
                if zero:
                    view.zero_()                # bias ahead of a BatchNorm: exactly zero in the engine
                else:
                    view.copy_(self.grads[k])
                self.produced.add(k)

    def run(self, h, stream):
        self.run_slice(h, stream, 0, self.n)


class RecordingStreams:
    """Stand-in for train._HipStreams: records the stream operations instead of performing them."""

    def __init__(self, log, overlap=True):
        self.log = log
        self.comm = "comm" if overlap else None

    def current(self):
        return "compute"

    def wait(self, waiter, waited):
        self.log.append(("wait", waiter, waited))

    def ptr(self, stream):
        return stream

    def on(self, stream):
        return contextlib.nullcontext()


def _lowered(seed_shift, buckets=3):
    """Model + dry-run engine + real backward plan + oracle gradients of this rank's shard."""
    from oracle import cpu_reference as O
    from robocupvision_amd import _lib as L
    from robocupvision_amd.engine import Engine
    import robocupvision_amd.model as M
    torch.manual_seed(12345678)
    model = M.ROBO_UNet()
    st = O.TrainState(model.state_dict(), O.NetConfig())
    x, t = O.synthetic_batch(2, 16, 24, seed=1 + seed_shift)          # each rank owns its own shard
    O.train_step(st, x, t, do_step=False)
    eng = Engine(model._graph(), list(model.parameters()), [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)], dry_run=True)
    eng.grad_buckets = buckets
    plan = eng._plan_for([x], True)
    ce = eng._ce_variant(plan)                     # the Trainer's fast path runs these lists
    assert ce is not None and plan.bwd_marks
    names = [n for n, _ in model.named_parameters()]
    grads = [st.sd[n].grad.detach().clone() for n in names]
    eng._last = (plan, [x])
    return L, eng, plan, ce, names, grads


def _exchange_pass(L, eng, plan, ops_list, grads, all_reduce, overlap=True):
    from robocupvision_amd.train import GradExchange
    fl = eng.flat
    fl.grad.zero_()
    for k in range(len(fl.params)):
        fl.grad_view(k).fill_(float("nan"))        # poison: a range exchanged before its producer ran stays NaN
    sim = SimulatedOps(ops_list, fl, grads, L, plan.reduce_outputs)
    log = []
    exch = GradExchange(eng, RecordingStreams(log, overlap), lambda t: (log.append(("all_reduce", t.numel())), all_reduce(t))[1],
                        join_side=lambda h, s: log.append(("join_side", s)))
    eng.grad_ready_cb = exch.grad_ready
    exch.begin()
    eng._run_backward(plan, sim)
    exch.finish()
    return sim, exch, log


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L, eng, plan, ce, names, grads = _lowered(rank)
    fl = eng.flat
    sim, exch, log = _exchange_pass(L, eng, plan, ce["bwd"], grads, dist.all_reduce)

    # every parameter the graph uses was produced, every range exchanged exactly once, descending and contiguous
    assert all(k in sim.produced for k in range(len(fl.params)) if eng.param_used[k])
    r = exch.ranges
    assert r[0][1] == fl.numel and r[-1][0] == 0 and all(r[k][0] == r[k + 1][1] for k in range(len(r) - 1)) and 1 <= len(r) <= 3
    assert all(not j for (_a, _b, j) in sim.slices)          # slices never join the filter-gradient stream themselves
    # per bucket: the COMMUNICATION stream waits for the compute stream, then for the side stream, then reduces
    per = [log[i:i + 3] for i in range(0, 3 * len(r), 3)]
    for (lo, hi), ev in zip(r, per):
        assert ev == [("wait", "comm", "compute"), ("join_side", "comm"), ("all_reduce", hi - lo)], ev
    assert log[3 * len(r):] == [("wait", "compute", "comm")]  # the optimizer's stream waits for the last all-reduce

    # semantics: after the 1/world scaling (folded into the optimizer launch) = mean over ranks of the per-shard oracle gradients
    local = torch.zeros(fl.numel)
    for k, g in enumerate(grads):
        if not (names[k].startswith("upPart") and names[k].endswith("conv.bias")):      # exactly zero here (MEMSET), ~1e-9 noise in the oracle
            local[fl.offsets[k]:fl.offsets[k] + g.numel()] = g.reshape(-1)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(gathered) / world
    got = fl.grad / world
    assert not bool(torch.isnan(got).any()), "a range was exchanged before the op that produces it had run"
    assert torch.equal(got, expect)
    if rank == 0:
        out["ranges"] = list(r)
        out["ok"] = True
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_exchange_world2_through_engine_marks():
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29650 + os.getpid() % 200
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out.get("ok")
    assert len(out["ranges"]) == 3


def test_marks_are_tight_and_a_wrong_mark_is_caught():
    """Single process: (1) for every bucket count the real marks give a poison-free exchange; (2) the same machinery fed a mark that
    declares a range final one op too early leaves NaNs behind -- i.e. this test file can turn red."""
    for buckets in (1, 2, 3, 5):
        L, eng, plan, ce, names, grads = _lowered(0, buckets)
        sim, exch, log = _exchange_pass(L, eng, plan, plan.bwd, grads, lambda t: None, overlap=False)
        assert not bool(torch.isnan(eng.flat.grad).any())
        assert len(exch.ranges) <= buckets and exch.ranges[-1][0] == 0
        assert all(e[0] != "wait" for e in log)            # no communication stream: join on the compute stream, then reduce
    L, eng, plan, ce, names, grads = _lowered(0, 3)
    good = list(plan.bwd_marks)
    seen_nan = []
    snapshot = []

    def check(t):
        snapshot.append(bool(torch.isnan(t).any()))
    plan.bwd_marks = [(max(end - 1, 1), lo) for (end, lo) in good]      # every range declared final one op too early
    sim, exch, log = _exchange_pass(L, eng, plan, plan.bwd, grads, check, overlap=False)
    seen_nan.append(any(snapshot))
    assert seen_nan[0], "a premature mark went unnoticed"


def test_select_buckets_properties():
    from robocupvision_amd.engine import select_buckets
    marks = [(1, 900), (3, 700), (4, 650), (9, 300), (12, 120), (15, 0)]
    for n in (1, 2, 3, 5):
        sel = select_buckets(marks, 1000, n)
        assert 1 <= len(sel) <= max(n, 1) and sel[-1] == (15, 0)
        assert all(sel[k][1] > sel[k + 1][1] for k in range(len(sel) - 1))
    assert select_buckets([], 10, 3) == []
