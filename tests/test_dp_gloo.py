"""CPU, world_size 2 over gloo: the host logic of the data-parallel path -- bucket selection in reverse layer
order, the flat-gradient all-reduce with 1/N scaling, and its semantics (mean over ranks of the per-shard
gradients, per-rank BatchNorm statistics: SURVEY.md 8e) checked against the CPU oracle."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cpu_reference as O
    from robocupvision_amd.engine import select_buckets
    import robocupvision_amd.model as M
    torch.manual_seed(12345678)
    sd = M.ROBO_UNet().state_dict()
    st = O.TrainState(sd, O.NetConfig())
    x, t = O.synthetic_batch(2, 16, 24, seed=1 + rank)          # each rank owns its own shard
    O.train_step(st, x, t, do_step=False)
    names = st.names
    sizes = [(st.sd[n].numel() + 3) // 4 * 4 for n in names]
    offs = [sum(sizes[:k]) for k in range(len(names))]
    numel = sum(sizes)
    flat = torch.zeros(numel)
    for n, o in zip(names, offs):
        flat[o:o + st.sd[n].numel()] = st.sd[n].grad.reshape(-1)
    local = flat.clone()
    # marks as the engine produces them: one per parameter group in reverse order (a growing suffix)
    marks = [(k + 1, offs[len(names) - 1 - k]) for k in range(len(names))]
    hi = numel
    seen = []
    for (_end, lo) in select_buckets(marks, numel, 3):
        dist.all_reduce(flat[lo:hi])                  # what Trainer._grad_ready does per bucket
        seen.append((lo, hi))
        hi = lo
    assert hi == 0 and len(seen) <= 3 and seen[0][1] == numel
    flat /= world
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(gathered) / world
    assert torch.allclose(flat, expect, rtol=0, atol=0)
    if rank == 0:
        out["buckets"] = seen
        out["ok"] = True
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29650 + os.getpid() % 200
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out.get("ok")
    b = out["buckets"]
    assert all(b[k][0] == b[k + 1][1] for k in range(len(b) - 1))      # contiguous, reverse layer order


def test_select_buckets_properties():
    from robocupvision_amd.engine import select_buckets
    marks = [(1, 900), (3, 700), (4, 650), (9, 300), (12, 120), (15, 0)]
    for n in (1, 2, 3, 5):
        sel = select_buckets(marks, 1000, n)
        assert 1 <= len(sel) <= max(n, 1) and sel[-1] == (15, 0)
        assert all(sel[k][1] > sel[k + 1][1] for k in range(len(sel) - 1))
    assert select_buckets([], 10, 3) == []
