"""GPU: the data-parallel Trainer over the real RCCL path, each run in a FRESH child process (this file sorts first so that the
children are started before the test process itself has touched the GPU).

  * world size 1 with the collectives forced (RCV_FORCE_COLLECTIVES=1): the bucketed backward, the communication stream, the
    side-stream joins and three real ncclAllReduce calls per step -- must be BIT-IDENTICAL to the plain Trainer, and inside the
    golden bars of the reference step;
  * world size 2 (skipped on a one-GPU box): rank r trains on sample r; the exchanged gradient must equal the mean over ranks of the
    CPU oracle's per-shard gradients (per-rank BatchNorm statistics, SURVEY.md 8e i-iv)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "dp_child.py")
pytestmark = pytest.mark.gpu
TAG = "robo_s_2x48x64"


def _run_child(tmp_path, name, mode, nproc=1, extra=(), env_extra=None):
    out = str(tmp_path / (name + ".pt"))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    if mode == "single":
        cmd = [sys.executable, CHILD, "--mode", "single", "--out", out, "--tag", TAG, *extra]
    else:
        port = 29720 + (os.getpid() + len(name)) % 200
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), CHILD, "--mode", "dist", "--out", out, "--tag", TAG, *extra]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "child %s failed (%d):\n%s\n%s" % (name, r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return torch.load(out, weights_only=True)


def _hash(sd):
    import hashlib
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.contiguous().numpy().tobytes())
    return h.hexdigest()[:16]


def test_world1_forced_collectives_bit_identical_to_plain_trainer(tmp_path):
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "whole_net.json")))[TAG]
    plain = _run_child(tmp_path, "plain", "single")
    for name, extra in (("dist1", ()), ("dist1_no_overlap", ("--no-overlap",))):
        d = _run_child(tmp_path, name, "dist", 1, extra, {"RCV_FORCE_COLLECTIVES": "1"})
        assert d["world"] == 1
        assert len(d["ranges"]) == 2 and all(len(r) == 3 for r in d["ranges"]), d["ranges"]     # three buckets per step really went out
        for r in d["ranges"]:
            assert r[0][1] == d["grad_last"].numel() and r[-1][0] == 0 and all(r[k][0] == r[k + 1][1] for k in range(len(r) - 1))
        assert torch.equal(d["losses"], plain["losses"])
        assert torch.equal(d["grad_last"], plain["grad_last"])
        assert _hash(d["sd"]) == _hash(plain["sd"])
    # and the step itself is the reference's: first-step loss (CE + decay*L1) inside the 1e-3 bar of the golden
    assert abs(float(plain["losses"][0]) - meta["loss"]) <= 1e-3 * abs(meta["loss"])


def _check_world2(d):
    from oracle import cpu_reference as O
    import robocupvision_amd.model as M
    assert d["world"] == 2
    kats = np.load(os.path.join(ROOT, "tests", "golden", "whole_net.npz"))
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "whole_net.json")))[TAG]
    x, t = torch.from_numpy(kats[TAG + "/x"]), torch.from_numpy(kats[TAG + "/t"])
    torch.manual_seed(12345678)
    sd = M.ROBO_UNet(**meta["ctor"]).state_dict()
    per_rank = []
    for r in range(2):
        st = O.TrainState(sd, O.NetConfig(**meta["ctor"]))
        O.train_step(st, x[r:r + 1], t[r:r + 1], do_step=False)
        per_rank.append({n: st.sd[n].grad.detach().clone() for n in st.names})
    offs = d["offsets"].tolist()
    for k, n in enumerate(d["names"]):
        if n.startswith("upPart") and n.endswith("conv.bias"):
            continue
        ref = (per_rank[0][n] + per_rank[1][n]) / 2
        got = d["grad_step0"][offs[k]:offs[k] + ref.numel()].view(ref.shape)
        scale = float(ref.abs().max()) + 1e-12
        err = float((got - ref).abs().max()) / scale
        assert err <= 1e-2, "%s: exchanged gradient off by %.3e of its scale" % (n, err)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU node)")
def test_world2_gradient_is_mean_of_per_shard_oracle_gradients(tmp_path):
    _check_world2(_run_child(tmp_path, "dist2", "dist", 2, ("--shard", "1", "--steps", "1")))


@pytest.mark.parametrize("overlap", [True, False])
def test_world2_rehearsal_two_ranks_share_the_gpu_over_gloo(tmp_path, overlap):
    """The same world-size-2 check on a ONE-GPU box: both ranks use device 0 and exchange through gloo (RCCL refuses two ranks on one
    device).  Everything but the collective's transport is the product path: Trainer(distributed=True), the bucketed backward, the
    communication stream and its joins with the library's filter-gradient stream, grad_scale = 1/2 inside the optimizer launch."""
    extra = ("--shard", "1", "--steps", "1") + (() if overlap else ("--no-overlap",))
    _check_world2(_run_child(tmp_path, "gloo2_%d" % overlap, "dist", 2, extra, {"RCV_DIST_BACKEND": "gloo"}))


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` (the driver's command shape, no launcher around it) starts its two ranks itself and prints ONE JSON
    line for the whole job.  Rehearsed on this box's single GPU over gloo; on a multi-GPU node the same command runs over RCCL."""
    env = dict(os.environ)
    env.update({"RCV_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--workload", "robo_unet_160x120_bs64", "--batch", "4"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 8
    assert out["scaling"] == "weak" and out["steps"] == 3 and out["value"] > 0
    assert abs(out["value"] - 8 / (out["ms_per_step"] * 1e-3)) <= 1e-2 * out["value"]
    assert "cpu_baseline" not in out and "roofline" in out
