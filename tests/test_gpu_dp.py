"""GPU, in process: the pieces of the data-parallel / caller-side step around the kernels.

  * the bucketed backward (engine.run_bucketed through the real op lists): gradient ranges handed out while backward is still
    running are final -- bitwise equal to the un-bucketed gradient -- once the filter-gradient stream has been joined;
  * the prune mask of train.py:50-65 (fused Adam launch, and the literal loop on the aliased gradient views with stock Adam)
    against the CPU oracle running those reference lines;
  * optimizer bookkeeping: the 10x learning-rate group (transfer > 0, train.py:357-358), parameters without a gradient stay untouched;
  * the one-forward-in-flight rule raises instead of computing wrong gradients; ignored labels (-100) behave as under NLLLoss."""
import numpy as np
import pytest
import torch

from oracle import cpu_reference as O
import robocupvision_amd.model as M
from robocupvision_amd import _lib as L
from robocupvision_amd.train import Trainer
from test_gpu_blocks import close, _t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CE_W = [1, 10, 30, 10, 2]


def build(ctor=None):
    torch.manual_seed(12345678)
    return M.ROBO_UNet(**(ctor or {}))


@pytest.mark.parametrize("buckets", [1, 3, 6])
def test_bucket_callback_ranges_are_final_when_reported(buckets):
    x, t = O.synthetic_batch(2, 48, 64)
    x, t = x.to(DEV), t.to(DEV)
    model = build().to(DEV).train()
    crit = M.CrossEntropyLoss2d(torch.tensor(CE_W, dtype=torch.float32)).to(DEV)
    eng = model._get_engine()
    # (a callback is in place when the plan is built: data-parallel plans keep the filter-gradient reductions in several launches --
    # a single-GPU plan folds them into ONE at the end of the list and would have a single bucket)
    eng.grad_ready_cb = lambda lo, hi: None
    crit(model(x), t).backward()                       # reference: nothing looks at the ranges
    torch.cuda.synchronize()
    ref = eng.flat.grad.clone()
    for k in range(len(eng.flat.params)):              # poison every parameter's slice (the 16-byte padding between slices stays 0)
        eng.flat.grad_view(k).fill_(float("nan"))
    snaps = []

    def cb(lo, hi):
        cur = torch.cuda.current_stream(torch.device(DEV))
        L.join_side(eng.handle, cur.cuda_stream)       # what GradExchange does on its communication stream
        snaps.append((lo, hi, eng.flat.grad[lo:hi].clone()))

    eng.grad_ready_cb, eng.grad_buckets = cb, buckets
    try:
        model.zero_grad(set_to_none=True)
        crit(model(x), t).backward()
        torch.cuda.synchronize()
    finally:
        eng.grad_ready_cb = None
    assert 1 <= len(snaps) <= buckets and (buckets == 1 or len(snaps) >= 2)
    assert snaps[0][1] == eng.flat.numel and snaps[-1][0] == 0
    assert all(snaps[k][0] == snaps[k + 1][1] for k in range(len(snaps) - 1))      # descending, contiguous
    for lo, hi, g in snaps:
        assert torch.equal(g, ref[lo:hi]), "range [%d,%d) was reported before it was final" % (lo, hi)
    assert torch.equal(eng.flat.grad, ref)


def _pruned_oracle(ctor, x, t, steps, ratio):
    torch.manual_seed(12345678)
    sd = M.ROBO_UNet(**ctor).state_dict()
    st = O.TrainState(sd, O.NetConfig(**ctor))
    indices = O.prune_model_new(st.params(), ratio)          # model.py:45-57 on the oracle's parameters
    out = [O.train_step(st, x, t, indices=indices) for _ in range(steps)]
    return st, indices, out


@pytest.mark.parametrize("fused", [True, False])
def test_prune_mask_step_vs_oracle(fused):
    """train.py:344-347 (pruneModelNew) + train.py:50-65 (no L1 term, gradients of pruned weights zeroed) + Adam, three steps."""
    ctor = dict(noScale=False)
    x, t = O.synthetic_batch(2, 48, 64, seed=5)
    st, ref_idx, ref = _pruned_oracle(ctor, x, t, 3, 0.05)
    model = build(ctor).to(DEV)
    indices = M.pruneModelNew(model.parameters(), 0.05)
    assert len(indices) == len(ref_idx) and all(torch.equal(a.cpu(), b) for a, b in zip(indices, ref_idx))
    assert sum(int(m.sum()) for m in indices) > 1000
    if fused:
        tr = Trainer(model, class_weights=CE_W, lr=1e-3, decay=1e-6, prune_indices=indices)
    else:       # stock torch.optim.Adam over the reference's five groups: Trainer runs the literal train.py:59-65 loop on the views
        from robocupvision_amd.optim import reference_param_groups
        tr = Trainer(model, class_weights=CE_W, optimizer=torch.optim.Adam(reference_param_groups(model, 1e-3), lr=1e-3),
                     prune_indices=indices)
    for k in range(3):
        tr.step(x.to(DEV), t.to(DEV))
        met = tr.pop_metrics()
        assert abs(met["loss"] - ref[k]["loss"]) <= 1e-3 * abs(ref[k]["loss"]), (k, met, ref[k]["loss"])
        assert met["reg"] == 0.0                             # train.py:51: reg stays Tensor([0.0]) in a pruning run
    big = [p for p in model.parameters() if p.dim() > 1]
    for p, m in zip(big, indices):
        assert float(p.detach()[m].abs().max()) == 0.0       # pruned weights stay exactly zero: zero gradient => zero Adam update
    for n, p in model.named_parameters():
        if n.startswith("upPart") and n.endswith("conv.bias"):
            # a bias directly ahead of BatchNorm: its exact gradient is 0 (what the engine returns); the reference holds ~1e-9 of
            # rounding noise there, which Adam -- without the L1 term of a non-pruning run to dominate it -- turns into steps of a
            # fraction of lr.  The value of such a bias never reaches the network output.
            continue
        r = st.sd[n].detach()
        # three Adam steps move every live element by ~3*lr; the sign decisions of tiny gradients may differ between fp32
        # evaluations, so compare the bulk: 99.5 % of the elements within 0.2*lr, none further than 2 steps
        d = (p.detach().cpu() - r).abs()
        assert float(d.max()) <= 6.5e-3, n
        # (per-channel parameters -- BatchNorm affine, biases -- have cancellation-heavy gradients: more of their elements sit near
        # a sign change of the normalised Adam step than of the filters')
        per_channel = p.dim() == 1
        n_far = int((d > 2e-4).sum())
        assert n_far <= max(2, int((0.10 if per_channel else 0.03) * d.numel())), (n, n_far, d.numel())
        assert float(d.median()) <= 1e-4, (n, float(d.median()))


def test_transfer_group_gets_ten_times_the_learning_rate():
    """train.py:357-358: downPart[0:transfer] steps at 10x lr.  First Adam step: |delta p| = lr_group (sign of the gradient)."""
    x, t = O.synthetic_batch(2, 48, 64, seed=2)
    model = build().to(DEV)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    tr = Trainer(model, class_weights=CE_W, lr=1e-3, decay=1e-6, transfer=2)
    tr.step(x.to(DEV), t.to(DEV))
    torch.manual_seed(12345678)
    st = O.TrainState(M.ROBO_UNet().state_dict(), O.NetConfig(), transfer=2)
    O.train_step(st, x, t)
    for k, p in model.named_parameters():
        d = (p.detach() - before[k]).abs()
        fast = k.startswith("downPart.Level0.") or k.startswith("downPart.Level1.")
        lr = 1e-2 if fast else 1e-3
        moved = d[d > 0]
        assert moved.numel() > 0 and abs(float(moved.median()) - lr) <= 0.02 * lr, (k, float(moved.median()), lr)
        # against the oracle's post-step parameters: Adam's first step is lr*sign(g), so an element whose gradient is ~0 may land
        # 2*lr away (opposite signs in two fp32 evaluation orders): a handful per tensor at most, nothing further than that
        r = st.sd[k].detach()
        dd = (p.detach().cpu() - r).abs()
        assert float(dd.max()) <= 2.02 * lr, (k, float(dd.max()))
        assert float((dd <= 0.02 * lr).float().mean()) >= 0.999 or k.endswith("conv.bias") and k.startswith("upPart"), k


def test_parameters_without_gradient_are_not_stepped():
    """PB_FCN_2 (a ROBO_UNet subclass) carries a pooled classification head the segmentation graph never reads: torch.optim.Adam
    skips it (grad None); the fused launch must leave it bit-unchanged as well, L1 decay included."""
    torch.manual_seed(12345678)
    net = M.PB_FCN_2(False).to(DEV)
    head = {k: v.detach().clone() for k, v in net.named_parameters() if k.startswith("classifier.")}
    assert head
    tr = Trainer(net, class_weights=CE_W, lr=1e-2, decay=1e-2)
    x, t = O.synthetic_batch(2, 48, 64, seed=4)
    for _ in range(2):
        tr.step(x.to(DEV), t.to(DEV))
    after = dict(net.named_parameters())
    for k, v in head.items():
        assert torch.equal(after[k].detach(), v), k
    moved = [k for k, p in net.named_parameters() if not k.startswith("classifier.")]
    assert moved


def test_backward_of_a_stale_forward_raises():
    x1, t1 = O.synthetic_batch(1, 16, 24)
    model = build().to(DEV).train()
    crit = M.CrossEntropyLoss2d().to(DEV)
    loss1 = crit(model(x1.to(DEV)), t1.to(DEV))
    with torch.no_grad():
        model(torch.zeros(2, 3, 24, 32, device=DEV))          # e.g. a validation batch in between
    with pytest.raises(L.RcvError, match="stale forward"):
        loss1.backward()
    # the documented order works
    model.zero_grad(set_to_none=True)
    crit(model(x1.to(DEV)), t1.to(DEV)).backward()
    assert all(p.grad is not None for p in model.parameters())


@pytest.mark.parametrize("weighted", [True, False])
def test_ignored_labels_behave_like_nllloss(weighted):
    """Labels outside [0, C) -- NLLLoss's ignore_index -100 -- carry weight 0: they drop out of the loss, of its normaliser and
    of the gradient (the reference's nn.NLLLoss does the same for -100), instead of indexing the class weights out of bounds."""
    g = torch.Generator().manual_seed(9)
    lg = torch.randn(2, 5, 12, 16, generator=g)
    t = torch.randint(0, 5, (2, 12, 16), generator=g)
    t[0, :4] = -100
    w = torch.tensor([1.0, 10, 30, 10, 2]) if weighted else None
    ref_in = lg.clone().requires_grad_(True)
    ref = torch.nn.NLLLoss(w)(torch.nn.functional.log_softmax(ref_in, dim=1), t)
    ref.backward()
    crit = M.CrossEntropyLoss2d(w).to(DEV)
    x = lg.to(DEV).requires_grad_(True)
    loss = crit(x, t.to(DEV))
    loss.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    close(x.grad, ref_in.grad, "dlogits with ignored labels", rtol=1e-4)
    assert float(x.grad[0, :, :4].abs().max()) == 0.0
    assert int(crit.last_stats[2]) == int((torch.max(lg, 1)[1] == t).sum())


def test_captured_step_replays_bit_identically():
    """Trainer.capture: the whole step as ONE hipGraph launch (forward, loss, backward on two streams, Adam + metrics with the step
    number on the device).  Five replayed steps against five eager steps from the same state: bit-identical parameters, BatchNorm
    buffers and metrics."""
    x, t = O.synthetic_batch(4, 48, 64, seed=6)
    x, t = x.to(DEV), t.to(DEV)
    x2, t2 = O.synthetic_batch(4, 48, 64, seed=8)
    x2, t2 = x2.to(DEV), t2.to(DEV)
    out = []
    for graph in (False, True):
        model = build().to(DEV)
        tr = Trainer(model, class_weights=CE_W, lr=1e-3, decay=1e-6)
        for _ in range(3):
            tr.step(x, t)
        tr.optimizer.use_device_step()                 # (both runs read the step number from the device: same arithmetic)
        if graph:
            tr.step(x, t)                              # capture() runs one eager step itself: the eager run takes it here
            step = tr.step
        else:
            step = tr.capture(x, t)
        tr.pop_metrics()
        for k in range(5):
            step(x2 if k % 2 else x, t2 if k % 2 else t)
        torch.cuda.synchronize()
        out.append((tr.pop_metrics(), {k: v.detach().clone() for k, v in model.state_dict().items()}))
    (ma, sa), (mb, sb) = out
    assert ma == mb, (ma, mb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_eval_head_is_cached_and_invalidated():
    """Inference skips the parameter-only head of the forward list (filter repack, BatchNorm constants) while nothing wrote the
    parameters -- and must notice every way they can change: a train-mode forward (running statistics), an optimizer launch of
    this package, load_state_dict / in-place torch ops."""
    x, t = O.synthetic_batch(2, 48, 64, seed=3)
    x, t = x.to(DEV), t.to(DEV)
    model = build().to(DEV)
    tr = Trainer(model, class_weights=CE_W, lr=1e-2, decay=1e-6)

    def eval_logits():
        if model.training:         # (every .eval() call drops the cache on purpose -- see test_data_writes_... below)
            model.eval()
        with torch.no_grad():
            return model(x).clone()

    def fresh_logits():            # the same parameters through a brand-new engine: nothing cached
        torch.manual_seed(0)
        m2 = M.ROBO_UNet().to(DEV)
        m2.load_state_dict(model.state_dict())
        m2.eval()
        with torch.no_grad():
            return m2(x).clone()

    a = eval_logits()
    eng = model._get_engine()
    plan = eng._last[0]
    assert plan.n_head >= 2 and not eng.params_dirty
    assert torch.equal(eval_logits(), a) and torch.equal(a, fresh_logits())
    tr.step(x, t)                                      # parameters and running statistics change through kernels
    b = eval_logits()
    assert not torch.equal(a, b) and torch.equal(b, fresh_logits())
    with torch.no_grad():
        model.segmenter.layers.Class.bias.add_(1.0)    # a torch in-place op (what load_state_dict does): version counter
        model.downPart.Level0.layers.Conv0.bn.running_var.mul_(2.0)
    c = eval_logits()
    assert not torch.equal(b, c) and torch.equal(c, fresh_logits())


def _fresh_eval_logits(model, x):
    torch.manual_seed(0)
    m2 = M.ROBO_UNet().to(DEV)
    m2.load_state_dict(model.state_dict())
    m2.eval()
    with torch.no_grad():
        return m2(x).clone()


def test_eval_after_graph_replay_sees_the_new_weights():
    """A replayed hipGraph writes parameters and BatchNorm buffers through raw pointers: no tensor version moves.  The eval forward
    after it must not reuse the packed filters / BatchNorm constants its cached head derived from the OLD weights
    (capture -> eval -> replays -> eval == a fresh engine on the current state dict)."""
    x, t = O.synthetic_batch(2, 48, 64, seed=3)
    x, t = x.to(DEV), t.to(DEV)
    model = build().to(DEV)
    tr = Trainer(model, class_weights=CE_W, lr=1e-2, decay=1e-6)
    for _ in range(3):
        tr.step(x, t)
    step = tr.capture(x, t)
    step(x, t)
    _, _, _ = tr.evaluate(x, t)
    a = tr.evaluate(x, t)[0].clone()
    assert torch.equal(a, _fresh_eval_logits(model, x))
    model.train()
    for _ in range(4):
        step(x, t)
    b = tr.evaluate(x, t)[0].clone()
    torch.cuda.synchronize()
    assert not torch.equal(a, b)
    assert torch.equal(b, _fresh_eval_logits(model, x))
    # replay WITHOUT an intervening .train()/.eval() call: the flag set by step_fn is what invalidates
    step(x, t)
    model.training = False
    for mod in model.modules():
        mod.training = False
    with torch.no_grad():
        c = model(x).clone()
    assert torch.equal(c, _fresh_eval_logits(model, x)) and not torch.equal(c, b)


def test_data_writes_are_seen_at_the_next_eval_call():
    """``p.data`` edits bump no version counter (the reference's pruneModel does exactly that, model.py:626-640).  The cache is
    dropped at every .eval()/.train() call (the reference's scripts call model.eval() at the top of each validation pass) and by
    invalidate(); without either, a .data edit between two eval forwards is the one documented blind spot."""
    x, _ = O.synthetic_batch(2, 48, 64, seed=3)
    x = x.to(DEV)
    model = build().to(DEV).eval()
    with torch.no_grad():
        a = model(x).clone()
        assert torch.equal(model(x), a)
        w = model.downPart.Level1.layers.Conv0.conv.weight
        v0 = w._version
        w.data.mul_(0.5)                               # pruneModel-style write
        model.segmenter.layers.Class.bias.data[1] = 3.0
        assert w._version == v0                        # ... invisible to the version key
        model.eval()                                   # what valid() does first
        b = model(x).clone()
        assert not torch.equal(a, b) and torch.equal(b, _fresh_eval_logits(model, x))
        w.data.mul_(2.0)
        model.invalidate()                             # explicit form
        c = model(x).clone()
        assert torch.equal(c, _fresh_eval_logits(model, x)) and not torch.equal(c, b)


def test_plan_of_a_live_capture_is_not_evicted():
    """LRU eviction of plans (PLAN_BYTES_BUDGET) skips a plan a captured hipGraph still replays into."""
    x, t = O.synthetic_batch(2, 48, 64, seed=3)
    x, t = x.to(DEV), t.to(DEV)
    model = build().to(DEV)
    tr = Trainer(model, class_weights=CE_W, lr=1e-3, decay=1e-6)
    for _ in range(3):
        tr.step(x, t)
    step = tr.capture(x, t)
    eng = model._get_engine()
    plan = eng._last[0]
    assert plan.pinned == 1
    eng.plan_bytes_budget = 1                          # every other plan is over budget from here on
    for hw in ((32, 48), (40, 56), (56, 64)):
        xx, tt = O.synthetic_batch(2, *hw, seed=5)
        tr.step(xx.to(DEV), tt.to(DEV))
    assert any(pl is plan for pl in eng.plans.values())
    step(x, t)                                         # still replays into live buffers
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in model.parameters())
    del step
    import gc
    gc.collect()
    assert plan.pinned == 0
