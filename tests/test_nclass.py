"""Class counts other than 5, and the 16-plane net.

The reference's scripts build their nets with ``numClass = 5 - nb - ng - nr - nl`` (train.py:301, trainer.py:126, tester.py:105: the
--noBall / --noGoal / --noRobot / --noLine flags) and drop the matching entries of the loss weights (train.py:312-313,
trainer.py:136-137); ``ROBO_UNet(nClass=...)`` / ``PB_FCN(planes, num_classes, ...)`` take any count (model.py:462, 271).  Goldens:
``tests/golden/nclass.{npz,json}`` from the imported reference (``make_golden.py nclass``).

  * CPU: the oracle restatement is bit-exact on them (8 threads);
  * CPU: the planner accepts 1..8 classes on 8 or 16 classifier input channels and refuses everything else AT PLAN TIME -- whatever
    ``rcv_op_workspace`` / ``rcv_op_kernel_label`` accept, the launch accepts (no refusal hides behind the query return);
  * GPU: the HIP step against the goldens at the bars of test_gpu_net.py, module-by-module and through the Trainer's fused path."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, sd_hash
from oracle import cpu_reference as O
import robocupvision_amd.model as M
from robocupvision_amd import _lib as L
from robocupvision_amd.engine import Engine

ROBO_TAGS = ["robo_s_c4_2x48x64", "robo_l_c2_1x48x64", "robo_s_c3_2x48x64_dice", "robo_s_c1_2x48x64", "robo_s_c8_2x48x64",
             "robo_p16_2x48x64", "robo_p16_c7_1x48x64"]
PB_TAGS = ["pbfcn_s_c4_2x48x64", "pbfcn_l_c3_1x64x96", "pbfcn2_s_c4_2x48x64"]


@pytest.fixture(scope="module")
def nc_kats():
    return np.load(os.path.join(GOLDEN, "nclass.npz"))


@pytest.fixture(scope="module")
def nc_meta():
    with open(os.path.join(GOLDEN, "nclass.json")) as f:
        return json.load(f)


def _t(a):
    a = np.asarray(a)
    return torch.from_numpy(np.ascontiguousarray(a)).reshape(a.shape)


@pytest.fixture()
def threads8():
    old = torch.get_num_threads()
    torch.set_num_threads(8)
    yield
    torch.set_num_threads(old)


# ------------------------------------------------------------------------------------------ CPU: oracle pin
@pytest.mark.parametrize("tag", ROBO_TAGS + ["robo_s_c4_4x120x160"])
def test_oracle_step_bit_exact(nc_kats, nc_meta, tag, threads8):
    m = nc_meta[tag]
    n_class, weights = m.get("n_class", 5), m.get("weights", [1, 10, 30, 10, 2])
    torch.manual_seed(12345678)
    sd = M.ROBO_UNet(**m["ctor"]).state_dict()
    assert sd_hash(sd) == m["sd_hash_init"]
    st = O.TrainState(sd, O.NetConfig(**m["ctor"]), ce_weight=weights, use_dice=m["dice"])
    x, t = O.synthetic_batch(m["B"], m["H"], m["W"], n_class=n_class)
    if (tag + "/x") in nc_kats.files:
        assert torch.equal(x, _t(nc_kats[tag + "/x"])) and torch.equal(t, _t(nc_kats[tag + "/t"]))
    res = O.train_step(st, x, t)
    assert res["ce"] == m["ce"] and res["reg"] == m["reg"] and res["correct"] == m["correct"]
    assert np.array_equal(res["pred_class"].numpy().astype(np.uint8), nc_kats[tag + "/argmax"])
    if (tag + "/logits") in nc_kats.files:
        assert torch.equal(res["pred"], _t(nc_kats[tag + "/logits"]))
        for k in nc_kats.files:
            if k.startswith(tag + "/grad/"):
                assert torch.equal(st.sd[k[len(tag) + 6:]].grad, _t(nc_kats[k])), k
    assert sd_hash({k: v.detach() for k, v in st.sd.items()}) == m["sd_hash_after_step"]


@pytest.mark.parametrize("tag", PB_TAGS)
def test_oracle_pb_fcn_steps_bit_exact(nc_kats, nc_meta, tag, threads8):
    m = nc_meta[tag]
    torch.manual_seed(12345678)
    model = M.PB_FCN_2(False, nClass=m["num_class"]) if m["v2"] else M.PB_FCN(32, m["num_class"], 1, m["noScale"], 0)
    sd = model.state_dict()
    assert sd_hash(sd) == m["sd_hash_init"]
    st = O.PBTrainState(sd, m["noScale"], ce_weight=m["weights"], v2=m["v2"])
    x, t = O.synthetic_batch(m["B"], m["H"], m["W"], n_class=m["num_class"])
    res = O.pb_train_step(st, x, t)
    assert res["loss"] == m["loss"] and res["correct"] == m["correct"]
    assert torch.equal(res["pred"], _t(nc_kats[tag + "/logits"]))
    assert sd_hash({k: v.detach() for k, v in st.sd.items()}) == m["sd_hash_after_step"]
    assert O.pb_train_step(st, x, t)["loss"] == m["loss_step2"]
    assert sd_hash({k: v.detach() for k, v in st.sd.items()}) == m["sd_hash_after_2_steps"]


# ------------------------------------------------------------------------------------------ CPU: plan-time acceptance / refusal
def _lower(model, shape=(2, 3, 48, 64), training=True):
    eng = Engine(model._graph(), list(model.parameters()), M._bn_modules(model), dry_run=True)
    plan = eng._plan_for([torch.zeros(shape)], training)
    return eng, plan


@pytest.mark.parametrize("n_class", [1, 2, 3, 4, 5, 6, 7, 8])
def test_every_class_count_plans_with_the_fused_loss(n_class):
    for model in (M.ROBO_UNet(nClass=n_class), M.PB_FCN(32, n_class, 1, False, 0), M.PB_FCN(32, n_class, 1, True, 0)):
        eng, plan = _lower(model)
        assert eng._ce_variant(plan), "the classifier + cross-entropy fast path must exist for %d classes" % n_class
        assert "cls_bwd" in plan.bwd.labels(eng.handle) and "cls_fwd" in plan.fwd.labels(eng.handle)
    eng, plan = _lower(M.ROBO_UNet(planes=16, nClass=n_class))          # 16-channel classifier input: separate loss kernels
    assert "cls_bwd" in plan.bwd.labels(eng.handle)


@pytest.mark.parametrize("make,what", [
    (lambda: M.ROBO_UNet(nClass=9), "8 -> 9"),
    (lambda: M.ROBO_UNet(planes=4), "4 -> 5"),
    (lambda: M.ROBO_UNet(planes=32), "32 -> 5"),
    (lambda: M.PB_FCN(64, 5, 1, False, 0), "classifier"),        # outPlanes = 16 with a fused decoder input is fine; must PLAN or refuse, not crash
])
def test_unsupported_classifier_shapes_are_refused_when_the_plan_is_built(make, what):
    model = make()
    try:
        eng, plan = _lower(model)
    except L.RcvError as e:
        assert what in str(e) or "unsupported" in str(e) or "no tile" in str(e), str(e)
        return
    # a plan that was accepted must answer every record's query -- i.e. nothing is left to fail at launch for shape reasons
    assert len(plan.fwd.labels(eng.handle)) == plan.fwd.n and len(plan.bwd.labels(eng.handle)) == plan.bwd.n
    assert "PB_FCN" in type(model).__name__


def test_what_the_query_accepts_the_launch_would_accept():
    """Raw records through the planner handle: the shape refusals of the small ops sit in front of the query return."""
    h = L.planner_handle(256)
    ok = lambda op: L.OpList([op]).labels(h)
    bad = [
        L.make_op(L.OP_CLS_BWD, 0, n=2, h=8, w=8, cin=8, cout=9),
        L.make_op(L.OP_CLS_BWD, 0, n=2, h=8, w=8, cin=12, cout=5),
        L.make_op(L.OP_CLS_BWD, L.F_FUSED_UP, n=2, h=8, w=8, cin=16, cout=5, stats=L.STATS_BWD_DEC),
        L.make_op(L.OP_CLS_BWD, 0, n=2, h=8, w=8, cin=8, cout=5, stats=L.STATS_BWD_ENC),
        L.make_op(L.OP_CLS_FWD, 0, n=2, h=8, w=8, cin=4, cout=5),
        L.make_op(L.OP_CLS_FWD, L.F_FUSED_UP | L.F_FUSED_CE, n=2, h=8, w=8, cin=16, cout=5),
        L.make_op(L.OP_CE_FWD, 0, n=2, h=8, w=8, cout=9),
        L.make_op(L.OP_CE_BWD, 0, n=2, h=8, w=8, cout=0),
        L.make_op(L.OP_DICE_FWD, 0, n=2, h=8, w=8, cout=1),
        L.make_op(L.OP_POOL_FWD, 0, n=2, h=7, w=8, cout=8),
        L.make_op(L.OP_POOL_FWD, 0, n=2, h=8, w=8, cout=6),
        L.make_op(L.OP_POOL_BWD, 0, n=2, h=8, w=8, cout=12),
        L.make_op(L.OP_COMBINE, 0, n=2, h=8, w=8, cout=6),
        L.make_op(L.OP_MATERIALIZE, 0, n=2, h=8, w=8, cout=6),
        L.make_op(L.OP_ADD_SLICE, 0, n=2, h=8, w=8, cin=16, cout=8),
        L.make_op(L.OP_NHWC_TO_NCHW, 0, n=2, h=8, w=8, cin=16, cout=5),
        L.make_op(L.OP_NCHW_TO_NHWC, 0, n=2, h=8, w=8, cin=5, cout=16),
        L.make_op(L.OP_CONFUSION, 0, n=2, h=8, w=8, cout=9),
        L.make_op(L.OP_BWD_STATS, 0, n=2, h=8, w=8, cout=6),
        L.make_op(L.OP_WGRAD, 0, n=2, h=16, w=16, cin=16, ho=16, wo=16, cout=16, stride=1, dil=1, inmode=L.LOAD_GRAD_ENC, inmode2=L.LOAD_GRAD_ENC),
        L.make_op(L.OP_WGRAD, 0, n=2, h=16, w=16, cin=16, ho=16, wo=16, cout=16, stride=1, dil=1, inmode=L.LOAD_AFFINE, inmode2=L.LOAD_NCHW),
        L.make_op(L.OP_CONV, 0, n=2, h=16, w=16, cin=16, cout=14, ho=16, wo=16, stride=1, dil=1, inmode=L.LOAD_AFFINE),
        L.make_op(L.OP_CONV, 0, n=2, h=16, w=16, cin=16, cout=16, ho=16, wo=16, stride=3, dil=1, inmode=L.LOAD_AFFINE),
    ]
    for op in bad:
        with pytest.raises(L.RcvError):
            ok(op)
    for cin in (8, 16):
        for cout in range(1, 9):
            ok(L.make_op(L.OP_CLS_BWD, 0, n=2, h=8, w=8, cin=cin, cout=cout))
            ok(L.make_op(L.OP_CLS_FWD, 0, n=2, h=8, w=8, cin=cin, cout=cout))
    ok(L.make_op(L.OP_CLS_BWD, L.F_FUSED_UP | L.F_FUSED_CE, n=2, h=8, w=8, cin=8, cout=3, stats=L.STATS_BWD_DEC))


# ------------------------------------------------------------------------------------------ GPU: parity
DEV = "cuda:0"


def _check_grads(tag, grads, kats, m, close, l1_of=None, strict_only=None):
    """Element-wise 1e-3 bars against the reference's gradients.  ``l1_of``: the parameters before the step -- the Trainer applies
    the decay*sign(p) term inside the optimizer launch, the reference's gradients carry it.  ``strict_only``: name prefixes judged
    element-wise; every other tensor is judged as a whole (relative L2 5e-3, 2e-2 for per-channel sums), the bar of
    test_ragged_shapes_vs_oracle -- used for the 16-plane fixtures, whose 6x8 bottom planes hold 96 values per channel: one ReLU whose
    pre-activation sits within rounding distance of zero (measured: one pixel of one channel of Up0) moves everything upstream of it
    by ~2e-3 of the tensor scale in ANY fp32 evaluation order."""
    for k, g in grads.items():
        if g is None or (k.startswith("up") and k.endswith("conv.bias")):
            continue
        if l1_of is not None:
            g = g + 1e-6 * torch.sign(l1_of[k])
        key = "%s/grad/%s" % (tag, k)
        if strict_only is not None and not k.startswith(strict_only):
            if key not in kats.files:
                continue
            r = _t(kats[key]).double()
            rel = float((g.double().cpu() - r).norm() / (r.norm() + 1e-30))
            tol = 2e-2 if (k.endswith("conv.bias") or k.endswith("bn.weight") or k.endswith("bn.bias")) else 5e-3
            assert rel <= tol, "%s grad %s: relative L2 error %.3e" % (tag, k, rel)
            continue
        if key in kats.files:
            close(g, _t(kats[key]), "%s grad %s" % (tag, k), rtol=1e-3, floor=1.0)
        else:
            n_ = m["grad_summary"][k][2]
            n64 = m.get("fp64", {}).get("grad_summary", {}).get(k, [0, 0, n_])[2]
            gn = float(g.double().norm())
            assert abs(gn - n_) <= 1e-3 * n_ + 1e-7 or abs(gn - n64) <= 1e-3 * n64 + 1e-7, (k, gn, n_, n64)
            close(g.reshape(-1)[:64], _t(kats["%s/grad_head/%s" % (tag, k)]), "%s grad head %s" % (tag, k), rtol=1e-3, floor=1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ROBO_TAGS)
def test_gpu_step_vs_golden(nc_kats, nc_meta, tag):
    from test_gpu_blocks import close
    from test_gpu_net import check_mask, hip_step
    m = nc_meta[tag]
    torch.manual_seed(12345678)
    model = M.ROBO_UNet(**m["ctor"])
    assert sd_hash(model.state_dict()) == m["sd_hash_init"]
    model = model.to(DEV)
    x, t = _t(nc_kats[tag + "/x"]).to(DEV), _t(nc_kats[tag + "/t"]).to(DEV)
    res = hip_step(model, x, t, dice=m["dice"], weights=m.get("weights"))
    close(res["pred"], _t(nc_kats[tag + "/logits"]), tag + " logits")
    assert abs(res["ce"] - m["ce"]) <= 1e-3 * abs(m["ce"]) + 1e-9, (res["ce"], m["ce"])
    check_mask(res["pc"], nc_kats[tag + "/argmax"], nc_kats[tag + "/near_tie_idx"], tag)
    assert torch.equal(res["crit"].last_argmax.long(), res["pc"])
    p16 = m["ctor"].get("planes", 8) == 16
    _check_grads(tag, res["grads"], nc_kats, m, close, strict_only=("segmenter", "upPart.Up1", "upPart.Up2") if p16 else None)
    sd = model.state_dict()
    for k in nc_kats.files:
        if k.startswith(tag + "/after/"):
            close(sd[k[len(tag) + 7:]], _t(nc_kats[k]), k)
    model.eval()
    with torch.no_grad():
        # (16-plane fixtures: the Adam step moved the few gradient entries the ReLU flip touches by +-lr in the other direction, so
        # the post-step logits are judged at 1e-3 of their scale instead of 1e-3 of each value)
        close(model(x), _t(nc_kats[tag + "/eval_logits"]), tag + " eval logits", rtol=1e-3, floor=1.0 if p16 else 1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["robo_s_c4_2x48x64", "robo_l_c2_1x48x64", "robo_s_c8_2x48x64", "robo_s_c1_2x48x64"])
def test_gpu_trainer_fused_step_vs_golden(nc_kats, nc_meta, tag):
    """Trainer fast path (cross entropy inside the classifier ops, fused Adam + L1): loss, arg-max, parameters after the step."""
    from test_gpu_blocks import close
    from test_gpu_net import check_mask
    from robocupvision_amd.train import Trainer
    m = nc_meta[tag]
    torch.manual_seed(12345678)
    model = M.ROBO_UNet(**m["ctor"]).to(DEV)
    x, t = _t(nc_kats[tag + "/x"]).to(DEV), _t(nc_kats[tag + "/t"]).to(DEV)
    tr = Trainer(model, class_weights=m["weights"], lr=1e-3, decay=1e-6)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    pred = tr.step(x, t).clone()
    assert model._get_engine()._last[0].ce, "the fused-loss lists must be in use"
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    met = tr.pop_metrics()
    close(pred, _t(nc_kats[tag + "/logits"]), tag + " logits")
    assert abs(met["loss"] - m["loss"]) <= 1e-3 * abs(m["loss"]) and abs(met["reg"] - m["reg"]) <= 1e-5 * abs(m["reg"])
    ndiff = check_mask(tr.criterion.last_argmax, nc_kats[tag + "/argmax"], nc_kats[tag + "/near_tie_idx"], tag)
    assert abs(met["correct_pixels"] - m["correct"]) <= ndiff
    _check_grads(tag, grads, nc_kats, m, close, l1_of=before)
    sd = model.state_dict()
    for k, ref_sum in m["param_after_step_sum"].items():
        if "running" in k:
            continue
        got = float(sd[k].double().sum())
        # Adam's first step moves every element by ~lr: sums must agree to a small fraction of numel*lr (test_trainer_fused_step_vs_golden)
        assert abs(got - ref_sum) <= 0.02 * 1e-3 * sd[k].numel() + 1e-6, (k, got, ref_sum)


@pytest.mark.gpu
def test_gpu_four_class_step_at_the_training_shape(nc_kats, nc_meta):
    from test_gpu_net import check_mask
    from robocupvision_amd.train import Trainer
    tag = "robo_s_c4_4x120x160"
    m = nc_meta[tag]
    torch.manual_seed(12345678)
    model = M.ROBO_UNet(**m["ctor"]).to(DEV)
    x, t = O.synthetic_batch(m["B"], m["H"], m["W"], n_class=4)
    tr = Trainer(model, class_weights=m["weights"], lr=1e-3, decay=1e-6)
    pred = tr.step(x.to(DEV), t.to(DEV)).clone()
    met = tr.pop_metrics()
    assert abs(met["loss"] - m["loss"]) <= 1e-3 * abs(m["loss"])
    assert abs(float(pred.double().abs().sum()) - m["logits_abs_sum"]) <= 1e-3 * m["logits_abs_sum"]
    ndiff = check_mask(torch.max(pred, 1)[1], nc_kats[tag + "/argmax"], nc_kats[tag + "/near_tie_idx"], tag)
    assert abs(met["correct_pixels"] - m["correct"]) <= ndiff
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters())))
    assert abs(gn - m["grad_norm"]) <= 1e-3 * m["grad_norm"], (gn, m["grad_norm"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,fused", [("pbfcn_s_c4_2x48x64", False), ("pbfcn_s_c4_2x48x64", True), ("pbfcn_l_c3_1x64x96", True),
                                       ("pbfcn2_s_c4_2x48x64", False)])
def test_gpu_pb_fcn_step_vs_golden(nc_kats, nc_meta, tag, fused):
    from test_gpu_blocks import close
    from test_gpu_net import check_mask
    from test_gpu_pbfcn import pb_step, _check_after
    from robocupvision_amd.optim import SGD
    m = nc_meta[tag]
    torch.manual_seed(12345678)
    model = M.PB_FCN_2(False, nClass=m["num_class"]) if m["v2"] else M.PB_FCN(32, m["num_class"], 1, m["noScale"], 0)
    assert sd_hash(model.state_dict()) == m["sd_hash_init"]
    model = model.to(DEV)
    x, t = _t(nc_kats[tag + "/x"]).to(DEV), _t(nc_kats[tag + "/t"]).to(DEV)
    opt = SGD(model, lr=1e-1, momentum=0.5, weight_decay=1e-3) if fused else None
    res = pb_step(model, x, t, opt, weights=m["weights"])
    close(res["pred"], _t(nc_kats[tag + "/logits"]), tag + " logits")
    assert abs(res["loss"] - m["loss"]) <= 1e-3 * abs(m["loss"])
    check_mask(res["pc"], nc_kats[tag + "/argmax"], nc_kats[tag + "/near_tie_idx"], tag)
    assert sorted(k for k, g in res["grads"].items() if g is None) == sorted(m["none_grads"])
    _check_grads(tag, res["grads"], nc_kats, m, close)
    _check_after(model.state_dict(), m["param_after_step_sum"], 0.1)
    res2 = pb_step(model, x, t, res["opt"], weights=m["weights"])
    assert abs(res2["loss"] - m["loss_step2"]) <= 2e-3 * abs(m["loss_step2"]), (res2["loss"], m["loss_step2"])


def test_inference_lowering_folds_batchnorm_and_skip_adds():
    """engine.EVAL_FOLD_BN: the inference plans of the relu(bn(conv)) graphs carry no RCV_OP_COMBINE launch (the skip add rides in the
    transposed conv's epilogue); ROBO_UNet (bn(relu(conv)), nothing to fold) keeps its load-transform form; training plans are untouched."""
    for model, shape in ((M.LabelProp(5, 32, 0.0), (2, 120, 160, 8)), (M.PB_FCN(32, 5, 1, False, 0), (2, 3, 128, 160))):
        eng, plan = _lower(model, shape, training=False)
        labels = plan.fwd.labels(eng.handle)
        assert not any("combine" in l for l in labels), labels
    eng, plan = _lower(M.ROBO_UNet(), (2, 3, 48, 64), training=False)
    assert any("combine" in l for l in plan.fwd.labels(eng.handle))
    eng, plan = _lower(M.PB_FCN(32, 5, 1, False, 0), (2, 3, 128, 160), training=True)
    assert any("combine" in l for l in plan.fwd.labels(eng.handle))
