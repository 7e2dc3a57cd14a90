"""Golden-vector generator.  Runs ONLY in the build container, where /root/reference exists.

It imports the reference's ``model.py`` (pure PyTorch), runs it on seeded inputs
on the CPU with 8 threads and stores inputs / expected outputs as small ``.npz`` / ``.json``
fixtures next to this script.  Nothing from the reference's source text is stored: fixtures
are tensors (inputs, parameters produced by the reference constructors, outputs, gradients).

    python tests/golden/make_golden.py            # regenerate everything

The pattern follows the reference's own known-answer dumper (testDumper.py:21-75) but with
seeded inputs and train-mode / backward coverage.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.environ.get("GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))      # GOLDEN_OUT: regenerate elsewhere (to compare)
THREADS = 8


def sd_hash(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()[:16]


def npy(t):
    return t.detach().cpu().numpy().copy()      # copy: later in-place updates must not leak into a fixture


def block_kat(ref, out, name, mod, x, seed):
    """One reference block: train-mode fwd + bwd, running stats, then eval-mode fwd."""
    g = torch.Generator().manual_seed(seed)
    # non-trivial BN affine so gamma/beta gradients and folding are exercised
    for m in mod.modules():
        if isinstance(m, nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
    for k, v in mod.state_dict().items():
        out["%s/p/%s" % (name, k)] = npy(v)
    x = x.clone().requires_grad_(True)
    mod.train()
    y = mod(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    out[name + "/x"] = npy(x)
    out[name + "/gy"] = npy(gy)
    out[name + "/y_train"] = npy(y)
    out[name + "/gx"] = npy(x.grad)
    for k, p in mod.named_parameters():
        out["%s/g/%s" % (name, k)] = npy(p.grad)
    for k, v in mod.state_dict().items():
        if "running" in k:
            out["%s/after/%s" % (name, k)] = npy(v)
    mod.eval()
    with torch.no_grad():
        out[name + "/y_eval"] = npy(mod(x))


def layer_kats(ref):
    out = {}
    g = torch.Generator().manual_seed(2024)
    torch.manual_seed(7)
    cases = [("conv_3_8_s1", 3, 8, 1, (2, 12, 16)), ("conv_8_16_s2", 8, 16, 2, (2, 12, 16)),
             ("conv_16_16_s1", 16, 16, 1, (2, 12, 16)), ("conv_32_64_s2", 32, 64, 2, (2, 8, 12)),
             ("conv_64_64_s1", 64, 64, 1, (2, 6, 10)), ("conv_128_128_s1", 128, 128, 1, (3, 5, 7)),
             ("conv_8_8_s1_odd", 8, 8, 1, (1, 7, 9))]
    for i, (name, cin, cout, s, (n, h, w)) in enumerate(cases):
        block_kat(ref, out, name, ref.Conv(cin, cout, 3, s), torch.randn(n, cin, h, w, generator=g), 100 + i)
    ups = [("up_16_8", 16, 8, (2, 6, 8)), ("up_64_32", 64, 32, (2, 4, 6)), ("up_128_64", 128, 64, (1, 3, 5))]
    for i, (name, cin, cout, (n, h, w)) in enumerate(ups):
        block_kat(ref, out, name, ref.upSampleTransposeConv(cin, cout), torch.randn(n, cin, h, w, generator=g), 200 + i)
    # dilated conv->BN->ReLU blocks used by LabelProp (ConvPoolSimple, model.py:166-176)
    cps = [("cps_32_64_d2", 32, 64, 1, 2, 2, (2, 9, 11)), ("cps_8_16_s2", 8, 16, 2, 1, 1, (2, 12, 16))]
    for i, (name, cin, cout, s, p, d, (n, h, w)) in enumerate(cps):
        block_kat(ref, out, name, ref.ConvPoolSimple(cin, cout, 3, s, p, d, False),
                  torch.randn(n, cin, h, w, generator=g), 300 + i)
    # max-pool (model.py:92-100)
    x = torch.randn(2, 8, 12, 16, generator=g, requires_grad=True)
    y = ref.Pool(8, 2)(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    out.update({"pool/x": npy(x), "pool/y": npy(y), "pool/gy": npy(gy), "pool/gx": npy(x.grad)})
    # classifier (model.py:403-414)
    cl = ref.UltClassifier(8, 5, False, size=1)
    x = torch.randn(2, 8, 12, 16, generator=g, requires_grad=True)
    y = cl(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    out.update({"cls/x": npy(x), "cls/y": npy(y), "cls/gy": npy(gy), "cls/gx": npy(x.grad)})
    for k, p in cl.named_parameters():
        out["cls/p/" + k] = npy(p)
        out["cls/g/" + k] = npy(p.grad)
    # CrossEntropyLoss2d (model.py:76-82) with the train.py:309 weights, + argmax (train.py:70)
    for name, wts in (("ce_w", [1, 10, 30, 10, 2]), ("ce_now", None)):
        crit = ref.CrossEntropyLoss2d(None if wts is None else torch.tensor(wts, dtype=torch.float32))
        lg = (torch.randn(2, 5, 12, 16, generator=g) * 3).requires_grad_(True)
        t = torch.randint(0, 5, (2, 12, 16), generator=g)
        loss = crit(lg, t)
        loss.backward()
        out.update({name + "/logits": npy(lg), name + "/t": npy(t), name + "/loss": npy(loss),
                    name + "/glogits": npy(lg.grad), name + "/argmax": npy(torch.max(lg, 1)[1])})
        if wts is not None:
            out[name + "/w"] = np.asarray(wts, dtype=np.float32)
    # argmax tie rule (first max wins)
    tie = torch.tensor([1.0, 3.0, 3.0, 2.0]).view(1, 4, 1, 1)
    out["tie/argmax"] = npy(torch.max(tie, 1)[1])
    np.savez_compressed(os.path.join(HERE, "layer_kats.npz"), **out)
    print("layer_kats.npz: %d arrays" % len(out))


def grad_summary(model):
    rows = {}
    for k, p in model.named_parameters():
        gr = p.grad.double()
        rows[k] = [float(gr.sum()), float(gr.abs().sum()), float(gr.norm())]
    return rows


def run_step(ref, ctor_kwargs, B, H, W, store_full, tag, out_npz, meta, dice=False, n_class=5, weights=None):
    """The train.py:43-74 step on the reference, seeds as SURVEY.md 8(c).  n_class / weights: train.py:301,309-313 (the --noBall /
    --noGoal / --noRobot / --noLine flags drop classes and the matching entries of the weight vector)."""
    torch.manual_seed(12345678)
    model = ref.ROBO_UNet(**ctor_kwargs)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, H, W, generator=g)
    t = torch.randint(0, n_class, (B, H, W), generator=g)
    if weights is None:
        weights = [1, 2, 6, 3, 2] if dice else [1, 10, 30, 10, 2]
    if dice:        # train.py:309,315 (--useDice)
        crit = ref.DiceLoss(torch.tensor(weights, dtype=torch.float32))
    else:
        crit = ref.CrossEntropyLoss2d(torch.tensor(weights, dtype=torch.float32))
    decay, lr, transfer = 1e-6, 1e-3, 0
    opt = torch.optim.Adam([
        {'params': model.downPart[0:transfer].parameters(), 'lr': lr * 10},
        {'params': model.downPart[transfer:].parameters()},
        {'params': model.PB.parameters()},
        {'params': model.upPart.parameters()},
        {'params': model.segmenter.parameters()}], lr=lr)
    model.train()
    opt.zero_grad()
    pred = model(x)
    ce = crit(pred, t)
    reg = 0
    for p in model.parameters():
        reg = reg + torch.sum(torch.abs(p))
    reg = decay * reg
    loss = ce + reg
    loss.backward()
    gsum = grad_summary(model)
    gnorm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters())))
    opt.step()
    _, pc = torch.max(pred, 1)
    top2 = torch.topk(pred.detach(), min(2, n_class), dim=1)[0]
    margin = (top2[:, 0] - top2[:, 1]) if n_class > 1 else torch.full_like(top2[:, 0], 1e30)      # (one class: no second logit)
    sd1 = model.state_dict()
    model.eval()
    with torch.no_grad():
        pred_eval = model(x)
    m = {
        "ctor": ctor_kwargs, "dice": bool(dice), "B": B, "H": H, "W": W, "threads": torch.get_num_threads(),
        "torch": torch.__version__, "sd_hash_init": sd_hash(sd0), "sd_hash_after_step": sd_hash(sd1),
        "n_params": int(sum(p.numel() for p in model.parameters())),
        "sum_p": float(sum(v.double().sum() for k, v in sd0.items() if v.dtype.is_floating_point and "running" not in k)),
        "logits_sum": float(pred.double().sum()), "logits_abs_sum": float(pred.double().abs().sum()),
        "ce": float(ce), "reg": float(reg), "loss": float(loss), "grad_norm": gnorm,
        "argmax_hist": [int((pc == c).sum()) for c in range(n_class)], "correct": int((pc == t).sum()),
        "n_margin_lt_1e-4": int((margin < 1e-4).sum()), "n_margin_lt_1e-5": int((margin < 1e-5).sum()),
        "min_margin": float(margin.min()),
        "bn0_running_mean": [float(v) for v in sd1["downPart.Level0.layers.Conv0.bn.running_mean"]],
        "bn0_running_var": [float(v) for v in sd1["downPart.Level0.layers.Conv0.bn.running_var"]],
        "eval_logits_sum": float(pred_eval.double().sum()),
        "eval_argmax_hist": [int((torch.max(pred_eval, 1)[1] == c).sum()) for c in range(n_class)],
        "grad_summary": gsum,
        "param_after_step_sum": {k: float(v.double().sum()) for k, v in sd1.items() if v.dtype.is_floating_point},
    }
    # the same step evaluated by the reference in float64: the exact value the fp32 paths approximate.
    # Several gradients are sums with heavy cancellation (biases ahead of BN, BN affine of wide layers):
    # the fp32 reference itself is off by up to ~2e-3 relative there, so parity tests judge against these.
    torch.manual_seed(12345678)
    model64 = ref.ROBO_UNet(**ctor_kwargs).double()
    if dice:
        crit64 = ref.DiceLoss(torch.tensor(weights, dtype=torch.float64))
    else:
        crit64 = ref.CrossEntropyLoss2d(torch.tensor(weights, dtype=torch.float64))
    model64.train()
    pred64 = model64(x.double())
    ce64 = crit64(pred64, t)
    reg64 = 0
    for p in model64.parameters():
        reg64 = reg64 + torch.sum(torch.abs(p))
    (ce64 + decay * reg64).backward()
    m["fp64"] = {"ce": float(ce64), "logits_sum": float(pred64.sum()), "logits_abs_sum": float(pred64.abs().sum()),
                 "grad_norm": float(torch.sqrt(sum((p.grad ** 2).sum() for p in model64.parameters()))),
                 "grad_summary": grad_summary(model64),
                 "max_logit_err_fp32_ref": float((pred64 - pred.double()).abs().max())}
    if n_class != 5 or "nClass" in ctor_kwargs:
        m["n_class"] = n_class
        m["weights"] = [float(v) for v in weights]
    meta[tag] = m
    out_npz[tag + "/argmax"] = npy(pc).astype(np.uint8)
    out_npz[tag + "/eval_argmax"] = npy(torch.max(pred_eval, 1)[1]).astype(np.uint8)
    # flat indices of near-tie pixels (top-2 logit margin < 1e-4): SURVEY F9 -- the reference's own
    # argmax is not stable there across thread counts, so parity tests report them separately
    out_npz[tag + "/near_tie_idx"] = np.nonzero(npy(margin).reshape(-1) < 1e-4)[0].astype(np.int32)
    if store_full:
        out_npz[tag + "/x"] = npy(x)
        out_npz[tag + "/t"] = npy(t).astype(np.int64)
        out_npz[tag + "/logits"] = npy(pred)
        out_npz[tag + "/eval_logits"] = npy(pred_eval)
        for k, p in model.named_parameters():
            if p.numel() <= 4096:
                out_npz["%s/grad/%s" % (tag, k)] = npy(p.grad)
            else:
                out_npz["%s/grad_head/%s" % (tag, k)] = npy(p.grad.reshape(-1)[:64])
        for k, v in sd1.items():
            if "running" in k:
                out_npz["%s/after/%s" % (tag, k)] = npy(v)
    print(tag, "ce=%.8f reg=%.10f gnorm=%.8f hist=%s correct=%d hash=%s" %
          (m["ce"], m["reg"], gnorm, m["argmax_hist"], m["correct"], m["sd_hash_init"]))


def whole_net(ref, big=True):
    out, meta = {}, {}
    ROBO_S = dict(noScale=False, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128)
    ROBO_L = dict(noScale=True, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128)
    UNET_S = dict(noScale=False, planes=8, depth=4, levels=3, bellySize=0, bellyPlanes=128, pool=True)
    UNET_L = dict(noScale=True, planes=8, depth=4, levels=3, bellySize=0, bellyPlanes=128, pool=True)
    run_step(ref, ROBO_S, 2, 48, 64, True, "robo_s_2x48x64", out, meta)
    run_step(ref, ROBO_L, 1, 48, 64, True, "robo_l_1x48x64", out, meta)
    run_step(ref, UNET_S, 2, 48, 64, True, "unet_s_2x48x64", out, meta)
    run_step(ref, UNET_L, 1, 32, 48, True, "unet_l_1x32x48", out, meta)
    if big:
        run_step(ref, ROBO_S, 4, 120, 160, False, "robo_s_4x120x160", out, meta)
        run_step(ref, ROBO_L, 2, 480, 640, False, "robo_l_2x480x640", out, meta)
        run_step(ref, UNET_L, 2, 480, 640, False, "unet_l_2x480x640", out, meta)
    np.savez_compressed(os.path.join(HERE, "whole_net.npz"), **out)
    with open(os.path.join(HERE, "whole_net.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def dice1(ref):
    """DiceLoss, single-class branch (model.py:25-33): one logit channel, sigmoid, classes (target == 1, target == 0)."""
    out = {}
    for name, B, H, W, wts, scale, seed in (("dice1", 2, 24, 32, [1, 3], 1.5, 11), ("dice1_sharp", 1, 16, 48, [2, 1], 8.0, 12)):
        g = torch.Generator().manual_seed(seed)
        x = (torch.randn(B, 1, H, W, generator=g) * scale).requires_grad_(True)
        t = torch.randint(0, 2, (B, H, W), generator=g)
        w = torch.tensor(wts, dtype=torch.float32)
        loss = ref.DiceLoss(w)(x, t)
        loss.backward()
        x64 = x.detach().double().requires_grad_(True)
        loss64 = ref.DiceLoss(w.double())(x64, t)
        loss64.backward()
        out[name + "/logits"] = npy(x); out[name + "/target"] = npy(t).astype(np.int64); out[name + "/weights"] = npy(w)
        out[name + "/loss"] = npy(loss); out[name + "/dlogits"] = npy(x.grad)
        out[name + "/loss64"] = npy(loss64); out[name + "/dlogits64"] = npy(x64.grad)
        print(name, float(loss), float(loss64))
    np.savez_compressed(os.path.join(HERE, "dice1.npz"), **out)


def dice_v2(ref, big=True):
    """DiceLoss (model.py:5-43) known answers, and whole steps of the v2 net (concat skips, 3x3
    classifier, bellySize 9; train.py:302-307) and of the default net trained with --useDice."""
    out, meta = {}, {}
    for name, C, B, H, W, wts, seed in (("dice5", 5, 2, 24, 32, [1, 2, 6, 3, 2], 3), ("dice3", 3, 1, 16, 48, [1, 6, 3], 4),
                                        ("dice5_sharp", 5, 2, 24, 32, [1, 2, 6, 3, 2], 5)):
        g = torch.Generator().manual_seed(seed)
        x = (torch.randn(B, C, H, W, generator=g) * (8.0 if name.endswith("sharp") else 1.5)).requires_grad_(True)
        t = torch.randint(0, C, (B, H, W), generator=g)
        w = torch.tensor(wts, dtype=torch.float32)
        loss = ref.DiceLoss(w)(x, t)
        loss.backward()
        x64 = x.detach().double().requires_grad_(True)
        loss64 = ref.DiceLoss(w.double())(x64, t)
        loss64.backward()
        out[name + "/logits"] = npy(x); out[name + "/target"] = npy(t).astype(np.int64); out[name + "/weights"] = npy(w)
        out[name + "/loss"] = npy(loss); out[name + "/dlogits"] = npy(x.grad)
        out[name + "/loss64"] = npy(loss64); out[name + "/dlogits64"] = npy(x64.grad)
        print(name, float(loss), float(loss64))
    V2_S = dict(noScale=False, planes=8, depth=4, levels=1, bellySize=9, bellyPlanes=64, v2=True, classSize=3)
    V2_L = dict(noScale=True, planes=8, depth=4, levels=1, bellySize=9, bellyPlanes=64, v2=True, classSize=3)
    ROBO_S = dict(noScale=False, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128)
    run_step(ref, V2_S, 2, 48, 64, True, "v2_s_2x48x64", out, meta)
    run_step(ref, V2_L, 1, 48, 64, True, "v2_l_1x48x64", out, meta)
    run_step(ref, ROBO_S, 2, 48, 64, True, "robo_s_2x48x64_dice", out, meta, dice=True)
    run_step(ref, V2_S, 2, 48, 64, True, "v2_s_2x48x64_dice", out, meta, dice=True)
    if big:
        run_step(ref, V2_L, 2, 480, 640, False, "v2_l_2x480x640", out, meta)
        run_step(ref, ROBO_S, 4, 120, 160, False, "robo_s_4x120x160_dice", out, meta, dice=True)
    np.savez_compressed(os.path.join(HERE, "dice_v2.npz"), **out)
    with open(os.path.join(HERE, "dice_v2.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def pbfcn(ref, big=True):
    """the PB_FCN / trainer.py path -- dilated conv->BN->ReLU encoder blocks in TRAINING mode, ConvPool, a whole
    trainer.py:205-221 step (CrossEntropyLoss2d [1,6,1.5,3,3], SGD lr .1 momentum .5 weight_decay 1e-3), and the flat float64
    parameter dump of paramSave.py."""
    out, meta = {}, {}
    g = torch.Generator().manual_seed(4048)
    torch.manual_seed(9)
    block_kat(ref, out, "convpool_16_32", ref.ConvPool(16, 32), torch.randn(2, 16, 12, 16, generator=g), 400)
    block_kat(ref, out, "convpool_32_64", ref.ConvPool(32, 64), torch.randn(2, 32, 10, 14, generator=g), 401)
    block_kat(ref, out, "cpsT_16_16_d2", ref.ConvPoolSimple(16, 16, 3, 1, 2, 2, False), torch.randn(2, 16, 9, 11, generator=g), 402)
    block_kat(ref, out, "cpsT_64_128_d2", ref.ConvPoolSimple(64, 128, 3, 1, 2, 2, False), torch.randn(2, 64, 6, 10, generator=g), 403)
    block_kat(ref, out, "cpsT_8_16_s2", ref.ConvPoolSimple(8, 16, 3, 2, 1, 1, False), torch.randn(2, 8, 12, 16, generator=g), 404)

    def step(tag, noScale, B, H, W, store_full, v2=False):
        return pbfcn_step(ref, out, meta, tag, noScale, B, H, W, store_full, v2)

    model = step("pbfcn_s_2x48x64", False, 2, 48, 64, True)
    step("pbfcn_l_1x64x96", True, 1, 64, 96, True)
    step("pbfcn2_s_2x48x64", False, 2, 48, 64, True, v2=True)
    if big:
        step("pbfcn_s_4x120x160", False, 4, 120, 160, False)
        step("pbfcn_l_2x240x320", True, 2, 240, 320, False)
    pbfcn_tail(ref, model, out, meta)


def pbfcn_step(ref, out, meta, tag, noScale, B, H, W, store_full, v2=False, num_class=5, weights=(1, 6, 1.5, 3, 3)):
    """trainer.py:205-221, two iterations; num_class / weights as trainer.py:126,133-138 (--noBall ... drop classes)."""
    def grad_summary_some(model):
        return {k: [float(p.grad.double().sum()), float(p.grad.double().abs().sum()), float(p.grad.double().norm())]
                for k, p in model.named_parameters() if p.grad is not None}

    torch.manual_seed(12345678)
    model = ref.PB_FCN_2(False, nClass=num_class) if v2 else ref.PB_FCN(32, num_class, 1, noScale, 0)      # trainer.py:126-129
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    gg = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, H, W, generator=gg)
    t = torch.randint(0, num_class, (B, H, W), generator=gg)
    crit = ref.CrossEntropyLoss2d(torch.tensor(list(weights), dtype=torch.float32))
    opt = torch.optim.SGD([{'params': model.parameters()}], lr=1e-1, momentum=0.5, weight_decay=1e-3)
    model.train()
    losses = []
    for it in range(2):          # two steps: the second one exercises the momentum buffer
        opt.zero_grad()
        pred = model(x)
        loss = crit(pred, t)
        loss.backward()
        if it == 0:
            pred0, gsum = pred.detach().clone(), grad_summary_some(model)
            grads0 = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
            none_grads = [k for k, p in model.named_parameters() if p.grad is None]
            gnorm = float(torch.sqrt(sum((gr.double() ** 2).sum() for gr in grads0.values())))
        opt.step()
        losses.append(float(loss))
        if it == 0:
            sd1 = {k: v.clone() for k, v in model.state_dict().items()}
    sd2 = model.state_dict()
    _, pc = torch.max(pred0, 1)
    top2 = torch.topk(pred0, 2, dim=1)[0]
    margin = top2[:, 0] - top2[:, 1]
    model.eval()
    with torch.no_grad():
        pred_eval = model(x)
    meta[tag] = {
        "noScale": noScale, "v2": v2, "B": B, "H": H, "W": W, "threads": torch.get_num_threads(), "torch": torch.__version__,
        "sd_hash_init": sd_hash(sd0), "sd_hash_after_step": sd_hash(sd1), "sd_hash_after_2_steps": sd_hash(sd2),
        "loss": losses[0], "loss_step2": losses[1], "grad_norm": gnorm, "grad_summary": gsum, "none_grads": none_grads,
        "logits_sum": float(pred0.double().sum()), "logits_abs_sum": float(pred0.double().abs().sum()),
        "argmax_hist": [int((pc == c).sum()) for c in range(num_class)], "correct": int((pc == t).sum()),
        "eval_logits_sum": float(pred_eval.double().sum()), "eval_logits_abs_sum": float(pred_eval.double().abs().sum()),
        "param_after_step_sum": {k: float(v.double().sum()) for k, v in sd1.items() if v.dtype.is_floating_point},
        "param_after_2_steps_sum": {k: float(v.double().sum()) for k, v in sd2.items() if v.dtype.is_floating_point},
    }
    out[tag + "/argmax"] = npy(pc).astype(np.uint8)
    out[tag + "/near_tie_idx"] = np.nonzero(npy(margin).reshape(-1) < 1e-4)[0].astype(np.int32)
    if store_full:
        out[tag + "/x"] = npy(x); out[tag + "/t"] = npy(t).astype(np.int64)
        out[tag + "/logits"] = npy(pred0); out[tag + "/eval_logits"] = npy(pred_eval)
        for k, gr in grads0.items():
            if gr.numel() <= 4096:
                out["%s/grad/%s" % (tag, k)] = npy(gr)
            else:
                out["%s/grad_head/%s" % (tag, k)] = npy(gr.reshape(-1)[:64])
        for k, v in sd1.items():
            if "running" in k:
                out["%s/after/%s" % (tag, k)] = npy(v)
    if num_class != 5:
        meta[tag]["num_class"] = num_class
        meta[tag]["weights"] = [float(v) for v in weights]
    print(tag, "loss=%.8f / %.8f gnorm=%.8f hist=%s" % (losses[0], losses[1], gnorm, meta[tag]["argmax_hist"]))
    return model


def pbfcn_tail(ref, model, out, meta):
    import tempfile
    # paramSave.saveParams on the twice-stepped small model: flat float64 dump in state_dict order
    sys.path.insert(0, REF)
    import paramSave
    with tempfile.TemporaryDirectory() as td:
        paramSave.saveParams(td, model.cpu(), "weights.dat", False)
        flat = np.fromfile(os.path.join(td, "weights.dat"))
        paramSave.saveParams(td, model.cpu(), "weights2.dat", True)
        flat_skip = np.fromfile(os.path.join(td, "weights2.dat"))
    meta["saveParams"] = {"model": "pbfcn_s_2x48x64 after 2 steps", "count": int(flat.size), "sha256": hashlib.sha256(flat.tobytes()).hexdigest(),
                          "count_skip_classifier": int(flat_skip.size), "sha256_skip_classifier": hashlib.sha256(flat_skip.tobytes()).hexdigest(),
                          "head": [float(v) for v in flat[:8]], "tail": [float(v) for v in flat[-8:]]}
    np.savez_compressed(os.path.join(HERE, "pbfcn.npz"), **out)
    with open(os.path.join(HERE, "pbfcn.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def nclass(ref):
    """Class counts other than 5 (train.py:301,312-313 / trainer.py:126,136-137: numClass = 5 - nb - ng - nr - nl and the weight
    vector loses the dropped classes' entries; model.py:462 nClass) and the 16-plane net (a 16-channel classifier input)."""
    out, meta = {}, {}
    ROBO_S = dict(noScale=False, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128)
    ROBO_L = dict(noScale=True, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128)
    W_CE, W_DICE, W_PB = [1, 10, 30, 10, 2], [1, 2, 6, 3, 2], [1, 6, 1.5, 3, 3]
    keep = lambda w, drop: [v for k, v in enumerate(w) if k not in drop]      # class order: background, ball, robot, goal, line
    run_step(ref, dict(ROBO_S, nClass=4), 2, 48, 64, True, "robo_s_c4_2x48x64", out, meta, n_class=4, weights=keep(W_CE, (1,)))        # --noBall
    run_step(ref, dict(ROBO_L, nClass=2), 1, 48, 64, True, "robo_l_c2_1x48x64", out, meta, n_class=2, weights=keep(W_CE, (1, 2, 3)))   # only lines
    run_step(ref, dict(ROBO_S, nClass=3), 2, 48, 64, True, "robo_s_c3_2x48x64_dice", out, meta, dice=True, n_class=3,
             weights=keep(W_DICE, (1, 4)))                                                                                            # --noBall --noLine --useDice
    run_step(ref, dict(ROBO_S, nClass=1), 2, 48, 64, True, "robo_s_c1_2x48x64", out, meta, n_class=1, weights=[1.0])
    run_step(ref, dict(ROBO_S, nClass=8), 2, 48, 64, True, "robo_s_c8_2x48x64", out, meta, n_class=8, weights=[1, 10, 30, 10, 2, 4, 3, 5])
    run_step(ref, dict(ROBO_S, planes=16), 2, 48, 64, True, "robo_p16_2x48x64", out, meta)
    run_step(ref, dict(ROBO_S, planes=16, nClass=7), 1, 48, 64, True, "robo_p16_c7_1x48x64", out, meta, n_class=7, weights=[1, 10, 30, 10, 2, 4, 3])
    run_step(ref, dict(ROBO_S, nClass=4), 4, 120, 160, False, "robo_s_c4_4x120x160", out, meta, n_class=4, weights=keep(W_CE, (1,)))
    pbfcn_step(ref, out, meta, "pbfcn_s_c4_2x48x64", False, 2, 48, 64, True, num_class=4, weights=keep(W_PB, (1,)))
    pbfcn_step(ref, out, meta, "pbfcn_l_c3_1x64x96", True, 1, 64, 96, True, num_class=3, weights=keep(W_PB, (1, 3)))
    pbfcn_step(ref, out, meta, "pbfcn2_s_c4_2x48x64", False, 2, 48, 64, True, v2=True, num_class=4, weights=keep(W_PB, (1,)))
    np.savez_compressed(os.path.join(HERE, "nclass.npz"), **out)
    with open(os.path.join(HERE, "nclass.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def labelprop(ref):
    """LabelProp (config 5).  The class cannot be constructed at HEAD (8 args passed to a
    7-arg ConvPoolSimple.__init__), so the generator wraps the constructor to drop the extra
    (dropout) argument; the arithmetic is untouched.  Weights: seeded random (the shipped .pth does
    not travel); stored because the constructor's RNG order is part of the reference."""
    orig = ref.ConvPoolSimple.__init__
    def patched(self, inplanes, planes, size, stride, padding, dilation, bias, *_ignored):
        orig(self, inplanes, planes, size, stride, padding, dilation, bias)
    ref.ConvPoolSimple.__init__ = patched
    try:
        torch.manual_seed(12345678)
        net = ref.LabelProp(5, 32, 0.0)
    finally:
        ref.ConvPoolSimple.__init__ = orig
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():     # make BN running stats / affine non-trivial, as a trained net has
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    net.eval()
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle.cpu_reference import labelprop_inputs
    y_t = torch.randn(120, 160, generator=g)
    y_n = y_t + 0.1 * torch.randn(120, 160, generator=g)
    lab_t = torch.randint(0, 5, (120, 160), generator=g)
    lab_n = torch.randint(0, 5, (120, 160), generator=g)
    x = labelprop_inputs(y_t, y_n, lab_t, lab_n)
    with torch.no_grad():
        y = net(x.clone())
    out = {"x": npy(x), "logits": npy(y), "argmax": npy(torch.max(y, 1)[1]).astype(np.uint8)}
    for k, v in net.state_dict().items():
        out["p/" + k] = npy(v)
    np.savez_compressed(os.path.join(HERE, "labelprop.npz"), **out)
    print("labelprop logits sum %.6f" % float(y.double().sum()))


def surface(ref):
    """Host-side module surface: get_computations (model.py:513-536), pruneModelNew (model.py:45-57) and
    count_zero_weights (model.py:59-66) of seeded reference models; numbers only."""
    import contextlib, io
    cfgs = {
        "robo_s": dict(noScale=False, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128),
        "robo_l": dict(noScale=True, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128),
        "unet_s": dict(noScale=False, planes=8, depth=4, levels=3, bellySize=0, bellyPlanes=128, pool=True),
        "unet_l": dict(noScale=True, planes=8, depth=4, levels=3, bellySize=0, bellyPlanes=128, pool=True),
        "v2_s": dict(noScale=False, planes=8, depth=4, levels=1, bellySize=9, bellyPlanes=128, v2=True, classSize=3),
    }
    meta = {}
    for tag, ctor in cfgs.items():
        torch.manual_seed(12345678)
        m = ref.ROBO_UNet(**ctor)
        e = {"ctor": ctor, "sd_hash_init": sd_hash(m.state_dict()), "computations": [float(c) for c in m.get_computations()],
             "zero_fraction_init": ref.count_zero_weights(m)}
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            masks = ref.pruneModelNew(m.parameters(), 0.1)
        e["prune_ratio"] = 0.1
        e["mask_counts"] = [int(k.sum()) for k in masks]
        e["mask_shapes"] = [list(k.shape) for k in masks]
        e["zero_fraction_pruned"] = ref.count_zero_weights(m)
        e["computations_pruned"] = [float(c) for c in m.get_computations(True)]
        e["sd_hash_pruned"] = sd_hash(m.state_dict())
        meta[tag] = e
        print("surface %-7s %d layers, %d masks, total %.4g -> pruned %.4g ops" %
              (tag, len(e["computations"]), len(masks), sum(e["computations"]), sum(e["computations_pruned"])))
    with open(os.path.join(HERE, "surface.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    torch.set_num_threads(THREADS)
    sys.path.insert(0, REF)
    import model as ref          # the reference, imported (never copied)
    which = sys.argv[1:] or ["layers", "net", "lp", "dice_v2", "pbfcn", "surface"]
    if "layers" in which:
        layer_kats(ref)
    if "net" in which:
        whole_net(ref)
    if "lp" in which:
        labelprop(ref)
    if "dice_v2" in which:
        dice_v2(ref)
    if "dice1" in which:        # (not in the default list: added in round 2, the other files stay byte-identical)
        dice1(ref)
    if "pbfcn" in which:
        pbfcn(ref)
    if "surface" in which:
        surface(ref)
    if "nclass" in which:       # (round 3, not in the default list: the other files stay byte-identical)
        nclass(ref)
