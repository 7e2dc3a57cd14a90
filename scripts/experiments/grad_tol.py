"""How close are the per-parameter gradient norms to the reference's fp64 evaluation?  (the bars of tests/test_gpu_net.py)"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_net as T
from oracle import cpu_reference as O
G = os.path.join(os.path.dirname(T.__file__), "golden")
metas = {}
for f in ("whole_net.json", "dice_v2.json"):
    metas.update(json.load(open(os.path.join(G, f))))
for tag in ["robo_s_4x120x160", "robo_l_2x480x640", "unet_l_2x480x640", "v2_l_2x480x640", "robo_s_2x48x64", "unet_s_2x48x64"]:
    m = metas[tag]
    model = T.build(m["ctor"]).to(T.DEV)
    x, t = O.synthetic_batch(m["B"], m["H"], m["W"])
    res = T.hip_step(model, x.to(T.DEV), t.to(T.DEV), dice=m.get("dice", False))
    worst = {"chan": (0, None), "filt": (0, None)}
    for k, g in res["grads"].items():
        if k.startswith("upPart") and k.endswith("conv.bias"):
            continue
        n32, n64 = m["grad_summary"][k][2], m["fp64"]["grad_summary"][k][2]
        got = float(g.double().norm())
        e = min(abs(got - n32) / (n32 + 1e-30), abs(got - n64) / (n64 + 1e-30))
        e64 = abs(got - n64) / (n64 + 1e-30)
        r32 = abs(n32 - n64) / (n64 + 1e-30)
        cls = "chan" if (k.endswith("bn.weight") or k.endswith("bn.bias") or k.endswith("conv.bias")) else "filt"
        if e > worst[cls][0]:
            worst[cls] = (e, (k, e64, r32))
    print(tag, {c: (round(v[0], 6), v[1] and (v[1][0], round(v[1][1], 6), round(v[1][2], 6))) for c, v in worst.items()})
