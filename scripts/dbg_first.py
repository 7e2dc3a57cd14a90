import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import cpu_reference as O
import robocupvision_amd.model as M
from test_gpu_net import hip_step
B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x, t = O.synthetic_batch(B, H, W, seed=11)
outs = []
for flag in ("", "1"):
    if flag: os.environ["RCV_NO_CONV_FIRST"] = "1"
    else: os.environ.pop("RCV_NO_CONV_FIRST", None)
    torch.manual_seed(12345678)
    m = M.ROBO_UNet(noScale=True).to("cuda:0")
    res = hip_step(m, x.to("cuda:0"), t.to("cuda:0"), do_step=False)
    eng = m._get_engine()
    plan = eng._last[0]
    bn0 = m.downPart.Level0.layers.Conv0.bn
    outs.append((res, bn0.running_mean.clone(), bn0.running_var.clone()))
a, b = outs
print("logits max diff", float((a[0]["pred"] - b[0]["pred"]).abs().max()), "ce", a[0]["ce"], b[0]["ce"])
print("running mean diff", float((a[1] - b[1]).abs().max()), "var diff", float((a[2] - b[2]).abs().max()))
for k in [k for k in a[0]["grads"] if k.endswith('conv.weight') or k.endswith('Class.weight')]:
    ga, gb = a[0]["grads"][k].double(), b[0]["grads"][k].double()
    print("%-46s rel %.3e ratio %.4f" % (k, float((ga - gb).norm() / gb.norm()), float((ga * gb).sum() / (gb * gb).sum())))
