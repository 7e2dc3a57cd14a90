"""Trainer vs oracle over consecutive steps: per step, how far apart two parameter tensors are relative to the distance travelled."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_net as T
from oracle import cpu_reference as O
from robocupvision_amd.train import Trainer
ctor = dict(noScale=False, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128)
model = T.build(ctor)
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
st = O.TrainState(model.state_dict(), O.NetConfig(**ctor), ce_weight=T.CE_W, lr=1e-3, decay=1e-6)
st2 = O.TrainState(model.state_dict(), O.NetConfig(**ctor), ce_weight=T.CE_W, lr=1e-3, decay=1e-6)     # second oracle run on perturbed inputs: fp32-noise yardstick
model = model.to(T.DEV)
tr = Trainer(model, class_weights=T.CE_W, lr=1e-3, decay=1e-6)
keys = ["downPart.Level0.layers.Conv0.conv.weight", "downPart.Level2.layers.Conv1.conv.weight", "PB.PB_1.layers.Conv1.conv.weight", "upPart.Up2.conv.weight", "segmenter.layers.Class.weight"]
for it in range(8):
    x, t = O.synthetic_batch(2, 48, 64, seed=100 + it)
    ref = O.train_step(st, x, t)
    O.train_step(st2, x * (1 + 1e-6), t)
    tr.step(x.to(T.DEV), t.to(T.DEV))
    met = tr.pop_metrics()
    sd = model.state_dict()
    out = []
    for k in keys:
        a, b, c, b0 = sd[k].detach().double().cpu(), st.sd[k].detach().double(), st2.sd[k].detach().double(), sd0[k].double()
        trav = float((b - b0).pow(2).mean().sqrt())
        out.append("%.2f/%.2f" % (float((a - b).pow(2).mean().sqrt()) / trav, float((c - b).pow(2).mean().sqrt()) / trav))
    print("step %d loss hip %.5f oracle %.5f | apart/travelled hip / oracle-with-1e-6-perturbed-input: %s" % (it, met["loss"], ref["ce"] + ref["reg"], "  ".join(out)))
