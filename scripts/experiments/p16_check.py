"""Per-parameter gradient error of a ROBO_UNet variant against the CPU oracle (fp32 and fp64) on the box.
usage: python scripts/experiments/p16_check.py [planes] [nClass] [B H W]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import cpu_reference as O
import robocupvision_amd.model as M

planes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ncls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, H, W = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (2, 48, 64)
torch.set_num_threads(8)
ctor = dict(noScale=False, planes=planes, nClass=ncls, depth=4, levels=2, bellySize=5, bellyPlanes=128)
torch.manual_seed(12345678)
model = M.ROBO_UNet(**ctor)
sd = model.state_dict()
x, t = O.synthetic_batch(B, H, W, n_class=ncls)
w = [1, 10, 30, 10, 2, 4, 3, 5][:ncls]
st = O.TrainState(sd, O.NetConfig(**ctor), ce_weight=w)
ref = O.train_step(st, x, t, do_step=False)
sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
st64 = O.TrainState(sd64, O.NetConfig(**ctor), ce_weight=w)
st64.ce_weight = st64.ce_weight.double()
O.train_step(st64, x.double(), t, do_step=False)
model = model.cuda().train()
crit = M.CrossEntropyLoss2d(torch.tensor(w, dtype=torch.float32)).cuda()
pred = model(x.cuda())
loss = crit(pred, t.cuda())
loss.backward()
torch.cuda.synchronize()
print("logits rel err %.3e  loss %.7f vs %.7f" % (float((pred.cpu() - ref["pred"]).abs().max() / ref["pred"].abs().max()), float(loss), ref["ce"]))
eng = model._get_engine()
plan = eng._last[0]
labs = plan.bwd.labels(eng.handle)
for n, p in model.named_parameters():
    g = p.grad.double().cpu()
    r32, r64 = st.sd[n].grad.double(), st64.sd[n].grad
    sc = float(r64.abs().max()) + 1e-30
    e = (g - r64).abs() / sc
    print("%-44s %-18s hip-fp64 %.2e (n>1e-3: %d)  ref32-fp64 %.2e" % (n, tuple(p.shape), float(e.max()), int((e > 1e-3).sum()), float((r32 - r64).abs().max() / sc)))
