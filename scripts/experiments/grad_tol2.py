"""Per-channel view of the worst BN-affine gradient of grad_tol.py: HIP vs the oracle in fp32 and fp64 (CPU, this box)."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_net as T
from oracle import cpu_reference as O
ctor = dict(noScale=False, planes=8, depth=4, levels=2, bellySize=5, bellyPlanes=128)
B, H, W = 4, 120, 160
x, t = O.synthetic_batch(B, H, W, seed=int(os.environ.get('TOL_SEED', '1')))
model = T.build(ctor)
sd = model.state_dict()
st32 = O.TrainState({k: v.clone() for k, v in sd.items()}, O.NetConfig(**ctor))
O.train_step(st32, x, t, do_step=False)
st64 = O.TrainState({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, O.NetConfig(**ctor))
st64.ce_weight = st64.ce_weight.double() if st64.ce_weight is not None else None
O.train_step(st64, x.double(), t, do_step=False)
import robocupvision_amd.engine as E
if os.environ.get("TOL_WINO"): E.WINOGRAD = os.environ["TOL_WINO"]
if os.environ.get("TOL_NOFUSE"): E.FUSE_UP_INTO_CLS = False
if os.environ.get("TOL_NOMERGED"): E.MERGED_TCONV_MAX_COUT = 0
print("winograd", E.WINOGRAD, "fuse_up", E.FUSE_UP_INTO_CLS, "merged", E.MERGED_TCONV_MAX_COUT)
res = T.hip_step(model.to(T.DEV), x.to(T.DEV), t.to(T.DEV), do_step=False)
for key in st64.names:
    if not (key.endswith("bn.weight") or key.endswith("bn.bias") or key.endswith("conv.weight")): continue
    g64 = st64.sd[key].grad.double().reshape(-1); g32 = st32.sd[key].grad.double().reshape(-1); gh = res["grads"][key].double().cpu().reshape(-1)
    sc = g64.abs().max()
    print("%-46s max %.2e  err/max: oracle32 %.1e hip %.1e  ratio %5.1f" % (key, sc, float((g32 - g64).abs().max() / sc), float((gh - g64).abs().max() / sc),
          float((gh - g64).abs().max() / ((g32 - g64).abs().max() + 1e-30))))
