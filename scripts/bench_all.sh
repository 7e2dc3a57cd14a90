#!/bin/bash
# all BASELINE configs on one GPU (GPU box): headline first
mkdir -p gpurun_out
for w in robo_unet_640x480_bs32 robo_unet_160x120_bs64 unet_640x480_bs32 robo_unet_320x240_bs32 robo_unet_v2_640x480_bs32; do
  timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline', {})
print('%-26s %9.1f img/s  %8.3f ms/step  sum_kernel %8.3f ms  step %.1f TF/s  dominant %s %.1f TF/s' % (d['config']['workload'], d['value'], d['ms_per_step'], r.get('sum_kernel_ms', 0), r.get('step_tflops', 0), r.get('kernel'), r.get('achieved', 0)))"
done
python - <<'PY'
import time, torch, sys
sys.path.insert(0, '.')
import robocupvision_amd.model as M
torch.manual_seed(12345678)
net = M.LabelProp(5, 32, 0.0).cuda().eval()
for B in (2, 64):
    x = torch.randn(B, 8, 120, 160, device='cuda')
    with torch.no_grad():
        for _ in range(5): net(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 50
        for _ in range(n): y = net(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print('labelprop_160x120 B=%d: %.3f ms/call  %.1f img/s' % (B, dt * 1e3, B / dt))
# PB_FCN / trainer.py path (SGD), trainer.py's shapes
from robocupvision_amd.optim import SGD
from robocupvision_amd.train import Trainer
for noScale, B, H, W in ((False, 64, 120, 160), (True, 32, 240, 320)):
    torch.manual_seed(12345678)
    net = M.PB_FCN(32, 5, 1, noScale, 0).cuda()
    tr = Trainer(net, class_weights=[1, 6, 1.5, 3, 3], optimizer=SGD(net, lr=1e-1, momentum=0.5, weight_decay=1e-3))
    x = torch.randn(B, 3, H, W, device='cuda'); t = torch.randint(0, 5, (B, H, W), device='cuda')
    for _ in range(5): tr.step(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): tr.step(x, t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print('pb_fcn noScale=%s %dx%dx%d: %.3f ms/step  %.1f img/s' % (noScale, B, H, W, dt * 1e3, B / dt))
PY
