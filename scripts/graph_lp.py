import sys, time, torch
sys.path.insert(0, '.')
import robocupvision_amd.model as M
torch.manual_seed(12345678)
net = M.LabelProp(5, 32, 0.0).cuda().eval()
for B in (2,):
    x = torch.randn(B, 8, 120, 160, device='cuda')
    with torch.no_grad():
        for _ in range(5): y_ref = net(x).clone()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            net(x)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            y = net(x)
        g.replay(); torch.cuda.synchronize()
        print('graph output equal:', bool(torch.equal(y, y_ref)))
        for name, fn in (('eager', lambda: net(x)), ('graph', g.replay)):
            for _ in range(10): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): fn()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
            print('labelprop B=%d %s: %.1f us/call' % (B, name, dt * 1e6))
