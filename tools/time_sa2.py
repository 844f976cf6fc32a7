import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
torch.manual_seed(0)
def mk(cin, chans):
    layers = []
    for a, b in zip([cin] + chans[:-1], chans):
        layers += [nn.Conv2d(a, b, 1, bias=False), nn.BatchNorm2d(b), nn.ReLU(inplace=True)]
    return nn.Sequential(*layers)
for (cin, chans, M, S) in ((135, [64, 64, 128], 2048, 64), (131, [128, 128, 256], 1024, 32)):
    for dt, cl in ((torch.bfloat16, True), (torch.bfloat16, False), (torch.float16, True)):
        mlp = mk(cin, chans).cuda().train()
        x = torch.randn(16, cin, M, S, device="cuda", dtype=dt)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last); mlp = mlp.to(memory_format=torch.channels_last)
        x.requires_grad_(True)
        def step():
            with torch.autocast("cuda", dtype=dt):
                y = mlp(x)
            y.max(dim=3)[0].float().square().mean().backward()
        for _ in range(2):
            t0 = time.time(); step(); torch.cuda.synchronize(); print("  warm", dt, cl, round(time.time() - t0, 3), flush=True)
        t0 = time.time()
        for _ in range(5): step()
        torch.cuda.synchronize()
        print("cin=%d M=%d S=%d %s channels_last=%s: %.2f ms" % (cin, M, S, dt, cl, (time.time() - t0) / 5 * 1e3), flush=True)
