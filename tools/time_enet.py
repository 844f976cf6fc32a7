"""ENet feature extraction throughput on the device (offline multiview preparation):
python tools/time_enet.py [frames per pass]"""
import sys
import time

import torch

from bridgeqa_amd import enet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
net = enet.feature_extractor(None, device=dev)
frames = torch.randint(0, 256, (n, 240, 320, 3), dtype=torch.uint8, device=dev)
for _ in range(2):
    enet.extract_features(net, frames, batch_size=n)
torch.cuda.synchronize()
t = time.time()
for _ in range(5):
    enet.extract_features(net, frames, batch_size=n)
torch.cuda.synchronize()
dt = (time.time() - t) / 5
print("ENet: %d frames per pass, %.1f ms per pass, %.0f frames/s" % (n, dt * 1e3, n / dt))
