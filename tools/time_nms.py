"""device parse_predictions at the c3 evaluation shape (B = 16 scenes x 256 proposals x 40000 points)"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bridgeqa_amd.ap_helper import parse_predictions
g = torch.Generator().manual_seed(0)
B, K, N, NS, NC = 16, 256, 40000, 18, 18
hubs = torch.rand(B, 12, 3, generator=g) * torch.tensor([6.0, 6.0, 2.5])
end = {"center": hubs[:, torch.randint(0, 12, (K,), generator=g)] + torch.randn(B, K, 3, generator=g) * 0.25,
       "heading_scores": torch.randn(B, K, 1, generator=g), "heading_residuals": torch.zeros(B, K, 1),
       "size_scores": torch.randn(B, K, NS, generator=g), "size_residuals": torch.randn(B, K, NS, 3, generator=g) * 0.05,
       "sem_cls_scores": torch.randn(B, K, NC, generator=g), "objectness_scores": torch.randn(B, K, 2, generator=g) * 2,
       "point_clouds": torch.cat([torch.rand(B, N, 3, generator=g) * torch.tensor([6.5, 6.5, 3.0]), torch.randn(B, N, 132, generator=g)], -1)}
end = {k: v.cuda() for k, v in end.items()}
cfg = dict(remove_empty_box=True, use_3d_nms=True, nms_iou=0.25, use_old_type_nms=False, cls_nms=True, per_class_proposal=True,
           conf_thresh=0.05, dataset_config=types.SimpleNamespace(num_heading_bin=1, num_class=NC, mean_size_arr=np.random.rand(NS, 3) * 0.8 + 0.4))
parse_predictions(dict(end), cfg); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    out = parse_predictions(dict(end), cfg)
torch.cuda.synchronize()
print("device parse_predictions, B=16 K=256 N=40000: %.1f ms per call (incl. the host-side list building), %d entries in scene 0"
      % ((time.time() - t0) / 5 * 1e3, len(out[0])))
