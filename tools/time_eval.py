"""Wall time of eval_helper.get_eval at the training shapes (B = 16 samples, K = 256 proposals, 64 GT slots, 8864 answers):
host-format outputs (one device->host copy) and device-resident outputs (none).  python tools/time_eval.py"""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridgeqa_amd.eval_helper import get_eval  # noqa: E402

dev = torch.device("cuda:0")
B, K, K2, NH, NS, NC, A = 16, 256, 64, 1, 18, 18, 8864
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.rand(*s, generator=g).to(dev)
n = lambda *s: torch.randn(*s, generator=g).to(dev)
ri = lambda hi, *s: torch.randint(0, hi, s, generator=g).to(dev)


def make():
    d = dict(objectness_scores=n(B, K, 2), objectness_label=ri(2, B, K), objectness_mask=r(B, K).round(),
             object_assignment=ri(K2, B, K), cluster_ref=n(B, K), cluster_labels=torch.nn.functional.one_hot(ri(K, B), K).float(),
             center=r(B, K, 3) * 4, heading_scores=n(B, K, NH), heading_residuals=n(B, K, NH) * 0.1, size_scores=n(B, K, NS),
             size_residuals=n(B, K, NS, 3) * 0.1, sem_cls_scores=n(B, K, NC), center_label=r(B, K2, 3) * 4,
             heading_class_label=ri(NH, B, K2), heading_residual_label=n(B, K2) * 0.1, size_class_label=ri(NS, B, K2),
             size_residual_label=n(B, K2, 3) * 0.1, sem_cls_label=ri(NC, B, K2),
             ref_box_label=torch.nn.functional.one_hot(ri(K2, B), K2), lang_scores=n(B, NC), object_cat=ri(NC, B),
             answer_scores=n(B, A), answer_scores_scene=n(B, A), answer_scores_2d=n(B, A), answer_scores_2d3d=n(B, A),
             answer_cats=(r(B, A) < 0.001).float())
    return d


cfg = types.SimpleNamespace(num_heading_bin=NH, num_size_cluster=NS, num_class=NC, mean_size_arr=np.random.RandomState(0).rand(NS, 3) + 0.5)
for host in (True, False):
    for _ in range(3):
        get_eval(make(), cfg, use_lang_classifier=True, host_outputs=host)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        d = make()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        get_eval(d, cfg, use_lang_classifier=True, host_outputs=host)
        t1 = time.perf_counter()          # host time until get_eval returns (includes its one copy when host=True)
        torch.cuda.synchronize()
        ts.append((t1 - t0, time.perf_counter() - t0))
    ts = np.array(ts) * 1e3
    print("host_outputs=%s: returns after %.2f ms (median), device done after %.2f ms" % (host, np.median(ts[:, 0]), np.median(ts[:, 1])))
