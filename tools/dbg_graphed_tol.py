"""How far apart are eager / graphed / wrapped gradients and the 4-step loops now that the detector's data-path gradients are
deterministic?  Prints what tests/test_graphed_gpu.py bounds.  python tools/dbg_graphed_tol.py"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_graphed_gpu as T
from bridgeqa_amd import fusion_ops as ops
dev = torch.device("cuda", 0)
ops.set_compute_dtype(torch.bfloat16)
want, wl = T._grads_once(dev, "eager")
again, al = T._grads_once(dev, "eager")
def err(n, x):
    ref = want[n.replace(".key.bias", ".value.bias")] if n.endswith(".key.bias") else want[n]
    return ((x[n] - want[n]).norm() / (ref.norm() + 1e-12)).item()
for name, got in (("eager2", again), ("graphed", T._grads_once(dev, "graphed")[0]), ("wrapped", T._grads_once(dev, "wrapped")[0])):
    det = sorted(((err(n, got), n) for n in want if not n.startswith("blip_model.")), reverse=True)[:3]
    blip = sorted(((err(n, got), n) for n in want if n.startswith("blip_model.")), reverse=True)[:3]
    print(name, "det", [(round(e, 6), n[-50:]) for e, n in det], "blip", [(round(e, 6), n[-50:]) for e, n in blip], flush=True)
res = {mode: T._run_loop(dev, mode) for mode in ("eager", "eager", "graphed", "phased")}
for mode in ("eager", "graphed", "phased"):
    print(mode, [round(x, 4) for x in res[mode]])
print("eager again", [round(x, 4) for x in T._run_loop(dev, "eager")])
