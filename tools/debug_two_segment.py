"""which piece of the hoisted two-segment path disagrees with the default wiring (prints rel-L2 per item)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridgeqa_amd import fusion_ops as ops, med, _ext

dev = torch.device("cuda:0")
cfg = med.BertConfig(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=3,
                     vocab_size=200, max_position_embeddings=64, encoder_width=256)
torch.manual_seed(0)
twin = med.BertModelTwin(config=cfg, add_pooling_layer=False).to(dev).eval()
B, L, P, O = 3, 9, 150, 11
g = torch.Generator().manual_seed(1)
ids = torch.randint(5, 190, (B, L), generator=g).to(dev)
am = torch.ones(B, L, dtype=torch.long, device=dev); am[1, 6:] = 0
img0 = torch.randn(B, P, 256, generator=g).to(dev); obj0 = torch.randn(B, O, 256, generator=g).to(dev)
om = torch.ones(B, O, dtype=torch.long, device=dev); om[0, 7:] = 0
ops.set_compute_dtype(torch.bfloat16)
ops.set_overlap(False)
rel = lambda x, y: ((x.float() - y.float()).norm() / (y.float().norm() + 1e-12)).item()


def run(hoisted, native=True):
    med._TWO_SEGMENT = hoisted
    ops._NATIVE_GEMM[0] = native
    twin.zero_grad()
    img, obj = img0.clone().requires_grad_(True), obj0.clone().requires_grad_(True)
    r = twin(ids, attention_mask=am, encoder_hidden_states=img,
             encoder_attention_mask=torch.ones(B, P, dtype=torch.long, device=dev),
             encoder_hidden_states_twin=obj, encoder_attention_mask_twin=om, return_dict=True, output_attentions="last")
    h2d, h3d = r.last_hidden_state
    (h2d.float().square().sum() + h3d.float().square().sum() + r.cross_attentions[-1][0].float().square().sum()).backward()
    grads = {n: p.grad.float().clone() for n, p in twin.named_parameters() if p.grad is not None}
    return h2d.detach().float(), h3d.detach().float(), img.grad.float(), obj.grad.float(), grads


base = run(False, True)
for name, args in (("hoisted native", (True, True)), ("hoisted torch", (True, False)), ("default torch", (False, False))):
    a = run(*args)
    print(name, "h2d %.4f h3d %.4f dimg %.4f dobj %.4f" % tuple(rel(x, y) for x, y in zip(a[:4], base[:4])),
          "worst param grad", max((rel(a[4][k], base[4][k]), k) for k in base[4] if base[4][k].norm() > 1e-2))
# the dX GEMM of the hoisted projection in isolation
G2 = torch.randn(450, 1024, device=dev).to(torch.bfloat16)
w = (torch.randn(1024, 256, device=dev) * 0.05).to(torch.bfloat16)
print("dx2 vs mm", rel(_ext.gemm_dx(G2, w), torch.mm(G2, w)))
