"""dW = dY^T X for the ViT linears (M = 16 x 1025 tokens): which formulation does hipBLASLt run fastest?"""
import time, torch
dev = torch.device("cuda")
M = 16 * 1025
def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for N, K in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    dY = torch.randn(M, N, device=dev).to(torch.bfloat16); X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    fl = 2.0 * M * N * K
    res = {}
    res["mm f32 (current)"] = timed(lambda: torch.mm(dY.t(), X, out_dtype=torch.float32))
    res["mm bf16"] = timed(lambda: torch.mm(dY.t(), X))
    res["mm^T f32"] = timed(lambda: torch.mm(X.t(), dY, out_dtype=torch.float32))
    res["mm^T bf16"] = timed(lambda: torch.mm(X.t(), dY))
    for S in (4, 16):
        a = dY.view(S, M // S, N).transpose(1, 2); b = X.view(S, M // S, K)
        try:
            res["bmm S=%d f32+sum" % S] = timed(lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0))
        except Exception as e:
            res["bmm S=%d f32+sum" % S] = float("nan")
        res["bmm S=%d bf16+sum" % S] = timed(lambda: torch.bmm(a, b).float().sum(0))
    dYc = dY.t().contiguous()
    res["pre-transposed dY^T (NN) f32"] = timed(lambda: torch.mm(dYc, X, out_dtype=torch.float32))
    print("N=%d K=%d: " % (N, K) + "  ".join("%s %.0fus (%.0f TF)" % (k, v, fl / v / 1e6) for k, v in res.items()), flush=True)
