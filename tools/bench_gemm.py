"""hipBLASLt (through torch) on the GEMM shapes of the c3 step -- the bar a hand-written kernel has to clear."""
import torch
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
M = 16400
for name, (m, n, k) in {"qkv": (M, 2304, 768), "proj": (M, 768, 768), "fc1": (M, 3072, 768), "fc2": (M, 768, 3072),
                        "kv2d": (16720, 1536, 768), "kv3d": (4416, 1536, 768), "text": (320, 768, 768),
                        "lmhead": (160, 30524, 768)}.items():
    x = torch.randn(m, k, device="cuda").bfloat16(); w = torch.randn(n, k, device="cuda").bfloat16(); g = torch.randn(m, n, device="cuda").bfloat16()
    b = torch.randn(n, device="cuda").bfloat16()
    t_f = timeit(lambda: torch.nn.functional.linear(x, w, b))
    t_dx = timeit(lambda: torch.mm(g, w))
    t_dw = timeit(lambda: torch.mm(g.t(), x))
    fl = 2.0 * m * n * k
    print("%-7s M=%5d N=%5d K=%4d  fwd %.3f ms %4.0f TF | dX %.3f ms %4.0f TF | dW %.3f ms %4.0f TF" %
          (name, m, n, k, t_f, fl / t_f / 1e9, t_dx, fl / t_dx / 1e9, t_dw, fl / t_dw / 1e9))
