"""Which hipBLASLt kernels does torch pick for the three ViT shapes where this repo's GEMM family is behind?  Run under
rocprofv3 --kernel-trace; the kernel names carry the macro-tile (MT), depth and wave layout.  python tools/hipblaslt_names.py"""
import torch
dev = torch.device("cuda:0")
for name, m, n, k in (("fc2", 16400, 768, 3072), ("proj", 16400, 768, 768), ("twinkv", 16720, 1536, 768), ("dx_fc1", 16400, 3072, 768)):
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        y = torch.nn.functional.linear(x, w, b)
    torch.cuda.synchronize()
