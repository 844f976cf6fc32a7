"""Epilogue cost of the 256 x 128 kernel at the fc1 / fc2 shapes of config c3: the same GEMM with and without the GELU / GELU-derivative / add epilogues (HIP events, median of 20)."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from bridgeqa_amd import _ext
dev = torch.device('cuda:0')
M = 16400
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
dy, w, pre = rnd(M, 768), rnd(768, 3072, sc=0.05), rnd(M, 3072)
x, w1, b1 = rnd(M, 768), rnd(3072, 768, sc=0.05), torch.randn(3072, device=dev)
def t(f, n=20):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
print("dx_fc2_dgelu %.1f us" % t(lambda: _ext.gemm_dx(dy, w, pre_act=pre)))
print("dx_fc2 plain %.1f us" % t(lambda: _ext.gemm_dx(dy, w)))
print("dx_fc2 add   %.1f us" % t(lambda: _ext.gemm_dx(dy, w, add=pre)))
print("fwd_fc1_gelu %.1f us" % t(lambda: _ext.gemm_fwd(x, w1, b1, gelu=True)))
print("fwd_fc1_bias %.1f us" % t(lambda: _ext.gemm_fwd(x, w1, b1)))
