import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bridgeqa_amd import _ext
lib = _ext._lib
lib.bq_debug_clock_mhz.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(4, device="cuda")
for blocks in (1, 16, 256, 2048):
    for it in (100000, 1000000):
        lib.bq_debug_clock_mhz(out.data_ptr(), blocks, it, None)
        torch.cuda.synchronize()
        print("blocks", blocks, "iters", it, "MHz", out[0].item(), "cycles/iter", out[2].item() / it)
