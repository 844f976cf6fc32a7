"""Do CU-masked HIP streams give real spatial partitioning (also for graph replays)?  Stream A: GPU-filling GEMM
chain restricted to CUs [NDET, 256); stream B: chain of tiny kernels + a few mid-size ones restricted to [0, NDET)."""
import ctypes, sys, time
import torch

NDET = int(sys.argv[1]) if len(sys.argv) > 1 else 48
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda")
torch.zeros(1, device=dev)


def masked_stream(lo, hi):
    words = (ctypes.c_uint32 * 8)()
    for i in range(lo, hi):
        words[i // 32] |= (1 << (i % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
x = torch.randn(64, 1024, device=dev)
big_in = torch.randn(16 * 2048 * 64, 64, device=dev, dtype=torch.bfloat16)


def big():
    y = a
    for _ in range(40):
        y = (y @ a) * 0.01
    return y


def small():
    y = x
    for _ in range(300):
        y = y * 1.0001
    z = big_in
    for _ in range(10):
        z = z * 1.0001   # 268 MB read + write: memory-bound mid-size kernels
    return y, z


def capture(fn, s):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    torch.cuda.synchronize()
    return g


def timed(f, n=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for label, sa, sb in (("plain streams", torch.cuda.Stream(), torch.cuda.Stream()),
                      ("CU-masked %d | %d" % (256 - NDET, NDET), masked_stream(NDET, 256), masked_stream(0, NDET))):
    ga, gb = capture(big, sa), capture(small, sb)
    def on(s, g):
        with torch.cuda.stream(s):
            g.replay()
    ta = timed(lambda: on(sa, ga)); tb = timed(lambda: on(sb, gb))
    def both():
        on(sa, ga); on(sb, gb)
    def both_r():
        on(sb, gb); on(sa, ga)
    print("%-22s big alone %.2f ms  small alone %.2f ms  both %.2f ms  both(small first) %.2f ms" %
          (label, ta, tb, timed(both), timed(both_r)), flush=True)
