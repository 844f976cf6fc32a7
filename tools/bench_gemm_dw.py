"""The grouped weight-gradient launch of the image encoder (12 blocks x 4 linears = 48 problems, 16400-row contraction, fp32
out, bias gradients from the same launch) on the 256 x 256 kernel (one workgroup per CU) against the 256 x 128 persistent
kernel (two per CU), for several ways of cutting the 48 problems into launches (<= 36 problems each).

    python tools/bench_gemm_dw.py [reps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bridgeqa_amd import _ext  # noqa: E402
from bridgeqa_amd.fusion_wgrad import plan_big_launches  # noqa: E402

M = 16400
SHAPES = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]


def main():
    args = [a for i, a in enumerate(sys.argv[1:]) if not a.startswith("--") and sys.argv[i] != "--rows"]
    reps = int(args[0]) if args else 5
    global M
    if "--rows" in sys.argv:           # contraction length (default: c3's 16400; c5: 131104)
        M = int(sys.argv[sys.argv.index("--rows") + 1])
    quick = "--quick" in sys.argv      # the per-shape launches of the 256 x 128 kernel only (tools/ablate_gemm_mid.sh)
    alias = "--alias" in sys.argv      # every block's dY is the same tensor: the operands of a launch fit the MALL
    dev = torch.device("cuda:0")
    rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
    flags = _ext.GEMM_P_XC | _ext.GEMM_Q_XC | _ext.GEMM_OUT_F32
    xs = {k: rnd(M, k) for k in (768, 3072)}
    probs = []
    dys = {n: rnd(M, n) for n, _ in SHAPES}
    for blk in range(12):
        for n, k in SHAPES:
            probs.append(dict(P=xs[k], Q=dys[n] if alias else rnd(M, n), out=torch.empty(n, k, device=dev), colsum=torch.empty(n, device=dev)))
    flops = sum(2.0 * M * n * k for n, k in SHAPES) * 12

    def run(tile, groups):
        for g in groups:
            _ext.gemm_grouped([probs[i] for i in g], flags, _ext.EPI_NONE, tile)

    def timeit(tile, groups):
        run(tile, groups)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(tile, groups)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2]

    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    t256 = [-(-p["Q"].shape[1] // 256) * -(-p["P"].shape[1] // 256) for p in probs]
    g256, moved = plan_big_launches(t256, cus)
    seq = list(range(48))
    cases = [("256 planned (moved %d to the 64-tile kernel: not run here)" % len(moved), 256, g256),
             ("256 36+12", 256, [seq[:36], seq[36:]]),
             ("128 36+12", 128, [seq[:36], seq[36:]]),
             ("128 24+24", 128, [seq[:24], seq[24:]]),
             ("128 16x3", 128, [seq[:16], seq[16:32], seq[32:]]),
             ("128 12x4", 128, [seq[i:i + 12] for i in range(0, 48, 12)])]
    # per layer kind (12 problems of one shape): the K loop's own rate
    for li, (n, k) in enumerate(SHAPES):
        idx = [b * 4 + li for b in range(12)]
        cases.append(("128 12 x (%d x %d)" % (n, k), 128, [idx]))
        cases.append(("256 12 x (%d x %d)" % (n, k), 256, [idx]))
    if quick:
        cases = [c for c in cases if c[0].startswith("128 12 x")]
    for name, tile, groups in cases:
        ms = timeit(tile, groups)
        fl = flops if len(groups) != 1 or len(groups[0]) != 12 else sum(2.0 * M * probs[i]["Q"].shape[1] * probs[i]["P"].shape[1] for i in groups[0])
        print("%-60s %8.3f ms  %6.1f TFLOP/s" % (name, ms, fl / ms / 1e9), flush=True)
    if quick:
        return
    # parity of the two kernels on the whole set (fp32 accumulation order differs: close, not equal)
    run(256, [seq[:36], seq[36:]])
    ref = [(p["out"].clone(), p["colsum"].clone()) for p in probs]
    run(128, [seq[:36], seq[36:]])
    worst = max(((p["out"] - r[0]).abs().max() / r[0].abs().max()).item() for p, r in zip(probs, ref))
    worst_b = max(((p["colsum"] - r[1]).abs().max() / r[1].abs().max()).item() for p, r in zip(probs, ref))
    print("256 vs 128: max rel diff dW %.2e  db %.2e" % (worst, worst_b))


if __name__ == "__main__":
    main()
