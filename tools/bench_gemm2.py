"""csrc/gemm.hip against torch (hipBLASLt) on the GEMM shapes of config c3, random bf16 operands, GPU time only
(20 launches captured in a HIP graph, replayed; both arms interleaved in one process).

    python tools/bench_gemm2.py [--small] [--json out.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridgeqa_amd import _ext  # noqa: E402

dev = torch.device("cuda:0")


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)


_STREAM = []


def graph_time(fn, reps=20, rounds=5):
    if not _STREAM:
        _STREAM.append(torch.cuda.Stream())   # (one capture stream: the stream-K workspace is registered per stream)
    s = _STREAM[0]
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(reps):
            fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / reps)
    best.sort()
    return best[0], best[len(best) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true", help="text-side shapes only")
    ap.add_argument("--big", action="store_true", help="image-encoder shapes only (256 x 256 against 256 x 128 tiles)")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    M = 16400
    big = [("fwd qkv+bias", "fwd", M, 2304, 768), ("fwd proj+bias", "fwd", M, 768, 768),
           ("fwd fc1+bias+gelu", "fwdg", M, 3072, 768), ("fwd fc2+bias", "fwd", M, 768, 3072),
           ("fwd twin K/V", "fwd", 16720, 1536, 768),
           ("dx qkv", "dx", M, 2304, 768), ("dx proj", "dx", M, 768, 768), ("dx fc1", "dx", M, 3072, 768),
           ("dx fc2+dgelu", "dxg", M, 768, 3072),
           ("dw qkv", "dw", M, 2304, 768), ("dw proj", "dw", M, 768, 768), ("dw fc1", "dw", M, 3072, 768),
           ("dw fc2", "dw", M, 768, 3072)]
    small = []
    for m in (320, 80, 4416):
        small += [("fwd qkv M=%d" % m, "fwd", m, 2304, 768), ("fwd proj M=%d" % m, "fwd", m, 768, 768),
                  ("fwd fc1+gelu M=%d" % m, "fwdg", m, 3072, 768), ("fwd fc2 M=%d" % m, "fwd", m, 768, 3072),
                  ("dx proj M=%d" % m, "dx", m, 768, 768), ("dx fc2+dgelu M=%d" % m, "dxg", m, 768, 3072),
                  ("dw proj M=%d" % m, "dw", m, 768, 768), ("dw fc1 M=%d" % m, "dw", m, 3072, 768)]
    rows = []
    for name, kind, m, n, k in (small if args.small else big if args.big else big + small):
        # (m, n, k): rows, out features, in features of the LINEAR LAYER; fwd: x(m,k) w(n,k); dx: dy(m,n) w(n,k);
        # dw: dy(m,n) x(m,k)
        x, w, dy = rnd(m, k), rnd(n, k, scale=0.05), rnd(m, n)
        b = torch.randn(n, device=dev)
        bb = b.to(torch.bfloat16)
        pre = rnd(m, k)
        flops = 2.0 * m * n * k
        tiles = [256, 128] if m >= 1024 else [64, 32]
        if kind == "dw":
            tiles = [256, 64] if m >= 1024 else [64]
        res = {}
        wt = w.t().contiguous() if kind in ("dx", "dxg") else None
        arms = list(tiles)
        if m >= 1024 and kind in ("dx", "dxg"):
            arms.append("128t")            # what the step launches: the K-contiguous transposed weight copy (fusion_state)
        if m >= 1024 and kind in ("fwd", "dx") and k * (n if kind == "dx" else 1) >= 1536 and (n if kind == "fwd" else k) <= 1024:
            arms.append("128t-streamk" if kind == "dx" else "128-streamk")   # the same launch cut into equal K-tile runs
            arms.append("128t-streamk256" if kind == "dx" else "128-streamk256")   # ... on the 256 x 256 kernel
        for arm in arms:
            tile = 128 if isinstance(arm, str) else arm
            use_wt = isinstance(arm, str) and arm.startswith("128t")
            _ext.streamk_enable(isinstance(arm, str) and arm.endswith("streamk"))
            _ext.streamk256_enable(isinstance(arm, str) and arm.endswith("streamk256"))
            if use_wt:
                if kind == "dx":
                    mine = lambda: _ext.gemm_dx(dy, w, tile=tile, wt=wt)
                else:
                    mine = lambda: _ext.gemm_dx(dy, w, pre_act=pre, tile=tile, wt=wt)
                res[arm] = graph_time(mine)
                continue
            if kind == "fwd":
                mine = lambda: _ext.gemm_fwd(x, w, b, tile=tile)
            elif kind == "fwdg":
                mine = lambda: _ext.gemm_fwd(x, w, b, gelu=True, tile=tile)
            elif kind == "dx":
                mine = lambda: _ext.gemm_dx(dy, w, tile=tile)
            elif kind == "dxg":
                mine = lambda: _ext.gemm_dx(dy, w, pre_act=pre, tile=tile)
            else:
                mine = lambda: _ext.gemm_dw(dy, x, tile=tile)
            res[arm] = graph_time(mine)
        _ext.streamk_enable(False)
        _ext.streamk256_enable(False)
        if kind == "fwd":
            ref = lambda: torch.nn.functional.linear(x, w, bb)
        elif kind == "fwdg":
            ref = lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x, w, bb))
        elif kind == "dx":
            ref = lambda: torch.mm(dy, w)
        elif kind == "dxg":
            ref = lambda: torch.ops.aten.gelu_backward(torch.mm(dy, w), pre)
        else:
            ref = lambda: torch.mm(dy.t(), x, out_dtype=torch.float32)
        tref = graph_time(ref)
        best_tile = min(res, key=lambda t_: res[t_][0])
        row = dict(name=name, m=m, n=n, k=k, torch_us=round(tref[0], 2), torch_tflops=round(flops / tref[0] / 1e6, 1),
                   bq_us={str(t_): round(v[0], 2) for t_, v in res.items()}, bq_med_us={str(t_): round(v[1], 2) for t_, v in res.items()},
                   bq_tflops=round(flops / res[best_tile][0] / 1e6, 1), best_tile=best_tile)
        rows.append(row)
        print("%-24s m=%5d n=%4d k=%4d  bq %s us (%.0f TF/s)   torch %.1f us (%.0f TF/s)" % (
            name, m, n, k, " ".join("%s:%.1f" % (t_, v[0]) for t_, v in res.items()), row["bq_tflops"], tref[0],
            row["torch_tflops"]), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
