"""The image encoder's eight GEMM launches per block at c3 (16 images x 1025 tokens = 16400 rows) in three forms:
  plain   : one problem of 16400 rows (129 row tiles of 128: 387 / 1161 / 1548 tiles on 512 slots)
  split   : the 16 x 1024 patch rows as ONE batched-row-map problem (128 row tiles: 384 / 1152 / 1536 tiles = 3 whole rounds
            at N = 3072) + the 16 class-token rows as a strided-row problem on the small-tile kernel (a second launch)
  grouped : the same two problems in ONE launch of the 256 x 128 kernel (the class rows cost 3 / 9 / 12 ragged tiles)
Each form captured in a HIP graph of REP launches, replayed interleaved; HIP events, median per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bridgeqa_amd import _ext  # noqa: E402

dev = torch.device("cuda:0")
B, L = 16, 1025
M = B * L
REP = 8
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev).to(torch.bfloat16)

# (name, N out, K contraction, kind)
SHAPES = [("qkv fwd", 2304, 768, "fwd"), ("proj fwd", 768, 768, "fwd"), ("fc1 fwd+gelu", 3072, 768, "gelu"),
          ("fc2 fwd", 768, 3072, "fwd"), ("dX qkv", 768, 2304, "dx"), ("dX proj", 768, 768, "dx"),
          ("dX fc1", 768, 3072, "dx"), ("dX fc2*gelu'", 3072, 768, "dgelu")]
ONLY = sys.argv[1:]


def problems(kind, x, w, bias, y, y2, aux, form):
    """x (M, K), y (M, N) ... as the launch lists of a form"""
    K, N = x.shape[1], y.shape[1]
    v3 = lambda t: t.view(B, L, t.shape[1])
    body = lambda t: None if t is None else v3(t)[:, 1:]
    head = lambda t: None if t is None else v3(t)[:, 0]
    pxc = kind in ("dx", "dgelu")
    flags = _ext.GEMM_P_XC if pxc else 0
    epi = {"fwd": _ext.EPI_BIAS, "gelu": _ext.EPI_BIAS_GELU, "dx": _ext.EPI_NONE, "dgelu": _ext.EPI_DGELU}[kind]
    full = dict(P=w, Q=x, out=y, bias=bias, out2=y2, aux=aux)
    pb = dict(P=w, Q=body(x), out=body(y), bias=bias, out2=body(y2), aux=body(aux))
    ph = dict(P=w, Q=head(x), out=head(y), bias=bias, out2=head(y2), aux=head(aux))
    if form == "plain":
        return [([full], flags, epi, None)]
    if form == "split":
        return [([pb], flags, epi, 128), ([ph], flags, epi, None)]
    return [([pb, ph], flags, epi, 128)]


for name, N, K, kind in SHAPES:
    if ONLY and not any(o in name for o in ONLY):
        continue
    fwd = kind in ("fwd", "gelu")
    x = rnd(M, K)
    w = rnd(N, K) if fwd else rnd(K, N)   # dX: dy (M, K_contraction) @ w (K_contraction, N_out) read transposed
    bias = torch.randn(N, generator=g).to(dev) if fwd else None
    aux = rnd(M, N) if kind == "dgelu" else None
    outs = {}
    graphs = {}
    side = torch.cuda.Stream()
    for form in ("plain", "split", "grouped"):
        y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        y2 = torch.zeros_like(y) if kind == "gelu" else None
        outs[form] = (y, y2)
        launches = problems(kind, x, w, bias, y, y2, aux, form)

        def run():
            for pr, fl, ep, tile in launches:
                _ext.gemm_grouped(pr, fl, ep, tile)
        with torch.cuda.stream(side):
            run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(REP):
                run()
        graphs[form] = gr
    torch.cuda.synchronize()
    res = {f: [] for f in graphs}
    for rep in range(15):
        for f, gr in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record()
            torch.cuda.synchronize()
            res[f].append(e0.elapsed_time(e1) * 1e3 / REP)
    ref = outs["plain"]
    diff = max(float((outs[f][k].float() - ref[k].float()).abs().max()) for f in ("split", "grouped")
               for k in range(2) if ref[k] is not None)
    flop = 2.0 * M * N * K
    line = "%-14s N=%4d K=%4d :" % (name, N, K)
    for f in ("plain", "split", "grouped"):
        v = sorted(res[f])
        med = v[len(v) // 2]
        line += "  %s %6.1f us (%4.0f TF/s)" % (f, med, flop / med / 1e6)
    print(line + "   max |diff| %.1e" % diff, flush=True)
