"""The text side's deferred weight gradients alone (contractions of 320 / 80 rows: 12 twin levels x 2 streams x 6 linears,
12 decoder layers x 6 linears + their hoisted K/V): fusion_wgrad.flush_deferred_items with the short contractions on the
64 x 64-tile kernel against the persistent 256 x 128 kernel's weight-gradient form (fusion_wgrad._SHORT_DW_TILE).
Each form captured in a HIP graph (as the step replays it); HIP events, median of 15 replays each, interleaved."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bridgeqa_amd import fusion_wgrad  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev).to(torch.bfloat16)
LIN = [(2304, 768), (768, 768), (768, 768), (768, 768), (3072, 768), (768, 3072)]
items, params = [], []


def add(M, N, K):
    w, b = torch.nn.Parameter(torch.empty(N, K, device=dev)), torch.nn.Parameter(torch.empty(N, device=dev))
    params.extend([w, b])
    items.append((rnd(M, N), rnd(M, K), [w], [b]))


if "--min-rows" in sys.argv:   # fusion_wgrad._SHORT_DW_MIN_ROWS: contractions from this many rows on take the tile under test
    k = sys.argv.index("--min-rows")
    fusion_wgrad._SHORT_DW_MIN_ROWS[0] = int(sys.argv[k + 1])
    del sys.argv[k:k + 2]
WHAT = sys.argv[1] if len(sys.argv) > 1 else "all"   # all | twin | decoder
if WHAT in ("all", "twin"):
    for lvl in range(12):
        for stream in range(2):
            for N, K in LIN:
                add(320, N, K)
if WHAT in ("all", "decoder"):
    for layer in range(12):
        for N, K in LIN:
            add(80, N, K)
        add(320, 1536, 768)
out_bytes = sum(it[0].shape[1] * it[1].shape[1] * 4 for it in items)
print("%d problems, %.2f GB of fp32 gradients written per flush" % (len(items), out_bytes / 1e9))


def flush():
    for p in params:
        p.grad = None
    fusion_wgrad.flush_deferred_items(items)


# the step replays the flush from a HIP graph (no host time between its launches): so does this
graphs = {}
side = torch.cuda.Stream()
TILES = (64, 128, 256) if fusion_wgrad._SHORT_DW_MIN_ROWS[0] >= 128 else (64, 256)   # (the 256 x 128 form needs >= 128 rows)
for tile in TILES:
    fusion_wgrad._SHORT_DW_TILE[0] = tile
    with torch.cuda.stream(side):
        for _ in range(3):
            flush()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        flush()
    graphs[tile] = gr
torch.cuda.synchronize()
res = {t: [] for t in TILES}
for rep in range(15):
    for tile in TILES:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); graphs[tile].replay(); e1.record()
        torch.cuda.synchronize()
        res[tile].append(e0.elapsed_time(e1))
for tile in TILES:
    v = sorted(res[tile])
    print("short contractions on tile %3d: median %.3f ms  min %.3f  (%.2f TB/s of gradient writes)" % (
        tile, v[len(v) // 2], v[0], out_bytes / (v[len(v) // 2] * 1e-3) / 1e12))
# parity of the two forms
fusion_wgrad._SHORT_DW_TILE[0] = 64
flush(); a = [p.grad.clone() for p in params]
fusion_wgrad._SHORT_DW_TILE[0] = TILES[-1]
flush(); b = [p.grad for p in params]
print("max |difference| between the forms: %.3e" % max((x - y).abs().max().item() for x, y in zip(a, b)))
