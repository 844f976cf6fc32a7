"""A few launches of the MFMA GEMM family (csrc/gemm.hip, csrc/gemm_mid.hip) at the ViT shapes of config c3 for rocprofv3
passes (kernel trace, --pmc): exactly the launches the c3 step runs, with the tile the library picks for them.  Random bf16
operands.

    python tools/gemm_once.py [reps] [--only NAME]      NAME in LAUNCHES below (one form per process keeps the counter
                                                        rows of different shapes apart: they share kernel and grid)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bridgeqa_amd import _ext  # noqa: E402

M = 16400
SHAPES = {"qkv": (2304, 768), "proj": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}
# name -> (kind, layer): the four forward launches of a ViT block, its four input-gradient launches, the grouped dW launch
LAUNCHES = {"fwd_qkv": ("fwd", "qkv"), "fwd_proj": ("fwd", "proj"), "fwd_fc1_gelu": ("fwdg", "fc1"), "fwd_fc2": ("fwd", "fc2"),
            "dx_qkv": ("dx", "qkv"), "dx_proj": ("dx", "proj"), "dx_fc1": ("dx", "fc1"), "dx_fc2_dgelu": ("dxg", "fc2"),
            "dx_qkv_add": ("dxa", "qkv"),   # (a second gradient added in the epilogue: the text side's form, here at ViT size)
            "dw_grouped48": ("dw", None), "text_fc1_gelu": ("text", None)}


def algorithmic(name):
    """(flops, bytes) of one launch: 2 M N K; operands + outputs read / written once (bf16; fp32 dW)"""
    kind, layer = LAUNCHES[name]
    if kind == "dw":
        fl = sum(2.0 * M * n * k for n, k in SHAPES.values()) * 12
        by = sum(2.0 * M * (n + k) + 4.0 * n * k for n, k in SHAPES.values()) * 12
        return fl, by
    if kind == "text":
        return 2.0 * 320 * 3072 * 768, 2.0 * (320 * 768 + 3072 * 768 + 2 * 320 * 3072)
    n, k = SHAPES[layer]
    fl = 2.0 * M * n * k
    if kind in ("fwd", "fwdg"):
        by = 2.0 * (M * k + n * k + M * n * (2 if kind == "fwdg" else 1))
    else:  # dx: dy (M, n) @ w (n, k) -> (M, k) [+ aux (M, k)]
        by = 2.0 * (M * n + n * k + M * k * (1 if kind == "dx" else 2))
    return fl, by


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = int(args[0]) if args else 3
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    dev = torch.device("cuda:0")
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
    for name, (kind, layer) in LAUNCHES.items():
        if only and name != only:
            continue
        if kind == "dw":   # exactly what the step issues: the deferred flush (launch plan, bias gradients in the same launches)
            from bridgeqa_amd import fusion_wgrad
            items, params = [], []
            for blk in range(12):
                for n, k in SHAPES.values():
                    wp, bp = torch.nn.Parameter(torch.empty(n, k, device=dev)), torch.nn.Parameter(torch.empty(n, device=dev))
                    params += [wp, bp]
                    items.append((rnd(M, n), rnd(M, k), [wp], [bp]))
            for _ in range(max(1, reps // 2)):
                for p_ in params:
                    p_.grad = None
                fusion_wgrad.flush_deferred_items(items)
        elif kind == "text":
            x, w, b = rnd(320, 768), rnd(3072, 768, sc=0.05), torch.randn(3072, device=dev)
            for _ in range(reps):
                _ext.gemm_fwd(x, w, b, gelu=True)
        else:
            n, k = SHAPES[layer]
            x, w, dy, b, pre = rnd(M, k), rnd(n, k, sc=0.05), rnd(M, n), torch.randn(n, device=dev), rnd(M, k)
            wt = w.t().contiguous()   # the step's input-gradient launches read the K-contiguous copy (fusion_state.transposed_shadow)
            for _ in range(reps):
                if kind == "fwd":
                    _ext.gemm_fwd(x, w, b)
                elif kind == "fwdg":
                    _ext.gemm_fwd(x, w, b, gelu=True)
                elif kind == "dx":
                    _ext.gemm_dx(dy, w, wt=wt)
                elif kind == "dxa":
                    _ext.gemm_dx(dy, w, add=pre, wt=wt)      # (the residual-branch gradient rides on the dX epilogue)
                else:
                    _ext.gemm_dx(dy, w, pre_act=pre, wt=wt)
        torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
