"""the floor table of ONE twin level of the fusion chain (forward and backward): for every launch its
problem, algorithmic bytes and flops, the duration measured inside the step (profiles/r06_c3_one_step_trace.csv, level 4 of
12) and a floor = dependent-launch floor of a replayed graph on this box (1.56 us, tools/probes/launch_floor.hip) + one
dependent memory round trip from MALL (0.8 us: the operands were written by the previous kernel and left the XCD's L2 at the
kernel boundary) + max(bytes / 6.3 TB/s achievable HBM, flops / 2.5 PFLOP/s).  -> profiles/r06_fusion_floor.txt

    python tools/fusion_floor.py [trace.csv] > profiles/r06_fusion_floor.txt
"""
import csv, sys

trace = sys.argv[1] if len(sys.argv) > 1 else "profiles/r06_c3_one_step_trace.csv"
rows = [r for r in csv.DictReader(open(trace)) if r["stream"] == "1"]
B, L, D, H = 16, 20, 768, 12
M = 2 * B * L                                   # both text streams stacked: 640 rows
R2D, R3D = B * (1025 + L), B * (256 + L)        # rows of the two K/V projections: image / object tokens + the other stream's states
RKV = R2D + R3D
MB = 1e6


def gemm(m, n, k, outs=1, w_sets=2, extra=0.0):
    """two streams with their own weights (w_sets = 2): bytes of x, W, outputs; flops"""
    return (m * k * 2 + w_sets * n * k * 2 + outs * m * n * 2 + extra) / MB, 2.0 * m * n * k / 1e9


def ln(reads, writes):
    return (reads + writes) * M * D * 2 / MB, 0.0


att_small = ((M * 3 * D * 2) + M * D * 2) / MB, 4.0 * (2 * B) * H * L * L * 64 / 1e9
att_pair = ((M * D * 2) + RKV * 2 * D * 2 + M * D * 2) / MB, 4.0 * B * H * L * (1025 + L + 256 + L) * 64 / 1e9
FWD = [("K/V projection over cat(image | object tokens, other stream's states): %d rows x 768 -> 1536 (gemm128, row-mapped)" % RKV, gemm(RKV, 2 * D, D)),
       ("Q/K/V projection, 640 rows x 768 -> 2304 (gemm64 grouped)", gemm(M, 3 * D, D)),
       ("self-attention over 20 tokens (attn_fwd)", att_small),
       ("attention output projection 768 -> 768 (gemm64)", gemm(M, D, D)),
       ("dropout + add + LayerNorm (two row groups)", ln(2, 1)),
       ("cross-attention query projection 768 -> 768 (gemm64)", gemm(M, D, D)),
       ("the two cross-attentions, 20 queries x 1045 / 276 keys (attn_fwd_narrow_pair)", att_pair),
       ("cross-attention output projection (gemm64)", gemm(M, D, D)),
       ("dropout + add + LayerNorm", ln(2, 1)),
       ("fc1 + GELU 768 -> 3072, two outputs (gemm64)", gemm(M, 4 * D, D, outs=2)),
       ("fc2 3072 -> 768 (gemm64)", gemm(M, D, 4 * D)),
       ("dropout + add + LayerNorm", ln(2, 1))]
BWD = [("fc2 input gradient x GELU' 768 -> 3072 (gemm64, EPI_DGELU; reads the pre-activation)", gemm(M, 4 * D, D, extra=M * 4 * D * 2)),
       ("fc1 input gradient 3072 -> 768 + residual-branch gradient (gemm64, EPI_ADD)", gemm(M, D, 4 * D, extra=M * D * 2)),
       ("LayerNorm + dropout backward", ln(4, 2)),
       ("cross-attention output projection, input gradient (gemm64)", gemm(M, D, D)),
       ("cross-attentions backward, dQ (attn_bwd_dq_narrow_pair)", (att_pair[0] + M * D * 2 / MB, 2 * att_pair[1])),
       ("cross-attentions backward, dK / dV (attn_bwd_dkv_pair; writes d(K/V) of all %d rows)" % RKV, (att_pair[0] + RKV * 2 * D * 2 / MB, 2.5 * att_pair[1])),
       ("query projection input gradient + residual-branch gradient (gemm64, EPI_ADD)", gemm(M, D, D, extra=M * D * 2)),
       ("LayerNorm + dropout backward", ln(4, 2)),
       ("attention output projection, input gradient (gemm64)", gemm(M, D, D)),
       ("self-attention backward, dQ and dK/dV in one launch (attn_bwd_small)", (2 * att_small[0], 2.5 * att_small[1])),
       ("Q/K/V projection input gradient + residual-branch gradient (gemm64, EPI_ADD)", gemm(M, D, 3 * D, extra=M * D * 2)),
       ("K/V projection input gradient, %d rows x 1536 -> 768, accumulated over the levels (gemm128, EPI_ADD)" % RKV, gemm(RKV, D, 2 * D, extra=RKV * D * 2)),
       ("LayerNorm + dropout backward (the previous level's last)", ln(4, 2))]


def window(pat, k):
    idx = [i for i, r in enumerate(rows) if pat in r["name"]]
    return idx[k], idx[k + 1]


def floor(mb, gf):
    return 1.56 + 0.8 + max(mb / 6.3, gf / 2.5)   # MB / (6.3 MB per us), GFLOP / (2.5 GFLOP per us)


def emit(title, spec, launches):
    print(title)
    print("%-118s %9s %8s %9s %9s" % ("launch", "MB", "GFLOP", "in-step", "floor us"))
    tm = tf = 0.0
    for (name, (mb, gf)), r in zip(spec, launches):
        d = float(r["dur_us"])
        f = floor(mb, gf)
        tm, tf = tm + d, tf + f
        print("%-118s %9.1f %8.2f %9.1f %9.1f   %s" % (name[:118], mb, gf, d, f, r["name"].replace("void ", "").replace("bq::", "")[:38]))
    print("%-118s %9s %8s %9.1f %9.1f\n" % ("total of the level (%d launches)" % len(spec), "", "", tm, tf))
    return tm, tf


print(__doc__.split("\n\n")[0] + "\n")
a, b = window("attn_fwd_narrow_pair", 3)
# a level's forward = the 6 launches in front of its pair attention (K/V .. query projection) + the pair + 5 behind it
f = emit("FORWARD, twin level 4 of 12", FWD, rows[a - 6:a + 6])
a, b = window("attn_bwd_dkv_pair", 3)
# backward order: fc2 dX .. LN (3 launches), cross projection dX, dq pair, dkv pair, then 7 more up to the next level's first launch
bw = rows[a - 5:a + 1] + rows[a + 1:a + 8]
g = emit("BACKWARD, twin level 4 of 12 (weight gradients are parked and flushed in grouped launches after the chain)", BWD, bw)
print("12 levels: forward %.2f ms in the step against %.2f ms of floor; backward %.2f ms against %.2f ms of floor.  The two big launches of a level "
      "(K/V projection, pair attention) run at 38 - 53 %% of this floor's rate -- the GEMM family's and the narrow attention's usual efficiency --, the "
      "ten small ones at 2 - 3 x their floor: each is one load -> compute -> store round trip of a 240-workgroup launch with a 100 - 140 KB operand "
      "panel per CU (DESIGN.md section 5.3).  What this floor leaves out for the small GEMMs: a 640 x 768 x 768 launch on 64 x 32 tiles moves "
      "240 x 144 KB = 34.6 MB from L2 to LDS (every weight panel is staged by 20 row tiles, every row panel by 12 column tiles) -- 2.9 us at the "
      "11.9 TB/s this fabric sustained for the GEMM family (profiles/EXPERIMENTS_r1-r5.md section 4.5); larger tiles halve the bytes and quarter the CUs "
      "that pull them.  With that term a K = 768 projection is 1.56 + 0.8 + 2.9 + ~0.5 (MFMA tail, epilogue) = 5.8 us against 6.6 - 7.7 measured."
      % (12 * f[0] / 1e3, 12 * f[1] / 1e3, 12 * g[0] / 1e3, 12 * g[1] / 1e3))
