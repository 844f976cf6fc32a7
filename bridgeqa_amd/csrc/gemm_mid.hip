// gemm128_kernel: the forward / input-gradient GEMMs of the image encoder and of the K/V projections over the image
// tokens (reference models/vit.py:30-32,51-53, models/med.py:112-118: nn.Linear + bias / GELU, and their autograd dX) on
// 256 (i) x 128 (j) output tiles, PERSISTENT workgroups, TWO of them co-resident per CU.
//
// Why a second tile shape: the 256 x 256 kernel of csrc/gemm.hip keeps the matrix pipe 58 % busy inside
// its K loop, but these contractions are SHORT (K = 768: 12 K tiles) against a prologue (first operands from HBM) and an
// epilogue (128 KB of output per workgroup, written by all 256 workgroups of a round at once), and 16400 rows are 64.06 row
// tiles: whole launches ended at 19-33 % MFMA busy.  One workgroup per CU cannot hide its own epilogue.  Here a workgroup
// is 4 waves (one per SIMD, <= 256 VGPRs) with 64 KB of LDS, so a CU holds two INDEPENDENT workgroups: one's epilogue and
// its barrier / LDS-read / DMA-issue gaps run under the other's MFMA clusters, and the two drift apart by themselves, so
// the output stores of the chip are spread over the launch instead of arriving in lock-step bursts.  Ablations of the
// non-persistent first version (tools/bench_gemm_ablate.py) put ~11 us of every ~28 us tile of the K = 768 shapes OUTSIDE
// the K loop (workgroup launch, address set-up, first operands' latency, epilogue through LDS), hence the second step:
//   * persistent workgroups (grid = min(tiles, 2 x CUs)) walking tiles v, v + grid, ... of an XCD-aware order;
//   * ONE continuous LDS-DMA stream across tiles: the slots that would stage K tiles past a tile's end stage the first K
//     tiles of the workgroup's NEXT tile (per-group cursors that roll over to the next tile's addresses), so a tile's
//     first operands are already in LDS when its K loop starts -- no prologue after the first tile;
//   * an epilogue that leaves the staging buffers alone (they belong to the next tile by then): a wave-private 2 KB image
//     turns the accumulator layout into whole 128-B lines; every load of the epilogue (bias, second operand) is issued
//     before its first store and the last K tile's two youngest restages behind those loads (vmcnt retires in issue
//     order: a load behind a store waits for the store's acknowledgement, a load behind a fresh DMA for its landing);
//     bounds-checked buffer stores that ALWAYS issue (rows / columns past the edge are dropped by the descriptor / an
//     out-of-range offset), so the counted waits of the next tile's first K tile know how many stores sit in front.
//
// Same formulation, LDS images and fragment maps as gemm256_kernel (gemm_common.h):  out[j][i] = sum_kc P(i,kc) Q(j,kc),
// P K-contiguous (forward: W) or contraction-major (dX: W read transposed with ds_read_b64_tr_b16), Q K-contiguous.
// Wave tile 128 (i) x 64 (j) = gemm256's; waves 2 (i) x 2 (j).  Staging units of 64 rows x 64 k (8 KB):
//   PA0(g) PA1(g)  the two 64-row halves of wave row g's P rows          -- SINGLE-buffered (weights: L2-resident)
//   QB0[b] QB1[b]  the first / second 32 rows of both wave columns       -- DOUBLE-buffered (b = running K tile & 1: the
//                                                                            activation stream comes from HBM / MALL)
// = 64 KB (+ 8 KB of epilogue images).  Per K tile four phases, ONE barrier each ("t+1" rolls over into the next tile):
//   phase   waits for (counted)         reads (LDS -> registers)   MFMA quadrant        restages (LDS-DMA, per wave)
//   p0(t)   vmcnt(6): PA0(t), QB0(t)    PA0, QB0                   A0 x B0              QB1(t+1)  x2   [read next in p1(t+1)]
//   p1(t)   --                          QB1                        A0 x B1              PA0(t+1)  x4   [p0(t+1): 3 phases]
//   p2(t)   vmcnt(6): PA1(t)            PA1                        A1 x B1              QB0(t+2)  x2   [p0(t+2): 6 phases]
//   p3(t)   --                          (B0 kept in registers)     A1 x B0              PA1(t+1)  x4   [p2(t+1): 3 phases]
// A unit is restaged in the phase AFTER its last read (that phase's barrier orders every wave's reads before the DMA);
// vmcnt counts a wave's vector-memory operations in issue order, so vmcnt(6) at p0(t) = "everything up to PA0(t) has
// landed" (issued after it: QB0(t+1) x2, PA1(t) x4) and at p2(t) = "up to PA1(t)" (after it: QB1(t+1) x2, PA0(t+1) x4); the
// barrier that follows makes every wave's DMAs visible to all.  In the FIRST K tile of a continuing tile the previous
// tile's S epilogue stores sit between those DMAs and the wait: vmcnt(6 + S) there.  Past the workgroup's last tile the
// cursors park on the out-of-range sentinel (zeros, no traffic) so that the counts keep their meaning.
//
// STREAM-K form (template flag SK, round 5).  The N = 768 / K = 3072 launches of the image encoder (fc2
// forward, the input gradient through fc1) are 387 tiles of 48 K tiles on 512 workgroup slots: one ragged round, the CUs
// that hold two tiles set the makespan at 96 K-tile executions while the balanced share is 72.6.  With SK a workgroup walks
// a contiguous range of K-TILE UNITS [b(s), b(s+1)), b(s) = s U / G snapped so that no segment is a single K tile
// (U = tiles x K tiles, G = grid, s = the XCD-aware slot of the block): a tail of one tile, whole tiles, a head of the next --
// the same continuous LDS-DMA stream, the cursors simply start a segment at its first K tile.  A tile cut across n
// workgroups is finished by whichever of them ARRIVES LAST (no workgroup ever waits for one that has not run yet, so the
// scheme cannot deadlock under any dispatch order): at the end of its segment a workgroup draws a ticket (agent-scope atomic
// on tick[t].arrive); not last -> it writes its fp32 accumulators to its slab with write-through (sc1) stores, drains them,
// and counts itself in tick[t].done; last -> it waits until done == n - 1 (the others are already past their K loops: a wait
// of microseconds), reads their slabs with sc1 loads, resets the two ticket words and runs the ordinary epilogue.  Protocol
// after MI355X_MICROARCH.md "Valid forms" (sc1 payload, every storing wave's vmcnt(0), workgroup barrier, one lane's
// agent-scope atomic; the reader: one lane polls with relaxed agent loads, workgroup barrier, sc1 loads of every byte).
#include "gemm_common.h"

// measurement builds only (tools/bench_gemm_dw.py): 1 = no MFMAs (the memory stream alone), 2 = no LDS-DMAs
// (matrix pipe + LDS reads alone; results are garbage)
#ifndef BQ_MID_ABLATE
#define BQ_MID_ABLATE 0
#endif

namespace bq {

// ST_AUX: cache policy of the output stores (buffer-store aux bits: 0 default, 2 nt, 16 sc1 = write-through, the line is
// dropped from the XCD's L2 -- MI355X_MICROARCH.md, stores of each flavour)
typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <bool P_XC, bool Q_XC, int EPI, bool OUT_F32, int ST_AUX, bool SK = false>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(const GemmArgs args) {
  static_assert(!OUT_F32 || (P_XC && Q_XC && EPI == EPI_NONE), "fp32 out = the weight-gradient form");
  static_assert(!Q_XC || OUT_F32, "a contraction-major Q = the weight-gradient form");
  static_assert(!SK || (!OUT_F32 && EPI != EPI_BIAS_GELU), "stream-K: bf16 out, one output");
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536 + 4 * 2048 + (SK ? 16 : 0)];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nwg = args.total_tiles, G = (int)gridDim.x;

  // ---- lane-dependent constants ---------------------------------------------------------------------------------------
  const int row16 = lane & 15, q4 = lane >> 4, cp = lane & 7, lr = lane >> 3;
  const int kc_base = row16 * 128 + ((q4 ^ (row16 & 7)) << 4);
  // contraction-major fragment address in closed form (read_frag_cf): a per-lane int[4] table indexed by the wave column
  // is an alloca that hipcc promotes to LDS (+ 4 KB per workgroup, an LDS round trip per fragment address -- DESIGN.md §5.7)
  const int xc_q = (lane & 15) >> 2;
  const int xcg = (xc_q >> 1) | ((q4 & 1) << 1), xc0 = (8 * q4 + xc_q) * 128 + 8 * (lane & 3);
  constexpr int PA0 = 0, PA1 = 16384, QB = 32768;   // QB + buf * 16384 + half * 8192
  const int uA0 = PA0 + wr * 8192, uA1 = PA1 + wr * 8192;
  const int bsub = wc * 2;

  // ---- virtual block id -> (problem, tile): XCD-aware as in gemm256_kernel (blocks b, b + 8, ... share an XCD's L2; G is
  // a multiple of 8 or the whole grid, so v = b + n G stays on b's XCD) -------------------------------------------------
  // nkt: K tiles of this work item (a whole tile, or -- SK -- a segment starting at K tile kt0 of tile t)
  struct Tile { int pi, i0, j0, nkt, kt0, t; };
  auto xcd_order = [](int v, int n) {   // bijection blocks -> positions: the blocks of one XCD (b, b + 8, ...) are neighbours
    const int q = n >> 3, r = n & 7, x = v & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (v >> 3);
  };
  auto decode_t = [&](int t) {
    int pi = 0;
    for (int k = 1; k < args.n; ++k)
      if (t >= args.p[k].tile0) pi = k;
    const GemmProblem &pr = args.p[pi];
    const int tl = t - pr.tile0;
    const int tiles_j = (pr.Nj + 127) >> 7;
    Tile tt;
    tt.pi = pi;
    tt.j0 = (tiles_j - 1 - tl / pr.tiles_i()) * 128;   // ragged last j block first
    tt.i0 = (tl % pr.tiles_i()) * 256;
    tt.nkt = (pr.Kc + 63) >> 6;
    tt.kt0 = 0;
    tt.t = t;
    return tt;
  };
  auto decode = [&](int v) { return decode_t(xcd_order(v, nwg)); };
  // ---- stream-K: unit ranges (one problem per launch; every tile has nkt_full K tiles) -----------------------------------
  const int nkt_full = SK ? (args.p[0].Kc + 63) >> 6 : 1;   // (1: the lambdas below are dead code without SK)
  const int U = SK ? nwg * nkt_full : 0;
  auto bound = [&](int x) {   // first unit of slot x; never 1 K tile away from a tile edge (a segment has >= 2 K tiles)
    int b = (int)((long)x * U / G);
    const int r = b % nkt_full;
    return r == 1 ? b - 1 : (r == nkt_full - 1 ? b + 1 : b);
  };
  auto seg = [&](int u, int ue) {   // the work item starting at unit u of the range that ends at ue
    const int t = u / nkt_full;
    Tile tt = decode_t(t);
    tt.kt0 = u - t * nkt_full;
    const int left = nkt_full - tt.kt0;
    tt.nkt = left < ue - u ? left : ue - u;
    return tt;
  };
  const int slot = SK ? xcd_order((int)blockIdx.x, G) : 0;
  int su = SK ? bound(slot) : 0;
  const int sue = SK ? bound(slot + 1) : 0;

  // ---- the DMA stream: one cursor per staging group, (tile, K tile) advancing independently of the compute side ------
  // a unit = 8 DMAs of 1 KB (8 rows x 128 B), two per wave: unit rows (w + 4 d) * 8 + lane / 8
  unsigned vP[2], vQ[2];                // [half]: the next byte offset to stage of DMA (g = 0, d = 0); the other DMAs of
                                        // the group add wave-uniform deltas (in the bounds-checked vector offset)
  unsigned stepP[2], dgP[2], ddP[2], ddQ[2], stepQ[2];   // per group: bytes per K tile / per wave row g / per DMA d of the cursor's problem
  int remP[2], remQ[2];                 // K tiles left before the cursor rolls over to the next tile
  int rsrcP_pi[2], rsrcQ_pi[2];         // problem of each cursor (its buffer descriptor)
  const unsigned DEAD = 0x80000000u;    // (+ a few deltas and steps stays out of range)
  const int ur0 = wave * 8 + lr;        // unit row of DMA d = 0; d = 1: + 32 (same row & 7, same xg: same chunk swizzle)

  auto set_p = [&](int half, const Tile &tt, bool live) {
    const GemmProblem &pr = args.p[tt.pi];
    const int ldp = pr.ldp;
    unsigned v;
    if (!P_XC) v = (unsigned)(((tt.i0 + half * 64 + ur0) * ldp + (cp ^ (ur0 & 7)) * 8) * 2);
    else v = (unsigned)((ur0 * ldp + tt.i0 + half * 64 + (cp ^ (xg(ur0) << 1)) * 8) * 2);
    stepP[half] = P_XC ? (unsigned)(64 * ldp * 2) : 128u;
    if (SK) v += (unsigned)tt.kt0 * stepP[half];
    vP[half] = live ? v : DEAD;
    dgP[half] = P_XC ? 256u : (unsigned)(128 * ldp * 2);
    ddP[half] = (unsigned)(32 * ldp * 2);
    remP[half] = live ? tt.nkt : 0x40000000;
    rsrcP_pi[half] = tt.pi;
  };
  auto set_q = [&](int half, const Tile &tt, bool live) {
    const GemmProblem &pr = args.p[tt.pi];
    unsigned v;
    if (!Q_XC) {
      // unit row r -> j0 + (r >> 5) * 64 + half * 32 + (r & 31); d = 1 is r + 32: 64 rows further (under a batched-row
      // map the two rows may lie in different batches: the d = 1 delta is per lane then)
      const int row0 = tt.j0 + half * 32 + ur0;
      const unsigned o0 = mapped_row(row0, pr.ldq, pr.q_rpb(), pr.q_bstride);
      v = (o0 + (unsigned)((cp ^ (ur0 & 7)) * 8)) * 2u;
      ddQ[half] = pr.q_rpb() ? (mapped_row(row0 + 64, pr.ldq, pr.q_rpb(), pr.q_bstride) - o0) * 2u : (unsigned)(64 * pr.ldq * 2);
    } else {
      // unit row r = contraction index; its 8 chunks of 8 columns: chunk c -> j0 + (c >> 2) * 64 + half * 32 + (c & 3) * 8
      // (the `half` 32 columns of both wave columns); d = 1 is r + 32
      const int c = cp ^ (xg(ur0) << 1);
      v = (unsigned)((ur0 * pr.ldq + tt.j0 + (c >> 2) * 64 + half * 32 + (c & 3) * 8) * 2);
      ddQ[half] = (unsigned)(32 * pr.ldq * 2);
    }
    stepQ[half] = Q_XC ? (unsigned)(64 * pr.ldq * 2) : 128u;
    if (SK) v += (unsigned)tt.kt0 * stepQ[half];
    vQ[half] = live ? v : DEAD;
    remQ[half] = live ? tt.nkt : 0x40000000;
    rsrcQ_pi[half] = tt.pi;
  };

  int v = blockIdx.x;
  Tile cur = SK ? seg(su, sue) : decode(v);
  bool has_next = SK ? (su + cur.nkt < sue) : (v + G < nwg);
  Tile nxt = has_next ? (SK ? seg(su + cur.nkt, sue) : decode(v + G)) : cur;

  // stage P unit pair `half` (PA0 / PA1: both wave rows, 4 DMAs per wave) at its cursor, advance the cursor
  auto dma_p = [&](int half) {
    const GemmProblem &pr = args.p[rsrcP_pi[half]];
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)pr.P, 0, pr.p_bytes, 0x00020000);
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int d = 0; d < 2; ++d)
        if (BQ_MID_ABLATE != 2)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(smem + (half ? PA1 : PA0) + g * 8192 + (wave + 4 * d) * 1024),
                                                   16, vP[half] + (g * dgP[half] + d * ddP[half]), 0, 0, 0);
    vP[half] += stepP[half];
    if (--remP[half] == 0) set_p(half, nxt, has_next);
  };
  // stage Q unit `half` (QB0 / QB1, 2 DMAs per wave) at its cursor into buffer `par` & 1, advance the cursor
  auto dma_q = [&](int half, unsigned par) {
    const GemmProblem &pr = args.p[rsrcQ_pi[half]];
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)pr.Q, 0, pr.q_bytes, 0x00020000);
#pragma unroll
    for (int d = 0; d < 2; ++d)
      if (BQ_MID_ABLATE != 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(smem + QB + (par & 1) * 16384 + half * 8192 + (wave + 4 * d) * 1024),
                                                 16, vQ[half] + d * ddQ[half], 0, 0, 0);
    vQ[half] += Q_XC ? stepQ[half] : 128u;
    if (--remQ[half] == 0) set_q(half, nxt, has_next);
  };

  // ---- prologue of the workgroup's first tile: the issue order the steady state would have produced before p0(0) -------
  set_q(0, cur, true); set_q(1, cur, true); set_p(0, cur, true); set_p(1, cur, true);
  unsigned gk = 0;   // running K tile count of this workgroup (Q buffer parity)
  dma_q(0, 0); dma_q(1, 0); dma_p(0); dma_q(0, 1); dma_p(1);

  f32x4 acc[8][4];
  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
#define BQ_MID_MFMA(AO, FB, BO)                                                                       \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int a = 0; a < 4; ++a)      \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) if (BQ_MID_ABLATE != 1 || (a == 0 && b == 0)) acc[AO + a][BO + b] =  \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][kk], FB[b][kk], acc[AO + a][BO + b], 0, 0, 0);
#define BQ_MID_COMPUTE_BEGIN()                              \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
  __builtin_amdgcn_sched_barrier(0);                        \
  __builtin_amdgcn_s_setprio(1);
#define BQ_MID_COMPUTE_END()                                \
  __builtin_amdgcn_s_setprio(0);                            \
  __builtin_amdgcn_sched_barrier(0);
  constexpr int NPASS = (EPI == EPI_BIAS_GELU) ? 2 : 1;
  // epilogue stores per wave, always issued: bf16 4 j blocks x 2 i halves x 2 (x passes); fp32 8 x 4 accumulator blocks + 4
  // of the column sums
  constexpr int NSTORE = OUT_F32 ? 36 : 16 * NPASS;
  // weight-gradient form with pr.colsum set: the column sums of Q over the contraction (the layer's BIAS gradient: sum over
  // the rows of dY) from all-ones MFMAs on the B fragments, in the waves wr == 0 of the i = 0 tiles (as gemm256_kernel)
  bf16x8 ones;
  f32x4 qs[4];
  if (OUT_F32) {
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  }
#define BQ_MID_QSUM(FB, BO)                                                                           \
  if (OUT_F32 && do_qsum) {                                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int b = 0; b < 2; ++b)    \
        qs[BO + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, FB[b][kk], qs[BO + b], 0, 0, 0);   \
  }

  for (bool first = true;; first = false) {
    const GemmProblem &pr = args.p[cur.pi];
    const int Ni = pr.Ni, Nj = pr.Nj;
    const int iw = cur.i0 + wr * 128, jw = cur.j0 + wc * 64;
    const bool vA0 = iw < Ni, vA1 = iw + 64 < Ni, vB0 = jw < Nj, vB1 = jw + 32 < Nj;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_qsum = OUT_F32 && pr.colsum != nullptr && cur.i0 == 0 && wr == 0;   // wave-uniform
    if (OUT_F32) {
#pragma unroll
      for (int b = 0; b < 4; ++b) qs[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int kt = 0; kt < cur.nkt; ++kt, ++gk) {
      const unsigned char *qb = smem + QB + (gk & 1) * 16384;
      const bool after_stores = kt == 0 && !first;   // the previous tile's epilogue stores are in front of the wait
      const bool last = kt == cur.nkt - 1;           // its p2 / p3 restages are issued from the epilogue (see there)
      // ---- p0: PA0, QB0 -> A0 x B0 ; restage QB1(t+1)
      if (after_stores) wait_vmcnt<6 + NSTORE>(); else wait_vmcnt<6>();
      BQ_BARRIER();
      // (fragment reads are unconditional -- a skipped read would keep the fragment registers live across the epilogue;
      // only the MFMAs of quadrants that lie wholly past a ragged edge are skipped)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[a][kk] = read_frag_cf_x<P_XC>(smem + uA0, a, kk, kc_base, xc0, xcg);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb0[b][kk] = read_frag_cf_x<Q_XC>(qb, bsub + b, kk, kc_base, xc0, xcg);
      dma_q(1, gk + 1);
      BQ_MID_COMPUTE_BEGIN();
      if (vA0 && vB0) { BQ_MID_MFMA(0, fb0, 0) }
      if (vB0) { BQ_MID_QSUM(fb0, 0) }
      BQ_MID_COMPUTE_END();
      // ---- p1: QB1 -> A0 x B1 ; restage PA0(t+1)
      BQ_BARRIER();
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb1[b][kk] = read_frag_cf_x<Q_XC>(qb + 8192, bsub + b, kk, kc_base, xc0, xcg);
      dma_p(0);
      BQ_MID_COMPUTE_BEGIN();
      if (vA0 && vB1) { BQ_MID_MFMA(0, fb1, 2) }
      if (vB1) { BQ_MID_QSUM(fb1, 2) }
      BQ_MID_COMPUTE_END();
      // ---- p2: PA1 -> A1 x B1 ; restage QB0(t+2)
      if (after_stores) wait_vmcnt<6 + NSTORE>(); else wait_vmcnt<6>();
      BQ_BARRIER();
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[a][kk] = read_frag_cf_x<P_XC>(smem + uA1, a, kk, kc_base, xc0, xcg);
      if (!last) dma_q(0, gk + 2);
      BQ_MID_COMPUTE_BEGIN();
      if (vA1 && vB1) { BQ_MID_MFMA(4, fb1, 2) }
      BQ_MID_COMPUTE_END();
      // ---- p3: (B0 kept in registers) -> A1 x B0 ; restage PA1(t+1)
      BQ_BARRIER();
      if (!last) dma_p(1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      if (vA1 && vB0) { BQ_MID_MFMA(4, fb0, 0) }
      BQ_MID_COMPUTE_END();
    }

    // ---- stream-K: a tile cut across several workgroups is finished by the one that arrives last (header) ------------------
    bool sk_partial = false;
    if (SK && cur.nkt != nkt_full) {
      unsigned *tick = (unsigned *)args.sk_ws + 2 * cur.t;
      float *slabs = (float *)((char *)args.sk_ws + SK_TICKET_BYTES);
      // participants: the slots whose unit range meets [t nkt, (t + 1) nkt) -- a run of consecutive slots around this one
      const int t0 = cur.t * nkt_full, t1 = t0 + nkt_full;
      int s_lo = slot, s_hi = slot;
      while (s_lo > 0 && bound(s_lo) > t0) --s_lo;
      while (s_hi + 1 < G && bound(s_hi + 1) < t1) ++s_hi;
      const int n_part = s_hi - s_lo + 1;
      const unsigned flag_lds = (unsigned)(size_t)((lds_u8_t *)smem) + 65536u + 4u * 2048u;
      if (wave == 0) {
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
        asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(flag_lds), "v"(old) : "memory");
      }
      BQ_BARRIER();
      unsigned arrived;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(arrived) : "v"(flag_lds) : "memory");
      arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)arrived);
      const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void *)slabs, 0, (unsigned)(2 * G) * (unsigned)SK_SLAB_BYTES > 0x7fffffffu
                                                         ? 0x7fffffffu : (unsigned)(2 * G) * (unsigned)SK_SLAB_BYTES, 0x00020000);
      // slab of (slot, which of its two possible partial segments): the first one iff the slot's range starts inside the tile
      auto slab_off = [&](int sl) { return (unsigned)((2 * sl + (bound(sl) >= t0 ? 0 : 1))) * (unsigned)SK_SLAB_BYTES; };
      if (arrived != (unsigned)(n_part - 1)) {
        // not last: park the accumulators.  The last K tile's two youngest restages first (they belong to the next item)
        sk_partial = true;
        __builtin_amdgcn_sched_barrier(0);
        dma_q(0, gk + 1);
        dma_p(1);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned base = slab_off(slot) + (unsigned)(wave * 32 * 1024 + lane * 16);
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[a][b]), rsW, base + (unsigned)((a * 4 + b) * 1024), 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BQ_BARRIER();
        if (wave == 0 && lane == 0) __hip_atomic_fetch_add(tick + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        // last: every other participant is past its K loop; wait for their slabs, take the tickets back, fold
        if (wave == 0 && lane == 0) {
          // (bounded: ~0.5 s; the others are already writing their slabs -- a timeout can only mean a protocol bug, and a
          // wrong tile fails a test where a hang would lose the GPU)
          for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(tick + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(n_part - 1); ++spin)
            __builtin_amdgcn_s_sleep(4);
          __hip_atomic_store(tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(tick + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        BQ_BARRIER();
        for (int sl = s_lo; sl <= s_hi; ++sl) {
          if (sl == slot) continue;
          const unsigned base = slab_off(sl) + (unsigned)(wave * 32 * 1024 + lane * 16);
          // 16 loads (16 KB per wave) in flight per round trip: the fragment registers are free here
#pragma unroll
          for (int a0 = 0; a0 < 8; a0 += 4) {
            u32x4_t part[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
              part[e] = __builtin_amdgcn_raw_buffer_load_b128(rsW, base + (unsigned)((a0 * 4 + e) * 1024), 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a0 + (e >> 2)][e & 3] += __builtin_bit_cast(f32x4, part[e]);
          }
        }
      }
    }

    // ---- epilogue.  accumulator (a, b)[r]: i = iw + a*16 + q4*4 + r, j = jw + b*16 + row16.  The staging buffers belong
    // to the next tile by now, so the bf16 results go through a wave-private 2 KB image of their own -- [16 j][64 i], 16-B
    // chunk c of row j at j*128 + ((c ^ (j & 7)) << 4), one (j block, i half) at a time -- and leave as whole 128-B lines
    // (8 rows per store instruction; stores of 64-B row pieces straight from the accumulators measured 15 % slower on the
    // whole launch).  Bounds-checked buffer stores that ALWAYS issue: the next tile's counted waits know their number.
    if (SK && sk_partial) {
      // (parked: the tile's last arriver stores it)
    } else if (OUT_F32) {
      // fp32 weight gradients straight from the accumulators (16 B per lane, 64-B row pieces; once per ~260 K tiles).  The
      // last K tile's two youngest restages first; every store issues (out-of-range offsets are dropped)
      __builtin_amdgcn_sched_barrier(0);
      dma_q(0, gk + 1);
      dma_p(1);
      __builtin_amdgcn_sched_barrier(0);
      const int ldo = pr.ldo;
      const auto rsO = __builtin_amdgcn_make_buffer_rsrc(pr.out, 0, (unsigned)((long)Nj * ldo * 4), 0x00020000);
      const auto rsS = __builtin_amdgcn_make_buffer_rsrc((void *)pr.colsum, 0, pr.colsum == nullptr ? 0u : (unsigned)(Nj * 4), 0x00020000);
#pragma unroll
      for (int b = 0; b < 4; ++b) {  // every row of the all-ones product holds the sums: lane row16 has column j's in element 0
        const int j = jw + b * 16 + row16;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(qs[b][0]), rsS, (do_qsum && q4 == 0) ? (unsigned)(j * 4) : DEAD, 0, 0);
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = jw + b * 16 + row16;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const int i = iw + a * 16 + q4 * 4;   // Ni % 8 == 0 (host check): a lane's four i are valid together
          const unsigned off = i < Ni ? (unsigned)((j * ldo + i) * 4) : DEAD;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[a][b]), rsO, off, 0, ST_AUX);
        }
      }
    } else {
      const int ldo = pr.ldo;
      const int orpb = pr.o_rpb(), obs = pr.o_bstride;   // batched-row map of out / out2 / aux (0: plain rows)
      const unsigned obytes = orpb ? (unsigned)((long)(Nj / orpb) * obs * 2) : (unsigned)((long)Nj * ldo * 2);
      const auto rsO = __builtin_amdgcn_make_buffer_rsrc(pr.out, 0, obytes, 0x00020000);
      const auto rsO2 = __builtin_amdgcn_make_buffer_rsrc(EPI == EPI_BIAS_GELU ? pr.out2 : pr.out, 0, obytes, 0x00020000);
      const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void *)((EPI == EPI_DGELU || EPI == EPI_ADD) ? (const void *)pr.aux : pr.out), 0, obytes, 0x00020000);
      const int wj = row16 * 128 + (q4 & 1) * 8, wx = row16 & 7, wc2 = q4 >> 1;
      // every load of the epilogue is issued BEFORE its first store: vmcnt retires in issue order, so a load behind a store
      // could only be consumed once that store has been acknowledged (the fragment registers are free by now: the bias of
      // the wave's 128 columns / the second operand of its whole 128 x 64 tile fit)
      // FOLD phase: bias / the second operand go into the accumulators in place, BEFORE the tile's first store -- vmcnt
      // retires in issue order, so a load issued behind a store could only be consumed once that store has been
      // acknowledged.  The last K tile's p2 / p3 restages (QB0 two K tiles ahead, PA1 one ahead: both belong to the next
      // tile) are issued behind the LAST of these loads: the loads then only queue behind DMAs that are >= 2 phases old.
      auto deferred_dmas = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        dma_q(0, gk + 1);   // (gk already counts this tile's last K tile: "K tile + 2" of the stream)
        dma_p(1);
        __builtin_amdgcn_sched_barrier(0);
      };
      if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
        float bias_r[8][4];
        const bool bb = pr.bias_bf16() != 0;
        const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void *)pr.bias, 0, pr.bias == nullptr ? 0 : Ni * (bb ? 2 : 4), 0x00020000);
        const unsigned bo = (unsigned)(iw + q4 * 4);
        if (bb) {
          uint2 raw[8];
#pragma unroll
          for (int a = 0; a < 8; ++a) raw[a] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsB, (bo + a * 16) * 2, 0, 0));
#pragma unroll
          for (int a = 0; a < 8; ++a) {
            bias_r[a][0] = __uint_as_float(raw[a].x << 16); bias_r[a][1] = __uint_as_float(raw[a].x & 0xffff0000u);
            bias_r[a][2] = __uint_as_float(raw[a].y << 16); bias_r[a][3] = __uint_as_float(raw[a].y & 0xffff0000u);
          }
        } else {
#pragma unroll
          for (int a = 0; a < 8; ++a) {
            const float4 t4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (bo + a * 16) * 4, 0, 0));
            bias_r[a][0] = t4.x; bias_r[a][1] = t4.y; bias_r[a][2] = t4.z; bias_r[a][3] = t4.w;
          }
        }
        deferred_dmas();
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] += bias_r[a][r];
      } else if (EPI == EPI_DGELU || EPI == EPI_ADD) {
        uint2 yA[8], yB[8];   // the second operand in the accumulator's own map (8 B per lane), one j block each
        auto fetch = [&](int b, uint2(&dst)[8]) {
          const unsigned off = (mapped_row(jw + b * 16 + row16, ldo, orpb, obs) + (unsigned)(iw + q4 * 4)) * 2u;
#pragma unroll
          for (int a = 0; a < 8; ++a)
            dst[a] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsX, off + a * 32, 0, 0));
          __builtin_amdgcn_sched_barrier(0);
        };
        auto fold = [&](auto b_tag, const uint2(&src)[8]) {
          constexpr int b = decltype(b_tag)::value;
#pragma unroll
          for (int a = 0; a < 8; ++a) {
            const uint2 yb = src[a];
            const float y0 = __uint_as_float(yb.x << 16), y1 = __uint_as_float(yb.x & 0xffff0000u);
            const float y2 = __uint_as_float(yb.y << 16), y3 = __uint_as_float(yb.y & 0xffff0000u);
            if (EPI == EPI_DGELU) {  // out = acc * gelu'(y)
              acc[a][b][0] *= dgelu_f(y0); acc[a][b][1] *= dgelu_f(y1); acc[a][b][2] *= dgelu_f(y2); acc[a][b][3] *= dgelu_f(y3);
              __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise interleaves all the evaluations and spills)
            } else {                 // out = acc + aux
              acc[a][b][0] += y0; acc[a][b][1] += y1; acc[a][b][2] += y2; acc[a][b][3] += y3;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        };
        typedef std::integral_constant<int, 0> B0; typedef std::integral_constant<int, 1> B1;
        typedef std::integral_constant<int, 2> B2; typedef std::integral_constant<int, 3> B3;
        fetch(0, yA); fetch(1, yB);
        fold(B0{}, yA);
        fetch(2, yA);
        fold(B1{}, yB);
        fetch(3, yB);
        deferred_dmas();
        fold(B2{}, yA);
        fold(B3{}, yB);
      } else {
        deferred_dmas();
      }
      const unsigned ep_lds = (unsigned)(size_t)((lds_u8_t *)smem) + 65536u + (unsigned)(wave * 2048);
      unsigned jrow[4][2];   // element offset of output row jw + b * 16 + t2 * 8 + lr
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) jrow[b][t2] = mapped_row(jw + b * 16 + t2 * 8 + lr, ldo, orpb, obs);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
          for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
              const int a = half * 4 + a4;
              float vv[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) vv[r] = (float)(__bf16)acc[a][b][r];  // what is stored (and what a backward differentiates at)
              if (EPI == EPI_BIAS_GELU && pass == 1) {  // x * Phi(x), Phi of the bf16 value
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[r] = gelu_f(vv[r]);
              }
              // (inline asm: a compiler-visible LDS access here would be fenced with vmcnt(0) against the LDS-DMAs in flight)
              const u32x2_t pk = {pack_bf16x2(vv[0], vv[1]), pack_bf16x2(vv[2], vv[3])};
              asm volatile("ds_write_b64 %0, %1" ::"v"(ep_lds + (unsigned)(wj + (((a4 * 2 + wc2) ^ wx) << 4))), "v"(pk) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            u32x4_t d[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
              const int jr = t2 * 8 + lr;                  // image row; 16-B chunk cp
              asm volatile("ds_read_b128 %0, %1" : "=v"(d[t2]) : "v"(ep_lds + (unsigned)(jr * 128 + ((cp ^ (jr & 7)) << 4))) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1])::"memory");   // (the image is rewritten next)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
              const int i_st = iw + half * 64 + cp * 8;
              const unsigned off = i_st < Ni ? (jrow[b][t2] + (unsigned)i_st) * 2u : DEAD;   // Ni % 8 == 0 (host check)
              __builtin_amdgcn_raw_buffer_store_b128(d[t2], pass == 1 ? rsO2 : rsO, off, 0, ST_AUX);
            }
          }
        }
      }
    }

    if (SK) {
      su += cur.nkt;
      if (su >= sue) break;
      cur = nxt;
      has_next = su + cur.nkt < sue;
      if (has_next) nxt = seg(su + cur.nkt, sue);
    } else {
      v += G;
      if (v >= nwg) break;
      cur = nxt;
      has_next = v + G < nwg;
      if (has_next) nxt = decode(v + G);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the parked cursors' out-of-range DMAs have written their zeros
}

// ---- stream-K workspaces: one per stream (launches of one stream are serialised; a captured launch keeps the pointer, so a
// graph must replay on the stream it was captured on, which is how pipeline.py and graphed.py replay) -------------------------
struct SkWorkspace { int device; hipStream_t stream; void *ptr; long bytes; };
static SkWorkspace g_sk_ws[16];
static int g_sk_n = 0;

void *gemm_sk_workspace(hipStream_t st, long *bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  for (int i = 0; i < g_sk_n; ++i)
    if (g_sk_ws[i].device == dev && g_sk_ws[i].stream == st) {
      *bytes = g_sk_ws[i].bytes;
      return g_sk_ws[i].ptr;
    }
  return nullptr;
}

// which stream-K forms may run: bit 0 = this file's (256 x 128 tiles: measured slower than whole tiles),
// bit 1 = gemm256_kernel's (csrc/gemm.hip: slower as well, 98.6 us against 76.7 on fc2).  Default 0; bq_gemm_streamk_mode sets
// it; BQ_GEMM_STREAMK=0 in the environment keeps every launch on whole tiles whatever the mode.
static int g_sk_mode = 0;
int gemm_sk_mode() {
  static int env_off = -1;
  if (env_off < 0) {
    const char *e = getenv("BQ_GEMM_STREAMK");
    env_off = (e != nullptr && e[0] == '0') ? 1 : 0;
  }
  return env_off ? 0 : g_sk_mode;
}
static bool sk_enabled() { return (gemm_sk_mode() & 1) != 0; }

int launch_gemm_mid(const GemmArgs &ga_in, bool p_xc, bool q_xc, bool out_f32, int epi, bool background, hipStream_t stream) {
  const int slots = 2 * device_cus();   // two co-resident workgroups per CU (64 KB of LDS, <= 256 VGPRs each)
  for (int k = 0; k < ga_in.n; ++k)
    if (ga_in.p[k].Kc < 128) return -1;   // (the QB0 cursor runs two K tiles ahead: a tile has at least two)
  GemmArgs ga = ga_in;
  ga.sk_ws = nullptr;
  // stream-K (header): one problem, a K-contiguous P (forward, or dX on the transposed weight copy), bf16 out, a long
  // contraction cut across a grid that whole tiles would fill unevenly -- the N = 768 / K = 3072 launches of the ViT MLP
  if (sk_enabled() && !background && ga.n == 1 && !p_xc && !q_xc && !out_f32 && (epi == EPI_NONE || epi == EPI_BIAS || epi == EPI_ADD)) {
    const int nkt = (ga.p[0].Kc + 63) >> 6, tiles = ga.total_tiles;
    const long units = (long)tiles * nkt;
    const int rounds = (tiles + slots - 1) / slots;
    long ws_bytes = 0;
    void *ws = gemm_sk_workspace(stream, &ws_bytes);
    // worth it when whole tiles leave >= 15 % of the makespan idle and a share is still >= 16 K tiles (the hand-off moves
    // 128 KB per cut: ~1/3 of a 48-K-tile share's gain at 36 K tiles, more than the gain of a 12-K-tile contraction)
    if (ws != nullptr && nkt >= 24 && tiles >= slots / 2 && units / slots >= 16 && (long)rounds * nkt * slots >= units * 115 / 100
        && 8L * tiles <= SK_TICKET_BYTES && ws_bytes >= SK_TICKET_BYTES + 2L * slots * SK_SLAB_BYTES) {
      ga.sk_ws = ws;
      const dim3 grid(slots), block(256);
      if (epi == EPI_NONE) hipLaunchKernelGGL((gemm128_kernel<false, false, EPI_NONE, false, 16, true>), grid, block, 0, stream, ga);
      else if (epi == EPI_BIAS) hipLaunchKernelGGL((gemm128_kernel<false, false, EPI_BIAS, false, 16, true>), grid, block, 0, stream, ga);
      else hipLaunchKernelGGL((gemm128_kernel<false, false, EPI_ADD, false, 16, true>), grid, block, 0, stream, ga);
      return 0;
    }
  }
  // background: one workgroup per CU (BQ_GEMM_BACKGROUND, include/bqhip_fusion.h)
  const int use = background ? (slots + 1) / 2 : slots;
  const dim3 grid(ga.total_tiles < use ? ga.total_tiles : use), block(256);
  // output stores write-through (sc1): the lines are dropped from the XCD's L2, which then keeps the weight / activation
  // panels the co-scheduled tiles share.  MEASURED against the default policy, every c3 shape: 3-9 % faster (qkv 60.5 ->
  // 59.7 us, proj 26.6 -> 24.2, fc1 + GELU 104.5 -> 99.9, dX qkv 60.4 -> 57.3)
#define BQ_MID_LAUNCH(PX, E)                                                                                        \
  do {                                                                                                              \
    hipLaunchKernelGGL((gemm128_kernel<PX, false, E, false, 16>), grid, block, 0, st_, ga);                         \
    return 0;                                                                                                       \
  } while (0)
  hipStream_t st_ = stream;
  if (q_xc || out_f32) {   // the weight-gradient form (fp32 out, column sums on request)
    if (!(p_xc && q_xc && out_f32 && epi == EPI_NONE)) return -1;
    hipLaunchKernelGGL((gemm128_kernel<true, true, EPI_NONE, true, 16>), grid, block, 0, st_, ga);
    return 0;
  }
  for (int k = 0; k < ga.n; ++k)
    if (ga.p[k].colsum != nullptr) return -1;
  if (!p_xc) {
    if (epi == EPI_NONE) BQ_MID_LAUNCH(false, EPI_NONE);
    if (epi == EPI_BIAS) BQ_MID_LAUNCH(false, EPI_BIAS);
    if (epi == EPI_BIAS_GELU) BQ_MID_LAUNCH(false, EPI_BIAS_GELU);
    if (epi == EPI_DGELU) BQ_MID_LAUNCH(false, EPI_DGELU);
    if (epi == EPI_ADD) BQ_MID_LAUNCH(false, EPI_ADD);
  } else {
    if (epi == EPI_NONE) BQ_MID_LAUNCH(true, EPI_NONE);
    if (epi == EPI_DGELU) BQ_MID_LAUNCH(true, EPI_DGELU);
    if (epi == EPI_ADD) BQ_MID_LAUNCH(true, EPI_ADD);
  }
  return -1;
}

}  // namespace bq

extern "C" int bq_gemm_streamk_mode(int mode) {
  const int old = bq::g_sk_mode;
  if (mode >= 0) bq::g_sk_mode = mode & 3;
  return old;
}

extern "C" long bq_gemm_workspace_bytes(void) {
  return bq::SK_TICKET_BYTES + 2L * 2L * bq::device_cus() * bq::SK_SLAB_BYTES;
}

extern "C" int bq_gemm_set_workspace(void *ws, long bytes, void *stream) {
  using namespace bq;
  int dev = 0;
  BQ_REQUIRE(hipGetDevice(&dev) == hipSuccess, BQ_EINVAL, "bq_gemm_set_workspace: no device");
  BQ_REQUIRE(ws == nullptr || bytes >= bq_gemm_workspace_bytes(), BQ_EINVAL, "bq_gemm_set_workspace: %ld bytes < %ld", bytes,
             bq_gemm_workspace_bytes());
  for (int i = 0; i < g_sk_n; ++i)
    if (g_sk_ws[i].device == dev && g_sk_ws[i].stream == (hipStream_t)stream) {
      g_sk_ws[i].ptr = ws;
      g_sk_ws[i].bytes = bytes;
      return 0;
    }
  BQ_REQUIRE(g_sk_n < 16, BQ_ELIMIT, "bq_gemm_set_workspace: more than 16 (device, stream) pairs");
  g_sk_ws[g_sk_n++] = SkWorkspace{dev, (hipStream_t)stream, ws, bytes};
  return 0;
}
