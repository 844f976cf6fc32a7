// MFMA bf16 GEMM family for gfx950 (MI355X): the dense contractions of the fusion half of the hot path --
// ViT patch-embed / QKV / proj / FFN (reference models/vit.py:30-32,51-53,144-145) and the MED/BERT Q/K/V,
// output and FFN projections (models/med.py:112-118,232,295,310) -- forward, input-gradient and weight-gradient
// forms, bf16 operands, fp32 accumulation on v_mfma_f32_16x16x32_bf16.
//
// One formulation for all three:   out[j][i] (+)= sum_kc P(i, kc) * Q(j, kc)      (i contiguous in `out`)
//   forward  y[m][n]  = x W^T + b : P = W  (K-contiguous, i = n),  Q = x  (K-contiguous, j = m)
//   dX       dx[m][k] = dy W      : P = W  (contraction-major: P(i=k, kc=n) = W[n][k]),  Q = dy (K-contiguous)
//   dW       dw[n][k] = dy^T x    : P = x  (contraction-major, i = k),  Q = dy (contraction-major, j = n), fp32 out
// "K-contiguous" (KC) operands are staged as [rows][64 k] LDS images and read with ds_read_b128; contraction-major
// (XC) operands as [64 kc][64 outs] images read TRANSPOSED with ds_read_b64_tr_b16 -- no transposed copy of a weight
// or an activation exists anywhere.  P is the MFMA A operand (i on the accumulator registers: 4 consecutive i per
// lane => 8-byte bf16 / 16-byte fp32 contiguous pieces of an output row), Q the B operand (j on the lanes).
//
// gemm256_kernel: 256 x 256 output tile per workgroup, 8 waves (2 along i x 4 along j, 128 x 64 per wave), K step 64,
// LDS-DMA staging (buffer_load ... lds: bounds-checked, no VGPRs), two 64 KB LDS buffers.  The K loop is a
// phased software pipeline in which the two wave groups (wr = 0 / 1, one wave of each per SIMD) run HALF A
// PHASE APART: while one group issues its 16-MFMA cluster the other issues its LDS reads and the next DMA, so the
// matrix pipe of every SIMD always has a wave feeding it (cdna_hip_programming.md §5 "8-phase template"; the schedule
// below is this file's own and its hazard analysis is in DESIGN.md §4.4):
//   staging unit = 64 rows x 64 k (8 KB, one DMA per wave): A0(g) A1(g) = the two 64-row halves of wave group g's
//   P rows, B0(h) B1(h) = the first / second 32 columns of the four wave columns (h = wc >> 1);
//   rounds 1-5: four phases of 16 MFMAs per K tile; round 6: TWO of 32 (the schedule and its wait arithmetic stand at the
//   K loop), the fragment reads as inline asm (gemm_common.h: the compiler put s_waitcnt vmcnt(0) in front of its own).
#include "gemm_common.h"

namespace bq {

typedef unsigned u32x4_sk __attribute__((ext_vector_type(4)));

// SK (round 5, second attempt at VERDICT r4 item 1): the STREAM-K form of this kernel for ONE problem with a long contraction
// whose 256 x 256 tiles do not fill the chip evenly (fc2 forward: 195 tiles of 48 K tiles on 256 CUs).  The grid is one
// workgroup per CU; workgroup s walks the K-TILE UNITS [b(s), b(s + 1)) of the tile-major unit sequence -- a tail of one tile,
// whole tiles, a head of the next -- running this kernel's ordinary pipeline over each segment; a tile cut across several
// workgroups is finished by the one that ARRIVES LAST (csrc/gemm_mid.hip's protocol: ticket, fp32 slabs parked with
// write-through stores, a bounded wait, a fold in fixed slot order).  The first attempt put stream-K on the 256 x 128 kernel
// and lost: those launches are bound by L2 -> LDS bytes, which a 256 x 256 tile halves -- and the vendor library's kernel for
// these shapes is exactly a 256 x 256 x 64 stream-K kernel (tools/hipblaslt_names.py).  MEASURED: parity-green and slower again
// -- fc2 forward 98.6 us against 76.7 on whole 256 x 256 tiles (75.4 on 256 x 128), the c3 step + 0.9 ms: with one workgroup per
// CU nothing hides a segment's second pipeline fill, the 256 KB park and the 256 KB fold.  Off by default
// (bq_gemm_streamk_mode bit 1).
//
// Round 6, tried on the weight-gradient form and withdrawn: PREFETCH ROLES -- the four waves of group 0 issue all the LDS-DMAs
// (and do the counted waits), the four of group 1 issue none and touch, one dword per 128-B line, the operand rows four K
// tiles ahead, so that the DMAs hit L2 (a prefetch by the wave that also issues DMAs is useless: vmcnt retires in issue
// order).  Bit-identical, 20 % SLOWER on every cut of the 48 problems (2.49 -> 3.02 ms; profiles/r06_dw_prefetch.txt): the
// launches are not waiting for HBM, they are at the LDS port (192 KB of fragment reads + 64 KB of DMA writes per K tile =
// 2048 clocks at 128 B/clk, the MFMA time of the same K tile).
template <bool P_XC, bool Q_XC, int EPI, bool OUT_F32, bool SK = false>
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmArgs args) {
  static_assert(!SK || (!P_XC && !Q_XC && !OUT_F32 && (EPI == EPI_NONE || EPI == EPI_BIAS)), "stream-K: K-contiguous operands, bf16 out");
  __shared__ __attribute__((aligned(16))) unsigned char smem[131072];
  __shared__ unsigned s_skflag;
  constexpr bool GTAB = !OUT_F32 && (EPI == EPI_BIAS_GELU || EPI == EPI_DGELU);
  __shared__ float s_gtab[GTAB ? GELU_TAB_N : 1];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- workgroup -> (problem, tile): XCD-aware (blocks b, b+8, ... share an XCD's L2: give each XCD a contiguous run
  // of tiles so that consecutive tiles, which share the Q row panel, hit the same L2) -----------------------------------
  const int nwg = args.total_tiles;
  auto xcd_order = [](int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  };
  // ---- stream-K: unit ranges (one problem; every tile has nkt_full K tiles; a segment is never a single K tile) ---------------
  const int G = (int)gridDim.x;
  const int nkt_full = SK ? (args.p[0].Kc + 63) >> 6 : 1;
  const int U = SK ? nwg * nkt_full : 0;
  auto bound = [&](int x) {
    int b = (int)((long)x * U / G);
    const int r = b % nkt_full;
    return r == 1 ? b - 1 : (r == nkt_full - 1 ? b + 1 : b);
  };
  const int slot = SK ? xcd_order((int)blockIdx.x, G) : 0;
  int su = SK ? bound(slot) : 0;
  const int sue = SK ? bound(slot + 1) : 0;
  if (SK && su >= sue) return;
  for (;;) {   // SK: one trip per segment of this workgroup's unit range; otherwise exactly one trip
  int t, kb = 0, kseg = 0;
  if (SK) {
    t = su / nkt_full;
    kb = su - t * nkt_full;
    kseg = min(nkt_full - kb, sue - su);
  } else {
    t = xcd_order((int)blockIdx.x, nwg);
  }
  int pi = 0;
  for (int k = 1; k < args.n; ++k)
    if (t >= args.p[k].tile0) pi = k;
  const GemmProblem &pr = args.p[pi];
  const int tl = t - pr.tile0;
  const int tiles_j = (pr.Nj + 255) >> 8;
  // ragged last j block first: it is the cheapest tile, and late-starting workgroups should be the full ones
  const int bj = tiles_j - 1 - tl / pr.tiles_i(), bi = tl % pr.tiles_i();
  const int i0 = bi * 256, j0 = bj * 256;
  const int Ni = pr.Ni, Nj = pr.Nj, Kc = pr.Kc;
  const int ldp = pr.ldp, ldq = pr.ldq;
  const int nkt = SK ? kseg : (Kc + 63) >> 6;

  // ---- staging: 8 units per K tile, one LDS-DMA per wave per unit (wave w -> unit rows 8w..8w+7) -----------------
  const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void *)pr.P, 0, pr.p_bytes, 0x00020000);
  const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void *)pr.Q, 0, pr.q_bytes, 0x00020000);
  const int ur = wave * 8 + (lane >> 3);  // unit row this lane stages
  const int cp = lane & 7;                // LDS chunk position
  unsigned voff[8];                       // A0(0) A0(1) A1(0) A1(1) B0(0) B0(1) B1(0) B1(1)
#pragma unroll
  for (int u = 0; u < 4; ++u) {           // P units: rows i0 + g*128 + half*64 + r
    const int g = u & 1, half = u >> 1;
    if (!P_XC) {
      const int c = cp ^ (ur & 7);
      voff[u] = (unsigned)(((i0 + g * 128 + half * 64 + ur) * ldp + c * 8) * 2);
    } else {
      const int c = cp ^ (xg(ur) << 1);
      voff[u] = (unsigned)((ur * ldp + i0 + g * 128 + half * 64 + c * 8) * 2);
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {           // Q units: j0 + h*128 + (r>>5)*64 + half*32 + (r&31)
    const int h = u & 1, half = u >> 1;
    if (!Q_XC) {
      const int c = cp ^ (ur & 7);
      voff[4 + u] = (unsigned)(((j0 + h * 128 + (ur >> 5) * 64 + half * 32 + (ur & 31)) * ldq + c * 8) * 2);
    } else {
      const int c = cp ^ (xg(ur) << 1);
      voff[4 + u] = (unsigned)((ur * ldq + j0 + h * 128 + (c >> 2) * 64 + half * 32 + (c & 3) * 8) * 2);
    }
  }
  const unsigned p_step = P_XC ? (unsigned)(64 * ldp * 2) : 128u;
  const unsigned q_step = Q_XC ? (unsigned)(64 * ldq * 2) : 128u;
  if (SK) {   // the segment starts at K tile kb of its tile
#pragma unroll
    for (int u = 0; u < 4; ++u) { voff[u] += (unsigned)kb * p_step; voff[4 + u] += (unsigned)kb * q_step; }
  }
  const unsigned lds_w = (unsigned)(wave * 1024);
  // batched-row map on the CONTRACTION rows of a contraction-major Q (the weight-gradient form reading its row range of
  // a (B, L1 + L2, N) gradient in place): the lane's row walks 64 rows per K tile; when it leaves its batch (q_rpb >= 64:
  // at most once per step) the cursor jumps over the rows between the batches.  One row counter per unit pair (B0 / B1
  // are staged in different phases).
  const int q_rpb = Q_XC ? (int)pr.q_rpb() : 0;
  const unsigned q_gap = (unsigned)((pr.q_bstride - q_rpb * ldq) * 2);
  int q_rl[2] = {ur, ur};

  // stage the unit pair (u, u+1) of K tile `kt` into buffer kt & 1; past the last tile: out-of-range DMAs (no
  // traffic, they keep the vmcnt bookkeeping uniform; the buffer they zero is never read again)
  auto stage_pair = [&](int u, int kt) {
    const bool live = kt < nkt;
    const unsigned base = (unsigned)((kt & 1) * 65536 + u * 8192) + lds_w;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const unsigned vo = live ? voff[u + d] : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(u < 4 ? rsP : rsQ, (lds_void_t *)(smem + base + d * 8192), 16, vo, 0, 0, 0);
      voff[u + d] += (u < 4) ? p_step : q_step;
    }
    if (Q_XC && u >= 4 && q_rpb != 0) {   // (workgroup-uniform branch)
      int &rl = q_rl[(u - 4) >> 1];
      rl += 64;
      const bool wrap = rl >= q_rpb;
      rl -= wrap ? q_rpb : 0;
      voff[u] += wrap ? q_gap : 0u;
      voff[u + 1] += wrap ? q_gap : 0u;
    }
  };

  // ---- fragment read addresses (lane-dependent parts) ------------------------------------------------------------
  const int row16 = lane & 15, q4 = lane >> 4;
  const int kc_base = row16 * 128 + ((q4 ^ (row16 & 7)) << 4);
  int xc_base[4];
  {
    const int q = (lane & 15) >> 2, p = lane & 3, g = (q >> 1) | ((q4 & 1) << 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) xc_base[s] = (8 * q4 + q) * 128 + ((s ^ g) << 5) + 8 * p;
  }
  const int uA0 = wr * 8192, uA1 = (2 + wr) * 8192, uB0 = (4 + (wc >> 1)) * 8192, uB1 = (6 + (wc >> 1)) * 8192;
  const int bsub = (wc & 1) * 2;

  // which quadrants of this wave's 128 (i) x 64 (j) tile hold any valid output (ragged edge tiles skip the rest)
  const int iw = i0 + wr * 128, jw = j0 + wc * 64;
  const bool vA0 = iw < Ni, vA1 = iw + 64 < Ni, vB0 = jw < Nj, vB1 = jw + 32 < Nj;

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // weight-gradient form (P and Q contraction-major, fp32 out) with pr.colsum set: the column sums of Q over the
  // contraction -- the layer's BIAS gradient, sum over rows of dY -- from four more MFMAs per K tile with an all-ones A
  // operand on the B fragments already in registers, in the waves wr == 0 of the i = 0 tiles only (every j is then summed
  // by exactly one wave: plain stores, no atomics).  The grouped column-sum launch re-read every dY (2.7 GB at c3) for it.
  constexpr bool QSUM = P_XC && Q_XC && OUT_F32;
  const bool do_qsum = QSUM && pr.colsum != nullptr && bi == 0 && wr == 0;   // wave-uniform
  f32x4 qs[4];
  bf16x8 ones;
#pragma unroll
  for (int b = 0; b < 4; ++b) qs[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#define BQ_MFMA_QSUM(FB, BO)                                                                          \
  if (QSUM && do_qsum) {                                                                              \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int b = 0; b < 2; ++b)    \
        qs[BO + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, FB[b][kk], qs[BO + b], 0, 0, 0);   \
  }

  // ---- prologue: tile 0 whole, then A0 B0 B1 of tile 1 (its A1 is staged by X(0)) ---------------------------------------
  stage_pair(0, 0); stage_pair(4, 0); stage_pair(6, 0); stage_pair(2, 0); stage_pair(0, 1); stage_pair(4, 1); stage_pair(6, 1);
  if (GTAB) gelu_tab_fill<EPI == EPI_DGELU>(s_gtab, tid, 512);  // under the prologue's DMA latency; read after the K loop
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // A0 B0 B1 of tile 0 have landed (behind them: A1(0) and tile 1's six)
  BQ_BARRIER();
  if (wr == 1) BQ_BARRIER();  // group 1 runs half a phase behind group 0

  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
  const unsigned smem_lds = lds_addr_of(smem);
#define BQ_MFMA_Q(AO, FB, BO)                                                                         \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int a = 0; a < 4; ++a)      \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) acc[AO + a][BO + b] =                             \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][kk], FB[b][kk], acc[AO + a][BO + b], 0, 0, 0);
  // (the LDS reads are waited for BEFORE the barrier: behind it no read of any wave is in flight, so the phase that follows
  // may restage what this one read)
#define BQ_PHASE_SYNC_A()                                   \
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
  BQ_BARRIER();                                             \
  __builtin_amdgcn_sched_barrier(0);                        \
  __builtin_amdgcn_s_setprio(1);
#define BQ_PHASE_SYNC_B()                                   \
  __builtin_amdgcn_s_setprio(0);                            \
  BQ_BARRIER();

  // Round 6: TWO phases per K tile, 32 MFMAs (512 cycles) each, instead of four of 16.  The other group's reads, DMA issue
  // and barrier skew of a phase (~300-400 cycles) used to stand against a 256-cycle MFMA cluster -- the matrix pipe idled
  // about half of every phase; against 512 cycles they fit.
  //   X(t) reads A0 B0 B1 of tile t -> Q00 Q01 ; stages A1(t+1)        (its image was last read in Y(t-1))
  //   Y(t) reads A1                 -> Q11 Q10 ; stages A0 B0 B1 (t+2)  (their images were read in X(t))
  // A wave's DMAs in issue order: ... a(t) = A1(t+1) [2], b(t) = A0 B0 B1 (t+2) [6], a(t+1) [2], b(t+1) [6] ...; the wait of
  // X(t+1) wants a(t) (read next, in Y(t+1)): b(t) + a(t+1) = 8 behind it; the wait of Y(t+1) wants b(t) (read in X(t+2)):
  // a(t+1) + b(t+1) = 8 behind it -- vmcnt(8) both times, a stage is waited for one K tile after it was issued.
  for (int kt = 0; kt < nkt; ++kt) {
    const unsigned buf = smem_lds + (unsigned)((kt & 1) * 65536);   // (asm reads: gemm_common.h, read_frag_asm)
    // ---- X: A0, B0, B1 -> Q00, Q01 ; stage A1(t+1)
    if (vA0) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[a][kk] = read_frag_asm<P_XC>(buf + uA0, a, kk, kc_base, xc_base);
    }
    if (vB0) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb0[b][kk] = read_frag_asm<Q_XC>(buf + uB0, bsub + b, kk, kc_base, xc_base);
    }
    if (vB1) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb1[b][kk] = read_frag_asm<Q_XC>(buf + uB1, bsub + b, kk, kc_base, xc_base);
    }
    stage_pair(2, kt + 1);
    BQ_PHASE_SYNC_A();
    if (vA0 && vB0) { BQ_MFMA_Q(0, fb0, 0) }
    if (vB0) { BQ_MFMA_QSUM(fb0, 0) }
    if (vA0 && vB1) { BQ_MFMA_Q(0, fb1, 2) }
    if (vB1) { BQ_MFMA_QSUM(fb1, 2) }
    BQ_PHASE_SYNC_B();
    // ---- Y: A1 (B0, B1 kept in registers) -> Q11, Q10 ; stage A0, B0, B1 (t+2)
    if (vA1) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[a][kk] = read_frag_asm<P_XC>(buf + uA1, a, kk, kc_base, xc_base);
    }
    stage_pair(0, kt + 2); stage_pair(4, kt + 2); stage_pair(6, kt + 2);
    BQ_PHASE_SYNC_A();
    if (vA1 && vB1) { BQ_MFMA_Q(4, fb1, 2) }
    if (vA1 && vB0) { BQ_MFMA_Q(4, fb0, 0) }
    BQ_PHASE_SYNC_B();
  }
  if (wr == 0) BQ_BARRIER();  // re-align the two groups: every LDS read of the K loop is complete after this
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing out-of-range DMAs have written their zeros
  BQ_BARRIER();

  if (SK && kseg != nkt_full) {
    // ---- a tile cut across several workgroups is finished by the one that arrives last (csrc/gemm_mid.hip, header) --------
    unsigned *tick = (unsigned *)args.sk_ws + 2 * t;
    const int t0 = t * nkt_full, t1 = t0 + nkt_full;
    int s_lo = slot, s_hi = slot;
    while (s_lo > 0 && bound(s_lo) > t0) --s_lo;
    while (s_hi + 1 < G && bound(s_hi + 1) < t1) ++s_hi;
    const int n_part = s_hi - s_lo + 1;
    if (tid == 0) s_skflag = __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)s_skflag);
    constexpr unsigned SLAB256 = 256u * 256u * 4u;
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((char *)args.sk_ws + SK_TICKET_BYTES, 0, 0x7fffffffu, 0x00020000);
    // slab of (slot, which of its two possible partial segments): the first one iff the slot's range starts inside the tile
    auto slab_off = [&](int sl) { return (unsigned)(2 * sl + (bound(sl) >= t0 ? 0 : 1)) * SLAB256; };
    if (arrived != (unsigned)(n_part - 1)) {
      // not last: park the accumulators (write-through), drain, count this workgroup in
      const unsigned base = slab_off(slot) + (unsigned)(tid * 16);
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_sk, acc[a][b]), rsW, base + (unsigned)((a * 4 + b) * 8192), 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(tick + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      su += kseg;
      if (su >= sue) return;
      continue;   // (the tile's last arriver stores it)
    }
    // last: every other participant is past its K loop; wait for their slabs (bounded: a timeout can only mean a protocol bug,
    // and a wrong tile fails a test where a hang would lose the GPU), take the tickets back, fold in slot order
    if (tid == 0) {
      for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(tick + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(n_part - 1); ++spin)
        __builtin_amdgcn_s_sleep(4);
      __hip_atomic_store(tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(tick + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int sl = s_lo; sl <= s_hi; ++sl) {
      if (sl == slot) continue;
      const unsigned base = slab_off(sl) + (unsigned)(tid * 16);
#pragma unroll
      for (int a0 = 0; a0 < 8; a0 += 2) {   // eight 16-byte loads in flight per lane (the fragment registers are free here)
        u32x4_sk part[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) part[e] = __builtin_amdgcn_raw_buffer_load_b128(rsW, base + (unsigned)((a0 * 4 + e) * 8192), 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[a0 + (e >> 2)][e & 3] += __builtin_bit_cast(f32x4, part[e]);
      }
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------------------------
  // accumulator (a, b)[r]: i = iw + a*16 + q4*4 + r, j = jw + b*16 + row16
  const int ldo = pr.ldo;
  if (OUT_F32) {
    float *out = reinterpret_cast<float *>(pr.out);
    if (QSUM && do_qsum && q4 == 0) {  // every row of the all-ones product holds the sums: lane row16 has j's in element 0
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = jw + b * 16 + row16;
        if (j < Nj) pr.colsum[j] = qs[b][0];
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = jw + b * 16 + row16;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int i = iw + a * 16 + q4 * 4;
        if (j < Nj && i < Ni) {  // Ni % 4 == 0 (checked on the host): a lane's four i are valid together
          float4 *dst = reinterpret_cast<float4 *>(out + (long)j * ldo + i);
          const f32x4 v = acc[a][b];
          *dst = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    return;
  }
  if (EPI == EPI_BIAS_CE) {
    // Cross-entropy partials of the LM head (csrc/lmhead.hip combines them): for every row j of this wave and the 128
    // vocabulary entries i of its half tile -- the fp32 logits acc + bias, before any rounding -- the running maximum,
    // sum exp(z - max) and sum z over the VALID entries (i < n_valid = pr.ksplit()), and the target's logit where this lane
    // holds it.  Field reuse for this epilogue: out2 = partial f32 [tiles_i * 2][Nj][3], aux = int32 targets [Nj]
    // (-100 = ignore), colsum = f32 [Nj] target logits, ksplit = number of valid vocabulary entries.
    const int n_valid = pr.ksplit();
    float *part = reinterpret_cast<float *>(pr.out2) + (long)(bi * 2 + wr) * Nj * 3;
    const int *tgt = reinterpret_cast<const int *>(pr.aux);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = jw + b * 16 + row16;
      const int t = j < Nj ? tgt[j] : -100;
      float m = -INFINITY, sz = 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int i = iw + a * 16 + q4 * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (pr.bias != nullptr && i < Ni) load_bias4(pr, i, bv);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float z = acc[a][b][r] + bv[r];
          if (i + r < n_valid) {
            m = fmaxf(m, z);
            sz += z;
            if (i + r == t) pr.colsum[j] = z;   // exactly one lane of the grid holds the target of row j
          }
        }
      }
      float se = 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int i = iw + a * 16 + q4 * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (pr.bias != nullptr && i < Ni) load_bias4(pr, i, bv);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (i + r < n_valid) se += __builtin_amdgcn_exp2f((acc[a][b][r] + bv[r] - m) * 1.4426950408889634f);
      }
      // the four lanes row16 + 16 q hold the other vocabulary entries of the same row
#pragma unroll
      for (int sh = 16; sh <= 32; sh <<= 1) {
        const float mo = __shfl_xor(m, sh), so = __shfl_xor(se, sh), zo = __shfl_xor(sz, sh);
        const float mn = fmaxf(m, mo);
        const float fa = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m - mn) * 1.4426950408889634f);
        const float fb = mo == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((mo - mn) * 1.4426950408889634f);
        se = se * fa + so * fb;
        m = mn;
        sz += zo;
      }
      if (q4 == 0 && j < Nj) {
        part[(long)j * 3 + 0] = m;
        part[(long)j * 3 + 1] = se;
        part[(long)j * 3 + 2] = sz;
      }
    }
  }
  // bf16 outputs go through LDS so that global stores are whole 256-B row pieces: wave-private [64 j][128 i] image
  // (16 KB, 16-B chunk c of row j at j*256 + ((c ^ (j & 15)) << 4): conflict-free both ways), written as 8-B pieces,
  // read back as 16-B pieces.
  unsigned char *ep = smem + wave * 16384;
  __bf16 *outb = reinterpret_cast<__bf16 *>(pr.out);
  const int npass = (EPI == EPI_BIAS_GELU) ? 2 : 1;
  const bool want_colsum = EPI != EPI_BIAS_CE && pr.colsum != nullptr;  // (the CE epilogue reuses the field)
#pragma unroll 1
  for (int pass = 0; pass < npass; ++pass) {
    if (EPI == EPI_DGELU || EPI == EPI_ADD) {
      // a second operand `aux` in the accumulator's own map (8 B per lane).  j block outermost: the eight pieces of one j
      // block are the 16 rows' whole 256-B spans, requested back to back and one block ahead of their use (with i
      // outermost the four 32-B pieces of a line were requested an epilogue apart, each load waited for on the spot)
      float cs[8][4];
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[a][r] = 0.f;
      // bounds-checked buffer loads (rows past Nj read zeros; columns past Ni read the next row -- never stored), the
      // next j block's eight pieces in flight while this one is processed
      const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void *)pr.aux, 0, (int)((long)Nj * ldo * 2), 0x00020000);
      uint2 yy[2][8];  // (the column-sum variant, tests only, fetches one block at a time: 32 more live registers)
      auto fetch = [&](int b, uint2(&dst)[8]) {
        const unsigned off = (unsigned)(((jw + b * 16 + row16) * ldo + iw + q4 * 4) * 2);
#pragma unroll
        for (int a = 0; a < 8; ++a)
          dst[a] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsX, off + a * 32, 0, 0));
      };
      auto sweep = [&](auto wc_tag) {  // (two copies: the column-sum variant keeps its values apart)
        constexpr bool WC = decltype(wc_tag)::value;
        if (!WC) fetch(0, yy[0]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = jw + b * 16 + row16;
        if (WC) fetch(b, yy[0]);
        else if (b + 1 < 4) fetch(b + 1, yy[(b + 1) & 1]);
        float gd[8][4];  // gelu'(y): all 32 table gathers of the block in flight before the first use
        if (EPI == EPI_DGELU) {
#pragma unroll
          for (int a = 0; a < 8; ++a) {
            const uint2 yb = yy[WC ? 0 : (b & 1)][a];
            gd[a][0] = gelu_tab_at(s_gtab, yb.x);
            gd[a][1] = gelu_tab_at(s_gtab, yb.x >> 16);
            gd[a][2] = gelu_tab_at(s_gtab, yb.y);
            gd[a][3] = gelu_tab_at(s_gtab, yb.y >> 16);
          }
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const uint2 yb = yy[WC ? 0 : (b & 1)][a];
          float v[4];
          if (EPI == EPI_DGELU) {  // out = acc * gelu'(y)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r] * gd[a][r];
          } else {                 // out = acc + aux: a second gradient of the same tensor rides on the dX GEMM
            v[0] = acc[a][b][0] + __uint_as_float(yb.x << 16);
            v[1] = acc[a][b][1] + __uint_as_float(yb.x & 0xffff0000u);
            v[2] = acc[a][b][2] + __uint_as_float(yb.y << 16);
            v[3] = acc[a][b][3] + __uint_as_float(yb.y & 0xffff0000u);
          }
          if (WC) {
            const bool ok = j < Nj && iw + a * 16 + q4 * 4 < Ni;
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[a][r] += ok ? (float)(__bf16)v[r] : 0.f;
          }
          uint2 pk;
          pk.x = pack_bf16x2(v[0], v[1]);
          pk.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2 *>(ep + (b * 16 + row16) * 256 + (((a * 2 + (q4 >> 1)) ^ row16) << 4) + (q4 & 1) * 8) = pk;
        }
      }
      };
      if (want_colsum) sweep(std::true_type{}); else sweep(std::false_type{});
      if (want_colsum) {
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float sacc = cs[a][r];
            sacc += dpp_f32_add<0x111>(sacc);
            sacc += dpp_f32_add<0x112>(sacc);
            sacc += dpp_f32_add<0x114>(sacc);
            sacc += dpp_f32_add<0x118>(sacc);
            const int i = iw + a * 16 + q4 * 4 + r;
            if (row16 == 15 && i < Ni) atomicAdd(pr.colsum + i, sacc);
          }
      }
    } else {
      // straight-line copies per (pass, column sums wanted): the runtime tests used to cut the sweep into ~10 basic blocks
      // per i block and the table gathers could not be batched
      auto sweep = [&](auto pass_tag, auto wc_tag) {
        constexpr int PASS = decltype(pass_tag)::value;
        constexpr bool WC = decltype(wc_tag)::value;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const int i = iw + a * 16 + q4 * 4;
          float bv[4] = {0.f, 0.f, 0.f, 0.f};
          if ((EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_CE) && pr.bias != nullptr && i < Ni) load_bias4(pr, i, bv);
          float cs[4] = {0.f, 0.f, 0.f, 0.f};
          float vv[4][4];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int j = jw + b * 16 + row16;
            const bool ok = j < Nj && i < Ni;
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[b][r] = (float)(__bf16)(acc[a][b][r] + bv[r]);  // what is stored (and what a backward differentiates at)
            if (WC) {
#pragma unroll
              for (int r = 0; r < 4; ++r) cs[r] += ok ? vv[b][r] : 0.f;
            }
          }
          if (EPI == EPI_BIAS_GELU && PASS == 1) {  // x * Phi(x), Phi of the bf16 value from the LDS table
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // eight gathers in flight, then their uses (the scheduler serialises them otherwise)
              float ph[2][4];
#pragma unroll
              for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) ph[b][r] = gelu_tab_at(s_gtab, __float_as_uint(vv[2 * h + b][r]) >> 16);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[2 * h + b][r] *= ph[b][r];
            }
          }
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            uint2 pk;
            pk.x = pack_bf16x2(vv[b][0], vv[b][1]);
            pk.y = pack_bf16x2(vv[b][2], vv[b][3]);
            *reinterpret_cast<uint2 *>(ep + (b * 16 + row16) * 256 + (((a * 2 + (q4 >> 1)) ^ row16) << 4) + (q4 & 1) * 8) = pk;
          }
          if (WC) {
            // sum over this wave's 64 j: the 16 lanes of a q4 group hold different j of the same four i
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float sacc = cs[r];
              sacc += dpp_f32_add<0x111>(sacc);
              sacc += dpp_f32_add<0x112>(sacc);
              sacc += dpp_f32_add<0x114>(sacc);
              sacc += dpp_f32_add<0x118>(sacc);
              if (row16 == 15 && i + r < Ni) atomicAdd(pr.colsum + i + r, sacc);
            }
          }
        }
      };
      typedef std::integral_constant<int, 0> P0;
      typedef std::integral_constant<int, 1> P1;
      if (pass == 1) sweep(P1{}, std::false_type{});
      else if (want_colsum) sweep(P0{}, std::true_type{});
      else sweep(P0{}, std::false_type{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __bf16 *dst = (pass == 1) ? reinterpret_cast<__bf16 *>(pr.out2) : outb;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int jr = it * 4 + q4;          // row of the wave image
      const int j = jw + jr, i = iw + row16 * 8;
      const uint4 v = *reinterpret_cast<const uint4 *>(ep + jr * 256 + ((row16 ^ (jr & 15)) << 4));
      if (j < Nj && i < Ni) *reinterpret_cast<uint4 *>(dst + (long)j * ldo + i) = v;  // Ni % 8 == 0 (host check)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (!SK) return;
  su += kseg;
  if (su >= sue) return;
  BQ_BARRIER();   // the wave-private output images alias the staging buffers the next segment's DMAs write
  }   // segments
}

}  // namespace bq

// ======================================================================================================================
// gemm64_kernel: 64 (i) x BJ (j) output tile, 4 waves (2 x 2), K step 64, three LDS stages filled by LDS-DMA.  For the
// text side of the fusion (M = 80 .. 640 rows: latency-bound, many workgroups of little work each), the object-token
// projections and every weight gradient with a short contraction.  Same operand formulation and LDS images as above.
// ======================================================================================================================
namespace bq {

// KT = K tiles per pipeline step (1 or 2): with KT = 2 a step stages, waits for and consumes TWO 64-wide K tiles between
// barriers -- half the barriers / counted waits of a long contraction (text side: K = 2304 / 3072 at M <= 640 rows, a
// latency chain of 36-48 steps otherwise).

// NS = LDS stages (NS - 1 K tiles in flight).  MEASURED: 5 instead of 3 changes nothing for the detector's cut
// contractions either (tools/bench_det_wgrad.py, to 0.1 us) -- they were bound by their atomics, not by bytes in flight
template <int BJ, bool P_XC, bool Q_XC, int EPI, bool OUT_F32, int KT = 1, int NS = 3>
__global__ __launch_bounds__(256) void gemm64_kernel(const GemmArgs args) {
  static_assert(BJ == 64 || (BJ == 32 && !Q_XC), "32-wide j tiles only for K-contiguous Q");
  static_assert(KT == 1 || KT == 2 || (KT == 4 && BJ == 32), "one, two or (32-row tiles) four K tiles per step");
  constexpr int QF = BJ / 32;               // 16-wide j fragments per wave
  constexpr int Q_UNIT = BJ * 128;          // bytes of the Q image per K tile
  constexpr int TILE_BYTES = 8192 + Q_UNIT;
  constexpr int STAGE = KT * TILE_BYTES;
  // NS LDS stages, NS - 1 K tiles in flight.  MEASURED (profiles/r02_gemm_bench_v2.json): 5 stages instead of 3 change
  // nothing for the forward / dX forms (a K tile costs ~0.24 us either way: the loop is bound by the ISSUE of its 3-4
  // LDS-DMA instructions per wave, ~100 cycles each, not by memory latency) and halve the weight-gradient form
  // (80 KB of LDS = 2 workgroups per CU for a kernel that lives on its output stores) => 3.
  static_assert(NS >= 3 && NS * STAGE <= 160 * 1024, "LDS stages");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int t = blockIdx.x;
  int pi = 0;
  for (int k = 1; k < args.n; ++k)
    if (t >= args.p[k].tile0) pi = k;
  const GemmProblem &pr = args.p[pi];
  const int ksplit = OUT_F32 ? pr.ksplit() : 1;
  const int Ni = pr.Ni, Nj = pr.Nj, Kc = pr.Kc;
  int tl = t - pr.tile0, ks = 0;
  if (ksplit > 1) {
    // a cut contraction (detector weight gradients: millions of rows, 1 - 8 output tiles): the output tiles of ONE piece
    // read the same rows of both operands, so they get neighbouring slots of the same XCD (workgroups go to the XCDs
    // round-robin by blockIdx) and meet in its L2 (PMC: operands fetched once) -- 170 -> 125 us on SA2's first layer
    // against the piece-major order that spreads them over the XCDs
    const int lt = t - pr.tile0;
    const int ntl = pr.tiles_i() * ((Nj + BJ - 1) / BJ);
    if ((ksplit & 7) == 0 && (pr.tile0 & 7) == 0) {
      const int slot = lt >> 3;
      tl = slot % ntl;
      ks = (slot / ntl) * 8 + (lt & 7);
    } else {
      tl = lt % ntl;
      ks = lt / ntl;
    }
  }
  const int bj = tl / pr.tiles_i(), bi = tl % pr.tiles_i();
  const int i0 = bi * 64, j0 = bj * BJ;
  const int ldp = pr.ldp, ldq = pr.ldq;
  const int nkt_all = (Kc + 63) >> 6;
  const int kt_per = (nkt_all + ksplit - 1) / ksplit;
  const int kt0 = ks * kt_per;                         // this workgroup's K tiles: [kt0, kt0 + nkt)
  const int nkt = max(0, min(kt_per, nkt_all - kt0));

  const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void *)pr.P, 0, pr.p_bytes, 0x00020000);
  const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void *)pr.Q, 0, pr.q_bytes, 0x00020000);
  const int cp = lane & 7;
  // P unit: 2 DMAs per wave (rows (2w+d)*8 + lane/8); Q unit: 2 (BJ = 64) or 1 (BJ = 32: rows w*8 + lane/8)
  unsigned vp[2], vq[2];
  // XC Q under a row map (short contractions over a strided (batch, rows) view): the contraction row each DMA stages next
  // and its column offset.  SCALARS on purpose: as small arrays hipcc promoted them to LDS (+ 4 KB per workgroup) and every
  // gemm64 launch of the step got 10-20 us slower (profiles/r04: 15.2 -> 27.1 us on the text side's dX form)
  const bool q_xc_map = Q_XC && pr.q_rpb() != 0;   // (workgroup-uniform)
  int qrow0 = 0, qrow1 = 0;
  unsigned qcol0 = 0, qcol1 = 0;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int ur = (wave * 2 + d) * 8 + (lane >> 3);
    if (!P_XC) vp[d] = (unsigned)(((i0 + ur) * ldp + (cp ^ (ur & 7)) * 8) * 2);
    else vp[d] = (unsigned)((ur * ldp + i0 + (cp ^ (xg(ur) << 1)) * 8) * 2);
    const int uq = (BJ == 64) ? ur : wave * 8 + (lane >> 3);
    // (batched-row map of Q, GemmProblem::q_rpb: on its j rows here, on its contraction rows in the XC form -- see stage())
    if (!Q_XC) vq[d] = (mapped_row(j0 + uq, ldq, pr.q_rpb(), pr.q_bstride) + (unsigned)((cp ^ (uq & 7)) * 8)) * 2u;
    else vq[d] = (unsigned)((uq * ldq + j0 + (cp ^ (xg(uq) << 1)) * 8) * 2);
    if (Q_XC) {
      const int row = kt0 * 64 + uq;
      const unsigned col = (unsigned)(j0 + (cp ^ (xg(uq) << 1)) * 8);
      if (d == 0) { qrow0 = row; qcol0 = col; } else { qrow1 = row; qcol1 = col; }
    }
  }
  const unsigned p_step = P_XC ? (unsigned)(64 * ldp * 2) : 128u;
  const unsigned q_step = Q_XC ? (unsigned)(64 * ldq * 2) : 128u;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    vp[d] += (unsigned)kt0 * p_step;
    vq[d] += (unsigned)kt0 * q_step;
  }
  constexpr int NDMA = KT * (2 + (BJ == 64 ? 2 : 1));  // LDS-DMAs per wave per step

  auto stage = [&](int step) {
#pragma unroll
    for (int h = 0; h < KT; ++h) {
      const bool live = step * KT + h < nkt;
      const unsigned base = (unsigned)((step % NS) * STAGE + h * TILE_BYTES);
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_void_t *)(smem + base + (wave * 2 + d) * 1024), 16,
                                                 live ? vp[d] : 0x80000000u, 0, 0, 0);
        vp[d] += p_step;
      }
#pragma unroll
      for (int d = 0; d < (BJ == 64 ? 2 : 1); ++d) {
        const int blk = (BJ == 64) ? wave * 2 + d : wave;
        if (Q_XC && q_xc_map) {   // short contractions over a strided (batch, rows) view: one division per DMA
          if (d == 0) { vq[0] = (mapped_row(qrow0, ldq, pr.q_rpb(), pr.q_bstride) + qcol0) * 2u; qrow0 += 64; }
          else { vq[1] = (mapped_row(qrow1, ldq, pr.q_rpb(), pr.q_bstride) + qcol1) * 2u; qrow1 += 64; }
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_void_t *)(smem + base + 8192 + blk * 1024), 16,
                                                 live ? vq[d] : 0x80000000u, 0, 0, 0);
        vq[d] += q_step;
      }
    }
  };

  const int row16 = lane & 15, q4 = lane >> 4;
  const int kc_base = row16 * 128 + ((q4 ^ (row16 & 7)) << 4);
  // (XC fragment addresses in closed form: read_frag_cf, gemm_common.h -- sub16 depends on the wave here)
  const int xc_q = (lane & 15) >> 2, xcg = (xc_q >> 1) | ((q4 & 1) << 1);
  const int xc0 = (8 * q4 + xc_q) * 128 + 8 * (lane & 3);

  f32x4 acc[2][QF];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < QF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // weight-gradient form with pr.colsum set: the column sums of Q over this workgroup's share of the contraction (the
  // bias gradient) from all-ones MFMAs on the B fragments, waves wr == 0 of the i = 0 tiles (see gemm256_kernel)
  constexpr bool QSUM = P_XC && Q_XC && OUT_F32;
  const bool do_qsum = QSUM && pr.colsum != nullptr && bi == 0 && wr == 0;   // wave-uniform
  f32x4 qs[QF];
  bf16x8 ones;
#pragma unroll
  for (int b = 0; b < QF; ++b) qs[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

#pragma unroll
  for (int p = 0; p < NS - 1; ++p) stage(p);
  const int nsteps = (nkt + KT - 1) / KT;
  for (int step = 0; step < nsteps; ++step) {
    // step `step` has landed for this wave (steps step+1 .. step+NS-2 may still be in flight); after the barrier: for
    // every wave, and every wave has finished reading the buffer that step step+NS-1 is about to overwrite
    static_assert((NS - 2) * NDMA <= 63, "vmcnt is a 6-bit counter");
    wait_vmcnt<(NS - 2) * NDMA>();
    BQ_BARRIER();
    stage(step + NS - 1);
#pragma unroll
    for (int h = 0; h < KT; ++h) {  // (a K tile past the end was staged as zeros: it adds nothing)
      const unsigned char *buf = smem + (step % NS) * STAGE + h * TILE_BYTES;
      bf16x8 fa[2][2], fb[QF][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[a][kk] = read_frag_cf_x<P_XC>(buf, wr * 2 + a, kk, kc_base, xc0, xcg);
#pragma unroll
      for (int b = 0; b < QF; ++b)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb[b][kk] = read_frag_cf_x<Q_XC>(buf + 8192, wc * QF + b, kk, kc_base, xc0, xcg);
      if (P_XC || Q_XC) {
        // (round 6) the contraction-major reads are inline asm: behind stage() hipcc fenced them with vmcnt(0) -- every step
        // of the weight-gradient forms waited for the DMAs it had just issued for step + NS - 1 (tools/isa_waits.py)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < QF; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][kk], fb[b][kk], acc[a][b], 0, 0, 0);
      if (QSUM && do_qsum) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int b = 0; b < QF; ++b) qs[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fb[b][kk], qs[b], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: straight from the accumulators (8-B bf16 / 16-B fp32 pieces of an output row) ---------------------
  const int ldo = pr.ldo;
  const int iw = i0 + wr * 32, jw = j0 + wc * (BJ / 2);
  const bool atomic_out = ksplit > 1 || pr.accum();   // fp32 out: add to what is there (cut contraction / second row source)
  if (QSUM && do_qsum && q4 == 0) {
#pragma unroll
    for (int b = 0; b < QF; ++b) {
      const int j = jw + b * 16 + row16;
      if (j < Nj) {
        if (!atomic_out) pr.colsum[j] = qs[b][0];
        else atomicAdd(pr.colsum + j, qs[b][0]);   // (zero-initialised by the caller, as `out` is)
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int i = iw + a * 16 + q4 * 4;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if ((EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) && pr.bias != nullptr && i < Ni) load_bias4(pr, i, b4);
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < QF; ++b) {
      const int j = jw + b * 16 + row16;
      const bool ok = j < Nj && i < Ni;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r] + b4[r];
      // element offset of row j of out / out2 / aux (64-bit on plain rows: outputs beyond 2 G elements exist)
      const long jo = pr.o_rpb() ? (long)mapped_row(j < Nj ? j : 0, ldo, pr.o_rpb(), pr.o_bstride) : (long)j * ldo;
      if (OUT_F32) {
        float *dst = reinterpret_cast<float *>(pr.out) + jo + i;
        if (ok && !atomic_out) *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        if (ok && atomic_out) {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(dst + r, v[r]);
        }
        continue;
      }
      if (EPI == EPI_DGELU && ok) {
        const bf16x4 y = *reinterpret_cast<const bf16x4 *>(pr.aux + jo + i);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= dgelu_f((float)y[r]);
      }
      if (EPI == EPI_ADD && ok) {
        const bf16x4 y = *reinterpret_cast<const bf16x4 *>(pr.aux + jo + i);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)y[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (float)(__bf16)v[r];
      if (ok) {
        uint2 pk;
        pk.x = pack_bf16x2(v[0], v[1]);
        pk.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2 *>(reinterpret_cast<__bf16 *>(pr.out) + jo + i) = pk;
        if (EPI == EPI_BIAS_GELU) {
          pk.x = pack_bf16x2(gelu_f(v[0]), gelu_f(v[1]));
          pk.y = pack_bf16x2(gelu_f(v[2]), gelu_f(v[3]));
          *reinterpret_cast<uint2 *>(reinterpret_cast<__bf16 *>(pr.out2) + jo + i) = pk;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[r] += ok ? v[r] : 0.f;
    }
    if (!OUT_F32 && pr.colsum != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = cs[r];
        s += dpp_f32_add<0x111>(s);
        s += dpp_f32_add<0x112>(s);
        s += dpp_f32_add<0x114>(s);
        s += dpp_f32_add<0x118>(s);
        if (row16 == 15 && i + r < Ni) atomicAdd(pr.colsum + i + r, s);
      }
    }
  }
}


// ======================================================================================================================
// pwconv64_kernel: the 1x1 convolution of a SharedMLP layer on point-major rows with the BatchNorm statistics of its
// output taken in the epilogue (reference lib/pointnet2/pytorch_utils.py:104-157 Conv2d(1x1, bias=False) ->
// BatchNorm2d(train) of a SharedMLP, :11-36):   y[r][n] = sum_k x[r][k] W[n][k]   over R = B*npoint*nsample rows
// (2.1 M for SA1 at config c3), K = 3+C input channels (x rows of ldx >= K elements, W zero-padded to Kc % 64 == 0),
// N in {64, 128, 256} output channels.  A streaming problem (SA1 layer 0 moves 840 MB for 36 GFLOP): persistent
// workgroups walk the 64-row tiles of ONE 64-channel block, so that the per-channel sums needed by BatchNorm stay in
// registers across tiles -- taken from the fp32 accumulators, i.e. BEFORE the rounding of y to bf16 -- and leave
// the kernel as one (pivot, sum, sum of squares) record per wave; pwconv_bn_finalize_kernel merges the records in a
// fixed order (Chan's update: exact pairwise merging of (n, mean, M2), robust when |mean| >> std) into scale / shift /
// mean / rstd and the running statistics.  This removes the separate statistics pass over y (268 MB for SA1 layer 0)
// and the library convolution.  Same LDS images / DMA staging / fragment maps as gemm64_kernel.
// ======================================================================================================================
struct PwconvArgs {
  const __bf16 *W;    // [Ni][ldw], zero beyond K
  const __bf16 *X;    // [R][ldx]
  __bf16 *Y;          // [R][Ni]  (MODE 1: the layer's output; MODE 2: [R / S][Ni], the maxima over every run of S rows)
  float *partial;     // [Gj * 2][3][Ni]: pivot | sum (y - pivot) | sum (y - pivot)^2, per (row walker, wave column)
  const float *center;  // [Ni] or null: Y holds y - center (see bq_pwconv_bn_fwd); the statistics are those of y
  int ldw, ldx, Ni, R, Kc, Gj, tiles_i;
  unsigned w_bytes, x_bytes;
  // MODE 1 / 2 (bq_pwconv_bn_apply): BatchNorm's affine map of the STORED values (v = (acc - center) scale + shift), ReLU
  const float *scale, *shift;
  int relu, S;
  // pwconv64s_kernel<NKT, true>: X holds the PREVIOUS layer's stored pre-activation; relu(x xscale + xshift) is applied to
  // every x tile as it arrives in LDS (the previous layer's BatchNorm + ReLU never materialised: bq_pwconv_bn_fwd_x)
  const float *xscale, *xshift;
};

// MODE 0: y and its statistics (the first pass of a SharedMLP layer).  MODE 1 / 2 (round 5): the SAME
// product once more with BatchNorm + ReLU (+ the max over nsample) applied to the fp32 ACCUMULATORS: the layer's output no
// longer passes through a bf16 pre-activation -- the one rounding class that reproduced the detector's convergence gap
// (tools/loss_gap_probe.py) -- for the price of reading x (ldx elements per row) instead of the stored y (Ni elements).
template <int MODE>
__global__ __launch_bounds__(256) void pwconv64_kernel(const PwconvArgs a) {
  constexpr int STAGE = 16384, NS = 3;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int bi = blockIdx.x % a.tiles_i, gj = blockIdx.x / a.tiles_i;
  const int i0 = bi * 64;
  const int nkt = a.Kc >> 6;
  const int tiles_j = (a.R + 63) >> 6;
  const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void *)a.W, 0, a.w_bytes, 0x00020000);
  const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void *)a.X, 0, a.x_bytes, 0x00020000);
  const int cp = lane & 7;
  const int row16 = lane & 15, q4 = lane >> 4;
  const int kc_base = row16 * 128 + ((q4 ^ (row16 & 7)) << 4);
  const int xc_dummy[4] = {0, 0, 0, 0};
  const int iw = i0 + wr * 32;

  float piv[2][4], s1[2][4], s2[2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) { piv[x][r] = 0.f; s1[x][r] = 0.f; s2[x][r] = 0.f; }
  bool have_pivot = false;
  float ctr[2][4];   // this lane's eight channels of the centre
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) ctr[x][r] = a.center ? a.center[iw + x * 16 + q4 * 4 + r] : 0.f;

  for (int bj = gj; bj < tiles_j; bj += a.Gj) {
    const int j0 = bj * 64;
    unsigned vp[2], vq[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int ur = (wave * 2 + d) * 8 + (lane >> 3);
      vp[d] = (unsigned)(((i0 + ur) * a.ldw + (cp ^ (ur & 7)) * 8) * 2);
      vq[d] = (unsigned)(((long)(j0 + ur) * a.ldx + (cp ^ (ur & 7)) * 8) * 2);
    }
    auto stage = [&](int kt) {
      const bool live = kt < nkt;
      const unsigned base = (unsigned)((kt % NS) * STAGE);
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_void_t *)(smem + base + (wave * 2 + d) * 1024), 16,
                                                 live ? vp[d] : 0x80000000u, 0, 0, 0);
        vp[d] += 128u;
      }
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_void_t *)(smem + base + 8192 + (wave * 2 + d) * 1024), 16,
                                                 live ? vq[d] : 0x80000000u, 0, 0, 0);
        vq[d] += 128u;
      }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[x][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    BQ_BARRIER();  // every wave has finished reading the stages of the previous tile
    stage(0);
    stage(1);
    for (int kt = 0; kt < nkt; ++kt) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      BQ_BARRIER();
      stage(kt + 2);
      const unsigned char *buf = smem + (kt % NS) * STAGE;
      bf16x8 fa[2][2], fb[2][2];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[x][kk] = read_frag<false>(buf, wr * 2 + x, kk, kc_base, xc_dummy);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb[b][kk] = read_frag<false>(buf + 8192, wc * 2 + b, kk, kc_base, xc_dummy);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[x][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[x][kk], fb[b][kk], acc[x][b], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int jw = j0 + wc * 32;
    if constexpr (MODE != 0) {
      // ---- epilogue of the apply pass: v = (acc - centre) scale + shift, ReLU, bf16; MODE 2: max over runs of S rows -----
      float v[2][2][4];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int i = iw + x * 16 + q4 * 4;
        const float4 sc = *reinterpret_cast<const float4 *>(a.scale + i), sh = *reinterpret_cast<const float4 *>(a.shift + i);
        const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t = __builtin_fmaf(acc[x][b][r], scv[r], shv[r]);   // shift = beta - mean scale of the product itself
            v[x][b][r] = a.relu ? fmaxf(t, 0.f) : t;
          }
      }
      if constexpr (MODE == 1) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const int i = iw + x * 16 + q4 * 4;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int j = jw + b * 16 + row16;
            if (j < a.R) {
              uint2 pk;
              pk.x = pack_bf16x2(v[x][b][0], v[x][b][1]);
              pk.y = pack_bf16x2(v[x][b][2], v[x][b][3]);
              *reinterpret_cast<uint2 *>(a.Y + (long)j * a.Ni + i) = pk;
            }
          }
        }
      } else {
        // rows past R never win (R is a multiple of S: a run is wholly inside or wholly outside)
        float *s_max = reinterpret_cast<float *>(smem);   // [wc][wr][x][q4][r]: the staging buffers are idle here
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const bool ok = jw + b * 16 + row16 < a.R;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float m = ok ? v[x][b][r] : -3.0e38f;
              m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2));
              m = fmaxf(m, __shfl_xor(m, 4)); m = fmaxf(m, __shfl_xor(m, 8));
              v[x][b][r] = m;   // the maximum over the 16 rows of block b, in every lane of the q4 group
            }
          }
        if (a.S == 16) {
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const int j = jw + b * 16;
              if (row16 == 0 && j < a.R) {
                uint2 pk;
                pk.x = pack_bf16x2(v[x][b][0], v[x][b][1]);
                pk.y = pack_bf16x2(v[x][b][2], v[x][b][3]);
                *reinterpret_cast<uint2 *>(a.Y + (long)(j >> 4) * a.Ni + iw + x * 16 + q4 * 4) = pk;
              }
            }
        } else {
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[x][0][r] = fmaxf(v[x][0][r], v[x][1][r]);   // the wave column's 32 rows
          if (a.S == 32) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
              if (row16 == 0 && jw < a.R) {
                uint2 pk;
                pk.x = pack_bf16x2(v[x][0][0], v[x][0][1]);
                pk.y = pack_bf16x2(v[x][0][2], v[x][0][3]);
                *reinterpret_cast<uint2 *>(a.Y + (long)(jw >> 5) * a.Ni + iw + x * 16 + q4 * 4) = pk;
              }
          } else {   // S == 64: the two wave columns meet in LDS
            BQ_BARRIER();   // (every wave is done with this tile's fragment reads)
            if (wc == 1 && row16 == 0) {
#pragma unroll
              for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_max[((wr * 2 + x) * 4 + q4) * 4 + r] = v[x][0][r];
            }
            __syncthreads();
            if (wc == 0 && row16 == 0 && j0 < a.R) {
#pragma unroll
              for (int x = 0; x < 2; ++x) {
                float m[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) m[r] = fmaxf(v[x][0][r], s_max[((wr * 2 + x) * 4 + q4) * 4 + r]);
                uint2 pk;
                pk.x = pack_bf16x2(m[0], m[1]);
                pk.y = pack_bf16x2(m[2], m[3]);
                *reinterpret_cast<uint2 *>(a.Y + (long)(j0 >> 6) * a.Ni + iw + x * 16 + q4 * 4) = pk;
              }
            }
          }
        }
      }
      continue;
    }
    // ---- epilogue: y (bf16) and the statistics of the fp32 values ---------------------------------------------------
    if (!have_pivot) {  // this wave's first row: the value every lane of a 16-lane group subtracts from its channels
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) piv[x][r] = __shfl(acc[x][0][r], lane & 48);
      have_pivot = true;
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int i = iw + x * 16 + q4 * 4;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int j = jw + b * 16 + row16;
        const bool ok = j < a.R;
        if (ok) {
          uint2 pk;
          pk.x = pack_bf16x2(acc[x][b][0] - ctr[x][0], acc[x][b][1] - ctr[x][1]);
          pk.y = pack_bf16x2(acc[x][b][2] - ctr[x][2], acc[x][b][3] - ctr[x][3]);
          *reinterpret_cast<uint2 *>(a.Y + (long)j * a.Ni + i) = pk;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = ok ? acc[x][b][r] - piv[x][r] : 0.f;
          s1[x][r] += d;
          s2[x][r] += d * d;
        }
      }
    }
  }
  if constexpr (MODE != 0) return;
  // ---- one record per (row walker gj, wave column wc): the 16 lanes of a q4 group hold different rows ---------------
  float *rec = a.partial + (long)(gj * 2 + wc) * 3 * a.Ni;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float u = s1[x][r], v = s2[x][r];
      u += dpp_f32_add<0x111>(u); u += dpp_f32_add<0x112>(u); u += dpp_f32_add<0x114>(u); u += dpp_f32_add<0x118>(u);
      v += dpp_f32_add<0x111>(v); v += dpp_f32_add<0x112>(v); v += dpp_f32_add<0x114>(v); v += dpp_f32_add<0x118>(v);
      if (row16 == 15) {
        const int i = iw + x * 16 + q4 * 4 + r;
        rec[i] = piv[x][r];
        rec[a.Ni + i] = u;
        rec[2 * a.Ni + i] = v;
      }
    }
}

// pwconv64s_kernel (round 5): MODE 0 of the kernel above as a STREAM over the row tiles.  There every tile started with an empty
// pipeline -- barrier, two DMA stages, a full memory latency before the first MFMA -- and fetched the 64-channel weight block
// again; SA1's layers ran at 2.9 TB/s.  Here the weight block (NKT units of 64 x 64) is loaded once per workgroup and stays in
// LDS, a stage is ALL of a tile's NKT x units, and stages run NS - 1 tiles ahead across tile boundaries: one counted wait and
// one barrier per tile, the y stores of a tile retire under the next tiles' loads (buffer stores, dropped past the end, so that
// every wave issues the same number of vector-memory operations per tile: the counted waits below rely on it).  Same tile walk
// (gj, gj + Gj, ...), same statistics records, same results as the kernel above.
template <int NKT>
struct PwsCfg {
  static constexpr int NS = NKT <= 2 ? 3 : 2;
  static constexpr int STAGE = NKT * 8192, WB = NKT * 8192, LDS = WB + NS * STAGE;
  static constexpr int NDMA = 2 * NKT, NST = 4;
  static constexpr int WAITN = (NS - 2) * (NDMA + NST) + NST;
};

template <int NKT, bool XT>
__global__ __launch_bounds__(256) void pwconv64s_kernel(const PwconvArgs a) {
  using C = PwsCfg<NKT>;
  constexpr int NS = C::NS, STAGE = C::STAGE;
  static_assert(C::LDS <= 160 * 1024 && C::WAITN <= 63, "LDS / counted waits");
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS];
  unsigned char *const wimg = smem, *const stages = smem + C::WB;
  const unsigned DEAD = 0x80000000u;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int bi = blockIdx.x % a.tiles_i, gj = blockIdx.x / a.tiles_i;
  const int i0 = bi * 64;
  const int tiles_j = (a.R + 63) >> 6;
  const int ntile = gj < tiles_j ? (tiles_j - 1 - gj) / a.Gj + 1 : 0;   // tiles gj, gj + Gj, ...
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void *)a.W, 0, a.w_bytes, 0x00020000);
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void *)a.X, 0, a.x_bytes, 0x00020000);
  const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void *)a.Y, 0, (unsigned)((long)a.R * a.Ni * 2), 0x00020000);
  const int cp = lane & 7;
  const int row16 = lane & 15, q4 = lane >> 4;
  const int kc_base = row16 * 128 + ((q4 ^ (row16 & 7)) << 4);
  const int xc_dummy[4] = {0, 0, 0, 0};
  const int iw = i0 + wr * 32;

  unsigned vq[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int ur = (wave * 2 + d) * 8 + (lane >> 3);
    const unsigned wo = (unsigned)(((i0 + ur) * a.ldw + (cp ^ (ur & 7)) * 8) * 2);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)   // the weight block: K-contiguous units, once
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_t *)(wimg + kt * 8192 + (wave * 2 + d) * 1024), 16, wo + kt * 128u, 0,
                                               0, 0);
    vq[d] = (unsigned)(((long)(gj * 64 + ur) * a.ldx + (cp ^ (ur & 7)) * 8) * 2);
  }
  const unsigned x_step = (unsigned)((long)a.Gj * 64 * a.ldx * 2);
  auto stage = [&](int t) {
    const bool live = t < ntile;
    unsigned char *base = stages + (t % NS) * STAGE;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_t *)(base + kt * 8192 + (wave * 2 + d) * 1024), 16,
                                                 live ? vq[d] + kt * 128u : DEAD, 0, 0, 0);
      vq[d] += x_step;
    }
  };

  float piv[2][4], s1[2][4], s2[2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) { piv[x][r] = 0.f; s1[x][r] = 0.f; s2[x][r] = 0.f; }
  bool have_pivot = false;
  float ctr[2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) ctr[x][r] = a.center ? a.center[iw + x * 16 + q4 * 4 + r] : 0.f;

  // XT: this lane's 8 channels (logical 16-byte chunk lane & 7) of every K unit of the previous layer's BatchNorm map
  [[maybe_unused]] float xs_c[XT ? NKT : 1][8], xh_c[XT ? NKT : 1][8];
  if constexpr (XT) {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xs_c[kt][e] = a.xscale[kt * 64 + cp * 8 + e];
        xh_c[kt][e] = a.xshift[kt * 64 + cp * 8 + e];
      }
  }

#pragma unroll
  for (int p = 0; p < NS - 1; ++p) stage(p);
  for (int t = 0; t < ntile; ++t) {
    // behind this tile's DMAs lie (NS - 2) later stages and the y stores of the tiles in between (none in the first two trips)
    if (t < 2) wait_vmcnt<(NS - 2) * C::NDMA>();
    else wait_vmcnt<C::WAITN>();
    if constexpr (XT) {
      // every wave turns ITS OWN 16 rows of the x units (the rows its DMAs wrote) into relu(x xscale + xshift) in place
      unsigned char *xb = stages + (t % NS) * STAGE;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int row = wave * 16 + (lane >> 3) + 8 * d;
          unsigned char *pp = xb + kt * 8192 + row * 128 + ((cp ^ ((lane >> 3) & 7)) << 4);
          const bf16x8 v = *reinterpret_cast<const bf16x8 *>(pp);
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (__bf16)fmaxf((float)v[e] * xs_c[kt][e] + xh_c[kt][e], 0.0f);
          *reinterpret_cast<bf16x8 *>(pp) = o;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    BQ_BARRIER();
    stage(t + NS - 1);
    const unsigned char *buf = stages + (t % NS) * STAGE;
    f32x4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[x][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      bf16x8 fa[2][2], fb[2][2];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[x][kk] = read_frag<false>(wimg + kt * 8192, wr * 2 + x, kk, kc_base, xc_dummy);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb[b][kk] = read_frag<false>(buf + kt * 8192, wc * 2 + b, kk, kc_base, xc_dummy);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[x][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[x][kk], fb[b][kk], acc[x][b], 0, 0, 0);
    }
    // ---- epilogue: y (bf16) and the statistics of the fp32 values ---------------------------------------------------
    const int jw = (gj + t * a.Gj) * 64 + wc * 32;
    if (!have_pivot) {
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) piv[x][r] = __shfl(acc[x][0][r], lane & 48);
      have_pivot = true;
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int i = iw + x * 16 + q4 * 4;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int j = jw + b * 16 + row16;
        const bool ok = j < a.R;
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        u32x2_t pk;
        pk[0] = pack_bf16x2(acc[x][b][0] - ctr[x][0], acc[x][b][1] - ctr[x][1]);
        pk[1] = pack_bf16x2(acc[x][b][2] - ctr[x][2], acc[x][b][3] - ctr[x][3]);
        __builtin_amdgcn_raw_buffer_store_b64(pk, rsY, ok ? (unsigned)(((long)j * a.Ni + i) * 2) : DEAD, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = ok ? acc[x][b][r] - piv[x][r] : 0.f;
          s1[x][r] += d;
          s2[x][r] += d * d;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float *rec = a.partial + (long)(gj * 2 + wc) * 3 * a.Ni;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float u = s1[x][r], v = s2[x][r];
      u += dpp_f32_add<0x111>(u); u += dpp_f32_add<0x112>(u); u += dpp_f32_add<0x114>(u); u += dpp_f32_add<0x118>(u);
      v += dpp_f32_add<0x111>(v); v += dpp_f32_add<0x112>(v); v += dpp_f32_add<0x114>(v); v += dpp_f32_add<0x118>(v);
      if (row16 == 15) {
        const int i = iw + x * 16 + q4 * 4 + r;
        rec[i] = piv[x][r];
        rec[a.Ni + i] = u;
        rec[2 * a.Ni + i] = v;
      }
    }
}

// rows a (row walker gj, wave column wc) record covers: tiles gj, gj+Gj, ...; rows [64 t + 32 wc, +32) below R
__device__ __forceinline__ long pwconv_record_rows(int rec, int Gj, long R) {
  const int gj = rec >> 1, wc = rec & 1;
  const long tiles_j = (R + 63) >> 6;
  if (gj >= tiles_j) return 0;
  const long nt = (tiles_j - 1 - gj) / Gj + 1;          // tiles walked
  const long last = gj + (nt - 1) * Gj;                 // the only tile that can be ragged
  const long r0 = last * 64 + wc * 32;
  const long in_last = R > r0 ? (R - r0 < 32 ? R - r0 : 32) : 0;
  return (nt - 1) * 32 + in_last;
}

struct PwconvBnParams {
  const float *partial;
  const float *gamma, *beta;
  float *running_mean, *running_var;
  long long *num_batches_tracked;
  float *scale, *shift, *mean, *rstd;
  float eps, momentum;
  int C, nrec, Gj;
  long R;
  const float *center;   // what the stored y had subtracted (null: nothing): shift / mean describe the STORED values
  float *shift_acc;      // optional [C]: beta - mean scale for the UNcentred product (bq_pwconv_bn_apply reads the fp32
                         // accumulators; the centre -- usually running_mean itself -- has moved by then)
};

// 16 threads per channel: thread phase ph merges the records g = ph (mod 16) in index order, then the sixteen partial
// (n, mean, M2) triples are merged in phase order (Chan et al.: exact for any partition, no cancellation; the fixed
// order makes the statistics bit-reproducible).  (4 phases took 46 us per layer: a serial chain of ~250 dependent
// merges per thread; 16 phases keep the chain at ~64.)
__device__ __forceinline__ void chan_merge(float &n, float &mean, float &m2, float ng, float mg, float m2g) {
  if (ng == 0.f) return;
  const float tot = n + ng, delta = mg - mean;
  mean += delta * (ng / tot);
  m2 += m2g + delta * delta * (n * ng / tot);
  n = tot;
}
// 16 channels x 64 phases per workgroup (round 5: 64 channels x 16 phases left every thread a chain of nrec / 16 = 64 records,
// 16 dependent batches of loads -- 23 us per layer, 0.5 ms per step over the detector's 23 layers): a thread merges nrec / 64
// records, four loads in flight, then the 64 phases of a channel meet in LDS in a fixed order
constexpr int PWF_PH = 64, PWF_CH = 16;
__global__ __launch_bounds__(1024) void pwconv_bn_finalize_kernel(const PwconvBnParams p) {
  __shared__ float sh[3][PWF_PH][PWF_CH];
  const int cl = threadIdx.x & (PWF_CH - 1), ph = threadIdx.x / PWF_CH;
  const int c = min(blockIdx.x * PWF_CH + cl, p.C - 1);
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int g0 = ph; g0 < p.nrec; g0 += 8 * PWF_PH) {
    float pv[8], su[8], sq[8], ng[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {  // eight records in flight per thread
      const int g = g0 + PWF_PH * u;
      const bool live = g < p.nrec;
      const float *rec = p.partial + (long)(live ? g : 0) * 3 * p.C;
      pv[u] = rec[c]; su[u] = rec[p.C + c]; sq[u] = rec[2 * p.C + c];
      ng[u] = live ? (float)pwconv_record_rows(g, p.Gj, p.R) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float inv = ng[u] > 0.f ? 1.0f / ng[u] : 0.f;
      chan_merge(n, mean, m2, ng[u], pv[u] + su[u] * inv, fmaxf(sq[u] - su[u] * su[u] * inv, 0.f));
    }
  }
  sh[0][ph][cl] = n; sh[1][ph][cl] = mean; sh[2][ph][cl] = m2;
  __syncthreads();
  // 64 -> 8 -> 1: phases 8 q .. 8 q + 7 merged by thread (q, cl), then the eight results by thread (0, cl)
  if (ph < 8) {
    n = 0.f; mean = 0.f; m2 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) chan_merge(n, mean, m2, sh[0][ph * 8 + q][cl], sh[1][ph * 8 + q][cl], sh[2][ph * 8 + q][cl]);
  }
  __syncthreads();
  if (ph < 8) { sh[0][ph][cl] = n; sh[1][ph][cl] = mean; sh[2][ph][cl] = m2; }
  __syncthreads();
  if (ph != 0 || blockIdx.x * PWF_CH + cl >= p.C) return;
  n = 0.f; mean = 0.f; m2 = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) chan_merge(n, mean, m2, sh[0][q][cl], sh[1][q][cl], sh[2][q][cl]);
  const float var = m2 / n;
  const float rstd = rsqrtf(var + p.eps);
  const float sc = p.gamma[c] * rstd;
  const float stored_mean = mean - (p.center ? p.center[c] : 0.f);   // (read before running_mean -- the usual centre -- moves)
  p.scale[c] = sc;
  p.shift[c] = p.beta[c] - stored_mean * sc;
  p.mean[c] = stored_mean;
  p.rstd[c] = rstd;
  if (p.shift_acc) p.shift_acc[c] = p.beta[c] - mean * sc;
  if (p.running_mean) {
    const float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
    p.running_mean[c] = (1.0f - p.momentum) * p.running_mean[c] + p.momentum * mean;
    p.running_var[c] = (1.0f - p.momentum) * p.running_var[c] + p.momentum * unbiased;
  }
  if (c == 0 && p.num_batches_tracked) p.num_batches_tracked[0] += 1;
}

// ---- grouped column sums: out_p[n] = sum_m G_p[m][n] for a list of bf16 matrices, ONE launch -----------------------
// (the bias gradients of every parked linear: reference autograd of nn.Linear's bias, torch `grad.sum(0)`).
// Workgroup = 256 columns x up to 512 rows: thread = 8 columns (16-B loads) of every 8th row; the 8 row groups meet in
// LDS, then one fp32 atomic per column (out is zeroed by the caller; a few hundred atomics per workgroup).
struct ColsumProblem {
  const __bf16 *g;
  float *out;
  int M, N, ld, wg0, cblocks;
};
constexpr int COLSUM_MAX_PROBLEMS = 96;
constexpr int COLSUM_ROWS = 512;
struct ColsumArgs {
  int n;
  ColsumProblem p[COLSUM_MAX_PROBLEMS];
};

__global__ __launch_bounds__(256) void colsum_grouped_kernel(const ColsumArgs args) {
  __shared__ float red[8][256];
  const int t = blockIdx.x;
  int pi = 0;
  for (int k = 1; k < args.n; ++k)
    if (t >= args.p[k].wg0) pi = k;
  const ColsumProblem &pr = args.p[pi];
  const int tl = t - pr.wg0;
  const int cb = tl % pr.cblocks, rb = tl / pr.cblocks;
  const int tc = threadIdx.x & 31, tr = threadIdx.x >> 5;
  const int col = cb * 256 + tc * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < pr.N) {  // N % 8 == 0
    const int r1 = min(pr.M, (rb + 1) * COLSUM_ROWS);
    for (int r = rb * COLSUM_ROWS + tr; r < r1; r += 8) {
      const bf16x8 v = *reinterpret_cast<const bf16x8 *>(pr.g + (long)r * pr.ld + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tr][tc * 8 + e] = s[e];
  __syncthreads();
  const int c = threadIdx.x;
  float tot = 0.f;
#pragma unroll
  for (int g8 = 0; g8 < 8; ++g8) tot += red[g8][c];
  if (cb * 256 + c < pr.N) atomicAdd(pr.out + cb * 256 + c, tot);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// long_k: every problem of the launch has >= 12 K tiles and the launch is small (host decision in launch_gemm):
// the bf16-output small-tile forms then run two K tiles per pipeline step
template <bool P_XC, bool Q_XC, int EPI, bool OUT_F32>
static int launch_variant(const GemmArgs &ga, int tile, hipStream_t st, bool long_k = false) {
  constexpr bool KT2_OK = !OUT_F32 && !Q_XC && EPI != EPI_BIAS_CE;
  if (tile == 256) {
    // (the K-contiguous dX forms -- a pre-transposed weight, csrc/transpose.hip -- exist for the small tiles and 256 x 128)
    if constexpr (P_XC || (EPI != EPI_DGELU && EPI != EPI_ADD))
      hipLaunchKernelGGL((gemm256_kernel<P_XC, Q_XC, EPI, OUT_F32>), dim3(ga.total_tiles), dim3(512), 0, st, ga);
    else
      return -1;
  } else if (tile == 64) {
    if constexpr (KT2_OK) {
      if (long_k) {
        hipLaunchKernelGGL((gemm64_kernel<64, P_XC, Q_XC, EPI, OUT_F32, 2>), dim3(ga.total_tiles), dim3(256), 0, st, ga);
        return 0;
      }
    }
    hipLaunchKernelGGL((gemm64_kernel<64, P_XC, Q_XC, EPI, OUT_F32>), dim3(ga.total_tiles), dim3(256), 0, st, ga);
  } else {
    if constexpr (!Q_XC) {
      if constexpr (KT2_OK) {
        // four K tiles per step (144 KB of LDS: one workgroup per CU) for launches of <= 512 tiles: c3 45.05 -> 44.57 ms
        constexpr int k4_tiles = 12;
        bool k4 = long_k && k4_tiles > 0 && ga.total_tiles <= 512;
        for (int k = 0; k < ga.n; ++k) k4 = k4 && ga.p[k].Kc >= 64 * k4_tiles;
        if (k4) {
          hipLaunchKernelGGL((gemm64_kernel<32, P_XC, Q_XC, EPI, OUT_F32, 4>), dim3(ga.total_tiles), dim3(256), 0, st, ga);
          return 0;
        }
        if (long_k) {
          hipLaunchKernelGGL((gemm64_kernel<32, P_XC, Q_XC, EPI, OUT_F32, 2>), dim3(ga.total_tiles), dim3(256), 0, st, ga);
          return 0;
        }
      }
      hipLaunchKernelGGL((gemm64_kernel<32, P_XC, Q_XC, EPI, OUT_F32>), dim3(ga.total_tiles), dim3(256), 0, st, ga);
    } else {
      return -1;
    }
  }
  return 0;
}

static int launch_gemm(const GemmArgs &ga, int flags, int epi, int tile, hipStream_t st) {
  const bool pxc = flags & BQ_GEMM_P_XC, qxc = flags & BQ_GEMM_Q_XC, f32 = flags & BQ_GEMM_OUT_F32;
  // measured on the c3 step (A/B in one call): never 46.2 ms, from 24 K tiles 45.5-45.9, from 12 K tiles 45.3-45.6
  constexpr int long_k_tiles = 12;
  bool long_k = long_k_tiles > 0 && tile != 256 && ga.total_tiles <= 2048;
  for (int k = 0; k < ga.n; ++k) long_k = long_k && ga.p[k].Kc >= 64 * long_k_tiles;
  if (tile == 128 && (gemm_sk_mode() & 2) && !(flags & BQ_GEMM_BACKGROUND) && ga.n == 1 && !pxc && !qxc && !f32 &&
      (epi == EPI_NONE || epi == EPI_BIAS) && ga.p[0].colsum == nullptr && ga.p[0].q_rpb() == 0 && ga.p[0].o_rpb() == 0) {
    // stream-K on the 256 x 256 kernel (header of gemm256_kernel): one problem with a long contraction whose 256 x 256 tiles
    // would leave >= 15 % of a one-workgroup-per-CU grid idle, every share still >= 16 K tiles
    const GemmProblem &p0 = ga.p[0];
    const int cus = device_cus();
    const int nkt = (p0.Kc + 63) >> 6, tiles = ((p0.Ni + 255) / 256) * ((p0.Nj + 255) / 256);
    const long units = (long)tiles * nkt;
    const int rounds = (tiles + cus - 1) / cus;
    long ws_bytes = 0;
    void *ws = gemm_sk_workspace(st, &ws_bytes);
    if (ws != nullptr && nkt >= 24 && tiles >= cus / 2 && units / cus >= 16 && (long)rounds * nkt * cus >= units * 115 / 100 &&
        8L * tiles <= SK_TICKET_BYTES && ws_bytes >= SK_TICKET_BYTES + 2L * cus * 256 * 256 * 4) {
      GemmArgs g2 = ga;
      g2.sk_ws = ws;
      g2.total_tiles = tiles;
      g2.p[0].tile0 = 0;
      g2.p[0].tiles_ks = (p0.Ni + 255) / 256;   // tiles along i, no K split
      if (epi == EPI_NONE) hipLaunchKernelGGL((gemm256_kernel<false, false, EPI_NONE, false, true>), dim3(cus), dim3(512), 0, st, g2);
      else hipLaunchKernelGGL((gemm256_kernel<false, false, EPI_BIAS, false, true>), dim3(cus), dim3(512), 0, st, g2);
      return 0;
    }
  }
  if (tile == 128)  // csrc/gemm_mid.hip: forward / dX (bf16 out, K-contiguous Q) and the weight-gradient form
    return launch_gemm_mid(ga, pxc, qxc, f32, epi, (flags & BQ_GEMM_BACKGROUND) != 0, st);
  if (!pxc && !qxc && !f32) {
    if (epi == EPI_NONE) return launch_variant<false, false, EPI_NONE, false>(ga, tile, st, long_k);
    if (epi == EPI_BIAS) return launch_variant<false, false, EPI_BIAS, false>(ga, tile, st, long_k);
    if (epi == EPI_BIAS_GELU) return launch_variant<false, false, EPI_BIAS_GELU, false>(ga, tile, st, long_k);
    if (epi == EPI_DGELU) return launch_variant<false, false, EPI_DGELU, false>(ga, tile, st, long_k);
    if (epi == EPI_ADD) return launch_variant<false, false, EPI_ADD, false>(ga, tile, st, long_k);
    if (epi == EPI_BIAS_CE && tile == 256) {
      hipLaunchKernelGGL((gemm256_kernel<false, false, EPI_BIAS_CE, false>), dim3(ga.total_tiles), dim3(512), 0, st, ga);
      return 0;
    }
  } else if (!pxc && !qxc && f32) {  // forward with fp32 results (the detector's regression heads: 64 / 32 tiles)
    if (epi == EPI_NONE && tile != 256) return launch_variant<false, false, EPI_NONE, true>(ga, tile, st);
    if (epi == EPI_BIAS && tile != 256) return launch_variant<false, false, EPI_BIAS, true>(ga, tile, st);
  } else if (pxc && !qxc && f32) {
    if (epi == EPI_NONE && tile != 256) return launch_variant<true, false, EPI_NONE, true>(ga, tile, st);
  } else if (pxc && !qxc && !f32) {
    if (epi == EPI_NONE) return launch_variant<true, false, EPI_NONE, false>(ga, tile, st, long_k);
    if (epi == EPI_DGELU) return launch_variant<true, false, EPI_DGELU, false>(ga, tile, st, long_k);
    if (epi == EPI_ADD) return launch_variant<true, false, EPI_ADD, false>(ga, tile, st, long_k);
  } else if (pxc && qxc && f32) {
    if (epi == EPI_NONE) return launch_variant<true, true, EPI_NONE, true>(ga, tile, st);
  }
  return -1;
}

}  // namespace bq

extern "C" int bq_gemm_max_problems(void) { return bq::GEMM_MAX_PROBLEMS; }

extern "C" int bq_gemm_bf16(const bq_gemm_desc *d, int n, int flags, int epilogue, int tile, void *stream) {
  using namespace bq;
  BQ_REQUIRE(d != nullptr && n >= 1, BQ_EINVAL, "bq_gemm_bf16: no problems");
  BQ_REQUIRE(tile == 256 || tile == 128 || tile == 64 || tile == 32, BQ_EINVAL,
             "bq_gemm_bf16: tile must be 256, 128 (= 256 x 128), 64 or 32 (got %d)", tile);
  const bool pxc = flags & BQ_GEMM_P_XC, qxc = flags & BQ_GEMM_Q_XC, f32 = flags & BQ_GEMM_OUT_F32;
  BQ_REQUIRE(!(flags & BQ_GEMM_BACKGROUND) || tile == 128, BQ_EINVAL, "bq_gemm_bf16: BQ_GEMM_BACKGROUND needs tile 128");
  hipStream_t st = (hipStream_t)stream;
  int done = 0;
  while (done < n) {
    GemmArgs ga;
    ga.n = 0;
    ga.total_tiles = 0;
    while (done < n && ga.n < GEMM_MAX_PROBLEMS) {
      const bq_gemm_desc &s = d[done];
      BQ_REQUIRE(s.P && s.Q && s.out, BQ_EINVAL, "bq_gemm_bf16: null operand (problem %d)", done);
      BQ_REQUIRE(s.Ni > 0 && s.Nj > 0 && s.Kc > 0, BQ_EINVAL, "bq_gemm_bf16: empty problem %d", done);
      BQ_REQUIRE(s.ldp % 8 == 0 && s.ldq % 8 == 0, BQ_EINVAL, "bq_gemm_bf16: operand leading dimensions must be multiples of 8 elements");
      BQ_REQUIRE(s.Ni % 8 == 0 && s.ldo % (f32 ? 4 : 8) == 0, BQ_EINVAL, "bq_gemm_bf16: Ni and ldo must be multiples of 8 (Ni = %d, ldo = %d)", s.Ni, s.ldo);
      BQ_REQUIRE(!qxc || s.Nj % 8 == 0, BQ_EINVAL, "bq_gemm_bf16: a contraction-major Q needs Nj %% 8 == 0 (Nj = %d)", s.Nj);
      BQ_REQUIRE((pxc && qxc) || s.Kc % 64 == 0, BQ_EINVAL,
                 "bq_gemm_bf16: a K-contiguous operand needs Kc %% 64 == 0 (Kc = %d; zero-pad the other operand and pass "
                 "p_bytes / q_bytes when the rows are shorter)", s.Kc);
      BQ_REQUIRE(((uintptr_t)s.P % 16 == 0) && ((uintptr_t)s.Q % 16 == 0) && ((uintptr_t)s.out % 16 == 0), BQ_EINVAL,
                 "bq_gemm_bf16: operands must be 16-byte aligned");
      BQ_REQUIRE(epilogue != EPI_BIAS_GELU || s.out2, BQ_EINVAL, "bq_gemm_bf16: BIAS_GELU needs out2");
      BQ_REQUIRE(epilogue != EPI_DGELU || s.aux, BQ_EINVAL, "bq_gemm_bf16: DGELU needs aux");
      BQ_REQUIRE(epilogue != EPI_ADD || s.aux, BQ_EINVAL, "bq_gemm_bf16: ADD needs aux");
      BQ_REQUIRE((epilogue != EPI_ADD && epilogue != EPI_DGELU && tile != 128) || ((long)s.Nj + 256) * s.ldo * (f32 ? 4 : 2) < 0x7FFFFFFFL,
                 BQ_EINVAL, "bq_gemm_bf16: out / aux larger than 2 GB (problem %d)", done);
      const long pb = pxc ? ((long)(s.Kc - 1) * s.ldp + s.Ni) * 2 : ((long)(s.Ni - 1) * s.ldp + s.Kc) * 2;
      long qb = qxc ? ((long)(s.Kc - 1) * s.ldq + s.Nj) * 2 : ((long)(s.Nj - 1) * s.ldq + s.Kc) * 2;
      const int tj = tile == 256 ? 256 : tile, ti = (tile == 256 || tile == 128) ? 256 : 64;
      // batched-row maps: the mapped rows of Q are its j rows, or its contraction rows when Q is contraction-major
      BQ_REQUIRE(s.q_rpb >= 0 && s.o_rpb >= 0 && s.q_rpb <= 65535 && s.o_rpb <= 65535, BQ_EINVAL,
                 "bq_gemm_bf16: q_rpb / o_rpb must be in 0..65535 (problem %d)", done);
      long q_span = 0, o_span = 0;   // elements from the first mapped row to one past the last (map on: bounds of the views)
      if (s.q_rpb > 0) {
        const long rows = qxc ? s.Kc : s.Nj, cols = qxc ? s.Nj : s.Kc;
        BQ_REQUIRE(rows % s.q_rpb == 0 && s.q_bstride % 8 == 0 && (long)s.q_bstride >= (long)s.q_rpb * s.ldq, BQ_EINVAL,
                   "bq_gemm_bf16: q map: rows %ld must be whole batches of q_rpb = %d rows, q_bstride %% 8 == 0 and >= q_rpb * ldq",
                   rows, s.q_rpb);
        BQ_REQUIRE(tile != 256 || (qxc && s.q_rpb >= 64), BQ_EINVAL,
                   "bq_gemm_bf16: tile 256 maps only the contraction rows of a contraction-major Q, q_rpb >= 64");
        BQ_REQUIRE(tile != 128 || !qxc, BQ_EINVAL, "bq_gemm_bf16: tile 128 has no map on a contraction-major Q");
        q_span = (rows / s.q_rpb - 1) * (long)s.q_bstride + (long)(s.q_rpb - 1) * s.ldq + cols;
        qb = q_span * 2;   // rows past the last batch land beyond this bound and read zeros
      }
      if (s.o_rpb > 0) {
        BQ_REQUIRE(s.Nj % s.o_rpb == 0 && s.o_bstride % 8 == 0 && (long)s.o_bstride >= (long)s.o_rpb * s.ldo, BQ_EINVAL,
                   "bq_gemm_bf16: o map: Nj = %d must be whole batches of o_rpb = %d rows, o_bstride %% 8 == 0 and >= o_rpb * ldo",
                   s.Nj, s.o_rpb);
        BQ_REQUIRE(tile != 256 && epilogue != EPI_BIAS_CE && s.colsum == nullptr, BQ_EINVAL,
                   "bq_gemm_bf16: an output map needs tile 128, 64 or 32 and no column sums");
        o_span = (long)(s.Nj / s.o_rpb) * s.o_bstride;
        BQ_REQUIRE((o_span + 256L * s.ldo) * (f32 ? 4 : 2) < 0x7FFFFFFFL, BQ_EINVAL, "bq_gemm_bf16: mapped out larger than 2 GB");
      }
      BQ_REQUIRE(!s.accum || (f32 && tile != 256 && tile != 128), BQ_EINVAL, "bq_gemm_bf16: accum needs fp32 out and tile 64 / 32");
      // operand bytes (plus the rows of a ragged edge tile) must stay below the out-of-range sentinel of the DMA offsets
      BQ_REQUIRE(pb + (long)ti * s.ldp * 2 < 0x7FFFFFFFL && qb + (long)tj * s.ldq * 2 + (s.q_rpb > 0 ? 2L * s.q_bstride * 2 : 0) < 0x7FFFFFFFL,
                 BQ_EINVAL, "bq_gemm_bf16: operand larger than 2 GB (problem %d)", done);
      GemmProblem &g = ga.p[ga.n];
      g.P = (const __bf16 *)s.P; g.Q = (const __bf16 *)s.Q; g.out = s.out; g.bias = s.bias; g.out2 = s.out2;
      g.aux = (const __bf16 *)s.aux; g.colsum = s.colsum;
      g.ldp = s.ldp; g.ldq = s.ldq; g.ldo = s.ldo; g.Ni = s.Ni; g.Nj = s.Nj; g.Kc = s.Kc;
      g.flags = (s.bias_bf16 != 0 ? 1 : 0) | (s.accum != 0 ? 2 : 0);
      g.rpb_pack = (int)((unsigned)s.q_rpb | ((unsigned)s.o_rpb << 16));
      g.q_bstride = s.q_bstride; g.o_bstride = s.o_bstride;
      g.p_bytes = (unsigned)(s.p_bytes > 0 ? s.p_bytes : pb);
      g.q_bytes = (unsigned)((s.q_bytes > 0 && s.q_rpb == 0) ? s.q_bytes : qb);
      int ks_ = 1;
      if (epilogue == EPI_BIAS_CE) {
        BQ_REQUIRE(s.out2 && s.aux && s.colsum && s.ksplit > 0 && s.ksplit <= s.Ni && s.ksplit <= 65535 && tile == 256, BQ_EINVAL,
                   "bq_gemm_bf16: the cross-entropy epilogue needs out2 (partials), aux (targets), colsum (target logits), "
                   "ksplit = valid vocabulary entries (<= 65535), tile 256");
        ks_ = s.ksplit;
      } else {
        BQ_REQUIRE(s.ksplit <= 1 || (f32 && tile != 256), BQ_EINVAL, "bq_gemm_bf16: ksplit needs fp32 out and tile 64");
        BQ_REQUIRE(s.ksplit <= 65535, BQ_ELIMIT, "bq_gemm_bf16: ksplit = %d > 65535", s.ksplit);
        ks_ = (f32 && tile != 256 && s.ksplit > 1) ? s.ksplit : 1;
      }
      BQ_REQUIRE((s.Ni + ti - 1) / ti <= 65535, BQ_ELIMIT, "bq_gemm_bf16: Ni = %d too wide", s.Ni);
      g.tiles_ks = (int)((unsigned)((s.Ni + ti - 1) / ti) | ((unsigned)ks_ << 16));
      g.tile0 = ga.total_tiles;
      ga.total_tiles += g.tiles_i() * ((s.Nj + tj - 1) / tj) * (epilogue == EPI_BIAS_CE ? 1 : g.ksplit());
      ++ga.n;
      ++done;
    }
    BQ_REQUIRE(launch_gemm(ga, flags, epilogue, tile, st) == 0, BQ_EINVAL,
               "bq_gemm_bf16: unsupported combination flags=%d epilogue=%d tile=%d", flags, epilogue, tile);
    int rc = check_launch("gemm_bf16");
    if (rc) return rc;
  }
  return 0;
}

extern "C" int bq_colsum_grouped_bf16(const bq_colsum_desc *d, int n, void *stream) {
  using namespace bq;
  BQ_REQUIRE(d != nullptr && n >= 1, BQ_EINVAL, "bq_colsum_grouped_bf16: no problems");
  hipStream_t st = (hipStream_t)stream;
  int done = 0;
  while (done < n) {
    ColsumArgs ca;
    ca.n = 0;
    int wgs = 0;
    while (done < n && ca.n < COLSUM_MAX_PROBLEMS) {
      const bq_colsum_desc &s = d[done];
      BQ_REQUIRE(s.g && s.out && s.M > 0 && s.N > 0, BQ_EINVAL, "bq_colsum_grouped_bf16: bad problem %d", done);
      BQ_REQUIRE(s.N % 8 == 0 && s.ld % 8 == 0 && ((uintptr_t)s.g % 16 == 0), BQ_EINVAL,
                 "bq_colsum_grouped_bf16: N and ld must be multiples of 8, g 16-byte aligned");
      ColsumProblem &c = ca.p[ca.n];
      c.g = (const __bf16 *)s.g; c.out = s.out; c.M = s.M; c.N = s.N; c.ld = s.ld;
      c.cblocks = (s.N + 255) / 256;
      c.wg0 = wgs;
      wgs += c.cblocks * ((s.M + COLSUM_ROWS - 1) / COLSUM_ROWS);
      ++ca.n;
      ++done;
    }
    hipLaunchKernelGGL(colsum_grouped_kernel, dim3(wgs), dim3(256), 0, st, ca);
    int rc = check_launch("colsum_grouped");
    if (rc) return rc;
  }
  return 0;
}

// ======================================================================================================================
// wgrad_rows_kernel: dW (Nj x Ni, fp32) = Q^T P over the R rows, for the detector's SharedMLP layers -- R in the millions,
// Ni / Nj <= 256 channels.  The 64-tile kernel above cuts this into (tile, piece) workgroups that each end in 4096
// scattered fp32 atomics (~50 G atomics/s: THE cost of that form -- its HBM traffic is fine, with the XCD-aware piece
// order the tiles of a piece meet in the L2 and PMC shows every operand row fetched once, profiles/r02_det_wgrad_pmc.txt).
// Here a workgroup stages WHOLE rows -- TI 64-column units of P and TJ of Q per K tile, the gemm64 LDS images and
// transposing fragment reads -- keeps all TI x TJ output tiles in its accumulators and walks a contiguous run of K tiles;
// every workgroup stores its product into its own slice and wgrad_rows_reduce_kernel sums the slices (no atomics, fixed
// order).  One workgroup per CU (3 stages x (TI + TJ) x 8 KB).  HBM roofline: R (Ni + Nj) 2 bytes per launch (PMC:
// 839.0 MB for 838.9 MB of operands at SA1's first layer, 4.7 TB/s).
// ======================================================================================================================
namespace bq {

struct WgradRowsArgs {
  const __bf16 *P, *Q;
  float *part;   // (workgroups, Nj, ldo) partial products, one slice per workgroup
  float *out;    // (Nj, ldo): written by wgrad_rows_reduce_kernel
  int R, Ni, Nj, ldp, ldq, ldo, pieces;
  unsigned p_bytes, q_bytes;
};

template <int TI, int TJ>
__global__ __launch_bounds__(256) void wgrad_rows_kernel(const WgradRowsArgs ar) {
  // three LDS stages while they fit (<= 6 units), two for 7 units (264 -> 128 channels: SA3 / SA4's first layers)
  constexpr int UNITS = TI + TJ, STAGE = UNITS * 8192, NS = UNITS <= 6 ? 3 : 2, NDMA = 2 * UNITS;
  static_assert(NS >= 2 && NS * STAGE <= 160 * 1024 && (NS - 2) * NDMA <= 63, "LDS stages / counted waits");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nkt_all = (ar.R + 63) >> 6;
  const int kt_per = (nkt_all + (int)gridDim.x - 1) / (int)gridDim.x;
  const int kt0 = (int)blockIdx.x * kt_per;
  const int nkt = max(0, min(kt_per, nkt_all - kt0));   // (0: this workgroup's slice is all zeros)

  // (the LDS-DMAs are inline asm -- gemm_common.h, lds_dma16: hipcc fenced the transposed reads of every step against the builtin)
  const i32x4_rs rsP = raw_rsrc_v4(ar.P, ar.p_bytes), rsQ = raw_rsrc_v4(ar.Q, ar.q_bytes);
  const int cp = lane & 7;
  unsigned vp[2], vq[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int ur = (wave * 2 + d) * 8 + (lane >> 3);
    vp[d] = (unsigned)((ur * ar.ldp + (cp ^ (xg(ur) << 1)) * 8) * 2) + (unsigned)kt0 * (unsigned)(64 * ar.ldp * 2);
    vq[d] = (unsigned)((ur * ar.ldq + (cp ^ (xg(ur) << 1)) * 8) * 2) + (unsigned)kt0 * (unsigned)(64 * ar.ldq * 2);
  }
  const unsigned p_step = (unsigned)(64 * ar.ldp * 2), q_step = (unsigned)(64 * ar.ldq * 2);

  auto stage = [&](int step) {
    const bool live = step < nkt;
    const unsigned base = (unsigned)((step % NS) * STAGE);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
      for (int u = 0; u < TI; ++u)
        lds_dma16(rsP, smem + base + u * 8192 + (wave * 2 + d) * 1024, live ? vp[d] + u * 128 : 0x80000000u);
#pragma unroll
      for (int v = 0; v < TJ; ++v)
        lds_dma16(rsQ, smem + base + (TI + v) * 8192 + (wave * 2 + d) * 1024, live ? vq[d] + v * 128 : 0x80000000u);
      vp[d] += p_step;
      vq[d] += q_step;
    }
  };

  const int row16 = lane & 15, q4 = lane >> 4;
  const int kc_base = 0;  // (K-contiguous form unused here)
  // (closed-form fragment address: a table indexed by the wave row / column would be promoted to LDS, gemm_common.h)
  const int xc_q = (lane & 15) >> 2;
  const int xcg = (xc_q >> 1) | ((q4 & 1) << 1), xc0 = (8 * q4 + xc_q) * 128 + 8 * (lane & 3);

  f32x4 acc[TI][TJ][2][2];
#pragma unroll
  for (int u = 0; u < TI; ++u)
#pragma unroll
    for (int v = 0; v < TJ; ++v)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[u][v][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int p = 0; p < NS - 1; ++p) stage(p);
  for (int step = 0; step < nkt; ++step) {
    wait_vmcnt<(NS - 2) * NDMA>();
    BQ_BARRIER();
    stage(step + NS - 1);
    const unsigned char *buf = smem + (step % NS) * STAGE;
#pragma unroll
    for (int v = 0; v < TJ; ++v) {
      bf16x8 fb[2][2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb[b][kk] = read_frag_cf<true>(buf + (TI + v) * 8192, wc * 2 + b, kk, kc_base, xc0, xcg);
#pragma unroll
      for (int u = 0; u < TI; ++u) {
        bf16x8 fa[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) fa[a][kk] = read_frag_cf<true>(buf + u * 8192, wr * 2 + a, kk, kc_base, xc0, xcg);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc[u][v][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][kk], fb[b][kk], acc[u][v][a][b], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  float *slice = ar.part + (long)blockIdx.x * ar.Nj * ar.ldo;
#pragma unroll
  for (int u = 0; u < TI; ++u)
#pragma unroll
    for (int v = 0; v < TJ; ++v)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int i = u * 64 + wr * 32 + a * 16 + q4 * 4;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int j = v * 64 + wc * 32 + b * 16 + row16;
          if (i < ar.Ni && j < ar.Nj)
            *reinterpret_cast<float4 *>(slice + (long)j * ar.ldo + i) =
                make_float4(acc[u][v][a][b][0], acc[u][v][a][b][1], acc[u][v][a][b][2], acc[u][v][a][b][3]);
        }
      }
}

// out = sum over the workgroups' slices (plain stores: no atomics -- 4096 scattered fp32 atomics per output tile and
// workgroup were the whole cost of the cut contraction, ~50 G atomics/s -- and a fixed summation order).  A block owns 16
// consecutive float4 columns; its 16 lane groups take the slices p = g, g + 16, ... (8 loads in flight each: the kernel
// is a latency chain otherwise, 31 us for 9 MB with one thread per column) and meet in LDS.
__global__ __launch_bounds__(256) void wgrad_rows_reduce_kernel(const WgradRowsArgs ar) {
  __shared__ float4 red[16][16];
  const int n4 = ar.Nj * ar.ldo / 4;
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + c;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (e < n4) {
    const float4 *src = reinterpret_cast<const float4 *>(ar.part) + e;
    int p = g;
    for (; p + 112 < ar.pieces; p += 128) {
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = src[(long)(p + 16 * k) * n4];
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        s0.x += v[k].x; s0.y += v[k].y; s0.z += v[k].z; s0.w += v[k].w;
        s1.x += v[k + 1].x; s1.y += v[k + 1].y; s1.z += v[k + 1].z; s1.w += v[k + 1].w;
      }
    }
    for (; p < ar.pieces; p += 16) {
      const float4 a = src[(long)p * n4];
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
  }
  red[g][c] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
  __syncthreads();
  if (g == 0 && e < n4) {
    float4 r = red[0][c];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 a = red[k][c];
      r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
    }
    if ((e * 4) % ar.ldo >= ar.Ni) r = make_float4(0.f, 0.f, 0.f, 0.f);   // padding columns of a slice are never written
    reinterpret_cast<float4 *>(ar.out)[e] = r;
  }
}

template <int TI>
static int launch_wgrad_rows_j(int tj, int wgs, hipStream_t st, const WgradRowsArgs &a) {
  if constexpr (TI + 1 <= 6)
    if (tj == 1) { hipLaunchKernelGGL((wgrad_rows_kernel<TI, 1>), dim3(wgs), dim3(256), 0, st, a); return 0; }
  if constexpr (TI + 2 <= 7)
    if (tj == 2) { hipLaunchKernelGGL((wgrad_rows_kernel<TI, 2>), dim3(wgs), dim3(256), 0, st, a); return 0; }
  if constexpr (TI + 4 <= 7)
    if (tj == 4) { hipLaunchKernelGGL((wgrad_rows_kernel<TI, 4>), dim3(wgs), dim3(256), 0, st, a); return 0; }
  return -1;
}

}  // namespace bq

extern "C" int bq_wgrad_rows_supported(int Ni, int Nj) {
  const int ti = (Ni + 63) / 64, tj = (Nj + 63) / 64;
  return ti >= 1 && ti <= 5 && (tj == 1 || tj == 2 || tj == 4) && ti + tj <= 7 && !(ti == 5 && tj == 1);
}

extern "C" int bq_wgrad_rows_workgroups(long R, int Ni, int Nj, int workgroups) {
  const long nkt = (R + 63) / 64;
  // default: one workgroup per CU; two where two fit the LDS (<= 3 units: 72 KB each), measured tools/bench_det_wgrad.py
  long wgs = workgroups > 0 ? workgroups : (((Ni + 63) / 64 + (Nj + 63) / 64 <= 3) ? 512 : 256);
  if (wgs > nkt) wgs = nkt;
  return (int)(wgs < 1 ? 1 : wgs);
}

extern "C" int bq_wgrad_rows_bf16(const void *P, const void *Q, float *out, float *part, long R, int Ni, int Nj, int ldp,
                                  int ldq, int ldo, int workgroups, void *stream) {
  using namespace bq;
  BQ_REQUIRE(P && Q && out && part && R > 0, BQ_EINVAL, "bq_wgrad_rows_bf16: null pointer / no rows");
  BQ_REQUIRE(ldo % 4 == 0 && ((uintptr_t)part % 16 == 0), BQ_EINVAL, "bq_wgrad_rows_bf16: ldo %% 4, part 16-byte aligned");
  BQ_REQUIRE(bq_wgrad_rows_supported(Ni, Nj), BQ_EINVAL, "bq_wgrad_rows_bf16: %d x %d channels not supported", Ni, Nj);
  BQ_REQUIRE(Ni % 4 == 0 && ldp % 8 == 0 && ldq % 8 == 0 && ldp >= Ni && ldq >= Nj && ldo >= Ni, BQ_EINVAL,
             "bq_wgrad_rows_bf16: Ni %% 4, ld %% 8, ld >= channels (Ni=%d Nj=%d ldp=%d ldq=%d ldo=%d)", Ni, Nj, ldp, ldq, ldo);
  BQ_REQUIRE(((uintptr_t)P % 16 == 0) && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)out % 16 == 0), BQ_EINVAL,
             "bq_wgrad_rows_bf16: operands must be 16-byte aligned");
  BQ_REQUIRE(R * (long)ldp * 2 < 0x7FFFFFFFL - 64L * ldp * 2 - 1024 && R * (long)ldq * 2 < 0x7FFFFFFFL - 64L * ldq * 2 - 1024,
             BQ_ELIMIT, "bq_wgrad_rows_bf16: an operand larger than 2 GB");
  WgradRowsArgs a;
  a.P = (const __bf16 *)P; a.Q = (const __bf16 *)Q; a.out = out; a.part = part;
  a.R = (int)R; a.Ni = Ni; a.Nj = Nj; a.ldp = ldp; a.ldq = ldq; a.ldo = ldo;
  a.p_bytes = (unsigned)(R * (long)ldp * 2);
  a.q_bytes = (unsigned)(R * (long)ldq * 2);
  const int wgs = bq_wgrad_rows_workgroups(R, Ni, Nj, workgroups);
  a.pieces = wgs;
  const int ti = (Ni + 63) / 64, tj = (Nj + 63) / 64;
  hipStream_t st = (hipStream_t)stream;
  int rc = -1;
  if (ti == 1) rc = launch_wgrad_rows_j<1>(tj, wgs, st, a);
  else if (ti == 2) rc = launch_wgrad_rows_j<2>(tj, wgs, st, a);
  else if (ti == 3) rc = launch_wgrad_rows_j<3>(tj, wgs, st, a);
  else if (ti == 4) rc = launch_wgrad_rows_j<4>(tj, wgs, st, a);
  else if (ti == 5 && tj == 2) { hipLaunchKernelGGL((wgrad_rows_kernel<5, 2>), dim3(wgs), dim3(256), 0, st, a); rc = 0; }
  BQ_REQUIRE(rc == 0, BQ_EINVAL, "bq_wgrad_rows_bf16: no kernel for %d x %d units", ti, tj);
  hipLaunchKernelGGL(wgrad_rows_reduce_kernel, dim3((Nj * ldo / 4 + 15) / 16), dim3(256), 0, st, a);
  return check_launch("wgrad_rows");
}

// out (Nj, ldo) = the sum of `pieces` slices of part, in slice order (library-internal: csrc/detbwd.hip)
extern "C" int bq_wgrad_rows_reduce(float *part, float *out, int Ni, int Nj, int ldo, int pieces, void *stream) {
  using namespace bq;
  WgradRowsArgs a{};
  a.part = part; a.out = out; a.Ni = Ni; a.Nj = Nj; a.ldo = ldo; a.pieces = pieces;
  hipLaunchKernelGGL(wgrad_rows_reduce_kernel, dim3((Nj * ldo / 4 + 15) / 16), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("wgrad_rows_reduce");
}

extern "C" int bq_pwconv_records(long R, int N) {
  // row walkers per 64-channel block: enough workgroups to fill the chip three times over, never more than the tiles
  const long tiles_j = (R + 63) / 64;
  const int tiles_i = (N + 63) / 64;
#ifndef BQ_PWCONV_WALKERS
#define BQ_PWCONV_WALKERS 1024   // (c2 step, two runs each: 512 9.04 ms, 768 8.93, 1024 8.89)
#endif
  long gj = (BQ_PWCONV_WALKERS + tiles_i - 1) / tiles_i;
  if (gj > tiles_j) gj = tiles_j;
  if (gj < 1) gj = 1;
  return (int)gj * 2;
}

extern "C" int bq_pwconv_bn_fwd(const void *x, long R, int K, int ldx, const void *w, int ldw, int Kc, int N, void *y,
                                float *partial, const float *gamma, const float *beta, float *running_mean,
                                float *running_var, long long *num_batches_tracked, float eps, float momentum,
                                float *scale, float *shift, float *mean, float *rstd, const float *center, float *shift_acc,
                                void *stream) {
  return bq_pwconv_bn_fwd_x(x, nullptr, nullptr, R, K, ldx, w, ldw, Kc, N, y, partial, gamma, beta, running_mean, running_var,
                            num_batches_tracked, eps, momentum, scale, shift, mean, rstd, center, shift_acc, stream);
}

extern "C" int bq_pwconv_bn_fwd_x(const void *x, const float *xscale, const float *xshift, long R, int K, int ldx, const void *w,
                                  int ldw, int Kc, int N, void *y, float *partial, const float *gamma, const float *beta,
                                  float *running_mean, float *running_var, long long *num_batches_tracked, float eps,
                                  float momentum, float *scale, float *shift, float *mean, float *rstd, const float *center,
                                  float *shift_acc, void *stream) {
  using namespace bq;
  BQ_REQUIRE(x && w && y && partial && gamma && beta && scale && shift && mean && rstd, BQ_EINVAL, "pwconv_bn_fwd: null pointer");
  BQ_REQUIRE(R > 0 && K > 0 && N > 0, BQ_EINVAL, "pwconv_bn_fwd: empty problem");
  BQ_REQUIRE(N % 64 == 0 && Kc % 64 == 0 && Kc >= K && ldw >= Kc && ldx >= K && ldx % 8 == 0 && ldw % 8 == 0, BQ_EINVAL,
             "pwconv_bn_fwd: need N %% 64 == 0, Kc %% 64 == 0 >= K, ldw >= Kc, ldx >= K, ld %% 8 == 0 (N=%d K=%d Kc=%d ldx=%d ldw=%d)",
             N, K, Kc, ldx, ldw);
  BQ_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)y % 16 == 0), BQ_EINVAL,
             "pwconv_bn_fwd: operands must be 16-byte aligned");
  BQ_REQUIRE(R * (long)ldx * 2 < 0x7FFFFFFFL - 64L * ldx * 2, BQ_ELIMIT, "pwconv_bn_fwd: x larger than 2 GB");
  PwconvArgs a;
  a.W = (const __bf16 *)w; a.X = (const __bf16 *)x; a.Y = (__bf16 *)y; a.partial = partial; a.center = center;
  a.ldw = ldw; a.ldx = ldx; a.Ni = N; a.R = (int)R; a.Kc = Kc;
  a.tiles_i = N / 64;
  a.Gj = bq_pwconv_records(R, N) / 2;
  a.w_bytes = (unsigned)((long)N * ldw * 2);
  a.x_bytes = (unsigned)(R * (long)ldx * 2);
  hipStream_t st = (hipStream_t)stream;
#ifndef BQ_PWCONV_STREAM
#define BQ_PWCONV_STREAM 1
#endif
  const int nkt = Kc / 64;
  const dim3 grid(a.tiles_i * a.Gj);
  a.xscale = xscale; a.xshift = xshift;
  if (xscale) {
    BQ_REQUIRE(xshift && K == Kc && ldx == K && (nkt == 1 || nkt == 2), BQ_EINVAL,
               "pwconv_bn_fwd_x: a deferred input needs K = ldx = 64 or 128 (K=%d ldx=%d)", K, ldx);
    if (nkt == 1) hipLaunchKernelGGL((pwconv64s_kernel<1, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((pwconv64s_kernel<2, true>), grid, dim3(256), 0, st, a);
  }
  else if (BQ_PWCONV_STREAM && nkt == 1) hipLaunchKernelGGL((pwconv64s_kernel<1, false>), grid, dim3(256), 0, st, a);
  else if (BQ_PWCONV_STREAM && nkt == 2) hipLaunchKernelGGL((pwconv64s_kernel<2, false>), grid, dim3(256), 0, st, a);
  else if (BQ_PWCONV_STREAM && nkt == 3) hipLaunchKernelGGL((pwconv64s_kernel<3, false>), grid, dim3(256), 0, st, a);
  else if (BQ_PWCONV_STREAM && nkt == 4) hipLaunchKernelGGL((pwconv64s_kernel<4, false>), grid, dim3(256), 0, st, a);
  else if (BQ_PWCONV_STREAM && nkt == 5) hipLaunchKernelGGL((pwconv64s_kernel<5, false>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(pwconv64_kernel<0>, grid, dim3(256), 0, st, a);
  PwconvBnParams p{partial, gamma, beta, running_mean, running_var, num_batches_tracked, scale, shift, mean, rstd,
                   eps, momentum, N, a.Gj * 2, a.Gj, R, center, shift_acc};
  hipLaunchKernelGGL(pwconv_bn_finalize_kernel, dim3((N + PWF_CH - 1) / PWF_CH), dim3(1024), 0, st, p);
  return check_launch("pwconv_bn_fwd");
}

extern "C" int bq_pwconv_bn_apply(const void *x, long R, int K, int ldx, const void *w, int ldw, int Kc, int N, const float *scale,
                                  const float *shift_acc, void *out, int S, int relu, int pool, void *stream) {
  const float *shift = shift_acc, *center = nullptr;
  using namespace bq;
  BQ_REQUIRE(x && w && out && scale && shift, BQ_EINVAL, "pwconv_bn_apply: null pointer");
  BQ_REQUIRE(R > 0 && K > 0 && N > 0, BQ_EINVAL, "pwconv_bn_apply: empty problem");
  BQ_REQUIRE(N % 64 == 0 && Kc % 64 == 0 && Kc >= K && ldw >= Kc && ldx >= K && ldx % 8 == 0 && ldw % 8 == 0, BQ_EINVAL,
             "pwconv_bn_apply: need N %% 64 == 0, Kc %% 64 == 0 >= K, ldw >= Kc, ldx >= K, ld %% 8 == 0");
  BQ_REQUIRE(!pool || ((S == 16 || S == 32 || S == 64) && R % S == 0), BQ_ELIMIT, "pwconv_bn_apply: pooling over S = %d rows", S);
  BQ_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)out % 16 == 0), BQ_EINVAL,
             "pwconv_bn_apply: operands must be 16-byte aligned");
  BQ_REQUIRE(R * (long)ldx * 2 < 0x7FFFFFFFL - 64L * ldx * 2, BQ_ELIMIT, "pwconv_bn_apply: x larger than 2 GB");
  PwconvArgs a;
  a.W = (const __bf16 *)w; a.X = (const __bf16 *)x; a.Y = (__bf16 *)out; a.partial = nullptr; a.center = center;
  a.ldw = ldw; a.ldx = ldx; a.Ni = N; a.R = (int)R; a.Kc = Kc;
  a.tiles_i = N / 64;
  a.Gj = bq_pwconv_records(R, N) / 2;
  a.w_bytes = (unsigned)((long)N * ldw * 2);
  a.x_bytes = (unsigned)(R * (long)ldx * 2);
  a.scale = scale; a.shift = shift; a.relu = relu; a.S = S;
  hipStream_t st = (hipStream_t)stream;
  if (pool) hipLaunchKernelGGL(pwconv64_kernel<2>, dim3(a.tiles_i * a.Gj), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(pwconv64_kernel<1>, dim3(a.tiles_i * a.Gj), dim3(256), 0, st, a);
  return check_launch("pwconv_bn_apply");
}
