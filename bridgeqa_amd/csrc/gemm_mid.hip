// gemm128_kernel: the forward / input-gradient GEMMs of the image encoder and of the K/V projections over the image
// tokens (reference models/vit.py:30-32,51-53, models/med.py:112-118: nn.Linear + bias / GELU, and their autograd dX) on a
// 256 (i) x 128 (j) output tile with TWO workgroups co-resident per CU.
//
// Why a second tile shape (profiles/r02_gemm_pmc.txt): the 256 x 256 kernel of csrc/gemm.hip keeps the
// matrix pipe 58 % busy inside its K loop, but these contractions are SHORT (K = 768: 12 K tiles, 18 us) against a
// prologue (first operands from HBM) and an epilogue (128 KB of output per workgroup, written by all 256 workgroups of a
// round at once) of ~9 us per round, and 16400 rows are 64.06 row tiles: whole launches end at 19-33 % MFMA busy.  One
// workgroup per CU cannot hide its own epilogue (the persistent form was built and measured slower, §4.4).  Here a
// workgroup is 4 waves (one per SIMD, <= 256 VGPRs) with 64 KB of LDS, so a CU holds two INDEPENDENT workgroups: one's
// prologue, epilogue stores and barrier / LDS-read gaps run under the other's MFMA clusters, the tiles are half as long
// (129 x 3k tiles over 512 slots: the tail is one small tile, and a CU left with one workgroup runs it at full rate), and
// the workgroups of a CU drift apart by themselves (the older one wins the issue arbitration), so the output stores of
// the chip are spread over the launch instead of arriving in lock-step bursts.
//
// Same formulation, LDS images and fragment maps as gemm256_kernel (gemm_common.h):  out[j][i] = sum_kc P(i,kc) Q(j,kc),
// P K-contiguous (forward: W) or contraction-major (dX: W read transposed with ds_read_b64_tr_b16), Q K-contiguous.
// Wave tile 128 (i) x 64 (j) = gemm256's; waves 2 (i) x 2 (j).  Staging units of 64 rows x 64 k (8 KB):
//   PA0(g) PA1(g)  the two 64-row halves of wave row g's P rows          -- SINGLE-buffered (weights: L2-resident)
//   QB0[b] QB1[b]  the first / second 32 rows of both wave columns       -- DOUBLE-buffered (b = K tile & 1: the
//                                                                            activation stream comes from HBM / MALL)
// = 64 KB.  Per K tile four phases, ONE barrier each:
//   phase   waits for (counted)         reads (LDS -> registers)   MFMA quadrant        restages (LDS-DMA, per wave)
//   p0(t)   vmcnt(6): PA0(t), QB0(t)    PA0, QB0                   A0 x B0              QB1(t+1)  x2   [read next in p1(t+1)]
//   p1(t)   --                          QB1                        A0 x B1              PA0(t+1)  x4   [p0(t+1): 3 phases]
//   p2(t)   vmcnt(6): PA1(t)            PA1                        A1 x B1              QB0(t+2)  x2   [p0(t+2): 6 phases]
//   p3(t)   --                          (B0 kept in registers)     A1 x B0              PA1(t+1)  x4   [p2(t+1): 3 phases]
// A unit is restaged in the phase AFTER its last read (that phase's barrier orders every wave's reads before the DMA);
// vmcnt counts a wave's DMAs in issue order, so vmcnt(6) at p0(t) = "everything up to PA0(t) has landed" (issued after
// it: QB0(t+1) x2, PA1(t) x4) and at p2(t) = "up to PA1(t)" (after it: QB1(t+1) x2, PA0(t+1) x4); the barrier that follows
// makes every wave's DMAs visible to all.  DMAs past the last K tile use the out-of-range sentinel (zeros, no traffic)
// so that the counts keep their meaning.
#include "gemm_common.h"

namespace bq {

template <bool P_XC, int EPI>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(const GemmArgs args) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // ---- workgroup -> (problem, tile), XCD-aware as in gemm256_kernel ---------------------------------------------------
  const int nwg = args.total_tiles;
  int t;
  {
    const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
    t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  int pi = 0;
  for (int k = 1; k < args.n; ++k)
    if (t >= args.p[k].tile0) pi = k;
  const GemmProblem &pr = args.p[pi];
  const int tl = t - pr.tile0;
  const int tiles_j = (pr.Nj + 127) >> 7;
  const int bj = tiles_j - 1 - tl / pr.tiles_i, bi = tl % pr.tiles_i;   // ragged last j block first
  const int i0 = bi * 256, j0 = bj * 128;
  const int Ni = pr.Ni, Nj = pr.Nj, Kc = pr.Kc;
  const int ldp = pr.ldp, ldq = pr.ldq;
  const int nkt = (Kc + 63) >> 6;

  // ---- staging: a unit = 8 DMAs of 1 KB (8 rows x 128 B), two per wave: unit rows (w + 4 d) * 8 + lane / 8 ------------
  const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void *)pr.P, 0, pr.p_bytes, 0x00020000);
  const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void *)pr.Q, 0, pr.q_bytes, 0x00020000);
  const int cp = lane & 7;
  unsigned vP[4][2], vQ[2][2];   // P units PA0(0) PA0(1) PA1(0) PA1(1); Q units QB0 QB1
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int ur = (wave + 4 * d) * 8 + (lane >> 3);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = u & 1, half = u >> 1;
      if (!P_XC) vP[u][d] = (unsigned)(((i0 + g * 128 + half * 64 + ur) * ldp + (cp ^ (ur & 7)) * 8) * 2);
      else vP[u][d] = (unsigned)((ur * ldp + i0 + g * 128 + half * 64 + (cp ^ (xg(ur) << 1)) * 8) * 2);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half)  // unit row r -> j0 + (r >> 5) * 64 + half * 32 + (r & 31)
      vQ[half][d] = (unsigned)(((j0 + (ur >> 5) * 64 + half * 32 + (ur & 31)) * ldq + (cp ^ (ur & 7)) * 8) * 2);
  }
  const unsigned p_step = P_XC ? (unsigned)(64 * ldp * 2) : 128u;
  constexpr unsigned q_step = 128u;
  constexpr int PA0 = 0, PA1 = 16384, QB = 32768;   // QB + buf * 16384 + half * 8192

  // stage P unit pair `half` (PA0 / PA1: both wave rows, 4 DMAs per wave) of K tile kt
  auto stage_p = [&](int half, int kt) {
    const bool live = kt < nkt;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const int u = half * 2 + g;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_void_t *)(smem + (half ? PA1 : PA0) + g * 8192 + (wave + 4 * d) * 1024),
                                                 16, live ? vP[u][d] : 0x80000000u, 0, 0, 0);
        vP[u][d] += p_step;
      }
  };
  // stage Q unit `half` (QB0 / QB1, 2 DMAs per wave) of K tile kt into buffer kt & 1
  auto stage_q = [&](int half, int kt) {
    const bool live = kt < nkt;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_void_t *)(smem + QB + (kt & 1) * 16384 + half * 8192 + (wave + 4 * d) * 1024),
                                               16, live ? vQ[half][d] : 0x80000000u, 0, 0, 0);
      vQ[half][d] += q_step;
    }
  };

  // ---- fragment read addresses (lane-dependent parts), as in gemm256_kernel -----------------------------------------
  const int row16 = lane & 15, q4 = lane >> 4;
  const int kc_base = row16 * 128 + ((q4 ^ (row16 & 7)) << 4);
  int xc_base[4];
  {
    const int q = (lane & 15) >> 2, p = lane & 3, g = (q >> 1) | ((q4 & 1) << 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) xc_base[s] = (8 * q4 + q) * 128 + ((s ^ g) << 5) + 8 * p;
  }
  const int uA0 = PA0 + wr * 8192, uA1 = PA1 + wr * 8192;
  const int bsub = wc * 2;

  const int iw = i0 + wr * 128, jw = j0 + wc * 64;
  const bool vA0 = iw < Ni, vA1 = iw + 64 < Ni, vB0 = jw < Nj, vB1 = jw + 32 < Nj;

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: the issue order the steady state would have produced before p0(0) ---------------------------------
  stage_q(0, 0); stage_q(1, 0); stage_p(0, 0); stage_q(0, 1); stage_p(1, 0);

  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
#define BQ_MID_MFMA(AO, FB, BO)                                                                       \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int a = 0; a < 4; ++a)      \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) acc[AO + a][BO + b] =                             \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][kk], FB[b][kk], acc[AO + a][BO + b], 0, 0, 0);
#define BQ_MID_COMPUTE_BEGIN()                              \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
  __builtin_amdgcn_sched_barrier(0);                        \
  __builtin_amdgcn_s_setprio(1);
#define BQ_MID_COMPUTE_END()                                \
  __builtin_amdgcn_s_setprio(0);                            \
  __builtin_amdgcn_sched_barrier(0);

  for (int kt = 0; kt < nkt; ++kt) {
    const unsigned char *qb = smem + QB + (kt & 1) * 16384;
    // ---- p0: PA0, QB0 -> A0 x B0 ; restage QB1(t+1)
    wait_vmcnt<6>();
    BQ_BARRIER();
    if (vA0) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[a][kk] = read_frag<P_XC>(smem + uA0, a, kk, kc_base, xc_base);
    }
    if (vB0) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb0[b][kk] = read_frag<false>(qb, bsub + b, kk, kc_base, xc_base);
    }
    stage_q(1, kt + 1);
    BQ_MID_COMPUTE_BEGIN();
    if (vA0 && vB0) { BQ_MID_MFMA(0, fb0, 0) }
    BQ_MID_COMPUTE_END();
    // ---- p1: QB1 -> A0 x B1 ; restage PA0(t+1)
    BQ_BARRIER();
    if (vB1) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb1[b][kk] = read_frag<false>(qb + 8192, bsub + b, kk, kc_base, xc_base);
    }
    stage_p(0, kt + 1);
    BQ_MID_COMPUTE_BEGIN();
    if (vA0 && vB1) { BQ_MID_MFMA(0, fb1, 2) }
    BQ_MID_COMPUTE_END();
    // ---- p2: PA1 -> A1 x B1 ; restage QB0(t+2)
    wait_vmcnt<6>();
    BQ_BARRIER();
    if (vA1) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[a][kk] = read_frag<P_XC>(smem + uA1, a, kk, kc_base, xc_base);
    }
    stage_q(0, kt + 2);
    BQ_MID_COMPUTE_BEGIN();
    if (vA1 && vB1) { BQ_MID_MFMA(4, fb1, 2) }
    BQ_MID_COMPUTE_END();
    // ---- p3: (B0 kept in registers) -> A1 x B0 ; restage PA1(t+1)
    BQ_BARRIER();
    stage_p(1, kt + 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (vA1 && vB0) { BQ_MID_MFMA(4, fb0, 0) }
    BQ_MID_COMPUTE_END();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing out-of-range DMAs have written their zeros
  BQ_BARRIER();

  // ---- epilogue (gemm256_kernel's, without the LDS table: GELU / GELU' are evaluated -- the co-resident workgroup's
  // MFMAs run meanwhile).  accumulator (a, b)[r]: i = iw + a*16 + q4*4 + r, j = jw + b*16 + row16.  bf16 outputs go through
  // a wave-private [64 j][128 i] LDS image (16 KB, chunk c of row j at j*256 + ((c ^ (j & 15)) << 4)) so that global stores
  // are whole 256-B row pieces.
  const int ldo = pr.ldo;
  unsigned char *ep = smem + wave * 16384;
  constexpr int NPASS = (EPI == EPI_BIAS_GELU) ? 2 : 1;
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    if (EPI == EPI_DGELU || EPI == EPI_ADD) {
      // second operand `aux` in the accumulator's own map (8 B per lane), j block outermost, the next block's eight
      // pieces in flight while one is processed; bounds-checked (rows past Nj read zeros)
      const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void *)pr.aux, 0, (int)((long)Nj * ldo * 2), 0x00020000);
      uint2 yy[2][8];
      auto fetch = [&](int b, uint2(&dst)[8]) {
        const unsigned off = (unsigned)(((jw + b * 16 + row16) * ldo + iw + q4 * 4) * 2);
#pragma unroll
        for (int a = 0; a < 8; ++a)
          dst[a] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsX, off + a * 32, 0, 0));
      };
      fetch(0, yy[0]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (b + 1 < 4) fetch(b + 1, yy[(b + 1) & 1]);
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const uint2 yb = yy[b & 1][a];
          const float y0 = __uint_as_float(yb.x << 16), y1 = __uint_as_float(yb.x & 0xffff0000u);
          const float y2 = __uint_as_float(yb.y << 16), y3 = __uint_as_float(yb.y & 0xffff0000u);
          float v[4];
          if (EPI == EPI_DGELU) {  // out = acc * gelu'(y)
            v[0] = acc[a][b][0] * dgelu_f(y0); v[1] = acc[a][b][1] * dgelu_f(y1);
            v[2] = acc[a][b][2] * dgelu_f(y2); v[3] = acc[a][b][3] * dgelu_f(y3);
          } else {                 // out = acc + aux
            v[0] = acc[a][b][0] + y0; v[1] = acc[a][b][1] + y1; v[2] = acc[a][b][2] + y2; v[3] = acc[a][b][3] + y3;
          }
          uint2 pk;
          pk.x = pack_bf16x2(v[0], v[1]);
          pk.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2 *>(ep + (b * 16 + row16) * 256 + (((a * 2 + (q4 >> 1)) ^ row16) << 4) + (q4 & 1) * 8) = pk;
        }
      }
    } else {
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int i = iw + a * 16 + q4 * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if ((EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) && pr.bias != nullptr && i < Ni) load_bias4(pr, i, bv);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (float)(__bf16)(acc[a][b][r] + bv[r]);  // what is stored (and what a backward differentiates at)
          if (EPI == EPI_BIAS_GELU && pass == 1) {  // x * Phi(x), Phi of the bf16 value
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
          }
          uint2 pk;
          pk.x = pack_bf16x2(v[0], v[1]);
          pk.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2 *>(ep + (b * 16 + row16) * 256 + (((a * 2 + (q4 >> 1)) ^ row16) << 4) + (q4 & 1) * 8) = pk;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __bf16 *dst = (pass == 1) ? reinterpret_cast<__bf16 *>(pr.out2) : reinterpret_cast<__bf16 *>(pr.out);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int jr = it * 4 + q4;          // row of the wave image
      const int j = jw + jr, i = iw + row16 * 8;
      const uint4 v = *reinterpret_cast<const uint4 *>(ep + jr * 256 + ((row16 ^ (jr & 15)) << 4));
      if (j < Nj && i < Ni) *reinterpret_cast<uint4 *>(dst + (long)j * ldo + i) = v;  // Ni % 8 == 0 (host check)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

int launch_gemm_mid(const GemmArgs &ga, bool p_xc, int epi, hipStream_t st) {
  const dim3 grid(ga.total_tiles), block(256);
  if (!p_xc) {
    if (epi == EPI_NONE) { hipLaunchKernelGGL((gemm128_kernel<false, EPI_NONE>), grid, block, 0, st, ga); return 0; }
    if (epi == EPI_BIAS) { hipLaunchKernelGGL((gemm128_kernel<false, EPI_BIAS>), grid, block, 0, st, ga); return 0; }
    if (epi == EPI_BIAS_GELU) { hipLaunchKernelGGL((gemm128_kernel<false, EPI_BIAS_GELU>), grid, block, 0, st, ga); return 0; }
  } else {
    if (epi == EPI_NONE) { hipLaunchKernelGGL((gemm128_kernel<true, EPI_NONE>), grid, block, 0, st, ga); return 0; }
    if (epi == EPI_DGELU) { hipLaunchKernelGGL((gemm128_kernel<true, EPI_DGELU>), grid, block, 0, st, ga); return 0; }
    if (epi == EPI_ADD) { hipLaunchKernelGGL((gemm128_kernel<true, EPI_ADD>), grid, block, 0, st, ga); return 0; }
  }
  return -1;
}

}  // namespace bq
