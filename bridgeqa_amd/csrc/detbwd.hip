// Backward of one SharedMLP layer (1x1 convolution -> training-mode BatchNorm -> ReLU (-> max over nsample)) on point-major
// bf16 rows in ONE pass over the activations, gfx950 (reference lib/pointnet2/pytorch_utils.py:104-157,
// 11-36 and its autograd).
//
// Until round 5 the backward of a layer ran as four passes over (R x C)-sized arrays (R = B * npoint * nsample: 2.1 M rows
// at SA1): bn_bwd_reduce (reads dOut, P) -> bn_bwd_dx (reads dOut, P, writes dP) -> the dX GEMM (reads dP, writes dX) -> the
// dW kernel (reads dP, X) = nine array passes.  Here, after the reduction (the BatchNorm gradient needs its two per-channel
// sums first), sa_bwd_kernel walks the 64-row tiles once: X, P and dOut tiles arrive in LDS by DMA, every wave turns its own 16
// rows of P into dP IN PLACE (dP = scale * (g - dbeta / R - xhat * dgamma / R), g = dOut masked by ReLU, or routed to the
// arg-max row of its group when the layer ends in the max-pool), and the tile then feeds two products from the same LDS image:
//   dW (Nj x Ni, fp32) += dP^T X     contraction over the tile's rows   (transposing reads of both images, as wgrad_rows)
//   dX (64 x Ni, bf16)  = dP  W      contraction over the Nj channels   (plain reads of the dP image, transposing reads of W)
// -- the contraction-major swizzle of the images is conflict-free for both read patterns (tools/lds_bank_sim.py).  Six array
// passes instead of nine; for a pooled layer (dOut and the arg-max table are R / S rows) three instead of seven.
// Per-workgroup dW slices are summed by wgrad_rows_reduce_kernel (fixed order, no atomics).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "gemm_common.h"

#ifndef BQ_SA_PREFER_TWO
#define BQ_SA_PREFER_TWO 1
#endif

namespace bq {

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct SaBwdArgs {
  const __bf16 *X;       // [R][ldx] the layer's input rows (whole padded rows)
  const __bf16 *P;       // [R][Nj] the stored pre-activation
  const __bf16 *dOut;    // [R][Nj], pooled: [R / S][Nj]
  const unsigned char *arg;  // pooled: [R / S][Nj] row (0 .. S-1) of the group's first maximum
  const __bf16 *W;       // [Nj][ldw] zero beyond the input channels
  const float *scale, *shift, *mean, *rstd;   // of the STORED pre-activation
  const float *dgb;      // [2][Nj] dbeta | dgamma
  const float *xscale, *xshift;   // XT: X holds the previous layer's stored pre-activation; the layer's input is relu(X xscale + xshift)
  // RED (with XT): the PREVIOUS layer's BatchNorm reduction rides on this pass -- its dOut is this kernel's dX, its pre-activation
  // this kernel's x tile: per (workgroup, wave) partial sums of g and g xhat, g = dX masked by that layer's ReLU
  const float *xmean, *xrstd;
  float *red_part;       // [workgroups * 4][2][ldx]
  __bf16 *dX;            // [R][ldx] or null
  float *part;           // [workgroups][Nj][ldo]
  int R, ldx, Nj, ldw, ldo, S, relu;
  unsigned x_bytes, p_bytes, d_bytes, a_bytes, w_bytes, dx_bytes;
};

template <int TI, int TJ, bool POOL, bool DX, bool RED = false>
struct SaBwdCfg {
  static constexpr int RED_LDS = RED ? TI * 64 * 16 : 0;   // the previous layer's (scale, shift, mean, rstd) per channel
  static constexpr int UNITS = TI + TJ * (POOL ? 1 : 2);
  static constexpr int WB = DX ? TI * TJ * 8192 : 0;
  // pooled layers: a wave's dOut / arg-max rows (one group per wave and tile) travel by DMA into 2 KB of the stage while three
  // stages fit; with two stages they are loaded into registers instead (GREG: the loads of tile t + 1 are issued after the
  // transform of tile t has consumed the registers) and the stage is the images alone
  static constexpr int STAGE3 = UNITS * 8192 + (POOL ? 8192 : 0);
  // (BQ_SA_PREFER_TWO: two stages wherever that lets two workgroups share a CU's LDS -- measurement macro)
  static constexpr bool TWO = BQ_SA_PREFER_TWO && 2 * UNITS * 8192 + WB + RED_LDS <= 80 * 1024 && 3 * STAGE3 + WB + RED_LDS > 80 * 1024;
  static constexpr int NS = (3 * STAGE3 + WB + RED_LDS <= 160 * 1024 && !TWO) ? 3 : 2;
  static constexpr bool GREG = POOL && NS == 2;
  static constexpr int STAGE = NS == 3 ? STAGE3 : UNITS * 8192;
  static constexpr int LDS = NS * STAGE + WB + RED_LDS;
  static constexpr bool FITS = LDS <= 160 * 1024;
  static constexpr int NDMA = 2 * TI + 2 * TJ + (POOL ? (GREG ? 2 * TJ : 2) : 2 * TJ);   // vector-memory operations per wave and stage
  static constexpr int NST = DX ? 4 * TI : 0;                           // dX stores per wave and tile
  static constexpr int WAITN = (NS - 2) * (NDMA + NST) + NST;
};

template <int TI, int TJ, bool POOL, bool DX, bool XT = false, bool RED = false>
__global__ __launch_bounds__(256) void sa_bwd_kernel(const SaBwdArgs ar) {
  static_assert(!RED || (XT && DX), "the previous layer's reduction needs the deferred input and its gradient");
  using C = SaBwdCfg<TI, TJ, POOL, DX, RED>;
  constexpr int STAGE = C::STAGE, NS = C::NS, WB = C::WB;
  static_assert(C::FITS && C::WAITN <= 63, "LDS / counted waits");
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS];
  unsigned char *const wimg = smem;
  unsigned char *const stages = smem + WB;
  [[maybe_unused]] float4 *const s_prev = reinterpret_cast<float4 *>(smem + WB + NS * STAGE);   // RED: [channel] = (sc, sh, mean, rstd)
  const unsigned DEAD = 0x80000000u;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nkt_all = (ar.R + 63) >> 6;
  const int kt_per = (nkt_all + (int)gridDim.x - 1) / (int)gridDim.x;
  const int kt0 = (int)blockIdx.x * kt_per;
  const int nkt = max(0, min(kt_per, nkt_all - kt0));

  [[maybe_unused]] const auto rsD = __builtin_amdgcn_make_buffer_rsrc((void *)ar.dOut, 0, ar.d_bytes, 0x00020000);
  [[maybe_unused]] const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(POOL ? (const void *)ar.arg : (const void *)ar.dOut), 0,
                                                     POOL ? ar.a_bytes : 0u, 0x00020000);
  const auto rsO = __builtin_amdgcn_make_buffer_rsrc((void *)ar.dX, 0, DX ? ar.dx_bytes : 0u, 0x00020000);
  const int cp = lane & 7;
  // the same descriptors for the LDS-DMAs, which are inline asm (gemm_common.h, lds_dma16: hipcc fenced every LDS access of
  // the tile loop against the builtin -- the loop's stages never overlapped anything)
  const i32x4_rs dsX = raw_rsrc_v4(ar.X, ar.x_bytes), dsP = raw_rsrc_v4(ar.P, ar.p_bytes), dsD = raw_rsrc_v4(ar.dOut, ar.d_bytes);
  [[maybe_unused]] const i32x4_rs dsA = raw_rsrc_v4(POOL ? (const void *)ar.arg : (const void *)ar.dOut, POOL ? ar.a_bytes : 0u);

  if constexpr (DX) {   // the weight image: [TJ][TI] units of [64 n][64 i], contraction-major
    const i32x4_rs rsW = raw_rsrc_v4(ar.W, ar.w_bytes);
#pragma unroll
    for (int v = 0; v < TJ; ++v)
#pragma unroll
      for (int u = 0; u < TI; ++u)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int ur = (wave * 2 + d) * 8 + (lane >> 3);
          const unsigned off = (unsigned)((((v * 64 + ur) * ar.ldw) + u * 64 + (cp ^ (xg(ur) << 1)) * 8) * 2);
          lds_dma16(rsW, wimg + (v * TI + u) * 8192 + (wave * 2 + d) * 1024, off);
        }
  }

  unsigned vx[2], vp[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int ur = (wave * 2 + d) * 8 + (lane >> 3);
    vx[d] = (unsigned)((ur * ar.ldx + (cp ^ (xg(ur) << 1)) * 8) * 2) + (unsigned)kt0 * (unsigned)(64 * ar.ldx * 2);
    vp[d] = (unsigned)((ur * ar.Nj + (cp ^ (xg(ur) << 1)) * 8) * 2) + (unsigned)kt0 * (unsigned)(64 * ar.Nj * 2);
  }
  const unsigned x_step = (unsigned)(64 * ar.ldx * 2), p_step = (unsigned)(64 * ar.Nj * 2);
  const int smask = ar.S - 1;   // (pooled: S is a power of two, 16 .. 64)
  const int sshift = POOL ? __builtin_ctz((unsigned)ar.S) : 0;

  [[maybe_unused]] u32x4_t g_d[TJ];   // GREG: the next tile's group rows
  [[maybe_unused]] u32x2_t g_a[TJ];
  auto stage = [&](int step) {
    const bool live = step < nkt;
    unsigned char *base = stages + (step % NS) * STAGE;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
      for (int u = 0; u < TI; ++u)
        lds_dma16(dsX, base + u * 8192 + (wave * 2 + d) * 1024, live ? vx[d] + u * 128 : DEAD);
#pragma unroll
      for (int v = 0; v < TJ; ++v)
        lds_dma16(dsP, base + (TI + v) * 8192 + (wave * 2 + d) * 1024, live ? vp[d] + v * 128 : DEAD);
      if constexpr (!POOL) {
#pragma unroll
        for (int v = 0; v < TJ; ++v)
          lds_dma16(dsD, base + (TI + TJ + v) * 8192 + (wave * 2 + d) * 1024, live ? vp[d] + v * 128 : DEAD);
      }
      vx[d] += x_step;
      vp[d] += p_step;
    }
    if constexpr (C::GREG) {
      // this wave's 16 rows lie in ONE group (S % 16 == 0): this lane's chunk of the group's dOut and arg rows, per unit
      const long row0 = (long)(kt0 + step) * 64 + wave * 16;
      const unsigned grp = (unsigned)(row0 >> sshift);
      const bool ok = live && row0 < ar.R;
#pragma unroll
      for (int v = 0; v < TJ; ++v) {
        const unsigned e = grp * (unsigned)ar.Nj + v * 64 + cp * 8;
        g_d[v] = __builtin_amdgcn_raw_buffer_load_b128(rsD, ok ? e * 2u : DEAD, 0, 0);
        g_a[v] = __builtin_amdgcn_raw_buffer_load_b64(rsA, ok ? e : DEAD, 0, 0);
      }
    } else if constexpr (POOL) {
      // this wave's 16 rows lie in ONE group (S % 16 == 0): its dOut row (lanes 0 .. Nj/8-1) and arg row (lanes 0 .. Nj/16-1)
      const long row0 = (long)(kt0 + step) * 64 + wave * 16;
      const unsigned grp = (unsigned)(row0 >> sshift);
      const bool ok = live && row0 < ar.R;
      lds_dma16(dsD, base + C::UNITS * 8192 + wave * 2048, ok && lane * 8 < ar.Nj ? (grp * (unsigned)ar.Nj + lane * 8) * 2u : DEAD);
      lds_dma16(dsA, base + C::UNITS * 8192 + wave * 2048 + 1024, ok && lane * 16 < ar.Nj ? grp * (unsigned)ar.Nj + lane * 16 : DEAD);
    }
  };

  // per-lane constants of the in-place transform: logical 16-B chunk c = lane & 7 of every dP unit
  //   dP = sc g + c0 + x c1,   c1 = -sc rstd dgamma / R,   c0 = -sc dbeta / R - c1 mean;   g masked where x sc + sh <= 0
  float k_sc[TJ][8], k_sh[TJ][8], k_c0[TJ][8], k_c1[TJ][8];
  {
    const float invR = 1.0f / (float)ar.R;
#pragma unroll
    for (int v = 0; v < TJ; ++v)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int n = v * 64 + cp * 8 + i;
        const float sc = ar.scale[n], mb = ar.dgb[n] * invR, mg = ar.dgb[ar.Nj + n] * invR;
        k_sc[v][i] = sc;
        k_sh[v][i] = ar.relu ? ar.shift[n] : INFINITY;
        k_c1[v][i] = -(sc * ar.rstd[n] * mg);
        k_c0[v][i] = -(sc * mb) - k_c1[v][i] * ar.mean[n];
      }
  }

  // XT: the layer's input was never stored -- the x tile is the previous layer's pre-activation and becomes
  // relu(x xscale + xshift) in place, wave by wave on its own rows, like dP
  [[maybe_unused]] float x_sc[XT ? TI : 1][8], x_sh[XT ? TI : 1][8];
  if constexpr (XT) {
#pragma unroll
    for (int u = 0; u < TI; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = u * 64 + cp * 8 + i;
        x_sc[u][i] = c < ar.ldx ? ar.xscale[c] : 0.f;
        x_sh[u][i] = c < ar.ldx ? ar.xshift[c] : 0.f;
      }
  }

  const int row16 = lane & 15, q4 = lane >> 4;
  const int xc_q = (lane & 15) >> 2;
  const int xcg = (xc_q >> 1) | ((q4 & 1) << 1), xc0 = (8 * q4 + xc_q) * 128 + 8 * (lane & 3);

  [[maybe_unused]] float r_b[RED ? TI : 1][4][4], r_g[RED ? TI : 1][4][4];   // RED: sums of g and g xhat, this lane's channels
  if constexpr (RED) {
    for (int c = tid; c < TI * 64; c += 256)
      s_prev[c] = c < ar.ldx ? make_float4(ar.xscale[c], ar.xshift[c], ar.xmean[c], ar.xrstd[c]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < TI; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { r_b[u][t][e] = 0.f; r_g[u][t][e] = 0.f; }
  }

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
    // this wave's DMAs of the tile have landed -- its own 16 rows of every image.  Operations count in issue order: behind the
    // tile's DMAs lie (NS - 2) later stages and the dX stores of the tiles in between (none yet in the first two iterations)
    if (step < 2) wait_vmcnt<(NS - 2) * C::NDMA>();
    else wait_vmcnt<C::WAITN>();
    unsigned char *buf = stages + (step % NS) * STAGE;
    {
      // ---- dP in place of P: rows 16 wave + (lane >> 3) + 8 d, logical chunk lane & 7 ----------------------------------------
      [[maybe_unused]] const unsigned char *garea = buf + C::UNITS * 8192 + wave * 2048;
#pragma unroll
      for (int v = 0; v < TJ; ++v) {
        [[maybe_unused]] bf16x8 gd;
        [[maybe_unused]] u32x2_t ga;
        if constexpr (C::GREG) {
          gd = __builtin_bit_cast(bf16x8, g_d[v]);
          ga = g_a[v];
        } else if constexpr (POOL) {
          gd = *reinterpret_cast<const bf16x8 *>(garea + (v * 64 + cp * 8) * 2);
          ga = *reinterpret_cast<const u32x2_t *>(garea + 1024 + v * 64 + cp * 8);
        }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int row = wave * 16 + (lane >> 3) + 8 * d;
          const int pos = cp ^ ((((lane >> 4) & 1) | (d << 1)) << 1);
          unsigned char *pp = buf + (TI + v) * 8192 + row * 128 + pos * 16;
          const bf16x8 pv = *reinterpret_cast<const bf16x8 *>(pp);
          bf16x8 dv;
          if constexpr (!POOL) dv = *reinterpret_cast<const bf16x8 *>(pp + TJ * 8192);
          [[maybe_unused]] const unsigned s = (unsigned)(row & smask);   // (tiles start at multiples of 64 >= S)
          bf16x8 o;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float xf = (float)pv[i];
            float g;
            if constexpr (POOL) {
              const unsigned a8 = ((i < 4 ? ga[0] : ga[1]) >> (8 * (i & 3))) & 0xffu;
              g = a8 == s ? (float)gd[i] : 0.0f;
            } else {
              g = (float)dv[i];
            }
            if (!(xf * k_sc[v][i] + k_sh[v][i] > 0.0f)) g = 0.0f;
            o[i] = (__bf16)(k_sc[v][i] * g + k_c0[v][i] + xf * k_c1[v][i]);
          }
          *reinterpret_cast<bf16x8 *>(pp) = o;
        }
      }
    }
    auto transform_x = [&]() {   // XT: relu(x xscale + xshift) in place, this wave's own rows; rows past the end become 0
#pragma unroll
      for (int u = 0; u < TI; ++u)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int row = wave * 16 + (lane >> 3) + 8 * d;
          const int pos = cp ^ ((((lane >> 4) & 1) | (d << 1)) << 1);
          unsigned char *pp = buf + u * 8192 + row * 128 + pos * 16;
          const bf16x8 v = *reinterpret_cast<const bf16x8 *>(pp);
          const bool inside = (long)(kt0 + step) * 64 + row < ar.R;   // (their dP is not zero)
          bf16x8 o;
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = (__bf16)(inside ? fmaxf((float)v[i] * x_sc[u][i] + x_sh[u][i], 0.0f) : 0.0f);
          *reinterpret_cast<bf16x8 *>(pp) = o;
        }
    };
    if constexpr (XT && !RED) transform_x();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BQ_BARRIER();   // every wave's X rows have landed and its dP rows are written; the previous tile's reads are done
    stage(step + NS - 1);

    if constexpr (DX) {
      // ---- dX rows 16 wave .. + 16 = dP W: contraction over the Nj channels ------------------------------------------------
      f32x4 accx[TI][4];
#pragma unroll
      for (int u = 0; u < TI; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) accx[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int rowx = wave * 16 + row16;
#pragma unroll
      for (int v = 0; v < TJ; ++v)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const bf16x8 fbx = *reinterpret_cast<const bf16x8 *>(buf + (TI + v) * 8192 + rowx * 128 +
                                                               (((q4 + 4 * kk) ^ (xg(rowx) << 1)) << 4));
#pragma unroll
          for (int u = 0; u < TI; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const bf16x8 fax = read_frag_cf<true>(wimg + (v * TI + u) * 8192, t, kk, 0, xc0, xcg);
              accx[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fax, fbx, accx[u][t], 0, 0, 0);
            }
        }
      const long r = (long)(kt0 + step) * 64 + rowx;
#pragma unroll
      for (int u = 0; u < TI; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int col = u * 64 + t * 16 + q4 * 4;
          u32x2_t pk;
          pk[0] = pack_bf16x2(accx[u][t][0], accx[u][t][1]);
          pk[1] = pack_bf16x2(accx[u][t][2], accx[u][t][3]);
          __builtin_amdgcn_raw_buffer_store_b64(pk, rsO, (col < ar.ldx && r < ar.R) ? (unsigned)((r * ar.ldx + col) * 2) : DEAD, 0, 0);
        }
      if constexpr (RED) {
        // ---- the previous layer's BatchNorm reduction: g = this row's dX (as stored: bf16) where that layer's ReLU passed,
        // xhat from its pre-activation -- the x tile, still untransformed, this wave's own rows ------------------------------
        const bool live = r < ar.R;
#pragma unroll
        for (int u = 0; u < TI; ++u)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int chunk = 2 * t + (q4 >> 1);
            const bf16x4 pv = *reinterpret_cast<const bf16x4 *>(buf + u * 8192 + rowx * 128 + ((chunk ^ (xg(rowx) << 1)) << 4) +
                                                                (q4 & 1) * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float4 k = s_prev[u * 64 + t * 16 + q4 * 4 + e];   // (sc, sh, mean, rstd)
              const float xf = (float)pv[e];
              float g = (float)(__bf16)accx[u][t][e];
              if (!live || !(xf * k.x + k.y > 0.0f)) g = 0.0f;
              r_b[u][t][e] += g;
              r_g[u][t][e] += g * ((xf - k.z) * k.w);
            }
          }
        transform_x();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BQ_BARRIER();   // every wave's x rows are the activation now
      }
    }

    // ---- dW += dP^T X: contraction over the tile's 64 rows ----------------------------------------------------------------------
#pragma unroll
    for (int v = 0; v < TJ; ++v) {
      bf16x8 fb[2][2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb[b][kk] = read_frag_cf<true>(buf + (TI + v) * 8192, wc * 2 + b, kk, 0, xc0, xcg);
#pragma unroll
      for (int u = 0; u < TI; ++u) {
        bf16x8 fa[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) fa[a][kk] = read_frag_cf<true>(buf + u * 8192, wr * 2 + a, kk, 0, xc0, xcg);
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
          if (i < ar.ldx && j < ar.Nj)
            *reinterpret_cast<float4 *>(slice + (long)j * ar.ldo + i) =
                make_float4(acc[u][v][a][b][0], acc[u][v][a][b][1], acc[u][v][a][b][2], acc[u][v][a][b][3]);
        }
      }
  if constexpr (RED) {
    // one record per (workgroup, wave): the 16 lanes of a q4 group hold different rows of the same four channels
    float *rec = ar.red_part + ((long)blockIdx.x * 4 + wave) * 2 * ar.ldx;
#pragma unroll
    for (int u = 0; u < TI; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float sb = r_b[u][t][e], sg = r_g[u][t][e];
          sb += dpp_f32_add<0x111>(sb); sb += dpp_f32_add<0x112>(sb); sb += dpp_f32_add<0x114>(sb); sb += dpp_f32_add<0x118>(sb);
          sg += dpp_f32_add<0x111>(sg); sg += dpp_f32_add<0x112>(sg); sg += dpp_f32_add<0x114>(sg); sg += dpp_f32_add<0x118>(sg);
          const int c = u * 64 + t * 16 + q4 * 4 + e;
          if (row16 == 15 && c < ar.ldx) {
            rec[c] = sb;
            rec[ar.ldx + c] = sg;
          }
        }
  }
}

// pooled layers: dbeta / dgamma partial sums from the arg-max table -- one pre-activation value per (group, channel) instead of
// the search over the group's S rows (bn_bwd_reduce_kernel<RELU, true> reads all of P for it)
__global__ __launch_bounds__(256) void bn_bwd_reduce_arg_kernel(const __bf16 *__restrict__ dy, const __bf16 *__restrict__ x,
                                                                const unsigned char *__restrict__ arg,
                                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                                const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                float *__restrict__ partial, long G, int C, int S, int relu,
                                                                int gpc) {
  __shared__ float s_b[256], s_g[256];
  // thread = one channel of one group row; 256 / C group rows per sweep (C <= 256)
  const int c = threadIdx.x % C, gp = threadIdx.x / C, gps = 256 / C;
  const long g0 = (long)blockIdx.x * gpc, g1 = min(G, g0 + gpc);
  const float sc = scale[c], sh = shift[c], mu = mean[c], rs = rstd[c];
  float ab = 0.f, ag = 0.f;
  if (gp < gps)
    for (long gi = g0 + gp; gi < g1; gi += gps) {
      const int a = arg[gi * C + c];
      const float xf = (float)x[(gi * S + a) * C + c];
      float g = (float)dy[gi * C + c];
      if (relu && !(xf * sc + sh > 0.0f)) g = 0.0f;
      ab += g;
      ag += g * ((xf - mu) * rs);
    }
  s_b[threadIdx.x] = ab;
  s_g[threadIdx.x] = ag;
  __syncthreads();
  if (threadIdx.x < C) {
    float tb = 0.f, tg = 0.f;
    for (int p = 0; p < gps; ++p) { tb += s_b[p * C + threadIdx.x]; tg += s_g[p * C + threadIdx.x]; }
    partial[(long)blockIdx.x * 2 * C + threadIdx.x] = tb;
    partial[(long)blockIdx.x * 2 * C + C + threadIdx.x] = tg;
  }
}

template <int TI, int TJ, bool POOL, bool DX, bool XT = false, bool RED = false>
static int launch_sa_bwd(int wgs, hipStream_t st, const SaBwdArgs &a) {
  if constexpr (SaBwdCfg<TI, TJ, POOL, DX, RED>::FITS) {
    hipLaunchKernelGGL((sa_bwd_kernel<TI, TJ, POOL, DX, XT, RED>), dim3(wgs), dim3(256), 0, st, a);
    return 0;
  }
  return -1;
}

template <int TI, int TJ>
static int launch_sa_bwd_pd(bool pool, bool dx, int wgs, hipStream_t st, const SaBwdArgs &a) {
  if (a.xscale) {   // a deferred input always wants its gradient, and is 64 or 128 channels wide
    if constexpr (TI <= 2) {
      if (a.red_part)
        return pool ? launch_sa_bwd<TI, TJ, true, true, true, true>(wgs, st, a) : launch_sa_bwd<TI, TJ, false, true, true, true>(wgs, st, a);
      return pool ? launch_sa_bwd<TI, TJ, true, true, true>(wgs, st, a) : launch_sa_bwd<TI, TJ, false, true, true>(wgs, st, a);
    }
    return -1;
  }
  if (pool) return dx ? launch_sa_bwd<TI, TJ, true, true>(wgs, st, a) : launch_sa_bwd<TI, TJ, true, false>(wgs, st, a);
  return dx ? launch_sa_bwd<TI, TJ, false, true>(wgs, st, a) : launch_sa_bwd<TI, TJ, false, false>(wgs, st, a);
}

static int sa_bwd_lds(int ti, int tj, bool pool, bool dx, bool red = false) {
  const int units = ti + tj * (pool ? 1 : 2), stage3 = units * 8192 + (pool ? 8192 : 0), wb = dx ? ti * tj * 8192 : 0;
  const int rl = red ? ti * 64 * 16 : 0;
  const bool two = BQ_SA_PREFER_TWO && 2 * units * 8192 + wb + rl <= 80 * 1024 && 3 * stage3 + wb + rl > 80 * 1024;
  return ((3 * stage3 + wb + rl <= 160 * 1024 && !two) ? 3 * stage3 + wb : 2 * units * 8192 + wb) + rl;
}

}  // namespace bq
using namespace bq;

// 1 when bq_sa_bwd_fused has a kernel for a layer with ldx input elements per row and Nj output channels
extern "C" int bq_sa_bwd_supported(int ldx, int Nj, int S, int pool, int need_dx) {
  const int ti = (ldx + 63) / 64, tj = Nj / 64;
  if (ldx <= 0 || ldx % 8 || Nj % 64 || ti < 1 || ti > 3 || (tj != 1 && tj != 2 && !(tj == 4 && ti == 2))) return 0;
  if (pool && !(S == 16 || S == 32 || S == 64)) return 0;
  return sa_bwd_lds(ti, tj, pool != 0, need_dx != 0) <= 160 * 1024;
}

static int sa_bwd_wgs(long R, int ldx, int Nj, int pool, int need_dx, bool red) {
  const int ti = (ldx + 63) / 64, tj = Nj / 64;
  const long nkt = (R + 63) / 64;
  long wgs = sa_bwd_lds(ti, tj, pool != 0, need_dx != 0, red) <= 80 * 1024 ? 512 : 256;
  if (wgs > nkt) wgs = nkt;
  return (int)(wgs < 1 ? 1 : wgs);
}

// (an upper bound for every form of the call: the form that carries the previous layer's reduction may run fewer workgroups)
extern "C" int bq_sa_bwd_workgroups(long R, int ldx, int Nj, int pool, int need_dx) {
  return sa_bwd_wgs(R, ldx, Nj, pool, need_dx, false);
}

// The fused backward of a SharedMLP layer after its BatchNorm reduction (dgb): dX bf16 (R, ldx) (null: not needed) and
// dW f32 (Nj, ldo) from x (R, ldx), the stored pre-activation p (R, Nj), dOut ((R, Nj), or (R / S, Nj) + arg when pooled),
// w bf16 (Nj, ldw).  part: bq_sa_bwd_workgroups(...) * Nj * ldo floats.
extern "C" int bq_sa_bwd_fused(const void *x, const void *p, const void *dout,
                               const void *arg, const void *w, const float *scale,
                               const float *shift, const float *mean, const float *rstd,
                               const float *dgb, void *dx, float *dw, float *part, long R,
                               int ldx, int Nj, int ldw, int ldo, int S, int relu, int pool,
                               void *stream) {
  return bq_sa_bwd_fused_x(x, nullptr, nullptr, p, dout, arg, w, scale, shift, mean, rstd, dgb, dx, dw, part, R, ldx, Nj, ldw, ldo,
                           S, relu, pool, stream);
}

// the same with a DEFERRED input: x holds the previous layer's stored pre-activation (ldx = 64 or 128 channels, no padding)
// and the layer's input is relu(x xscale + xshift), formed tile by tile in LDS (xscale == NULL: plain bq_sa_bwd_fused); dx is
// then the gradient w.r.t. that activation and is required
extern "C" int bq_sa_bwd_fused_x(const void *x, const float *xscale, const float *xshift, const void *p, const void *dout,
                                 const void *arg, const void *w, const float *scale, const float *shift, const float *mean,
                                 const float *rstd, const float *dgb, void *dx, float *dw, float *part, long R, int ldx, int Nj,
                                 int ldw, int ldo, int S, int relu, int pool, void *stream) {
  return bq_sa_bwd_fused_xr(x, xscale, xshift, nullptr, nullptr, nullptr, nullptr, p, dout, arg, w, scale, shift, mean, rstd, dgb,
                            dx, dw, part, R, ldx, Nj, ldw, ldo, S, relu, pool, stream);
}

// 1 when bq_sa_bwd_fused_xr can also carry the previous layer's reduction for this shape (its LDS table still fits)
extern "C" int bq_sa_bwd_reduce_supported(int ldx, int Nj, int S, int pool) {
  const int ti = (ldx + 63) / 64, tj = Nj / 64;
  if (!(ldx == 64 || ldx == 128) || !bq_sa_bwd_supported(ldx, Nj, S, pool, 1)) return 0;
  return sa_bwd_lds(ti, tj, pool != 0, true, true) <= 160 * 1024;
}

// ... and with the PREVIOUS layer's BatchNorm reduction riding on the pass (VERDICT r4 item 3a's first half: "the BatchNorm
// reduce in the epilogue of the next layer's dX GEMM"): that layer's dOut is this call's dx, its pre-activation this call's x.
// xmean / xrstd: that layer's mean / rstd; red_part: scratch of bq_sa_bwd_workgroups(...) * 4 * 2 * ldx floats; red_dgb: f32
// (2, ldx) = its dbeta | dgamma (what bq_bn_backward_reduce would return for it).  red_dgb == NULL: bq_sa_bwd_fused_x.
extern "C" int bq_sa_bwd_fused_xr(const void *x, const float *xscale, const float *xshift, const float *xmean, const float *xrstd,
                                  float *red_part, float *red_dgb, const void *p, const void *dout, const void *arg, const void *w,
                                  const float *scale, const float *shift, const float *mean, const float *rstd, const float *dgb,
                                  void *dx, float *dw, float *part, long R, int ldx, int Nj, int ldw, int ldo, int S, int relu,
                                  int pool, void *stream) {
  BQ_REQUIRE(!red_dgb || (xscale && xmean && xrstd && red_part && bq_sa_bwd_reduce_supported(ldx, Nj, S, pool)), BQ_EINVAL,
             "bq_sa_bwd_fused_xr: the carried reduction needs a deferred input, its mean / rstd, scratch and a shape with room "
             "for its table (ldx=%d Nj=%d pool=%d)", ldx, Nj, pool);
  BQ_REQUIRE(!xscale || (xshift && dx && (ldx == 64 || ldx == 128)), BQ_EINVAL,
             "bq_sa_bwd_fused_x: a deferred input needs xshift, dx and ldx = 64 or 128 (ldx=%d)", ldx);
  BQ_REQUIRE(x && p && dout && scale && shift && mean && rstd && dgb && dw && part && R > 0, BQ_EINVAL,
             "bq_sa_bwd_fused: null pointer / no rows");
  BQ_REQUIRE(bq_sa_bwd_supported(ldx, Nj, S, pool, dx != nullptr), BQ_EINVAL,
             "bq_sa_bwd_fused: ldx=%d Nj=%d S=%d pool=%d dx=%d not supported", ldx, Nj, S, pool, dx != nullptr);
  BQ_REQUIRE(!pool || (arg && R % S == 0), BQ_EINVAL, "bq_sa_bwd_fused: pooled layers need arg and R %% S == 0");
  BQ_REQUIRE(!dx || (w && ldw >= ((ldx + 63) / 64) * 64), BQ_EINVAL, "bq_sa_bwd_fused: w (Nj, ldw >= %d) required for dX",
             ((ldx + 63) / 64) * 64);
  BQ_REQUIRE(ldo % 4 == 0 && ldo >= ldx && ((uintptr_t)part % 16 == 0) && ((uintptr_t)dw % 16 == 0), BQ_EINVAL,
             "bq_sa_bwd_fused: ldo %% 4, ldo >= ldx, 16-byte aligned outputs");
  BQ_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)p % 16 == 0) && ((uintptr_t)dout % 16 == 0) && ((uintptr_t)w % 16 == 0) &&
                 ((uintptr_t)arg % 16 == 0) && ((uintptr_t)dx % 8 == 0),
             BQ_EINVAL, "bq_sa_bwd_fused: operands must be 16-byte aligned");
  BQ_REQUIRE(R * (long)ldx * 2 < 0x7FFFFFFFL - 64L * ldx * 2 - 1024 && R * (long)Nj * 2 < 0x7FFFFFFFL - 64L * Nj * 2 - 1024,
             BQ_ELIMIT, "bq_sa_bwd_fused: an operand larger than 2 GB");
  SaBwdArgs a;
  a.X = (const __bf16 *)x; a.P = (const __bf16 *)p; a.dOut = (const __bf16 *)dout; a.arg = (const unsigned char *)arg;
  a.W = (const __bf16 *)w; a.scale = scale; a.shift = shift; a.mean = mean; a.rstd = rstd; a.dgb = dgb;
  a.dX = (__bf16 *)dx; a.part = part; a.xscale = xscale; a.xshift = xshift;
  a.xmean = xmean; a.xrstd = xrstd; a.red_part = red_dgb ? red_part : nullptr;
  a.R = (int)R; a.ldx = ldx; a.Nj = Nj; a.ldw = ldw; a.ldo = ldo; a.S = pool ? S : 1; a.relu = relu;
  a.x_bytes = (unsigned)(R * (long)ldx * 2);
  a.p_bytes = (unsigned)(R * (long)Nj * 2);
  a.d_bytes = (unsigned)((pool ? R / S : R) * (long)Nj * 2);
  a.a_bytes = pool ? (unsigned)((R / S) * (long)Nj) : 0u;
  a.w_bytes = w ? (unsigned)((long)Nj * ldw * 2) : 0u;
  a.dx_bytes = a.x_bytes;
  const int wgs = sa_bwd_wgs(R, ldx, Nj, pool, dx != nullptr, red_dgb != nullptr);
  const int ti = (ldx + 63) / 64, tj = Nj / 64;
  hipStream_t st = (hipStream_t)stream;
  int rc = -1;
  if (ti == 1 && tj == 1) rc = launch_sa_bwd_pd<1, 1>(pool, dx != nullptr, wgs, st, a);
  else if (ti == 1 && tj == 2) rc = launch_sa_bwd_pd<1, 2>(pool, dx != nullptr, wgs, st, a);
  else if (ti == 2 && tj == 1) rc = launch_sa_bwd_pd<2, 1>(pool, dx != nullptr, wgs, st, a);
  else if (ti == 2 && tj == 2) rc = launch_sa_bwd_pd<2, 2>(pool, dx != nullptr, wgs, st, a);
  else if (ti == 2 && tj == 4) rc = launch_sa_bwd_pd<2, 4>(pool, dx != nullptr, wgs, st, a);
  else if (ti == 3 && tj == 1) rc = launch_sa_bwd_pd<3, 1>(pool, dx != nullptr, wgs, st, a);
  else if (ti == 3 && tj == 2) rc = launch_sa_bwd_pd<3, 2>(pool, dx != nullptr, wgs, st, a);
  BQ_REQUIRE(rc == 0, BQ_EINVAL, "bq_sa_bwd_fused: no kernel for %d x %d units", ti, tj);
  if (a.red_part) {
    const int frc = bq_bn_fold(red_part, red_dgb, wgs * 4, 2 * ldx, stream);
    if (frc) return frc;
  }
  return bq_wgrad_rows_reduce(part, dw, ldx, Nj, ldo, wgs, stream);
}

extern "C" int bq_bn_backward_reduce_arg(const void *dy, const void *x, const void *arg,
                                                                                const float *scale, const float *shift,
                                                                                const float *mean, const float *rstd,
                                                                                float *partial, float *dgb, long R, int C, int S,
                                                                                int relu, void *stream) {
  BQ_REQUIRE(dy && x && arg && scale && shift && mean && rstd && partial && dgb && R > 0, BQ_EINVAL,
             "bn_backward_reduce_arg: null pointer / no rows");
  BQ_REQUIRE(S > 0 && S <= 256 && R % S == 0 && C > 0 && C <= 256 && 256 % C == 0, BQ_ELIMIT,
             "bn_backward_reduce_arg: C=%d S=%d unsupported", C, S);
  const int chunks = bq_bn_chunks(R, S, 1);
  const long G = R / S;
  const int gpc = (int)((G + chunks - 1) / chunks);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_bwd_reduce_arg_kernel, dim3(chunks), dim3(256), 0, st, (const __bf16 *)dy, (const __bf16 *)x,
                     (const unsigned char *)arg, scale, shift, mean, rstd, partial, G, C, S, relu, gpc);
  return bq_bn_fold(partial, dgb, chunks, 2 * C, stream);
}
