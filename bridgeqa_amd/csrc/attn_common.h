// Shared device code of the fused attention kernels (csrc/attn.hip: one workgroup per row block; csrc/attn_persist.hip: the
// resident-grid forms of round 6): operand maps, LDS images, the LDS-DMA tile pipeline and the per-tile bodies of the three
// passes.  Two translation units on purpose: instantiating the resident-grid kernels beside the block kernels in ONE unit
// changed the block kernels' register allocation (dQ pass: 12 -> 188 B of scratch per lane, 238 -> 353 us at L = 1025).
#pragma once
#include <stdlib.h>

#include "bq_common.h"
#include "bqhip_fusion.h"

namespace bq {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef __attribute__((address_space(3))) void lds_void_t;

#ifndef BQ_ATTN_DQ_MINW
#define BQ_ATTN_DQ_MINW 4     // launch bound of the dQ pass: 128 VGPRs (two spilled), four workgroups per CU -- measured 2 / 3 / 4:
                              // backward 0.331 / 0.307 / 0.297 ms at L = 1025, 6.18 / 6.19 / 5.99 ms at L = 4097 (tools/rebuild_with.sh)
#endif
#ifndef BQ_ATTN_FWD_MINW
#define BQ_ATTN_FWD_MINW 3    // workgroups per CU the PLAIN forward is compiled for (measurement builds: 4)
#endif
constexpr int AT_D = 64;      // head dim
constexpr int AT_QW = 32;     // query rows per wave
constexpr int AT_NW = 4;      // waves per workgroup
constexpr int AT_QB = AT_QW * AT_NW;  // 128 query rows per workgroup
constexpr int AT_KB = 64;     // keys per LDS tile

__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float xhalf_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
  return fmaxf(__int_as_float(r[0]), __int_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}

// byte offset of 16-B chunk `ch` (0..7) of row `row` in a [rows][64 x bf16] LDS image
// The XOR key g(row) = row bits (1, 2, 1^3) makes BOTH read patterns of these kernels conflict-free on gfx950's 64 x 4-B
// banks (tools/lds_bank_sim.py --attn; searched over all GF(2)-linear keys): the ds_read_b128 row reads (lane groups
// {0-3,12-15,20-27}, ... of MI355X_MICROARCH.md §LDS: 16 rows of one 16-B column must fall on 16 distinct (row parity,
// position) pairs) and the ds_read_b64_tr_b16 transposed reads (32 lanes = 4 rows x 4 chunks x 2 halves: the two rows of
// equal parity must land in different 64-B halves).  The plain key row & 7 of round 1 was 2-way on both (PMC:
// SQ_LDS_BANK_CONFLICT = 40 % of SQ_LDS_IDX_ACTIVE in all three kernels).  The ds_write_b128 staging (8 lanes = one row's
// 8 chunks) is conflict-free under any key.
__device__ __forceinline__ int swz_key(int row) { return ((row >> 1) & 3) | ((((row >> 1) ^ (row >> 3)) & 1) << 2); }
__device__ __forceinline__ int swz(int row, int ch) { return row * 128 + ((ch ^ swz_key(row)) << 4); }

struct AttnDims {
  int B, H, Lq, Lk, Lqp, Lkp;       // real lengths; padded lengths (multiples of 64) of the transposed operands / mask
  long q_bs, q_rs, q_hs;            // element strides: batch, row (token), head   (Q, dQ)
  long k_bs, k_rs, k_hs;            // K, V, dK, dV
  long o_bs, o_rs, o_hs;            // O (forward) / dO (backward)
  const float *mask;                // optional additive key mask [B][Lkp], ALREADY multiplied by log2(e); or null
  float scale;                      // softmax scale (natural units)
  float inv_keep;                   // 1/(1-p) of the attention-probability dropout, 1 when off
  unsigned drop_thresh;             // p * 2^32 (0 = no dropout)
  unsigned seed;                    // per-call offset ...
  const unsigned *seed_ptr;         // ... combined with a per-step device counter (graph-replay safe), may be null
  int causal;                       // 1: key j is visible to query i only when j <= i (decoder self-attention,
                                    // reference med.py:640-672 causal_mask), on top of the key mask
  // Optional SECOND key/value segment: the keys are cat(segment 1 (Lk rows of K / V), segment 2 (Lk2 rows of K2 / V2))
  // as in the twin cross-attention over cat(image tokens, other stream's text states) (reference med.py:549-562) --
  // without the concatenated tensor ever existing.  Tiles never straddle segments: the 64-key tile list is the
  // nkt1 = ceil(Lk / 64) tiles of segment 1 followed by ceil(Lk2 / 64) tiles of segment 2, and the "padded key
  // index" 64 * tile + i addresses the mask row (each segment padded to a multiple of 64 there) and the dropout hash.
  int Lk2, nkt1;
  long k2_bs, k2_rs, k2_hs;
  const __bf16 *K2, *V2;
  __bf16 *dK2, *dV2;
};

// tile kt of the two-segment key list: operand bases of this (batch, head), row stride, first row and row count of
// its segment, whether it is that segment's last tile, and the padded key index one past the segment's last key
struct KeyTile {
  const __bf16 *k, *v;
  long rs;
  int row0, nrows;
  bool last;
  int kend;
};
__device__ __forceinline__ KeyTile key_tile(const AttnDims &dm, const __bf16 *Kb, const __bf16 *Vb, const __bf16 *K2b,
                                            const __bf16 *V2b, int kt) {
  if (kt < dm.nkt1) return KeyTile{Kb, Vb, dm.k_rs, kt * 64, dm.Lk, kt == dm.nkt1 - 1, dm.Lk};
  const int t2 = kt - dm.nkt1;
  return KeyTile{K2b, V2b, dm.k2_rs, t2 * 64, dm.Lk2, t2 == (dm.Lk2 + 63) / 64 - 1, dm.nkt1 * 64 + dm.Lk2};
}

__device__ __forceinline__ unsigned eff_seed(const AttnDims &dm) {
  return dm.seed_ptr ? dm.seed_ptr[0] * 2654435761u + dm.seed : dm.seed;
}

// keep-decision of the attention dropout: a stateless hash of (seed, batch*head, query, key), so forward and
// backward regenerate the same mask and nothing is stored
__device__ __forceinline__ bool drop_keep(unsigned seed, int bh, int q, int key, unsigned thresh) {
  unsigned x = seed ^ ((unsigned)bh * 0x9E3779B1u) ^ ((unsigned)q * 0x85EBCA77u) ^ ((unsigned)key * 0xC2B2AE3Du);
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x >= thresh;
}

// 16 contiguous-in-groups-of-4 floats for the register rows of lane half h: v[rr] = src[32*blk + crow(rr, h)]
__device__ __forceinline__ void load_rowvals(const float *src, int blk, int h, float *v) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 x = *reinterpret_cast<const float4 *>(src + 32 * blk + 8 * g + 4 * h);
    v[4 * g + 0] = x.x; v[4 * g + 1] = x.y; v[4 * g + 2] = x.z; v[4 * g + 3] = x.w;
  }
}

// Staging of one 64 x 64 bf16 tile (source rows of 64 contiguous elements) into a swizzled LDS image via
// registers: 512 chunks of 16 B, two per thread.  Loads are issued one tile ahead of their LDS commit.
__device__ __forceinline__ uint4 stage_load(const __bf16 *base, long row_stride, int row0, int nrows_valid, int c) {
  const int row = c >> 3, ch = c & 7;
  const int gr = min(row0 + row, nrows_valid - 1);  // clamp: rows past the end repeat the last one (masked later)
  return *reinterpret_cast<const uint4 *>(base + (long)gr * row_stride + ch * 8);
}
// The same load as a wave-uniform tile base (scalar registers) plus a per-thread 32-bit byte offset that is computed
// ONCE per kernel (the clamped variant once more for the last tile): no 64-bit address arithmetic per tile (the
// clamped re-computation cost ~6 quarter-rate integer multiplies per tile on a VALU-bound kernel, §4.3 of DESIGN.md).
__device__ __forceinline__ unsigned stage_off(long row_stride, int row_in_tile, int c) {
  return (unsigned)((row_in_tile * (int)row_stride + (c & 7) * 8) * 2);
}
__device__ __forceinline__ uint4 stage_ld(const __bf16 *tile_base, unsigned byte_off) {
  return *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(tile_base) + byte_off);
}
__device__ __forceinline__ void stage_store(unsigned char *lds, int c, uint4 v) {
  *reinterpret_cast<uint4 *>(lds + swz(c >> 3, c & 7)) = v;
}

// A-operand fragment of the TRANSPOSE of a row-major LDS image (64 rows x 64 bf16, swizzled by swz), read with gfx950's
// ds_read_b64_tr_b16: lane (r, h) gets column d = dbase + r of rows row0 .. row0+3 (elements 0-3) and row0+8 .. row0+11
// (elements 4-7) -- the k-order 16s + 8(j>>2) + 4h + (j&3) of the accumulator-as-B-operand maps when row0 = 32*blk +
// 16*s + 4*h.  Per 16-lane group the instruction gathers a 4-row x 16-column block: lane 4q+p of the group supplies the
// address of row q, columns 4p..4p+3, and lane i receives column i of the 4 rows (tools/tr_read_probe.py).  No
// transposed copy of K / V / Q / dO exists any more, in HBM or in LDS.  EXEC must be all ones here.
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x8 tfrag_tr(const unsigned char *img, int row0, int dbase, int r) {
  const int q = (r & 15) >> 2, p = r & 3;
  const int ch = ((dbase + 16 * (r >> 4)) >> 3) + (p >> 1);
  const unsigned char *a0 = img + swz(row0 + q, ch) + 8 * (p & 1);
  const unsigned char *a1 = img + swz(row0 + 8 + q, ch) + 8 * (p & 1);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)a0);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)a1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Two row-major bf16 operands with 64-element rows (K and V of a (batch, head); Q and dO in the dK/dV pass), one 64-row tile
// of each per step, from global memory straight into the swizzled LDS images by LDS-DMA (buffer_load ... lds: 16 B per
// lane, lane-linear destination -- 8 lanes = one 128-B row, so the lane at position p of row r fetches chunk
// p ^ swz_key(r) of that row).  No staging registers, no ds_write, one add of address arithmetic per tile: 16-32 VGPRs and
// ~30 instructions per tile less than staging through registers.  The workgroup's four waves issue 2 + 2 DMAs of 8 rows
// each per tile; rows past the operand's last row are outside the descriptor and arrive as zeros (the last tile masks
// them).  Protocol (TileDma::turn): before tile kt is consumed every wave waits for its own DMAs, the barrier makes all of
// them visible and says that tile kt - 1's image is no longer read, then tile kt + 1 is issued into that image.
// A tile is four UNITS of 16 rows (2 DMAs of 8 rows per operand); unit u belongs to wave u, or -- in a ragged edge block
// whose idle waves have ended (attn_live_waves) -- to the live waves round-robin.
// SAME: both operands have the same row stride (K and V): one set of offsets (two VGPRs less -- what the dQ pass spilled).
template <bool SAME>
struct TileDma {
  __amdgpu_buffer_rsrc_t rsA, rsB;
  unsigned voffA[2], voffB[SAME ? 1 : 2], stepA, stepB, unitA, unitB;   // voff: rows d * 8 + lane / 8 of unit 0
  int first, stride;
  __device__ __forceinline__ void init(const __bf16 *A, long a_rs, const __bf16 *B, long b_rs, int rows, int lane, int wid,
                                       int live = AT_NW) {
    rsA = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, (unsigned)((((long)rows - 1) * a_rs + AT_D) * 2), 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc((void *)B, 0, (unsigned)((((long)rows - 1) * b_rs + AT_D) * 2), 0x00020000);
    first = __builtin_amdgcn_readfirstlane(wid);   // (the DMA's LDS base must be wave-uniform)
    stride = live;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int row = d * 8 + (lane >> 3);   // (swz_key reads row bits 1-3: the same in every unit)
      const unsigned ch = (unsigned)(((lane & 7) ^ swz_key(row)) << 4);
      voffA[d] = (unsigned)(row * (int)a_rs * 2) + ch;
      if (!SAME) voffB[d] = (unsigned)(row * (int)b_rs * 2) + ch;
    }
    stepA = (unsigned)(AT_KB * (int)a_rs * 2);
    stepB = (unsigned)(AT_KB * (int)b_rs * 2);
    unitA = (unsigned)(16 * (int)a_rs * 2);
    unitB = (unsigned)(16 * (int)b_rs * 2);
  }
  __device__ __forceinline__ void issue(int kt, unsigned char *imgA, unsigned char *imgB) const {
    for (int u = first; u < AT_NW; u += stride) {   // (one trip when all four waves live)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t *)(imgA + (u * 2 + d) * 1024), 16,
                                                 voffA[d] + (unsigned)kt * stepA + (unsigned)u * unitA, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t *)(imgB + (u * 2 + d) * 1024), 16,
                                                 (SAME ? voffA[d] : voffB[d]) + (unsigned)kt * stepB + (unsigned)u * unitB, 0, 0, 0);
      }
    }
  }
};

// live waves of the 128-row block that starts at row `start` of `L`: a wave whose 32 rows all lie past the end has
// nothing to compute.  It used to stay for the barriers and its share of the staging; now it ENDS at once (before any
// barrier: an ended wave is not waited for) and the live waves stage its units, so a ragged edge block -- the 1025th
// token of the ViT: one live wave -- holds one wave slot of its CU for the key loop instead of four.
__device__ __forceinline__ int attn_live_waves(int L, int start) { return min(AT_NW, (L - start + AT_QW - 1) / AT_QW); }

// One 64-key tile (two 32-key blocks) of the online-softmax forward for the 32 queries of a wave: S^T = K.Q^T from
// the LDS image s_k, mask / causal / length clamp, running max + rescale, P (with dropout) straight from the
// accumulator registers into O^T += V^T.P^T from the LDS image s_v.  qrow = this lane's query index.
//
// PLAIN (compile time): no key mask, not causal, no dropout -- the ViT's case, 3/4 of the attention time of the path.
// The wave-uniform special cases then vanish at compile time instead of being branched around, the tile becomes
// (almost) one basic block the scheduler can interleave, and `LASTP` (compile time, the peeled last tile) replaces
// the run-time `last`.  EARLY: both 32-key score blocks are issued to the matrix pipe before the first softmax.
__device__ __forceinline__ f32x16 score_block(const unsigned char *s_k, const bf16x8 (&qf)[4], int kb2, int r, int h) {
  f32x16 acc = {0};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const bf16x8 a = *reinterpret_cast<const bf16x8 *>(s_k + swz(kb2 * 32 + r, 2 * s + h));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], acc, 0, 0, 0);
  }
  return acc;
}

template <bool PLAIN = false, bool LASTP = false, bool EARLY = false, bool PK = false>
__device__ __forceinline__ void fwd_tile(const unsigned char *s_k, const unsigned char *s_v, const bf16x8 (&qf)[4],
                                         const AttnDims &dm, const float *mrow_, float scale_log2e, unsigned seed, int bh,
                                         int qrow, int kt, bool last_, int r, int h, f32x16 &o0, f32x16 &o1, float &m,
                                         float &lsum, int kend = -1) {
    if (kend < 0) kend = dm.Lk;  // padded key index one past the last valid key of this tile's segment
    const float *mrow = PLAIN ? nullptr : mrow_;
    const bool last = PLAIN ? LASTP : last_;
    const bool causal = !PLAIN && dm.causal;
    const bool drop = !PLAIN && dm.drop_thresh != 0;
    f32x16 early[2];
    if (EARLY) {
      early[0] = score_block(s_k, qf, 0, r, h);
      early[1] = score_block(s_k, qf, 1, r, h);
    }
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      // the second 32-key block of a segment's last tile may lie wholly past its end (L = 1025: one key in the 17th
      // tile): wave-uniform skip of its 8 MFMAs and softmax
      if (last && kb2 == 1 && kt * AT_KB + 32 >= kend) continue;
      const f32x16 acc = EARLY ? early[kb2] : score_block(s_k, qf, kb2, r, h);
      // sc is kept in log2 units when there is a mask (one fma per element) and in raw dot-product units otherwise
      // (the scale is folded into the max once and into the exp2 argument by an fma): fewer VALU ops per pair
      float sc[16];
      float mloc = -INFINITY;
      const int kbase = kt * AT_KB + kb2 * 32;
      const bool masked = mrow != nullptr;
      if (masked) {
        float mk[16];
        load_rowvals(mrow + kbase, 0, h, mk);
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i] = __builtin_fmaf(acc[i], scale_log2e, mk[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i] = acc[i];
      }
      if (last) {  // wave-uniform: only the last tile has keys past the end
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kbase + crow(i, h) >= kend) sc[i] = -INFINITY;
      }
      if (causal) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kbase + crow(i, h) > qrow) sc[i] = -INFINITY;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, sc[i]);
      if (!masked) mloc *= scale_log2e;  // scale > 0: max commutes with it
      mloc = xhalf_max(mloc);
      const float mnew = fmaxf(m, mloc);
      const float alpha = __builtin_amdgcn_exp2f(m - mnew);  // m = -inf on the first block -> 0
      m = mnew;
      float psum = 0.0f;
      bf16x8 pb0, pb1;
      float pv[16];
      if (PK) {
        // packed fp32: v_pk_fma_f32 / v_pk_add_f32 take two elements per issue slot (the forward is bound by vector issue,
        // DESIGN.md section 4.3); the sum pairs differently from the scalar form below
        if (masked) {
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const f32x2 t = f32x2{sc[i], sc[i + 1]} - f32x2{mnew, mnew};
            pv[i] = __builtin_amdgcn_exp2f(t[0]);
            pv[i + 1] = __builtin_amdgcn_exp2f(t[1]);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const f32x2 t = __builtin_elementwise_fma(f32x2{sc[i], sc[i + 1]}, f32x2{scale_log2e, scale_log2e}, f32x2{-mnew, -mnew});
            pv[i] = __builtin_amdgcn_exp2f(t[0]);
            pv[i + 1] = __builtin_amdgcn_exp2f(t[1]);
          }
        }
        f32x2 ps = {pv[0], pv[1]};
#pragma unroll
        for (int i = 2; i < 16; i += 2) ps += f32x2{pv[i], pv[i + 1]};
        psum = ps[0] + ps[1];
      } else {
        if (masked) {
#pragma unroll
          for (int i = 0; i < 16; ++i) pv[i] = __builtin_amdgcn_exp2f(sc[i] - mnew);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) pv[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[i], scale_log2e, -mnew));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) psum += pv[i];  // the softmax denominator uses the un-dropped probabilities
      }
      if (drop) {  // wave-uniform
#pragma unroll
        for (int i = 0; i < 16; ++i)
          pv[i] = drop_keep(seed, bh, qrow, kbase + crow(i, h), dm.drop_thresh) ? pv[i] * dm.inv_keep : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        pb0[i] = (__bf16)pv[i];
        pb1[i] = (__bf16)pv[8 + i];
      }
      lsum = lsum * alpha + psum;
      if (__ballot(alpha != 1.0f)) {  // wave-uniform: skip the rescale once the running max has settled
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int key0 = 32 * kb2 + 16 * s2 + 4 * h;  // s_v is the ROW-major V tile [key][d]
        const bf16x8 v0 = tfrag_tr(s_v, key0, 0, r), v1 = tfrag_tr(s_v, key0, 32, r);
        const bf16x8 pb = s2 == 0 ? pb0 : pb1;
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb, o1, 0, 0, 0);
      }
    }
}


// A fragment (8 x bf16) of a TRANSPOSED image [row][k]: elements k = 16*s2 + 8*(j>>2) + 4*h + (j&3) of row `row`
__device__ __forceinline__ bf16x8 tfrag(const unsigned char *img, int row, int kchunk, int h) {
  const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(img + swz(row, kchunk) + h * 8);
  const bf16x4 hi = *reinterpret_cast<const bf16x4 *>(img + swz(row, kchunk + 1) + h * 8);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// store a transposed accumulator pair (T[d][col], lane = col) as row `Xrow`[0..63] of a bf16 matrix
__device__ __forceinline__ void store_T(__bf16 *Xrow, const f32x16 &a0, const f32x16 &a1, int h, float mul) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bf16x4 w0, w1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w0[j] = (__bf16)(a0[4 * g + j] * mul);
      w1[j] = (__bf16)(a1[4 * g + j] * mul);
    }
    *reinterpret_cast<bf16x4 *>(Xrow + 8 * g + 4 * h) = w0;
    *reinterpret_cast<bf16x4 *>(Xrow + 32 + 8 * g + 4 * h) = w1;
  }
}

typedef AttnDims BwdDims;  // o_* strides describe dO

// One 64-key tile of the dQ pass for the 32 queries of a wave: P recomputed from S^T = K.Q^T and the forward's LSE,
// dP^T = V.dO^T, dS^T = P o (dP - delta) * scale, then dQ^T += K^T.dS^T (A = LDS image of the pre-transposed K).
template <bool PLAIN = false, bool LASTP = false>
__device__ __forceinline__ void dq_tile(const unsigned char *s_k, const unsigned char *s_v, const bf16x8 (&qf)[4], const bf16x8 (&gf)[4], const BwdDims &dm,
                                        const float *mrow_, float c, float scale, float lse, float delta, unsigned seed,
                                        int bh, int qrow, int kt, bool last_, int r, int h, f32x16 &a0, f32x16 &a1,
                                        int kend = -1) {
    if (kend < 0) kend = dm.Lk;
    const float *mrow = PLAIN ? nullptr : mrow_;  // PLAIN / LASTP: see fwd_tile
    const bool last = PLAIN ? LASTP : last_;
    const bool causal = !PLAIN && dm.causal;
    const bool drop = !PLAIN && dm.drop_thresh != 0;
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      if (last && kb2 == 1 && kt * AT_KB + 32 >= kend) continue;  // (as in fwd_tile: a block wholly past the end)
      f32x16 sacc = {0}, pacc = {0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 ak = *reinterpret_cast<const bf16x8 *>(s_k + swz(kb2 * 32 + r, 2 * s + h));
        const bf16x8 av = *reinterpret_cast<const bf16x8 *>(s_v + swz(kb2 * 32 + r, 2 * s + h));
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, qf[s], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, gf[s], pacc, 0, 0, 0);
      }
      bf16x8 d0, d1;
      const int kbase = kt * AT_KB + kb2 * 32;
      float mk[16];
      if (mrow) load_rowvals(mrow + kbase, 0, h, mk);
      float pv[16], gv[16];
      if (mrow) {
#pragma unroll
        for (int i = 0; i < 16; ++i) pv[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[i], c, mk[i] - lse));
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) pv[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[i], c, -lse));
      }
      if (last) {  // wave-uniform special cases stay out of the common path
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kbase + crow(i, h) >= kend) pv[i] = 0.0f;
      }
      if (causal) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kbase + crow(i, h) > qrow) pv[i] = 0.0f;
      }
      if (drop) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          gv[i] = drop_keep(seed, bh, qrow, kbase + crow(i, h), dm.drop_thresh) ? pacc[i] * dm.inv_keep : 0.0f;
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) gv[i] = pacc[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        d0[i] = (__bf16)(pv[i] * (gv[i] - delta));  // the softmax scale is applied once, to the dQ accumulators
        d1[i] = (__bf16)(pv[8 + i] * (gv[8 + i] - delta));
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int key0 = 32 * kb2 + 16 * s2 + 4 * h;
        const bf16x8 ds = s2 == 0 ? d0 : d1;
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag_tr(s_k, key0, 0, r), ds, a0, 0, 0, 0);   // K^T from the K tile
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag_tr(s_k, key0, 32, r), ds, a1, 0, 0, 0);
      }
    }
}


// One 64-query tile of the dK/dV pass for the 32 keys of a wave (lane = key): S = Q.K^T and dP = dO.V^T (A = Q / dO
// rows from LDS, B = register fragments of K and V), P from the forward's LSE, then dV^T += dO^T.P and dK^T += Q^T.dS
// with the transposed A operands read straight from the row-major tiles.  PLAIN / LASTP: see fwd_tile.
template <bool PLAIN = false, bool LASTP = false>
__device__ __forceinline__ void dkv_tile(const unsigned char *s_q, const unsigned char *s_g, const float *s_lse,
                                         const float *s_del, const bf16x8 (&kf)[4], const bf16x8 (&vf)[4],
                                         const BwdDims &dm, float c, float mkey_, unsigned seed, int bh, int kcol, int qt,
                                         bool last_, int r, int h, f32x16 &dk0, f32x16 &dk1, f32x16 &dv0, f32x16 &dv1) {
    const float mkey = PLAIN ? 0.0f : mkey_;
    const bool last = PLAIN ? LASTP : last_;
    const bool causal = !PLAIN && dm.causal;
    const bool drop = !PLAIN && dm.drop_thresh != 0;
#pragma unroll
    for (int qb2 = 0; qb2 < 2; ++qb2) {
      if (last && qb2 == 1 && qt * 64 + 32 >= dm.Lq) continue;  // the last query tile's second block wholly past the end
      f32x16 sacc = {0}, pacc = {0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 aq = *reinterpret_cast<const bf16x8 *>(s_q + swz(qb2 * 32 + r, 2 * s + h));
        const bf16x8 ag = *reinterpret_cast<const bf16x8 *>(s_g + swz(qb2 * 32 + r, 2 * s + h));
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kf[s], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag, vf[s], pacc, 0, 0, 0);
      }
      float lv[16], dl[16];
      load_rowvals(s_lse, qb2, h, lv);
      load_rowvals(s_del, qb2, h, dl);
      bf16x8 p0, p1, d0, d1;
      const int qbase = qt * AT_KB + qb2 * 32;
      float pv[16], pdv[16], gv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) pv[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[i], c, mkey - lv[i]));
      if (last) {  // query rows past the end were staged as copies of the last row: drop them (wave-uniform branch)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (qbase + crow(i, h) >= dm.Lq) pv[i] = 0.0f;
      }
      if (causal) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (qbase + crow(i, h) < kcol) pv[i] = 0.0f;
      }
      if (drop) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const bool keep = drop_keep(seed, bh, qbase + crow(i, h), kcol, dm.drop_thresh);
          gv[i] = keep ? pacc[i] * dm.inv_keep : 0.0f;
          pdv[i] = keep ? pv[i] * dm.inv_keep : 0.0f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) { gv[i] = pacc[i]; pdv[i] = pv[i]; }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        p0[i] = (__bf16)pdv[i];
        p1[i] = (__bf16)pdv[8 + i];
        d0[i] = (__bf16)(pv[i] * (gv[i] - dl[i]));  // the softmax scale is applied once, to the dK accumulators
        d1[i] = (__bf16)(pv[8 + i] * (gv[8 + i] - dl[8 + i]));
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int q0r = 32 * qb2 + 16 * s2 + 4 * h;  // dO^T / Q^T fragments straight from the row-major dO / Q tiles
        const bf16x8 pp = s2 == 0 ? p0 : p1, ds = s2 == 0 ? d0 : d1;
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag_tr(s_g, q0r, 0, r), pp, dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag_tr(s_g, q0r, 32, r), pp, dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag_tr(s_q, q0r, 0, r), ds, dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag_tr(s_q, q0r, 32, r), ds, dk1, 0, 0, 0);
      }
    }
}


// host side of csrc/attn_persist.hip (called by bq_attn_fwd / bq_attn_bwd when the shape suits a resident grid)
int attn_fwd_persist_launch(const void *Q, const void *K, const void *V, void *O, float *LSE, const AttnDims &dm, int nqb,
                            int tail, int slots, hipStream_t st);
int attn_bwd_dq_persist_launch(const void *Q, const void *K, const void *V, const void *dO, const float *LSE, const void *O,
                               float *DELTA, void *dQ, const BwdDims &dm, int nqb, int tail, int slots, hipStream_t st);
int attn_bwd_dkv_persist_launch(const void *Q, const void *K, const void *V, const void *dO, const float *LSE,
                                const float *DELTA, void *dK, void *dV, const BwdDims &dm, int nkb, int tail, int slots,
                                hipStream_t st);

}  // namespace bq
