// Shared definitions of the MFMA bf16 GEMM family (csrc/gemm.hip: 256 x 256 and 64 x {64, 32} tiles, the SharedMLP
// kernels; csrc/gemm_mid.hip: the 256 x 128 tile with two co-resident workgroups per CU): the problem descriptor, the LDS
// images of K-contiguous / contraction-major operands and their fragment reads, the GELU fit and its LDS table.
#pragma once
#include <type_traits>

#include "bq_common.h"
#include "bqhip_fusion.h"

namespace bq {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;
typedef __attribute__((address_space(3))) void lds_void_t;

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_DGELU = 3, EPI_BIAS_CE = 4, EPI_ADD = 5 };
enum { GF_P_XC = 1, GF_Q_XC = 2, GF_OUT_F32 = 4, GF_ACCUM = 8 };

struct GemmProblem {
  const __bf16 *P, *Q;
  void *out;           // out[j * ldo + i]: bf16, or fp32 with GF_OUT_F32
  const void *bias;    // over i (EPI_BIAS / EPI_BIAS_GELU) or null: fp32, or bf16 when bias_bf16
  void *out2;          // EPI_BIAS_GELU: gelu(out) as bf16, same layout as out
  const __bf16 *aux;   // EPI_DGELU: the pre-activation y[j][i] (ld = ldo): out = acc * gelu'(y); EPI_ADD: out = acc + aux[j][i]
  float *colsum;       // optional fp32 [Ni]: += sum_j out[j][i] (bias gradient of the producing layer)
  int ldp, ldq, ldo;   // leading dimensions in elements
  int Ni, Nj, Kc;
  int tile0;           // first workgroup of this problem in the launch
  unsigned p_bytes, q_bytes;  // bounds of the operand buffers (a K-contiguous operand whose rows are shorter than Kc:
                              // its partner is zero-padded, its own tail reads run into the next row)
  // batched-row maps (bq_gemm_desc, include/bqhip_fusion.h): logical row r of Q / of out (and aux / out2) lives at element
  // (r / rpb) * bstride + (r % rpb) * ld -- a (batch, rows, cols) view with a batch stride, e.g. the first 1025 of every
  // sample's 1045 key rows.  rpb = 0: plain rows.  On a contraction-major Q the map is on the CONTRACTION rows.
  int q_bstride, o_bstride;
  // 32-bit words only: the table is indexed by a run-time problem number, and with 8- / 16-bit members hipcc could no longer
  // read it with scalar loads from the kernel-argument segment -- it copied the whole 4 KB table into LDS in every workgroup
  // (+ 4 KB of LDS, + 10-20 us on every small-tile launch of the step: round 4's first version of this struct)
  int rpb_pack;        // q_rpb | o_rpb << 16
  int tiles_ks;        // tiles along i | ksplit << 16.  ksplit > 1 (fp32 out, small-tile kernel): the contraction is cut
                       // into ksplit pieces, one workgroup each, accumulated with fp32 atomics into a zero-initialised `out`
  int flags;           // bit 0: bias is bf16; bit 1: accum (fp32 out, small-tile kernel: add to `out` / `colsum` with fp32
                       // atomics instead of storing)
  __host__ __device__ __forceinline__ int q_rpb() const { return rpb_pack & 0xffff; }
  __host__ __device__ __forceinline__ int o_rpb() const { return (int)((unsigned)rpb_pack >> 16); }
  __host__ __device__ __forceinline__ int tiles_i() const { return tiles_ks & 0xffff; }
  __host__ __device__ __forceinline__ int ksplit() const { return (int)((unsigned)tiles_ks >> 16); }
  __host__ __device__ __forceinline__ int bias_bf16() const { return flags & 1; }
  __host__ __device__ __forceinline__ int accum() const { return (flags >> 1) & 1; }
};
// the descriptor table travels as a kernel argument (4 KB limit): 36 problems x 112 B + 8
static_assert(sizeof(GemmProblem) == 112, "GemmProblem grew: GEMM_MAX_PROBLEMS x sizeof must stay below the 4 KB kernarg limit");

// element offset of logical row r under a batched-row map (rpb = 0: r * ld).  A handful of calls per tile / per short
// contraction step: the integer division is noise there.
__device__ __forceinline__ unsigned mapped_row(int r, int ld, int rpb, int bstride) {
  if (rpb == 0) return (unsigned)(r * ld);
  const int b = r / rpb;
  return (unsigned)(b * bstride + (r - b * rpb) * ld);
}

__device__ __forceinline__ void load_bias4(const GemmProblem &pr, int i, float (&bv)[4]) {
  if (pr.bias_bf16()) {
    const bf16x4 b = *reinterpret_cast<const bf16x4 *>(reinterpret_cast<const __bf16 *>(pr.bias) + i);
    bv[0] = (float)b[0]; bv[1] = (float)b[1]; bv[2] = (float)b[2]; bv[3] = (float)b[3];
  } else {
    const float4 b = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(pr.bias) + i);
    bv[0] = b.x; bv[1] = b.y; bv[2] = b.z; bv[3] = b.w;
  }
}

constexpr int GEMM_MAX_PROBLEMS = 36;
struct GemmArgs {
  int n;
  int total_tiles;
  GemmProblem p[GEMM_MAX_PROBLEMS];
  // stream-K form of gemm128_kernel (csrc/gemm_mid.hip) only: tickets ({arrive, done} u32 pairs per tile, zero between
  // launches) in the first SK_TICKET_BYTES, fp32 partial-tile slabs (2 per workgroup, 128 KB each) behind them
  void *sk_ws;
};
constexpr long SK_TICKET_BYTES = 65536, SK_SLAB_BYTES = 256 * 128 * 4;
static_assert(sizeof(GemmArgs) <= 4096, "kernel-argument limit");

// ---- LDS images -----------------------------------------------------------------------------------------------------
// KC unit: [64 rows][64 k] bf16, 128-B rows, 16-B chunk ch of row r at r*128 + ((ch ^ (r & 7)) << 4)
// XC unit: [64 kc][64 outs] bf16, 128-B rows, 16-B chunk ch of row kc at kc*128 + ((ch ^ (xg(kc) << 1)) << 4)
// both conflict-free for the fragment reads below (tools/lds_bank_sim.py)
__device__ __forceinline__ int xg(int kc) { return ((kc >> 1) & 1) | (((kc >> 3) & 1) << 1); }

// GELU x * Phi(x) (reference: nn.GELU / HF "gelu", the erf form) and its derivative, fp32, for epilogues whose result
// is rounded to bf16.  Phi(x) = sigmoid(x * (a1 + a3 x^2 + a5 x^4)) on |x| <= 8 (clamped beyond: Phi is 0 / 1 to fp32
// there): a minimax fit of the Gaussian CDF, max |Phi - Phi_erf| = 3.1e-5, max |gelu - gelu_erf| = 3.1e-5, and the
// derivative of the fitted function differs from gelu_erf' by <= 1.2e-4 (fit and error scan: DESIGN.md §4.4) -- two
// orders of magnitude below the bf16 rounding (2^-9 relative) applied to the result.  8 VALU operations per element
// (one v_exp_f32, one v_rcp_f32) instead of ~20 for an erf polynomial: the epilogue runs with the matrix pipe idle.
constexpr float GELU_A1 = 1.59525515f, GELU_A3 = 7.38511083e-2f, GELU_A5 = -6.82350683e-4f;
__device__ __forceinline__ float gauss_cdf(float x, float &x2) {
  const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
  x2 = xc * xc;
  constexpr float L = -1.4426950408889634f;   // exp(-z) = exp2(L z): folded into the coefficients (explicit fmas: the
                                              // library is built with -ffp-contract=off)
  const float z = xc * __builtin_fmaf(x2, __builtin_fmaf(x2, GELU_A5 * L, GELU_A3 * L), GELU_A1 * L);
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
}
__device__ __forceinline__ float gelu_f(float x) {
  float x2;
  return x * gauss_cdf(x, x2);
}
__device__ __forceinline__ float dgelu_f(float x) {
  // d/dx [x s(z(x))] = s + x z'(x) s (1 - s), evaluated at the CLAMPED argument: beyond |x| = 8 the second term is below
  // 1e-11 and s is 0 / 1 to fp32, so no select is needed (a select of the two results became a branch per element under
  // hipcc, a select of constants three more instructions); s (1 - s) = s - s^2 as one fma.  15 VALU operations.
  const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
  const float x2 = xc * xc;
  constexpr float L = -1.4426950408889634f;   // exp(-z) = exp2(L z): folded into the coefficients
  const float z = xc * __builtin_fmaf(x2, __builtin_fmaf(x2, GELU_A5 * L, GELU_A3 * L), GELU_A1 * L);
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
  const float u = xc * __builtin_fmaf(x2, __builtin_fmaf(x2, 5.0f * GELU_A5, 3.0f * GELU_A3), GELU_A1);
  const float v = __builtin_fmaf(-s, s, s);
  return __builtin_fmaf(u, v, s);
}

// Both functions above are applied to bf16 VALUES (the stored pre-activation), i.e. they have 65536 possible arguments:
// the 256-tile kernel tabulates them in LDS once per workgroup (filled with the very functions above while the prologue's
// DMAs are in flight) and its epilogue gathers instead of evaluating ~25 VALU operations per element with the matrix pipe
// idle.  Index = [sign][magnitude bits clamped to 2^-16 .. 8]: below 2^-16 Phi and gelu' are 0.5 to 2e-5, at and beyond
// 8 the functions above are clamped themselves.  4866 entries (19 KB beside the 128 KB of staging / output images).
constexpr unsigned GELU_TAB_LO = (127 - 16) << 7, GELU_TAB_HI = (127 + 3) << 7;
constexpr int GELU_TAB_HALF = (int)(GELU_TAB_HI - GELU_TAB_LO) + 1, GELU_TAB_N = 2 * GELU_TAB_HALF;
template <bool DERIV>
__device__ __forceinline__ void gelu_tab_fill(float *tab, int tid, int nthreads) {
  for (int e = tid; e < GELU_TAB_N; e += nthreads) {
    const unsigned sgn = e >= GELU_TAB_HALF ? 1u : 0u;
    const unsigned bits = (sgn << 15) | ((unsigned)e - sgn * GELU_TAB_HALF + GELU_TAB_LO);
    const float x = __uint_as_float(bits << 16);
    float x2;
    tab[e] = DERIV ? dgelu_f(x) : gauss_cdf(x, x2);
  }
}
// bits16: the bf16 pattern in the low half of a dword (upper half ignored)
__device__ __forceinline__ float gelu_tab_at(const float *tab, unsigned bits16) {
  const unsigned mag = min(max(bits16 & 0x7fffu, GELU_TAB_LO), GELU_TAB_HI) - GELU_TAB_LO;
  return tab[mag + ((bits16 >> 15) & 1u) * GELU_TAB_HALF];
}

// value of the lane `n` to the left in the same 16-lane row, 0 where there is none (bound_ctrl): row-prefix sums
template <int CTRL>
__device__ __forceinline__ float dpp_f32_add(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  bf16x2_t v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

template <bool XC>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char *unit, int sub16, int kk, int kc_base, const int (&xc_base)[4]) {
  // KC: kc_base = lane term row16*128 + ((q4 ^ (row16 & 7)) << 4) for kk = 0; kk = 1 toggles bit 6
  if (!XC) {
    return *reinterpret_cast<const bf16x8 *>(unit + sub16 * 2048 + (kc_base ^ (kk << 6)));
  } else {
    // XC: xc_base[s] = (8*gq + q)*128 + ((s ^ g) << 5) + 8*p ; rows +kk*32, second read +4 rows
    const unsigned char *a0 = unit + xc_base[sub16] + kk * 4096;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)a0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)(a0 + 512));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// the same with the XC lane term in CLOSED FORM (xc0 = (8*gq + q)*128 + 8*p, xcg = g: xc_base[s] = xc0 + ((s ^ g) << 5)) for
// kernels whose sub16 is a run-time value (gemm64_kernel: wave row / column): indexing the int[4] table dynamically makes it
// an alloca, which hipcc may promote to LDS (+ 4 KB per workgroup and an LDS round trip per fragment address)
template <bool XC>
__device__ __forceinline__ bf16x8 read_frag_cf(const unsigned char *unit, int sub16, int kk, int kc_base, int xc0, int xcg) {
  if (!XC) {
    return *reinterpret_cast<const bf16x8 *>(unit + sub16 * 2048 + (kc_base ^ (kk << 6)));
  } else {
    const unsigned char *a0 = unit + xc0 + ((sub16 ^ xcg) << 5) + kk * 4096;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)a0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)(a0 + 512));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// ---- the same fragment reads as INLINE ASM (round 6) --------------------------------------------------------------------
// hipcc's waitcnt pass knows that an LDS-DMA (buffer_load ... lds) writes LDS and, without alias information, puts
// s_waitcnt vmcnt(0) in front of LDS reads that follow one: in gemm256_kernel every phase's fragment reads then waited for ALL
// the DMAs in flight -- the four-stage look-ahead of its counted waits never existed, a K tile of the weight-gradient launches
// cost four L2 / HBM round trips (2.1-2.5 us against 0.85 us of MFMA time).  Reads the compiler cannot see are ordered by the
// kernel's own counted waits and barriers only; the caller waits lgkmcnt(0) behind a sched_barrier before the first use.
template <bool XC>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char *unit, int sub16, int kk, int kc_base, const int (&xc_base)[4]);
typedef __attribute__((address_space(3))) unsigned char lds_u8c_t;
__device__ __forceinline__ unsigned lds_addr_of(const void *p) { return (unsigned)(size_t)((lds_u8c_t *)p); }
// (measurement builds only, tools/bench_dw_prefetch.py: 0 = the compiler-visible reads of rounds 1-5)
#ifndef BQ_G256_ASM_READS
#define BQ_G256_ASM_READS 1
#endif
template <bool XC>
__device__ __forceinline__ bf16x8 read_frag_asm(unsigned unit, int sub16, int kk, int kc_base, const int (&xc_base)[4]) {
  if (!BQ_G256_ASM_READS) {
    return read_frag<XC>((const unsigned char *)((lds_u8c_t *)(size_t)unit), sub16, kk, kc_base, xc_base);
  } else if (!XC) {
    bf16x8 v;
    const unsigned a = unit + (unsigned)(sub16 * 2048 + (kc_base ^ (kk << 6)));
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
    return v;
  } else {
    bf16x4 lo, hi;
    const unsigned a = unit + (unsigned)(xc_base[sub16] + kk * 4096);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(hi) : "v"(a) : "memory");
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// An LDS-DMA the compiler cannot see (round 6): 16 B per lane from buffer offset `voff` (bounds-checked by the descriptor) to
// LDS byte address lds + 16 * lane (lds wave-uniform).  Issued through __builtin_amdgcn_raw_ptr_buffer_load_lds hipcc fences
// EVERY later LDS access of the kernel against it (s_waitcnt vmcnt(0) in front of plain reads, transposed reads and writes
// alike); a kernel whose in-place LDS work is ordered by its own counted waits and barriers loses its stages to that fence.
// M0 is written inside the asm and not declared (a reserved register: hipcc ignores it on a clobber list): a kernel uses EITHER
// this OR the builtin for all its DMAs -- the compiler hoists its own M0 initialisations as if nobody else wrote M0.
// BQ_DMA_ASM=0 (measurement builds) restores the builtin.
#ifndef BQ_DMA_ASM
#define BQ_DMA_ASM 1
#endif
typedef int i32x4_rs __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_rs raw_rsrc_v4(const void *p, unsigned bytes) {   // = make_buffer_rsrc(p, 0, bytes, 0x00020000)
  const unsigned long a = (unsigned long)p;
  i32x4_rs r;
  r[0] = (int)(unsigned)a; r[1] = (int)(unsigned)((a >> 32) & 0xffffu); r[2] = (int)bytes; r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void lds_dma16(i32x4_rs rs, const void *lds, unsigned voff) {
  if (!BQ_DMA_ASM) {
    const auto r = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long)(unsigned)rs[1] << 32) | (unsigned)rs[0]), 0, rs[2], rs[3]);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)(lds_u8c_t *)lds, 16, voff, 0, 0, 0);
    return;
  }
  const unsigned l = lds_addr_of(lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(l), "v"(voff), "s"(rs) : "memory");
}

// the closed-form contraction-major read (read_frag_cf<true>) as inline asm, for the same reason as read_frag_asm; the
// K-contiguous form stays a compiler-visible load (hipcc does not fence those: tools/isa_waits.py)
template <bool XC>
__device__ __forceinline__ bf16x8 read_frag_cf_x(const unsigned char *unit, int sub16, int kk, int kc_base, int xc0, int xcg) {
  if (!XC) {
    return read_frag_cf<false>(unit, sub16, kk, kc_base, xc0, xcg);
  } else {
    bf16x4 lo, hi;
    const unsigned a = lds_addr_of(unit) + (unsigned)(xc0 + ((sub16 ^ xcg) << 5) + kk * 4096);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(hi) : "v"(a) : "memory");
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

#define BQ_BARRIER()                                        \
  do {                                                      \
    __builtin_amdgcn_sched_barrier(0);                      \
    asm volatile("s_barrier" ::: "memory");                 \
    __builtin_amdgcn_sched_barrier(0);                      \
  } while (0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// csrc/gemm_mid.hip: the 256 (i) x 128 (j) tile kernel (bf16 out, K-contiguous Q); -1 when it has no such form
int launch_gemm_mid(const GemmArgs &ga, bool p_xc, bool q_xc, bool out_f32, int epi, bool background, hipStream_t st);
// the stream-K workspace registered for `st` (bq_gemm_set_workspace), or nullptr
void *gemm_sk_workspace(hipStream_t st, long *bytes);
// which stream-K forms may run (bq_gemm_streamk_mode): bit 0 the 256 x 128 kernel's, bit 1 the 256 x 256 kernel's (both
// measured slower than whole tiles: off by default)
int gemm_sk_mode();

}  // namespace bq
