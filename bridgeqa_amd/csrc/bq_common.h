// Shared device/host helpers for libbqhip.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "bqhip.h"

namespace bq {

void set_error(const char *fmt, ...);
int check_launch(const char *what);
int device_cus();   // compute units of the CURRENT device (cached per device ordinal, pn2_ops.hip)

// library-internal entry points shared between translation units (not part of include/*.h)
extern "C" int bq_bn_fold(const float *partial, float *out, int chunks, int W, void *stream);                       // bn.hip
extern "C" int bq_wgrad_rows_reduce(float *part, float *out, int Ni, int Nj, int ldo, int pieces, void *stream);    // gemm.hip

#define BQ_REQUIRE(cond, code, ...)   \
  do {                                \
    if (!(cond)) {                    \
      bq::set_error(__VA_ARGS__);     \
      return (code);                  \
    }                                 \
  } while (0)

// ---- canonical arithmetic: every product and sum rounded separately ----------
// The translation unit is compiled with -ffp-contract=off; the pragma makes the intent local.
__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  const float xx = dx * dx;
  const float yy = dy * dy;
  const float zz = dz * dz;
  return (xx + yy) + zz;
}

__device__ __forceinline__ float sqnorm(float x, float y, float z) {
#pragma clang fp contract(off)
  const float xx = x * x;
  const float yy = y * y;
  const float zz = z * z;
  return (xx + yy) + zz;
}

// ---- wave64 DPP reductions ---------------------------------------------------------------------
// row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143 (gfx9 family encodings).
// Lanes without a DPP source keep `old` (= their own value), harmless for idempotent max/min.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}

// max over the 64 lanes, returned wave-uniform (SGPR)
__device__ __forceinline__ float wave_max_f32(float v) {
  v = fmaxf(v, dpp_f32<0x111, 0xF>(v));
  v = fmaxf(v, dpp_f32<0x112, 0xF>(v));
  v = fmaxf(v, dpp_f32<0x114, 0xF>(v));
  v = fmaxf(v, dpp_f32<0x118, 0xF>(v));
  v = fmaxf(v, dpp_f32<0x142, 0xA>(v));
  v = fmaxf(v, dpp_f32<0x143, 0xC>(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = min(v, dpp_u32<0x111, 0xF>(v));
  v = min(v, dpp_u32<0x112, 0xF>(v));
  v = min(v, dpp_u32<0x114, 0xF>(v));
  v = min(v, dpp_u32<0x118, 0xF>(v));
  v = min(v, dpp_u32<0x142, 0xA>(v));
  v = min(v, dpp_u32<0x143, 0xC>(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// reductions over lanes 0..15 only (one DPP row), result taken from lane 15
__device__ __forceinline__ float row0_max_f32(float v) {
  v = fmaxf(v, dpp_f32<0x111, 0xF>(v));
  v = fmaxf(v, dpp_f32<0x112, 0xF>(v));
  v = fmaxf(v, dpp_f32<0x114, 0xF>(v));
  v = fmaxf(v, dpp_f32<0x118, 0xF>(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
}
__device__ __forceinline__ unsigned row0_min_u32(unsigned v) {
  v = min(v, dpp_u32<0x111, 0xF>(v));
  v = min(v, dpp_u32<0x112, 0xF>(v));
  v = min(v, dpp_u32<0x114, 0xF>(v));
  v = min(v, dpp_u32<0x118, 0xF>(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}

// ---- single-instruction DPP reduction steps (inline asm) -----------------------------------------
// hipcc expands update_dpp + max into mov / s_nop / mov_dpp / canonicalise / max (5 instructions per step);
// the serial arg-max chain of furthest point sampling is made of these steps, so they are written as the
// one VOP2-DPP instruction the hardware has.  `s_nop 1` = the 2 wait states a DPP read needs after a VALU
// write of the same VGPR.  Lanes without a DPP source (row start, masked rows) are not written: they keep v.
#define BQ_DPP_STEP(op, ctrl) "s_nop 1\n\t" op " %0, %0, %0 " ctrl "\n\t"
#define BQ_DPP_ROW(op)                                       \
  BQ_DPP_STEP(op, "row_shr:1 row_mask:0xf bank_mask:0xf")    \
  BQ_DPP_STEP(op, "row_shr:2 row_mask:0xf bank_mask:0xf")    \
  BQ_DPP_STEP(op, "row_shr:4 row_mask:0xf bank_mask:0xf")    \
  BQ_DPP_STEP(op, "row_shr:8 row_mask:0xf bank_mask:0xf")
#define BQ_DPP_WAVE(op)                                      \
  BQ_DPP_ROW(op)                                             \
  BQ_DPP_STEP(op, "row_bcast:15 row_mask:0xa bank_mask:0xf") \
  BQ_DPP_STEP(op, "row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1"

// signed-int max over the wave; used on float BITS: for d in {-1} U [0, +inf) the int order is the float order
__device__ __forceinline__ int wave_max_i32(int v) {
  asm volatile(BQ_DPP_WAVE("v_max_i32_dpp") : "+v"(v));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32_fast(unsigned v) {
  asm volatile(BQ_DPP_WAVE("v_min_u32_dpp") : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Wave-wide arg-max on the total order (d desc, key asc): returns the winning lane; *dmax_bits / *kmin get the
// winner's distance bits and key (wave-uniform).  The key reduction only runs on an exact tie in d.
__device__ __forceinline__ int wave_argmax(float d, unsigned key, int *dmax_bits, unsigned *kmin) {
  const int bits = __float_as_int(d);
  const int mx = wave_max_i32(bits);
  const unsigned long long tied = __ballot(bits == mx);
  int L;
  if (__builtin_popcountll(tied) == 1) {
    L = __builtin_ctzll(tied);
    *kmin = (unsigned)__builtin_amdgcn_readlane((int)key, L);
  } else {
    const unsigned mine = (bits == mx) ? key : 0xFFFFFFFFu;
    const unsigned k = wave_min_u32_fast(mine);
    L = __builtin_ctzll(__ballot(mine == k));
    *kmin = k;
  }
  *dmax_bits = mx;
  return L;
}

// (d, key) -> one u64 whose unsigned order is (d desc, key asc); d < 0 ("no candidate") maps below everything
__device__ __forceinline__ unsigned long long pack_best(int dbits, unsigned key) {
  const unsigned hi = dbits < 0 ? 0u : (unsigned)dbits + 1u;
  return ((unsigned long long)hi << 32) | (unsigned long long)(~key);
}

// Workgroup barrier that waits for LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would make
// every sampling round wait for the previous round's global store to be acknowledged (~0.3 us).  Legal where
// no wave reads global memory another wave wrote since the last full __syncthreads().
__device__ __forceinline__ void barrier_lds_only() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Sample ids are produced one per round by every lane (wave-uniform value).  Wave 0 parks id j in lane j%64 and
// flushes 64 ids with one coalesced store, so no round has a global store in flight at its barrier.
struct IdSink {
  int32_t *out;
  int acc;
  __device__ __forceinline__ void push(int j, int id, int m) {
    if (threadIdx.x < 64) {
      if ((int)threadIdx.x == (j & 63)) acc = id;
      if ((j & 63) == 63 || j == m - 1) {
        const int base = j & ~63;
        if (base + (int)threadIdx.x <= j) out[base + threadIdx.x] = acc;
      }
    }
  }
};

__device__ __forceinline__ unsigned lane_id() { return __lane_id(); }

}  // namespace bq
