// Shared device/host helpers for libbqhip.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "bqhip.h"

namespace bq {

void set_error(const char *fmt, ...);
int check_launch(const char *what);

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

__device__ __forceinline__ unsigned lane_id() { return __lane_id(); }

}  // namespace bq
