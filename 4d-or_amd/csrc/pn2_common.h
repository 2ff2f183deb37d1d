// pn2_common.h — shared device helpers for the gfx950 kernels.
// Written for CDNA4 only: wave = 64 lanes, DPP row ops, hardware fp32 atomics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pn2_hip.h"

typedef unsigned long long u64;

#define PN2_WAVE 64

// ---- pinned arithmetic (must match oracle/pn2_oracle.c orc_sq3 / orc_dot3) ----
// a*a + b*b + c*c  ==>  fma(c, c, fma(a, a, b*b)); the translation unit is built
// with -ffp-contract=off so nothing else is fused behind our back.
__device__ __forceinline__ float pn2_sq3(float dx, float dy, float dz) {
  float t = __fmul_rn(dy, dy);
  t = __fmaf_rn(dx, dx, t);
  return __fmaf_rn(dz, dz, t);
}
__device__ __forceinline__ float pn2_dot3(float p1, float w1, float p2, float w2,
                                          float p3, float w3) {
  float t = __fmul_rn(p2, w2);
  t = __fmaf_rn(p1, w1, t);
  return __fmaf_rn(p3, w3, t);
}

// ---- wave64 helpers -----------------------------------------------------------
__device__ __forceinline__ int pn2_lane() { return __lane_id(); }

// Number of set bits of `mask` strictly below this lane.
__device__ __forceinline__ int pn2_prefix_popc(u64 mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// DPP controls (GFX9 encoding).
#define PN2_DPP_QUAD_XOR1 0xB1   // quad_perm:[1,0,3,2]
#define PN2_DPP_QUAD_XOR2 0x4E   // quad_perm:[2,3,0,1]
#define PN2_DPP_ROW_HALF_MIRROR 0x141
#define PN2_DPP_ROW_MIRROR 0x140

template <int CTRL>
__device__ __forceinline__ u64 pn2_dpp_u64(u64 v) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, 0xF, 0xF, false);
  hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, 0xF, 0xF, false);
  return ((u64)hi << 32) | lo;
}

// Max over the 16 lanes of each DPP row; every lane of a row ends with the row max.
__device__ __forceinline__ u64 pn2_row16_max_u64(u64 v) {
  u64 o;
  o = pn2_dpp_u64<PN2_DPP_QUAD_XOR1>(v); v = o > v ? o : v;
  o = pn2_dpp_u64<PN2_DPP_QUAD_XOR2>(v); v = o > v ? o : v;
  o = pn2_dpp_u64<PN2_DPP_ROW_HALF_MIRROR>(v); v = o > v ? o : v;
  o = pn2_dpp_u64<PN2_DPP_ROW_MIRROR>(v); v = o > v ? o : v;
  return v;
}

__device__ __forceinline__ u64 pn2_readlane_u64(u64 v, int lane) {
  unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
  unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return ((u64)hi << 32) | lo;
}

// Wave-wide max of a u64; result is wave-uniform (lives in SGPRs).
__device__ __forceinline__ u64 pn2_wave_max_u64(u64 v) {
  v = pn2_row16_max_u64(v);
  u64 a = pn2_readlane_u64(v, 0), b = pn2_readlane_u64(v, 16);
  u64 c = pn2_readlane_u64(v, 32), d = pn2_readlane_u64(v, 48);
  a = a > b ? a : b;
  c = c > d ? c : d;
  return a > c ? a : c;
}

// Wave-wide max of a u32 (4 DPP stages inside the rows of 16 + a scalar max of the four row results);
// the result is wave-uniform.  One VALU instruction per stage where the u64 form needs five.
__device__ __forceinline__ unsigned pn2_wave_max_u32(unsigned v) {
  unsigned o;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, PN2_DPP_QUAD_XOR1, 0xF, 0xF, false); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, PN2_DPP_QUAD_XOR2, 0xF, 0xF, false); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, PN2_DPP_ROW_HALF_MIRROR, 0xF, 0xF, false); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, PN2_DPP_ROW_MIRROR, 0xF, 0xF, false); v = o > v ? o : v;
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}

// Wave-wide arg-max of the totally ordered key (hi, lo): returns the wave-uniform maximum and the
// ballot of the lanes that hold it (exactly one lane when keys are unique).
__device__ __forceinline__ u64 pn2_wave_argmax_u32x2(unsigned hi, unsigned lo, unsigned &mhi, unsigned &mlo) {
  mhi = pn2_wave_max_u32(hi);
  const unsigned l2 = hi == mhi ? lo : 0u;
  mlo = pn2_wave_max_u32(l2);
  return __ballot(hi == mhi && lo == mlo);
}

__device__ __forceinline__ float pn2_readlane_f32(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// ---- host-side launch error capture ---------------------------------------------
extern thread_local int pn2_tls_hip_error;
static inline int pn2_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pn2_tls_hip_error = (int)e;
    return PN2_ELAUNCH;
  }
  return PN2_OK;
}
