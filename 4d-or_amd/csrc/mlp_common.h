// mlp_common.h — shared by the fused-MLP kernels (mlp_gemm.hip, mlp_bwd_fused.hip).
#pragma once
#include "pn2_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { PRO_NONE = 0, PRO_BNRELU = 1, PRO_GY = 2, PRO_POOLG = 3, PRO_FIRST = 4, PRO_LIFT = 5 };
enum { EPI_NONE = 0, EPI_STATS = 1, EPI_MASK = 2, EPI_POOL = 3, EPI_MASKL = 4 };

// ---- raw buffer access (gfx950) ------------------------------------------------
// Every global access of the GEMM goes through a buffer descriptor built from wave-uniform
// scalars: the per-lane part of the address is a 32-bit byte offset that is CONSTANT for the
// whole kernel, the tile / chunk part lives in the descriptor base and the SGPR offset, and rows
// or columns outside the matrix are simply out of range — loads return 0, stores are dropped
// (the SGPR offset takes part in the range check on gfx950: tools/ubench/buffer_oob.hip).
// Result: zero VALU instructions per load/store.  That matters more than anything else here:
// v_mfma_f32_32x32x2_f32 and fp32/int VALU instructions do NOT overlap on a SIMD
// (tools/ubench/mfma_valu_overlap.hip: time(MFMA + VALU) >= time(MFMA) + time(VALU), from the
// same wave or from different waves), so every VALU instruction in the loop is paid in full.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned kRsrcMaxBytes = 0x40000000u;          // descriptors are clamped to 1 GiB windows
constexpr int kOobOffset = 0x40000000;                   // per-lane offset of an invalid column

// `bytes` > 0 by construction at every call site (tile cursors never pass the last tile); the clamp is
// written on the unsigned high bits so that it stays on the scalar unit (s_cmp has no signed 64-bit form)
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, long long bytes) {
  const unsigned long long b = (unsigned long long)bytes;
  const unsigned n = (b >> 30) ? kRsrcMaxBytes : (unsigned)b;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ int bload_i(rsrc_t r, int voff, int soff) {
  return (int)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
}
__device__ __forceinline__ void bstore(float v, rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}


}  // namespace
