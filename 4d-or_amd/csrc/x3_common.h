// x3_common.h — the split-bf16 ("f32x3") product on the bf16 matrix cores, shared by x3_chain.hip.
//
// An fp32 operand is written as hi + mid + lo, three bf16 values (round to nearest each: |mid| <= 2^-9 |x|,
// |lo| <= 2^-17 |x|, remainder <= 2^-26 |x|); a product keeps the six partial products hi*hi, hi*mid, mid*hi, hi*lo,
// lo*hi, mid*mid (the dropped ones are <= 2^-24 of the term) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
// 6/16 of the matrix time of v_mfma_f32_32x32x2_f32, fp32-grade error (profiles/r05_split_bf16_gemm.jsonl: 1.0-1.5x the
// exact kernel's error against fp64).
//
// Operand fragments (both MFMA roles are symmetric): lane l holds, for row / column (l & 31), the 8 contraction indices
// kmap(chunk, l >> 5, 0..7) of a 16-wide chunk as 8 bf16 = 16 bytes.  A 32x32 accumulator tile's lane l, register r, is
// element (row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31) of D = A B.
//
// Chaining layers without a transpose: compute a hidden layer TRANSPOSED, D^T[channel][row] = W[channel][k] * X^T[k][row]
// (W fragment as the A operand, the activations as the B operand).  Lane (row, h) then holds, in registers 8q .. 8q+7 of
// tile T, the channels 32 T + 16 q + {4h + 0..3, 8 + 4h + 0..3} of ITS row — eight values of one row, i.e. exactly an
// operand fragment of the next layer for the chunk 2T + q, provided the next layer's weight fragments use the same
// (permuted) contraction order:  kmap_perm(c, h, i) = 16 c + (i < 4 ? 4h + i : 8 + 4h + i - 4).
// Activations therefore stay in registers from the gathered input rows to the pooled maximum.
#pragma once
#include "pn2_common.h"

namespace {

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));
typedef float x3_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned x3_u32x4 __attribute__((ext_vector_type(4)));

struct x3_frag {            // one operand fragment: three pieces of 8 bf16
  x3_u32x4 p[3];
};

__device__ __forceinline__ unsigned x3_pack(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x3_f32x2{a, b}, x3_bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float x3_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float x3_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// two floats -> dword `d` of the three pieces (piece p = RNE(x - sum of the pieces before it); the subtractions are exact)
__device__ __forceinline__ void x3_split_pair(float a, float b, x3_frag &f, int d) {
  const unsigned p0 = x3_pack(a, b);
  a -= x3_lo(p0); b -= x3_hi(p0);
  const unsigned p1 = x3_pack(a, b);
  a -= x3_lo(p1); b -= x3_hi(p1);
  f.p[0][d] = p0; f.p[1][d] = p1; f.p[2][d] = x3_pack(a, b);
}
__device__ __forceinline__ void x3_split8(const float (&v)[8], x3_frag &f) {
#pragma unroll
  for (int d = 0; d < 4; ++d) x3_split_pair(v[2 * d], v[2 * d + 1], f, d);
}

__device__ __forceinline__ x3_bf16x8 x3_as_bf(x3_u32x4 v) { return __builtin_bit_cast(x3_bf16x8, v); }

// acc += A * B with the six products, small ones first (their sum is formed before it meets the large one)
__device__ __forceinline__ void x3_mma(const x3_frag &a, const x3_frag &b, x3_f32x16 &acc) {
#pragma unroll
  for (int d = 2; d >= 0; --d)
#pragma unroll
    for (int i = 0; i <= d; ++i)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf(a.p[i]), x3_as_bf(b.p[d - i]), acc, 0, 0, 0);
}

// contraction index of fragment slot (chunk c, half h, element i)
__device__ __host__ __forceinline__ int x3_kmap(int c, int h, int i, bool perm) {
  return perm ? 16 * c + (i < 4 ? 4 * h + i : 8 + 4 * h + (i - 4)) : 16 * c + 8 * h + i;
}

constexpr int kX3UnitBytes = 3 * 1024;     // one (chunk, tile) weight unit: 3 pieces x 64 lanes x 16 bytes
constexpr int kX3SlotUnits = 8;            // units per LDS ring slot (24 KB): K = 128 -> one tile, K = 64 -> two, K = 32 -> four
constexpr int kX3SlotBytes = kX3SlotUnits * kX3UnitBytes;

}  // namespace
