// Does fp32 VALU work overlap with v_mfma_f32_32x32x2_f32 on gfx950?
//   mode 0: MFMA only      mode 1: VALU only     mode 2: same wave, interleaved
//   mode 3: waves 0-3 MFMA, waves 4-7 VALU (two waves per SIMD, one of each)
// hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int KIND>   // KIND 0: v_fma_f32, 1: integer v_mad_u32_u24/v_add, 2: ds_read_b32, 3: v_pk_fma_f32
__global__ __launch_bounds__(512) void k(float *out, int iters, float a, float b) {
  __shared__ float lds[512];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned u[8];
  for (int i = 0; i < 8; ++i) u[i] = threadIdx.x + i;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pk[8];
  for (int i = 0; i < 8; ++i) pk[i] = f2{(float)threadIdx.x, (float)i};
  f32x16 acc0 = {0}, acc1 = {0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
  if (MODE != 3 && wave >= 4) return;                      // modes 0-2: one wave per SIMD
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
    }
    if (do_valu) {
#pragma unroll
      for (int r = 0; r < 2; ++r)                          // 16 independent FMAs per iteration (vs 2 MFMAs = 128 cycles)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (KIND == 0) v[i] = __builtin_fmaf(v[i], a, b);
          if (KIND == 1) u[i] = u[i] * 3u + (unsigned)it;
          if (KIND == 2) v[i] += lds[(threadIdx.x + ((int)v[i] & 63) + i + it) & 511];
          if (KIND == 3) pk[i] = __builtin_elementwise_fma(pk[i], f2{a, a}, f2{b, b});
        }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i] + pk[i][0] + pk[i][1];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int KIND>
float run(float *out, int iters) {
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL((k<MODE, KIND>), dim3(256), dim3(512), 0, 0, out, iters, 1.0001f, 0.9999f);
  hipEventRecord(s);
  hipLaunchKernelGGL((k<MODE, KIND>), dim3(256), dim3(512), 0, 0, out, iters, 1.0001f, 0.9999f);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  return ms;
}

template <int KIND>
void suite(const char *name, float *out, int iters) {
  printf("%-12s mfma %.3f | other %.3f | same-wave %.3f | separate-waves %.3f ms\n", name, run<0, KIND>(out, iters),
         run<1, KIND>(out, iters), run<2, KIND>(out, iters), run<3, KIND>(out, iters));
}
int main() {
  float *out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 100000;
  suite<0>("v_fma_f32", out, iters);
  suite<1>("int valu", out, iters);
  suite<2>("ds_read", out, iters);
  suite<3>("v_pk_fma_f32", out, iters);
  return 0;
}
