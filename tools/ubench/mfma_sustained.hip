// What does the matrix pipe SUSTAIN on this part?  A register-only MFMA loop (no memory, no LDS, no VALU in the loop) on
// 8 .. 256 workgroups, one or two waves per SIMD, for v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16.
// Prints TFLOP/s and the clock that rate implies (rate / (active SIMDs x flop per SIMD-cycle)).  If the implied clock falls
// as more CUs are lit, the ceiling is the power / current limiter, not the instruction's issue rate.
//   hipcc --offload-arch=gfx950 -O3 mfma_sustained.hip -o mfma_sustained && ./mfma_sustained
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: f32 32x32x2, 1: bf16 32x32x16
__global__ __launch_bounds__(512) void k(float *out, const float *in, int iters) {
  // operands from memory so that they are not compile-time constants; four independent accumulators
  const float a0 = in[threadIdx.x], a1 = in[threadIdx.x + 512], b0 = in[threadIdx.x + 1024], b1 = in[threadIdx.x + 1536];
  bf16x8 pa, pb;
  for (int i = 0; i < 8; ++i) { pa[i] = (__bf16)in[(threadIdx.x * 8 + i) & 2047]; pb[i] = (__bf16)in[(threadIdx.x * 8 + i + 77) & 2047]; }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pb, pa, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pa, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pb, pb, c3, 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, float *out, const float *in, int wgs, int threads, int iters) {
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL((k<KIND>), dim3(wgs), dim3(threads), 0, 0, out, in, iters);      // warm-up (clock ramp)
  hipEventRecord(s);
  hipLaunchKernelGGL((k<KIND>), dim3(wgs), dim3(threads), 0, 0, out, in, iters);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double flop_per_mfma = KIND == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
  const double flop_per_simd_cycle = KIND == 0 ? 64.0 : 1024.0;
  const double waves = (double)wgs * threads / 64;
  const double flops = waves * iters * 4.0 * flop_per_mfma;
  const double tf = flops / (ms * 1e-3) / 1e12;
  const double simds = (double)wgs * 4;                       // one workgroup per CU (grid <= 256), all four SIMDs lit
  printf("{\"mfma\": \"%s\", \"workgroups\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"TFLOPps\": %.1f, \"implied_clock_GHz\": %.3f}\n",
         name, wgs, threads / 256, ms, tf, tf * 1e12 / (simds * flop_per_simd_cycle) / 1e9);
}

int main() {
  float *out, *in;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&in, 2048 * 4);
  float h[2048];
  for (int i = 0; i < 2048; ++i) h[i] = 0.5f + (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int wgs[] = {8, 32, 64, 128, 256};
  for (int w : wgs) run<0>("f32_32x32x2", out, in, w, 256, 200000);
  run<0>("f32_32x32x2", out, in, 256, 512, 100000);
  for (int w : wgs) run<1>("bf16_32x32x16", out, in, w, 256, 200000);
  run<1>("bf16_32x32x16", out, in, 256, 512, 100000);
  return 0;
}
