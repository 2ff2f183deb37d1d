// ds_add_f32 (no return) throughput on gfx950: conflict-free rows, 64 lanes = 64 consecutive columns.
//   mode 0: ds_add_f32 to a pseudo-random row per instruction (what a sparse row scatter does)
//   mode 1: ds_read_b32 + v_fma + ds_write_b32 of the same element (non-atomic read-modify-write)
//   mode 2: ds_write_b32 only (reference)
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic.hip -o lds_atomic && ./lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const int *rows, int iters, int nrows) {
  extern __shared__ float lds[];   // [nrows][64]
  for (int i = threadIdx.x; i < nrows * 64; i += 256) lds[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float w = 1.0f + lane * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
      const int r = __builtin_amdgcn_readfirstlane(rows[(it * 32 + j + wave * 7) & 1023]);
      float *p = &lds[r * 64 + lane];
      if (MODE == 0) __hip_atomic_fetch_add(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 1) *p = __builtin_fmaf(w, 1.5f, *p);
      if (MODE == 2) *p = w;
    }
  }
  __syncthreads();
  float s = 0;
  for (int i = threadIdx.x; i < nrows * 64; i += 256) s += lds[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, float *out, const int *rows, int wgs_per_cu) {
  const int iters = 2000, nrows = 64;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * wgs_per_cu), dim3(256), nrows * 64 * 4, 0, out, rows, iters, nrows);
  hipEventRecord(s);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * wgs_per_cu), dim3(256), nrows * 64 * 4, 0, out, rows, iters, nrows);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double instr_per_cu = (double)iters * 32 * 4 * wgs_per_cu;   // wave-instructions per CU
  printf("%-28s wgs/cu %d: %.3f ms  -> %.1f clk per wave-instruction per CU (at 2.4 GHz)\n", name, wgs_per_cu, ms,
         ms * 1e-3 * 2.4e9 / instr_per_cu);
}

int main() {
  float *out; hipMalloc(&out, 256 * 4 * 256 * 4);
  int h[1024]; unsigned x = 12345;
  for (int i = 0; i < 1024; ++i) { x = x * 1664525u + 1013904223u; h[i] = (x >> 16) & 63; }
  int *rows; hipMalloc(&rows, sizeof(h)); hipMemcpy(rows, h, sizeof(h), hipMemcpyHostToDevice);
  for (int w = 1; w <= 4; w *= 2) {
    run<0>("ds_add_f32", out, rows, w);
    run<1>("ds_read+fma+ds_write", out, rows, w);
    run<2>("ds_write_b32", out, rows, w);
  }
  return 0;
}
