// Is the SGPR offset of a raw buffer access part of the range check on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float *buf, float *out, int nrec_bytes, int soff) {
  auto rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, nrec_bytes, 0x00020000);
  const int l = threadIdx.x;
  // load: voffset in range, soffset pushes it out
  out[l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, l * 4, soff, 0));
  // store 777 at voffset l*4 + soffset: only in-range lanes may land
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, 777.0f), rs, l * 4, soff, 0);
}
int main() {
  float *buf, *out; hipMalloc(&buf, 4096); hipMalloc(&out, 256);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  hipMemcpy(buf, h, 4096, hipMemcpyHostToDevice);
  // 16 records (64 B); soffset 32 B: lanes 0..7 in range (offsets 32..60), lanes 8..15 have voffset < 64 but voffset+soffset >= 64
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, buf, out, 64, 32);
  float o[64]; hipMemcpy(o, out, 256, hipMemcpyDeviceToHost); hipMemcpy(h, buf, 4096, hipMemcpyDeviceToHost);
  printf("loads : "); for (int i = 0; i < 20; ++i) printf("%g ", o[i]); printf("\n");
  printf("memory: "); for (int i = 0; i < 32; ++i) printf("%g ", h[i]); printf("\n");
  return 0;
}
