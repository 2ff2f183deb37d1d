// fps.hip — furthest point sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel (EXT/src/sampling_gpu.cu:69-229).
// Same selection rule, different machine mapping:
//
//  * The reference streams xyz + temp from memory every round with one
//    512-thread block per cloud.  Here a cloud's points live in VGPRs for the
//    whole kernel (xyz + running min distance = 4 registers per point, up to
//    24 points per lane x 1024 lanes), so a round is pure VALU + one wave64 DPP
//    reduction + one LDS exchange + ONE barrier.
//  * The reference's shared-memory tree (:115-166) breaks ties by tree shape.
//    We reduce a 64-bit totally ordered key instead,
//        packed = (bits(dist)+1) << 32 | ~rank(k)
//        rank(k) = bitrev_L(k mod bs) << (31-L) | (k >> L),  bs = 2^L = the
//                  reference's block size opt_n_threads(N)
//    so ANY reduction order returns the reference's winner: larger distance
//    first, then the thread with the smaller bit-reversed tid (the tree keeps
//    slot idx1 unless v2 > v1, :63-64), then the smaller k inside that thread
//    (strict '>' at :108-109).  packed == 0 encodes "no candidate"
//    (best = -1, besti = 0 at :90-91).
//  * Skipped points (|p|^2 <= 1e-3, :100-101) and out-of-range slots carry a
//    running distance of -1, which can never beat best = -1.
//  * Clouds too large for the register file use the streaming kernel with the
//    running distances in a caller-provided workspace.
#include "pn2_common.h"

namespace {

// v_min_f32 without the canonicalising v_max the compiler puts in front of
// llvm.minnum for a loop-carried operand.  IEEE-mode v_min_f32 returns the
// non-NaN operand, i.e. fminf() semantics (the running distance is never NaN).
__device__ __forceinline__ float fps_min(float d, float t) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(d), "v"(t));
  return r;
}

struct __attribute__((aligned(32))) FpsSlot {
  u64 packed;
  float x, y, z;
  float pad0, pad1, pad2;
};

__device__ __forceinline__ unsigned fps_rank(unsigned k, int L) {
  // L in [0, 9]
  const unsigned rev = L ? (__brev(k) >> (32 - L)) : 0u;  // bitrev_L(k mod 2^L)
  return (rev << (31 - L)) | (k >> L);
}
__device__ __forceinline__ unsigned fps_unrank(unsigned key, int L) {
  const unsigned revt = key >> (31 - L);
  const unsigned tid = L ? (__brev(revt) >> (32 - L)) : 0u;
  const unsigned q = key & ((1u << (31 - L)) - 1u);
  return (q << L) | tid;
}
__device__ __forceinline__ u64 fps_pack(float best, unsigned k, int L) {
  return ((u64)(__float_as_uint(best) + 1u) << 32) | (u64)(unsigned)(~fps_rank(k, L));
}

// Block-level arg-max exchange.  Each wave contributes (wmax, x, y, z); returns
// the block winner (uniform) and its coordinates.  One barrier; `buf` is the
// parity-selected half of a double-buffered slot array.
template <int NW>
__device__ __forceinline__ u64 fps_block_exchange(FpsSlot *buf, u64 wmax, bool writer,
                                                  float sx, float sy, float sz,
                                                  float &ox, float &oy, float &oz) {
  const int lane = pn2_lane();
  const int wave = threadIdx.x >> 6;
  if (writer) {
    buf[wave].packed = wmax;
    buf[wave].x = sx;
    buf[wave].y = sy;
    buf[wave].z = sz;
  }
  __syncthreads();
  const int s = lane & 15;
  u64 mine = (s < NW) ? buf[s].packed : 0ull;
  const u64 gmax = pn2_readlane_u64(pn2_row16_max_u64(mine), 0);
  const u64 who = __ballot(mine == gmax && s < NW);
  const int w = __ffsll((long long)who) - 1;  // >= 0: some slot holds gmax
  const int ws = w & 15;
  ox = buf[ws].x;
  oy = buf[ws].y;
  oz = buf[ws].z;
  return gmax;
}

// ---- register-resident kernel: one workgroup per cloud ------------------------
template <int BS, int PPT>
__global__ __launch_bounds__(BS) void fps_resident_kernel(int N, int m, int L,
                                                         const float *__restrict__ xyz,
                                                         int *__restrict__ idxs) {
  constexpr int NW = BS / 64;
  __shared__ FpsSlot slots[2][16];

  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int lane = pn2_lane();
  const float *P = xyz + (size_t)b * N * 3;
  int *out = idxs + (size_t)b * m;

  float px[PPT], py[PPT], pz[PPT], td[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = t + i * BS;
    float x = 0.f, y = 0.f, z = 0.f;
    bool valid = false;
    if (k < N) {
      x = P[(size_t)k * 3 + 0];
      y = P[(size_t)k * 3 + 1];
      z = P[(size_t)k * 3 + 2];
      const float mag = pn2_sq3(x, y, z);
      valid = !((double)mag <= 1e-3);
    }
    px[i] = x; py[i] = y; pz[i] = z;
    td[i] = valid ? 1e10f : -1.f;
  }

  const float p0x = P[0], p0y = P[1], p0z = P[2];
  float ox = p0x, oy = p0y, oz = p0z;  // old = 0
  if (t == 0) out[0] = 0;

  for (int j = 1; j < m; ++j) {
    float best = -1.f;
    int bi = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float d = pn2_sq3(px[i] - ox, py[i] - oy, pz[i] - oz);
      const float d2 = fps_min(d, td[i]);
      td[i] = d2;
      if (d2 > best) { best = d2; bi = i; }
    }
    u64 pk = 0ull;
    if (best >= 0.f) pk = fps_pack(best, (unsigned)(t + bi * BS), L);
    const u64 wmax = pn2_wave_max_u64(pk);

    float sx = p0x, sy = p0y, sz = p0z;
    bool writer;
    if (wmax == 0ull) {
      writer = (lane == 0);
    } else {
      writer = (pk == wmax);
      if (writer) {
        sx = px[0]; sy = py[0]; sz = pz[0];
#pragma unroll
        for (int i = 1; i < PPT; ++i) {
          if (bi == i) { sx = px[i]; sy = py[i]; sz = pz[i]; }
        }
      }
    }
    const u64 gmax = fps_block_exchange<NW>(slots[j & 1], wmax, writer, sx, sy, sz, ox, oy, oz);
    if (t == 0) out[j] = gmax ? (int)fps_unrank(~(unsigned)gmax, L) : 0;
  }
}

// ---- streaming kernel: any N, running distances in global scratch --------------
// One 1024-thread workgroup per cloud; thread t visits k = t, t+1024, ... in
// ascending order (1024 is a multiple of every reference block size, so a
// thread's points share one reference tid and ascending k == ascending rank).
template <int BS>
__global__ __launch_bounds__(BS) void fps_stream_kernel(int N, int m, int L,
                                                       const float *__restrict__ xyz,
                                                       float *__restrict__ temp,
                                                       int *__restrict__ idxs) {
  constexpr int NW = BS / 64;
  __shared__ FpsSlot slots[2][16];
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int lane = pn2_lane();
  const float *P = xyz + (size_t)b * N * 3;
  float *T = temp + (size_t)b * N;
  int *out = idxs + (size_t)b * m;

  const float p0x = P[0], p0y = P[1], p0z = P[2];
  float ox = p0x, oy = p0y, oz = p0z;
  if (t == 0) out[0] = 0;

  for (int j = 1; j < m; ++j) {
    float best = -1.f;
    int bk = 0;
    float bx = p0x, by = p0y, bz = p0z;
    if (j == 1) {
      for (int k = t; k < N; k += BS) {
        const float x = P[(size_t)k * 3 + 0], y = P[(size_t)k * 3 + 1], z = P[(size_t)k * 3 + 2];
        const float mag = pn2_sq3(x, y, z);
        const bool valid = !((double)mag <= 1e-3);
        const float d = pn2_sq3(x - ox, y - oy, z - oz);
        const float d2 = valid ? fminf(d, 1e10f) : -1.f;
        T[k] = d2;
        if (d2 > best) { best = d2; bk = k; bx = x; by = y; bz = z; }
      }
    } else {
#pragma unroll 4
      for (int k = t; k < N; k += BS) {
        const float x = P[(size_t)k * 3 + 0], y = P[(size_t)k * 3 + 1], z = P[(size_t)k * 3 + 2];
        const float d = pn2_sq3(x - ox, y - oy, z - oz);
        const float d2 = fps_min(d, T[k]);
        T[k] = d2;
        if (d2 > best) { best = d2; bk = k; bx = x; by = y; bz = z; }
      }
    }
    u64 pk = 0ull;
    if (best >= 0.f) pk = fps_pack(best, (unsigned)bk, L);
    const u64 wmax = pn2_wave_max_u64(pk);
    bool writer;
    if (wmax == 0ull) {
      writer = (lane == 0);
      bx = p0x; by = p0y; bz = p0z;
    } else {
      writer = (pk == wmax);
    }
    const u64 gmax = fps_block_exchange<NW>(slots[j & 1], wmax, writer, bx, by, bz, ox, oy, oz);
    if (t == 0) out[j] = gmax ? (int)fps_unrank(~(unsigned)gmax, L) : 0;
  }
}

constexpr int kFpsResidentMaxN = 1024 * 24;

// EXT/include/cuda_utils.h:15-19 (same truncating double-log expression).
int ref_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int v = 1 << pow_2;
  if (v > 512) v = 512;
  if (v < 1) v = 1;
  return v;
}

}  // namespace

extern "C" size_t pn2_fps_workspace_bytes(int B, int N, int m) {
  if (B <= 0 || N <= 0 || m <= 1) return 0;
  if (N <= kFpsResidentMaxN) return 0;
  return (size_t)B * (size_t)N * sizeof(float);
}

extern "C" int pn2_furthest_point_sampling(int B, int N, int m, const float *xyz,
                                           void *workspace, size_t workspace_bytes,
                                           int *idxs, void *stream) {
  if (B < 0 || N < 0) return PN2_EINVAL;
  if (m <= 0 || B == 0) return PN2_OK;  // EXT/src/sampling_gpu.cu:73
  if (N <= 0) return PN2_EINVAL;
  if (!xyz || !idxs) return PN2_ENULL;
  hipStream_t s = (hipStream_t)stream;
  const int bs = ref_opt_n_threads(N);
  int L = 0;
  while ((1 << L) < bs) ++L;

#define PN2_FPS_LAUNCH(BS, PPT) \
  hipLaunchKernelGGL((fps_resident_kernel<BS, PPT>), dim3(B), dim3(BS), 0, s, N, m, L, xyz, idxs)
  if (N <= 512) PN2_FPS_LAUNCH(512, 1);
  else if (N <= 1024) PN2_FPS_LAUNCH(512, 2);
  else if (N <= 2048) PN2_FPS_LAUNCH(512, 4);
  else if (N <= 4096) PN2_FPS_LAUNCH(512, 8);
  else if (N <= 8192) PN2_FPS_LAUNCH(1024, 8);
  else if (N <= 16384) PN2_FPS_LAUNCH(1024, 16);
  else if (N <= kFpsResidentMaxN) PN2_FPS_LAUNCH(1024, 24);
  else {
    const size_t need = (size_t)B * (size_t)N * sizeof(float);
    if (!workspace) return PN2_ENULL;
    if (workspace_bytes < need) return PN2_ENOSPC;
    hipLaunchKernelGGL((fps_stream_kernel<1024>), dim3(B), dim3(1024), 0, s, N, m, L, xyz,
                       (float *)workspace, idxs);
  }
#undef PN2_FPS_LAUNCH
  return pn2_check_launch();
}
