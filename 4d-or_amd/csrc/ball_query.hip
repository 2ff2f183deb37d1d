// ball_query.hip — radius search for gfx950.
//
// Replaces query_ball_point_kernel (EXT/src/ball_query_gpu.cu:9-54): for each
// centre, the first `nsample` points IN ASCENDING INDEX ORDER with d^2 < r^2
// (strict), the row padded with the first hit, a zero row when the ball is
// empty.
//
// Machine mapping.  The reference gives each centre ONE thread that walks all N
// points serially (divergent early exit, uncoalesced).  Here a wave64 owns CPW
// centres and walks the cloud 64 points at a time: lane l holds point base+l
// (one coalesced 12-byte load per lane, prefetched one tile ahead), tests it
// against each of the wave's centres (centre coordinates and hit counters are
// wave-uniform => SGPRs), and `__ballot` + `mbcnt` turn the 64 hit flags into
// ordered output slots, so index order is preserved without any sorting.  The
// scan stops as soon as every centre of the wave has `nsample` hits, which for
// typical radii is a small fraction of N.
#include "pn2_common.h"

#include <stdlib.h>

namespace {

template <int CPW>
__global__ __launch_bounds__(256) void ball_query_kernel(int N, int m, int bpc, float r2, int ns,
                                                        const float *__restrict__ new_xyz,
                                                        const float *__restrict__ xyz,
                                                        int *__restrict__ idx) {
  const int b = blockIdx.x / bpc;  // bpc = workgroups per cloud
  const int blk = blockIdx.x - b * bpc;
  const int lane = pn2_lane();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j0 = (blk * 4 + wave) * CPW;
  if (j0 >= m) return;  // wave-uniform; the kernel has no barriers

  const float *P = xyz + (size_t)b * N * 3;
  const float *Q = new_xyz + (size_t)b * m * 3;
  int *out = idx + (size_t)b * m * ns;

  float cx[CPW], cy[CPW], cz[CPW];
  int cnt[CPW], first[CPW];
  int remaining = 0;
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int j = j0 + c;
    const bool live = j < m;
    const int jj = live ? j : j0;
    cx[c] = Q[(size_t)jj * 3 + 0];
    cy[c] = Q[(size_t)jj * 3 + 1];
    cz[c] = Q[(size_t)jj * 3 + 2];
    cnt[c] = live ? 0 : ns;  // dead slots look "full"
    first[c] = 0;
    remaining += (live && ns > 0) ? 1 : 0;
  }

  // software prefetch: tile (base) is consumed while tile (base+64) is in flight
  float x = 0.f, y = 0.f, z = 0.f;
  if (lane < N) {
    x = P[(size_t)lane * 3 + 0];
    y = P[(size_t)lane * 3 + 1];
    z = P[(size_t)lane * 3 + 2];
  }
  for (int base = 0; base < N && remaining > 0; base += 64) {
    const int k = base + lane;
    const int kn = k + 64;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (kn < N) {
      nx = P[(size_t)kn * 3 + 0];
      ny = P[(size_t)kn * 3 + 1];
      nz = P[(size_t)kn * 3 + 2];
    }
    const bool in = k < N;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      if (cnt[c] < ns) {  // wave-uniform
        const float d2 = pn2_sq3(cx[c] - x, cy[c] - y, cz[c] - z);
        const bool hit = in && (d2 < r2);
        const u64 mask = __ballot(hit);
        if (mask) {
          if (cnt[c] == 0) first[c] = base + (__ffsll((long long)mask) - 1);
          const int pos = cnt[c] + pn2_prefix_popc(mask);
          if (hit && pos < ns) out[(size_t)(j0 + c) * ns + pos] = k;
          cnt[c] += __popcll(mask);
          if (cnt[c] >= ns) --remaining;
        }
      }
    }
    x = nx; y = ny; z = nz;
  }

  // pad with the first hit (or 0 for an empty ball): EXT/src/ball_query_gpu.cu:34-38
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    if (j0 + c < m) {
      const int have = cnt[c] < ns ? cnt[c] : ns;
      const int fill = cnt[c] > 0 ? first[c] : 0;
      for (int s = have + lane; s < ns; s += 64) out[(size_t)(j0 + c) * ns + s] = fill;
    }
  }
}

}  // namespace

extern "C" int pn2_ball_query(int B, int N, int m, float radius, int nsample,
                              const float *new_xyz, const float *xyz, int *idx,
                              void *stream) {
  if (B < 0 || N < 0 || m < 0 || nsample < 0) return PN2_EINVAL;
  if (B == 0 || m == 0 || nsample == 0) return PN2_OK;
  if (!new_xyz || !idx || (N > 0 && !xyz)) return PN2_ENULL;
  hipStream_t s = (hipStream_t)stream;
  const float r2 = radius * radius;  // fp32, EXT/src/ball_query_gpu.cu:22
  const long long centres = (long long)B * m;
  // enough waves to fill 256 CUs x 8 waves/SIMD first, then amortise loads
  // measured at 32 x 50k / 2048 centres / ns 64: CPW 1/2/4/8 -> 0.50/0.46/0.36/0.43 ms (8 amortises loads
  // best but its waves wait for their slowest centre)
  int cpw = 1;
  if (centres >= 8192 * 4) cpw = 4;
  else if (centres >= 8192 * 2) cpw = 2;
  const int per_block = 4 * cpw;
  const int bpc = (m + per_block - 1) / per_block;
  if ((long long)bpc * B > 0x7fffffffLL) return PN2_EINVAL;
  dim3 grid((unsigned)(bpc * B));
  switch (cpw) {
    case 8: hipLaunchKernelGGL((ball_query_kernel<8>), grid, dim3(256), 0, s, N, m, bpc, r2, nsample, new_xyz, xyz, idx); break;
    case 4: hipLaunchKernelGGL((ball_query_kernel<4>), grid, dim3(256), 0, s, N, m, bpc, r2, nsample, new_xyz, xyz, idx); break;
    case 2: hipLaunchKernelGGL((ball_query_kernel<2>), grid, dim3(256), 0, s, N, m, bpc, r2, nsample, new_xyz, xyz, idx); break;
    default: hipLaunchKernelGGL((ball_query_kernel<1>), grid, dim3(256), 0, s, N, m, bpc, r2, nsample, new_xyz, xyz, idx); break;
  }
  return pn2_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------
// sample_uniformly / ret_unique_cnt of the Group-Free-3D QueryAndGroup (GF3D/pointnet2/pointnet2_utils.py:327-336).
// The reference walks every (batch, region) on the HOST: torch.unique of the ball-query row, torch.randint over the
// unique entries for the padded tail, one tiny tensor op at a time.  A ball-query row is already "unique entries in
// ascending order, then padding with the first hit", so the unique set is the strictly ascending prefix and its length
// is the count; the padded tail is refilled with uniformly drawn members of that prefix.  One wave per row, a
// counter-based generator (seed, row, slot) instead of the host's Mersenne stream: same distribution, different draws.
namespace {

__device__ __forceinline__ unsigned pn2_mix32(unsigned seed, unsigned row, unsigned slot) {
  unsigned h = seed ^ (row * 0x9E3779B9u) ^ (slot * 0x85EBCA6Bu);
  h ^= h >> 16; h *= 0x7FEB352Du;
  h ^= h >> 15; h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}

__global__ __launch_bounds__(256) void unique_resample_kernel(long long rows, int ns, unsigned seed,
                                                             int *__restrict__ idx, float *__restrict__ unique_cnt) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = pn2_lane();
  int *r = idx + row * ns;
  int cnt = 0;
  for (int base = 0; base < ns; base += 64) {
    const int s = base + lane;
    bool uniq = false;
    if (s < ns) uniq = (s == 0) || (r[s] > r[s - 1]);
    const u64 mask = __ballot(uniq);
    // the unique entries are a prefix: stop counting at the first non-ascending slot
    const u64 inv = ~mask & ((ns - base) >= 64 ? ~0ull : ((1ull << (ns - base)) - 1ull));
    if (inv) { cnt += __ffsll((long long)inv) - 1; break; }
    cnt += __popcll(mask);
  }
  if (unique_cnt && lane == 0) unique_cnt[row] = (float)cnt;
  for (int s = cnt + lane; s < ns; s += 64)
    r[s] = r[pn2_mix32(seed, (unsigned)row, (unsigned)s) % (unsigned)cnt];
}

}  // namespace

extern "C" int pn2_ball_query_unique_resample(long long rows, int nsample, unsigned seed, int *idx, float *unique_cnt,
                                              void *stream) {
  if (rows < 0 || nsample < 0) return PN2_EINVAL;
  if (rows == 0 || nsample == 0) return PN2_OK;
  if (!idx) return PN2_ENULL;
  const long long blocks = (rows + 3) / 4;
  if (blocks > 0x7fffffffLL) return PN2_EINVAL;
  hipLaunchKernelGGL(unique_resample_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, nsample, seed,
                     idx, unique_cnt);
  return pn2_check_launch();
}
