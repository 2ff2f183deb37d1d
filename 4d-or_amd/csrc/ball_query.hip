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

// ---------------------------------------------------------------------------------------------------------------
// Cell-list ("uniform grid") ball query.  The brute-force kernel above walks the cloud in index order and stops after
// `nsample` hits, which is efficient when balls are crowded (a 0.2-ball in a 50k-point room holds ~400 points: the scan
// ends after ~8k points).  It is hopeless when they are not: the scene-graph encoders query r = 0.1 / 0.2 balls in
// 4000 / 8000-point object clouds, a ball holds 4-30 points < nsample, and EVERY centre scans the WHOLE cloud
// (9.4e9 distance tests per 32-scan step).  Here each cloud is binned once into cells of edge >= r (G^3 cells over its
// bounding box), and a centre only tests the points of its 27 neighbouring cells:
//   build : one workgroup per cloud — bounding box, LDS histogram, LDS scan, scatter of (x, y, z, index) records
//           (order inside a cell does not matter, see below);
//   query : one wave per centre — the 27 cell ranges are flattened into one candidate list (lane-parallel, binary search
//           in the 27 prefix sums), hits are ballot-compacted into LDS, and because they arrive in cell order, not index
//           order, they are RANK-SORTED by index (each lane counts how many hits are smaller than its own), which yields
//           exactly the reference's "first nsample hits in ascending index" + first-hit padding.
//   A centre with more than kGridCap hits (a crowded ball — where brute force with early exit is the better algorithm)
//   falls back to the index-order scan inside the same kernel.  Same distance expression, same strict '<': bit-exact.
constexpr int kGridCap = 256;          // hits a wave collects before it prefers the index-order scan
constexpr int kGridMaxG = 16;          // cells per axis (<= 4096 cells: the histogram lives in LDS)
constexpr int kGridHdr = 16;           // ints per cloud header: min[3], inv_cs[3], dims[3], cells, pad

struct GridHdr {
  float mn[3];
  float inv[3];
  int dim[3];
  int cells;
  int pad[6];
};
static_assert(sizeof(GridHdr) == kGridHdr * 4, "header layout");

__device__ __forceinline__ int grid_coord(float v, float mn, float inv, int dim) {
  // comparisons first: a NaN / infinite coordinate lands in cell 0 resp. the last one without an undefined float -> int cast
  const float f = floorf((v - mn) * inv);
  return f >= 0.f ? (f < (float)dim ? (int)f : dim - 1) : 0;
}

// workspace per cloud: GridHdr | cell_start[kGridMaxG^3 + 1] | records[N] (float4: x, y, z, bits(index))
__global__ __launch_bounds__(1024) void bq_grid_build_kernel(int N, int G, float r, const float *__restrict__ xyz,
                                                            int *__restrict__ hdrs, int *__restrict__ starts,
                                                            float4 *__restrict__ recs) {
  __shared__ int cnt[kGridMaxG * kGridMaxG * kGridMaxG];
  __shared__ float red[6][16];
  __shared__ GridHdr h;
  __shared__ int wsum[16];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float *P = xyz + (size_t)b * N * 3;
  float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int k = t; k < N; k += 1024) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float v = P[(size_t)k * 3 + d];
      lo[d] = fminf(lo[d], v);
      hi[d] = fmaxf(hi[d], v);
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
      hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
    }
    if (lane == 0) { red[d][wave] = lo[d]; red[3 + d][wave] = hi[d]; }
  }
  __syncthreads();
  if (t == 0) {
    int cells = 1;
    for (int d = 0; d < 3; ++d) {
      float mn = red[d][0], mx = red[3 + d][0];
      for (int w = 1; w < 16; ++w) { mn = fminf(mn, red[d][w]); mx = fmaxf(mx, red[3 + d][w]); }
      if (!(mx >= mn)) { mn = 0.f; mx = 0.f; }               // N == 0 / NaN coordinates: one cell
      float ext = mx - mn;
      if (!(ext < 3.0e38f)) { mn = 0.f; ext = 0.f; }          // infinite extent: one cell along this axis (inv = 0 below)
      const float cs = fmaxf(r * 1.0001f, ext / (float)G);      // cell edge >= r (strictly, against rounding)
      int dim = cs > 0.f ? (int)floorf(ext / cs) + 1 : 1;
      if (dim > G) dim = G;
      if (dim < 1) dim = 1;
      h.mn[d] = mn;
      h.inv[d] = (cs > 0.f && dim > 1) ? 1.0f / cs : 0.f;
      h.dim[d] = dim;
      cells *= dim;
    }
    h.cells = cells;
  }
  for (int c = t; c < kGridMaxG * kGridMaxG * kGridMaxG; c += 1024) cnt[c] = 0;
  __syncthreads();
  const int dx = h.dim[0], dy = h.dim[1], cells = h.cells;
  for (int k = t; k < N; k += 1024) {
    const int cx = grid_coord(P[(size_t)k * 3 + 0], h.mn[0], h.inv[0], h.dim[0]);
    const int cy = grid_coord(P[(size_t)k * 3 + 1], h.mn[1], h.inv[1], h.dim[1]);
    const int cz = grid_coord(P[(size_t)k * 3 + 2], h.mn[2], h.inv[2], h.dim[2]);
    atomicAdd(&cnt[(cz * dy + cy) * dx + cx], 1);
  }
  __syncthreads();
  // exclusive scan of cnt[0..cells) (4 cells per thread, wave scan, wave totals)
  int *st = starts + (size_t)b * (kGridMaxG * kGridMaxG * kGridMaxG + 1);
  int v[4], run = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int c = t * 4 + i; v[i] = c < cells ? cnt[c] : 0; run += v[i]; }
  int inc = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(inc, o); if (lane >= o) inc += n; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  int ex = base + inc - run;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = t * 4 + i;
    if (c < cells) { st[c] = ex; cnt[c] = ex; }              // cnt becomes the scatter cursor
    ex += v[i];
  }
  if (t == 0) {
    st[cells] = N;
    *(GridHdr *)(hdrs + (size_t)b * kGridHdr) = h;
  }
  __syncthreads();
  float4 *R = recs + (size_t)b * N;
  for (int k = t; k < N; k += 1024) {
    const float x = P[(size_t)k * 3 + 0], y = P[(size_t)k * 3 + 1], z = P[(size_t)k * 3 + 2];
    const int cx = grid_coord(x, h.mn[0], h.inv[0], h.dim[0]);
    const int cy = grid_coord(y, h.mn[1], h.inv[1], h.dim[1]);
    const int cz = grid_coord(z, h.mn[2], h.inv[2], h.dim[2]);
    const int pos = atomicAdd(&cnt[(cz * dy + cy) * dx + cx], 1);
    R[pos] = make_float4(x, y, z, __int_as_float(k));
  }
}

__global__ __launch_bounds__(256) void bq_grid_query_kernel(int N, int m, float r2, int ns,
                                                           const float *__restrict__ new_xyz,
                                                           const float *__restrict__ xyz, const int *__restrict__ hdrs,
                                                           const int *__restrict__ starts,
                                                           const float4 *__restrict__ recs, int *__restrict__ idx,
                                                           long long centres) {
  __shared__ int s_pref[4][32];
  __shared__ int s_beg[4][32];
  __shared__ int s_hit[4][kGridCap];
  const int lane = pn2_lane();
  const int wv = threadIdx.x >> 6;
  const long long j = (long long)blockIdx.x * 4 + wv;
  if (j >= centres) return;                                     // wave-uniform; no block barriers below
  const int b = (int)(j / m);
  const GridHdr *h = (const GridHdr *)(hdrs + (size_t)b * kGridHdr);
  const float qx = new_xyz[j * 3 + 0], qy = new_xyz[j * 3 + 1], qz = new_xyz[j * 3 + 2];
  int *out = idx + j * ns;
  const int *st = starts + (size_t)b * (kGridMaxG * kGridMaxG * kGridMaxG + 1);
  const float4 *R = recs + (size_t)b * N;
  const int dx = h->dim[0], dy = h->dim[1], dz = h->dim[2];

  // the 27 neighbour cells: lane l < 27 -> (ox, oy, oz) in {-1, 0, 1}^3
  int beg = 0, cnt = 0;
  {
    const float fx = floorf((qx - h->mn[0]) * h->inv[0]), fy = floorf((qy - h->mn[1]) * h->inv[1]),
                fz = floorf((qz - h->mn[2]) * h->inv[2]);
    const int cx = (int)fminf(fmaxf(fx, -2.f), (float)dx + 1.f), cy = (int)fminf(fmaxf(fy, -2.f), (float)dy + 1.f),
              cz = (int)fminf(fmaxf(fz, -2.f), (float)dz + 1.f);
    if (lane < 27) {
      const int x = cx + lane % 3 - 1, y = cy + (lane / 3) % 3 - 1, z = cz + lane / 9 - 1;
      if (x >= 0 && x < dx && y >= 0 && y < dy && z >= 0 && z < dz) {
        const int c = (z * dy + y) * dx + x;
        beg = st[c];
        cnt = st[c + 1] - beg;
      }
    }
  }
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) { const int n = __shfl_up(inc, o); if (lane >= o) inc += n; }
  if (lane < 32) { s_pref[wv][lane] = inc - cnt; s_beg[wv][lane] = beg; }   // exclusive prefix, cell begin
  __builtin_amdgcn_wave_barrier();                                           // same-wave LDS hand-off (in-order DS queue)
  const int total = __builtin_amdgcn_readlane(inc, 31);

  int have = 0;
  bool overflow = false;
  for (int t0 = 0; t0 < total; t0 += 64) {
    const int t = t0 + lane;
    bool hit = false;
    int id = 0;
    if (t < total) {
      int c = 0;                                                // largest c with pref[c] <= t  (pref is non-decreasing)
#pragma unroll
      for (int step = 16; step > 0; step >>= 1)
        if (c + step < 27 && s_pref[wv][c + step] <= t) c += step;
      const float4 p = R[s_beg[wv][c] + (t - s_pref[wv][c])];
      const float d2 = pn2_sq3(qx - p.x, qy - p.y, qz - p.z);
      hit = d2 < r2;
      id = __float_as_int(p.w);
    }
    const u64 mask = __ballot(hit);
    const int pos = have + pn2_prefix_popc(mask);
    if (hit && pos < kGridCap) s_hit[wv][pos] = id;
    have += __popcll(mask);
    if (have > kGridCap) { overflow = true; break; }
  }

  if (overflow) {
    // crowded ball: the index-order scan with early exit (same loop as ball_query_kernel, one centre per wave)
    const float *P = xyz + (size_t)b * N * 3;
    int got = 0, first = 0;
    for (int base = 0; base < N && got < ns; base += 64) {
      const int k = base + lane;
      bool hit = false;
      if (k < N) {
        const float d2 = pn2_sq3(qx - P[(size_t)k * 3 + 0], qy - P[(size_t)k * 3 + 1], qz - P[(size_t)k * 3 + 2]);
        hit = d2 < r2;
      }
      const u64 mask = __ballot(hit);
      if (mask) {
        if (got == 0) first = base + (__ffsll((long long)mask) - 1);
        const int pos = got + pn2_prefix_popc(mask);
        if (hit && pos < ns) out[pos] = k;
        got += __popcll(mask);
      }
    }
    const int fill = got > 0 ? first : 0;
    for (int s = (got < ns ? got : ns) + lane; s < ns; s += 64) out[s] = fill;
    return;
  }

  // rank sort of the `have` collected indices (unique): rank = number of hits with a smaller index
  __builtin_amdgcn_wave_barrier();
  int mine[kGridCap / 64], rank[kGridCap / 64];
  int mn = 0x7fffffff;
#pragma unroll
  for (int q = 0; q < kGridCap / 64; ++q) {
    const int e = q * 64 + lane;
    mine[q] = e < have ? s_hit[wv][e] : 0x7fffffff;
    rank[q] = 0;
    mn = mine[q] < mn ? mine[q] : mn;
  }
  for (int e = 0; e < have; ++e) {
    const int v = s_hit[wv][e];                                  // LDS broadcast
#pragma unroll
    for (int q = 0; q < kGridCap / 64; ++q) rank[q] += v < mine[q] ? 1 : 0;
  }
#pragma unroll
  for (int q = 0; q < kGridCap / 64; ++q)
    if (q * 64 + lane < have && rank[q] < ns) out[rank[q]] = mine[q];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int n = __shfl_xor(mn, o); mn = n < mn ? n : mn; }
  const int fill = have > 0 ? mn : 0;
  for (int s = (have < ns ? have : ns) + lane; s < ns; s += 64) out[s] = fill;
}

int grid_cells_per_axis(int N) {
  int g = (int)lround(cbrt((double)N / 8.0));
  if (g < 4) g = 4;
  if (g > kGridMaxG) g = kGridMaxG;
  return g;
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

// Workspace of the cell-list path: 0 = this shape runs the plain index-order scan (small clouds, huge nsample).
// Which algorithm: the cell list pays when balls are SPARSE.  The host cannot see the coordinates, so it estimates the
// hits per ball for a cloud that fills the unit ball (the 4D-OR clouds are normalised that way, zero_mean of
// data_preparation_utils.py:12-18): E = N r^3.  Measured (tools/microbench.py): E = 8 and 64 with nsample 16 / 32 —
// cell list 1.5-4x faster; E = 130 / nsample 32 and E = 400 / nsample 64 — the early-exit scan is faster.  Results are
// identical either way.
extern "C" size_t pn2_ball_query_workspace_bytes(int B, int N, int m, float radius, int nsample) {
  if (B <= 0 || m <= 0 || N < 2048 || nsample <= 0 || nsample > kGridCap || !(radius > 0.f)) return 0;
  if ((double)N * radius * radius * radius > 4.0 * nsample) return 0;
  return pn2_ball_query_grid_bytes(B, N, nsample);
}

// Raw requirement of the cell-list kernels (0: shape not covered — nsample beyond the collection cap); a caller that
// passes this much workspace to pn2_ball_query_ws gets the cell list whatever the density estimate says.
extern "C" size_t pn2_ball_query_grid_bytes(int B, int N, int nsample) {
  if (B <= 0 || N <= 0 || nsample <= 0 || nsample > kGridCap) return 0;
  const size_t per_cloud = (size_t)kGridHdr * 4 + (size_t)(kGridMaxG * kGridMaxG * kGridMaxG + 1) * 4 + (size_t)N * 16;
  return (size_t)B * per_cloud + 256;
}

extern "C" int pn2_ball_query_ws(int B, int N, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                                 int *idx, void *workspace, size_t workspace_bytes, void *stream) {
  if (B < 0 || N < 0 || m < 0 || nsample < 0) return PN2_EINVAL;
  const size_t need = pn2_ball_query_grid_bytes(B, N, nsample);
  // the cell edge is sized from the radius: a negative / non-finite radius (r*r is still a valid threshold for the
  // scan) would make cells smaller than the ball and the 27-cell search miss hits — such calls take the scan
  if (need == 0 || !workspace || workspace_bytes == 0 || m == 0 || !(radius > 0.f) || !(radius < 3.0e38f))
    return pn2_ball_query(B, N, m, radius, nsample, new_xyz, xyz, idx, stream);
  if (workspace_bytes < need) return PN2_ENOSPC;
  if (!new_xyz || !idx || !xyz) return PN2_ENULL;
  if (((uintptr_t)workspace & 15) != 0) return PN2_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // records first (16-byte aligned), then the cell starts, then the headers
  float4 *recs = (float4 *)workspace;
  int *starts = (int *)((char *)workspace + (size_t)B * N * 16);
  int *hdrs = starts + (size_t)B * (kGridMaxG * kGridMaxG * kGridMaxG + 1);
  const int G = grid_cells_per_axis(N);
  hipLaunchKernelGGL(bq_grid_build_kernel, dim3((unsigned)B), dim3(1024), 0, s, N, G, radius, xyz, hdrs, starts, recs);
  const long long centres = (long long)B * m;
  const long long blocks = (centres + 3) / 4;
  if (blocks > 0x7fffffffLL) return PN2_EINVAL;
  const float r2 = radius * radius;   // fp32, EXT/src/ball_query_gpu.cu:22
  hipLaunchKernelGGL(bq_grid_query_kernel, dim3((unsigned)blocks), dim3(256), 0, s, N, m, r2, nsample, new_xyz, xyz, hdrs,
                     starts, recs, idx, centres);
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
