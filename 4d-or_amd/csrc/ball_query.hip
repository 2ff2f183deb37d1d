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
// SLAB cell list (round 3): a cell list that keeps the reference's index order WITHOUT sorting.
// The scan above is the right shape for "first nsample hits in ascending index" but tests every point of the prefix it
// walks (~0.5 hits per 64 tests at the headline shape); the cell list below tests only the 27 neighbouring cells but
// loses the index order (rank sort, and a crowded ball overflows its collection cap).  Here the cloud is cut into SLABS of
// 2048 CONSECUTIVE indices and every slab gets its own cell list:
//   build : one workgroup per (slab, cloud) — cell of a point = floor(coordinate / h) modulo 16 per axis (a HASH grid: no
//           bounding box pass; cells that alias only add candidates that fail the distance test), LDS histogram + scan +
//           scatter of (x, y, z, index in slab) records, one packed (start | count << 16) word per cell;
//   query : one wave per centre walks the slabs in order.  Lanes = 27 cells x 2 slots test the cell's records and set
//           bit `index in slab` of a 2048-bit mask in LDS — 64 lanes x 32 bits, so lane i afterwards holds indices
//           [32 i, 32 i + 32) of the slab: a popcount + a wave prefix sum give every set bit its output slot, in ascending
//           index order, for free.  The walk stops as soon as the row holds nsample hits (crowded balls: after a few
//           slabs), the row is padded with the first hit.
// Exactness.  Same distance expression and strict '<' as the scan.  A hit implies |p - c| <= r (1 + 2^-22) per axis in real
// arithmetic; the window [c - rw, c + rw], rw = r 1.0001^2, is mapped to cells in DOUBLE precision with h = rw 1.0001, so it
// spans at most three cells per axis and contains the cell of every hit as long as |coordinate / h| < 2^32 (error of the
// double arithmetic 2^-19 cells against a margin of 1e-4).  Points beyond that (or non-finite) mark their slab, centres
// beyond that mark themselves: such a (centre, slab) pair tests ALL records of the slab through the same mask — slow,
// never wrong.  The mask makes duplicates harmless, the order inside a cell irrelevant.
// Round 4: the slab is a template parameter — 2048 W consecutive indices, W in {1, 4}.  A crowded ball finds its nsample hits
// in ONE slab when the slab holds about nsample / r^3 indices (unit-ball clouds): the dependent chain table word ->
// records -> LDS mask -> output is then walked once or twice per centre instead of four times (headline shape: 16 hits per
// 2048-index slab, 64 wanted), and the table shrinks from 13 MB to 3.7 MB per batch.  Lane i of the mask owns indices
// [32 W i, 32 W (i + 1)) of the slab, W words read back with one LDS instruction.  The row is assembled in LDS and
// written with coalesced stores; with FUSE the same wave then emits the GROUPED rows of its neighbourhood — what
// group_points_kernel (EXT/src/group_points_gpu.cu:8-28) x 2 + the subtraction / concatenation of QueryAndGroup.forward
// (OPS/pointnet2_utils.py:317-328) produce — so the index row never travels back through HBM between the two kernels.
// Workgroups are dealt to XCDs by cloud (hardware round-robin: workgroup p runs on XCD p mod 8), so a cloud's records and
// table are fetched into ONE L2 instead of all eight.
constexpr int kSlab = 2048;                       // indices per slab unit = 64 lanes x 32 mask bits
constexpr int kSlabCells = 16 * 16 * 16;          // hash grid per slab
constexpr int kSlabTable = kSlabCells + 16;       // words per (cloud, slab): cells | [4096] = "holds a wild point"
constexpr int kFuseMaxNs = 256;                   // index row staged in LDS (per wave) up to this many samples
constexpr int kFuseMaxRow = 16;                   // floats per grouped row the fused emission covers (3 + C)

// inclusive prefix sum over the 64 lanes on the DPP network (row shifts inside the rows of 16, then the row totals
// broadcast into the rows above): six VALU instructions, no LDS crossbar
__device__ __forceinline__ int slab_wave_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}

__device__ __forceinline__ bool slab_tame(double t) { return fabs(t) < 4294967296.0; }   // false for NaN / inf too
__device__ __forceinline__ int slab_cell1(double t) {                                    // floor(t) mod 16, t tame
  const double f = floor(t);
  return (int)(f - 16.0 * floor(f * 0.0625));
}

// one workgroup of 256 W threads per (slab, cloud): 8 points per thread
template <int W>
__global__ __launch_bounds__(256 * W) void bq_slab_build_kernel(int N, int nslab, double inv_h,
                                                               const float *__restrict__ xyz,
                                                               unsigned *__restrict__ table, float4 *__restrict__ recs) {
  constexpr int T = 256 * W, SL = kSlab * W, CPT = kSlabCells / T, NW = T / 64;
  __shared__ int hist[kSlabCells];
  __shared__ int wsum[NW];
  __shared__ int wild;
  const int tid = threadIdx.x, lane = pn2_lane(), wv = tid >> 6;
  const int sl = blockIdx.x, b = blockIdx.y;
  const int base = sl * SL;
  const int len = N - base < SL ? N - base : SL;
  const float *P = xyz + ((size_t)b * N + base) * 3;
  for (int i = tid; i < kSlabCells; i += T) hist[i] = 0;
  if (tid == 0) wild = 0;
  __syncthreads();
  float px[8], py[8], pz[8];
  int cell[8], rank[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int k = tid + u * T;
    cell[u] = -1;
    if (k < len) {
      px[u] = P[(size_t)k * 3 + 0]; py[u] = P[(size_t)k * 3 + 1]; pz[u] = P[(size_t)k * 3 + 2];
      const double tx = (double)px[u] * inv_h, ty = (double)py[u] * inv_h, tz = (double)pz[u] * inv_h;
      if (slab_tame(tx) && slab_tame(ty) && slab_tame(tz)) {
        cell[u] = (slab_cell1(tz) * 16 + slab_cell1(ty)) * 16 + slab_cell1(tx);
      } else {
        cell[u] = 0;
        wild = 1;
      }
      rank[u] = atomicAdd(&hist[cell[u]], 1);
    }
  }
  __syncthreads();
  // exclusive scan of the 4096 counts: thread t owns cells [CPT t, CPT t + CPT)
  int c[CPT], sum = 0;
#pragma unroll
  for (int i = 0; i < CPT; ++i) { c[i] = hist[tid * CPT + i]; sum += c[i]; }
  int inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(inc, d);
    if (lane >= d) inc += v;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  int off = inc - sum;
  for (int w = 0; w < wv; ++w) off += wsum[w];
  unsigned *Tb = table + ((size_t)b * nslab + sl) * kSlabTable;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    hist[tid * CPT + i] = off;
    Tb[tid * CPT + i] = (unsigned)off | ((unsigned)c[i] << 16);    // start < 8192, count <= 8192: 16 bits each
    off += c[i];
  }
  if (tid < 16) Tb[kSlabCells + tid] = (unsigned)wild;
  __syncthreads();
  float4 *R = recs + (size_t)b * N + base;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (cell[u] >= 0) R[hist[cell[u]] + rank[u]] = make_float4(px[u], py[u], pz[u], __int_as_float(tid + u * T));
  }
}

struct Row3 { float a, b, c; };   // 12 bytes, 4-byte aligned: global_load_dwordx3

struct BqFuse {              // grouped-row emission (FUSE): out[g][s][0:Cx] = (xyz[idx] - centre) (/ radius), [Cx:Cx+C] = feats[idx]
  const float *xyz;          // (B, N, 3)
  const float *feats;        // (B, N, C) point-major or null (C = 0)
  float *rows;               // (B, m, ns, Cx + C)
  int C, Cx, normalize;
  float radius;
};

// LDS traffic of a wave is ordered (one LDS queue per CU, in order per wave): lanes may read what other lanes of the wave
// wrote once the counter has drained.  The arrays are indexed as __shared__ objects (ds_* instructions; a pointer into
// them that loses its address space becomes flat_* with a full wait each) and the compiler is told not to move LDS
// accesses across the points where lanes exchange data.
__device__ __forceinline__ void slab_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }

// FUSE: 0 = indices only; 6 = grouped rows of exactly 3 + 3 floats (xyz + colours: the first level of the backbone and of
// the object encoder) with every row offset a compile-time constant; 1 = grouped rows of any covered width
template <int W, int FUSE>
__global__ __launch_bounds__(256, 8) void bq_slab_query_kernel(int N, int m, int nslab, float r2, int ns, double inv_h,
                                                           double rw, const float *__restrict__ new_xyz,
                                                           const unsigned *__restrict__ table,
                                                           const float4 *__restrict__ recs, int *__restrict__ idx,
                                                           int centres, int blocks, BqFuse fz) {
  constexpr int SL = kSlab * W;
  __shared__ unsigned s_mask[4][64 * W];
  __shared__ int s_idx[FUSE ? 4 : 1][FUSE ? kFuseMaxNs : 1];
  __shared__ float s_rows[FUSE ? 4 : 1][FUSE == 6 ? 64 * 6 : (FUSE ? 64 * kFuseMaxRow : 1)];
  const int lane = pn2_lane();
  const int wv = threadIdx.x >> 6;
  // XCD-aware order: physical workgroup p runs on XCD p mod 8; XCD x takes the x-th eighth of the (cloud-major) centres
  const int per = (blocks + 7) >> 3;
  const int lb = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (lb >= blocks) return;
  const int g = lb * 4 + wv;
  if (g >= centres) return;                                     // wave-uniform; no block barriers below
  const int b = (int)((unsigned)g / (unsigned)m);
  const float qx = new_xyz[(size_t)g * 3 + 0], qy = new_xyz[(size_t)g * 3 + 1], qz = new_xyz[(size_t)g * 3 + 2];
  int *row = idx + (size_t)g * ns;
#pragma unroll
  for (int k = 0; k < W; ++k) s_mask[wv][lane * W + k] = 0u;

  // lane -> (cell of the 3 x 3 x 3 window, slot); the window starts at the cell of c - rw on every axis
  const double tx = ((double)qx - rw) * inv_h, ty = ((double)qy - rw) * inv_h, tz = ((double)qz - rw) * inv_h;
  const bool tame = slab_tame(tx) && slab_tame(ty) && slab_tame(tz);
  const int kc = lane >> 1, slot = lane & 1;
  int mycell = -1;
  if (tame && kc < 27) {
    const int ox = kc % 3, oy = (kc / 3) % 3, oz = kc / 9;
    mycell = (((slab_cell1(tz) + oz) & 15) * 16 + ((slab_cell1(ty) + oy) & 15)) * 16 + ((slab_cell1(tx) + ox) & 15);
  }
  const unsigned *T = table + (size_t)b * nslab * kSlabTable;
  const float4 *R = recs + (size_t)b * N;
  int cnt = 0, first = 0;
  unsigned wnext = mycell >= 0 ? T[mycell] : 0u;
  for (int sl = 0; sl < nslab && cnt < ns; ++sl, T += kSlabTable, R += SL) {
    const unsigned w = wnext;
    if (sl + 1 < nslab && mycell >= 0) wnext = T[kSlabTable + mycell];
    const bool all = !tame || T[kSlabCells] != 0u;              // wave-uniform
    if (!all) {
      const int beg = (int)(w & 0xffffu), len = (int)(w >> 16);
      // four records in flight per lane (a cell of an 8192-index slab holds ~16 at the headline density): one pointer, four
      // loads at constant offsets — a load past the cell's last record reads a neighbouring record (the table follows the
      // records in the workspace: never out of the allocation) and is discarded by the count test
      const float4 *P = R + beg + slot;
      for (int j = slot; j < len; j += 8, P += 8) {
        const float4 p0 = P[0], p1 = P[2], p2 = P[4], p3 = P[6];
        const int left = len - j;                                 // records of this lane's slot sequence still inside: u < left / 2
        if (pn2_sq3(qx - p0.x, qy - p0.y, qz - p0.z) < r2) {
          const int li = __float_as_int(p0.w);
          atomicOr(&s_mask[wv][li >> 5], 1u << (li & 31));
        }
        if (left > 2 && pn2_sq3(qx - p1.x, qy - p1.y, qz - p1.z) < r2) {
          const int li = __float_as_int(p1.w);
          atomicOr(&s_mask[wv][li >> 5], 1u << (li & 31));
        }
        if (left > 4 && pn2_sq3(qx - p2.x, qy - p2.y, qz - p2.z) < r2) {
          const int li = __float_as_int(p2.w);
          atomicOr(&s_mask[wv][li >> 5], 1u << (li & 31));
        }
        if (left > 6 && pn2_sq3(qx - p3.x, qy - p3.y, qz - p3.z) < r2) {
          const int li = __float_as_int(p3.w);
          atomicOr(&s_mask[wv][li >> 5], 1u << (li & 31));
        }
      }
    } else {
      const int len = N - sl * SL < SL ? N - sl * SL : SL;
      for (int j = lane; j < len; j += 64) {
        const float4 p = R[j];
        if (pn2_sq3(qx - p.x, qy - p.y, qz - p.z) < r2) {
          const int li = __float_as_int(p.w);
          atomicOr(&s_mask[wv][li >> 5], 1u << (li & 31));
        }
      }
    }
    slab_lds_sync();
    unsigned word[W];
    bool any = false;
#pragma unroll
    for (int k = 0; k < W; ++k) { word[k] = s_mask[wv][lane * W + k]; any |= word[k] != 0u; }
    const u64 some = __ballot(any);
    if (some == 0ull) continue;
    int pc = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) { s_mask[wv][lane * W + k] = 0u; pc += __popc(word[k]); }
    const int inc = slab_wave_scan(pc);
    const int at = sl * SL + lane * 32 * W;
    if (cnt == 0) {
      int mine = 0;
#pragma unroll
      for (int k = W - 1; k >= 0; --k)
        if (word[k] != 0u) mine = at + k * 32 + __builtin_ctz(word[k]);
      first = __builtin_amdgcn_readlane(mine, __builtin_ctzll(some));
    }
    int o = cnt + inc - pc;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      unsigned wk = word[k];
      while (wk != 0u && o < ns) {
        const int hit = at + k * 32 + __builtin_ctz(wk);
        if (FUSE) s_idx[wv][o] = hit; else row[o] = hit;
        ++o;
        wk &= wk - 1u;
      }
    }
    cnt += __builtin_amdgcn_readlane(inc, 63);
    slab_lds_sync();                                            // the mask words are clear before the next slab's hits land
  }
  // pad with the first hit (zero row for an empty ball): EXT/src/ball_query_gpu.cu:34-38
  if (!FUSE) {
    for (int s = (cnt < ns ? cnt : ns) + lane; s < ns; s += 64) row[s] = first;
    return;
  }
  for (int s = (cnt < ns ? cnt : ns) + lane; s < ns; s += 64) s_idx[wv][s] = first;
  slab_lds_sync();
  // the index row (coalesced) and the grouped rows: 64 slots at a time, each lane gathers its slot's point, the row
  // block goes through LDS so that the stores are contiguous 256-byte runs (a neighbourhood's rows are one contiguous block)
  const int Cx = FUSE == 6 ? 3 : fz.Cx, C = FUSE == 6 ? 3 : fz.C, RW = Cx + C;
  const float *X = fz.xyz + (size_t)b * N * 3;
  const float *F = fz.feats ? fz.feats + (size_t)b * N * C : nullptr;
  float *out = fz.rows + (size_t)g * ns * RW;
#pragma unroll 1
  for (int s0 = 0; s0 < ns; s0 += 64) {
    const int s = s0 + lane;
    const int nrow = ns - s0 < 64 ? ns - s0 : 64;
    if (s < ns) {
      const int i = s_idx[wv][s];
      row[s] = i;
      // one 12-byte load per row where the row is three floats (coordinates, colours): a scattered access costs the
      // CU's address unit a pass per LANE, whatever its width
      if (Cx) {
        const Row3 p = *reinterpret_cast<const Row3 *>(X + (size_t)i * 3);
        float rx = p.a - qx, ry = p.b - qy, rz = p.c - qz;
        if (fz.normalize) { rx = __fdiv_rn(rx, fz.radius); ry = __fdiv_rn(ry, fz.radius); rz = __fdiv_rn(rz, fz.radius); }
        s_rows[wv][lane * RW + 0] = rx; s_rows[wv][lane * RW + 1] = ry; s_rows[wv][lane * RW + 2] = rz;
      }
      if (C == 3) {
        const Row3 f = *reinterpret_cast<const Row3 *>(F + (size_t)i * 3);
        s_rows[wv][lane * RW + Cx + 0] = f.a; s_rows[wv][lane * RW + Cx + 1] = f.b; s_rows[wv][lane * RW + Cx + 2] = f.c;
      } else if ((C & 3) == 0) {
#pragma unroll 1
        for (int c = 0; c < C; c += 4) {
          const float4 f = *reinterpret_cast<const float4 *>(F + (size_t)i * C + c);
          s_rows[wv][lane * RW + Cx + c + 0] = f.x; s_rows[wv][lane * RW + Cx + c + 1] = f.y;
          s_rows[wv][lane * RW + Cx + c + 2] = f.z; s_rows[wv][lane * RW + Cx + c + 3] = f.w;
        }
      } else {
#pragma unroll 1
        for (int c = 0; c < C; ++c) s_rows[wv][lane * RW + Cx + c] = F[(size_t)i * C + c];
      }
    }
    slab_lds_sync();
    const int total = nrow * RW;
    if (FUSE == 6 && nrow == 64) {
      float *o = out + (size_t)s0 * 6 + lane;
#pragma unroll
      for (int k = 0; k < 6; ++k) o[k * 64] = s_rows[wv][k * 64 + lane];
    } else {
#pragma unroll 2
      for (int e = lane; e < total; e += 64) out[(size_t)s0 * RW + e] = s_rows[wv][e];
    }
    slab_lds_sync();
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

// Which algorithm.  The host cannot see the coordinates, so it estimates the hits per ball for a cloud that fills the
// unit ball (the 4D-OR clouds are normalised that way, zero_mean of data_preparation_utils.py:12-18): E = N r^3.
//   PN2_BQ_SCAN  : index-order scan with early exit — no workspace; small clouds, very crowded balls (short walks);
//   PN2_BQ_CELLS : one cell list per cloud + rank sort — sparse balls, nsample <= 256 (round 2; by name only);
//   PN2_BQ_SLABS : one cell list per 2048-index slab + bit-mask order — everything whose scan would walk >= 3072 points.
// Results are identical whichever runs.  Measurements: profiles/HISTORY.md 4c.
namespace {
bool bq_radius_ok(float radius) { return radius > 0.f && radius < 3.0e38f; }

size_t bq_slab_bytes(int B, int N) {
  const size_t nslab = ((size_t)N + kSlab - 1) / kSlab;
  return (size_t)B * N * 16 + (size_t)B * nslab * kSlabTable * 4 + 256;
}
size_t bq_cells_bytes(int B, int N, int nsample) {
  if (B <= 0 || N <= 0 || nsample <= 0 || nsample > kGridCap) return 0;
  const size_t per_cloud = (size_t)kGridHdr * 4 + (size_t)(kGridMaxG * kGridMaxG * kGridMaxG + 1) * 4 + (size_t)N * 16;
  return (size_t)B * per_cloud + 256;
}

// The scan walks L = N min(1, nsample / E) points per centre (E = N r^3: estimated hits per ball), the slab walk visits
// ceil(L / 2048) slabs at a roughly constant price each: measured crossover L ~ 3000 (tools/bq_bench.py, profiles/HISTORY.md 4c).
// The per-cloud cell list is never the automatic choice any more (the slabs match or beat it on every measured shape but
// one); it stays available by name.
int bq_auto(int B, int N, int m, float radius, int nsample) {
  if (B <= 0 || B > 65535 || m <= 0 || N < 2048 || nsample <= 0 || !bq_radius_ok(radius)) return PN2_BQ_SCAN;
  const double E = (double)N * radius * radius * radius;
  const double L = E > nsample ? (double)N * nsample / E : (double)N;
  return L >= 3072.0 ? PN2_BQ_SLABS : PN2_BQ_SCAN;
}

// Slab width: a ball finds nsample hits in about nsample / (2048 r^3) slabs of 2048 indices (unit-ball clouds); from 1.5
// on, slabs of 8192 indices walk the dependent table -> records -> mask chain less often (and a sparse ball in a cloud of up
// to 8192 points sees the whole cloud as ONE cell list).  Results never depend on it.
int bq_slab_w(int N, float radius, int nsample) {
  const char *force = getenv("PN2_BQ_SLAB_W");                 // test / measurement hook: 1 or 4 (results never depend on it)
  if (force && (force[0] == '1' || force[0] == '4') && force[1] == 0) return force[0] - '0';
  if (N <= kSlab) return 1;
  const double per = (double)nsample / (2048.0 * (double)radius * radius * radius);
  return per > 1.5 ? 4 : 1;
}

int bq_run_slabs(int B, int N, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx,
                 void *workspace, hipStream_t s, const BqFuse *fuse = nullptr, int force_w = 0) {
  const int W = force_w ? force_w : bq_slab_w(N, radius, nsample);
  const int SL = kSlab * W;
  const int nslab = (N + SL - 1) / SL;
  const long long centres = (long long)B * m;
  const long long blocks = (centres + 3) / 4;
  const long long grid = ((blocks + 7) / 8) * 8;                  // XCD-aware order: eight equal shares
  if (centres > 0x7fffffffLL - 64 || B > 65535) return PN2_EINVAL;     // (the kernel indexes centres with 32 bits)
  // hit bound rq = r 1.0001, window rw = rq 1.0001, cell edge h = rw 1.0001 (see the kernel header)
  const double rw = (double)radius * 1.0001 * 1.0001, inv_h = 1.0 / (rw * 1.0001);
  float4 *recs = (float4 *)workspace;
  unsigned *table = (unsigned *)((char *)workspace + (size_t)B * N * 16);
  const float r2 = radius * radius;   // fp32, EXT/src/ball_query_gpu.cu:22
  BqFuse fz = fuse ? *fuse : BqFuse{nullptr, nullptr, nullptr, 0, 0, 0, 1.f};
#define PN2_BQ_LAUNCH(WW, FF)                                                                                          \
  hipLaunchKernelGGL((bq_slab_query_kernel<WW, FF>), dim3((unsigned)grid), dim3(256), 0, s, N, m, nslab, r2, nsample,  \
                     inv_h, rw, new_xyz, table, recs, idx, (int)centres, (int)blocks, fz)
  const int ff = !fuse ? 0 : (fz.Cx == 3 && fz.C == 3 ? 6 : 1);
  if (W == 4) {
    hipLaunchKernelGGL(bq_slab_build_kernel<4>, dim3((unsigned)nslab, (unsigned)B), dim3(1024), 0, s, N, nslab, inv_h, xyz,
                       table, recs);
    if (ff == 0) PN2_BQ_LAUNCH(4, 0); else if (ff == 6) PN2_BQ_LAUNCH(4, 6); else PN2_BQ_LAUNCH(4, 1);
  } else {
    hipLaunchKernelGGL(bq_slab_build_kernel<1>, dim3((unsigned)nslab, (unsigned)B), dim3(256), 0, s, N, nslab, inv_h, xyz,
                       table, recs);
    if (ff == 0) PN2_BQ_LAUNCH(1, 0); else if (ff == 6) PN2_BQ_LAUNCH(1, 6); else PN2_BQ_LAUNCH(1, 1);
  }
#undef PN2_BQ_LAUNCH
  return pn2_check_launch();
}

int bq_run_cells(int B, int N, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx,
                 void *workspace, hipStream_t s) {
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
}  // namespace

// Workspace of a given algorithm for a shape; 0 = the algorithm does not cover the shape (or needs none: the scan).
extern "C" size_t pn2_ball_query_algo_bytes(int algo, int B, int N, int m, float radius, int nsample) {
  if (B <= 0 || N <= 0 || m <= 0 || nsample <= 0 || !bq_radius_ok(radius)) return 0;
  if (algo == PN2_BQ_CELLS) return bq_cells_bytes(B, N, nsample);
  if (algo == PN2_BQ_SLABS) return B <= 65535 ? bq_slab_bytes(B, N) : 0;
  return 0;
}

// The algorithm the library would pick for the shape, and its workspace.
extern "C" int pn2_ball_query_auto(int B, int N, int m, float radius, int nsample) {
  return bq_auto(B, N, m, radius, nsample);
}
extern "C" size_t pn2_ball_query_workspace_bytes(int B, int N, int m, float radius, int nsample) {
  return pn2_ball_query_algo_bytes(bq_auto(B, N, m, radius, nsample), B, N, m, radius, nsample);
}

// Raw requirement of the per-cloud cell list (kept from round 2: a caller that passes exactly this much workspace to
// pn2_ball_query_ws gets the cell list whatever the density estimate says).
extern "C" size_t pn2_ball_query_grid_bytes(int B, int N, int nsample) { return bq_cells_bytes(B, N, nsample); }

// Explicit algorithm.  A shape the algorithm does not cover, a missing / short / misaligned workspace, or a radius the
// cell edge cannot be sized from (non-positive, non-finite: r * r is still a valid threshold for the scan) -> the scan.
extern "C" int pn2_ball_query_algo(int algo, int B, int N, int m, float radius, int nsample, const float *new_xyz,
                                   const float *xyz, int *idx, void *workspace, size_t workspace_bytes, void *stream) {
  if (B < 0 || N < 0 || m < 0 || nsample < 0) return PN2_EINVAL;
  if (algo != PN2_BQ_SCAN && algo != PN2_BQ_CELLS && algo != PN2_BQ_SLABS) return PN2_EINVAL;
  const size_t need = pn2_ball_query_algo_bytes(algo, B, N, m, radius, nsample);
  if (need == 0 || !workspace || workspace_bytes < need || ((uintptr_t)workspace & 15) != 0)
    return pn2_ball_query(B, N, m, radius, nsample, new_xyz, xyz, idx, stream);
  if (!new_xyz || !idx || !xyz) return PN2_ENULL;
  hipStream_t s = (hipStream_t)stream;
  if (algo == PN2_BQ_SLABS) return bq_run_slabs(B, N, m, radius, nsample, new_xyz, xyz, idx, workspace, s);
  return bq_run_cells(B, N, m, radius, nsample, new_xyz, xyz, idx, workspace, s);
}

// Automatic choice with the workspace the caller has: the shape's own algorithm when the workspace holds it, else the
// per-cloud cell list when it holds that (pn2_ball_query_grid_bytes), else the scan.
extern "C" int pn2_ball_query_ws(int B, int N, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                                 int *idx, void *workspace, size_t workspace_bytes, void *stream) {
  if (B < 0 || N < 0 || m < 0 || nsample < 0) return PN2_EINVAL;
  int algo = bq_auto(B, N, m, radius, nsample);
  size_t need = pn2_ball_query_algo_bytes(algo, B, N, m, radius, nsample);
  if (algo == PN2_BQ_SCAN || need == 0 || workspace_bytes < need) {
    algo = PN2_BQ_CELLS;
    need = pn2_ball_query_algo_bytes(algo, B, N, m, radius, nsample);
    if (need == 0 || workspace_bytes < need) algo = PN2_BQ_SCAN;
  }
  if (workspace && ((uintptr_t)workspace & 15) != 0) return PN2_EINVAL;
  return pn2_ball_query_algo(algo, B, N, m, radius, nsample, new_xyz, xyz, idx, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Ball query + grouping as ONE pass (round 4): query_ball_point_kernel (EXT/src/ball_query_gpu.cu:9-44), the two
// group_points_kernel gathers (EXT/src/group_points_gpu.cu:8-28) and the centre subtraction / concatenation of
// QueryAndGroup.forward (OPS/pointnet2_utils.py:317-328; GF3D's `/= radius`) — SURVEY.md 8d "fused ball_query+group".
// The wave that finds a centre's hits (slab cell lists, above) also emits the neighbourhood's grouped rows; idx is kept
// for the backward.  Covers nsample <= 256 and rows of at most 16 floats (first levels: xyz + colours); everything else
// stays pn2_ball_query_* + pn2_group_concat_rows.  Bit-identical to that pair.
extern "C" int pn2_ball_query_group_supported(int B, int N, int m, float radius, int nsample, int C, int use_xyz) {
  const int RW = (use_xyz ? 3 : 0) + C;
  return (B > 0 && B <= 65535 && N > 0 && m > 0 && nsample > 0 && nsample <= kFuseMaxNs && C >= 0 && RW > 0 &&
          RW <= kFuseMaxRow && bq_radius_ok(radius)) ? 1 : 0;
}
extern "C" size_t pn2_ball_query_group_workspace_bytes(int B, int N) { return B > 0 && N > 0 ? bq_slab_bytes(B, N) : 0; }

extern "C" int pn2_ball_query_group(int B, int N, int m, float radius, int nsample, int C, int use_xyz, int normalize,
                                    const float *new_xyz, const float *xyz, const float *feats, int *idx, float *rows,
                                    void *workspace, size_t workspace_bytes, int slab_w, void *stream) {
  if (B < 0 || N < 0 || m < 0 || nsample < 0 || C < 0) return PN2_EINVAL;
  if (B == 0 || m == 0 || nsample == 0) return PN2_OK;
  if (!pn2_ball_query_group_supported(B, N, m, radius, nsample, C, use_xyz)) return PN2_EINVAL;
  if (slab_w != 0 && slab_w != 1 && slab_w != 4) return PN2_EINVAL;
  if (!new_xyz || !xyz || !idx || !rows || (C > 0 && !feats)) return PN2_ENULL;
  if (!workspace || workspace_bytes < bq_slab_bytes(B, N) || ((uintptr_t)workspace & 15) != 0) return PN2_EINVAL;
  BqFuse fz{xyz, C > 0 ? feats : nullptr, rows, C, use_xyz ? 3 : 0, normalize ? 1 : 0, radius};
  return bq_run_slabs(B, N, m, radius, nsample, new_xyz, xyz, idx, workspace, (hipStream_t)stream, &fz, slab_w);
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
