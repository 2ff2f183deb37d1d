// fps.hip — furthest point sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel (EXT/src/sampling_gpu.cu:69-229).
// Same selection rule, different machine mapping:
//
//  * The reference streams xyz + temp from memory every round with one
//    512-thread block per cloud.  Here a cloud's points live in VGPRs for the
//    whole kernel (xyz + running min distance = 4 registers per point, up to
//    24 points per lane x 1024 lanes, or a cluster of workgroups per cloud), so a round is pure VALU + one wave64 DPP
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
#include <atomic>
#include "pn2_common.h"

#include <math.h>

#include <map>
#include <mutex>
#include <utility>

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

// Wave-wide max of a packed 64-bit key through two 32-bit DPP reductions (hi, then lo among the lanes that hold
// the maximal hi): ~14 VALU instructions where the 64-bit DPP form needs ~45.
__device__ __forceinline__ u64 fps_wave_max_key(u64 pk) {
  unsigned mhi, mlo;
  pn2_wave_argmax_u32x2((unsigned)(pk >> 32), (unsigned)pk, mhi, mlo);
  return ((u64)mhi << 32) | mlo;
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
// A round is: PPT distance updates per lane -> wave arg-max of the (hi, lo) key by two 32-bit DPP
// reductions -> the winning lane fetches its point from an LDS copy of the coordinates (3 ds_reads;
// a select chain over PPT registers when the copy would not fit) -> one 32-byte slot per wave in
// LDS -> ONE barrier -> every wave reduces the NW slots again.  Small clouds run with FEW waves
// (256 threads = one wave per SIMD up to 2048 points): the round is a dependent chain, and a second
// wave on the SIMD only doubles the issue time of the first (measured 0.89 -> see DESIGN.md us/round).
struct __attribute__((aligned(32))) FpsSlot2 {
  unsigned hi, lo;
  float x, y, z;
  float pad0, pad1, pad2;
};

template <int BS, int PPT>
__global__ __launch_bounds__(BS) void fps_resident_kernel(int N, int m, int L,
                                                         const float *__restrict__ xyz,
                                                         int *__restrict__ idxs,
                                                         const int *__restrict__ ordered_fail) {
  constexpr int NW = BS / 64;
  constexpr bool LDSXYZ = PPT * BS * 12 <= 96 * 1024;      // coordinates copy fits next to the slots
  __shared__ FpsSlot2 slots[2][16];
  __shared__ float lxyz[LDSXYZ ? 3 * PPT * BS : 1];

  const int b = blockIdx.x;
  const int t = threadIdx.x;
  // pn2_furthest_point_sampling_ordered: rounds 1 .. r0 - 1 of this cloud were verified to pick points 1 .. r0 - 1 (r0 >= m: the
  // whole cloud is a sampling order — its samples are 0 .. m - 1 and no round runs)
  const int r0 = ordered_fail ? (ordered_fail[b] < m ? ordered_fail[b] : m) : 1;
  if (r0 > 1) {
    for (int k = t; k < r0; k += BS) idxs[(size_t)b * m + k] = k;
    if (r0 >= m) return;
  }
  const int lane = pn2_lane();
  const int wave = t >> 6;
  const float *P = xyz + (size_t)b * N * 3;
  int *out = idxs + (size_t)b * m;

  // Slot order inside a lane = priority order on equal distances.  With BS a multiple of the
  // reference block size 2^L all of a lane's points share one reference tid and ascending k is the
  // reference order (strict '>' at :108-109).  A 256-thread workgroup under a 512-thread reference
  // holds TWO reference tids per lane (t and t+256, the latter with the larger bit-reversed rank):
  // the points of tid t (even strides) come first, then those of tid t+256 (odd strides).
  const bool two_tids = BS < (1 << L);
  constexpr int NE = (PPT + 1) / 2;
  auto stride_of = [&](int slot) { return two_tids ? (slot < NE ? 2 * slot : 2 * (slot - NE) + 1) : slot; };
  float px[PPT], py[PPT], pz[PPT], td[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = t + stride_of(i) * BS;
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
    if (LDSXYZ) {                                           // only ever re-read by this thread
      lxyz[i * BS + t] = x; lxyz[(PPT + i) * BS + t] = y; lxyz[(2 * PPT + i) * BS + t] = z;
    }
  }

  const float p0x = P[0], p0y = P[1], p0z = P[2];
  float ox = p0x, oy = p0y, oz = p0z;  // old = 0
  if (t == 0) out[0] = 0;
  if (r0 > 1) {
    // the state of round r0: running distances against the verified centres 0 .. r0 - 2 (no barrier: every point on its
    // own), the centre of round r0 is sample r0 - 1
    for (int c = 0; c + 1 < r0; ++c) {
      const float cx = P[(size_t)c * 3], cy = P[(size_t)c * 3 + 1], cz = P[(size_t)c * 3 + 2];      // (wave-uniform address)
#pragma unroll
      for (int i = 0; i < PPT; ++i) td[i] = fps_min(pn2_sq3(px[i] - cx, py[i] - cy, pz[i] - cz), td[i]);
    }
    ox = P[(size_t)(r0 - 1) * 3]; oy = P[(size_t)(r0 - 1) * 3 + 1]; oz = P[(size_t)(r0 - 1) * 3 + 2];
  }

  for (int j = r0; j < m; ++j) {
    float best = -1.f;
    int bi = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float d = pn2_sq3(px[i] - ox, py[i] - oy, pz[i] - oz);
      const float d2 = fps_min(d, td[i]);
      td[i] = d2;
      if (d2 > best) { best = d2; bi = i; }
    }
    // key = (bits(best)+1, ~rank(k)); hi == 0 encodes "no candidate" (best = -1, besti = 0 at :90-91)
    const bool has = best >= 0.f;
    const unsigned hi = has ? __float_as_uint(best) + 1u : 0u;
    const unsigned lo = has ? ~fps_rank((unsigned)(t + stride_of(bi) * BS), L) : 0u;
    unsigned whi, wlo;
    const u64 who = pn2_wave_argmax_u32x2(hi, lo, whi, wlo);
    float sx, sy, sz;
    if (LDSXYZ) {
      sx = lxyz[bi * BS + t]; sy = lxyz[(PPT + bi) * BS + t]; sz = lxyz[(2 * PPT + bi) * BS + t];
    } else {
      sx = px[0]; sy = py[0]; sz = pz[0];
#pragma unroll
      for (int i = 1; i < PPT; ++i) {
        if (bi == i) { sx = px[i]; sy = py[i]; sz = pz[i]; }
      }
    }
    if (whi == 0u) { sx = p0x; sy = p0y; sz = p0z; }
    FpsSlot2 *buf = slots[j & 1];
    if (lane == (__ffsll((long long)who) - 1)) {              // unique lane (lane 0 when whi == 0)
      buf[wave].hi = whi; buf[wave].lo = wlo;
      buf[wave].x = sx; buf[wave].y = sy; buf[wave].z = sz;
    }
    __syncthreads();
    const int s = lane & 15;
    const unsigned shi = s < NW ? buf[s].hi : 0u, slo = s < NW ? buf[s].lo : 0u;
    unsigned ghi, glo;
    const u64 gwho = pn2_wave_argmax_u32x2(shi, slo, ghi, glo);
    const int ws = (__ffsll((long long)gwho) - 1) & 15;      // lanes s, s+16, ... hold the same slot
    ox = buf[ws].x; oy = buf[ws].y; oz = buf[ws].z;
    if (t == 0) out[j] = ghi ? (int)fps_unrank(~glo, L) : 0;
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
    const u64 wmax = fps_wave_max_key(pk);
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


// ---- cooperative kernel: G workgroups per cloud, points still register-resident ----
// A 50k-point cloud does not fit one workgroup's registers, and a batch of 32
// clouds would leave 224 of the 256 CUs idle anyway.  A cluster of G workgroups
// (G*B <= 256 so that every workgroup is resident) shares one cloud: workgroup g,
// thread t owns points k = g*BS + t + i*G*BS (same reference tid for all of a
// thread's points, ascending k).  Per round each workgroup reduces its own slice
// exactly like the resident kernel, then wave 0 publishes (packed, x, y, z) as
// five 8-byte {round, value} granules written with ONE agent-scope store each and
// sweeps the cluster's granules until every tag equals the round (the data is the
// flag; placement-independent; MI355X_MICROARCH "handoff" recipe R2).  Slots are
// double-buffered by round parity and zeroed by a memset node before every
// launch; spins are bounded and raise `status` instead of hanging the GPU.
typedef __attribute__((address_space(1))) u64 gu64;

constexpr int kCoopMaxG = 32;
// every cluster workgroup must be resident at once: 512 threads, < 64 VGPRs, 2 KB LDS => at least
// 3 workgroups fit a CU; we allow 2 per CU (256 CUs)
constexpr int kCoopMaxWorkgroups = 512;
constexpr int kCoopFields = 5;                                      // hi, lo, x, y, z
// (fps_multi_kernel: [2 parities][6 fields][64 sub-blobs] granules per cloud — the larger of the two layouts)
constexpr size_t kCoopCloudBytes = 2ull * 6 * 64 * sizeof(u64);
static_assert(kCoopCloudBytes >= 2ull * kCoopFields * kCoopMaxG * sizeof(u64), "cluster hand-off slots");
constexpr unsigned kCoopSpinLimit = 1u << 22;

__device__ __forceinline__ void coop_store(u64 *p, u64 v) {
  __hip_atomic_store((gu64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 coop_load(const u64 *p) {
  return __hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct __attribute__((aligned(16))) FpsFinal {
  float x, y, z;
  int k;
  int abort;
  int pad[3];
};

// NC clouds per cluster: the two synchronisation costs of a round — the intra-workgroup
// barrier/LDS exchange (~0.5 us) and the inter-workgroup granule hand-off (~1.3-2 us) — do not
// depend on how many candidates they carry, so a cluster of G workgroups serves NC clouds at
// once: every workgroup owns a 1/G slice of EACH of its NC clouds (NC*PPT point slots per
// lane), scans them back to back, and ONE barrier + ONE hand-off per round moves all NC
// candidates.  Wave c (< NC) sweeps the granules of cloud c, so the sweeps run in parallel.
// Cluster q = blockIdx.x % (B/NC) serves clouds q*NC .. q*NC+NC-1; workgroup g = blockIdx.x / (B/NC).
// TAIL (clouds beyond the register capacity G * BS * PPT of a cluster, e.g. 64 x 200k points on 256 CUs): the first
// PPT slots of a thread stay in registers, its further points k0 + i * G * BS (i >= PPT — same reference tid, ascending k,
// so the tie order is unchanged) are streamed from memory each round with their running distances in `tail_td`
// (B x N floats).  The streamed part is what the round costs (20 bytes per point and round); holding 40 % of a 200k
// cloud in registers and using all 256 CUs instead of 64 is 2.4x faster than the one-workgroup streaming kernel.
// ---- spatial bucketing for the cluster kernel (round 3) --------------------------------------------------------
// A round of the cluster kernel costs ~0.9 us of synchronisation plus ~0.1 us per point slot of every lane (measured:
// 32 x 50k on two 1024-thread workgroups per cloud, 26 slots: 3.6 us; the same cluster on 4096 points: 0.87 us) — and
// almost all of the distance updates are no-ops: after a few hundred samples a new sample only lowers the running distance
// of points in its neighbourhood.  The reference's result does not depend on WHICH thread holds a point (the arg-max key
// carries the reference's rank of the point's index), so the points are first permuted into spatially compact groups:
// one workgroup per cloud bins the cloud into 16^3 cells of its bounding box, in Morton order of the cells (histogram,
// scan, scatter of (x, y, z, index) records; the order inside a cell is whatever the atomics give — it does not matter).
// A wave of the cluster kernel then owns 64 * PPT CONSECUTIVE records, i.e. a compact blob with a small bounding box, and
// skips the whole update of a round when the new sample is farther from that box than the wave's largest running
// distance: no running distance can change (see the kernel), the wave re-publishes its cached candidate.
constexpr int kBucketCells = 4096;

__device__ __forceinline__ int fps_cell(float x, float y, float z, const float *bb) {
  // bb: min[3], 16 / extent[3] (0 for a flat axis); NaN / out-of-range coordinates land in cell 0 / 15 of the axis
  const float tx = (x - bb[0]) * bb[3], ty = (y - bb[1]) * bb[4], tz = (z - bb[2]) * bb[5];
  const int qx = tx >= 0.f ? (tx < 15.f ? (int)tx : 15) : 0;
  const int qy = ty >= 0.f ? (ty < 15.f ? (int)ty : 15) : 0;
  const int qz = tz >= 0.f ? (tz < 15.f ? (int)tz : 15) : 0;
  int m = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b)
    m |= (((qx >> b) & 1) << (3 * b)) | (((qy >> b) & 1) << (3 * b + 1)) | (((qz >> b) & 1) << (3 * b + 2));
  return m;
}

__global__ __launch_bounds__(1024) void fps_bucket_kernel(int N, int Nstride, const float *__restrict__ xyz,
                                                          float4 *__restrict__ rec) {
  __shared__ float red[6][16];
  __shared__ float bb[6];
  __shared__ int hist[kBucketCells];
  __shared__ int wsum[16];
  const int t = threadIdx.x, lane = pn2_lane(), wave = t >> 6;
  const float *P = xyz + (size_t)blockIdx.x * N * 3;
  float4 *R = rec + (size_t)blockIdx.x * Nstride;
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int k = t; k < N; k += 1024) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = P[(size_t)k * 3 + c];
      if (v >= -3.0e38f && v <= 3.0e38f) { mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); }   // finite values only
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    for (int o = 32; o > 0; o >>= 1) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], o)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o)); }
    if (lane == 0) { red[c][wave] = mn[c]; red[3 + c][wave] = mx[c]; }
  }
  for (int i = t; i < kBucketCells; i += 1024) hist[i] = 0;
  __syncthreads();
  if (t < 3) {
    float a = red[t][0], b = red[3 + t][0];
    for (int w = 1; w < 16; ++w) { a = fminf(a, red[t][w]); b = fmaxf(b, red[3 + t][w]); }
    bb[t] = a;
    bb[3 + t] = b > a ? 16.f / (b - a) : 0.f;
  }
  __syncthreads();
  for (int k = t; k < N; k += 1024)
    atomicAdd(&hist[fps_cell(P[(size_t)k * 3], P[(size_t)k * 3 + 1], P[(size_t)k * 3 + 2], bb)], 1);
  __syncthreads();
  int c4[4], sum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { c4[i] = hist[4 * t + i]; sum += c4[i]; }
  int inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(inc, d);
    if (lane >= d) inc += v;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int off = inc - sum;
  for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
  for (int i = 0; i < 4; ++i) { hist[4 * t + i] = off; off += c4[i]; }
  __syncthreads();
  for (int k = t; k < N; k += 1024) {
    const float x = P[(size_t)k * 3], y = P[(size_t)k * 3 + 1], z = P[(size_t)k * 3 + 2];
    const int pos = atomicAdd(&hist[fps_cell(x, y, z, bb)], 1);
    R[pos] = make_float4(x, y, z, __int_as_float(k));
  }
}

// ---- bucketed kernel: ONE workgroup per cloud, running distances in L2, only the touched buckets updated ----------
// With the cloud binned (fps_bucket_kernel) into buckets of 64 S consecutive records, a round only has to update the
// buckets the new sample can reach: a bucket whose bounding box is farther from the sample than its largest running
// distance cannot change (simulated on a 50k-point ball, 2048 samples: 26 of 782 buckets per round on average, 18 in the
// second half).  So the points do not have to sit in registers, and a cloud does not have to be spread over several
// workgroups that then pay ~1.9 us per round for their hand-off: the records (x, y, z, index) and the running distances
// stay in memory (800 KB + 200 KB for 50k points: L2-resident), ONE workgroup of 16 waves owns the cloud, thread b keeps
// bucket b's bounding box in registers and its cached candidate (arg-max key, coordinates, largest distance) in LDS.
//   A  every thread tests its bucket against the new sample; the reachable ones are appended to a list (LDS)
//   B  wave w updates the listed buckets w, w + 16, ...: 64 records per slot, min with the stored distance, wave
//      arg-max of the reference's key (distance bits, rank of the point's index) -> the bucket's cache
//   C  block arg-max over the bucket caches = the reference's winner (the key is the reference's order)
// Three barriers per round, no spinning, any number of clouds (no residency requirement).  Exactness as in the cluster
// kernel: identical fp32 distances, skipped updates are provably no-ops (lb (1 - 2e-6) <= every computed distance of the
// bucket), ties through the rank.
constexpr int kBucketMaxBuckets = 1024;

template <int S>
__global__ __launch_bounds__(1024) void fps_bucketed_kernel(int N, int Nstride, int m, int L,
                                                           const float *__restrict__ xyz,
                                                           const float4 *__restrict__ rec, float *__restrict__ tdist,
                                                           int *__restrict__ idxs) {
  constexpr int NW = 16, PB = 64 * S;
  __shared__ u64 s_key[kBucketMaxBuckets];
  __shared__ float s_cx[kBucketMaxBuckets], s_cy[kBucketMaxBuckets], s_cz[kBucketMaxBuckets], s_maxd[kBucketMaxBuckets];
  __shared__ int s_list[kBucketMaxBuckets];
  __shared__ float s_bb[6][kBucketMaxBuckets];
  __shared__ int s_nact;
  __shared__ FpsSlot s_slots[2][16];

  const int t = threadIdx.x, lane = pn2_lane();
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int q = blockIdx.x;
  const float *P = xyz + (size_t)q * N * 3;
  const float4 *R = rec + (size_t)q * Nstride;
  float *T = tdist + (size_t)q * Nstride;
  int *out = idxs + (size_t)q * m;
  const int nb = (N + PB - 1) / PB;

  // ---- prologue: running distances, bounding boxes and (empty) caches of the buckets; wave w: buckets w, w + 16, ...
  for (int b = wave; b < nb; b += NW) {
    float x0 = 3.0e38f, y0 = 3.0e38f, z0 = 3.0e38f, x1 = -3.0e38f, y1 = -3.0e38f, z1 = -3.0e38f;
    bool any = false;
#pragma unroll
    for (int sl = 0; sl < S; ++sl) {
      const int pos = b * PB + sl * 64 + lane;
      bool valid = false;
      if (pos < N) {
        const float4 r = R[pos];
        valid = !((double)pn2_sq3(r.x, r.y, r.z) <= 1e-3);      // EXT/src/sampling_gpu.cu:100-101
        if (valid) {
          x0 = fminf(x0, r.x); y0 = fminf(y0, r.y); z0 = fminf(z0, r.z);
          x1 = fmaxf(x1, r.x); y1 = fmaxf(y1, r.y); z1 = fmaxf(z1, r.z);
        }
      }
      if (pos < Nstride) T[pos] = valid ? 1e10f : -1.f;
      any |= valid;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      x0 = fminf(x0, __shfl_xor(x0, o)); y0 = fminf(y0, __shfl_xor(y0, o)); z0 = fminf(z0, __shfl_xor(z0, o));
      x1 = fmaxf(x1, __shfl_xor(x1, o)); y1 = fmaxf(y1, __shfl_xor(y1, o)); z1 = fmaxf(z1, __shfl_xor(z1, o));
    }
    const bool some = __ballot(any) != 0ull;
    if (lane == 0) {
      s_bb[0][b] = x0; s_bb[1][b] = y0; s_bb[2][b] = z0; s_bb[3][b] = x1; s_bb[4][b] = y1; s_bb[5][b] = z1;
      s_key[b] = 0ull;
      s_cx[b] = 0.f; s_cy[b] = 0.f; s_cz[b] = 0.f;
      s_maxd[b] = some ? 1e10f : -1.f;                          // (a bucket without candidates is never reachable)
    }
  }
  if (t == 0) { s_nact = 0; out[0] = 0; }                        // :87
  __syncthreads();
  const bool mine = t < nb;
  const float bx0 = mine ? s_bb[0][t] : 0.f, by0 = mine ? s_bb[1][t] : 0.f, bz0 = mine ? s_bb[2][t] : 0.f;
  const float bx1 = mine ? s_bb[3][t] : 0.f, by1 = mine ? s_bb[4][t] : 0.f, bz1 = mine ? s_bb[5][t] : 0.f;
  const float p0x = P[0], p0y = P[1], p0z = P[2];
  float ox = p0x, oy = p0y, oz = p0z;

  for (int j = 1; j < m; ++j) {
    // ---- A: reachable buckets
    {
      bool act = false;
      if (mine) {
        const float ex = fmaxf(fmaxf(bx0 - ox, ox - bx1), 0.f), ey = fmaxf(fmaxf(by0 - oy, oy - by1), 0.f),
                    ez = fmaxf(fmaxf(bz0 - oz, oz - bz1), 0.f);
        const float lb = (ex * ex + ey * ey + ez * ez) * 0.999998f;
        // (first round: every bucket with a participating point — a bucket of NaN points only has no box, yet the
        // reference samples them, EXT/src/sampling_gpu.cu:100-109)
        act = !(lb >= s_maxd[t]) || (j == 1 && s_maxd[t] >= 0.f);
      }
      const u64 am = __ballot(act);
      if (am) {                                                 // wave-uniform
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_nact, __popcll(am));
        base = __builtin_amdgcn_readfirstlane(base);
        if (act) s_list[base + pn2_prefix_popc(am)] = t;
      }
    }
    __syncthreads();
    // ---- B: update the listed buckets
    {
      const int nact = s_nact;
      auto update = [&](int b, const float4 (&r)[S], const float (&t0)[S]) {
        u64 best = 0ull;
        float cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll
        for (int sl = 0; sl < S; ++sl) {
          const int pos = b * PB + sl * 64 + lane;
          const float d = pn2_sq3(r[sl].x - ox, r[sl].y - oy, r[sl].z - oz);
          const float d2 = fps_min(d, t0[sl]);                  // (-1 stays -1: skipped points and padding)
          if (pos < Nstride) T[pos] = d2;
          const u64 pk = d2 >= 0.f ? fps_pack(d2, (unsigned)__float_as_int(r[sl].w), L) : 0ull;
          if (pk > best) { best = pk; cx = r[sl].x; cy = r[sl].y; cz = r[sl].z; }
        }
        const u64 wmax = fps_wave_max_key(best);
        const u64 who = __ballot(best == wmax);
        const int wl = __ffsll((long long)who) - 1;
        const float wx = pn2_readlane_f32(cx, wl), wy = pn2_readlane_f32(cy, wl), wz = pn2_readlane_f32(cz, wl);
        if (lane == 0) {
          s_key[b] = wmax;
          s_cx[b] = wx; s_cy[b] = wy; s_cz[b] = wz;
          s_maxd[b] = wmax ? __uint_as_float((unsigned)(wmax >> 32) - 1u) : -1.f;
        }
      };
      auto fetch = [&](int b, float4 (&r)[S], float (&t0)[S]) {
#pragma unroll
        for (int sl = 0; sl < S; ++sl) {
          const int pos = b * PB + sl * 64 + lane;
          const bool in = pos < Nstride;
          r[sl] = in ? R[pos] : make_float4(0.f, 0.f, 0.f, 0.f);
          t0[sl] = in ? T[pos] : -1.f;
        }
      };
      // up to four buckets in flight per wave: the first rounds list every bucket (49 per wave at 50k points), and a
      // bucket is one L2 round trip
      constexpr int FL = S <= 2 ? 4 : 2;
      for (int a = wave; a < nact; a += FL * NW) {
        int bq[FL];
        bool on[FL];
        float4 rq[FL][S];
        float uq[FL][S];
#pragma unroll
        for (int f = 0; f < FL; ++f) {
          on[f] = a + f * NW < nact;                            // wave-uniform
          bq[f] = s_list[on[f] ? a + f * NW : a];
          if (on[f]) fetch(bq[f], rq[f], uq[f]);
        }
#pragma unroll
        for (int f = 0; f < FL; ++f)
          if (on[f]) update(bq[f], rq[f], uq[f]);
      }
    }
    __syncthreads();
    // ---- C: block arg-max over the bucket caches
    {
      if (t == 0) s_nact = 0;
      const u64 key = mine ? s_key[t] : 0ull;
      const u64 wmax = fps_wave_max_key(key);
      const u64 who = __ballot(key == wmax && mine);
      const int wl = who ? __ffsll((long long)who) - 1 : 0;
      const int wb = wave * 64 + wl;                           // bucket of the wave's candidate
      const bool has = wmax != 0ull;
      const float sx = has ? s_cx[wb] : p0x, sy = has ? s_cy[wb] : p0y, sz = has ? s_cz[wb] : p0z;
      const u64 gmax = fps_block_exchange<NW>(s_slots[j & 1], wmax, lane == 0, sx, sy, sz, ox, oy, oz);
      if (t == 0) out[j] = gmax ? (int)fps_unrank(~(unsigned)gmax, L) : 0;
    }
  }
}

template <int BS, int PPT, int NC, bool TAIL = false, bool BUCK = false>
__global__ __launch_bounds__(BS) void fps_coop_kernel(int B, int N, int m, int L, int G,
                                                     const float *__restrict__ xyz,
                                                     int *__restrict__ idxs,
                                                     u64 *__restrict__ slots,
                                                     int *__restrict__ status,
                                                     float *__restrict__ tail_td = nullptr,
                                                     const float4 *__restrict__ rec = nullptr) {
  static_assert(!TAIL || NC == 1, "streamed tail: one cloud per cluster");
  static_assert(!BUCK || (NC == 1 && !TAIL), "bucketed points: one cloud per cluster, everything in registers");
  constexpr int NW = BS / 64;
  static_assert(NC <= NW, "one sweeping wave per cloud");
  __shared__ FpsSlot lds_slots[2][NC][16];
  __shared__ unsigned lds_vals[NC][kCoopFields][kCoopMaxG];
  __shared__ FpsFinal lds_fin[2][NC];
  // BUCK: reference index of every point slot of the workgroup (the arg-max key needs it for the tied candidates only)
  __shared__ int permk[BUCK ? BS * PPT : 1];

  const int nclusters = B / NC;
  const int q = blockIdx.x % nclusters;   // a cluster's workgroups share blockIdx % 8 (one XCD) when nclusters % 8 == 0
  const int g = blockIdx.x / nclusters;
  const int t = threadIdx.x;
  const int lane = pn2_lane();
  const int wave = t >> 6;

  float px[NC * PPT], py[NC * PPT], pz[NC * PPT], td[NC * PPT];
  const int k0 = g * BS + t;
  const int kstride = G * BS;
  float p0x[NC], p0y[NC], p0z[NC], ox[NC], oy[NC], oz[NC];
  // BUCK: bounding box of the wave's valid points, its cached candidate and its largest running distance
  float bbx0 = 3.0e38f, bby0 = 3.0e38f, bbz0 = 3.0e38f, bbx1 = -3.0e38f, bby1 = -3.0e38f, bbz1 = -3.0e38f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float *P = xyz + (size_t)(q * NC + c) * N * 3;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int k = k0 + i * kstride;
      float x = 0.f, y = 0.f, z = 0.f;
      bool valid = false;
      if constexpr (BUCK) {
        // wave (g, w) owns the records [b 64 PPT, +64 PPT) of the binned cloud, b = w G + g: neighbouring blobs — the ones
        // a sample activates together — sit in different workgroups and on different SIMDs
        const int pos = (wave * G + g) * (64 * PPT) + i * 64 + lane;
        int kk = 0;
        if (pos < N) {
          const float4 r = rec[(size_t)q * N + pos];
          x = r.x; y = r.y; z = r.z; kk = __float_as_int(r.w);
          const float mag = pn2_sq3(x, y, z);
          valid = !((double)mag <= 1e-3);
        }
        permk[wave * (64 * PPT) + i * 64 + lane] = kk;
        if (valid) {
          bbx0 = fminf(bbx0, x); bby0 = fminf(bby0, y); bbz0 = fminf(bbz0, z);
          bbx1 = fmaxf(bbx1, x); bby1 = fmaxf(bby1, y); bbz1 = fmaxf(bbz1, z);
        }
      } else if (k < N) {
        x = P[(size_t)k * 3 + 0];
        y = P[(size_t)k * 3 + 1];
        z = P[(size_t)k * 3 + 2];
        const float mag = pn2_sq3(x, y, z);
        valid = !((double)mag <= 1e-3);
      }
      px[c * PPT + i] = x; py[c * PPT + i] = y; pz[c * PPT + i] = z;
      td[c * PPT + i] = valid ? 1e10f : -1.f;
    }
    p0x[c] = P[0]; p0y[c] = P[1]; p0z[c] = P[2];
    ox[c] = p0x[c]; oy[c] = p0y[c]; oz[c] = p0z[c];
    if (g == 0 && t == 0) idxs[(size_t)(q * NC + c) * m] = 0;
  }

  u64 c_wmax = 0ull;
  float c_sx = 0.f, c_sy = 0.f, c_sz = 0.f, c_maxd = 1e10f;
  if constexpr (BUCK) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      bbx0 = fminf(bbx0, __shfl_xor(bbx0, o)); bby0 = fminf(bby0, __shfl_xor(bby0, o)); bbz0 = fminf(bbz0, __shfl_xor(bbz0, o));
      bbx1 = fmaxf(bbx1, __shfl_xor(bbx1, o)); bby1 = fmaxf(bby1, __shfl_xor(bby1, o)); bbz1 = fmaxf(bbz1, __shfl_xor(bbz1, o));
    }
    bbx0 = pn2_readlane_f32(bbx0, 0); bby0 = pn2_readlane_f32(bby0, 0); bbz0 = pn2_readlane_f32(bbz0, 0);
    bbx1 = pn2_readlane_f32(bbx1, 0); bby1 = pn2_readlane_f32(bby1, 0); bbz1 = pn2_readlane_f32(bbz1, 0);
    c_sx = p0x[0]; c_sy = p0y[0]; c_sz = p0z[0];
    __syncthreads();                                            // permk complete
  }

  for (int j = 1; j < m; ++j) {
    // ---- scan + wave arg-max of every cloud, winners into this round's LDS slots ----
    if constexpr (NC == 1) {
      // One cloud per cluster (the default): the scan only tracks the lane's largest running distance (v_max);
      // WHICH slot holds it, its rank and its coordinates are worked out once per wave on the scalar unit from
      // readlanes of the winning lane.  ~140 VALU instructions per round instead of ~230 (per-point compare +
      // two selects, 64-bit key reduction, 3*PPT-select coordinate chain).  The FPS co-runs with the MFMA kernels of
      // the training step and fp32 VALU time is exactly what it takes away from them.
      // BUCK: every point p of the wave lies in its bounding box, so its distance to the new sample is at least the
      // box's: computed in fp32 both carry a few ulp, lb (1 - 2e-6) <= d(p) for every p.  If that bound is not below the
      // wave's largest running distance, min(d, td) = td for all of them: nothing changes, the cached candidate stands.
      bool active = true;
      if constexpr (BUCK) {
        const float ex = fmaxf(fmaxf(bbx0 - ox[0], ox[0] - bbx1), 0.f), ey = fmaxf(fmaxf(bby0 - oy[0], oy[0] - bby1), 0.f),
                    ez = fmaxf(fmaxf(bbz0 - oz[0], oz[0] - bbz1), 0.f);
        const float lb = (ex * ex + ey * ey + ez * ez) * 0.999998f;
        // (first round: every blob scans — a blob of NaN points only has no box at all, yet the reference samples them)
        active = j == 1 || __builtin_amdgcn_readfirstlane((int)!(lb >= c_maxd)) != 0;
#ifdef FPS_STATS
        if (lane == 0) { atomicAdd(status + 2, 1); if (active) atomicAdd(status + 1, 1); }
#endif
      }
      u64 wmax = c_wmax;
      float sx = c_sx, sy = c_sy, sz = c_sz;
      if (active) {
      float best = -1.f;
      if constexpr (PPT % 2 == 0) {
        // two points per packed instruction (v_pk_add/mul/fma_f32): the same IEEE operations in the same order as
        // pn2_sq3 — fma(dz, dz, fma(dx, dx, dy * dy)) — on both halves
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 o2x = {ox[0], ox[0]}, o2y = {oy[0], oy[0]}, o2z = {oz[0], oz[0]};
#pragma unroll
        for (int i = 0; i < PPT; i += 2) {
          const f2 dx = f2{px[i], px[i + 1]} - o2x, dy = f2{py[i], py[i + 1]} - o2y, dz = f2{pz[i], pz[i + 1]} - o2z;
          const f2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
          const float a = fps_min(d.x, td[i]), b = fps_min(d.y, td[i + 1]);
          td[i] = a; td[i + 1] = b;
          best = fmaxf(best, fmaxf(a, b));
        }
      } else {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
          const float d = pn2_sq3(px[i] - ox[0], py[i] - oy[0], pz[i] - oz[0]);
          const float d2 = fps_min(d, td[i]);
          td[i] = d2;
          best = fmaxf(best, d2);
        }
      }
      const unsigned hi = best >= 0.f ? __float_as_uint(best) + 1u : 0u;
      const unsigned whi = pn2_wave_max_u32(hi);                  // wave-uniform
      // (threadIdx-derived values are divergent to the compiler even when they are not: without the readfirstlane
      // the whole resolution below is compiled with exec masks and v_readfirstlane waterfalls)
      const int wave_u = __builtin_amdgcn_readfirstlane(wave);
      wmax = 0ull;
      sx = p0x[0]; sy = p0y[0]; sz = p0z[0];
      if (whi != 0u) {
        const unsigned target = whi - 1u;                         // bits of the wave's largest distance
        u64 tied = __ballot(hi == whi);                           // usually exactly one lane
        unsigned best_lo = 0u;
        int wl = 0, wbi = 0;
        while (tied) {                                            // scalar loop over the tied lanes
          const int l = __ffsll((long long)tied) - 1;
          tied &= tied - 1;
          if constexpr (BUCK) {
            // the slots of a lane are in no particular index order: every slot that holds the maximum is a candidate
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
              if ((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(td[i]), l) == target) {   // wave-uniform
                const unsigned k = (unsigned)__builtin_amdgcn_readfirstlane(permk[wave_u * (64 * PPT) + i * 64 + l]);
                const unsigned lo = ~fps_rank(k, L);
                if (lo >= best_lo) { best_lo = lo; wl = l; wbi = i; }
              }
            }
          } else {
          int bi = 0;
#pragma unroll
          for (int i = PPT - 1; i >= 0; --i)                      // first (smallest k) slot holding the maximum
            bi = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(td[i]), l) == target ? i : bi;
          const unsigned k = (unsigned)(g * BS + wave_u * 64 + l + bi * kstride);
          const unsigned lo = ~fps_rank(k, L);
          if (lo >= best_lo) { best_lo = lo; wl = l; wbi = bi; }  // keys are unique: '>' or first
          }
        }
        wmax = ((u64)whi << 32) | best_lo;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
          if (wbi == i) {                                         // wave-uniform: a scalar branch
            sx = pn2_readlane_f32(px[i], wl); sy = pn2_readlane_f32(py[i], wl); sz = pn2_readlane_f32(pz[i], wl);
          }
        }
      }
      if constexpr (BUCK) {
        c_wmax = wmax; c_sx = sx; c_sy = sy; c_sz = sz;
        c_maxd = whi != 0u ? __uint_as_float(whi - 1u) : -1.f;
      }
      }   // active
      if constexpr (TAIL) {
        // streamed points of this thread: k = k0 + i * kstride, i = PPT, PPT + 1, ...
        const float *P = xyz + (size_t)q * N * 3;
        float *T = tail_td + (size_t)q * N;
        float tb = -1.f, tx = 0.f, ty = 0.f, tz = 0.f;
        int tk = 0;
        if (j == 1) {
          for (int k = k0 + PPT * kstride; k < N; k += kstride) {
            const float x = P[(size_t)k * 3 + 0], y = P[(size_t)k * 3 + 1], z = P[(size_t)k * 3 + 2];
            const float mag = pn2_sq3(x, y, z);
            const bool valid = !((double)mag <= 1e-3);
            const float d = pn2_sq3(x - ox[0], y - oy[0], z - oz[0]);
            const float d2 = valid ? fminf(d, 1e10f) : -1.f;
            T[k] = d2;
            if (d2 > tb) { tb = d2; tk = k; tx = x; ty = y; tz = z; }
          }
        } else {
#pragma unroll 4
          for (int k = k0 + PPT * kstride; k < N; k += kstride) {
            const float x = P[(size_t)k * 3 + 0], y = P[(size_t)k * 3 + 1], z = P[(size_t)k * 3 + 2];
            const float d = pn2_sq3(x - ox[0], y - oy[0], z - oz[0]);
            const float d2 = fps_min(d, T[k]);
            T[k] = d2;
            if (d2 > tb) { tb = d2; tk = k; tx = x; ty = y; tz = z; }
          }
        }
        const u64 tpk = tb >= 0.f ? fps_pack(tb, (unsigned)tk, L) : 0ull;
        const u64 twmax = fps_wave_max_key(tpk);
        if (twmax > wmax) {                                       // wave-uniform
          const u64 who = __ballot(tpk == twmax);
          const int wl = __ffsll((long long)who) - 1;
          wmax = twmax;
          sx = pn2_readlane_f32(tx, wl); sy = pn2_readlane_f32(ty, wl); sz = pn2_readlane_f32(tz, wl);
        }
      }
      if (lane == 0) {
        FpsSlot &sl = lds_slots[j & 1][0][wave];
        sl.packed = wmax; sl.x = sx; sl.y = sy; sl.z = sz;
      }
    } else {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float best = -1.f;
      int bi = 0;
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int s = c * PPT + i;
        const float d = pn2_sq3(px[s] - ox[c], py[s] - oy[c], pz[s] - oz[c]);
        const float d2 = fps_min(d, td[s]);
        td[s] = d2;
        if (d2 > best) { best = d2; bi = i; }
      }
      u64 pk = 0ull;
      if (best >= 0.f) pk = fps_pack(best, (unsigned)(k0 + bi * kstride), L);
      const u64 wmax = fps_wave_max_key(pk);
      float sx = p0x[c], sy = p0y[c], sz = p0z[c];
      bool writer;
      if (wmax == 0ull) {
        writer = (lane == 0);
      } else {
        writer = (pk == wmax);
        if (writer) {
          sx = px[c * PPT]; sy = py[c * PPT]; sz = pz[c * PPT];
#pragma unroll
          for (int i = 1; i < PPT; ++i) {
            if (bi == i) { sx = px[c * PPT + i]; sy = py[c * PPT + i]; sz = pz[c * PPT + i]; }
          }
        }
      }
      if (writer) {
        FpsSlot &sl = lds_slots[j & 1][c][wave];
        sl.packed = wmax; sl.x = sx; sl.y = sy; sl.z = sz;
      }
    }
    }
    __syncthreads();

    // ---- wave c: block winner of cloud c -> publish -> sweep the cluster -> final winner ----
    if (wave < NC) {
      const int c = wave;
      const FpsSlot *buf = lds_slots[j & 1][c];
      const int s16 = lane & 15;
      const u64 mine = (s16 < NW) ? buf[s16].packed : 0ull;
      const u64 bmax = pn2_readlane_u64(pn2_row16_max_u64(mine), 0);
      const u64 whoB = __ballot(mine == bmax && s16 < NW);
      const int ws = (__ffsll((long long)whoB) - 1) & 15;
      const float bx = buf[ws].x, by = buf[ws].y, bz = buf[ws].z;

      u64 *par = slots + ((size_t)(q * NC + c) * 2 + (size_t)(j & 1)) * (kCoopFields * kCoopMaxG);
      if (lane < kCoopFields) {
        unsigned v = lane == 0 ? (unsigned)(bmax >> 32)
                   : lane == 1 ? (unsigned)bmax
                   : lane == 2 ? __float_as_uint(bx)
                   : lane == 3 ? __float_as_uint(by) : __float_as_uint(bz);
        coop_store(par + lane * kCoopMaxG + g, ((u64)(unsigned)j << 32) | v);
      }
      int total = kCoopFields * G;
      bool failed = false;
      for (int qq = 0; qq < total; qq += 64) {
        const int l = qq + lane;
        const bool act = l < total;
        const int f = act ? l / G : 0;
        const int gg = act ? l - f * G : 0;
        unsigned spins = 0;
        u64 v;
        for (;;) {
          v = act ? coop_load(par + f * kCoopMaxG + gg) : ((u64)(unsigned)j << 32);
          if (__all((unsigned)(v >> 32) == (unsigned)j)) break;
          if (++spins > kCoopSpinLimit) { failed = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (failed) break;
        if (act) lds_vals[c][f][gg] = (unsigned)v;
      }
      u64 cand = 0ull;
      if (!failed && lane < G) cand = ((u64)lds_vals[c][0][lane] << 32) | lds_vals[c][1][lane];
      const u64 cmax = fps_wave_max_key(cand);
      const u64 who = __ballot(!failed && lane < G && cand == cmax);
      const int wg = who ? (__ffsll((long long)who) - 1) : 0;
      if (lane == 0) {
        FpsFinal &fin = lds_fin[j & 1][c];
        fin.abort = failed ? 1 : 0;
        fin.x = __uint_as_float(lds_vals[c][2][wg]);
        fin.y = __uint_as_float(lds_vals[c][3][wg]);
        fin.z = __uint_as_float(lds_vals[c][4][wg]);
        fin.k = cmax ? (int)fps_unrank(~(unsigned)cmax, L) : 0;
        if (failed) atomicExch(status, 1);
      }
    }
    __syncthreads();
    int aborted = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const FpsFinal &fin = lds_fin[j & 1][c];
      aborted |= fin.abort;
      ox[c] = fin.x; oy[c] = fin.y; oz[c] = fin.z;
      if (g == 0 && t == 0) idxs[(size_t)(q * NC + c) * m + j] = fin.k;
    }
    if (aborted) return;
  }
}

// ---- cluster kernel with SEVERAL samples per hand-off (round 4) ----------------------------------------------------
// A round of fps_coop_kernel is ~1.1 us of scan and ~1.9 us of inter-workgroup hand-off, and the hand-off is at the price
// the hardware asks for a store -> poll round trip through L2.  The lever is FEWER hand-offs.  Every wave owns SUB
// "sub-blobs" (consecutive records of the spatially binned cloud, PPT / SUB slots per lane each) and caches, per sub-blob,
// its arg-max candidate c (key + coordinates) and `second`, the largest running distance among its OTHER points.  One
// hand-off carries the candidates of ALL sub-blobs of the cloud (<= 64: one per lane of the sweeping wave), and every
// workgroup then runs the same "mini FPS" over them:
//   * sample 1 = the largest key — the reference's next sample;
//   * every lane updates the running distance of ITS candidate with the accepted sample (the scan's arithmetic, so the
//     value is the one the owning wave will compute).  Running distances only decrease: while d(c) stays STRICTLY above
//     `second`, c is still its sub-blob's arg-max and (d(c), rank) its exact key — the sub-blob is CLEAN.  Otherwise it
//     is DIRTY: all that is known is that every one of its points is at most `second` away;
//   * the best clean candidate is the reference's next sample iff its distance is STRICTLY above every dirty bound: all
//     other points of clean sub-blobs have smaller keys (keys are unique), all points of dirty ones smaller distances.
//     Otherwise the list ends and the waves catch up.
// Up to kMultiK samples per hand-off (numpy model on binned 50k clouds: 4.8 on average with 32 sub-blobs, 5.4 with 64;
// a rule that declares every sub-blob an accepted sample can REACH dirty gives 2.4 / 2.9).  The waves then apply the
// accepted samples that reach their sub-blobs (box test, as in fps_coop_kernel) in one pass.  Bit-exact by construction:
// every accepted sample is proven to be the arg-max of the full update; checked against the lane-accurate oracle like
// every other variant.
constexpr int kMultiK = 16;
constexpr int kMultiFields = 6;                                     // hi, lo, x, y, z, second
constexpr int kMultiWords = kMultiFields * 64;                      // one parity: field-major, 64 sub-blobs
// per cloud: [2 parities][6][64] candidate granules
constexpr size_t kMultiCloudBytes = 2ull * kMultiWords * sizeof(u64);
static_assert(kMultiCloudBytes <= kCoopCloudBytes, "hand-off area of a cloud");

struct FpsMultiFin {
  int cnt, abort, pad0, pad1;
  float x[kMultiK], y[kMultiK], z[kMultiK];
};

__device__ __forceinline__ float fps_box_lb(float b0x, float b0y, float b0z, float b1x, float b1y, float b1z,
                                            float ox, float oy, float oz) {
  const float ex = fmaxf(fmaxf(b0x - ox, ox - b1x), 0.f), ey = fmaxf(fmaxf(b0y - oy, oy - b1y), 0.f),
              ez = fmaxf(fmaxf(b0z - oz, oz - b1z), 0.f);
  return (ex * ex + ey * ey + ez * ez) * 0.999998f;
}

template <int PPT, int SUB>
__global__ __launch_bounds__(1024) void fps_multi_kernel(int B, int N, int m, int L, int G,
                                                         const float *__restrict__ xyz,
                                                         const float4 *__restrict__ rec,
                                                         int *__restrict__ idxs,
                                                         u64 *__restrict__ slots,
                                                         int *__restrict__ status) {
  constexpr int NW = 16;
  constexpr int WSUB = NW * SUB;                                    // sub-blobs of a workgroup
  static_assert(SUB == 1 || SUB == 2, "one or two sub-blobs per wave");
  __shared__ int permk[1024 * PPT];                                 // reference index of every point slot
  __shared__ unsigned s_val[kMultiFields][WSUB];                    // this workgroup's cached candidates
  __shared__ unsigned s_cand[kMultiFields][64];                     // this hand-off's candidates of all sub-blobs
  __shared__ FpsMultiFin s_fin;

  const int q = blockIdx.x % B;                                     // cloud; a cluster shares blockIdx % 8 when B % 8 == 0
  const int g = blockIdx.x / B;
  const int t = threadIdx.x;
  const int lane = pn2_lane();
  const int wave = t >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int nsub = G * WSUB;                                        // <= 64 (host)

  float px[PPT], py[PPT], pz[PPT], td[PPT];
  float b0x[SUB], b0y[SUB], b0z[SUB], b1x[SUB], b1y[SUB], b1z[SUB], c_maxd[SUB];
  bool has_valid[SUB];
#pragma unroll
  for (int h = 0; h < SUB; ++h) {
    b0x[h] = b0y[h] = b0z[h] = 3.0e38f;
    b1x[h] = b1y[h] = b1z[h] = -3.0e38f;
    has_valid[h] = false;
  }
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    // wave (g, w) owns the records [b 64 PPT, +64 PPT) of the binned cloud, b = w G + g (neighbouring blobs on different
    // workgroups and SIMDs); its sub-blob h the slots [h PPT / SUB, (h + 1) PPT / SUB)
    const int pos = (wave * G + g) * (64 * PPT) + i * 64 + lane;
    float x = 0.f, y = 0.f, z = 0.f;
    int kk = 0;
    bool valid = false;
    if (pos < N) {
      const float4 r = rec[(size_t)q * N + pos];
      x = r.x; y = r.y; z = r.z; kk = __float_as_int(r.w);
      const float mag = pn2_sq3(x, y, z);
      valid = !((double)mag <= 1e-3);
    }
    permk[wave * (64 * PPT) + i * 64 + lane] = kk;
    px[i] = x; py[i] = y; pz[i] = z;
    td[i] = valid ? 1e10f : -1.f;
#pragma unroll
    for (int h = 0; h < SUB; ++h) {
      if (i >= h * PPT / SUB && i < (h + 1) * PPT / SUB && valid) {
        has_valid[h] = true;
        b0x[h] = fminf(b0x[h], x); b0y[h] = fminf(b0y[h], y); b0z[h] = fminf(b0z[h], z);
        b1x[h] = fmaxf(b1x[h], x); b1y[h] = fmaxf(b1y[h], y); b1z[h] = fmaxf(b1z[h], z);
      }
    }
    // four record loads in flight at a time: with all PPT of them hoisted to the top (16 bytes each) the prologue, not
    // the sampling loop, would set the kernel's register peak and push loop-carried values into scratch
    if (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int h = 0; h < SUB; ++h) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      b0x[h] = fminf(b0x[h], __shfl_xor(b0x[h], o)); b0y[h] = fminf(b0y[h], __shfl_xor(b0y[h], o));
      b0z[h] = fminf(b0z[h], __shfl_xor(b0z[h], o));
      b1x[h] = fmaxf(b1x[h], __shfl_xor(b1x[h], o)); b1y[h] = fmaxf(b1y[h], __shfl_xor(b1y[h], o));
      b1z[h] = fmaxf(b1z[h], __shfl_xor(b1z[h], o));
    }
    b0x[h] = pn2_readlane_f32(b0x[h], 0); b0y[h] = pn2_readlane_f32(b0y[h], 0); b0z[h] = pn2_readlane_f32(b0z[h], 0);
    b1x[h] = pn2_readlane_f32(b1x[h], 0); b1y[h] = pn2_readlane_f32(b1y[h], 0); b1z[h] = pn2_readlane_f32(b1z[h], 0);
    // some point of the sub-blob takes part (a NaN point does — EXT/src/sampling_gpu.cu:100-101 only skips |p|^2 <= 1e-3 —
    // without entering the box: its running distance never changes, the forced first scan files it as a candidate)
    has_valid[h] = __ballot(has_valid[h]) != 0ull;
    c_maxd[h] = has_valid[h] ? 1e10f : -1.f;
    if (lane == 0) {
      const int sb = wave * SUB + h;
#pragma unroll
      for (int f = 0; f < kMultiFields; ++f) s_val[f][sb] = 0u;
    }
  }
  const float *P0 = xyz + (size_t)q * N * 3;
  const float p0x = P0[0], p0y = P0[1], p0z = P0[2];
  if (t == 0) {
    s_fin.cnt = 1; s_fin.abort = 0;
    s_fin.x[0] = p0x; s_fin.y[0] = p0y; s_fin.z[0] = p0z;
    if (g == 0) idxs[(size_t)q * m] = 0;
  }
  __syncthreads();

  u64 *const base = slots + (size_t)q * (2 * kMultiWords);
  int j = 1;                                                        // samples emitted so far
  unsigned e = 0;                                                   // hand-offs so far
  while (j < m) {
    // ---- which (sample, sub-blob) pairs are live: lane l tests sample l / SUB against sub-blob l % SUB
    const int cnt = __builtin_amdgcn_readfirstlane(s_fin.cnt);
    u64 amask;
    {
      const int ls = lane / SUB < kMultiK ? lane / SUB : kMultiK - 1;
      const float sxl = s_fin.x[ls], syl = s_fin.y[ls], szl = s_fin.z[ls];
      const int h1 = SUB == 2 ? (lane & 1) : 0;
      const float lb = fps_box_lb(h1 ? b0x[SUB - 1] : b0x[0], h1 ? b0y[SUB - 1] : b0y[0], h1 ? b0z[SUB - 1] : b0z[0],
                                  h1 ? b1x[SUB - 1] : b1x[0], h1 ? b1y[SUB - 1] : b1y[0], h1 ? b1z[SUB - 1] : b1z[0],
                                  sxl, syl, szl);
      const float md = h1 ? c_maxd[SUB - 1] : c_maxd[0];
      const bool hv = h1 ? has_valid[SUB - 1] : has_valid[0];
      // first round: every sub-blob with a valid point takes the first sample (its cached candidate does not exist yet)
      amask = __ballot(lane < cnt * SUB && (!(lb >= md) || (e == 0u && hv)));
    }
#pragma unroll
    for (int h = 0; h < SUB; ++h) {
      constexpr u64 kEven = 0x5555555555555555ull;
      u64 mh = SUB == 2 ? (amask & (kEven << h)) : amask;
      if (mh == 0ull) continue;                                     // wave-uniform: cached candidate stands
      const int lo_i = h * PPT / SUB, hi_i = (h + 1) * PPT / SUB;
      while (mh) {
        const int bl = __ffsll((long long)mh) - 1;
        mh &= mh - 1;
        // (the sample again from LDS, by a uniform address: three registers per lane the 26-slot shape does not have)
        const int sl = bl / SUB;
        const float ox = pn2_readlane_f32(s_fin.x[sl], 0), oy = pn2_readlane_f32(s_fin.y[sl], 0),
                    oz = pn2_readlane_f32(s_fin.z[sl], 0);
        // two slots per packed instruction (v_pk_add / mul / fma_f32): the IEEE operations of pn2_sq3 in its order —
        // fma(dz, dz, fma(dx, dx, dy * dy)) — on both halves; an odd slot count leaves one scalar tail
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 o2x = {ox, ox}, o2y = {oy, oy}, o2z = {oz, oz};
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
          if (i >= lo_i && i < hi_i) {
            if (((i - lo_i) & 1) == 0 && i + 1 < hi_i) {
              const f2 dx = f2{px[i], px[i + 1]} - o2x, dy = f2{py[i], py[i + 1]} - o2y, dz = f2{pz[i], pz[i + 1]} - o2z;
              const f2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
              td[i] = fps_min(d.x, td[i]);
              td[i + 1] = fps_min(d.y, td[i + 1]);
            } else if (((i - lo_i) & 1) == 0) {
              const float d = pn2_sq3(px[i] - ox, py[i] - oy, pz[i] - oz);
              td[i] = fps_min(d, td[i]);
            }
          }
        }
      }
      // the lane's two largest running distances: b1 >= b2 (v_max + v_med3 per slot)
      // (pinned instructions: as plain expressions the compiler canonicalises every operand that came out of the v_min
      // asm (a v_max x, x each) and SINKS the second chain below the tied-lane loop, keeping all PPT prefix maxima alive
      // for it — 26 registers this kernel does not have)
      float best = -1.f, best2 = -1.f;
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        if (i >= lo_i && i < hi_i) {
          asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(best2) : "v"(best), "v"(td[i]));
          asm volatile("v_max_f32 %0, %0, %1" : "+v"(best) : "v"(td[i]));
        }
      }
      const unsigned hi = best >= 0.f ? __float_as_uint(best) + 1u : 0u;
      const unsigned whi = pn2_wave_max_u32(hi);                    // wave-uniform
      // (every result goes to the LDS slot where it is produced: carried to one common store they cost five registers
      // the 26-slot shape does not have)
      const int sb = wave_u * SUB + h;
      unsigned best_lo = 0u;
      if (whi != 0u) {
        const unsigned target = whi - 1u;
        u64 tied = __ballot(hi == whi);
        int wl = 0, wbi = lo_i;
        while (tied) {                                              // scalar loop over the tied lanes (usually one)
          const int l = __ffsll((long long)tied) - 1;
          tied &= tied - 1;
#pragma unroll
          for (int i = 0; i < PPT; ++i) {
            if (i >= lo_i && i < hi_i) {
              if ((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(td[i]), l) == target) {   // wave-uniform
                const unsigned k = (unsigned)__builtin_amdgcn_readfirstlane(permk[wave_u * (64 * PPT) + i * 64 + l]);
                const unsigned lo = ~fps_rank(k, L);
                if (lo >= best_lo) { best_lo = lo; wl = l; wbi = i; }
              }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
          if (i >= lo_i && i < hi_i) {
            if (wbi == i && lane == wl) {                           // (wbi == i: wave-uniform, a scalar branch)
              s_val[2][sb] = __float_as_uint(px[i]); s_val[3][sb] = __float_as_uint(py[i]);
              s_val[4][sb] = __float_as_uint(pz[i]);
            }
          }
        }
        // largest running distance among the sub-blob's OTHER points: the winner's lane contributes its second largest
        const float other = lane == wl ? best2 : best;
        const unsigned second = pn2_wave_max_u32(other >= 0.f ? __float_as_uint(other) + 1u : 0u);
        if (lane == 0) s_val[5][sb] = second;
      } else if (lane == 0) {
        s_val[2][sb] = __float_as_uint(p0x); s_val[3][sb] = __float_as_uint(p0y); s_val[4][sb] = __float_as_uint(p0z);
        s_val[5][sb] = 0u;
      }
      c_maxd[h] = whi != 0u ? __uint_as_float(whi - 1u) : -1.f;
      if (lane == 0) { s_val[0][sb] = whi; s_val[1][sb] = best_lo; }
    }
    ++e;
    __syncthreads();

    // ---- wave 0: publish this workgroup's candidates, collect the peers', mini FPS over all of them
    if (wave == 0) {
      u64 *par = base + (size_t)(e & 1u) * kMultiWords;
      const bool live = lane < nsub;
      const bool own = live && (lane / WSUB) == g;
      if (own) {
        const int sb = lane - g * WSUB;
#pragma unroll
        for (int f = 0; f < kMultiFields; ++f) {
          const unsigned val = s_val[f][sb];
          coop_store(par + f * 64 + lane, ((u64)e << 32) | (u64)val);
          s_cand[f][lane] = val;
        }
      } else if (!live) {
#pragma unroll
        for (int f = 0; f < kMultiFields; ++f) s_cand[f][lane] = 0u;
      }
      bool failed = false;
      unsigned spins = 0;
      {
        u64 w[kMultiFields];
#pragma unroll
        for (int f = 0; f < kMultiFields; ++f) w[f] = 0ull;
        for (;;) {
          bool ok = true;
          if (live && !own) {
#pragma unroll
            for (int f = 0; f < kMultiFields; ++f) w[f] = coop_load(par + f * 64 + lane);
#pragma unroll
            for (int f = 0; f < kMultiFields; ++f) ok = ok && (unsigned)(w[f] >> 32) == e;
          }
          if (__all(ok)) break;
          if (++spins > kCoopSpinLimit) { failed = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        // everything the mini FPS needs goes through LDS (a lane's own entries, the winner's coordinates by a uniform
        // read): at 26 point slots per lane the kernel has no registers to hold 64 candidates x 6 fields next to them
        if (live && !own) {
#pragma unroll
          for (int f = 0; f < kMultiFields; ++f) s_cand[f][lane] = (unsigned)w[f];
        }
      }
      int acc = 0;
      if (!failed) {
        // lane = sub-blob: its candidate's key, coordinates and CURRENT running distance (kept exact below), and the
        // largest distance among its other points as of this hand-off
        unsigned hi = s_cand[0][lane];
        const unsigned lo = s_cand[1][lane], sec = s_cand[5][lane];
        const float cx = __uint_as_float(s_cand[2][lane]), cy = __uint_as_float(s_cand[3][lane]),
                    cz = __uint_as_float(s_cand[4][lane]);
        bool dirty = false;
        const int kmax = m - j < kMultiK ? m - j : kMultiK;
        for (int k = 0; k < kmax; ++k) {
          unsigned mhi, mlo;
          const u64 who = pn2_wave_argmax_u32x2(dirty ? 0u : hi, dirty ? 0u : lo, mhi, mlo);
          if (k > 0) {
            const unsigned db = pn2_wave_max_u32(dirty ? sec : 0u);
            if (!(mhi > db)) break;                                 // (also: no clean candidate left)
          }
          float sx = p0x, sy = p0y, sz = p0z;
          int kidx = 0;
          if (mhi != 0u) {
            const int wl = __ffsll((long long)who) - 1;
            sx = pn2_readlane_f32(cx, wl); sy = pn2_readlane_f32(cy, wl); sz = pn2_readlane_f32(cz, wl);
            kidx = (int)fps_unrank(~mlo, L);
          }
          if (lane == 0) {
            s_fin.x[k] = sx; s_fin.y[k] = sy; s_fin.z[k] = sz;
            if (g == 0) idxs[(size_t)q * m + j + k] = kidx;
          }
          ++acc;
          if (mhi == 0u) break;                                     // no valid point anywhere: index 0 (sampling_gpu.cu:90-91)
          // the candidate's own update, with the arithmetic of the scan: while it stays STRICTLY above every other
          // point of its sub-blob (whose distances only decrease) it is still the sub-blob's arg-max, with this key
          if (hi != 0u) {
            const float nd = fps_min(pn2_sq3(cx - sx, cy - sy, cz - sz), __uint_as_float(hi - 1u));
            hi = __float_as_uint(nd) + 1u;
            if (!(hi > sec)) dirty = true;                          // (the sample's own sub-blob: nd = 0)
          }
        }
      }
      if (lane == 0) {
        s_fin.cnt = acc;
        if (failed) { s_fin.abort = 1; atomicExch(status, 1); }
      }
    }
    __syncthreads();
    if (s_fin.abort) return;
    j += __builtin_amdgcn_readfirstlane(s_fin.cnt);
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

// Kernel selection, from measured per-round costs on MI355X (tools/microbench.py
// MB_FPS_SWEEP): resident 512-thread workgroup ~0.55 us + 0.044 us per point slot,
// resident 1024-thread ~0.5 + 0.12 per slot, cooperative ~1.9 + 0.044 per slot
// (the inter-workgroup hand-off costs ~1.3 us), streaming ~0.25 us per 1000 points.
//  => one workgroup up to 16k points; above that a cluster with the SMALLEST G
//     that keeps <= 16 point slots per lane (larger clusters sweep more granules);
//     every cluster workgroup must be resident, hence B*G <= 256.
struct FpsPlan {
  int mode;  // 0 resident, 1 cooperative, 2 streaming, 3 cooperative with a streamed tail, 4 bucketed (PPT = slots per bucket),
             // 5 cluster with several samples per hand-off (fps_multi_kernel; NC = sub-blobs per wave)
  int G, BS, PPT;
  int NC;    // cooperative: clouds per cluster
};

const int kPptSteps[] = {1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 20, 24, 26};   // 26: only the 1024-thread cluster kernel (21 + 4 PPT VGPRs <= 128)

int round_ppt(int ppt) {
  for (int c : kPptSteps)
    if (c >= ppt) return c;
  return -1;
}

// Test hook (pn2_fps_set_plan_override): force a kernel variant / cluster shape so that every variant can be checked
// against the oracle on shapes the heuristic would route elsewhere.  Process-global, unset by default; never read from
// the environment.
struct FpsOverride {
  int mode;      // -1 none | 0 resident | 1 cooperative | 2 streaming | 3 cooperative with a streamed tail | 4 bucketed | 5 multi
  int G, NC, coop_bs, bs;   // 0 = heuristic
};
FpsOverride g_fps_override = {-1, 0, 0, 0, 0};
std::atomic<bool> g_fps_bucketing{true};     // (pn2_fps_set_bucketing: tests and measurements compare both forms)
std::atomic<bool> g_fps_multi{true};         // (pn2_fps_set_multi: several samples per hand-off for cluster-sized clouds)

// Number of CUs of the current device (a CPX/DPX partition reports its own count); every cluster workgroup must be
// resident at once, so the cluster shapes are sized from this instead of a hard-coded 256.
int device_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus[dev] = v;
  }
  return cus[dev];
}

// `bucketing`: -1 = the process-wide switch, 0 / 1 = as given (pn2_fps_workspace_bytes sizes both layouts without touching
// the switch other host threads read)
FpsPlan fps_plan(int B, int N, int m, bool few_cus = false, bool fewest = false, int bucketing = -1) {
  const bool use_bucketing = bucketing < 0 ? g_fps_bucketing.load(std::memory_order_relaxed) : bucketing != 0;
  FpsPlan p = {2, 1, 1024, 0, 1};
  if (m <= 1) { p.mode = 0; p.BS = 512; p.PPT = 1; return p; }
  const FpsOverride ov = g_fps_override;
  const int ncus = device_cus();
  const long long max_coop_wgs = 2LL * ncus < kCoopMaxWorkgroups ? 2LL * ncus : kCoopMaxWorkgroups;
  const bool want_coop = ov.mode == 1;
  const bool want_resident = ov.mode == 0;
  const bool want_stream = ov.mode == 2;
  if (want_stream) return p;

  // cooperative candidate: NC clouds per cluster (NC | B), G workgroups per cluster, NC*PPT <= 28
  // point slots per lane, (B/NC)*G <= 512 resident workgroups.
  FpsPlan c = {-1, 1, 512, 0, 1};
  {
    int best_nc = 1, best_g = 2;
    while (best_g < kCoopMaxG && (N + best_g * 512 - 1) / (best_g * 512) > 16) best_g *= 2;
    // Measured at 32 x 50k -> 2048 (profiles/r01_fps_variant_sweep.jsonl): (NC,G) = (1,8) 5.3 ms,
    // (2,8) 6.8, (2,16) 6.3, (4,16) 6.6, (4,32) 9.7: batching clouds per cluster does NOT pay (the
    // hand-off cost grows with the cluster size and the per-cloud serial work dominates), so the
    // default stays one cloud per cluster; NC > 1 remains available through the test hook.
    int G = best_g, NC = best_nc;
    if (ov.NC) {
      const int want = ov.NC;
      if ((want == 1 || want == 2 || want == 4) && B % want == 0) { NC = want; G = (best_g / best_nc) * want; }
    }
    if (ov.G) {
      const int want = ov.G;
      if (want >= 2 && want <= kCoopMaxG && (want & (want - 1)) == 0) G = want;
    }
    // few_cus (PN2_FPS_FEW_CUS: the sampling runs on a side stream next to the MFMA kernels of a training step):
    // half as many cluster workgroups of 1024 threads.  Alone that is slower (1024-thread scans: 5.9 vs 4.8 ms at
    // 32 x 50k -> 2048), but a resident FPS workgroup pins 160 of the 512 VGPRs per lane on its CU for its whole
    // run and halves the occupancy of every co-running 8-wave GEMM workgroup there; on 128 CUs instead of 256 the
    // other half of the chip runs the step undisturbed and the step is 0.8 ms shorter (tools/corun_probe.py).
    // Round 3: FEWER still — the 1024-thread kernel needs 21 + 4 PPT registers, so a workgroup can hold up to 26 points
    // per lane (125 VGPRs: the whole register file of its CU) and a 50k-point cloud fits TWO workgroups: 64 CUs for the
    // SA1 sampling of 32 clouds instead of 128.  A round gets slower (26 instead of 14 points per lane to scan, 3.0 vs
    // 2.6 us) but the CU-time the co-running step loses halves.
    int cbs = 512;
    if (NC == 1 && !ov.G && few_cus && G >= 4) {
      int g2 = G / 2;
      while (fewest && g2 > 2 && round_ppt((N + (g2 / 2) * 1024 - 1) / ((g2 / 2) * 1024)) > 0 &&
             round_ppt((N + (g2 / 2) * 1024 - 1) / ((g2 / 2) * 1024)) <= 26)
        g2 /= 2;
      const int wp = round_ppt((N + g2 * 1024 - 1) / (g2 * 1024));
      if (wp >= 8 && wp <= 26) { cbs = 1024; G = g2; }
    }
    if (ov.coop_bs == 1024 && NC == 1) cbs = 1024;
    const int ppt = round_ppt((N + G * cbs - 1) / (G * cbs));
    if (NC > 1 && NC * ppt > 28) NC = 1;                  // multi-cloud kernels are built for <= 28 slots
    // residency: 512-thread cluster workgroups fit at least three to a CU (two are counted on), 1024-thread ones one
    const long long cap = cbs == 1024 ? (long long)ncus : max_coop_wgs;
    if (G <= kCoopMaxG && (long long)(B / NC) * G <= cap && ppt > 0 && NC * ppt <= 28 &&
        (cbs == 512 ? ppt <= 24 : (ppt >= 8 && ppt <= 26))) {
      c.mode = 1; c.G = G; c.PPT = ppt; c.NC = NC; c.BS = cbs;
    }
  }
  // resident candidate
  // workgroup width: a round is one dependent chain per wave, so fewer, fatter waves win until the scan
  // itself dominates: 256 threads (one wave per SIMD) up to 2048 points, 512 up to 8192, then 1024.
  FpsPlan r = {-1, 1, 512, 0, 1};
  {
    int bs = N <= 2048 ? 256 : (N <= 512 * 16 ? 512 : 1024);
    if (ov.bs == 256 || ov.bs == 512 || ov.bs == 1024) bs = ov.bs;
    while (bs < 1024 && (N + bs - 1) / bs > (bs == 256 ? 16 : 16)) bs *= 2;
    if (N <= kFpsResidentMaxN && (N + bs - 1) / bs <= 24) { r.mode = 0; r.BS = bs; r.PPT = round_ppt((N + bs - 1) / bs); }
  }

  // cooperative with a streamed tail: the cloud exceeds the register capacity of every cluster shape above.  One
  // 1024-thread workgroup per CU (G = 256 / B clusters members), 20 register slots per thread, the rest streamed.
  FpsPlan h = {-1, 1, 1024, 20, 1};
  {
    int G = 1;
    while (G * 2 <= kCoopMaxG && (long long)B * (G * 2) <= ncus) G *= 2;
    if (ov.G) {
      const int want = ov.G;
      if (want >= 1 && want <= kCoopMaxG && (want & (want - 1)) == 0 && (long long)B * want <= ncus) G = want;
    }
    if ((long long)B * G <= ncus) { h.mode = 3; h.G = G; }   // 1024 threads x 20 slots: one workgroup per CU
  }
  const bool want_hybrid = ov.mode == 3;
  if (want_hybrid && h.mode == 3) return h;

  // bucketed: one workgroup per cloud, 64 S records per bucket, at most 1024 buckets.  Chosen where a cluster would be
  // (clouds beyond 16k points): 32 x 50k -> 2048 in 3.x ms on 32 CUs against 5.3 ms on 128 / 7.3 ms on 64 (profiles/HISTORY.md 4c)
  FpsPlan k = {-1, 1, 1024, 0, 1};
  {
    int S = 1;
    while (S < 8 && (long long)N > 65536LL * S) S *= 2;
    if ((long long)N <= 65536LL * S) { k.mode = 4; k.PPT = S; }
  }
  if (ov.mode == 4 && k.mode == 4) return k;
  // ... when the batch does not fit the chip's registers any more (~20k points per CU): 64 x 200k points 42 -> 13 ms per
  // step of samplings; a batch that does fit (32 x 50k: 1.6M of 5.2M slots) is as fast on a cluster (4.9 vs 4.8 ms)
  if (ov.mode < 0 && use_bucketing && k.mode == 4 && N > 16384 && (long long)B * N > (long long)ncus * 1024 * 20) return k;

  // several samples per hand-off: two (up to 53k points) or four 1024-thread workgroups per cloud over the binned records,
  // 64 sub-blobs per cloud.  Replaces the one-sample cluster wherever it fits (fewer CUs AND fewer hand-offs).
  FpsPlan mu = {-1, 2, 1024, 0, 2};
  {
    int G = 2;
    if ((long long)N > 2LL * 1024 * 26) G = 4;
    if (ov.mode == 5 && ov.G == 4) G = 4;
    const int raw = (N + G * 1024 - 1) / (G * 1024);
    // G 16 SUB <= 64 sub-blobs, one per lane of the sweeping wave: two per wave wherever two workgroups hold the cloud.
    // (With scalar distance arithmetic the 24..26-slot two-sub-blob instantiations needed 30-70 registers more than a
    // 1024-thread workgroup has — 232-316 B of scratch, 8 vs 4.6 ms; with the packed form 8-40 B: 2.50 vs 2.95 ms at 25 slots.)
    int sub = G == 2 ? 2 : 1;
    if (ov.mode == 5 && ov.NC == 1) sub = 1;
    if (ov.mode == 5 && ov.NC == 2 && G == 2) sub = 2;
    int ppt = -1;
    if (sub == 2) { for (int c2 : {10, 12, 14, 16, 20, 24, 25, 26}) if (c2 >= raw) { ppt = c2; break; } }
    else { for (int c2 : {14, 16, 20, 24, 25, 26}) if (c2 >= raw) { ppt = c2; break; } }
    if (ppt > 0 && (long long)B * G <= ncus) { mu.mode = 5; mu.G = G; mu.PPT = ppt; mu.NC = sub; }
  }
  if (ov.mode == 5 && mu.mode == 5) return mu;
  if (ov.mode < 0 && use_bucketing && g_fps_multi.load(std::memory_order_relaxed) && mu.mode == 5 && N > 16384 &&
      c.mode == 1 && !(r.mode == 0 && N <= 16384))
    return mu;

  if (want_coop && c.mode == 1) return c;
  if (want_resident && r.mode == 0) return r;
  if (r.mode == 0 && (N <= 16384 || c.mode != 1)) return r;
  if (c.mode == 1) return c;
  if (h.mode == 3) return h;
  return p;
}

// Every workgroup of a cluster kernel spins on its peers, so the whole grid must be resident at once.  The launch is
// admitted only if the occupancy the runtime reports for THIS kernel on THIS device covers the grid (one workgroup per
// CU of margin where more than two fit: the API can answer one high, MI355X_MICROARCH.md "Residency"); otherwise the
// caller falls back to the streaming kernel.  A CPX/DPX partition reports its own CU count.  Not covered: CU masks
// set on the stream and other processes sharing the GPU (the bounded spins then raise `status`, which the python
// layer turns into a device-side assertion).
bool coop_fits(const void *kernel, int block, unsigned grid) {
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, int> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  int nb = 0;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({kernel, dev});
    if (it != cache.end()) nb = it->second;
    else {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess) nb = 0;
      (void)hipGetLastError();
      cache[{kernel, dev}] = nb;
    }
  }
  const int allowed = nb >= 3 ? nb - 1 : nb;
  return (long long)grid <= (long long)allowed * device_cus();
}

// Two cluster launches of this process must not overlap (main stream + geometry-prefetch stream): each would hold part
// of the CUs while waiting for workgroups that cannot become resident.  They are chained through one event per device;
// inside a stream capture the graph's own edges order them.
struct CoopSerial {
  hipStream_t s;
  hipEvent_t ev = nullptr;
  explicit CoopSerial(hipStream_t stream) : s(stream) {
    static std::mutex mu;
    static hipEvent_t events[64] = {nullptr};
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return; }
    if (cs != hipStreamCaptureStatusNone) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!events[dev] && hipEventCreateWithFlags(&events[dev], hipEventDisableTiming) != hipSuccess) {
        events[dev] = nullptr;
        (void)hipGetLastError();
        return;
      }
      ev = events[dev];
    }
    (void)hipStreamWaitEvent(s, ev, 0);
  }
  ~CoopSerial() {
    if (ev) (void)hipEventRecord(ev, s);
  }
};

}  // namespace

namespace {
size_t fps_align256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

extern "C" int pn2_fps_set_plan_override(int mode, int G, int NC, int coop_bs, int bs) {
  if (mode < -1 || mode > 5 || G < 0 || NC < 0) return PN2_EINVAL;
  g_fps_override = {mode, G, NC, coop_bs, bs};
  return PN2_OK;
}

// Test / measurement hook: spatial bucketing of the cluster kernels on (default) / off.  Results never depend on it.
extern "C" int pn2_fps_set_bucketing(int on) {
  g_fps_bucketing.store(on != 0, std::memory_order_relaxed);
  return PN2_OK;
}
extern "C" int pn2_fps_get_bucketing(void) { return g_fps_bucketing.load(std::memory_order_relaxed) ? 1 : 0; }
// Measurement hook: the several-samples-per-hand-off cluster kernel on (default) / off.  Results never depend on it.
extern "C" int pn2_fps_set_multi(int on) {
  g_fps_multi.store(on != 0, std::memory_order_relaxed);
  return PN2_OK;
}
extern "C" int pn2_fps_get_multi(void) { return g_fps_multi.load(std::memory_order_relaxed) ? 1 : 0; }

// cluster plans also reserve the streaming kernel's B x N floats behind the hand-off slots: the fallback when the
// cluster would not be resident on this device
namespace {
int bucket_stride(int N, int S) { return (N + 64 * S - 1) / (64 * S) * (64 * S); }
size_t bucketed_bytes(int B, int N, int S) {
  const size_t ns = (size_t)bucket_stride(N, S);
  return fps_align256((size_t)B * ns * 16) + (size_t)B * ns * sizeof(float);
}
}  // namespace

extern "C" size_t pn2_fps_workspace_bytes(int B, int N, int m) {
  if (B <= 0 || N <= 0 || m <= 1) return 0;
  // the largest layout any plan of this shape can ask for (scheduling hints, the bucketing switch and the plan override
  // change the plan, not the workspace a caller has to bring)
  const FpsPlan p = fps_plan(B, N, m, false, false, 0);
  const FpsPlan kb = fps_plan(B, N, m, false, false, 1);
  size_t need = 0;
  // cluster plans: hand-off slots | status | B x N floats (streaming fallback / streamed tail) | B x N binned records
  if (p.mode == 1 || p.mode == 3 || p.mode == 5 || kb.mode == 5)
    need = (size_t)B * kCoopCloudBytes + 256 + fps_align256((size_t)B * (size_t)N * sizeof(float)) + (size_t)B * (size_t)N * 16;
  else if (p.mode == 2) need = (size_t)B * (size_t)N * sizeof(float);
  if (kb.mode == 4 || g_fps_override.mode == 4) {
    int S = 1;
    while (S < 8 && (long long)N > 65536LL * S) S *= 2;
    const size_t nk = bucketed_bytes(B, N, S);
    need = need > nk ? need : nk;
  }
  return need;
}

extern "C" int pn2_furthest_point_sampling(int B, int N, int m, const float *xyz,
                                           void *workspace, size_t workspace_bytes,
                                           int *idxs, void *stream) {
  return pn2_furthest_point_sampling_ex(B, N, m, xyz, workspace, workspace_bytes, idxs, 0, stream);
}

namespace {
int fps_launch(int B, int N, int m, const float *xyz, void *workspace, size_t workspace_bytes, int *idxs, int flags,
               const int *ordered_fail, void *stream);
}

extern "C" int pn2_furthest_point_sampling_ex(int B, int N, int m, const float *xyz,
                                              void *workspace, size_t workspace_bytes,
                                              int *idxs, int flags, void *stream) {
  return fps_launch(B, N, m, xyz, workspace, workspace_bytes, idxs, flags, nullptr, stream);
}

namespace {
// ---- clouds that ARE a sampling order --------------------------------------------------------------------------------
// The centres of SA level l + 1 are sampled from the centres of level l, and those are stored in the order level l's
// sampling picked them (pointnet2_modules.py:38-48: gather_operation(xyz, furthest_point_sample(xyz, npoint))).  Sampling
// m points from a cloud that is itself a farthest-point ORDER returns 0, 1, .., m - 1: point k was the farthest of ALL
// original points from {0 .. k-1}, so it is the farthest of the subset, and the running distances of the subset evolve
// exactly as in the first run.  Two things can break the identity — a tie (another point of the subset at exactly the same
// running distance: the two runs break ties by different index orders) and degenerate rounds (no candidate, a NaN centre,
// duplicates: distance 0) — so it is VERIFIED per cloud, with the kernel's own arithmetic: M[k] = running distance of point
// k when it is picked (fps_order_m_kernel), then every point j replays its running distance against the centres 0 .. m-2
// and must stay strictly below M[k] at every step k != j (fps_order_check_kernel).  No barriers, every point in parallel:
// ~35 us for the three lower levels of the headline against 1.2 ms of sampling rounds; a cloud that fails takes the rounds.
__global__ __launch_bounds__(256) void fps_order_m_kernel(int N, int m, const float *__restrict__ xyz,
                                                         float *__restrict__ Mk, int *__restrict__ fail) {
  extern __shared__ float cen[];                          // [3][kmax]: the centres this workgroup's points look back at
  const int b = blockIdx.x, t = threadIdx.x;
  const float *P = xyz + (size_t)b * N * 3;
  const int k = blockIdx.y * 256 + t, kmax = min(m, (int)(blockIdx.y + 1) * 256);
  for (int i = t; i < kmax; i += 256) { cen[i] = P[(size_t)i * 3]; cen[kmax + i] = P[(size_t)i * 3 + 1]; cen[2 * kmax + i] = P[(size_t)i * 3 + 2]; }
  if (blockIdx.y == 0 && t == 0) fail[b] = m;             // first round that is NOT verified (m: none)
  __syncthreads();
  if (k >= m) return;
  const float x = cen[k], y = cen[kmax + k], z = cen[2 * kmax + k];
  const float init = !((double)pn2_sq3(x, y, z) <= 1e-3) ? 1e10f : -1.f;  // (EXT/src/sampling_gpu.cu:100-101)
  // min over the centres 0 .. k-1: four independent chains (the minimum does not depend on the order — v_min skips a NaN
  // distance wherever it stands, and a skipped point's -1 is below every distance)
  float td[4] = {init, init, init, init};
  int i = 0;
  for (; i + 4 <= k; i += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) td[u] = fps_min(pn2_sq3(x - cen[i + u], y - cen[kmax + i + u], z - cen[2 * kmax + i + u]), td[u]);
  }
  for (; i < k; ++i) td[0] = fps_min(pn2_sq3(x - cen[i], y - cen[kmax + i], z - cen[2 * kmax + i]), td[0]);
  Mk[(size_t)b * m + k] = fps_min(fps_min(td[0], td[1]), fps_min(td[2], td[3]));
}

__global__ __launch_bounds__(256) void fps_order_check_kernel(int N, int m, int L, const float *__restrict__ xyz,
                                                             const float *__restrict__ Mk, int *__restrict__ fail) {
  extern __shared__ float cen[];                          // [3][m] centres | [m] M
  const int b = blockIdx.x, t = threadIdx.x;
  const float *P = xyz + (size_t)b * N * 3;
  float *mk = cen + 3 * m;
  for (int i = t; i < m; i += 256) {
    cen[i] = P[(size_t)i * 3]; cen[m + i] = P[(size_t)i * 3 + 1]; cen[2 * m + i] = P[(size_t)i * 3 + 2];
    mk[i] = Mk[(size_t)b * m + i];
  }
  __syncthreads();
  const int j = blockIdx.y * 256 + t;
  int first = m;                                            // first round this point would (or might) win against point k
  if (j < N) {
    const float x = P[(size_t)j * 3], y = P[(size_t)j * 3 + 1], z = P[(size_t)j * 3 + 2];
    float td = !((double)pn2_sq3(x, y, z) <= 1e-3) ? 1e10f : -1.f;
    // round k: the centre is sample k - 1, the pick must be point k.  Point k wins round k against point j iff its running
    // distance is larger, or equal with the smaller rank (the order of the sampling kernel's key on ties; a negative M[k] —
    // point k skipped — can never be picked).  Eight rounds per step: their distances and M are independent loads /
    // arithmetic, only the eight v_min + compares form a chain; ties and failures take a branch
    auto round = [&](int k, float d, float mkk) {
      td = fps_min(d, td);
      if (!(td < mkk)) {
        const bool loses = td == mkk && mkk >= 0.f && fps_rank((unsigned)j, L) > fps_rank((unsigned)k, L);
        if (j != k && !loses && first == m) first = k;
      }
    };
    int k = 1;
    for (; k + 8 <= m; k += 8) {
      float d[8], mv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        d[u] = pn2_sq3(x - cen[k - 1 + u], y - cen[m + k - 1 + u], z - cen[2 * m + k - 1 + u]);
        mv[u] = mk[k + u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) round(k + u, d[u], mv[u]);
    }
    for (; k < m; ++k) round(k, pn2_sq3(x - cen[k - 1], y - cen[m + k - 1], z - cen[2 * m + k - 1]), mk[k]);
  }
  // (wave minimum, one atomic per wave that saw a failure)
  int wmin = first;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmin = min(wmin, __shfl_xor(wmin, o));
  if (wmin < m && pn2_lane() == 0) atomicMin(fail + b, wmin);
}

int fps_launch(int B, int N, int m, const float *xyz, void *workspace, size_t workspace_bytes, int *idxs, int flags,
               const int *ordered_fail, void *stream) {
  if (B < 0 || N < 0) return PN2_EINVAL;
  if (flags & ~(PN2_FPS_FEW_CUS | PN2_FPS_FEWEST_CUS)) return PN2_EINVAL;
  if (m <= 0 || B == 0) return PN2_OK;  // EXT/src/sampling_gpu.cu:73
  if (N <= 0) return PN2_EINVAL;
  if (!xyz || !idxs) return PN2_ENULL;
  hipStream_t s = (hipStream_t)stream;
  const int bs = ref_opt_n_threads(N);
  int L = 0;
  while ((1 << L) < bs) ++L;
  const FpsPlan plan = fps_plan(B, N, m, flags != 0, (flags & PN2_FPS_FEWEST_CUS) != 0);
  const size_t need = pn2_fps_workspace_bytes(B, N, m);       // same for both cooperative shapes
  if (need) {
    if (!workspace) return PN2_ENULL;
    if (workspace_bytes < need) return PN2_ENOSPC;
  }

  if (plan.mode == 4) {
    const int S = plan.PPT, ns = bucket_stride(N, S);
    float4 *rec = (float4 *)workspace;
    float *td = (float *)((char *)workspace + fps_align256((size_t)B * ns * 16));
    if (((uintptr_t)workspace & 15) != 0) return PN2_EINVAL;
    hipLaunchKernelGGL(fps_bucket_kernel, dim3((unsigned)B), dim3(1024), 0, s, N, ns, xyz, rec);
    switch (S) {
      case 1: hipLaunchKernelGGL(fps_bucketed_kernel<1>, dim3((unsigned)B), dim3(1024), 0, s, N, ns, m, L, xyz, rec, td, idxs); break;
      case 2: hipLaunchKernelGGL(fps_bucketed_kernel<2>, dim3((unsigned)B), dim3(1024), 0, s, N, ns, m, L, xyz, rec, td, idxs); break;
      case 4: hipLaunchKernelGGL(fps_bucketed_kernel<4>, dim3((unsigned)B), dim3(1024), 0, s, N, ns, m, L, xyz, rec, td, idxs); break;
      default: hipLaunchKernelGGL(fps_bucketed_kernel<8>, dim3((unsigned)B), dim3(1024), 0, s, N, ns, m, L, xyz, rec, td, idxs); break;
    }
    return pn2_check_launch();
  }
  const size_t head = (size_t)B * kCoopCloudBytes + 256;
  float *tail = (plan.mode == 1 || plan.mode == 3 || plan.mode == 5) ? (float *)((char *)workspace + head) : (float *)workspace;
  bool fits = true;
  if (plan.mode == 3) {
    u64 *slots = (u64 *)workspace;
    int *status = (int *)((char *)workspace + (size_t)B * kCoopCloudBytes);
    if (hipMemsetAsync(workspace, 0, head, s) != hipSuccess) return pn2_check_launch();
    auto kfn = fps_coop_kernel<1024, 20, 1, true>;
    const unsigned grid = (unsigned)(B * plan.G);
    if ((fits = coop_fits((const void *)kfn, 1024, grid))) {
      CoopSerial chain(s);
      hipLaunchKernelGGL(kfn, dim3(grid), dim3(1024), 0, s, B, N, m, L, plan.G, xyz, idxs, slots, status, tail,
                         (const float4 *)nullptr);
      return pn2_check_launch();
    }
  }
  if (plan.mode == 5) {
    u64 *slots = (u64 *)workspace;
    int *status = (int *)((char *)workspace + (size_t)B * kCoopCloudBytes);
    if (hipMemsetAsync(workspace, 0, head, s) != hipSuccess) return pn2_check_launch();
    const dim3 grid((unsigned)(B * plan.G));
    CoopSerial chain(s);
    float4 *rec = (float4 *)((char *)workspace + head + fps_align256((size_t)B * (size_t)N * sizeof(float)));
    hipLaunchKernelGGL(fps_bucket_kernel, dim3((unsigned)B), dim3(1024), 0, s, N, N, xyz, rec);
#define PN2_FPS_MULTI(PPT, SUB)                                                                       \
  {                                                                                                   \
    auto kfn = fps_multi_kernel<PPT, SUB>;                                                            \
    if ((fits = coop_fits((const void *)kfn, 1024, grid.x)))                                          \
      hipLaunchKernelGGL(kfn, grid, dim3(1024), 0, s, B, N, m, L, plan.G, xyz, (const float4 *)rec,   \
                         idxs, slots, status);                                                        \
  }
    if (plan.NC == 2) {
      switch (plan.PPT) {
        case 10: PN2_FPS_MULTI(10, 2); break;
        case 12: PN2_FPS_MULTI(12, 2); break;
        case 14: PN2_FPS_MULTI(14, 2); break;
        case 16: PN2_FPS_MULTI(16, 2); break;
        case 20: PN2_FPS_MULTI(20, 2); break;
        case 24: PN2_FPS_MULTI(24, 2); break;
        case 25: PN2_FPS_MULTI(25, 2); break;
        case 26: PN2_FPS_MULTI(26, 2); break;
        default: return PN2_EINVAL;
      }
    } else {
      switch (plan.PPT) {
        case 14: PN2_FPS_MULTI(14, 1); break;
        case 16: PN2_FPS_MULTI(16, 1); break;
        case 20: PN2_FPS_MULTI(20, 1); break;
        case 24: PN2_FPS_MULTI(24, 1); break;
        case 25: PN2_FPS_MULTI(25, 1); break;
        case 26: PN2_FPS_MULTI(26, 1); break;
        default: return PN2_EINVAL;
      }
    }
#undef PN2_FPS_MULTI
    if (fits) return pn2_check_launch();
  }
  if (plan.mode == 1) {
    u64 *slots = (u64 *)workspace;
    int *status = (int *)((char *)workspace + (size_t)B * kCoopCloudBytes);
    if (hipMemsetAsync(workspace, 0, head, s) != hipSuccess) return pn2_check_launch();
    const dim3 grid((unsigned)((B / plan.NC) * plan.G));
    CoopSerial chain(s);
    // bucketed points (fps_bucket_kernel): one cloud per cluster, at least eight point slots per lane to skip
    float4 *rec = (float4 *)((char *)workspace + head + fps_align256((size_t)B * (size_t)N * sizeof(float)));
    // ... and 1024-thread workgroups, i.e. the shapes of a sampling that runs next to a training step (scheduling hints)
    // or a forced one: there the skipped updates are VALU time handed to the co-running kernels (same-box A/B of the
    // default step 15.45 -> 15.12 ms); the latency-optimal 512-thread clusters gain nothing (the round is the hand-off)
    const bool buck = plan.NC == 1 && plan.PPT >= 8 && plan.BS == 1024 && g_fps_bucketing.load(std::memory_order_relaxed);
    if (buck) hipLaunchKernelGGL(fps_bucket_kernel, dim3((unsigned)B), dim3(1024), 0, s, N, N, xyz, rec);
#define PN2_FPS_COOP_B(BSZ, PPT)                                                                    \
  {                                                                                                 \
    auto kfn = fps_coop_kernel<BSZ, PPT, 1, false, true>;                                           \
    if ((fits = coop_fits((const void *)kfn, BSZ, grid.x)))                                         \
      hipLaunchKernelGGL(kfn, grid, dim3(BSZ), 0, s, B, N, m, L, plan.G, xyz, idxs, slots, status,  \
                         (float *)nullptr, (const float4 *)rec);                                    \
  }
#define PN2_FPS_COOP(PPT, NC)                                                                       \
  {                                                                                                 \
    auto kfn = fps_coop_kernel<512, PPT, NC>;                                                       \
    if ((fits = coop_fits((const void *)kfn, 512, grid.x)))                                         \
      hipLaunchKernelGGL(kfn, grid, dim3(512), 0, s, B, N, m, L, plan.G, xyz, idxs, slots, status,  \
                         (float *)nullptr, (const float4 *)nullptr);                                \
  }
#define PN2_FPS_COOP_NC(NC)                                                                         \
  switch (plan.PPT) {                                                                               \
    case 1: PN2_FPS_COOP(1, NC); break;                                                             \
    case 2: PN2_FPS_COOP(2, NC); break;                                                             \
    case 3: PN2_FPS_COOP(3, NC); break;                                                             \
    case 4: PN2_FPS_COOP(4, NC); break;                                                             \
    case 6: if (NC * 6 <= 28) { PN2_FPS_COOP(6, (NC * 6 <= 28 ? NC : 1)); break; } return PN2_EINVAL;   \
    case 8: if (NC * 8 <= 28) { PN2_FPS_COOP(8, (NC * 8 <= 28 ? NC : 1)); break; } return PN2_EINVAL;   \
    case 10: if (NC * 10 <= 28) { PN2_FPS_COOP(10, (NC * 10 <= 28 ? NC : 1)); break; } return PN2_EINVAL; \
    case 12: if (NC * 12 <= 28) { PN2_FPS_COOP(12, (NC * 12 <= 28 ? NC : 1)); break; } return PN2_EINVAL; \
    case 14: if (NC * 14 <= 28) { PN2_FPS_COOP(14, (NC * 14 <= 28 ? NC : 1)); break; } return PN2_EINVAL; \
    default: return PN2_EINVAL;                                                                     \
  }
    if (plan.BS == 1024) {
#define PN2_FPS_COOP_W(PPT)                                                                           \
  {                                                                                                   \
    auto kfn = fps_coop_kernel<1024, PPT, 1>;                                                         \
    if ((fits = coop_fits((const void *)kfn, 1024, grid.x)))                                          \
      hipLaunchKernelGGL(kfn, grid, dim3(1024), 0, s, B, N, m, L, plan.G, xyz, idxs, slots, status,   \
                         (float *)nullptr, (const float4 *)nullptr);                                  \
  }
      if (buck) {
        switch (plan.PPT) {
          case 8: PN2_FPS_COOP_B(1024, 8); break;
          case 10: PN2_FPS_COOP_B(1024, 10); break;
          case 12: PN2_FPS_COOP_B(1024, 12); break;
          case 14: PN2_FPS_COOP_B(1024, 14); break;
          case 16: PN2_FPS_COOP_B(1024, 16); break;
          case 20: PN2_FPS_COOP_B(1024, 20); break;
          case 24: PN2_FPS_COOP_B(1024, 24); break;
          case 26: PN2_FPS_COOP_B(1024, 26); break;
          default: return PN2_EINVAL;
        }
      } else
      switch (plan.PPT) {
        case 8: PN2_FPS_COOP_W(8); break;
        case 10: PN2_FPS_COOP_W(10); break;
        case 12: PN2_FPS_COOP_W(12); break;
        case 14: PN2_FPS_COOP_W(14); break;
        case 16: PN2_FPS_COOP_W(16); break;
        case 20: PN2_FPS_COOP_W(20); break;
        case 24: PN2_FPS_COOP_W(24); break;
        case 26: PN2_FPS_COOP_W(26); break;
        default: return PN2_EINVAL;
      }
#undef PN2_FPS_COOP_W
    }
    else if (plan.NC == 4) { PN2_FPS_COOP_NC(4) }
    else if (plan.NC == 2) { PN2_FPS_COOP_NC(2) }
    else {
      switch (plan.PPT) {
        case 1: PN2_FPS_COOP(1, 1); break;
        case 2: PN2_FPS_COOP(2, 1); break;
        case 3: PN2_FPS_COOP(3, 1); break;
        case 4: PN2_FPS_COOP(4, 1); break;
        case 6: PN2_FPS_COOP(6, 1); break;
        case 8: PN2_FPS_COOP(8, 1); break;
        case 10: PN2_FPS_COOP(10, 1); break;
        case 12: PN2_FPS_COOP(12, 1); break;
        case 14: PN2_FPS_COOP(14, 1); break;
        case 16: PN2_FPS_COOP(16, 1); break;
        case 20: PN2_FPS_COOP(20, 1); break;
        case 24: PN2_FPS_COOP(24, 1); break;
        default: return PN2_EINVAL;
      }
    }
#undef PN2_FPS_COOP_NC
#undef PN2_FPS_COOP
#undef PN2_FPS_COOP_B
    if (fits) return pn2_check_launch();
  }

#define PN2_FPS_RES(BS, PPT)                                                                   \
  case PPT:                                                                                    \
    hipLaunchKernelGGL((fps_resident_kernel<BS, PPT>), dim3(B), dim3(BS), 0, s, N, m, L, xyz,  \
                       idxs, ordered_fail);                                                    \
    break;
  if (plan.mode == 2 || !fits) {
    hipLaunchKernelGGL((fps_stream_kernel<1024>), dim3(B), dim3(1024), 0, s, N, m, L, xyz, tail, idxs);
  } else if (plan.BS == 256) {
    switch (plan.PPT) {
      PN2_FPS_RES(256, 1) PN2_FPS_RES(256, 2) PN2_FPS_RES(256, 3) PN2_FPS_RES(256, 4)
      PN2_FPS_RES(256, 6) PN2_FPS_RES(256, 8) PN2_FPS_RES(256, 10) PN2_FPS_RES(256, 12)
      PN2_FPS_RES(256, 14) PN2_FPS_RES(256, 16)
      default: return PN2_EINVAL;
    }
  } else if (plan.BS == 512) {
    switch (plan.PPT) {
      PN2_FPS_RES(512, 1) PN2_FPS_RES(512, 2) PN2_FPS_RES(512, 3) PN2_FPS_RES(512, 4)
      PN2_FPS_RES(512, 6) PN2_FPS_RES(512, 8) PN2_FPS_RES(512, 10) PN2_FPS_RES(512, 12)
      PN2_FPS_RES(512, 14) PN2_FPS_RES(512, 16)
      default: return PN2_EINVAL;
    }
  } else {
    switch (plan.PPT) {
      PN2_FPS_RES(1024, 1) PN2_FPS_RES(1024, 2) PN2_FPS_RES(1024, 3) PN2_FPS_RES(1024, 4)
      PN2_FPS_RES(1024, 6) PN2_FPS_RES(1024, 8)
      PN2_FPS_RES(1024, 10) PN2_FPS_RES(1024, 12) PN2_FPS_RES(1024, 14) PN2_FPS_RES(1024, 16)
      PN2_FPS_RES(1024, 20) PN2_FPS_RES(1024, 24)
      default: return PN2_EINVAL;
    }
  }
#undef PN2_FPS_RES
  return pn2_check_launch();
}
}  // namespace

// Sampling of clouds the caller believes to be in farthest-point ORDER (the centres of the SA level above): identical
// results to pn2_furthest_point_sampling_ex for ANY input — the order is verified per cloud on the device (see
// fps_order_m_kernel), a cloud that passes gets 0 .. m-1 without a single sampling round, one that fails takes the rounds.
// Only plans with one workgroup per cloud and the points in registers take the shortcut (N <= 4096 or so: the lower SA
// levels); other shapes run the plain call.  Workspace: pn2_fps_ordered_workspace_bytes.
namespace { constexpr int kFpsOrderedMinSamples = 256; }

extern "C" size_t pn2_fps_ordered_workspace_bytes(int B, int N, int m) {
  if (B <= 0 || N <= 0 || m <= 0) return 0;
  return fps_align256(pn2_fps_workspace_bytes(B, N, m)) + fps_align256((size_t)B * m * sizeof(float)) + fps_align256((size_t)B * sizeof(int));
}

extern "C" int pn2_furthest_point_sampling_ordered(int B, int N, int m, const float *xyz, void *workspace,
                                                   size_t workspace_bytes, int *idxs, int flags, void *stream) {
  if (B < 0 || N < 0) return PN2_EINVAL;
  if (flags & ~(PN2_FPS_FEW_CUS | PN2_FPS_FEWEST_CUS)) return PN2_EINVAL;
  if (m <= 0 || B == 0) return PN2_OK;
  if (N <= 0) return PN2_EINVAL;
  if (!xyz || !idxs) return PN2_ENULL;
  const FpsPlan plan = fps_plan(B, N, m, flags != 0, (flags & PN2_FPS_FEWEST_CUS) != 0);
  const size_t base = fps_align256(pn2_fps_workspace_bytes(B, N, m));
  // (m <= N: otherwise the samples are not a prefix; 16 m bytes of centres + M per workgroup in LDS)
  // (fewer than 256 samples: the ~100 rounds cost less than the two verification launches do on a host-bound step —
  // scene-graph encoders, 128 of 512 points: 1 scan/step 140-147 scans/s plain, 125-139 verified)
  if (plan.mode != 0 || m > N || m < kFpsOrderedMinSamples || (size_t)m * 16 > 60 * 1024)
    return fps_launch(B, N, m, xyz, workspace, workspace_bytes < base ? workspace_bytes : base, idxs, flags, nullptr, stream);
  if (!workspace) return PN2_ENULL;
  if (workspace_bytes < pn2_fps_ordered_workspace_bytes(B, N, m)) return PN2_ENOSPC;
  if (((uintptr_t)workspace & 255) != 0) return PN2_EINVAL;
  float *Mk = (float *)((char *)workspace + base);
  int *fail = (int *)((char *)Mk + fps_align256((size_t)B * m * sizeof(float)));
  hipStream_t s = (hipStream_t)stream;
  const unsigned my = (unsigned)((m + 255) / 256), ny = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(fps_order_m_kernel, dim3((unsigned)B, my), dim3(256), (size_t)(3 * (m < 256 * (int)my ? m : 256 * my)) * sizeof(float), s, N, m,
                     xyz, Mk, fail);
  const int bs = ref_opt_n_threads(N);
  int L = 0;
  while ((1 << L) < bs) ++L;
  hipLaunchKernelGGL(fps_order_check_kernel, dim3((unsigned)B, ny), dim3(256), (size_t)4 * m * sizeof(float), s, N, m, L, xyz,
                     (const float *)Mk, fail);
  return fps_launch(B, N, m, xyz, workspace, base, idxs, flags, fail, stream);
}

// Byte offset of the int32 status word inside the workspace of a (B, N, m) call, or -1 when the plan has no
// inter-workgroup waits.  0 after a clean run, 1 when a bounded wait expired (the remaining indices are then 0).
extern "C" long long pn2_fps_status_offset_ex(int B, int N, int m, int flags) {
  if (B <= 0 || N <= 0 || m <= 1 || (flags & ~(PN2_FPS_FEW_CUS | PN2_FPS_FEWEST_CUS))) return -1;
  // the plan of the call that was made WITH these flags: the flag-less plan of the same shape can be a cluster mode where
  // the flagged one is resident / streaming (no status word is written, the bytes there are workspace: ADVICE r04)
  const FpsPlan p = fps_plan(B, N, m, flags != 0, (flags & PN2_FPS_FEWEST_CUS) != 0);
  return (p.mode == 1 || p.mode == 3 || p.mode == 5) ? (long long)((size_t)B * kCoopCloudBytes) : -1;
}
extern "C" long long pn2_fps_status_offset(int B, int N, int m) { return pn2_fps_status_offset_ex(B, N, m, 0); }

// Test hook: status word of the last cooperative launch that used `workspace`
// (0 = ok, 1 = a bounded spin expired).  Host-synchronous; not on the hot path.
extern "C" int pn2_fps_coop_status(int B, const void *workspace, void *stream) {
  int v = -1;
  if (!workspace) return -1;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
  if (hipMemcpy(&v, (const char *)workspace + (size_t)B * kCoopCloudBytes, sizeof(int),
                hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return v;
}

