// group_csr.hip — the feature-gradient scatter of QueryAndGroup as a gather: inverse neighbourhood index + CSR sum.
//
// Replaces group_points_grad_kernel (EXT/src/group_points_gpu.cu:44-75: one atomicAdd per gradient element into
// grad_points[b, :, idx[b, j, s]]) on the rows path.  With crowded balls (the headline SA2-SA4 levels: every slot a genuine
// hit, each point referenced ~16 times) the atomic form issues B*m*ns*C device-scope float atomics — they execute at the
// memory side, not in the XCD's L2 — and ran at 0.15 of the HBM rate.  The neighbourhood index is DATA (it depends on the
// clouds only), so its inverse is built once per batch next to the ball query (on the prefetch stream in the training
// loop):
//     refs  (B*m*ns)  row ids r = (b*m + j)*ns + s sorted by (b*N + idx[r], r)      [stable radix sort, rocPRIM]
//     ptr   (B*N + 1) refs[ptr[b*N + n] : ptr[b*N + n + 1]] = the rows that gathered point (b, n)
// and the backward becomes grad_feats[b, n, :] = sum over those rows of grad_out[r, col0 : col0 + C]: streaming 16-byte
// loads, one plain store per output row, no zero fill, and a summation order fixed by the sort — the result is
// bit-reproducible run to run (the atomic form is not).
#include "pn2_common.h"
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {
constexpr int kBlock = 256;

typedef float f4v __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) F4Dw { f4v v; };

__global__ __launch_bounds__(kBlock) void inv_keys_kernel(unsigned rows, unsigned per_cloud /* m*ns */, unsigned N,
                                                         const int *__restrict__ idx, unsigned *__restrict__ keys,
                                                         unsigned *__restrict__ vals) {
  for (unsigned r = blockIdx.x * kBlock + threadIdx.x; r < rows; r += gridDim.x * kBlock) {
    keys[r] = (r / per_cloud) * N + (unsigned)idx[r];
    vals[r] = r;
  }
}

// ptr[k] = number of sorted keys < k  (k = 0 .. npoints)
__global__ __launch_bounds__(kBlock) void inv_ptr_kernel(unsigned rows, unsigned npoints, const unsigned *__restrict__ keys,
                                                        int *__restrict__ ptr) {
  for (unsigned k = blockIdx.x * kBlock + threadIdx.x; k <= npoints; k += gridDim.x * kBlock) {
    unsigned lo = 0, hi = rows;
    while (lo < hi) {
      const unsigned mid = (lo + hi) >> 1;
      if (keys[mid] < k) lo = mid + 1; else hi = mid;
    }
    ptr[k] = (int)lo;
  }
}

// A wave per point; R sub-waves of 64 / R lanes walk the point's rows R at a time (C <= 4 * 64 / R), four 16-byte loads in
// flight per lane; the sub-wave sums are combined in a fixed order.
// BF: grad_out holds bf16 (the mixed-precision stack's activation gradients): 8-byte loads of four values.
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

template <int R, bool BF>
__global__ __launch_bounds__(kBlock) void group_rows_grad_csr_kernel(int C, int ldg, int col0, unsigned npoints,
                                                                    const void *__restrict__ gv,
                                                                    const int *__restrict__ ptr,
                                                                    const int *__restrict__ refs,
                                                                    float *__restrict__ out) {
  constexpr int LPR = 64 / R;
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  const float *g = (const float *)gv;
  const unsigned short *gb = (const unsigned short *)gv;
  const int lane = pn2_lane();
  const int sub = lane / LPR, l = lane % LPR;
  const bool fl = 4 * l < C;
  const unsigned nwaves = gridDim.x * (kBlock / 64);
  for (unsigned n = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)); n < npoints; n += nwaves) {
    const int p0 = ptr[n], p1 = ptr[n + 1];
    f4v acc = f4v{0.f, 0.f, 0.f, 0.f};
    for (int base = p0; base < p1; base += 64) {
      const int cnt = p1 - base < 64 ? p1 - base : 64;
      const int myref = lane < cnt ? refs[base + lane] : 0;
      for (int t = 0; t * R < cnt; t += 4) {
        f4v v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = (t + u) * R + sub;
          const int r = __shfl(myref, i & 63);
          v[u] = f4v{0.f, 0.f, 0.f, 0.f};
          if (i < cnt && fl) {
            if constexpr (BF) {
              const u2v w = *(const u2v *)(gb + (size_t)r * ldg + col0 + 4 * l);
              v[u] = f4v{bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y)};
            } else {
              v[u] = ((const F4Dw *)(g + (size_t)r * ldg + col0 + 4 * l))->v;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc.x = __fadd_rn(acc.x, v[u].x); acc.y = __fadd_rn(acc.y, v[u].y);
          acc.z = __fadd_rn(acc.z, v[u].z); acc.w = __fadd_rn(acc.w, v[u].w);
        }
      }
    }
#pragma unroll
    for (int d = 32; d >= LPR; d >>= 1) {
      acc.x = __fadd_rn(acc.x, __shfl_xor(acc.x, d)); acc.y = __fadd_rn(acc.y, __shfl_xor(acc.y, d));
      acc.z = __fadd_rn(acc.z, __shfl_xor(acc.z, d)); acc.w = __fadd_rn(acc.w, __shfl_xor(acc.w, d));
    }
    if (sub == 0 && fl) *(f4v *)(out + (size_t)n * C + 4 * l) = acc;
  }
}

// any C: lane c, c + 64, ... (element loads)
template <bool BF>
__global__ __launch_bounds__(kBlock) void group_rows_grad_csr_any_kernel(int C, int ldg, int col0, unsigned npoints,
                                                                        const void *__restrict__ gv,
                                                                        const int *__restrict__ ptr,
                                                                        const int *__restrict__ refs,
                                                                        float *__restrict__ out) {
  const int lane = pn2_lane();
  const unsigned nwaves = gridDim.x * (kBlock / 64);
  for (unsigned n = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)); n < npoints; n += nwaves) {
    const int p0 = ptr[n], p1 = ptr[n + 1];
    for (int c = lane; c < C; c += 64) {
      float acc = 0.f;
      for (int p = p0; p < p1; ++p) {
        const size_t e = (size_t)refs[p] * ldg + col0 + c;
        acc = __fadd_rn(acc, BF ? bf16_lo(((const unsigned short *)gv)[e]) : ((const float *)gv)[e]);
      }
      out[(size_t)n * C + c] = acc;
    }
  }
}

// ---- the inverse index in ONE launch (round 5) ----------------------------------------------------------------------
// The radix sort above is ~18 launches per call (47 rocPRIM passes per backbone step for three levels, 0.25 ms per call on
// the prefetch stream).  The keys are bounded point indices and the rows of a cloud are contiguous, so the index is a
// STABLE COUNTING SORT per cloud, done by one workgroup per cloud entirely in LDS:
//   rows of the cloud are cut into W contiguous chunks, one per wave; every wave counts the keys of its chunk into its OWN
//   histogram (cur[w][k]); a per-key prefix over the waves and a block scan over the keys turn the histograms into write
//   cursors (and `ptr`); every wave then walks its chunk in row order, 64 rows per step, and places them: lanes with equal keys
//   are found with a bit-sliced match (one ballot per key bit), their rank among equals is a masked popcount — rows of a
//   point therefore land in ascending row order, exactly the stable sort's result, with no atomics in the placement and a
//   summation order in the consumers that is fixed by construction (bit-reproducible like the sort's).
// LDS: W x Ns cursors, W = min(16, 36864 / Ns) waves, Ns = the points of one slice of the cloud (round 6; the radix sort
// remains as the route when the 144 KB LDS attribute is refused or PN2_INVERSE_INDEX_RADIX=1).
constexpr int kInvLdsInts = 36864;      // 144 KB of cursors
constexpr int kInvBatch = 8;            // 64-row steps whose index loads are in flight together

__global__ __launch_bounds__(1024) void inv_cloud_kernel(int N, int P, int W, int nbits, int last_cloud, int S, int Ns,
                                                         const int *__restrict__ idx, int *__restrict__ ptr,
                                                         int *__restrict__ refs) {
  // round 6: a cloud's POINTS are cut into S slices of Ns, one workgroup per (cloud, slice).  Every workgroup walks all rows
  // of its cloud but counts / places only the keys of its slice; where its slice starts in `refs` is the number of rows with
  // a smaller key, which it counts on the way — no dependency between workgroups.  S = 1 is the round-5 kernel.  Slices put
  // B S workgroups on the chip instead of B and lift the 36864-point limit (LDS holds W x Ns cursors).
  extern __shared__ int cur[];            // [W][Ns]
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, T = blockDim.x;
  // workgroups are dealt to the 8 XCDs round-robin: the slices of a cloud are given ids that share an XCD, so the cloud's rows
  // are read into ONE L2 (not eight) and found there by the other slices
  const int nwg = gridDim.x, per = nwg >> 3;
  const int v = (int)blockIdx.x < 8 * per ? ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int b = v / S, sl = v - b * S;
  const int k0 = sl * Ns, nk = N - k0 < Ns ? N - k0 : Ns;
  const int *I = idx + (size_t)b * P;
  const int chunk = (((P + W - 1) / W) + 63) & ~63;          // rows per wave, whole steps of 64
  const int r0 = wv * chunk, r1 = r0 + chunk < P ? r0 + chunk : P;
  for (int i = tid; i < W * Ns; i += T) cur[i] = 0;
  if (tid == 0) carry = 0;
  __syncthreads();
  // (1) histogram of the wave's chunk (LDS integer atomics: order-free); rows below the slice are only counted
  //     (kInvBatch steps of 64 rows per trip, their loads issued together: the walk is latency-bound — 16 waves on a CU)
  int below = 0;
  for (int rb = r0; rb < r1; rb += 64 * kInvBatch) {
    unsigned kq[kInvBatch];
#pragma unroll
    for (int j = 0; j < kInvBatch; ++j) {
      const int r = rb + 64 * j + lane;
      kq[j] = r < r1 ? (unsigned)I[r] : 0xffffffffu;
    }
#pragma unroll
    for (int j = 0; j < kInvBatch; ++j) {
      if (rb + 64 * j + lane >= r1) continue;
      const unsigned k = kq[j] < (unsigned)N ? kq[j] : (unsigned)N - 1u;
      const unsigned kk = k - (unsigned)k0;
      if (kk < (unsigned)nk) atomicAdd(&cur[wv * Ns + (int)kk], 1);
      below += (int)k < k0;
    }
  }
  if (S > 1) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) below += __shfl_xor(below, d);
    if (lane == 0 && below) atomicAdd(&carry, below);
  }
  __syncthreads();
  // (2) cursors: keys in tiles of T (thread = key: conflict-free), per key an exclusive prefix over the waves, over the keys a
  //     block scan carried from tile to tile
  int *P_ = ptr + (size_t)b * N + k0;
  for (int kb = 0; kb < nk; kb += T) {
    const int k = kb + tid;
    int tot = 0;
    if (k < nk) {
      for (int w = 0; w < W; ++w) {
        const int c = cur[w * Ns + k];
        cur[w * Ns + k] = tot;
        tot += c;
      }
    }
    int inc = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(inc, d);
      if (lane >= d) inc += v;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = carry + inc - tot;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    if (k < nk) {
      P_[k] = b * P + base;
      for (int w = 0; w < W; ++w) cur[w * Ns + k] += base;
    }
    __syncthreads();
    if (tid == T - 1) carry = base + tot;                    // (the last thread's exclusive base + its own count = tile total)
    __syncthreads();
  }
  if (b == last_cloud && sl == S - 1 && tid == 0) ptr[(size_t)(b + 1) * N] = (b + 1) * P;
  // (3) placement in row order
  int *R = refs + (size_t)b * P;
  // One step: up to 64 (key, row) pairs of the slice, in row order: lanes with equal keys are found with a bit-sliced match, ranked
  // by a masked popcount and placed behind the key's cursor.
  auto place = [&](unsigned k, int r, bool ok) {
    u64 same = __ballot(ok);
    for (int bit = 0; bit < nbits; ++bit) {
      const bool set = (k >> bit) & 1u;
      const u64 mb = __ballot(set);
      same &= set ? mb : ~mb;
    }
    if (ok) {
      const int rank = pn2_prefix_popc(same);
      const int at = cur[wv * Ns + (int)k];
      if (rank == 0) cur[wv * Ns + (int)k] = at + __popcll(same);
      R[at + rank] = b * P + r;
    }
  };
  // With slices only 1 / S of the rows belong to this workgroup: they are first COMPACTED (order kept) into a 128-entry queue of
  // the wave in LDS — a ballot and a popcount per step — and a placement step runs per 64 queued rows (the match costs ~100
  // instructions per step, the walk is issue-bound: without the queue S workgroups would each pay it for every row).
  int2 *queue = (int2 *)(cur + ((W * Ns + 1) & ~1)) + wv * 128;
  int qn = 0;
  for (int rb4 = r0; rb4 < r1; rb4 += 64 * kInvBatch) {                 // wave-uniform trip count; kInvBatch steps' keys loaded together
    unsigned kq[kInvBatch];
#pragma unroll
    for (int j = 0; j < kInvBatch; ++j) {
      const int r = rb4 + 64 * j + lane;
      kq[j] = r < r1 ? (unsigned)I[r] : 0u;
    }
#pragma unroll
    for (int j = 0; j < kInvBatch; ++j) {
      const int r = rb4 + 64 * j + lane;
      const unsigned k = (kq[j] < (unsigned)N ? kq[j] : (unsigned)N - 1u) - (unsigned)k0;
      const bool ok = r < r1 && k < (unsigned)nk;
      if (S == 1) {
        place(k, r, ok);
        continue;
      }
      const u64 in = __ballot(ok);
      if (in == 0) continue;                                 // (wave-uniform)
      if (ok) queue[qn + pn2_prefix_popc(in)] = make_int2((int)k, r);
      qn += __popcll(in);
      __builtin_amdgcn_wave_barrier();
      if (qn >= 64) {
        const int2 e = queue[lane];
        const int2 up = queue[64 + lane];
        __builtin_amdgcn_wave_barrier();
        qn -= 64;
        if (lane < qn) queue[lane] = up;
        __builtin_amdgcn_wave_barrier();
        place((unsigned)e.x, e.y, true);
      }
    }
  }
  if (qn > 0) {
    const int2 e = lane < qn ? queue[lane] : make_int2(0, 0);
    place((unsigned)e.x, e.y, lane < qn);
  }
}

// Slices of a cloud's points: B S workgroups of about one per CU when the cloud has rows enough to share out (a workgroup
// walks every row of its cloud: slices of fewer than 256 points or clouds of fewer than 8192 rows stay whole), W x Ns cursors
// + the waves' queues within the 144 KB of LDS.
constexpr int kInvQueueInts = 16 * 128 * 2 + 2;   // (+ the padding that aligns the queues to 8 bytes)
inline void inv_slices(int B, int N, int P, int *S, int *Ns) {
  int s = N > 8192 ? (N + 4095) / 4096 : 1;                    // (cursor init + scan are O(points of the slice) per workgroup)
  if (P >= 8192) {
    // about a workgroup per CU ...
    const int full = (N + 2046) / 2047;                        // ... and 16 waves in each (16 x 2047 cursors fill the LDS)
    const int per_cu = 256 / B;
    s = per_cu > full ? per_cu : full;
  }
  const char *e = getenv("PN2_INVERSE_INDEX_SLICES");
  if (e && e[0] >= '1' && e[0] <= '9') s = atoi(e);
  if (s > (N + 255) / 256) s = (N + 255) / 256;
  while (s > 1 && (long)B * s > 4096) --s;
  if (s < 1) s = 1;
  int ns = (N + s - 1) / s;
  while (ns > (s > 1 ? kInvLdsInts - kInvQueueInts : kInvLdsInts)) {
    ++s;
    ns = (N + s - 1) / s;
  }
  *S = (N + ns - 1) / ns;
  *Ns = ns;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

inline int key_bits(size_t npoints) {
  int bits = 1;
  while (bits < 32 && ((size_t)1 << bits) < npoints) ++bits;
  return bits;
}

inline hipError_t sort_temp_bytes(size_t rows, int bits, size_t *bytes) {
  *bytes = 0;
  return rocprim::radix_sort_pairs(nullptr, *bytes, (const unsigned *)nullptr, (unsigned *)nullptr,
                                   (const unsigned *)nullptr, (unsigned *)nullptr, rows, 0, bits, (hipStream_t)0);
}

// Which algorithm a call takes is decided in ONE place, before any size check, for the workspace query and the entry alike
// (the query used to promise 256 bytes for N <= kInvLdsInts while the entry could still fall through to the
// radix sort when the 144 KB dynamic-LDS attribute was refused, and sort into a 256-byte workspace).  The attribute is asked
// for once per process; PN2_INVERSE_INDEX_RADIX=1 (read on every call: tests flip it) forces the sort route.
bool inv_use_lds(int B, int N, double P) {
  // every slice's workgroup walks all P rows of its cloud: B N P / (64 x 32768 cursors x 256 CUs) steps of ~0.55 us per CU
  // (measured, tools/diag/inv_slices_time.py: 8 x 50000 x 131072 -> 55 us, 32 x 50000 x 131072 -> 217 us) against the
  // sort's ~160 us + 2.5 us per million rows: beyond 1.2e11 the sort is the faster route
  if ((double)B * (double)N * P > 1.2e11) return false;
  static const bool lds_ok = hipFuncSetAttribute((const void *)inv_cloud_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 kInvLdsInts * 4) == hipSuccess;
  if (!lds_ok) return false;
  const char *force = getenv("PN2_INVERSE_INDEX_RADIX");
  return !(force && force[0] == '1');
}
}  // namespace

extern "C" size_t pn2_group_inverse_index_workspace_bytes(int B, int N, int m, int ns) {
  if (B <= 0 || N <= 0 || m <= 0 || ns <= 0) return 0;
  if (inv_use_lds(B, N, (double)m * ns)) return 256;                              // one-launch counting sort in LDS: no scratch (a token size)
  const size_t rows = (size_t)B * m * ns;
  size_t temp = 0;
  if (sort_temp_bytes(rows, key_bits((size_t)B * N), &temp) != hipSuccess) return 0;
  return 3 * align256(rows * 4) + align256(temp);
}

extern "C" int pn2_group_inverse_index(int B, int N, int m, int ns, const int *idx, int *ptr, int *refs, void *workspace,
                                       size_t workspace_bytes, void *stream) {
  if (B < 0 || N < 0 || m < 0 || ns < 0) return PN2_EINVAL;
  const size_t rows = (size_t)B * m * ns, npoints = (size_t)B * N;
  if (rows >= 0x7fffffffull || npoints >= 0x7fffffffull) return PN2_EINVAL;
  if (npoints == 0) return PN2_OK;
  if (!ptr) return PN2_ENULL;
  if (rows == 0) return hipMemsetAsync(ptr, 0, (npoints + 1) * 4, (hipStream_t)stream) == hipSuccess ? PN2_OK : PN2_ELAUNCH;
  if (!idx || !refs || !workspace) return PN2_ENULL;
  const int bits = key_bits(npoints);
  size_t temp = 0;
  const size_t seg = align256(rows * 4);
  if ((((size_t)workspace) & 255) != 0) return PN2_EINVAL;
  const bool lds = inv_use_lds(B, N, (double)m * ns);
  if (lds) {
    if (workspace_bytes < 256) return PN2_ENOSPC;
  } else {
    if (sort_temp_bytes(rows, bits, &temp) != hipSuccess) return PN2_ELAUNCH;
    if (workspace_bytes < 3 * seg + align256(temp)) return PN2_ENOSPC;
  }
  if (lds) {
    // one launch: a stable counting sort per cloud in LDS (inv_cloud_kernel); the workspace is not touched
    const int P = m * ns;
    int S, Ns;
    inv_slices(B, N, P, &S, &Ns);
    const int room = S > 1 ? kInvLdsInts - kInvQueueInts : kInvLdsInts;
    int W = room / Ns;
    W = W > 16 ? 16 : W;
    const int steps = (P + 63) / 64;                          // no more waves than 64-row steps
    W = W > steps ? steps : W;
    hipLaunchKernelGGL(inv_cloud_kernel, dim3((unsigned)(B * S)), dim3(64u * W),
                       ((size_t)W * Ns + (S > 1 ? kInvQueueInts : 0)) * sizeof(int), (hipStream_t)stream, N, P, W,
                       key_bits((size_t)Ns), B - 1, S, Ns, idx, ptr, refs);
    return pn2_check_launch();
  }
  unsigned *keys_in = (unsigned *)workspace;
  unsigned *keys_out = (unsigned *)((char *)workspace + seg);
  unsigned *vals_in = (unsigned *)((char *)workspace + 2 * seg);
  void *tmp = (char *)workspace + 3 * seg;
  unsigned grid = (unsigned)((rows + kBlock - 1) / kBlock);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(inv_keys_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, (unsigned)rows, (unsigned)(m * ns),
                     (unsigned)N, idx, keys_in, vals_in);
  if (rocprim::radix_sort_pairs(tmp, temp, keys_in, keys_out, vals_in, (unsigned *)refs, rows, 0, bits,
                                (hipStream_t)stream) != hipSuccess)
    return PN2_ELAUNCH;
  grid = (unsigned)((npoints + 1 + kBlock - 1) / kBlock);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(inv_ptr_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, (unsigned)rows, (unsigned)npoints,
                     keys_out, ptr);
  return pn2_check_launch();
}

namespace {
template <bool BF>
int launch_rows_grad_csr(int B, int N, int C, int ldg, int col0, int64_t rows, const void *grad_out, const int *ptr,
                         const int *refs, float *grad_feats, void *stream) {
  if (B < 0 || N < 0 || C < 0 || col0 < 0 || ldg < col0 + C || rows < 0 || rows >= 0x7fffffffll) return PN2_EINVAL;
  const size_t npoints = (size_t)B * N;
  if (npoints == 0 || C == 0) return PN2_OK;
  if (npoints >= 0x7fffffffull) return PN2_EINVAL;
  if (!ptr || !grad_feats || (rows > 0 && (!grad_out || !refs))) return PN2_ENULL;
  const unsigned waves_wanted = 256u * 32u;
  unsigned grid = (unsigned)((npoints < waves_wanted ? npoints : waves_wanted) + 3) / 4;
  if (grid == 0) grid = 1;
  // 4 values per lane: fp32 rows at dword alignment (16-byte loads), bf16 rows at 8-byte alignment
  const bool v4 = (C & 3) == 0 && C <= 256 && (((size_t)grad_feats) & 15) == 0 &&
                  (BF ? ((ldg & 3) == 0 && (col0 & 3) == 0 && (((size_t)grad_out) & 7) == 0) : (((size_t)grad_out) & 3) == 0);
#define PN2_CSR(R) hipLaunchKernelGGL((group_rows_grad_csr_kernel<R, BF>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, C, \
                                      ldg, col0, (unsigned)npoints, grad_out, ptr, refs, grad_feats)
  if (v4 && C <= 64) PN2_CSR(4);
  else if (v4 && C <= 128) PN2_CSR(2);
  else if (v4) PN2_CSR(1);
  else
    hipLaunchKernelGGL(group_rows_grad_csr_any_kernel<BF>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, C, ldg, col0,
                       (unsigned)npoints, grad_out, ptr, refs, grad_feats);
#undef PN2_CSR
  return pn2_check_launch();
}
}  // namespace

extern "C" int pn2_group_rows_grad_csr(int B, int N, int C, int ldg, int col0, int64_t rows, const float *grad_out,
                                       const int *ptr, const int *refs, float *grad_feats, void *stream) {
  return launch_rows_grad_csr<false>(B, N, C, ldg, col0, rows, grad_out, ptr, refs, grad_feats, stream);
}

extern "C" int pn2_group_rows_grad_csr_bf16(int B, int N, int C, int ldg, int col0, int64_t rows, const void *grad_out,
                                            const int *ptr, const int *refs, float *grad_feats, void *stream) {
  return launch_rows_grad_csr<true>(B, N, C, ldg, col0, rows, grad_out, ptr, refs, grad_feats, stream);
}

namespace {
// ---- the LITERAL op's gradient as a gather (round 6) ------------------------------------------------------------------
// group_points_grad_kernel of the reference (EXT/src/group_points_gpu.cu:43-64) adds every gradient element to its point with
// an fp32 atomic: B C m ns device-scope atomics (12.6 M at the C = 3 micro shape: 0.62 ms = 0.0175 of 8 TB/s), in an order that
// changes from run to run.  Through the inverse index (ptr / refs of pn2_group_inverse_index over idx (B, m ns)) every point
// sums ITS rows in ascending row order: no atomics, every output element written once, bit-reproducible.
// Channel-major layouts of the reference: grad_out (B, C, S = m ns), grad_points (B, C, N).  Lane = point (coalesced stores).
// `div` = slots per column of grad_out (1: group_points / gather_points; 3: three_interpolate, whose slot (j, t) reads column
// j); `weight` (per slot, or NULL) multiplies the gathered value (three_interpolate_grad_kernel, EXT/src/interpolate_gpu.cu:116-143).
__global__ __launch_bounds__(256) void group_points_grad_csr_kernel(int C, int N, unsigned S, unsigned div, unsigned npoints,
                                                                   const float *__restrict__ grad_out,
                                                                   const float *__restrict__ weight,
                                                                   const int *__restrict__ ptr, const int *__restrict__ refs,
                                                                   float *__restrict__ grad_points) {
  const unsigned cols = S / div;
  for (unsigned g = blockIdx.x * 256 + threadIdx.x; g < npoints; g += gridDim.x * 256) {
    const unsigned b = g / (unsigned)N, n = g - b * (unsigned)N;
    const int p0 = ptr[g], p1 = ptr[g + 1];
    const float *G = grad_out + (size_t)b * C * cols;
    float *O = grad_points + (size_t)b * C * N + n;
    // four channels per walk of the point's references: a reference (and its weight) is read once for them, and their four
    // gathers are independent loads (C = 3, 2.6 references per point: the one-channel walk re-read the list per channel)
    for (int c = 0; c < C; c += 4) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int p = p0; p < p1; ++p) {
        const unsigned slot = (unsigned)refs[p];
        const size_t col = (slot - b * S) / div;
        const float w = weight ? weight[slot] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = c + j < C ? G[(size_t)(c + j) * cols + col] : 0.f;
          acc[j] = weight ? __fmaf_rn(v, w, acc[j]) : __fadd_rn(acc[j], v);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < C) O[(size_t)(c + j) * N] = acc[j];
    }
  }
}

// three_interpolate's literal gradient with the gradient rows of a channel group staged in LDS: a workgroup takes (cloud, CH
// channels), reads their n columns once, coalesced (CH n floats <= 64 KB), and every thread then gathers for (known point,
// channel) pairs — lanes along the points: the random reads hit LDS, the stores are coalesced.  The one-thread-per-point form
// above walks C channels serially (B m threads in all: 1.56 ms at 32 x 256 x 1024 -> 512, slower than the 25 M atomics it replaces).
__global__ __launch_bounds__(256) void interp_grad_csr_lds_kernel(int C, int n, int m, int CH, const float *__restrict__ grad_out,
                                                                  const float *__restrict__ weight, const int *__restrict__ ptr,
                                                                  const int *__restrict__ refs, float *__restrict__ grad_points) {
  extern __shared__ float gtile[];                       // [CH][n]
  const int groups = (C + CH - 1) / CH;
  const int b = blockIdx.x / groups, c0 = (blockIdx.x - b * groups) * CH;
  const int ch = C - c0 < CH ? C - c0 : CH;
  const float *G = grad_out + ((size_t)b * C + c0) * n;
  for (int i = threadIdx.x; i < ch * n; i += 256) gtile[i] = G[i];
  __syncthreads();
  const unsigned S = 3u * (unsigned)n;
  for (int o = threadIdx.x; o < ch * m; o += 256) {
    const int c = o / m, p = o - c * m;
    const int p0 = ptr[(size_t)b * m + p], p1 = ptr[(size_t)b * m + p + 1];
    float acc = 0.f;
    for (int q = p0; q < p1; ++q) {
      const unsigned slot = (unsigned)refs[q];
      acc = __fmaf_rn(gtile[c * n + (int)((slot - (unsigned)b * S) / 3u)], weight[slot], acc);
    }
    grad_points[((size_t)b * C + c0 + c) * m + p] = acc;
  }
}

int points_grad_csr(int B, int C, int N, size_t S, unsigned div, const float *grad_out, const float *weight, const int *ptr,
                    const int *refs, float *grad_points, void *stream) {
  const size_t total = (size_t)B * N;
  if (total == 0 || C == 0) return PN2_OK;
  if (total >= 0x7fffffffull || (size_t)B * S >= 0x7fffffffull) return PN2_EINVAL;
  if (!ptr || !grad_points || (S > 0 && (!grad_out || !refs))) return PN2_ENULL;
  unsigned grid = (unsigned)((total + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(group_points_grad_csr_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, C, N, (unsigned)S, div,
                     (unsigned)total, grad_out, weight, ptr, refs, grad_points);
  return pn2_check_launch();
}
}  // namespace

extern "C" int pn2_group_points_grad_csr(int B, int C, int N, int npoints, int nsample, const float *grad_out, const int *ptr,
                                         const int *refs, float *grad_points, void *stream) {
  if (B < 0 || C < 0 || N < 0 || npoints < 0 || nsample < 0) return PN2_EINVAL;
  return points_grad_csr(B, C, N, (size_t)npoints * nsample, 1u, grad_out, nullptr, ptr, refs, grad_points, stream);
}

// three_interpolate's gradient (EXT/src/interpolate_gpu.cu:116-143: three atomicAdds per gradient element) as the same gather:
// (ptr, refs) = pn2_group_inverse_index(B, N = m, npoints = n, nsample = 3, idx); grad_out (B, C, n), weight (B, n, 3),
// grad_points (B, C, m) — every element written, slots summed in ascending slot order.
extern "C" int pn2_three_interpolate_grad_csr(int B, int C, int n, int m, const float *grad_out, const float *weight,
                                              const int *ptr, const int *refs, float *grad_points, void *stream) {
  if (B < 0 || C < 0 || n < 0 || m < 0) return PN2_EINVAL;
  if (B == 0 || C == 0 || m == 0) return PN2_OK;
  if (n > 0 && !weight) return PN2_ENULL;
  int CH = 16;
  while (CH > 1 && (size_t)CH * n * 4 > 64 * 1024) CH >>= 1;
  if (B > 0 && C > 0 && m > 0 && n > 0 && (size_t)CH * n * 4 <= 64 * 1024 && (size_t)B * n * 3 < 0x7fffffffull) {
    if (!ptr || !refs || !grad_out || !grad_points) return PN2_ENULL;
    if (CH > C) CH = C;
    const int groups = (C + CH - 1) / CH;
    hipLaunchKernelGGL(interp_grad_csr_lds_kernel, dim3((unsigned)(B * groups)), dim3(256), (size_t)CH * n * 4, (hipStream_t)stream, C,
                       n, m, CH, grad_out, weight, ptr, refs, grad_points);
    return pn2_check_launch();
  }
  return points_grad_csr(B, C, m, (size_t)n * 3, 3u, grad_out, weight, ptr, refs, grad_points, stream);
}
