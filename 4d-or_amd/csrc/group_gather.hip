// group_gather.hip — gather_points / group_points (+grads) in the reference's
// channel-major layout, and the point-major ("rows") grouping extras used by
// the fast path.
//
// Reference kernels replaced: gather_points_kernel / gather_points_grad_kernel
// (EXT/src/sampling_gpu.cu:8-57), group_points_kernel / group_points_grad_kernel
// (EXT/src/group_points_gpu.cu:8-75).
//
// Mapping notes (HBM-bound byte movers):
//  * the reference runs ONE block per batch element; here the grid covers every
//    output element so all 256 CUs stream;
//  * every thread owns one (centre, sample) slot: the index is read once
//    (coalesced) and reused for all C channels; writes are coalesced along the
//    (npoints*nsample) axis;
//  * grads use the hardware fp32 global atomic add (-munsafe-fp-atomics).
#include "pn2_common.h"

namespace {

constexpr int kBlock = 256;

// 16 bytes at a dword-aligned address (rows of 3 + C floats): hipcc emits global_load/store_dwordx4 for it
typedef float f4v __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) F4Dw { f4v v; };

inline unsigned grid1d(size_t work, int block = kBlock) {
  size_t g = (work + block - 1) / block;
  return (unsigned)(g ? g : 1);
}

// XCD-aware workgroup order for the row-slab kernels: consecutive workgroup ids are dispatched round-robin over the 8 XCDs
// (each with its own 4 MB L2), so with slabs handed out in id order every cloud's rows are spread over all eight L2s and
// every L2 sees every cloud's feature table (33 MB at SA2: each row is fetched ~16 times from the memory side).  Here
// XCD x takes the x-th eighth of the slabs: the workgroups resident on one XCD work on 2-3 neighbouring clouds, whose
// tables (1 MB each) stay in that L2.  The launchers pad the grid to a multiple of 8.
__device__ __forceinline__ unsigned xcd_slab_wave() {
  const unsigned per = gridDim.x >> 3;
  const unsigned wg = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
  return __builtin_amdgcn_readfirstlane(wg * (kBlock / 64) + (threadIdx.x >> 6));
}

// ---------------------------------------------------------------- gather ----
__global__ __launch_bounds__(kBlock) void gather_points_kernel(int C, int N, int m,
                                                              const float *__restrict__ points,
                                                              const int *__restrict__ idx,
                                                              float *__restrict__ out, size_t total) {
  // flat over (b, c, j)
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (size_t)gridDim.x * kBlock) {
    const size_t bc = e / m;
    const int j = (int)(e - bc * m);
    const size_t b = bc / C;
    const int a = idx[b * m + j];
    out[e] = points[bc * N + a];
  }
}

__global__ __launch_bounds__(kBlock) void gather_points_grad_kernel(int C, int N, int m,
                                                                   const float *__restrict__ grad_out,
                                                                   const int *__restrict__ idx,
                                                                   float *__restrict__ grad_points,
                                                                   size_t total) {
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (size_t)gridDim.x * kBlock) {
    const size_t bc = e / m;
    const int j = (int)(e - bc * m);
    const size_t b = bc / C;
    const int a = idx[b * m + j];
    atomicAdd(grad_points + bc * N + a, grad_out[e]);
  }
}

// ----------------------------------------------------------------- group ----
__global__ __launch_bounds__(kBlock) void group_points_kernel(int C, int N, size_t S /* npoints*nsample */,
                                                             const float *__restrict__ points,
                                                             const int *__restrict__ idx,
                                                             float *__restrict__ out, int B) {
  // grid.y would overflow for big B, so (b) is folded into the flat id
  const size_t total = (size_t)B * S;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (size_t)gridDim.x * kBlock) {
    const size_t b = e / S;
    const size_t s = e - b * S;
    const int ii = idx[e];
    const float *src = points + b * C * N + ii;
    float *dst = out + b * C * S + s;
    for (int c = 0; c < C; ++c) dst[(size_t)c * S] = src[(size_t)c * N];
  }
}

__global__ __launch_bounds__(kBlock) void group_points_grad_kernel(int C, int N, size_t S,
                                                                  const float *__restrict__ grad_out,
                                                                  const int *__restrict__ idx,
                                                                  float *__restrict__ grad_points, int B) {
  const size_t total = (size_t)B * S;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (size_t)gridDim.x * kBlock) {
    const size_t b = e / S;
    const size_t s = e - b * S;
    const int ii = idx[e];
    float *dst = grad_points + b * C * N + ii;
    const float *g = grad_out + b * C * S + s;
    for (int c = 0; c < C; ++c) atomicAdd(dst + (size_t)c * N, g[(size_t)c * S]);
  }
}

// ------------------------------------------------------- point-major rows ----
// out[row, 0:Cx]   = (xyz[b, idx[row]] - new_xyz[b, j]) (/ radius)
// out[row, Cx:W]   = feats[b, idx[row], :]          row = (b*m + j)*ns + s
//
// Wide rows (C >= 32): ONE WAVE PER ROW GROUP.  A wave walks a contiguous range of rows eight at a time:
// lanes 0..7 fetch the eight neighbour indices with one load, `readlane` turns each into a wave-uniform
// scalar, and the row's source / destination addresses are scalar bases + lane offsets, so the column loop
// is pure coalesced load/store with no per-element integer arithmetic.  (The first version decomposed the
// flat element index with three 64-bit divisions per ELEMENT — ~150 VALU instructions per float moved.)
// The (b, j, s) decomposition of a row is kept in scalar counters.
__global__ __launch_bounds__(kBlock) void group_concat_rows_wide_kernel(
    int N, int m, int ns, int C, int Cx, int normalize, float radius,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ feats, const int *__restrict__ idx, float *__restrict__ out,
    unsigned rows, unsigned rows_per_wave) {
  const int lane = pn2_lane();
  const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
  const int W = Cx + C;
  unsigned r0 = wave * rows_per_wave;
  if (r0 >= rows) return;
  unsigned r1 = r0 + rows_per_wave;
  if (r1 > rows) r1 = rows;
  unsigned bj = r0 / (unsigned)ns;               // b*m + j of the current row (scalar)
  unsigned s = r0 - bj * (unsigned)ns;
  unsigned b = bj / (unsigned)m;
  unsigned j = bj - b * (unsigned)m;
  for (unsigned base = r0; base < r1; base += 8) {
    const unsigned nrow = (r1 - base) < 8u ? (r1 - base) : 8u;
    const int myi = lane < (int)nrow ? idx[base + lane] : 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if ((unsigned)q >= nrow) break;            // wave-uniform
      const int ii = __builtin_amdgcn_readlane(myi, q);
      const size_t src = (size_t)b * N + (size_t)ii;
      float *o = out + (size_t)(base + q) * W;
      if (lane < Cx) {
        float v = xyz[src * 3 + lane] - new_xyz[(size_t)bj * 3 + lane];
        if (normalize) v = __fdiv_rn(v, radius);
        o[lane] = v;
      }
      const float *f = feats + src * C;
      for (int c = lane; c < C; c += 64) o[Cx + c] = f[c];
      if (++s == (unsigned)ns) {
        s = 0; ++bj;
        if (++j == (unsigned)m) { j = 0; ++b; }
      }
    }
  }
}

// Wide rows whose feature width is a multiple of 4: the same walk, EIGHT ROWS IN FLIGHT.  The first version above moves
// one row at a time (index -> address -> dword loads -> dword stores), so a wave has one row's worth of loads outstanding
// and the kernel ran at 0.2-0.3 of the HBM rate with 1.7x the algorithmic traffic (dword stores into 524-byte rows).  Here
// a batch of eight rows issues ALL its 16-byte feature loads (R rows per wave instruction: 64 / R lanes per row, R = 4, 2, 1
// for C <= 64, 128, 256) and the six xyz loads of each row's first lane before the first store; the stores are 16-byte
// (dword-aligned: the row pitch 3 + C is odd) plus three dwords per row for the relative xyz.
template <int R>
__global__ __launch_bounds__(kBlock) void group_concat_rows_wide4_kernel(
    int N, int m, int ns, int C, int normalize, float radius,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ feats, const int *__restrict__ idx, float *__restrict__ out,
    unsigned rows, unsigned rows_per_wave) {
  constexpr int LPR = 64 / R;                    // lanes per row
  constexpr int P = 8 / R;                       // wave instructions per batch of 8 rows
  constexpr int Cx = 3;
  const int lane = pn2_lane();
  const unsigned wave = xcd_slab_wave();
  const int W = Cx + C;
  const int sub = lane / LPR, l = lane % LPR;
  const bool fl = 4 * l < C;                     // this lane carries features
  unsigned r0 = wave * rows_per_wave;
  if (r0 >= rows) return;
  unsigned r1 = r0 + rows_per_wave;
  if (r1 > rows) r1 = rows;
  unsigned bj = r0 / (unsigned)ns;               // b*m + j of row `base` (scalar)
  unsigned s = r0 - bj * (unsigned)ns;
  unsigned b = bj / (unsigned)m;
  unsigned j = bj - b * (unsigned)m;
  for (unsigned base = r0; base < r1; base += 8) {
    const unsigned nrow = (r1 - base) < 8u ? (r1 - base) : 8u;
    const int myi = lane < (int)nrow ? idx[base + lane] : 0;
    // A batch that lies inside one neighbourhood and repeats one index (ball-query padding = the first hit again,
    // EXT/src/ball_query_gpu.cu:34-38: 60 of 64 slots on the scene-graph encoders) is eight copies of ONE output row:
    // it is gathered once and stored eight times.
    const bool same = nrow == 8u && s + 8u <= (unsigned)ns &&
                      __all(lane >= 8 || myi == __builtin_amdgcn_readlane(myi, 0));
    if (same) {                                   // wave-uniform
      const size_t src = (size_t)b * N + (size_t)__builtin_amdgcn_readlane(myi, 0);
      f4v v0 = f4v{0.f, 0.f, 0.f, 0.f};
      float rel0[3] = {0.f, 0.f, 0.f};
      if (fl) v0 = *(const f4v *)(feats + src * C + 4 * l);
      if (l == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          rel0[d] = xyz[src * 3 + d] - new_xyz[(size_t)bj * 3 + d];
          if (normalize) rel0[d] = __fdiv_rn(rel0[d], radius);
        }
      }
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float *o = out + (size_t)(base + p * R + sub) * W;
        if (fl) ((F4Dw *)(o + Cx + 4 * l))->v = v0;
        if (l == 0) {
#pragma unroll
          for (int d = 0; d < 3; ++d) o[d] = rel0[d];
        }
      }
      s += 8u;
      if (s >= (unsigned)ns) {
        s -= (unsigned)ns; ++bj;
        if (++j == (unsigned)m) { j = 0; ++b; }
      }
      continue;
    }
    float4 v[P];
    float rel[P][3];
    unsigned q_of[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const unsigned q = p * R + sub;            // row of the batch this lane works on (ns >= 8: at most one wrap)
      q_of[p] = q;
      const unsigned wrap = (s + q) >= (unsigned)ns ? 1u : 0u;
      const unsigned bjq = bj + wrap;
      const unsigned bq = b + ((wrap && j + 1 == (unsigned)m) ? 1u : 0u);
      const int ii = __shfl(myi, (int)q);
      const size_t src = (size_t)bq * N + (size_t)ii;
      v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < nrow && fl) v[p] = *(const float4 *)(feats + src * C + 4 * l);
      if (q < nrow && l == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) rel[p][d] = xyz[src * 3 + d] - new_xyz[(size_t)bjq * 3 + d];
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const unsigned q = q_of[p];
      float *o = out + (size_t)(base + q) * W;
      if (q < nrow && fl) ((F4Dw *)(o + Cx + 4 * l))->v = f4v{v[p].x, v[p].y, v[p].z, v[p].w};
      if (q < nrow && l == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) o[d] = normalize ? __fdiv_rn(rel[p][d], radius) : rel[p][d];
      }
    }
    s += nrow;
    if (s >= (unsigned)ns) {
      s -= (unsigned)ns; ++bj;
      if (++j == (unsigned)m) { j = 0; ++b; }
    }
  }
}

// Narrow rows (W <= 16, e.g. the 3+3 columns of SA1): a lane per row, 32-bit index arithmetic, the wave's
// 64 rows staged through LDS so that the stores are contiguous.
__global__ __launch_bounds__(kBlock) void group_concat_rows_narrow_kernel(
    int N, int m, int ns, int C, int Cx, int normalize, float radius,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ feats, const int *__restrict__ idx, float *__restrict__ out, unsigned rows) {
  __shared__ float tile[kBlock / 64][64 * 17];
  const int lane = pn2_lane();
  const int wv = threadIdx.x >> 6;
  const int W = Cx + C;
  float *t = tile[wv];
  const unsigned nwaves = gridDim.x * (kBlock / 64);
  for (unsigned w0 = (blockIdx.x * (kBlock / 64) + wv) * 64u; w0 < rows; w0 += nwaves * 64u) {
    const unsigned row = w0 + lane;
    if (row < rows) {
      const unsigned bj = row / (unsigned)ns;
      const unsigned b = bj / (unsigned)m;
      const int ii = idx[row];
      const size_t src = (size_t)b * N + (size_t)ii;
      for (int w = 0; w < Cx; ++w) {
        float v = xyz[src * 3 + w] - new_xyz[(size_t)bj * 3 + w];
        if (normalize) v = __fdiv_rn(v, radius);
        t[lane * 17 + w] = v;
      }
      for (int c = 0; c < C; ++c) t[lane * 17 + Cx + c] = feats[src * C + c];
    }
    // same wave wrote and reads: no barrier needed, only the LDS counter
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
    const unsigned nrow = (rows - w0) < 64u ? (rows - w0) : 64u;
    float *o = out + (size_t)w0 * W;
    for (unsigned e = lane; e < nrow * (unsigned)W; e += 64) {
      const unsigned r = e / (unsigned)W, w = e - r * (unsigned)W;
      o[e] = t[r * 17 + w];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
}

// grad_feats[b, idx[row], :] += grad_out[row, col0 : col0 + C]   (hardware fp32 atomics), a wave per row group
// A wave walks whole neighbourhoods.  A ball-query row is "hits in ascending index, then padding with the FIRST hit"
// (EXT/src/ball_query_gpu.cu:34-38): with a small radius most of the ns slots of a neighbourhood are that padding, i.e.
// the SAME destination row.  Their gradients are summed in registers and leave as ONE atomic per channel; only the
// genuine other hits pay an atomic each (measured on the scene-graph encoders: 16-32 slots, 4-10 distinct hits).
// Any index row is handled correctly (slots equal to slot 0 are pre-reduced, the rest go out individually).
__global__ __launch_bounds__(kBlock) void group_rows_grad_kernel(
    int N, int m, int ns, int C, int ldg, int col0, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_feats, unsigned groups, unsigned groups_per_wave) {
  const int lane = pn2_lane();
  const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
  unsigned g0 = wave * groups_per_wave;
  if (g0 >= groups) return;
  unsigned g1 = g0 + groups_per_wave;
  if (g1 > groups) g1 = groups;
  for (unsigned gq = g0; gq < g1; ++gq) {
    const unsigned b = gq / (unsigned)m;
    const int *irow = idx + (size_t)gq * ns;
    const float *grow = grad_out + (size_t)gq * ns * ldg + col0;
    float *base = grad_feats + (size_t)b * N * C;
    const int first = irow[0];
    for (int c0 = 0; c0 < C; c0 += 64) {
      const int c = c0 + lane;
      const bool ok = c < C;
      float acc = 0.f;
      for (int s0 = 0; s0 < ns; s0 += 64) {
        const int cnt = ns - s0 < 64 ? ns - s0 : 64;
        const int myi = lane < cnt ? irow[s0 + lane] : first;
#pragma unroll 4
        for (int q = 0; q < cnt; ++q) {
          const int ii = __builtin_amdgcn_readlane(myi, q);
          const float g = ok ? grow[(size_t)(s0 + q) * ldg + c] : 0.f;
          if (ii == first) acc += g;                        // wave-uniform branch
          else if (ok) atomicAdd(base + (size_t)ii * C + c, g);
        }
      }
      if (ok) atomicAdd(base + (size_t)first * C + c, acc);
    }
  }
}

// The same walk over bf16 gradient rows (mixed-precision stack): lane l owns the channel pair (2l, 2l+1) [+128 per pass],
// one dword load per slot.  C, ldg and col0 even.
__global__ __launch_bounds__(kBlock) void group_rows_grad_bf16_kernel(
    int N, int m, int ns, int C, int ldg, int col0, const unsigned short *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_feats, unsigned groups, unsigned groups_per_wave) {
  const int lane = pn2_lane();
  const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
  unsigned g0 = wave * groups_per_wave;
  if (g0 >= groups) return;
  unsigned g1 = g0 + groups_per_wave;
  if (g1 > groups) g1 = groups;
  for (unsigned gq = g0; gq < g1; ++gq) {
    const unsigned b = gq / (unsigned)m;
    const int *irow = idx + (size_t)gq * ns;
    const unsigned short *grow = grad_out + (size_t)gq * ns * ldg + col0;
    float *base = grad_feats + (size_t)b * N * C;
    const int first = irow[0];
    for (int c0 = 0; c0 < C; c0 += 128) {
      const int c = c0 + 2 * lane;
      const bool ok = c < C;
      float acc0 = 0.f, acc1 = 0.f;
      for (int s0 = 0; s0 < ns; s0 += 64) {
        const int cnt = ns - s0 < 64 ? ns - s0 : 64;
        const int myi = lane < cnt ? irow[s0 + lane] : first;
#pragma unroll 4
        for (int q = 0; q < cnt; ++q) {
          const int ii = __builtin_amdgcn_readlane(myi, q);
          const unsigned w = ok ? *(const unsigned *)(grow + (size_t)(s0 + q) * ldg + c) : 0u;
          const float ga = __builtin_bit_cast(float, w << 16), gb = __builtin_bit_cast(float, w & 0xffff0000u);
          if (ii == first) { acc0 += ga; acc1 += gb; }         // wave-uniform branch
          else if (ok) {
            atomicAdd(base + (size_t)ii * C + c, ga);
            atomicAdd(base + (size_t)ii * C + c + 1, gb);
          }
        }
      }
      if (ok) {
        atomicAdd(base + (size_t)first * C + c, acc0);
        atomicAdd(base + (size_t)first * C + c + 1, acc1);
      }
    }
  }
}

// Vector form of the two kernels above for the padded part: R sub-waves of 64 / R lanes take R slots of the neighbourhood per
// wave instruction, a lane owns CH = 4 (fp32 rows) or 8 (bf16 rows) adjacent channels = one 16-byte load, four passes in
// flight; slots that repeat the first hit are summed in registers and the sub-waves' sums combined at the end.  (The
// dword-per-lane walk needs ns * C / 64 wave loads per neighbourhood — 192 at C = 192, ns = 64; this one 64 resp. 32.)  The
// genuine other hits keep the lane-per-channel form: their atomics must stay one contiguous 256-byte request per wave
// instruction — issued from the 16-byte layout (lane stride CH) the same atomics ran 4x slower on crowded balls.
template <int R, bool BF>
__global__ __launch_bounds__(kBlock) void group_rows_grad_vec_kernel(
    int N, int m, int ns, int C, int ldg, int col0, const void *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_feats, unsigned groups, unsigned groups_per_wave) {
  constexpr int LPR = 64 / R;
  constexpr int CH = BF ? 8 : 4;
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  const int lane = pn2_lane();
  const int sub = lane / LPR, l = lane % LPR;
  const bool fl = CH * l < C;
  const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
  unsigned g0 = wave * groups_per_wave;
  if (g0 >= groups) return;
  unsigned g1 = g0 + groups_per_wave;
  if (g1 > groups) g1 = groups;
  for (unsigned gq = g0; gq < g1; ++gq) {
    const unsigned b = gq / (unsigned)m;
    const int *irow = idx + (size_t)gq * ns;
    float *base = grad_feats + (size_t)b * N * C;
    const size_t row0 = (size_t)gq * ns;
    const int first = irow[0];
    float acc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = 0.f;
    for (int s0 = 0; s0 < ns; s0 += 64) {
      const int cnt = ns - s0 < 64 ? ns - s0 : 64;
      const int myi = lane < cnt ? irow[s0 + lane] : first;
      unsigned long long others = __ballot(myi != first);          // slots of this block that are genuine other hits
      if (others != ~0ull >> (64 - cnt)) {                           // some slot repeats the first hit (wave-uniform)
        for (int t = 0; t * R < cnt; t += 4) {
          float v[4][CH];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int q = (t + u) * R + sub;
            const int iq = __shfl(myi, q & 63);                    // every lane takes part (a lane that is masked off
            const bool on = q < cnt && fl && iq == first;           // would read as 0 for the others)
#pragma unroll
            for (int i = 0; i < CH; ++i) v[u][i] = 0.f;
            if (on) {
              const size_t e = (row0 + s0 + q) * (size_t)ldg + col0 + CH * l;
              if constexpr (BF) {
                const u4v w = *(const u4v *)((const unsigned short *)grad_out + e);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  v[u][2 * i] = __builtin_bit_cast(float, w[i] << 16);
                  v[u][2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
                }
              } else {
                const f4v w = ((const F4Dw *)((const float *)grad_out + e))->v;
                v[u][0] = w.x; v[u][1] = w.y; v[u][2] = w.z; v[u][3] = w.w;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[i] += v[u][i];
        }
      }
      while (others) {                                               // wave-uniform walk over the other hits
        // FOUR hits per round: their row loads are independent and fly together — one hit at a time, every hit paid a full
        // memory latency before its atomics could issue (half-full balls: 33 genuine hits of 64 slots at the scene-graph
        // encoders' second level, where this walk is most of the kernel)
        constexpr int U = 4;
        size_t eu[U];
        float *du[U];
        int nq = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          eu[u] = 0; du[u] = base;
          if (others) {
            const int q = __builtin_ctzll(others);
            others &= others - 1;
            eu[u] = (row0 + s0 + q) * (size_t)ldg + col0;
            du[u] = base + (size_t)__builtin_amdgcn_readlane(myi, q) * C;
            nq = u + 1;
          }
        }
        for (int c = lane; c < C; c += 64) {
          float gv[U];
#pragma unroll
          for (int u = 0; u < U; ++u)
            gv[u] = u < nq ? (BF ? __builtin_bit_cast(float, (unsigned)((const unsigned short *)grad_out)[eu[u] + c] << 16)
                                 : ((const float *)grad_out)[eu[u] + c])
                           : 0.f;
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (u < nq) atomicAdd(du[u] + c, gv[u]);
        }
      }
    }
#pragma unroll
    for (int d = 32; d >= LPR; d >>= 1)
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[i] += __shfl_xor(acc[i], d);
    if (sub == 0 && fl)
#pragma unroll
      for (int i = 0; i < CH; ++i) atomicAdd(base + (size_t)first * C + CH * l + i, acc[i]);
  }
}

template <bool BF>
bool launch_rows_grad_vec(int N, int m, int ns, int C, int ldg, int col0, const void *grad_out, const int *idx,
                          float *grad_feats, unsigned groups, unsigned gpw, unsigned grid, hipStream_t stream) {
  constexpr int CH = BF ? 8 : 4;
  // 16-byte loads: bf16 rows need 16-byte alignment of every (row, column group), fp32 rows dword alignment
  if (C % CH != 0 || C > 64 * CH) return false;
  if (BF && ((ldg & 7) || (col0 & 7) || (((size_t)grad_out) & 15))) return false;
#define PN2_VEC(R) hipLaunchKernelGGL((group_rows_grad_vec_kernel<R, BF>), dim3(grid), dim3(kBlock), 0, stream, N, m, ns, C, \
                                      ldg, col0, grad_out, idx, grad_feats, groups, gpw)
  if (C <= 16 * CH) PN2_VEC(4);
  else if (C <= 32 * CH) PN2_VEC(2);
  else PN2_VEC(1);
#undef PN2_VEC
  return true;
}

// max over the ns axis of x (R, ns, C); first maximal s wins (torch max_pool2d).
__global__ __launch_bounds__(kBlock) void rows_max_kernel(int ns, int C, const float *__restrict__ x,
                                                         float *__restrict__ out, int *__restrict__ arg,
                                                         size_t total /* R*C */) {
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (size_t)gridDim.x * kBlock) {
    const size_t r = e / C;
    const int c = (int)(e - r * C);
    const float *p = x + r * ns * C + c;
    float best = p[0];
    int bi = 0;
    for (int s = 1; s < ns; ++s) {
      const float v = p[(size_t)s * C];
      if (v > best || (v != v && best == best)) { best = v; bi = s; }  // NaN propagates like torch
    }
    out[e] = best;
    if (arg) arg[e] = bi;
  }
}

__global__ __launch_bounds__(kBlock) void rows_max_grad_kernel(int ns, int C,
                                                              const float *__restrict__ grad_out,
                                                              const int *__restrict__ arg,
                                                              float *__restrict__ grad_x,
                                                              size_t total /* R*ns*C */) {
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (size_t)gridDim.x * kBlock) {
    const size_t rs = e / C;
    const int c = (int)(e - rs * C);
    const size_t r = rs / ns;
    const int s = (int)(rs - r * ns);
    const size_t rc = r * C + c;
    grad_x[e] = (arg[rc] == s) ? grad_out[rc] : 0.f;
  }
}

constexpr unsigned kMaxGrid = 256u * 32u;  // grid-stride beyond 8192 workgroups
inline unsigned capped(size_t work) {
  unsigned g = grid1d(work);
  return g > kMaxGrid ? kMaxGrid : g;
}

}  // namespace

extern "C" int pn2_gather_points(int B, int C, int N, int m, const float *points,
                                 const int *idx, float *out, void *stream) {
  if (B < 0 || C < 0 || N < 0 || m < 0) return PN2_EINVAL;
  const size_t total = (size_t)B * C * m;
  if (total == 0) return PN2_OK;
  if (!points || !idx || !out) return PN2_ENULL;
  hipLaunchKernelGGL(gather_points_kernel, dim3(capped(total)), dim3(kBlock), 0, (hipStream_t)stream,
                     C, N, m, points, idx, out, total);
  return pn2_check_launch();
}

extern "C" int pn2_gather_points_grad(int B, int C, int N, int m, const float *grad_out,
                                      const int *idx, float *grad_points, void *stream) {
  if (B < 0 || C < 0 || N < 0 || m < 0) return PN2_EINVAL;
  const size_t total = (size_t)B * C * m;
  if (total == 0) return PN2_OK;
  if (!grad_out || !idx || !grad_points) return PN2_ENULL;
  hipLaunchKernelGGL(gather_points_grad_kernel, dim3(capped(total)), dim3(kBlock), 0,
                     (hipStream_t)stream, C, N, m, grad_out, idx, grad_points, total);
  return pn2_check_launch();
}

extern "C" int pn2_group_points(int B, int C, int N, int npoints, int nsample,
                                const float *points, const int *idx, float *out,
                                void *stream) {
  if (B < 0 || C < 0 || N < 0 || npoints < 0 || nsample < 0) return PN2_EINVAL;
  const size_t S = (size_t)npoints * nsample;
  if ((size_t)B * S == 0 || C == 0) return PN2_OK;
  if (!points || !idx || !out) return PN2_ENULL;
  hipLaunchKernelGGL(group_points_kernel, dim3(capped((size_t)B * S)), dim3(kBlock), 0,
                     (hipStream_t)stream, C, N, S, points, idx, out, B);
  return pn2_check_launch();
}

extern "C" int pn2_group_points_grad(int B, int C, int N, int npoints, int nsample,
                                     const float *grad_out, const int *idx,
                                     float *grad_points, void *stream) {
  if (B < 0 || C < 0 || N < 0 || npoints < 0 || nsample < 0) return PN2_EINVAL;
  const size_t S = (size_t)npoints * nsample;
  if ((size_t)B * S == 0 || C == 0) return PN2_OK;
  if (!grad_out || !idx || !grad_points) return PN2_ENULL;
  hipLaunchKernelGGL(group_points_grad_kernel, dim3(capped((size_t)B * S)), dim3(kBlock), 0,
                     (hipStream_t)stream, C, N, S, grad_out, idx, grad_points, B);
  return pn2_check_launch();
}

// bf16 rows for the mixed-precision shared MLP (csrc/mlp_bf16.hip): same gather as the wide kernel, the row written as
// bf16 with a pitch `ldo` that is a multiple of 8 elements (16-byte row groups for the GEMM loader); the pad columns
// W..ldo-1 are written as zeros.  Lane l produces the column pair (2l, 2l+1) [+128 per pass] and stores one dword.
__global__ __launch_bounds__(kBlock) void group_concat_rows_bf16_kernel(
    int N, int m, int ns, int C, int Cx, int normalize, float radius, int ldo,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ feats, const int *__restrict__ idx, unsigned *__restrict__ out /* bf16 pairs */,
    unsigned rows, unsigned rows_per_wave) {
  const int lane = pn2_lane();
  const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
  const int W = Cx + C;
  // 16-byte feature loads: xyz in front, feature width a multiple of 4 (<= 248 so that 62 lanes cover a row), 16-byte
  // aligned feature rows
  const bool fast = Cx == 3 && C >= 4 && (C & 3) == 0 && C <= 248 && (((size_t)feats) & 15) == 0;
  unsigned r0 = wave * rows_per_wave;
  if (r0 >= rows) return;
  unsigned r1 = r0 + rows_per_wave;
  if (r1 > rows) r1 = rows;
  unsigned bj = r0 / (unsigned)ns;
  unsigned s = r0 - bj * (unsigned)ns;
  unsigned b = bj / (unsigned)m;
  unsigned j = bj - b * (unsigned)m;
  for (unsigned base = r0; base < r1; base += 8) {
    const unsigned nrow = (r1 - base) < 8u ? (r1 - base) : 8u;
    const int myi = lane < (int)nrow ? idx[base + lane] : 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if ((unsigned)q >= nrow) break;            // wave-uniform
      const int ii = __builtin_amdgcn_readlane(myi, q);
      const size_t src = (size_t)b * N + (size_t)ii;
      unsigned *o = out + (size_t)(base + q) * (ldo / 2);
      if (fast) {
        // [x y z | f0 f1 ...]: lane j loads features 4j..4j+3 with ONE 16-byte load; the row's dword d holds columns
        // (2d, 2d+1), i.e. dword 2j+1 = (previous lane's 4th value | f[4j]) and dword 2j+2 = (f[4j+1] | f[4j+2]):
        // one shuffle, one 8-byte store per lane.  Lane 63 writes (x, y); the lane behind the last feature lane
        // closes the row (last feature | zero pad).
        const float *fr = feats + src * C;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int nl = C >> 2;                                   // feature lanes
        if (lane < nl) v = *(const float4 *)(fr + 4 * lane);
        float rx = 0.f, ry = 0.f, rz = 0.f;
        if (lane == 0 || lane == 63) {
          rx = xyz[src * 3 + 0] - new_xyz[(size_t)bj * 3 + 0];
          ry = xyz[src * 3 + 1] - new_xyz[(size_t)bj * 3 + 1];
          rz = xyz[src * 3 + 2] - new_xyz[(size_t)bj * 3 + 2];
          if (normalize) { rx = __fdiv_rn(rx, radius); ry = __fdiv_rn(ry, radius); rz = __fdiv_rn(rz, radius); }
        }
        float prev = __shfl_up(v.w, 1);
        if (lane == 0) prev = rz;
        auto pk = [](float lo, float hi) {
          return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) |
                 ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
        };
        if (lane < nl) {
          o[2 * lane + 1] = pk(prev, v.x);
          o[2 * lane + 2] = pk(v.y, v.z);
        } else if (lane == nl) {
          for (int d = 2 * nl + 1; d < ldo / 2; ++d) o[d] = d == 2 * nl + 1 ? pk(prev, 0.f) : 0u;
        }
        if (lane == 63) o[0] = pk(rx, ry);
      } else {
      const float *f = feats + src * C - Cx;     // f[c] = feature column c - Cx
      for (int c = 2 * lane; c < ldo; c += 128) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int cc = c + h;
          float t = 0.f;
          if (cc < Cx) {
            t = xyz[src * 3 + cc] - new_xyz[(size_t)bj * 3 + cc];
            if (normalize) t = __fdiv_rn(t, radius);
          } else if (cc < W) {
            t = f[cc];
          }
          v[h] = t;
        }
        const unsigned lo = __builtin_bit_cast(unsigned short, (__bf16)v[0]);
        const unsigned hi = __builtin_bit_cast(unsigned short, (__bf16)v[1]);
        o[c >> 1] = lo | (hi << 16);
      }
      }
      if (++s == (unsigned)ns) {
        s = 0; ++bj;
        if (++j == (unsigned)m) { j = 0; ++b; }
      }
    }
  }
}

// bf16 rows, 16-byte stores: lane g of a row produces the output columns 8g .. 8g+7 = features 8g-3 .. 8g+4 with two
// dword-aligned 16-byte loads (lane 0: x y z f0 | f1..f4) and writes ONE aligned 16-byte group; R rows per wave
// instruction (64 / R lanes per row, pitch <= 8 * 64 / R), eight rows in flight.  (The dword-store version above moves
// 2 x 4 bytes per lane and row and ran at 1.1-1.9 TB/s, slower than the fp32 kernel moving twice the bytes.)
template <int R>
__global__ __launch_bounds__(kBlock) void group_concat_rows_bf16_wide8_kernel(
    int N, int m, int ns, int C, int normalize, float radius, int ldo,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ feats, const int *__restrict__ idx, unsigned *__restrict__ out,
    unsigned rows, unsigned rows_per_wave) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  constexpr int LPR = 64 / R;
  constexpr int P = 8 / R;
  const int lane = pn2_lane();
  const unsigned wave = xcd_slab_wave();
  const int sub = lane / LPR, g = lane % LPR;
  const bool active = g < ldo / 8;
  const int c0 = 8 * g - 3;                        // feature column of output column 8g
  const bool whole = g > 0 && c0 + 8 <= C;         // both 16-byte loads lie inside the feature row
  // one output row group (8 columns of this lane) gathered and packed
  auto load_pack = [&](size_t src, unsigned bjq) {
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = 0.f;
    const float *fr = feats + src * C;
    if (active) {
      if (whole) {
        const f4v a = ((const F4Dw *)(fr + c0))->v, c = ((const F4Dw *)(fr + c0 + 4))->v;
        t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w; t[4] = c.x; t[5] = c.y; t[6] = c.z; t[7] = c.w;
      } else if (g == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          t[d] = xyz[src * 3 + d] - new_xyz[(size_t)bjq * 3 + d];
          if (normalize) t[d] = __fdiv_rn(t[d], radius);
        }
        t[3] = fr[0];
        const f4v c = ((const F4Dw *)(fr + 1))->v;
        t[4] = c.x; t[5] = c.y; t[6] = c.z; t[7] = c.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (c0 + i < C) t[i] = fr[c0 + i];
      }
    }
    u4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned lo = __builtin_bit_cast(unsigned short, (__bf16)t[2 * i]);
      const unsigned hi = __builtin_bit_cast(unsigned short, (__bf16)t[2 * i + 1]);
      w[i] = lo | (hi << 16);
    }
    return w;
  };
  unsigned r0 = wave * rows_per_wave;
  if (r0 >= rows) return;
  unsigned r1 = r0 + rows_per_wave;
  if (r1 > rows) r1 = rows;
  unsigned bj = r0 / (unsigned)ns;
  unsigned s = r0 - bj * (unsigned)ns;
  unsigned b = bj / (unsigned)m;
  unsigned j = bj - b * (unsigned)m;
  for (unsigned base = r0; base < r1; base += 8) {
    const unsigned nrow = (r1 - base) < 8u ? (r1 - base) : 8u;
    const int myi = lane < (int)nrow ? idx[base + lane] : 0;
    // one neighbourhood, one repeated index (ball-query padding): eight copies of one row — gather once, store eight times
    const bool same = nrow == 8u && s + 8u <= (unsigned)ns &&
                      __all(lane >= 8 || myi == __builtin_amdgcn_readlane(myi, 0));
    if (same) {                                   // wave-uniform
      const size_t src = (size_t)b * N + (size_t)__builtin_amdgcn_readlane(myi, 0);
      u4 w = load_pack(src, bj);
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (active) *(u4 *)(out + (size_t)(base + p * R + sub) * (ldo / 2) + 4 * g) = w;
      s += 8u;
      if (s >= (unsigned)ns) {
        s -= (unsigned)ns; ++bj;
        if (++j == (unsigned)m) { j = 0; ++b; }
      }
      continue;
    }
    float f[P][8];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const unsigned q = p * R + sub;
      const unsigned wrap = (s + q) >= (unsigned)ns ? 1u : 0u;
      const unsigned bjq = bj + wrap;
      const unsigned bq = b + ((wrap && j + 1 == (unsigned)m) ? 1u : 0u);
      const int ii = __shfl(myi, (int)q);
      const size_t src = (size_t)bq * N + (size_t)ii;
      const float *fr = feats + src * C;
#pragma unroll
      for (int i = 0; i < 8; ++i) f[p][i] = 0.f;
      if (q < nrow && active) {
        if (whole) {
          const f4v a = ((const F4Dw *)(fr + c0))->v, c = ((const F4Dw *)(fr + c0 + 4))->v;
          f[p][0] = a.x; f[p][1] = a.y; f[p][2] = a.z; f[p][3] = a.w;
          f[p][4] = c.x; f[p][5] = c.y; f[p][6] = c.z; f[p][7] = c.w;
        } else if (g == 0) {
#pragma unroll
          for (int d = 0; d < 3; ++d) f[p][d] = xyz[src * 3 + d] - new_xyz[(size_t)bjq * 3 + d];
          f[p][3] = fr[0];
          const f4v c = ((const F4Dw *)(fr + 1))->v;      // C >= 5 (the launcher only takes pitches > 16)
          f[p][4] = c.x; f[p][5] = c.y; f[p][6] = c.z; f[p][7] = c.w;
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (c0 + i < C) f[p][i] = fr[c0 + i];
        }
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const unsigned q = p * R + sub;
      if (normalize && g == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) f[p][d] = __fdiv_rn(f[p][d], radius);
      }
      u4 w;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned lo = __builtin_bit_cast(unsigned short, (__bf16)f[p][2 * i]);
        const unsigned hi = __builtin_bit_cast(unsigned short, (__bf16)f[p][2 * i + 1]);
        w[i] = lo | (hi << 16);
      }
      if (q < nrow && active) *(u4 *)(out + (size_t)(base + q) * (ldo / 2) + 4 * g) = w;
    }
    s += nrow;
    if (s >= (unsigned)ns) {
      s -= (unsigned)ns; ++bj;
      if (++j == (unsigned)m) { j = 0; ++b; }
    }
  }
}

// Narrow bf16 rows (pitch 8 or 16: the 3+3 / 3+4 columns of an SA1 level): a lane per row, one or two 16-byte stores.
template <int LDO>
__global__ __launch_bounds__(kBlock) void group_concat_rows_bf16_narrow_kernel(
    int N, int m, int ns, int C, int Cx, int normalize, float radius,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ feats, const int *__restrict__ idx, unsigned *__restrict__ out, unsigned rows) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const int W = Cx + C;
  for (unsigned r = blockIdx.x * kBlock + threadIdx.x; r < rows; r += gridDim.x * kBlock) {
    const unsigned bj = r / (unsigned)ns;
    const unsigned b = bj / (unsigned)m;
    const size_t src = (size_t)b * N + (size_t)idx[r];
    float v[LDO];
#pragma unroll
    for (int c = 0; c < LDO; ++c) {
      float t = 0.f;
      if (c < Cx) {
        t = xyz[src * 3 + c] - new_xyz[(size_t)bj * 3 + c];
        if (normalize) t = __fdiv_rn(t, radius);
      } else if (c < W) {
        t = feats[src * C + (c - Cx)];
      }
      v[c] = t;
    }
#pragma unroll
    for (int h = 0; h < LDO / 8; ++h) {
      u4 w;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned lo = __builtin_bit_cast(unsigned short, (__bf16)v[8 * h + 2 * i]);
        const unsigned hi = __builtin_bit_cast(unsigned short, (__bf16)v[8 * h + 2 * i + 1]);
        w[i] = lo | (hi << 16);
      }
      *(u4 *)(out + (size_t)r * (LDO / 2) + 4 * h) = w;
    }
  }
}

extern "C" int pn2_group_concat_rows_bf16(int B, int N, int m, int ns, int C, int use_xyz, int normalize, float radius,
                                          int ldo, const float *xyz, const float *new_xyz, const float *feats,
                                          const int *idx, void *out, void *stream) {
  if (B < 0 || N < 0 || m < 0 || ns < 0 || C < 0) return PN2_EINVAL;
  const int Cx = use_xyz ? 3 : 0;
  if (ldo < Cx + C || ldo % 8 != 0) return PN2_EINVAL;
  const size_t rows_sz = (size_t)B * m * ns;
  if (rows_sz == 0 || Cx + C == 0) return PN2_OK;
  if (!idx || !out) return PN2_ENULL;
  if (Cx && (!xyz || !new_xyz)) return PN2_ENULL;
  if (C && !feats) return PN2_ENULL;
  if (normalize && !(radius > 0.f)) return PN2_EINVAL;
  if (rows_sz >= 0x7fffffffull) return PN2_EINVAL;
  const unsigned rows = (unsigned)rows_sz;
  if (ldo <= 16) {
    unsigned grid = (rows + kBlock - 1) / kBlock;
    if (grid > 8192) grid = 8192;
    if (ldo == 8)
      hipLaunchKernelGGL(group_concat_rows_bf16_narrow_kernel<8>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, m, ns,
                         C, Cx, normalize, radius, xyz, new_xyz, feats, idx, (unsigned *)out, rows);
    else
      hipLaunchKernelGGL(group_concat_rows_bf16_narrow_kernel<16>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, m, ns,
                         C, Cx, normalize, radius, xyz, new_xyz, feats, idx, (unsigned *)out, rows);
    return pn2_check_launch();
  }
  const unsigned want_waves = 256u * 16u;
  unsigned rpw = (rows + want_waves - 1) / want_waves;
  rpw = (rpw + 7u) & ~7u;
  const unsigned waves = (rows + rpw - 1) / rpw;
  const unsigned grid = ((waves + kBlock / 64 - 1) / (kBlock / 64) + 7u) & ~7u;        // multiple of 8: xcd_slab_wave()
  // 16-byte variant: xyz in front, neighbourhoods of at least 8, pitch <= 512 columns
  const bool wide8 = Cx == 3 && C >= 5 && ns >= 8 && ldo <= 512;
  if (wide8 && ldo <= 128)
    hipLaunchKernelGGL(group_concat_rows_bf16_wide8_kernel<4>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, m, ns, C,
                       normalize, radius, ldo, xyz, new_xyz, feats, idx, (unsigned *)out, rows, rpw);
  else if (wide8 && ldo <= 256)
    hipLaunchKernelGGL(group_concat_rows_bf16_wide8_kernel<2>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, m, ns, C,
                       normalize, radius, ldo, xyz, new_xyz, feats, idx, (unsigned *)out, rows, rpw);
  else if (wide8)
    hipLaunchKernelGGL(group_concat_rows_bf16_wide8_kernel<1>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, m, ns, C,
                       normalize, radius, ldo, xyz, new_xyz, feats, idx, (unsigned *)out, rows, rpw);
  else
    hipLaunchKernelGGL(group_concat_rows_bf16_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, m, ns, C, Cx,
                       normalize, radius, ldo, xyz, new_xyz, feats, idx, (unsigned *)out, rows, rpw);
  return pn2_check_launch();
}

extern "C" int pn2_group_concat_rows(int B, int N, int m, int ns, int C, int use_xyz,
                                     int normalize, float radius, const float *xyz,
                                     const float *new_xyz, const float *feats,
                                     const int *idx, float *out, void *stream) {
  if (B < 0 || N < 0 || m < 0 || ns < 0 || C < 0) return PN2_EINVAL;
  const int Cx = use_xyz ? 3 : 0;
  const size_t total = (size_t)B * m * ns * (size_t)(Cx + C);
  if (total == 0) return PN2_OK;
  if (!idx || !out) return PN2_ENULL;
  if (Cx && (!xyz || !new_xyz)) return PN2_ENULL;
  if (C && !feats) return PN2_ENULL;
  if (normalize && !(radius > 0.f)) return PN2_EINVAL;
  const size_t rows_sz = (size_t)B * m * ns;
  if (rows_sz >= 0x7fffffffull) return PN2_EINVAL;
  const unsigned rows = (unsigned)rows_sz;
  if (Cx + C <= 16) {
    const unsigned waves = (rows + 63) / 64;
    unsigned grid = (waves + kBlock / 64 - 1) / (kBlock / 64);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(group_concat_rows_narrow_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, m, ns,
                       C, Cx, normalize, radius, xyz, new_xyz, feats, idx, out, rows);
  } else {
    // ~16 waves per CU, each a contiguous slab of rows (multiple of 8)
    const unsigned want_waves = 256u * 16u;
    unsigned rpw = (rows + want_waves - 1) / want_waves;
    rpw = (rpw + 7u) & ~7u;
    const unsigned waves = (rows + rpw - 1) / rpw;
    const unsigned grid = ((waves + kBlock / 64 - 1) / (kBlock / 64) + 7u) & ~7u;      // multiple of 8: xcd_slab_wave()
    // 16-byte variant: xyz in front, feature width a multiple of 4 up to 256, neighbourhoods of at least 8 (a batch of
    // eight rows then crosses at most one centre), 16-byte aligned feature rows
    const bool wide4 = Cx == 3 && C >= 4 && (C & 3) == 0 && C <= 256 && ns >= 8 && (((size_t)feats) & 15) == 0;
    if (wide4 && C <= 64)
      hipLaunchKernelGGL(group_concat_rows_wide4_kernel<4>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, m, ns, C,
                         normalize, radius, xyz, new_xyz, feats, idx, out, rows, rpw);
    else if (wide4 && C <= 128)
      hipLaunchKernelGGL(group_concat_rows_wide4_kernel<2>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, m, ns, C,
                         normalize, radius, xyz, new_xyz, feats, idx, out, rows, rpw);
    else if (wide4)
      hipLaunchKernelGGL(group_concat_rows_wide4_kernel<1>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, m, ns, C,
                         normalize, radius, xyz, new_xyz, feats, idx, out, rows, rpw);
    else
      hipLaunchKernelGGL(group_concat_rows_wide_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, m, ns, C,
                         Cx, normalize, radius, xyz, new_xyz, feats, idx, out, rows, rpw);
  }
  return pn2_check_launch();
}

extern "C" int pn2_group_rows_grad(int B, int N, int m, int ns, int C, int ldg, int col0,
                                   const float *grad_out, const int *idx,
                                   float *grad_feats, void *stream) {
  if (B < 0 || N < 0 || m < 0 || ns < 0 || C < 0 || col0 < 0 || ldg < col0 + C) return PN2_EINVAL;
  const size_t total = (size_t)B * m * ns * (size_t)C;
  if (total == 0) return PN2_OK;
  if (!grad_out || !idx || !grad_feats) return PN2_ENULL;
  const size_t groups_sz = (size_t)B * m;
  if (groups_sz >= 0x7fffffffull) return PN2_EINVAL;
  const unsigned groups = (unsigned)groups_sz;
  const unsigned want_waves = 256u * 32u;
  const unsigned gpw = (groups + want_waves - 1) / want_waves;
  const unsigned waves = (groups + gpw - 1) / gpw;
  const unsigned grid = (waves + kBlock / 64 - 1) / (kBlock / 64);
  if (!launch_rows_grad_vec<false>(N, m, ns, C, ldg, col0, grad_out, idx, grad_feats, groups, gpw, grid, (hipStream_t)stream))
    hipLaunchKernelGGL(group_rows_grad_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, m, ns, C, ldg,
                       col0, grad_out, idx, grad_feats, groups, gpw);
  return pn2_check_launch();
}

extern "C" int pn2_group_rows_grad_bf16(int B, int N, int m, int ns, int C, int ldg, int col0, const void *grad_out,
                                        const int *idx, float *grad_feats, void *stream) {
  if (B < 0 || N < 0 || m < 0 || ns < 0 || C < 0 || col0 < 0 || ldg < col0 + C) return PN2_EINVAL;
  if ((C & 1) || (ldg & 1) || (col0 & 1) || (((size_t)grad_out) & 3)) return PN2_EINVAL;
  const size_t total = (size_t)B * m * ns * (size_t)C;
  if (total == 0) return PN2_OK;
  if (!grad_out || !idx || !grad_feats) return PN2_ENULL;
  const size_t groups_sz = (size_t)B * m;
  if (groups_sz >= 0x7fffffffull) return PN2_EINVAL;
  const unsigned groups = (unsigned)groups_sz;
  const unsigned want_waves = 256u * 32u;
  const unsigned gpw = (groups + want_waves - 1) / want_waves;
  const unsigned waves = (groups + gpw - 1) / gpw;
  const unsigned grid = (waves + kBlock / 64 - 1) / (kBlock / 64);
  if (!launch_rows_grad_vec<true>(N, m, ns, C, ldg, col0, grad_out, idx, grad_feats, groups, gpw, grid, (hipStream_t)stream))
    hipLaunchKernelGGL(group_rows_grad_bf16_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, m, ns, C, ldg, col0,
                       (const unsigned short *)grad_out, idx, grad_feats, groups, gpw);
  return pn2_check_launch();
}

extern "C" int pn2_rows_max(int64_t R, int ns, int C, const float *x, float *out,
                            int *arg, void *stream) {
  if (R < 0 || ns <= 0 || C < 0) return PN2_EINVAL;
  const size_t total = (size_t)R * C;
  if (total == 0) return PN2_OK;
  if (!x || !out) return PN2_ENULL;
  hipLaunchKernelGGL(rows_max_kernel, dim3(capped(total)), dim3(kBlock), 0, (hipStream_t)stream, ns,
                     C, x, out, arg, total);
  return pn2_check_launch();
}

extern "C" int pn2_rows_max_grad(int64_t R, int ns, int C, const float *grad_out,
                                 const int *arg, float *grad_x, void *stream) {
  if (R < 0 || ns <= 0 || C < 0) return PN2_EINVAL;
  const size_t total = (size_t)R * ns * C;
  if (total == 0) return PN2_OK;
  if (!grad_out || !arg || !grad_x) return PN2_ENULL;
  hipLaunchKernelGGL(rows_max_grad_kernel, dim3(capped(total)), dim3(kBlock), 0,
                     (hipStream_t)stream, ns, C, grad_out, arg, grad_x, total);
  return pn2_check_launch();
}
