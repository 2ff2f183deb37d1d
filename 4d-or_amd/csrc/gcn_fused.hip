// gcn_fused.hip — the TripletGCN layer as a handful of fused per-scan kernels (round 4).
//
// Reference: scene_graph_prediction/scene_graph_helpers/model/gcns/network_TripletGCN.py:11-58 —
//   message  : nn1(cat[x_i, e, x_j])  with nn1 = Linear(2 dn + de -> dh) BN ReLU Linear(dh -> 2 dh + de) BN ReLU   (:36-37, 45-47)
//   split    : [:dh] | [dh : dh + de] | [dh + de :]; node message = first + last, new edge feature = middle         (:48-52)
//   aggregate: scatter(add) of the node messages onto edge_index[1]                                                   (:54-58)
//   update   : nn2 = Linear(dh -> dh) BN ReLU Linear(dh -> dn)                                                        (:38, 42-43)
// with BatchNorm1d(track_running_stats=False) (:20): batch statistics of ONE scan's rows, in training and in evaluation.
// Through torch + the row kernels of gcn_rows.hip that is ~60 launches forward and ~180 backward per layer for matrices of
// 9 .. 110 rows: the step is bound by the host thread and by launch gaps, not by arithmetic (profiles/r03_sgp_gpu_busy_fraction.md).
//
// Here a scan's rows (<= 128: the dataset has at most 11 objects = 110 ordered pairs) are ONE workgroup's row range, so
// everything BatchNorm needs is local to the workgroup that owns a 32-column tile of a Linear's output:
//   pn2_gcn_linear          gather (the concatenation is never built) -> Linear on the fp32 matrix cores -> + bias ->
//                           per-scan batch statistics (two passes over the accumulators) -> BN -> ReLU, one launch;
//   pn2_gcn_linear_grad_w   the block's backward up to the weights, two launches: ReLU mask + BatchNorm backward per scan
//                           (per-scan sums local to the workgroup) -> gz (kept for the input gradient), dgamma / dbeta /
//                           dbias; then dW += gz^T A as ONE product over all rows of the batch on the matrix cores, no
//                           atomics; the adjoint of split + aggregate ([g_agg[dst] | g_edge | g_agg[dst]]) is read in
//                           place of a materialised gradient;
//   pn2_gcn_linear_grad_x   input gradient gz W, written as rows or scattered through the triplet gather's adjoint
//                           (x[dst] / e / x[src] column blocks);
//   pn2_gcn_layer_forward / _backward   the whole layer's launch sequence from one C call each way (pn2_gcn_layer).
// v_mfma_f32_32x32x2_f32 throughout (exact fp32 products and sums, like the shared-MLP kernels); a wave's K loop splits the
// reduction range in two halves, one per 32-lane group, so every lane streams CONTIGUOUS floats of its A row / W row
// (16-byte loads).  Tested at 1e-4 against the oracle GCN (torch on the CPU restatement) like the unfused path.
#include "pn2_common.h"

namespace {
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int kGcnRows = 128;      // rows of one scan a workgroup covers: 4 waves x 32

__device__ __forceinline__ f16v mfma2(float a, float b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// row of accumulator register q in a 32 x 32 tile: lane group h = lane / 32
__device__ __forceinline__ int acc_row(int q, int h) { return 8 * (q >> 2) + 4 * h + (q & 3); }

struct GcnTriplet {               // cat[x[dst], e, x[src]] as a virtual (E, 2 dn + de) matrix
  const float *x, *e;
  const int64_t *dst, *src;
  int dn, de;
};

struct GcnFwdArgs {
  const float *A;                 // AMODE 0: (R, lda) rows
  int lda;
  GcnTriplet t;                   // AMODE 1
  const float *W, *bias;          // (N, K) row-major (torch Linear), (N) or null
  const int64_t *ptr;             // (S + 1) row offsets of the scans
  const float *gamma, *beta;      // BatchNorm (BN instantiation)
  float eps;
  float *Ypre, *Out, *mean, *rstd;   // (R, N) pre-BN, (R, N) result, (S, N), (S, N)
  int K, N, relu;
};

// K loop of a 32 x 32 tile, software-pipelined: a ring of four 16-float blocks per operand (the loads of block i + 3 are in
// flight while block i feeds the matrix core: at one wave per SIMD nothing else hides the ~1 us of an L2 round trip).
// `pa(k)` / `pb(k)`: 16-byte loads of the lane's A row / W row at offset k of its half of the reduction range.
template <class FA, class FB>
__device__ __forceinline__ void gcn_kloop(int k_begin, int k_end, bool rv, FA pa, FB pb, f16v &acc) {
  f4 ab[4][4], bb[4][4];
  const int nblk = (k_end - k_begin) >> 4;
  auto load = [&](int u, int blk) {
    const int k = k_begin + 16 * blk;
#pragma unroll
    for (int j = 0; j < 4; ++j) { ab[u][j] = pa(k + 4 * j); bb[u][j] = pb(k + 4 * j); }
  };
  auto compute = [&](int u) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f4 av = ab[u][j];
      if (!rv) av = f4{0.f, 0.f, 0.f, 0.f};
      acc = mfma2(av.x, bb[u][j].x, acc); acc = mfma2(av.y, bb[u][j].y, acc);
      acc = mfma2(av.z, bb[u][j].z, acc); acc = mfma2(av.w, bb[u][j].w, acc);
    }
  };
#pragma unroll
  for (int u = 0; u < 3; ++u)
    if (u < nblk) load(u, u);
  for (int blk = 0; blk < nblk; blk += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (blk + u + 3 < nblk) load((u + 3) & 3, blk + u + 3);
      if (blk + u < nblk) compute(u);
    }
  }
}

// Row tiles of a scan (T = 1 .. 4) and the split of the reduction range over the eight waves of a workgroup: a 32 x 32 tile
// is ONE dependent chain of matrix-core instructions (64 cycles each, K / 2 of them), so the range is cut into 8 / T' parts
// (T' = T rounded up to a power of two: every node matrix, <= 32 rows, runs on eight eighths) and the partial tiles meet in LDS.
constexpr int kGcnWaves = 8;
struct GcnSplit { int tile, part, nparts; };
__device__ __forceinline__ GcnSplit gcn_split(int Rs, int w, int half_len) {
  const int T = (Rs + 31) >> 5;
  const int tp = T <= 1 ? 1 : (T == 2 ? 2 : 4);
  int nparts = kGcnWaves / tp;
  while (nparts > 1 && ((half_len / nparts) & 15 || (half_len % nparts))) nparts >>= 1;
  GcnSplit sp;
  sp.nparts = nparts;
  sp.tile = w % tp;
  sp.part = w / tp;
  if (sp.part >= nparts) { sp.part = 0; sp.tile = 4; }             // (a spare wave: an empty tile)
  return sp;
}
// sum of the partial accumulators of the waves that share a tile; the part-0 wave returns the total
__device__ __forceinline__ void gcn_join(float (*part)[16][64], const GcnSplit &sp, int Rs, int w, int lane, f16v &acc) {
  if (sp.nparts > 1) {                                             // (workgroup-uniform)
    const int T = (Rs + 31) >> 5;
    const int tp = T <= 1 ? 1 : (T == 2 ? 2 : 4);
    if (sp.part != 0 && sp.tile < 4) {
#pragma unroll
      for (int q = 0; q < 16; ++q) part[w][q][lane] = acc[q];
    }
    __syncthreads();
    if (sp.part == 0 && sp.tile < 4) {
      for (int o = 1; o < sp.nparts; ++o) {
        const int ow = sp.tile + tp * o;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += part[ow][q][lane];
      }
    }
  }
}

// grid (N / 32, S), 256 threads: wave w = rows [32 w, 32 w + 32) of scan blockIdx.y (see gcn_split), columns [32 bx, +32)
template <int AMODE, bool BN>
__global__ __launch_bounds__(512) void gcn_linear_kernel(const GcnFwdArgs a) {
  __shared__ float red[2][kGcnWaves][32];
  __shared__ float part[kGcnWaves][16][64];
  const int lane = pn2_lane(), wv = threadIdx.x >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int s = blockIdx.y, c0 = blockIdx.x * 32;
  const long long r0 = a.ptr[s];
  const int Rs = (int)(a.ptr[s + 1] - r0);
  const int K = a.K, Kh = K >> 1;
  if (Rs > kGcnRows) {
    // A scan this workgroup cannot cover (pn2_gcn_fused_supported says so for callers that know their sizes on the host;
    // the pointer table lives on the device, so the C entry cannot check it): every result row of the scan is NaN instead
    // of statistics over its first 128 rows and uninitialised memory behind them.  Block-uniform exit.
    const float qnan = __builtin_nanf("");
    for (int i = threadIdx.x; i < Rs * 32; i += blockDim.x) {
      const size_t o = (size_t)(r0 + (i >> 5)) * a.N + c0 + (i & 31);
      a.Out[o] = qnan;
      if (BN) a.Ypre[o] = qnan;
    }
    return;
  }
  const GcnSplit sp = gcn_split(Rs, wv, Kh);
  const int w = sp.tile;                                           // row tile of this wave
  const int row = 32 * w + c;
  const bool rv = row < Rs;
  f16v acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  if (32 * w < Rs) {                                              // wave-uniform: the tile has rows
    const float *wrow = a.W + (size_t)(c0 + c) * K + h * Kh;
    const int plen = Kh / sp.nparts, k_begin = sp.part * plen;
    auto pb = [&](int k) { return *reinterpret_cast<const f4 *>(wrow + k); };
    if constexpr (AMODE == 0) {
      const float *arow = a.A + (size_t)(r0 + (rv ? row : 0)) * a.lda + h * Kh;
      gcn_kloop(k_begin, k_begin + plen, rv, [&](int k) { return *reinterpret_cast<const f4 *>(arow + k); }, pb, acc);
    } else {
      const long long er = r0 + (rv ? row : 0);
      const int dn = a.t.dn, de = a.t.de;
      // element kg of the virtual concatenated row is x[o0 + kg], x[o1 + kg] or x[o2 + kg] (offsets in floats from a.t.x, the
      // edge features addressed relative to it; segment boundaries are multiples of 32: a block never straddles one).  The
      // offset is combined with masks, not selects: a three-way select of loop-invariant pointers became a lookup table in
      // SCRATCH (one dependent scratch load in front of every global load, 104 B per lane).
      const long long o0 = (long long)a.t.dst[er] * dn;
      const long long o1 = (long long)(a.t.e - a.t.x) + er * de - dn;
      const long long o2 = (long long)a.t.src[er] * dn - dn - de;
      gcn_kloop(k_begin, k_begin + plen, rv, [&](int k) {
        const int kg = h * Kh + k;
        const long long m0 = -(long long)(kg < dn), m2 = -(long long)(kg >= dn + de), m1 = ~(m0 | m2);
        return *reinterpret_cast<const f4 *>(a.t.x + (((o0 & m0) | (o1 & m1) | (o2 & m2)) + kg));
      }, pb, acc);
    }
  }
  gcn_join(part, sp, Rs, wv, lane, acc);
  const bool owner = sp.part == 0 && w < 4;                        // this wave holds the finished tile `w`
  const int col = c0 + c;
  const float bias = a.bias ? a.bias[col] : 0.f;
  float v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = acc[q] + bias;
  if constexpr (BN) {
    // batch statistics of the scan's rows, two passes over the accumulators (mean, then centred squares)
    float s1 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (owner && 32 * w + acc_row(q, h) < Rs) s1 += v[q];
    s1 += __shfl_xor(s1, 32);
    if (h == 0) red[0][wv][c] = s1;
    __syncthreads();
    const float inv_n = 1.f / (float)Rs;
    const float mean = (((red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c])) +
                        ((red[0][4][c] + red[0][5][c]) + (red[0][6][c] + red[0][7][c]))) * inv_n;
    float s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (owner && 32 * w + acc_row(q, h) < Rs) { const float d = v[q] - mean; s2 = fmaf(d, d, s2); }
    s2 += __shfl_xor(s2, 32);
    if (h == 0) red[1][wv][c] = s2;
    __syncthreads();
    const float var = (((red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c])) +
                       ((red[1][4][c] + red[1][5][c]) + (red[1][6][c] + red[1][7][c]))) * inv_n;   // biased, like F.batch_norm
    const float rstd = 1.f / sqrtf(var + a.eps);
    if (wv == 0 && h == 0) { a.mean[(size_t)s * a.N + col] = mean; a.rstd[(size_t)s * a.N + col] = rstd; }
    const float g = a.gamma[col], b = a.beta[col];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int ri = 32 * w + acc_row(q, h);
      if (owner && ri < Rs) {
        const size_t o = (size_t)(r0 + ri) * a.N + col;
        a.Ypre[o] = v[q];
        float y = fmaf((v[q] - mean) * rstd, g, b);
        if (a.relu) y = fmaxf(y, 0.f);
        a.Out[o] = y;
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int ri = 32 * w + acc_row(q, h);
      if (owner && ri < Rs) a.Out[(size_t)(r0 + ri) * a.N + col] = a.relu ? fmaxf(v[q], 0.f) : v[q];
    }
  }
}

struct GcnGradWArgs {
  // gradient of the block's result: GMODE 0 G (R, N); GMODE 1 the adjoint of split + aggregate, read in place:
  // column n < dh: gagg[dst[r], n]; dh <= n < dh + de: gedge[r, n - dh]; else gagg[dst[r], n - dh - de]
  const float *G, *gagg, *gedge;
  int dh, dE;
  const float *Ypre, *mean, *rstd, *gamma, *beta;     // BN: pre-BN values and the forward's statistics; !BN && relu: Ypre = result
  int relu;
  const int64_t *ptr;
  const float *A;                                     // AMODE 0 input rows (R, lda)
  int lda;
  GcnTriplet t;                                       // AMODE 1 (also the dst of GMODE 1)
  float *Gz;                                          // (R, N): gradient at the Linear's output
  float *dW, *dbias, *dgamma, *dbeta;                 // += (zero on entry)
  int K, N;
};

// Backward of a block up to the weights, two launches.
//
// (1) gcn_bn_bwd_kernel, grid (N / 32, S), 256 threads: ReLU mask + BatchNorm backward of one 32-column tile of ONE scan (its
//     sums are local to the workgroup) -> Gz (R, N), which the input gradient needs anyway; dbias / dgamma / dbeta += (3 N
//     atomics per scan).  Lane (c, h) of wave w owns column c0 + c and rows 32 w + 2 i + h, i < 16.
// (2) gcn_wgrad_kernel, grid (N / 32, K / 32), 256 threads: dW tile += Gz^T A over ALL R rows of the batch — the scans only
//     matter to BatchNorm, the weight gradient is one (N, R) x (R, K) product — each wave a quarter of the rows, partial tiles
//     summed through LDS, ONE plain read-modify-write of the tile.  The first form of this kernel kept the per-scan structure
//     and added every scan's 32 x 32 tile with fp32 atomics: S N K device-scope atomics (block 2 of a layer at 8 scans: 5.2 M)
//     at ~90 G/s were 50-80 us of a 230 us backward, growing linearly with the scan count.
template <int GMODE, bool BN>
__global__ __launch_bounds__(256) void gcn_bn_bwd_kernel(const GcnGradWArgs a) {
  __shared__ int s_dst[kGcnRows];
  __shared__ float red[3][4][32];
  const int lane = pn2_lane(), w = threadIdx.x >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int s = blockIdx.y, c0 = blockIdx.x * 32;
  const long long r0 = a.ptr[s];
  const int Rs = (int)(a.ptr[s + 1] - r0);
  const int N = a.N, col = c0 + c;
  if (Rs <= 0) return;
  if (Rs > kGcnRows) {                                             // see gcn_linear_kernel: NaN, not a silent partial result
    for (int i = threadIdx.x; i < Rs * 32; i += blockDim.x) a.Gz[(size_t)(r0 + (i >> 5)) * N + c0 + (i & 31)] = __builtin_nanf("");
    return;
  }
  if constexpr (GMODE == 1) {
    for (int i = threadIdx.x; i < kGcnRows; i += 256) s_dst[i] = i < Rs ? (int)a.t.dst[r0 + i] : 0;
    __syncthreads();
  }
  float mean = 0.f, rstd = 1.f, gam = 1.f, bet = 0.f;
  if constexpr (BN) {
    mean = a.mean[(size_t)s * N + col]; rstd = a.rstd[(size_t)s * N + col];
    gam = a.gamma[col]; bet = a.beta[col];
  }
  // every load unconditional (row clamped, value zeroed afterwards): all 32 are in flight together
  float gv[16], yv[16];
  int seg = 0, gcol = col;
  if constexpr (GMODE == 1) {
    seg = c0 < a.dh ? 0 : (c0 < a.dh + a.dE ? 1 : 2);                // (wave-uniform: dh, dE are multiples of 32)
    gcol = seg == 0 ? col : (seg == 1 ? col - a.dh : col - a.dh - a.dE);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = 32 * w + 2 * i + h, rc = r < Rs ? r : Rs - 1;
    if constexpr (GMODE == 0) gv[i] = a.G[(size_t)(r0 + rc) * N + col];
    else gv[i] = seg == 1 ? a.gedge[(size_t)(r0 + rc) * a.dE + gcol] : a.gagg[(size_t)s_dst[rc] * a.dh + gcol];
    yv[i] = (BN || a.relu) ? a.Ypre[(size_t)(r0 + rc) * N + col] : 1.f;
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const bool ok = 32 * w + 2 * i + h < Rs;
    float g = ok ? gv[i] : 0.f;
    if constexpr (BN) {
      const float x = ok ? (yv[i] - mean) * rstd : 0.f;
      if (a.relu && !(fmaf(x, gam, bet) > 0.f)) g = 0.f;
      yv[i] = x;
      s2 = fmaf(g, x, s2);
    } else if (a.relu) {
      if (!(yv[i] > 0.f)) g = 0.f;
    }
    gv[i] = g;
    s1 += g;
  }
  s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 32);
  if (h == 0) { red[0][w][c] = s1; red[1][w][c] = s2; }
  __syncthreads();
  s1 = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);       // sum of the masked gradient over the scan's rows
  s2 = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);       // ... times the normalised value
  float sb = s1;                                                            // sum of gz (the Linear's bias gradient)
  if constexpr (BN) {
    const float k1 = s1 / (float)Rs, k2 = s2 / (float)Rs, sc = gam * rstd;
    sb = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      gv[i] = 32 * w + 2 * i + h < Rs ? sc * (gv[i] - k1 - yv[i] * k2) : 0.f;
      sb += gv[i];
    }
    sb += __shfl_xor(sb, 32);
    if (h == 0) red[2][w][c] = sb;
    __syncthreads();
    sb = (red[2][0][c] + red[2][1][c]) + (red[2][2][c] + red[2][3][c]);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = 32 * w + 2 * i + h;
    if (r < Rs) a.Gz[(size_t)(r0 + r) * N + col] = gv[i];
  }
  if (w == 0 && h == 0) {
    if (a.dbias) atomicAdd(a.dbias + col, sb);
    if constexpr (BN) { atomicAdd(a.dgamma + col, s2); atomicAdd(a.dbeta + col, s1); }
  }
}

struct GcnWgradArgs {
  const float *Gz;                // (R, N)
  const float *A;                 // AMODE 0: (R, lda)
  int lda;
  GcnTriplet t;                   // AMODE 1: the virtual cat[x[dst], e, x[src]]
  float *dW;                      // (N, K) +=
  long long R;
  int N, K;
};

// dW[n0 + i, k0 + j] += sum_r Gz[r, n0 + i] A[r, k0 + j].  One matrix-core step consumes two rows (lane (c, h): Gz[r + h, n0 + c]
// and A[r + h, k0 + c], one coalesced 128-byte line per half wave); a wave walks its quarter of the rows in chunks of 16 with
// a ring of four chunks in flight (AMODE 1: the gather's row numbers one chunk further ahead).
template <int AMODE>
__global__ __launch_bounds__(256) void gcn_wgrad_kernel(const GcnWgradArgs a) {
  __shared__ float part[3][16][64];
  const int lane = pn2_lane(), w = threadIdx.x >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int N = a.N, K = a.K;
  const long long R = a.R;
  const long long quarter = ((R + 3) / 4 + 15) / 16 * 16;          // whole chunks
  const long long rb = w * quarter < R ? w * quarter : R;
  const long long re = rb + quarter < R ? rb + quarter : R;
  const int nchunk = (int)((re - rb + 15) >> 4);
  float dwv[16];
  if (w == 0) {                                                    // the tile's old value, requested before the walk
#pragma unroll
    for (int q = 0; q < 16; ++q) dwv[q] = a.dW[(size_t)(n0 + acc_row(q, h)) * K + k0 + c];
  }
  f16v acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  int seg = 1, kcol = k0 + c;                                      // AMODE 1: which part of the concatenation the tile lies in
  const int64_t *idx = nullptr;
  if constexpr (AMODE == 1) {
    seg = k0 < a.t.dn ? 0 : (k0 < a.t.dn + a.t.de ? 1 : 2);
    kcol = seg == 0 ? k0 + c : (seg == 1 ? k0 - a.t.dn + c : k0 - a.t.dn - a.t.de + c);
    idx = seg == 0 ? a.t.dst : a.t.src;
  }
  float gv[4][8], av[4][8];
  int iv[2][8];
  auto row_of = [&](int ch, int i) {                               // clamped: a real row of this wave's range
    const long long r = rb + 16 * ch + 2 * i + h;
    return r < re ? r : re - 1;
  };
  auto load_idx = [&](int u, int ch) {
    if constexpr (AMODE == 1) {
      if (seg != 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) iv[u][i] = (int)idx[row_of(ch, i)];
      }
    }
  };
  auto load_val = [&](int u, int ui, int ch) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long rc = row_of(ch, i);
      gv[u][i] = a.Gz[(size_t)rc * N + n0 + c];
      if constexpr (AMODE == 0) av[u][i] = a.A[(size_t)rc * a.lda + kcol];
      else av[u][i] = seg == 1 ? a.t.e[(size_t)rc * a.t.de + kcol] : a.t.x[(size_t)iv[ui][i] * a.t.dn + kcol];
    }
  };
  auto compute = [&](int u, int ch) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool ok = rb + 16 * ch + 2 * i + h < re;
      acc = mfma2(ok ? gv[u][i] : 0.f, av[u][i], acc);
    }
  };
  if (nchunk > 0) {
    // prologue: row numbers of chunks 0 .. 3, values of chunks 0 .. 2
    load_idx(0, 0);
    if (1 < nchunk) load_idx(1, 1);
    load_val(0, 0, 0);
    if (1 < nchunk) load_val(1, 1, 1);
    if (2 < nchunk) load_idx(0, 2);
    if (3 < nchunk) load_idx(1, 3);
    if (2 < nchunk) load_val(2, 0, 2);
    for (int ch = 0; ch < nchunk; ch += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // values of chunk ch + u + 3 (their row numbers arrived one step ago), row numbers of chunk ch + u + 4
        if (ch + u + 3 < nchunk) load_val((u + 3) & 3, (u + 1) & 1, ch + u + 3);
        if (ch + u + 4 < nchunk) load_idx(u & 1, ch + u + 4);
        if (ch + u < nchunk) compute(u, ch + u);
      }
    }
  }
  if (w != 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) part[w - 1][q][lane] = acc[q];
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float v = (acc[q] + part[0][q][lane]) + (part[1][q][lane] + part[2][q][lane]);
      a.dW[(size_t)(n0 + acc_row(q, h)) * K + k0 + c] = dwv[q] + v;
    }
  }
}

struct GcnGradXArgs {
  const float *Gz, *W;            // (R, N), (N, K)
  const int64_t *ptr;
  float *Gin;                     // OMODE 0: (R, K)
  float *gx, *ge;                 // OMODE 1: gx (nodes, dn) += (zero on entry), ge (R, de) =
  GcnTriplet t;
  int N, K;
};

// grid (K / 32, S), 256 threads: wave = a row tile of the scan (and a part of the reduction range, gcn_split), input
// columns [32 bx, +32).  Same four-deep ring as gcn_kloop; the W operand is one dword per reduction step (its rows are K apart).
template <int OMODE>
__global__ __launch_bounds__(512) void gcn_linear_grad_x_kernel(const GcnGradXArgs a) {
  __shared__ float part[kGcnWaves][16][64];
  const int lane = pn2_lane(), wv = threadIdx.x >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int s = blockIdx.y, k0 = blockIdx.x * 32;
  const long long r0 = a.ptr[s];
  const int Rs = (int)(a.ptr[s + 1] - r0);
  const int N = a.N, K = a.K, Nh = N >> 1;
  const GcnSplit sp = gcn_split(Rs, wv, Nh);
  const int w = sp.tile;
  const int row = 32 * w + c;
  const bool rv = row < Rs;
  f16v acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  if (32 * w < Rs) {
    const float *arow = a.Gz + (size_t)(r0 + (rv ? row : 0)) * N + h * Nh;
    const float *wcol = a.W + (size_t)(h * Nh) * K + k0 + c;
    const int plen = Nh / sp.nparts, n_begin = sp.part * plen, nblk = plen >> 4;
    f4 ab[4][4];
    float bb[4][16];
    auto load = [&](int u, int blk) {
      const int n = n_begin + 16 * blk;
#pragma unroll
      for (int j = 0; j < 4; ++j) ab[u][j] = *reinterpret_cast<const f4 *>(arow + n + 4 * j);
#pragma unroll
      for (int j = 0; j < 16; ++j) bb[u][j] = wcol[(size_t)(n + j) * K];
    };
    auto compute = [&](int u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f4 av = ab[u][j];
        if (!rv) av = f4{0.f, 0.f, 0.f, 0.f};
        acc = mfma2(av.x, bb[u][4 * j], acc); acc = mfma2(av.y, bb[u][4 * j + 1], acc);
        acc = mfma2(av.z, bb[u][4 * j + 2], acc); acc = mfma2(av.w, bb[u][4 * j + 3], acc);
      }
    };
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (u < nblk) load(u, u);
    for (int blk = 0; blk < nblk; blk += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (blk + u + 3 < nblk) load((u + 3) & 3, blk + u + 3);
        if (blk + u < nblk) compute(u);
      }
    }
  }
  gcn_join(part, sp, Rs, wv, lane, acc);
  if (sp.part != 0 || 32 * w >= Rs) return;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int ri = 32 * w + acc_row(q, h);
    if (ri >= Rs) continue;
    const long long er = r0 + ri;
    if constexpr (OMODE == 0) {
      a.Gin[(size_t)er * K + k0 + c] = acc[q];
    } else {
      const int dn = a.t.dn, de = a.t.de;
      if (k0 < dn) atomicAdd(a.gx + (size_t)a.t.dst[er] * dn + k0 + c, acc[q]);
      else if (k0 < dn + de) a.ge[(size_t)er * de + (k0 - dn) + c] = acc[q];
      else atomicAdd(a.gx + (size_t)a.t.src[er] * dn + (k0 - dn - de) + c, acc[q]);
    }
  }
}

// split of nn1's result: new edge feature = columns [dh, dh + de) (+ ReLU between layers), one launch
__global__ __launch_bounds__(256) void gcn_edge_slice_kernel(long long total4, int de4, int ld4, int off4, int relu,
                                                            const f4 *__restrict__ hsrc, f4 *__restrict__ out) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const long long r = i / de4;
    const int j = (int)(i - r * de4);
    f4 v = hsrc[r * ld4 + off4 + j];
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    out[i] = v;
  }
}

bool gcn_dims_ok(int K, int N) { return K >= 32 && N >= 32 && K % 32 == 0 && N % 32 == 0; }
}  // namespace

extern "C" int pn2_gcn_fused_supported(int dn, int de, int dh, int max_rows_per_scan) {
  return (dn >= 32 && de >= 32 && dh >= 32 && dn % 32 == 0 && de % 32 == 0 && dh % 32 == 0 && max_rows_per_scan <= kGcnRows) ? 1 : 0;
}

// Out (R, N) = [ReLU] [BN_scan] (A W^T + bias).  A: rows (R, lda) when `x` is null, else the virtual concatenation
// cat[x[dst], e, x[src]] (K = 2 dn + de).  gamma null: no BatchNorm (Ypre / mean / rstd unused).
extern "C" int pn2_gcn_linear(long long R, int S, int K, int N, const float *A, int lda, const float *x, const float *e,
                              const long long *dst, const long long *src, int dn, int de, const float *W,
                              const float *bias, const long long *ptr, const float *gamma, const float *beta, float eps,
                              int relu, float *Ypre, float *Out, float *mean, float *rstd, void *stream) {
  if (R < 0 || S < 0 || !gcn_dims_ok(K, N)) return PN2_EINVAL;
  if (R == 0 || S == 0) return PN2_OK;
  if (!W || !ptr || !Out) return PN2_ENULL;
  const bool trip = x != nullptr;
  if (trip ? (!e || !dst || !src || K != 2 * dn + de || dn % 32 || de % 32) : (!A || lda < K || (lda & 3))) return PN2_EINVAL;
  if (gamma && (!beta || !Ypre || !mean || !rstd)) return PN2_ENULL;
  GcnFwdArgs a{A, lda, {x, e, (const int64_t *)dst, (const int64_t *)src, dn, de}, W, bias, (const int64_t *)ptr, gamma, beta,
               eps, Ypre, Out, mean, rstd, K, N, relu ? 1 : 0};
  const dim3 grid((unsigned)(N / 32), (unsigned)S), block(512);
  hipStream_t s = (hipStream_t)stream;
  if (trip) {
    if (gamma) hipLaunchKernelGGL((gcn_linear_kernel<1, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gcn_linear_kernel<1, false>), grid, block, 0, s, a);
  } else {
    if (gamma) hipLaunchKernelGGL((gcn_linear_kernel<0, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gcn_linear_kernel<0, false>), grid, block, 0, s, a);
  }
  return pn2_check_launch();
}

// Backward of one block up to the weights.  G (R, N) gradient of the block's result — or, with gagg != null, the adjoint of
// split + aggregate read in place (N = 2 dh + dE; dst required).  gamma null: no BatchNorm (relu: `Ypre` = the block's result).
// Writes Gz (R, N); dW (N, K), dbias (N), dgamma (N), dbeta (N) += (the caller zeroes them once per step).
extern "C" int pn2_gcn_linear_grad_w(long long R, int S, int K, int N, const float *G, const float *gagg, const float *gedge,
                                     int dh, int dE, const float *Ypre, const float *mean, const float *rstd,
                                     const float *gamma, const float *beta, int relu, const long long *ptr, const float *A,
                                     int lda, const float *x, const float *e, const long long *dst, const long long *src,
                                     int dn, int de, float *Gz, float *dW, float *dbias, float *dgamma, float *dbeta,
                                     void *stream) {
  if (R < 0 || S < 0 || !gcn_dims_ok(K, N)) return PN2_EINVAL;
  if (R == 0 || S == 0) return PN2_OK;
  if (!ptr || !Gz || !dW) return PN2_ENULL;
  const bool trip = x != nullptr, adj = gagg != nullptr;
  if (trip ? (!e || !dst || !src || K != 2 * dn + de || dn % 32 || de % 32) : (!A || lda < K)) return PN2_EINVAL;
  if (adj ? (!gedge || !dst || N != 2 * dh + dE || dh % 32 || dE % 32) : !G) return PN2_EINVAL;
  if (gamma && (!beta || !Ypre || !mean || !rstd || !dgamma || !dbeta)) return PN2_ENULL;
  if (!gamma && relu && !Ypre) return PN2_ENULL;
  GcnGradWArgs a{G, gagg, gedge, dh, dE, Ypre, mean, rstd, gamma, beta, relu ? 1 : 0, (const int64_t *)ptr, A, lda,
                 {x, e, (const int64_t *)dst, (const int64_t *)src, dn, de}, Gz, dW, dbias, dgamma, dbeta, K, N};
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid1((unsigned)(N / 32), (unsigned)S), block(256);
  if (adj) {
    if (gamma) hipLaunchKernelGGL((gcn_bn_bwd_kernel<1, true>), grid1, block, 0, s, a);
    else hipLaunchKernelGGL((gcn_bn_bwd_kernel<1, false>), grid1, block, 0, s, a);
  } else {
    if (gamma) hipLaunchKernelGGL((gcn_bn_bwd_kernel<0, true>), grid1, block, 0, s, a);
    else hipLaunchKernelGGL((gcn_bn_bwd_kernel<0, false>), grid1, block, 0, s, a);
  }
  GcnWgradArgs b{Gz, A, lda, {x, e, (const int64_t *)dst, (const int64_t *)src, dn, de}, dW, R, N, K};
  const dim3 grid2((unsigned)(N / 32), (unsigned)(K / 32));
  if (trip) hipLaunchKernelGGL((gcn_wgrad_kernel<1>), grid2, block, 0, s, b);
  else hipLaunchKernelGGL((gcn_wgrad_kernel<0>), grid2, block, 0, s, b);
  return pn2_check_launch();
}

// Input gradient Gz W: rows Gin (R, K), or (x != null: only its dims are used through dn / de) scattered through the adjoint
// of the triplet gather: gx (nodes, dn) += columns [0, dn) at dst and [dn + de, K) at src (zero on entry), ge (R, de) = the middle.
extern "C" int pn2_gcn_linear_grad_x(long long R, int S, int K, int N, const float *Gz, const float *W, const long long *ptr,
                                     float *Gin, float *gx, float *ge, const long long *dst, const long long *src, int dn,
                                     int de, void *stream) {
  if (R < 0 || S < 0 || !gcn_dims_ok(K, N)) return PN2_EINVAL;
  if (R == 0 || S == 0) return PN2_OK;
  if (!Gz || !W || !ptr) return PN2_ENULL;
  const bool trip = gx != nullptr;
  if (trip ? (!ge || !dst || !src || K != 2 * dn + de || dn % 32 || de % 32) : !Gin) return PN2_EINVAL;
  GcnGradXArgs a{Gz, W, (const int64_t *)ptr, Gin, gx, ge, {nullptr, nullptr, (const int64_t *)dst, (const int64_t *)src, dn, de},
                 N, K};
  const dim3 grid((unsigned)(K / 32), (unsigned)S), block(512);
  hipStream_t s = (hipStream_t)stream;
  if (trip) hipLaunchKernelGGL((gcn_linear_grad_x_kernel<1>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((gcn_linear_grad_x_kernel<0>), grid, block, 0, s, a);
  return pn2_check_launch();
}

// out (R, de) = [ReLU] h[:, off : off + de] of rows (R, ld): the new edge feature of a TripletGCN layer (:51)
extern "C" int pn2_gcn_edge_slice(long long R, int ld, int off, int de, int relu, const float *hrows, float *out, void *stream) {
  if (R < 0 || (ld & 3) || (off & 3) || (de & 3) || de <= 0 || off + de > ld) return PN2_EINVAL;
  if (R == 0) return PN2_OK;
  if (!hrows || !out) return PN2_ENULL;
  const long long total4 = R * (de / 4);
  long long blocks = (total4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gcn_edge_slice_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, total4, de / 4, ld / 4,
                     off / 4, relu ? 1 : 0, (const f4 *)hrows, (f4 *)out);
  return pn2_check_launch();
}

// ---- one layer per C call: the sequence _FusedTripletLayer issued block by block from python ------------------------------
namespace {
inline size_t gcn_pad(size_t floats) { return (floats + 63) / 64 * 64; }     // 256-byte pieces of the backward workspace
static_assert(sizeof(pn2_gcn_layer) == 48 + 56 * sizeof(void *), "pn2_gcn_layer: two i64, five int, three float, 56 pointers (the python binding mirrors it)");
}  // namespace

extern "C" size_t pn2_gcn_layer_backward_workspace_bytes(long long nodes, long long edges, int dn, int de, int dh) {
  if (nodes < 0 || edges < 0 || dn <= 0 || de <= 0 || dh <= 0) return 0;
  const size_t n = (size_t)nodes, e = (size_t)edges;
  // node side: gz4 (dn) | g_t (dh) | gz3 (dh) | g_agg (dh);  edge side: gz2 (2 dh + de) | g_h1 (dh) | gz1 (dh)
  return 4 * (gcn_pad(n * dn) + 3 * gcn_pad(n * dh) + gcn_pad(e * (2 * (size_t)dh + de)) + 2 * gcn_pad(e * dh));
}

extern "C" int pn2_gcn_layer_forward(const pn2_gcn_layer *L, void *stream) {
  if (!L) return PN2_ENULL;
  if (L->nodes < 0 || L->edges < 0 || L->S < 0) return PN2_EINVAL;
  if (L->nodes == 0 || L->edges == 0 || L->S == 0) return PN2_OK;
  const int dn = L->dn, de = L->de, dh = L->dh, wide = 2 * dh + de;
  if (!pn2_gcn_fused_supported(dn, de, dh, 0)) return PN2_EINVAL;
  if (!L->order || !L->rowptr || !L->agg || !L->e_out || !L->h1 || !L->h2 || !L->t) return PN2_ENULL;
  int rc = pn2_gcn_linear(L->edges, L->S, 2 * dn + de, dh, nullptr, 0, L->x, L->e, L->dst, L->src, dn, de, L->W1, L->b1, L->edge_ptr,
                          L->g1, L->be1, L->eps1, 1, L->h1p, L->h1, L->m1, L->r1, stream);
  if (rc) return rc;
  rc = pn2_gcn_linear(L->edges, L->S, dh, wide, L->h1, dh, nullptr, nullptr, nullptr, nullptr, 0, 0, L->W2, L->b2, L->edge_ptr, L->g2,
                      L->be2, L->eps2, 1, L->h2p, L->h2, L->m2, L->r2, stream);
  if (rc) return rc;
  rc = pn2_segment_sum2_rows(L->edges, dh, L->nodes, wide, 0, dh + de, L->h2, (const int64_t *)L->order, (const int64_t *)L->rowptr,
                             L->agg, stream);
  if (rc) return rc;
  rc = pn2_gcn_edge_slice(L->edges, wide, dh, de, L->relu_out, L->h2, L->e_out, stream);
  if (rc) return rc;
  rc = pn2_gcn_linear(L->nodes, L->S, dh, dh, L->agg, dh, nullptr, nullptr, nullptr, nullptr, 0, 0, L->W3, L->b3, L->node_ptr, L->g3,
                      L->be3, L->eps3, 1, L->tp, L->t, L->m3, L->r3, stream);
  if (rc) return rc;
  return pn2_gcn_linear(L->nodes, L->S, dh, dn, L->t, dh, nullptr, nullptr, nullptr, nullptr, 0, 0, L->W4, L->b4, L->node_ptr, nullptr,
                        nullptr, 0.f, L->relu_out, nullptr, L->out, nullptr, nullptr, stream);
}

extern "C" int pn2_gcn_layer_backward(const pn2_gcn_layer *L, void *stream) {
  if (!L) return PN2_ENULL;
  if (L->nodes < 0 || L->edges < 0 || L->S < 0) return PN2_EINVAL;
  if (L->nodes == 0 || L->edges == 0 || L->S == 0) return PN2_OK;
  const int dn = L->dn, de = L->de, dh = L->dh, wide = 2 * dh + de;
  if (!pn2_gcn_fused_supported(dn, de, dh, 0)) return PN2_EINVAL;
  if (!L->work || !L->g_out || !L->g_e || !L->gx || !L->ge) return PN2_ENULL;
  const size_t n = (size_t)L->nodes, e = (size_t)L->edges;
  float *gz4 = (float *)L->work, *g_t = gz4 + gcn_pad(n * dn), *gz3 = g_t + gcn_pad(n * dh), *g_agg = gz3 + gcn_pad(n * dh);
  float *gz2 = g_agg + gcn_pad(n * dh), *g_h1 = gz2 + gcn_pad(e * wide), *gz1 = g_h1 + gcn_pad(e * dh);
  // nn2[3] (no BatchNorm; its ReLU is the model's between-layer one), nn2[0..2]
  int rc = pn2_gcn_linear_grad_w(L->nodes, L->S, dh, dn, L->g_out, nullptr, nullptr, 0, 0, L->out, nullptr, nullptr, nullptr, nullptr,
                                 L->relu_out, L->node_ptr, L->t, dh, nullptr, nullptr, nullptr, nullptr, 0, 0, gz4, L->dW4, L->db4,
                                 nullptr, nullptr, stream);
  if (rc) return rc;
  rc = pn2_gcn_linear_grad_x(L->nodes, L->S, dh, dn, gz4, L->W4, L->node_ptr, g_t, nullptr, nullptr, nullptr, nullptr, 0, 0, stream);
  if (rc) return rc;
  rc = pn2_gcn_linear_grad_w(L->nodes, L->S, dh, dh, g_t, nullptr, nullptr, 0, 0, L->tp, L->m3, L->r3, L->g3, L->be3, 1, L->node_ptr,
                             L->agg, dh, nullptr, nullptr, nullptr, nullptr, 0, 0, gz3, L->dW3, L->db3, L->dg3, L->dbe3, stream);
  if (rc) return rc;
  rc = pn2_gcn_linear_grad_x(L->nodes, L->S, dh, dh, gz3, L->W3, L->node_ptr, g_agg, nullptr, nullptr, nullptr, nullptr, 0, 0, stream);
  if (rc) return rc;
  // nn1[3..5]: its gradient is the adjoint of split + aggregate, [g_agg[dst] | g_e | g_agg[dst]], read in place.  A ReLU on
  // e_out (relu_out) needs no mask of its own: e_out is a slice of h2 = ReLU(..), and this block's mask zeroes the same entries
  rc = pn2_gcn_linear_grad_w(L->edges, L->S, dh, wide, nullptr, g_agg, L->g_e, dh, de, L->h2p, L->m2, L->r2, L->g2, L->be2, 1,
                             L->edge_ptr, L->h1, dh, nullptr, nullptr, L->dst, nullptr, 0, 0, gz2, L->dW2, L->db2, L->dg2, L->dbe2,
                             stream);
  if (rc) return rc;
  rc = pn2_gcn_linear_grad_x(L->edges, L->S, dh, wide, gz2, L->W2, L->edge_ptr, g_h1, nullptr, nullptr, nullptr, nullptr, 0, 0, stream);
  if (rc) return rc;
  // nn1[0..2] on the virtual concatenation; its input gradient scattered back through the gather
  rc = pn2_gcn_linear_grad_w(L->edges, L->S, 2 * dn + de, dh, g_h1, nullptr, nullptr, 0, 0, L->h1p, L->m1, L->r1, L->g1, L->be1, 1,
                             L->edge_ptr, nullptr, 0, L->x, L->e, L->dst, L->src, dn, de, gz1, L->dW1, L->db1, L->dg1, L->dbe1, stream);
  if (rc) return rc;
  return pn2_gcn_linear_grad_x(L->edges, L->S, 2 * dn + de, dh, gz1, L->W1, L->edge_ptr, nullptr, L->gx, L->ge, L->dst, L->src, dn, de,
                               stream);
}
