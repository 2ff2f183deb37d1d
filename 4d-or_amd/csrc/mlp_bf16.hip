// mlp_bf16.hip — bf16-MFMA variants of the shared per-point MLP kernels (mixed precision).
//
// The reference trains under 16-bit AMP (scene_graph_prediction/main.py:64 `precision=16`): the 1x1 convolutions of
// the SA/FP shared MLPs (OPS/pointnet2_modules.py:9-19) run in half precision with fp32 master weights, BatchNorm
// keeps fp32 statistics, and the grouping ops stay fp32 (`custom_fwd(cast_inputs=torch.float32)`,
// OPS/pointnet2_utils.py:198).  The MI355X counterpart: activations between the layers of a stack are STORED as bf16
// (the raw pre-BatchNorm outputs y_l and the gradients dL/dz_l — the only large tensors of the step), the matrix
// products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, BatchNorm statistics / constants stay fp32 / fp64,
// weights and weight gradients stay fp32, and the geometry kernels are untouched.
//
// With bf16 operands the arithmetic intensity of these tall-skinny GEMMs (K, N <= 320) is ~50 FLOP/B against a ridge
// of ~310 FLOP/B (2.5 PFLOP/s / 8 TB/s): every kernel here is HBM-bound, the MFMA pipe is ~15 % busy.  They are
// therefore built for bytes in flight, not for MFMA issue: 16-byte loads of bf16 rows, the weight matrix resident in
// LDS for the lifetime of a persistent workgroup, the next A chunk prefetched into registers behind the current
// chunk's MFMAs, two workgroups per CU.
//
//   pn2_mlp_gemm_bf16  : Y[M][N] = pro(X)[M][K] * W[N][K]^T   (same prologue / epilogue algebra as csrc/mlp_gemm.hip)
//   pn2_mlp_wgrad_bf16 : dW[N][K] += gy^T * act                (operands transposed into LDS as packed row pairs)
//   pn2_*_bf16 helpers : BN+ReLU(+max) and their backward preparations on bf16 y
#include "pn2_common.h"
#include "mlp_common.h"

namespace {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }
__device__ __forceinline__ unsigned short bf_bits(float f) { return __builtin_bit_cast(unsigned short, (bf16)f); }
__device__ __forceinline__ unsigned bf_pack(float lo, float hi) {
  return (unsigned)bf_bits(lo) | ((unsigned)bf_bits(hi) << 16);
}
__device__ __forceinline__ float bf_round(float f) { return (float)(bf16)f; }

__device__ __forceinline__ u32x4 bload128(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ unsigned short bload16(rsrc_t r, int voff, int soff) {
  return (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
}
__device__ __forceinline__ void bstore16(unsigned short v, rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b16(v, r, voff, soff, 0);
}
__device__ __forceinline__ void bstore128(u32x4 v, rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}

constexpr int TM = 128;        // rows per tile: 4 waves x 32
constexpr int KC = 64;         // K chunk (4 MFMA k-steps of 16)
constexpr int AP = KC + 8;     // LDS pitch of the A chunk in bf16 elements (16-byte rows, conflict-free b128 reads)

struct GemmBf16Args {
  const void *X;      // [M][ldx]  bf16 (or fp32 when XF32); PRO_GY: g
  const bf16 *X2;     // PRO_GY / PRO_POOLG: y_l [M][ldx]
  const float *p0, *p1, *p2;   // per-K prologue vectors
  const int *arg;     // PRO_POOLG [M/ns][K]
  const float *gP;    // PRO_POOLG [M/ns][K]
  const float *W;     // [N][K] fp32 master weights
  void *Y;            // [M][ldy] bf16 (fp32 when YF32)
  double *stats;      // [2][N]
  const bf16 *Yprev;  // EPI_MASK [M][ldy]
  const float *e_fin; // EPI_MASK: [mean | rstd | scale | shift] x N
  long long M;
  int K, N, ldx, ldy, ns;
  int Nfull;          // stride of the per-column vectors (stats, e_fin): the full output width when N is a column block
  int Kp;             // K rounded up to KC
  int wres;           // weights resident in LDS
  // Batched scans with PER-SCAN statistics (segment table): blockIdx.y = scan s owns rows [seg[s], seg[s+1]) and its own
  // block of every per-channel array — p0/p1/p2 + s*pstride, stats + s*sstride, e_fin + s*estride; W is shared.
  const long long *seg;
  long long seg_max;  // host: rows of the longest segment (grid)
  int nseg, pstride, sstride, estride;
  int vcap;           // the persistent grid a call of ONE scan would get at most (see the virtual workgroups of the kernel)
  // EPI_POOL (the max-pooled last layer without its output, as csrc/mlp_gemm.hip EPI_POOL): per partial group of
  // psz = min(ns, 32) rows and column the maximum of the fp32 accumulators and its row; W arrives with the rows of
  // negative-gamma columns negated, sgn restores the column sums.  Y is not written.
  float *pmax;        // [M/psz][N]
  int *parg;          // [M/psz][N]
  const float *sgn;   // [N]
  // PRO_FIRST (the stack's FIRST layer re-formed from its input rows, as csrc/mlp_gemm.hip PRO_FIRST): X = the rows [M][8]
  // bf16 (K0 <= 8 real columns, zero padded), W0 [K][K0] fp32 the first layer's weight, p0 / p1 its BatchNorm scale / shift;
  // y_0 is never stored — the A tile is relu(bn_0(X W0^T)) formed in fp32 and rounded to bf16 once
  const float *W0;
  int K0;
};

// ---------------------------------------------------------------------------------------------- forward / dgrad
// Workgroup = 4 x CW waves: wave (wr, wc) owns rows wr*32.. of the 128-row tile and the NT 32-column tiles wc*NT.. .
// CW = 2 keeps the accumulators of N > 128 within 128 VGPRs per lane (two workgroups of 8 waves per CU).
template <int NT, int CW, int PRO, int EPI, bool XF32, bool YF32>
__global__ __launch_bounds__(256 * CW, 2) void mlp_gemm_bf16_kernel(const GemmBf16Args a_in) {
  GemmBf16Args a = a_in;
  if (a.seg) {                                      // this workgroup's scan: shift every row / per-channel pointer
    const int sg = blockIdx.y;
    const long long r0 = a.seg[sg];
    a.M = a.seg[sg + 1] - r0;
    a.X = (const char *)a.X + (size_t)r0 * a.ldx * (XF32 ? 4 : 2);
    a.Y = (char *)a.Y + (size_t)r0 * a.ldy * (YF32 ? 4 : 2);
    if constexpr (PRO == PRO_GY || PRO == PRO_POOLG) a.X2 += (size_t)r0 * a.ldx;
    if constexpr (PRO == PRO_POOLG) { a.arg += (size_t)(r0 / a.ns) * a.K; a.gP += (size_t)(r0 / a.ns) * a.K; }
    if constexpr (PRO != PRO_NONE) { a.p0 += (size_t)sg * a.pstride; a.p1 += (size_t)sg * a.pstride; }
    if constexpr (PRO == PRO_GY || PRO == PRO_POOLG) a.p2 += (size_t)sg * a.pstride;
    if constexpr (EPI != EPI_NONE) a.stats += (size_t)sg * a.sstride;
    if constexpr (EPI == EPI_MASK) { a.Yprev += (size_t)r0 * a.ldy; a.e_fin += (size_t)sg * a.estride; }
  }
  constexpr int NTH = 256 * CW;                     // threads
  constexpr int NTT = NT * CW;                      // column tiles of the workgroup
  // LDS: [ W tile | prologue parameters | A chunk, re-used by the ReLU-backward epilogue as the y_{l-1} / output tile ]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Kp = a.Kp;
  const int WP = a.wres ? Kp + 8 : AP;
  bf16 *sW = (bf16 *)smem;                          // [NTT*32][WP]
  float *sP = (float *)(smem + NTT * 32 * WP * 2);  // [3][Kp] prologue parameters, zero padded
  float *sW0 = sP + 3 * Kp;                         // PRO_FIRST: [Kp] first-layer weight rows as bf16x8 (16 bytes each), zero padded
  bf16 *sA = (bf16 *)((unsigned char *)sP + (PRO != PRO_NONE ? 3 * Kp * 4 : 0) + (PRO == PRO_FIRST ? Kp * 16 : 0));   // [TM][AP]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (tid >> 6) & 3, wc = tid >> 8;
  const int K = a.K, N = a.N;

  if constexpr (PRO != PRO_NONE) {
    for (int k = tid; k < Kp; k += NTH) {
      sP[k] = k < K ? a.p0[k] : 0.f;
      sP[Kp + k] = k < K ? a.p1[k] : 0.f;
      sP[2 * Kp + k] = (PRO != PRO_BNRELU && PRO != PRO_FIRST && k < K) ? a.p2[k] : 0.f;
    }
    if constexpr (PRO == PRO_FIRST) {
      // first-layer weight rows as bf16 fragments: row k = eight input columns (K0 real ones), 16 bytes
      for (int k = tid; k < Kp; k += NTH) {
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (k < K && j < a.K0) ? a.W0[(size_t)k * a.K0 + j] : 0.f;
        ((u32x4 *)sW0)[k] = u32x4{bf_pack(w[0], w[1]), bf_pack(w[2], w[3]), bf_pack(w[4], w[5]), bf_pack(w[6], w[7])};
      }
    }
  }
  auto stage_w = [&](int k0, int kw) {              // sW[n][0..kw) = bf16(W[n][k0..k0+kw)), zero outside N x K
    // a wave per weight row, a lane per FOUR consecutive k (one dword-aligned 16-byte load, one 8-byte LDS store), four rows
    // in flight.  (First version: one float per lane and row, one row at a time — every workgroup spent ~30 us staging a
    // resident 128 x 128 weight, which is most of a launch when M is small.)
    typedef float w4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(4))) W4 { w4 v; };
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    constexpr int WAVES = NTH / 64;
    for (int kb = 0; kb < kw; kb += 256) {
      const int k = kb + 4 * lane;                    // column of this lane inside the staged range
      const bool kin = k < kw;
      for (int n0 = tid >> 6; n0 < NTT * 32; n0 += 4 * WAVES) {
        w4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = n0 + u * WAVES;
          v[u] = w4{0.f, 0.f, 0.f, 0.f};
          if (kin && n < NTT * 32 && n < N) {
            const float *wr = a.W + (size_t)n * K + k0 + k;
            if (k0 + k + 3 < K) v[u] = ((const W4 *)wr)->v;
            else {
#pragma unroll
              for (int i = 0; i < 4; ++i) v[u][i] = k0 + k + i < K ? wr[i] : 0.f;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = n0 + u * WAVES;
          if (kin && n < NTT * 32) *(u2 *)&sW[n * WP + k] = u2{bf_pack(v[u][0], v[u][1]), bf_pack(v[u][2], v[u][3])};
        }
      }
    }
  };
  if (a.wres) stage_w(0, Kp);

  // per-column epilogue constants and running column sums (this lane's column of each 32-wide tile)
  float s1[NT], s2[NT], e_mean[NT], e_rstd[NT], e_sc[NT], e_sh[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = (wc * NT + nt) * 32 + (lane & 31);
    if constexpr (EPI == EPI_MASK) {
      const bool ok = c < N;
      e_mean[nt] = ok ? a.e_fin[c] : 0.f;
      e_rstd[nt] = ok ? a.e_fin[a.Nfull + c] : 0.f;
      e_sc[nt] = ok ? a.e_fin[2 * a.Nfull + c] : 0.f;
      e_sh[nt] = ok ? a.e_fin[3 * a.Nfull + c] : 0.f;
    }
  }

  const long long ntiles = (a.M + TM - 1) / TM;
  const int nchunks = Kp / KC;
  // A loader, bf16 input: thread -> (row = tid / (2 CW), 32 / CW consecutive k = J x 16 bytes)
  constexpr int J = 4 / CW;
  const int lr = tid / (2 * CW), lk = (tid % (2 * CW)) * (8 * J);
  const int xs = XF32 ? 4 : 2;

  // Virtual workgroups: the fp32 column sums are grouped per workgroup (tiles w, w + grid, ...) before they meet in fp64 —
  // a scan of a segmented call keeps the grouping its OWN call would have (grid = min(vcap, its tiles)): this workgroup
  // plays the virtual workgroups blockIdx.x, blockIdx.x + gridDim.x, ... of that grid one after the other, each with its
  // own sums and flush, so the scan's statistics — and every bf16 rounding downstream — are those of the single-scan
  // call.  Without a table: one virtual workgroup = this one.
  const int vstride = a.seg ? (int)(ntiles < a.vcap ? ntiles : a.vcap) : (int)gridDim.x;
  for (int vwg = blockIdx.x; vwg < vstride; vwg += gridDim.x) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) s1[nt] = s2[nt] = 0.f;

  for (long long tile = vwg; tile < ntiles; tile += vstride) {
    const long long row0 = tile * TM;
    const long long rows_left = a.M - row0;
    const rsrc_t rX = make_rsrc((const char *)a.X + (size_t)row0 * a.ldx * xs, rows_left * a.ldx * xs);
    rsrc_t rX2 = rX;
    if constexpr (PRO == PRO_GY || PRO == PRO_POOLG)
      rX2 = make_rsrc((const char *)a.X2 + (size_t)row0 * a.ldx * 2, rows_left * a.ldx * 2);

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    u32x4 ra[J], rb[J];                              // prefetched raw 16-byte groups (g / x and y)
    // PRO_POOLG: the (arg-max row, pooled gradient) entries of the next chunk travel with its row loads — entry
    // t = tid + e * NTH of the ngr x KC entries a chunk has (ns >= 32: one per thread); fetched inside patch_pool they were
    // two dependent-latency global loads between two barriers, once per chunk
    constexpr bool PFON = NT * CW < 10;              // (the widest variant has no registers to spare: loads in patch_pool)
    constexpr int PF = 2;
    int pf_arg[PF];
    float pf_g[PF];
    const long long pg0 = PRO == PRO_POOLG ? row0 / a.ns : 0;
    const long long plast = (row0 + TM - 1 < a.M - 1 ? row0 + TM - 1 : a.M - 1);
    const int ngr = PRO == PRO_POOLG ? (int)(plast / a.ns - pg0) + 1 : 0;
    auto issue = [&](int kc) {
      if constexpr (PRO == PRO_FIRST) {
        // the 32 input rows of this wave's row block (pitch 8 bf16 = 16 bytes); lanes 32-63 supply the zero half of the k step
        ra[0] = lane < 32 ? bload128(rX, (wave * 32 + lane) * 16, 0) : u32x4{0u, 0u, 0u, 0u};
      } else if constexpr (!XF32) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int k = kc * KC + lk + 8 * j;
          const int off = k < a.ldx ? (lr * a.ldx + k) * 2 : kOobOffset;
          if constexpr (PRO != PRO_POOLG) ra[j] = bload128(rX, off, 0);
          if constexpr (PRO == PRO_GY || PRO == PRO_POOLG) rb[j] = bload128(rX2, off, 0);
        }
      }
    };
    auto issue_pool = [&](int kc) {
      if constexpr (PRO == PRO_POOLG && PFON) {
#pragma unroll
        for (int e = 0; e < PF; ++e) {
          const int t = tid + e * NTH, gi = t >> 6, k = kc * KC + (t & 63);
          pf_arg[e] = -1;
          pf_g[e] = 0.f;
          if (gi < ngr && k < K) {
            const size_t o = (size_t)(pg0 + gi) * K + k;
            pf_arg[e] = a.arg[o];
            pf_g[e] = a.gP[o];
          }
        }
      }
    };
    auto commit = [&](int kc) {                      // prologue + write the chunk into sA
      if constexpr (XF32) {
        // fp32 rows (first layer of an un-grouped stack, FP modules): flat, coalesced scalar loads; PRO_NONE only
        const int k = tid & 63, kk = kc * KC + k;
        const int off0 = ((tid >> 6) * a.ldx + kk) * 4;
#pragma unroll 8
        for (int i = 0; i < 32 / CW; ++i) {
          const float v = bload(rX, kk < K ? off0 + i * (4 * CW) * a.ldx * 4 : kOobOffset, 0);
          sA[((tid >> 6) + 4 * CW * i) * AP + k] = (bf16)v;
        }
      } else {
        const long long grow = row0 + lr;
        if constexpr (PRO == PRO_FIRST) {
          // y_0 = X W0^T on the matrix pipe (K0 <= 8 input columns zero-padded to ONE 16-wide step: one instruction per
          // 32 x 32 block, bf16 operands like the first layer's own GEMM on this path), BatchNorm + ReLU on the accumulators,
          // rounded to bf16 once into the A chunk.  Wave (row block, column wave wc) takes the 32-column blocks wc, wc + CW, ..
#pragma unroll
          for (int cb = 0; cb < KC / 32; ++cb) {
            if (cb % CW != wc) continue;                // wave-uniform
            const int k = kc * KC + cb * 32 + (lane & 31);
            const u32x4 wf = lane < 32 ? ((const u32x4 *)sW0)[k] : u32x4{0u, 0u, 0u, 0u};
            f32x16 y;
#pragma unroll
            for (int e = 0; e < 16; ++e) y[e] = 0.f;
            y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[0]), __builtin_bit_cast(bf16x8, wf), y, 0, 0, 0);
            const float q0 = sP[k], q1 = sP[Kp + k];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int r = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
              const float v = row0 + r < a.M ? fmaxf(fmaf(y[e], q0, q1), 0.f) : 0.f;
              sA[r * AP + cb * 32 + (lane & 31)] = (bf16)v;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < (PRO == PRO_FIRST ? 0 : J); ++j) {
          const int k = kc * KC + lk + 8 * j;
          float v[8];
          if constexpr (PRO == PRO_NONE) {
            *(u32x4 *)&sA[lr * AP + lk + 8 * j] = ra[j];
            continue;
          }
          const float *q0 = sP + k, *q1 = sP + Kp + k, *q2 = sP + 2 * Kp + k;
          if constexpr (PRO == PRO_BNRELU) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              v[2 * i] = fmaxf(fmaf(bf_lo(ra[j][i]), q0[2 * i], q1[2 * i]), 0.f);
              v[2 * i + 1] = fmaxf(fmaf(bf_hi(ra[j][i]), q0[2 * i + 1], q1[2 * i + 1]), 0.f);
            }
          } else if constexpr (PRO == PRO_GY) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              v[2 * i] = fmaf(q0[2 * i], bf_lo(ra[j][i]), fmaf(q1[2 * i], bf_lo(rb[j][i]), q2[2 * i]));
              v[2 * i + 1] = fmaf(q0[2 * i + 1], bf_hi(ra[j][i]), fmaf(q1[2 * i + 1], bf_hi(rb[j][i]), q2[2 * i + 1]));
            }
          } else {   // PRO_POOLG: dense part c2*y + c3 here; the pooled gradient is patched in afterwards (patch_pool)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float y = (i & 1) ? bf_hi(rb[j][i >> 1]) : bf_lo(rb[j][i >> 1]);
              v[i] = fmaf(q1[i], y, q2[i]);
            }
          }
          // rows past M read as zeros, but their prologue value (relu(shift), c3) is not zero: they must not reach the
          // column sums of the epilogue
          u32x4 w;
#pragma unroll
          for (int i = 0; i < 4; ++i) w[i] = grow < a.M ? bf_pack(v[2 * i], v[2 * i + 1]) : 0u;
          *(u32x4 *)&sA[lr * AP + lk + 8 * j] = w;
        }
      }
    };

    // PRO_POOLG: the gradient of a max-pooled layer is ONE non-zero per (neighbourhood, channel), at the arg-max row:
    // instead of testing every element against the arg-max (two scattered loads per element) the dense tile
    // c2*y + c3 is staged first and c1*gP is added at the arg-max rows of the neighbourhoods this tile intersects
    auto patch_pool = [&](int kc) {
      const long long g0 = pg0;
      if constexpr (PFON)
#pragma unroll
      for (int e = 0; e < PF; ++e) {                 // the prefetched entries
        const int t = tid + e * NTH, gi = t >> 6, kk = t & 63;
        if (pf_arg[e] >= 0) {
          const long long row = (g0 + gi) * a.ns + pf_arg[e] - row0;
          if (row >= 0 && row < TM && row0 + row < a.M) {
            bf16 *cell = &sA[(int)row * AP + kk];
            *cell = (bf16)fmaf(sP[kc * KC + kk], pf_g[e], (float)*cell);
          }
        }
      }
      for (int t = tid + (PFON ? PF * NTH : 0); t < ngr * KC; t += NTH) {   // ns < 16 (CW = 1) only
        const int gi = t >> 6, kk = t & 63, k = kc * KC + kk;
        if (k < K) {
          const size_t o = (size_t)(g0 + gi) * K + k;
          const long long row = (g0 + gi) * a.ns + a.arg[o] - row0;
          if (row >= 0 && row < TM && row0 + row < a.M) {
            bf16 *cell = &sA[(int)row * AP + kk];
            *cell = (bf16)fmaf(sP[k], a.gP[o], (float)*cell);
          }
        }
      }
    };

    issue(0);
    issue_pool(0);
    for (int kc = 0; kc < nchunks; ++kc) {
      __syncthreads();                               // the previous chunk's fragment reads are done
      commit(kc);
      if (!a.wres) stage_w(kc * KC, KC);
      if (kc + 1 < nchunks) issue(kc + 1);
      if constexpr (PRO == PRO_POOLG) {
        __syncthreads();
        patch_pool(kc);
        if (kc + 1 < nchunks) issue_pool(kc + 1);    // in flight during this chunk's matrix products
      }
      __syncthreads();
      const int kb = a.wres ? kc * KC : 0;
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        const bf16x8 af = *(const bf16x8 *)&sA[(wave * 32 + (lane & 31)) * AP + ks * 16 + (lane >> 5) * 8];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const bf16x8 bfr = *(const bf16x8 *)&sW[((wc * NT + nt) * 32 + (lane & 31)) * WP + kb + ks * 16 + (lane >> 5) * 8];
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[nt], 0, 0, 0);
        }
      }
    }

    // ---- epilogue: C layout — lane holds column nt*32 + (lane & 31), rows (i&3) + 8*(i>>2) + 4*(lane>>5) of its wave
    const int ys = YF32 ? 4 : 2;
    const rsrc_t rY = EPI == EPI_POOL ? make_rsrc((const void *)a.pmax, 4)
                                      : make_rsrc((char *)a.Y + (size_t)row0 * a.ldy * ys, rows_left * a.ldy * ys);
    rsrc_t rYp = rY;
    if constexpr (EPI == EPI_MASK) rYp = make_rsrc((const char *)a.Yprev + (size_t)row0 * a.ldy * 2, rows_left * a.ldy * 2);
    if constexpr (EPI == EPI_MASK) {
      // ReLU-backward epilogue through LDS: the y_{l-1} tile comes in with coalesced 16-byte loads, every lane masks its
      // accumulator elements against it IN PLACE (C layout, 2-byte LDS accesses), and the tile leaves with coalesced
      // 16-byte stores — instead of 16 x NT two-byte global loads and stores per lane.
      bf16 *sY = sA;                                 // [TM][YP]: the A chunk is dead after the last MFMA
      constexpr int YP = NTT * 32 + 8;
      const int CGn = N >> 3;                        // N % 8 == 0 (bf16 rows)
      __syncthreads();                               // every wave is done with the A chunk
      for (int t = tid; t < TM * CGn; t += NTH) {
        const int r = t / CGn, cg = t - r * CGn;
        *(u32x4 *)&sY[r * YP + cg * 8] = bload128(rYp, (r * a.ldy + cg * 8) * 2, 0);
      }
      __syncthreads();
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = (wc * NT + nt) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
          bf16 *cell = &sY[r * YP + c];
          const float yp = (float)*cell;
          const float v = fmaf(yp, e_sc[nt], e_sh[nt]) > 0.f ? acc[nt][i] : 0.f;
          const bf16 vb = (bf16)v;
          const float vr = (float)vb;
          s1[nt] += vr;
          s2[nt] = fmaf(vr, (yp - e_mean[nt]) * e_rstd[nt], s2[nt]);
          *cell = vb;
        }
      }
      __syncthreads();
      for (int t = tid; t < TM * CGn; t += NTH) {
        const int r = t / CGn, cg = t - r * CGn;
        bstore128(*(const u32x4 *)&sY[r * YP + cg * 8], rY, (r * a.ldy + cg * 8) * 2, 0);
      }
    } else if constexpr (EPI == EPI_POOL) {
      // statistics + group maxima of the fp32 accumulators; nothing is stored to Y (see csrc/mlp_gemm.hip EPI_POOL: the C
      // layout is the same — registers 0-7 of a lane are one 16-row sub-group of the wave's 32 rows, 8-15 the other)
      const bool psz16 = a.ns == 16;
      const int psh = psz16 ? 4 : 5;
      const long long npart = a.M >> psh, pfirst = row0 >> psh;
      const long long pleft = npart > pfirst ? npart - pfirst : 0;
      const rsrc_t rspv = make_rsrc(pleft ? (const void *)(a.pmax + (size_t)pfirst * N) : (const void *)a.pmax, pleft ? pleft * N * 4 : 4);
      const rsrc_t rspr = make_rsrc(pleft ? (const void *)(a.parg + (size_t)pfirst * N) : (const void *)a.parg, pleft ? pleft * N * 4 : 4);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = (wc * NT + nt) * 32 + (lane & 31);
        float bst[2];
        int bi[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          bst[q] = acc[nt][8 * q];
          bi[q] = 0;
          s1[nt] += bst[q];
          s2[nt] = fmaf(bst[q], bst[q], s2[nt]);
#pragma unroll
          for (int r = 1; r < 8; ++r) {
            const float v = acc[nt][8 * q + r];
            const bool gt = v > bst[q];
            bst[q] = gt ? v : bst[q];
            bi[q] = gt ? r : bi[q];
            s1[nt] += v;
            s2[nt] = fmaf(v, v, s2[nt]);
          }
        }
        float b[2];
        int rw[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {        // the two half-waves interleave in blocks of four rows: larger value, then smaller row
          const int row = (bi[q] & 3) + 8 * (bi[q] >> 2) + 4 * (lane >> 5);
          const float ob = __shfl_xor(bst[q], 32);
          const int orow = __shfl_xor(row, 32);
          const bool take = ob > bst[q] || (ob == bst[q] && orow < row);
          b[q] = take ? ob : bst[q];
          rw[q] = take ? orow : row;
        }
        const bool second = b[1] > b[0];
        const float b32 = second ? b[1] : b[0];
        const int r32 = second ? 16 + rw[1] : rw[0];
        const bool hi = (lane >> 5) != 0;
        const float vout = psz16 ? (hi ? b[1] : b[0]) : b32;
        const int rout = psz16 ? (hi ? rw[1] : rw[0]) : r32;
        const int pl = psz16 ? wave * 2 + (lane >> 5) : wave;
        const int poff = (pleft && c < N && (psz16 || lane < 32)) ? (pl * N + c) * 4 : kOobOffset;
        bstore(vout, rspv, poff, 0);
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)rout, rspr, poff, 0, 0);
      }
    } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c = (wc * NT + nt) * 32 + (lane & 31);
      const bool cok = c < N;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        const int eoff = cok ? r * a.ldy + c : -1;
        float v = acc[nt][i];
        if constexpr (EPI == EPI_MASK) {
          const float yp = bf_lo((unsigned)bload16(rYp, eoff < 0 ? kOobOffset : eoff * 2, 0));
          v = fmaf(yp, e_sc[nt], e_sh[nt]) > 0.f ? v : 0.f;
          const float vr = YF32 ? v : bf_round(v);
          s1[nt] += vr;
          s2[nt] = fmaf(vr, (yp - e_mean[nt]) * e_rstd[nt], s2[nt]);
        } else if constexpr (EPI == EPI_STATS) {
          const float vr = YF32 ? v : bf_round(v);
          s1[nt] += vr;
          s2[nt] = fmaf(vr, vr, s2[nt]);
        }
        if constexpr (YF32) bstore(v, rY, eoff < 0 ? kOobOffset : eoff * 4, 0);
        else bstore16(bf_bits(v), rY, eoff < 0 ? kOobOffset : eoff * 2, 0);
      }
    }
    }
  }

  if constexpr (EPI != EPI_NONE) {
    // lanes l and l+32 hold the same column; 4 waves hold different rows: reduce through LDS, one fp64 atomic per column
    __syncthreads();
    float *red = (float *)sA;                        // [2][4][NTT*32] <= 10 KB of the 18 KB A chunk (the weights stay)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float t1 = s1[nt] + __shfl_xor(s1[nt], 32);
      const float t2 = s2[nt] + __shfl_xor(s2[nt], 32);
      if (lane < 32) {
        red[(0 * 4 + wave) * NTT * 32 + (wc * NT + nt) * 32 + lane] = t1;
        red[(1 * 4 + wave) * NTT * 32 + (wc * NT + nt) * 32 + lane] = t2;
      }
    }
    __syncthreads();
    for (int e = tid; e < 2 * NTT * 32; e += NTH) {
      const int which = e / (NTT * 32), c = e - which * NTT * 32;
      if (c < N) {
        const float *p = red + which * 4 * NTT * 32 + c;
        double tsum = (double)p[0] + (double)p[NTT * 32] + (double)p[2 * NTT * 32] + (double)p[3 * NTT * 32];
        if constexpr (EPI == EPI_POOL) { if (which == 0) tsum *= (double)a.sgn[c]; }
        atomicAdd(a.stats + (size_t)which * a.Nfull + c, tsum);
      }
    }
  }
  }   // virtual workgroups (the next one's first chunk commit follows a barrier)
}

// ---------------------------------------------------------------------------------------------- weight gradient
// dW[n][k] += sum_m gy[m][n] * act[m][k].  Both MFMA operands are indexed [feature][row]: A = gy^T, B = act^T, the
// contraction runs over rows.  A thread loads TWO consecutive rows x 8 consecutive features (16-byte pieces of the
// row-major tensors, coalesced across the features), applies the prologue and writes the 8 (row, row+1) pairs as
// packed dwords into the transposed LDS image T[feature][row]; row groups of 8 are XOR-swizzled with the feature
// group so that the 16 column groups of a wave store into different banks, and fragment reads stay 16-byte aligned.
struct WgradBf16Args {
  const bf16 *G;      // gmode GY: dL/dz_l [M][N]
  const bf16 *Yl;     // y_l [M][N]
  const float *consts;  // [3][N] c1 | c2 | c3
  const int *arg;     // gmode POOLG [M/ns][N]
  const float *gP;    // gmode POOLG [M/ns][N]
  const void *X;      // act source [M][ldx]: bf16 y_{l-1} (amode BNRELU) or the stack's input rows (amode NONE; fp32 when XF32)
  const float *a_fin; // amode BNRELU: [mean | rstd | scale | shift] x K
  float *dW;          // [N][K] ACCUMULATES
  long long M;
  int N, K, ldx, ns;
  // segment table (see GemmBf16Args): blockIdx.z = scan; consts + s*3N, a_fin + s*4K; dW is the scans' SUM
  const long long *seg;
  long long seg_max;
  int nseg;
};

template <int MT>
__device__ __forceinline__ int swz(int feature, int m) { return m ^ (((feature >> 3) & (MT / 8 - 1)) << 3); }

template <int NTW, int KTB, int MT, int GMODE, int AMODE, bool XF32>
__global__ __launch_bounds__(256, 2) void mlp_wgrad_bf16_kernel(const WgradBf16Args a_in) {
  WgradBf16Args a = a_in;
  if (a.seg) {
    const int sg = blockIdx.z;
    const long long r0 = a.seg[sg];
    a.M = a.seg[sg + 1] - r0;
    a.Yl += (size_t)r0 * a.N;
    if constexpr (GMODE == PRO_GY) a.G += (size_t)r0 * a.N;
    if constexpr (GMODE == PRO_POOLG) { a.arg += (size_t)(r0 / a.ns) * a.N; a.gP += (size_t)(r0 / a.ns) * a.N; }
    a.X = (const char *)a.X + (size_t)r0 * a.ldx * (XF32 ? 4 : 2);
    a.consts += (size_t)sg * 3 * a.N;
    if constexpr (AMODE == PRO_BNRELU) a.a_fin += (size_t)sg * 4 * a.K;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MP = MT + 8;                          // pitch in bf16 elements
  constexpr int NB = 4 * NTW * 32;                    // gy features held (>= N)
  constexpr int KB = KTB * 32;                        // act features per block
  bf16 *sG = (bf16 *)smem;                            // [NB][MP]
  bf16 *sX = sG + NB * MP;                            // [KB][MP]
  float *sC = (float *)(sX + KB * MP);                // [3][NB] gy constants, then [2][KB] act scale / shift

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N, K = a.K;
  const int k0 = blockIdx.y * KB;

  for (int n = tid; n < NB; n += 256) {
    sC[n] = n < N ? a.consts[n] : 0.f;
    sC[NB + n] = n < N ? a.consts[N + n] : 0.f;
    sC[2 * NB + n] = n < N ? a.consts[2 * N + n] : 0.f;
  }
  float *sS = sC + 3 * NB;
  for (int k = tid; k < KB; k += 256) {
    const bool ok = AMODE == PRO_BNRELU && k0 + k < K;
    sS[k] = ok ? a.a_fin[2 * K + k0 + k] : 0.f;
    sS[KB + k] = ok ? a.a_fin[3 * K + k0 + k] : 0.f;
  }
  __syncthreads();

  f32x16 acc[NTW][KTB];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < KTB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int CG = (N + 7) / 8;                         // 8-feature column groups of gy
  const int XG = XF32 ? 0 : (min(K - k0, KB) + 7) / 8;
  const long long ntiles = (a.M + MT - 1) / MT;

  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long row0 = tile * MT;
    const long long rows_left = a.M - row0;
    __syncthreads();                                  // previous tile's fragment reads are done

    // ---- gy tile -> sG (transposed, packed row pairs)
    {
      // PRO_POOLG: this tile's (arg-max row, pooled gradient) entries are requested BEFORE the dense part is staged —
      // entry t = tid + e * 256 of the ngr x Nr it has; loaded inside the patch loop below they were ngr rounds of two
      // global loads each, exposed between two barriers
      constexpr int PFW = NTW * KTB >= 6 ? 4 : 8;
      int pf_arg[PFW];
      float pf_g[PFW];
      const int Nr = (N + 255) & ~255;
      const long long q0 = GMODE == PRO_POOLG ? row0 / a.ns : 0;
      const long long qlast = (row0 + MT - 1 < a.M - 1 ? row0 + MT - 1 : a.M - 1);
      const int ngr = GMODE == PRO_POOLG ? (int)(qlast / a.ns - q0) + 1 : 0;
      if constexpr (GMODE == PRO_POOLG) {
#pragma unroll
        for (int e = 0; e < PFW; ++e) {
          const int t = tid + e * 256, gi = t / Nr, n = t - gi * Nr;
          pf_arg[e] = -1;
          pf_g[e] = 0.f;
          if (gi < ngr && n < N) {
            const size_t o = (size_t)(q0 + gi) * N + n;
            pf_arg[e] = a.arg[o];
            pf_g[e] = a.gP[o];
          }
        }
      }
      const rsrc_t rG = make_rsrc((const char *)(GMODE == PRO_GY ? a.G : a.Yl) + (size_t)row0 * N * 2, rows_left * N * 2);
      const rsrc_t rY = make_rsrc((const char *)a.Yl + (size_t)row0 * N * 2, rows_left * N * 2);
      const int tasks = (MT / 2) * CG;
      for (int t = tid; t < tasks; t += 256) {
        const int rp = t / CG, cg = t - rp * CG;
        const int m = 2 * rp, n = cg * 8;
        const int off0 = (m * N + n) * 2, off1 = off0 + N * 2;
        u32x4 g0, g1;
        const u32x4 y0 = bload128(rY, off0, 0), y1 = bload128(rY, off1, 0);
        if constexpr (GMODE == PRO_GY) { g0 = bload128(rG, off0, 0); g1 = bload128(rG, off1, 0); }
        float v0[8], v1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float c1 = sC[n + i], c2 = sC[NB + n + i], c3 = sC[2 * NB + n + i];
          float ga = 0.f, gb = 0.f;                  // PRO_POOLG: the pooled gradient is patched in below
          if constexpr (GMODE == PRO_GY) {
            ga = (i & 1) ? bf_hi(g0[i >> 1]) : bf_lo(g0[i >> 1]);
            gb = (i & 1) ? bf_hi(g1[i >> 1]) : bf_lo(g1[i >> 1]);
          }
          const float ya = (i & 1) ? bf_hi(y0[i >> 1]) : bf_lo(y0[i >> 1]);
          const float yb = (i & 1) ? bf_hi(y1[i >> 1]) : bf_lo(y1[i >> 1]);
          // rows past M must contribute nothing (their y reads are out of range = 0, but c3 is not)
          v0[i] = row0 + m < a.M ? fmaf(c1, ga, fmaf(c2, ya, c3)) : 0.f;
          v1[i] = row0 + m + 1 < a.M ? fmaf(c1, gb, fmaf(c2, yb, c3)) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          *(unsigned *)&sG[(n + i) * MP + swz<MT>(n + i, m)] = bf_pack(v0[i], v1[i]);
      }
      // features N .. NB-1 of the last partly used tile row block must read as zero
      for (int t = tid; t < (NB - CG * 8) * (MT / 2); t += 256) {
        const int f = CG * 8 + t / (MT / 2), mp = t % (MT / 2);
        *(unsigned *)&sG[f * MP + 2 * mp] = 0u;
      }
      if constexpr (GMODE == PRO_POOLG) {
        // one non-zero per (neighbourhood, feature): add c1 * gP at the arg-max row of every neighbourhood in this tile
        __syncthreads();
#pragma unroll
        for (int e = 0; e < PFW; ++e) {
          const int t = tid + e * 256, gi = t / Nr, n = t - gi * Nr;
          if (pf_arg[e] >= 0) {
            const long long row = (q0 + gi) * a.ns + pf_arg[e] - row0;
            if (row >= 0 && row < MT && row0 + row < a.M) {
              bf16 *cell = &sG[n * MP + swz<MT>(n, (int)row)];
              *cell = (bf16)fmaf(sC[n], pf_g[e], (float)*cell);
            }
          }
        }
        for (int t = tid + PFW * 256; t < ngr * Nr; t += 256) {   // more than PFW entries per thread (small ns, wide N)
          const int gi = t / Nr, n = t - gi * Nr;
          if (n < N) {
            const size_t o = (size_t)(q0 + gi) * N + n;
            const long long row = (q0 + gi) * a.ns + a.arg[o] - row0;
            if (row >= 0 && row < MT && row0 + row < a.M) {
              bf16 *cell = &sG[n * MP + swz<MT>(n, (int)row)];
              *cell = (bf16)fmaf(sC[n], a.gP[o], (float)*cell);
            }
          }
        }
      }
    }
    // ---- activation tile -> sX (transposed, packed row pairs)
    if constexpr (!XF32) {
      const rsrc_t rX = make_rsrc((const char *)a.X + (size_t)row0 * a.ldx * 2, rows_left * a.ldx * 2);
      const int tasks = (MT / 2) * XG;
      for (int t = tid; t < tasks; t += 256) {
        const int rp = t / XG, cg = t - rp * XG;
        const int m = 2 * rp, k = cg * 8;
        const int off0 = (m * a.ldx + k0 + k) * 2, off1 = off0 + a.ldx * 2;
        const u32x4 x0 = bload128(rX, off0, 0), x1 = bload128(rX, off1, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float xa = (i & 1) ? bf_hi(x0[i >> 1]) : bf_lo(x0[i >> 1]);
          float xb = (i & 1) ? bf_hi(x1[i >> 1]) : bf_lo(x1[i >> 1]);
          if constexpr (AMODE == PRO_BNRELU) {
            const float sc = sS[k + i], sh = sS[KB + k + i];
            xa = row0 + m < a.M ? fmaxf(fmaf(xa, sc, sh), 0.f) : 0.f;
            xb = row0 + m + 1 < a.M ? fmaxf(fmaf(xb, sc, sh), 0.f) : 0.f;
          }
          if (k0 + k + i >= K) xa = xb = 0.f;
          *(unsigned *)&sX[(k + i) * MP + swz<MT>(k + i, m)] = bf_pack(xa, xb);
        }
      }
      for (int t = tid; t < (KB - XG * 8) * (MT / 2); t += 256) {
        const int f = XG * 8 + t / (MT / 2), mp = t % (MT / 2);
        *(unsigned *)&sX[f * MP + 2 * mp] = 0u;
      }
    } else {
      // fp32 input rows of arbitrary width (amode NONE): one (row pair, feature) per task, coalesced over features
      const rsrc_t rX = make_rsrc((const char *)a.X + (size_t)row0 * a.ldx * 4, rows_left * a.ldx * 4);
      const int tasks = (MT / 2) * KB;
      for (int t = tid; t < tasks; t += 256) {
        const int rp = t / KB, k = t - rp * KB;
        const int m = 2 * rp;
        const bool ok = k0 + k < K;
        const float xa = bload(rX, ok ? (m * a.ldx + k0 + k) * 4 : kOobOffset, 0);
        const float xb = bload(rX, ok ? ((m + 1) * a.ldx + k0 + k) * 4 : kOobOffset, 0);
        *(unsigned *)&sX[k * MP + swz<MT>(k, m)] = bf_pack(xa, xb);
      }
    }
    __syncthreads();

    // ---- dW tile block += gy^T act over the MT rows
#pragma unroll
    for (int ms = 0; ms < MT / 16; ++ms) {
      const int m = ms * 16 + (lane >> 5) * 8;
      bf16x8 af[NTW], bfr[KTB];
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        const int f = (wave * NTW + i) * 32 + (lane & 31);
        af[i] = *(const bf16x8 *)&sG[f * MP + swz<MT>(f, m)];
      }
#pragma unroll
      for (int j = 0; j < KTB; ++j) {
        const int f = j * 32 + (lane & 31);
        bfr[j] = *(const bf16x8 *)&sX[f * MP + swz<MT>(f, m)];
      }
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int j = 0; j < KTB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- flush: C layout — column (lane & 31) = k, rows = n
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < KTB; ++j) {
      const int k = k0 + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = (wave * NTW + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (n < N && k < K) atomicAdd(a.dW + (size_t)n * K + k, acc[i][j][e]);
      }
    }
}

// ---------------------------------------------------------------------------------------------- one-pass backward
// dgrad AND wgrad of a hidden layer from ONE read of (g, y_l, y_{l-1}) — what pn2_mlp_gemm_bf16(pro 2|3, epi 2) and
// pn2_mlp_wgrad_bf16(amode 1) compute for the same operands:
//   Gout[M][K] = [BN(y_{l-1}) > 0] * (gy * W),  sums += (sum Gout, sum Gout * yhat_{l-1}),  dW[N][K] += gy^T relu(BN(y_{l-1}))
// Both kernels are HBM-bound and between them read g and y_l twice and y_{l-1} twice; here a 64-row tile of gy is staged
// row-major (dgrad A operand) AND transposed (wgrad A operand), the activation tile transposed (wgrad B operand) and
// raw row-major (ReLU mask / yhat of the dgrad epilogue, overwritten in place by the masked result and stored with
// 16-byte rows), the transposed weights stay resident.  8 waves: the 2 x K/32 dgrad tiles and the N/32 x K/32 wgrad
// tiles are dealt round-robin; wgrad accumulators persist across the workgroup's tiles.  N, K in {32, 64, 96, 128}.
struct BwdBf16Args {
  const bf16 *G, *Yl;
  const float *consts;      // [3][N]
  const int *arg;
  const float *gP;
  const float *Wt;          // [K][N] fp32 (transposed weights, as the dgrad call takes them)
  const bf16 *Yprev;        // [M][K]
  const float *a_fin;       // [mean | rstd | scale | shift] x K
  bf16 *Gout;               // [M][K]
  double *sums;             // [2][K]
  float *dW;                // [N][K] ACCUMULATES
  long long M;
  int N, K, ns;
  // FOLD (layer l-1 is the stack's first layer, its input rows X [M][8] bf16 — K0 <= 8 real columns, zero padded — need no
  // gradient): Gout is not stored, P1 [K][K0] += Gout^T X is reduced instead (the first-layer fold of csrc/mlp_bwd_fused.hip)
  const bf16 *X;
  float *P1;
  int K0;
  const float *W0;          // FY (FOLD with the first layer's output NOT stored): its weight [K][K0]; y_0 = X W0^T is re-formed per tile
  // segment table (see GemmBf16Args): blockIdx.y = scan; consts + s*3N, a_fin + s*4K, sums + s*2K; dW is the scans' SUM
  const long long *seg;
  long long seg_max;
  int nseg;
  int vcap;                 // grid cap of a single-scan call (virtual workgroups, see mlp_gemm_bf16_kernel)
};

// RECOMP (GMODE PRO_POOLG: the layer is a max-pooled LAST layer whose output was never stored, pn2_mlp_gemm_pool_bf16): the
// y_l tile is re-formed from the y_{l-1} tile the kernel stages anyway — the forward's own matrix product (same operands, same
// MFMA order: the fp32 accumulators the forward took its maxima and statistics from, bit for bit) — instead of loaded.
template <int NTN, int KTK, int GMODE, bool FOLD = false, bool RECOMP = false, bool FY = false>
// (HIP's second launch-bounds argument is waves per SIMD, not workgroups per CU.  The FOLD variants took 144-156 registers
// under "2": three waves per SIMD, i.e. ONE eight-wave workgroup per CU.  Capped at 128 they spill 24-88 bytes outside
// the tile loop's matrix products and two workgroups share a CU: 4.2M x 64 x 64 fold 0.59 -> 0.48 ms.  The 128 x 128
// non-fold variants lose with the same cap (0.22 -> 0.25 ms; pooled form at 18.9M rows 5.06 -> 5.23 ms) and keep their registers.)
__global__ __launch_bounds__(512, (FOLD || (RECOMP && KTK <= 2)) ? 4 : 2) void mlp_bwd_bf16_kernel(const BwdBf16Args a_in) {
  BwdBf16Args a = a_in;
  if (a.seg) {
    const int sg = blockIdx.y;
    const long long r0 = a.seg[sg];
    a.M = a.seg[sg + 1] - r0;
    a.Yl += (size_t)r0 * a.N;
    if constexpr (GMODE == PRO_GY) a.G += (size_t)r0 * a.N;
    if constexpr (GMODE == PRO_POOLG) { a.arg += (size_t)(r0 / a.ns) * a.N; a.gP += (size_t)(r0 / a.ns) * a.N; }
    a.Yprev += (size_t)r0 * a.K;
    if constexpr (!FOLD) a.Gout += (size_t)r0 * a.K;
    a.consts += (size_t)sg * 3 * a.N;
    a.a_fin += (size_t)sg * 4 * a.K;
    a.sums += (size_t)sg * 2 * a.K;
  }
  constexpr int MT = 64, MP = MT + 8;
  constexpr int NB = NTN * 32, KB = KTK * 32, NP = NB + 8, KP = KB + 8;
  constexpr int DT = (2 * KTK + 7) / 8;              // dgrad tiles per wave
  constexpr int WT = (NTN * KTK + 7) / 8;            // wgrad tiles per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16 *sWt = (bf16 *)smem;                          // [KB][NP]
  bf16 *sGY = sWt + KB * NP;                         // [MT][NP]
  bf16 *sGT = sGY + MT * NP;                         // [NB][MP]
  bf16 *sXT = sGT + NB * MP;                         // [KB][MP]
  bf16 *sO = sXT + KB * MP;                          // [MT][KP]
  float *sC = (float *)(sO + MT * KP);               // c1 | c2 | c3 [NB], then mean | rstd | scale | shift [KB]
  float *sF = sC + 3 * NB;
  float *sXr = sF + 4 * KB;                          // FOLD: [MT][8] input rows of the tile (fp32), then [KB][8] for the flush

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N, K = a.K;                        // == NB, KB (checked by the host wrapper)
  for (int n = tid; n < NB; n += 512) { sC[n] = a.consts[n]; sC[NB + n] = a.consts[N + n]; sC[2 * NB + n] = a.consts[2 * N + n]; }
  for (int k = tid; k < KB; k += 512) {
    sF[k] = a.a_fin[k]; sF[KB + k] = a.a_fin[K + k]; sF[2 * KB + k] = a.a_fin[2 * K + k]; sF[3 * KB + k] = a.a_fin[3 * K + k];
  }

  {
    // transposed weights -> LDS: a thread per four consecutive n (16-byte load, 8-byte store), four rows in flight
    // (N = NB is a multiple of 32: aligned).  One float per lane and row at a time cost every workgroup ~30 us.
    typedef float w4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    constexpr int LPRW = NB / 4;                       // lanes per weight row (8 .. 32)
    constexpr int RPI = 512 / LPRW;                    // rows per pass of the workgroup
    const int wn = (tid % LPRW) * 4, wk0 = tid / LPRW;
    for (int k0 = wk0; k0 < KB; k0 += 4 * RPI) {
      w4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * RPI;
        v[u] = k < KB ? *(const w4 *)(a.Wt + (size_t)k * N + wn) : w4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * RPI;
        if (k < KB) *(u2 *)&sWt[k * NP + wn] = u2{bf_pack(v[u][0], v[u][1]), bf_pack(v[u][2], v[u][3])};
      }
    }
  }

  // RECOMP: the wave's block of the re-formed y_l tile has the same columns in every tile (2 NTN <= 8 blocks, one per wave):
  // its weight fragments — 8 consecutive k of column n, gathered from the k-major LDS image — are formed ONCE
  typedef unsigned short us8 __attribute__((ext_vector_type(8)));
  constexpr bool RC1 = RECOMP && 2 * NTN <= 8;
  bf16x8 wfrag[RC1 ? KB / 16 : 1];
  if constexpr (RC1) {
    __syncthreads();                                   // the weights are in LDS
    const int ct = wave % NTN;
#pragma unroll
    for (int ks = 0; ks < KB / 16; ++ks) {
      us8 bw;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        bw[j] = __builtin_bit_cast(unsigned short, sWt[(ks * 16 + (lane >> 5) * 8 + j) * NP + ct * 32 + (lane & 31)]);
      wfrag[ks] = __builtin_bit_cast(bf16x8, bw);
    }
  }
  f32x16 accw[WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) accw[i][e] = 0.f;
  float s1[DT], s2[DT];
  float px[FOLD ? DT : 1][8];
#pragma unroll
  for (int i = 0; i < (FOLD ? DT : 1); ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) px[i][j] = 0.f;

  constexpr int CGn = NB / 8, CGk = KB / 8;
  constexpr int GT = (32 * CGn + 511) / 512;         // gy tasks per thread (row pair x 8 columns)
  constexpr int XT = (32 * CGk + 511) / 512;
  const long long ntiles = (a.M + MT - 1) / MT;

  u32x4 rg0[GT], rg1[GT], ry0[GT], ry1[GT], rx0[XT], rx1[XT];
  u32x4 rxr = {0u, 0u, 0u, 0u};
  u32x4 rxa = {0u, 0u, 0u, 0u};
  // PRO_POOLG: the (arg-max row, pooled gradient) entries of a tile's groups travel with its row loads (entry t = tid + 512 e
  // of the ngr x NB entries; ns >= 16: at most MT / 16 + 1 groups) — fetched inside the patch they were two dependent global
  // loads between two barriers, once per tile
  constexpr int PFE = GMODE == PRO_POOLG ? ((MT / 16 + 1) * NB + 511) / 512 : 1;
  int pfa[PFE];
  float pfg[PFE];
  // FY: y_0 = X W0^T on the matrix pipe — K0 <= 8 input columns zero-padded to one 16-wide step: ONE instruction per 32 x 32 block
  // of the tile (2 KTK <= 4 blocks, waves 0 .. 2 KTK - 1), bf16 operands like the first layer's own GEMM on this path.  The
  // weight fragment (row n = this lane's column of the block, eight k) is formed once.
  bf16x8 w0frag = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
  if constexpr (FY) {
    static_assert(!FY || 2 * KTK <= 8, "one y_0 block per wave");
    const int n = (wave % KTK) * 32 + (lane & 31);
    if (wave < 2 * KTK && lane < 32 && n < K) {
      float w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = j < a.K0 ? a.W0[(size_t)n * a.K0 + j] : 0.f;
      w0frag = __builtin_bit_cast(bf16x8, u32x4{bf_pack(w[0], w[1]), bf_pack(w[2], w[3]), bf_pack(w[4], w[5]), bf_pack(w[6], w[7])});
    }
  }
  auto issue = [&](long long tile) {
    const long long row0 = tile * MT, left = a.M - row0;
    if constexpr (FOLD) {
      const rsrc_t rX0 = make_rsrc((const char *)a.X + (size_t)row0 * 16, left * 16);
      rxr = bload128(rX0, tid < MT ? tid * 16 : kOobOffset, 0);
    }
    const rsrc_t rX = FY ? make_rsrc((const char *)a.X + (size_t)row0 * 16, left * 16)
                         : make_rsrc((const char *)a.Yprev + (size_t)row0 * K * 2, left * K * 2);
    const rsrc_t rG = RECOMP ? rX : make_rsrc((const char *)(GMODE == PRO_GY ? a.G : a.Yl) + (size_t)row0 * N * 2, left * N * 2);
    const rsrc_t rY = RECOMP ? rX : make_rsrc((const char *)a.Yl + (size_t)row0 * N * 2, left * N * 2);
#pragma unroll
    for (int i = 0; i < (RECOMP ? 0 : GT); ++i) {
      const int t = tid + 512 * i;
      const int rp = t / CGn, cg = t - rp * CGn;
      const int off = t < 32 * CGn ? (2 * rp * N + cg * 8) * 2 : kOobOffset;
      ry0[i] = bload128(rY, off, 0);
      ry1[i] = bload128(rY, off, N * 2);
      if constexpr (GMODE == PRO_GY) { rg0[i] = bload128(rG, off, 0); rg1[i] = bload128(rG, off, N * 2); }
    }
#pragma unroll
    for (int i = 0; i < (FY ? 0 : XT); ++i) {
      const int t = tid + 512 * i;
      const int rp = t / CGk, cg = t - rp * CGk;
      const int off = t < 32 * CGk ? (2 * rp * K + cg * 8) * 2 : kOobOffset;
      rx0[i] = bload128(rX, off, 0);
      rx1[i] = bload128(rX, off, K * 2);
    }
    if constexpr (GMODE == PRO_POOLG) {
      const long long q0 = row0 / a.ns;
      const long long last = (row0 + MT - 1 < a.M - 1 ? row0 + MT - 1 : a.M - 1);
      const int ngr = (int)(last / a.ns - q0) + 1;
#pragma unroll
      for (int e = 0; e < PFE; ++e) {
        const int t = tid + 512 * e, gi = t / NB, n = t - gi * NB;
        pfa[e] = -1;
        pfg[e] = 0.f;
        if (gi < ngr) {
          const size_t o = (size_t)(q0 + gi) * N + n;
          pfa[e] = a.arg[o];
          pfg[e] = a.gP[o];
        }
      }
    }
    if constexpr (FY) {
      // the 32 input rows of this wave's block of the y_0 tile (16 bytes per row), lanes 32-63 supply the zero half of k
      const int t = wave;
      rxa = (t < 2 * KTK && lane < 32) ? bload128(rX, ((t / KTK) * 32 + lane) * 16, 0) : u32x4{0u, 0u, 0u, 0u};
    }
  };

  // virtual workgroups: a scan of a segmented call keeps the per-workgroup grouping of ITS OWN call's column sums
  const int vstride = a.seg ? (int)(ntiles < a.vcap ? ntiles : a.vcap) : (int)gridDim.x;
  for (int vwg = blockIdx.x; vwg < vstride; vwg += gridDim.x) {
#pragma unroll
  for (int i = 0; i < DT; ++i) s1[i] = s2[i] = 0.f;
  long long tile = vwg;
  if (tile < ntiles) issue(tile);
  for (; tile < ntiles; tile += vstride) {
    const long long row0 = tile * MT;
    __syncthreads();                                 // previous tile: MFMA reads and the output store are done
    if constexpr (FOLD) {
      if (tid < MT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { sXr[tid * 8 + 2 * e] = bf_lo(rxr[e]); sXr[tid * 8 + 2 * e + 1] = bf_hi(rxr[e]); }
      }
    }
    // ---- commit: gy tile (row-major + transposed), activation tile (raw row-major + activated transposed)
#pragma unroll
    for (int i = 0; i < (RECOMP ? 0 : GT); ++i) {
      const int t = tid + 512 * i;
      if (t < 32 * CGn) {
        const int rp = t / CGn, cg = t - rp * CGn, m = 2 * rp, n = cg * 8;
        const bool ok0 = row0 + m < a.M, ok1 = row0 + m + 1 < a.M;
        float v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float c1 = sC[n + e], c2 = sC[NB + n + e], c3 = sC[2 * NB + n + e];
          float ga = 0.f, gb = 0.f;
          if constexpr (GMODE == PRO_GY) {
            ga = (e & 1) ? bf_hi(rg0[i][e >> 1]) : bf_lo(rg0[i][e >> 1]);
            gb = (e & 1) ? bf_hi(rg1[i][e >> 1]) : bf_lo(rg1[i][e >> 1]);
          }
          const float ya = (e & 1) ? bf_hi(ry0[i][e >> 1]) : bf_lo(ry0[i][e >> 1]);
          const float yb = (e & 1) ? bf_hi(ry1[i][e >> 1]) : bf_lo(ry1[i][e >> 1]);
          v0[e] = ok0 ? fmaf(c1, ga, fmaf(c2, ya, c3)) : 0.f;
          v1[e] = ok1 ? fmaf(c1, gb, fmaf(c2, yb, c3)) : 0.f;
        }
        u32x4 w0, w1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { w0[e] = bf_pack(v0[2 * e], v0[2 * e + 1]); w1[e] = bf_pack(v1[2 * e], v1[2 * e + 1]); }
        *(u32x4 *)&sGY[m * NP + n] = w0;
        *(u32x4 *)&sGY[(m + 1) * NP + n] = w1;
#pragma unroll
        for (int e = 0; e < 8; ++e) *(unsigned *)&sGT[(n + e) * MP + swz<MT>(n + e, m)] = bf_pack(v0[e], v1[e]);
      }
    }
    if constexpr (FY) {
      const int t = wave;
      if (t < 2 * KTK) {                               // wave-uniform
        const int rt = t / KTK, ct = t - rt * KTK;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rxa), w0frag, acc, 0, 0, 0);
        const int n = ct * 32 + (lane & 31);
        const float sc = sF[2 * KB + n], sh = sF[3 * KB + n];
        typedef unsigned u2f __attribute__((ext_vector_type(2)));
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
          const int m0 = rt * 32 + 8 * q + 4 * (lane >> 5);
          float av[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float y = acc[4 * q + j];
            sO[(m0 + j) * KP + n] = (bf16)y;                                   // raw y_0 (rows past M: zero input rows -> 0)
            av[j] = row0 + m0 + j < a.M ? fmaxf(fmaf(y, sc, sh), 0.f) : 0.f;
          }
          *(u2f *)&sXT[n * MP + swz<MT>(n, m0)] = u2f{bf_pack(av[0], av[1]), bf_pack(av[2], av[3])};
        }
      }
    }
#pragma unroll
    for (int i = 0; i < (FY ? 0 : XT); ++i) {
      const int t = tid + 512 * i;
      if (t < 32 * CGk) {
        const int rp = t / CGk, cg = t - rp * CGk, m = 2 * rp, k = cg * 8;
        const bool ok0 = row0 + m < a.M, ok1 = row0 + m + 1 < a.M;
        *(u32x4 *)&sO[m * KP + k] = rx0[i];
        *(u32x4 *)&sO[(m + 1) * KP + k] = rx1[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sc = sF[2 * KB + k + e], sh = sF[3 * KB + k + e];
          const float xa = (e & 1) ? bf_hi(rx0[i][e >> 1]) : bf_lo(rx0[i][e >> 1]);
          const float xb = (e & 1) ? bf_hi(rx1[i][e >> 1]) : bf_lo(rx1[i][e >> 1]);
          const float aa = ok0 ? fmaxf(fmaf(xa, sc, sh), 0.f) : 0.f;
          const float ab = ok1 ? fmaxf(fmaf(xb, sc, sh), 0.f) : 0.f;
          *(unsigned *)&sXT[(k + e) * MP + swz<MT>(k + e, m)] = bf_pack(aa, ab);
        }
      }
    }
    if constexpr (RECOMP) {
      // y_l tile = relu(bn(y_{l-1})) W^T on the matrix pipe (one 32 x 32 block per wave and pass), then the dense part of
      // dL/dy_l = c2 y_l + c3 into the row-major and the transposed gy tiles; the arg-max patch below adds c1 gP
      __syncthreads();                               // the raw y_{l-1} tile is in sO
      typedef unsigned u2r __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int yi = 0; yi < (2 * NTN + 7) / 8; ++yi) {
        const int t = wave + 8 * yi;
        if (t < 2 * NTN) {                           // wave-uniform
          const int rt = t / NTN, ct = t - rt * NTN;
          f32x16 acc;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KB / 16; ++ks) {
            const int k0 = ks * 16 + (lane >> 5) * 8;
            const u32x4 raw = *(const u32x4 *)&sO[(rt * 32 + (lane & 31)) * KP + k0];
            u32x4 aw;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float lo = fmaxf(fmaf(bf_lo(raw[e]), sF[2 * KB + k0 + 2 * e], sF[3 * KB + k0 + 2 * e]), 0.f);
              const float hi = fmaxf(fmaf(bf_hi(raw[e]), sF[2 * KB + k0 + 2 * e + 1], sF[3 * KB + k0 + 2 * e + 1]), 0.f);
              aw[e] = bf_pack(lo, hi);
            }
            bf16x8 bfr;
            if constexpr (RC1) {
              bfr = wfrag[ks];
            } else {
              us8 bw;
#pragma unroll
              for (int j = 0; j < 8; ++j)
                bw[j] = __builtin_bit_cast(unsigned short, sWt[(k0 + j) * NP + ct * 32 + (lane & 31)]);
              bfr = __builtin_bit_cast(bf16x8, bw);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw), bfr, acc, 0, 0, 0);
          }
          const int n = ct * 32 + (lane & 31);
          const float c2 = sC[NB + n], c3 = sC[2 * NB + n];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int m0 = rt * 32 + 8 * q + 4 * (lane >> 5);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = row0 + m0 + j < a.M ? fmaf(c2, acc[4 * q + j], c3) : 0.f;
              sGY[(m0 + j) * NP + n] = (bf16)v[j];
            }
            *(u2r *)&sGT[n * MP + swz<MT>(n, m0)] = u2r{bf_pack(v[0], v[1]), bf_pack(v[2], v[3])};
          }
        }
      }
    }
    if constexpr (GMODE == PRO_POOLG) {
      __syncthreads();
      const long long q0 = row0 / a.ns;
      const long long last = (row0 + MT - 1 < a.M - 1 ? row0 + MT - 1 : a.M - 1);
      const int ngr = (int)(last / a.ns - q0) + 1;
      if (a.ns >= 16) {
#pragma unroll
        for (int e = 0; e < PFE; ++e) {                  // the prefetched entries
          const int t = tid + 512 * e, gi = t / NB, n = t - gi * NB;
          if (pfa[e] >= 0) {
            const long long row = (q0 + gi) * a.ns + pfa[e] - row0;
            if (row >= 0 && row < MT && row0 + row < a.M) {
              bf16 *c0 = &sGY[(int)row * NP + n];
              const bf16 nv = (bf16)fmaf(sC[n], pfg[e], (float)*c0);
              *c0 = nv;
              sGT[n * MP + swz<MT>(n, (int)row)] = nv;
            }
          }
        }
      } else {
      for (int t = tid; t < ngr * NB; t += 512) {
        const int gi = t / NB, n = t - gi * NB;
        const size_t o = (size_t)(q0 + gi) * N + n;
        const long long row = (q0 + gi) * a.ns + a.arg[o] - row0;
        if (row >= 0 && row < MT && row0 + row < a.M) {
          bf16 *c0 = &sGY[(int)row * NP + n];
          const bf16 nv = (bf16)fmaf(sC[n], a.gP[o], (float)*c0);
          *c0 = nv;
          sGT[n * MP + swz<MT>(n, (int)row)] = nv;
        }
      }
      }
    }
    __syncthreads();
    if (tile + vstride < ntiles) issue(tile + vstride);       // next tile's loads fly behind this tile's MFMAs

    // ---- dgrad tiles of this wave
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const int t = wave + 8 * d;
      if (t < 2 * KTK) {                                       // wave-uniform
        const int rt = t / KTK, ct = t - rt * KTK;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NB / 16; ++ks) {
          const bf16x8 af = *(const bf16x8 *)&sGY[(rt * 32 + (lane & 31)) * NP + ks * 16 + (lane >> 5) * 8];
          const bf16x8 bfr = *(const bf16x8 *)&sWt[(ct * 32 + (lane & 31)) * NP + ks * 16 + (lane >> 5) * 8];
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc, 0, 0, 0);
        }
        const int k = ct * 32 + (lane & 31);
        const float mean = sF[k], rstd = sF[KB + k], sc = sF[2 * KB + k], sh = sF[3 * KB + k];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          bf16 *cell = &sO[m * KP + k];
          const float yp = (float)*cell;
          const float v = fmaf(yp, sc, sh) > 0.f ? acc[e] : 0.f;
          const bf16 vb = (bf16)v;
          const float vr = (float)vb;
          s1[d] += vr;
          s2[d] = fmaf(vr, (yp - mean) * rstd, s2[d]);
          if constexpr (FOLD) {
            if ((e & 3) == 0) __builtin_amdgcn_sched_barrier(0);      // keep the X-row reads from being hoisted en bloc
            typedef float f2 __attribute__((ext_vector_type(2)));
            const float4 xa = *(const float4 *)&sXr[m * 8], xb = *(const float4 *)&sXr[m * 8 + 4];
            const f2 v2 = {vr, vr};
            f2 *pp = reinterpret_cast<f2 *>(px[FOLD ? d : 0]);
            pp[0] = __builtin_elementwise_fma(v2, f2{xa.x, xa.y}, pp[0]);   // v_pk_fma_f32
            pp[1] = __builtin_elementwise_fma(v2, f2{xa.z, xa.w}, pp[1]);
            pp[2] = __builtin_elementwise_fma(v2, f2{xb.x, xb.y}, pp[2]);
            pp[3] = __builtin_elementwise_fma(v2, f2{xb.z, xb.w}, pp[3]);
          } else {
            *cell = vb;
          }
        }
      }
    }
    // ---- wgrad tiles of this wave
#pragma unroll
    for (int ms = 0; ms < MT / 16; ++ms) {
      const int m = ms * 16 + (lane >> 5) * 8;
#pragma unroll
      for (int i = 0; i < WT; ++i) {
        const int t = wave + 8 * i;
        if (t < NTN * KTK) {
          const int nt = t / KTK, kt = t - nt * KTK;
          const int fn = nt * 32 + (lane & 31), fk = kt * 32 + (lane & 31);
          const bf16x8 af = *(const bf16x8 *)&sGT[fn * MP + swz<MT>(fn, m)];
          const bf16x8 bfr = *(const bf16x8 *)&sXT[fk * MP + swz<MT>(fk, m)];
          accw[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, accw[i], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                           // masked tile complete in sO
    if constexpr (!FOLD) {
      const long long left = a.M - row0;
      const rsrc_t rO = make_rsrc((char *)a.Gout + (size_t)row0 * K * 2, left * K * 2);
      for (int t = tid; t < MT * CGk; t += 512) {
        const int r = t / CGk, cg = t - r * CGk;
        bstore128(*(const u32x4 *)&sO[r * KP + cg * 8], rO, (r * K + cg * 8) * 2, 0);
      }
    }
  }

  // ---- flush: column sums (LDS reduction over the waves that share a column tile); dW after the last virtual workgroup
  __syncthreads();
  float *red = (float *)sGY;                                   // [2][KB], zeroed (<= 1 KB of the gy tile; the weights stay)
  for (int i = tid; i < 2 * KB; i += 512) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    const int t = wave + 8 * d;
    if (t < 2 * KTK) {
      const int ct = t % KTK;
      const float t1 = s1[d] + __shfl_xor(s1[d], 32), t2 = s2[d] + __shfl_xor(s2[d], 32);
      if (lane < 32) { atomicAdd(&red[ct * 32 + lane], t1); atomicAdd(&red[KB + ct * 32 + lane], t2); }
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * KB; i += 512) atomicAdd(a.sums + (size_t)(i / KB) * K + (i % KB), (double)red[i]);
  }   // virtual workgroups
  if constexpr (FOLD) {
    __syncthreads();
    float *redP = sXr;                                         // [KB][8]
    for (int i = tid; i < KB * 8; i += 512) redP[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const int t = wave + 8 * d;
      if (t < 2 * KTK) {
        const int k = (t % KTK) * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&redP[k * 8 + j], px[FOLD ? d : 0][j]);
      }
    }
    __syncthreads();
    for (int i = tid; i < KB * 8; i += 512) {
      const int k = i >> 3, j = i & 7;
      if (j < a.K0) atomicAdd(a.P1 + (size_t)k * a.K0 + j, redP[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < WT; ++i) {
    const int t = wave + 8 * i;
    if (t < NTN * KTK) {
      const int nt = t / KTK, kt = t - nt * KTK;
      const int k = kt * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        atomicAdd(a.dW + (size_t)n * K + k, accw[i][e]);
      }
    }
  }
}

template <int NTN, int KTK, int GMODE, bool FOLD = false, bool RECOMP = false, bool FY = false>
int launch_bwd(const BwdBf16Args &a, hipStream_t s) {
  constexpr int MT = 64, MP = MT + 8, NB = NTN * 32, KB = KTK * 32, NP = NB + 8, KP = KB + 8;
  const size_t lds = (size_t)(KB * NP + MT * NP + NB * MP + KB * MP + MT * KP) * 2 + (size_t)(3 * NB + 4 * KB) * 4 +
                     (FOLD ? (size_t)(MT > KB ? MT : KB) * 8 * 4 : 0);
  auto kfn = mlp_bwd_bf16_kernel<NTN, KTK, GMODE, FOLD, RECOMP, FY>;
  static bool big_lds = false;
  if (lds > 64 * 1024 && !big_lds) {
    if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return pn2_check_launch();
    big_lds = true;
  }
  const long long ntiles = ((a.seg ? a.seg_max : a.M) + MT - 1) / MT;
  long long grid = lds > 80 * 1024 ? 256 : 512;
  BwdBf16Args b = a;
  b.vcap = (int)grid;
  const int nseg = a.seg ? a.nseg : 1;
  if (nseg > 1) grid = (grid + nseg - 1) / nseg;
  if (grid > ntiles) grid = ntiles;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kfn, dim3((unsigned)grid, (unsigned)nseg), dim3(512), lds, s, b);
  return pn2_check_launch();
}

template <int GMODE>
int dispatch_bwd_fold(const BwdBf16Args &a, hipStream_t s) {
  switch ((a.N / 32) * 8 + a.K / 32) {
    case 1 * 8 + 1: return launch_bwd<1, 1, GMODE, true>(a, s);
    case 1 * 8 + 2: return launch_bwd<1, 2, GMODE, true>(a, s);
    case 2 * 8 + 1: return launch_bwd<2, 1, GMODE, true>(a, s);
    case 2 * 8 + 2: return launch_bwd<2, 2, GMODE, true>(a, s);
    default: return PN2_EINVAL;
  }
}

template <int GMODE>
int dispatch_bwd(const BwdBf16Args &a, hipStream_t s) {
  switch ((a.N / 32) * 8 + a.K / 32) {
    case 1 * 8 + 1: return launch_bwd<1, 1, GMODE>(a, s);
    case 1 * 8 + 2: return launch_bwd<1, 2, GMODE>(a, s);
    case 2 * 8 + 1: return launch_bwd<2, 1, GMODE>(a, s);
    case 2 * 8 + 2: return launch_bwd<2, 2, GMODE>(a, s);
    case 2 * 8 + 4: return launch_bwd<2, 4, GMODE>(a, s);
    case 4 * 8 + 2: return launch_bwd<4, 2, GMODE>(a, s);
    case 4 * 8 + 4: return launch_bwd<4, 4, GMODE>(a, s);
    case 3 * 8 + 3: return launch_bwd<3, 3, GMODE>(a, s);
    case 4 * 8 + 3: return launch_bwd<4, 3, GMODE>(a, s);
    case 3 * 8 + 4: return launch_bwd<3, 4, GMODE>(a, s);
    default: return PN2_EINVAL;
  }
}

// ---------------------------------------------------------------------------------------------- element-wise helpers
// ReLU(BN(y)) for a stack that returns activations (FP modules): y bf16 -> out fp32
__global__ __launch_bounds__(256) void bn_relu_apply_bf16_kernel(size_t total, int N, const bf16 *__restrict__ y,
                                                                const float *__restrict__ fin, float *__restrict__ out) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % N);
    out[e] = fmaxf(fmaf((float)y[e], fin[2 * N + c], fin[3 * N + c]), 0.f);
  }
}

constexpr int kPrepRowsBf = 16;

// gpre = gout * [BN(y) > 0] (bf16), sums += (sum gpre, sum gpre * yhat) — backward entry of an un-pooled stack
__global__ __launch_bounds__(256) void bn_relu_bwd_prep_bf16_kernel(long long M, int N, int rpb, const bf16 *__restrict__ y,
                                                                   const float *__restrict__ gout,
                                                                   const float *__restrict__ fin,
                                                                   bf16 *__restrict__ gpre, double *__restrict__ sums) {
  __shared__ float part[2][256];
  const long long r0 = (long long)blockIdx.x * rpb;
  const int cw = N < 256 ? N : 256;
  const int groups = 256 / cw;
  const int grp = threadIdx.x / cw;
  const bool active = grp < groups;
  for (int c0 = 0; c0 < N; c0 += cw) {
    const int c = c0 + threadIdx.x - grp * cw;
    float s1 = 0.f, s2 = 0.f;
    if (active && c < N) {
      const float mean = fin[c], rstd = fin[N + c], sc = fin[2 * N + c], sh = fin[3 * N + c];
      for (int r = grp; r < rpb; r += groups) {
        const long long row = r0 + r;
        if (row >= M) break;
        const size_t off = (size_t)row * N + c;
        const float yy = (float)y[off];
        const float g = bf_round(fmaf(yy, sc, sh) > 0.f ? gout[off] : 0.f);
        gpre[off] = (bf16)g;
        s1 += g;
        s2 = fmaf(g, (yy - mean) * rstd, s2);
      }
    }
    part[0][threadIdx.x] = s1;
    part[1][threadIdx.x] = s2;
    __syncthreads();
    if (grp == 0 && c < N) {
      for (int q = 1; q < groups; ++q) { s1 += part[0][threadIdx.x + q * cw]; s2 += part[1][threadIdx.x + q * cw]; }
      atomicAdd(sums + c, (double)s1);
      atomicAdd(sums + N + c, (double)s2);
    }
    __syncthreads();
  }
}

// The same with four columns per thread (8-byte bf16 / 16-byte fp32 accesses), four rows in flight and at most 256 blocks whose
// 2 N sums leave through LDS as consecutive doubles — prep_vec_kernel of csrc/mlp_gemm.hip (round 5) for bf16 rows: the
// 16-row blocks above end with 2 N same-address fp64 atomics each, which, not the rows, was the kernel's time (66 us at the
// FP shapes of the backbone).
constexpr int kPrepBlocksBf = 256;
__host__ __device__ inline long long prep_vec_rpb_bf(long long R, int C) {
  const long long unit = (long long)(256 / (C / 4)) * 4;
  const long long want = (R + kPrepBlocksBf - 1) / kPrepBlocksBf;
  return (want + unit - 1) / unit * unit;
}
__global__ __launch_bounds__(256) void prep_vec_bf16_kernel(long long M, int N, long long rpb, const bf16 *__restrict__ y,
                                                           const float *__restrict__ gout, const float *__restrict__ fin,
                                                           bf16 *__restrict__ gpre, double *__restrict__ sums) {
  __shared__ float4 part[2][256];
  const long long r0 = (long long)blockIdx.x * rpb;
  if (r0 >= M) return;
  const long long r1 = r0 + rpb < M ? r0 + rpb : M;
  const int C4 = N >> 2;
  const int groups = 256 / C4;
  const int grp = threadIdx.x / C4, c4 = threadIdx.x - grp * C4;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (grp < groups) {
    const float4 mean = reinterpret_cast<const float4 *>(fin)[c4], rstd = reinterpret_cast<const float4 *>(fin + N)[c4];
    const float4 sc = reinterpret_cast<const float4 *>(fin + 2 * N)[c4], sh = reinterpret_cast<const float4 *>(fin + 3 * N)[c4];
    const float ma[4] = {mean.x, mean.y, mean.z, mean.w}, ra[4] = {rstd.x, rstd.y, rstd.z, rstd.w};
    const float sca[4] = {sc.x, sc.y, sc.z, sc.w}, sha[4] = {sh.x, sh.y, sh.z, sh.w};
    const uint2 *Y = reinterpret_cast<const uint2 *>(y) + c4;
    const float4 *G = reinterpret_cast<const float4 *>(gout) + c4;
    uint2 *O = reinterpret_cast<uint2 *>(gpre) + c4;
    for (long long r = r0 + grp; r < r1; r += 4 * groups) {
      uint2 yv[4];
      float4 gv[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long row = r + (long long)u * groups;
        ok[u] = row < r1;
        const size_t o = (size_t)(ok[u] ? row : r) * C4;
        yv[u] = Y[o];
        gv[u] = G[o];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) continue;
        const float ya[4] = {bf_lo(yv[u].x), bf_hi(yv[u].x), bf_lo(yv[u].y), bf_hi(yv[u].y)};
        const float ga[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
        float oa[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g = bf_round(fmaf(ya[j], sca[j], sha[j]) > 0.f ? ga[j] : 0.f);
          oa[j] = g;
          s1[j] += g;
          s2[j] = fmaf(g, (ya[j] - ma[j]) * ra[j], s2[j]);
        }
        O[(size_t)(r + (long long)u * groups) * C4] = uint2{bf_pack(oa[0], oa[1]), bf_pack(oa[2], oa[3])};
      }
    }
  }
  part[0][threadIdx.x] = float4{s1[0], s1[1], s1[2], s1[3]};
  part[1][threadIdx.x] = float4{s2[0], s2[1], s2[2], s2[3]};
  __syncthreads();
  float4 t1 = part[0][threadIdx.x], t2 = part[1][threadIdx.x];
  if (grp == 0) {
    for (int q = 1; q < groups; ++q) {
      const float4 a = part[0][threadIdx.x + q * C4], b = part[1][threadIdx.x + q * C4];
      t1.x += a.x; t1.y += a.y; t1.z += a.z; t1.w += a.w;
      t2.x += b.x; t2.y += b.y; t2.z += b.z; t2.w += b.w;
    }
  }
  __syncthreads();
  float *red = reinterpret_cast<float *>(&part[0][0]);
  if (grp == 0) {
    reinterpret_cast<float4 *>(red)[c4] = t1;
    reinterpret_cast<float4 *>(red + N)[c4] = t2;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * N; t += 256) atomicAdd(sums + t, (double)red[t]);
}

// ReLU(BN(y)) + max over groups of ns rows (+ first arg-max, + raw value there): y bf16 -> fp32 / int32 (R, C)
__global__ __launch_bounds__(256) void bn_relu_rows_max_bf16_kernel(size_t total /* R*C/2 */, int ns, int C,
                                                                   const bf16 *__restrict__ y,
                                                                   const float *__restrict__ fin, float *__restrict__ out,
                                                                   int *__restrict__ arg, float *__restrict__ yraw,
                                                                   const long long *__restrict__ seg) {
  const int CV = C / 2;
  if (seg) {                                         // blockIdx.y = scan: its groups and its (4,C) finalize block
    const size_t g0 = (size_t)(seg[blockIdx.y] / ns);
    total = ((size_t)(seg[blockIdx.y + 1] / ns) - g0) * CV;
    y += g0 * ns * C; out += g0 * C; arg += g0 * C; yraw += g0 * C;
    fin += (size_t)blockIdx.y * 4 * C;
  }
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t r = e / CV;
    const int c = (int)(e - r * CV) * 2;
    const float sc0 = fin[2 * C + c], sc1 = fin[2 * C + c + 1], sh0 = fin[3 * C + c], sh1 = fin[3 * C + c + 1];
    const unsigned *p = (const unsigned *)(y + r * ns * C + c);
    unsigned w = p[0];
    float r0 = bf_lo(w), r1 = bf_hi(w);
    float b0 = fmaxf(fmaf(r0, sc0, sh0), 0.f), b1 = fmaxf(fmaf(r1, sc1, sh1), 0.f);
    int i0 = 0, i1 = 0;
    for (int s = 1; s < ns; ++s) {
      w = p[(size_t)s * CV];
      const float q0 = bf_lo(w), q1 = bf_hi(w);
      const float z0 = fmaxf(fmaf(q0, sc0, sh0), 0.f), z1 = fmaxf(fmaf(q1, sc1, sh1), 0.f);
      if (z0 > b0) { b0 = z0; i0 = s; r0 = q0; }
      if (z1 > b1) { b1 = z1; i1 = s; r1 = q1; }
    }
    out[r * C + c] = b0; out[r * C + c + 1] = b1;
    arg[r * C + c] = i0; arg[r * C + c + 1] = i1;
    yraw[r * C + c] = r0; yraw[r * C + c + 1] = r1;
  }
}

inline unsigned capped_grid(size_t work, unsigned cap = 16384) {
  size_t g = (work + 255) / 256;
  if (g > cap) g = cap;
  return (unsigned)(g ? g : 1);
}

// LDS of one workgroup: 160 KB per CU.  Weights are kept resident whenever the whole footprint (weights + prologue table +
// A chunk / epilogue tile) fits — also when that leaves room for only ONE workgroup per CU: at K = N = 256 (135 KB of
// weights) the resident single launch takes 669 us for 2M rows, two resident 128-column blocks 1221 us, the streamed
// form (weights re-staged per K chunk of every row tile) 2394 us.
constexpr size_t kLdsBudget = 158 * 1024;

inline int gemm_tile_columns(int n) {                // columns a workgroup covers for an n-column (block of the) output
  const int nt = (n + 31) / 32;
  return 32 * (nt <= 4 ? nt : (nt + 1) / 2 * 2);
}

inline size_t gemm_lds_need(int n, int Kp, int pro, int epi) {      // with resident weights
  const int cols = gemm_tile_columns(n);
  size_t tile = (size_t)TM * AP * 2;
  if (epi == EPI_MASK && (size_t)TM * (cols + 8) * 2 > tile) tile = (size_t)TM * (cols + 8) * 2;
  return (size_t)cols * (Kp + 8) * 2 + tile + (pro != PRO_NONE ? 3 * (size_t)Kp * 4 : 0) + (pro == PRO_FIRST ? (size_t)Kp * 16 : 0);
}

template <int NT, int CW, int PRO, int EPI, bool XF32, bool YF32>
int launch_gemm(GemmBf16Args a, hipStream_t s) {
  constexpr int NTT = NT * CW;
  const size_t wbytes_res = (size_t)NTT * 32 * (a.Kp + 8) * 2;
  a.wres = 1;
  if (EPI == EPI_MASK && a.N % 8 != 0) return PN2_EINVAL;
  size_t tile = (size_t)TM * AP * 2;                        // A chunk, aliased by the epilogue tile of EPI_MASK
  if (EPI == EPI_MASK && (size_t)TM * (NTT * 32 + 8) * 2 > tile) tile = (size_t)TM * (NTT * 32 + 8) * 2;
  const size_t fixed = tile + (PRO != PRO_NONE ? 3 * (size_t)a.Kp * 4 : 0) + (PRO == PRO_FIRST ? (size_t)a.Kp * 16 : 0);
  if (fixed + wbytes_res > kLdsBudget) a.wres = 0;          // (the host wrapper picks column blocks that fit)
  const size_t wbytes = a.wres ? wbytes_res : (size_t)NTT * 32 * AP * 2;
  size_t lds = fixed + wbytes;
  if (lds > 160 * 1024) return PN2_EINVAL;
  const size_t red = (size_t)2 * 4 * NTT * 32 * 4;
  if (lds < red) lds = red;
  auto kfn = mlp_gemm_bf16_kernel<NT, CW, PRO, EPI, XF32, YF32>;
  static bool big_lds = false;                            // per template instance: opt in to > 64 KB of LDS once
  if (lds > 64 * 1024 && !big_lds) {
    if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return pn2_check_launch();
    big_lds = true;
  }
  const long long ntiles = ((a.seg ? a.seg_max : a.M) + TM - 1) / TM;
  long long grid = lds > 80 * 1024 ? 256 : 512;             // persistent: as many workgroups per CU as the LDS admits
  a.vcap = (int)grid;
  const int nseg = a.seg ? a.nseg : 1;
  if (nseg > 1) grid = (grid + nseg - 1) / nseg;            // (the scans share the chip)
  if (grid > ntiles) grid = ntiles;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kfn, dim3((unsigned)grid, (unsigned)nseg), dim3(256 * CW), lds, s, a);
  return pn2_check_launch();
}

template <int PRO, int EPI, bool XF32, bool YF32>
int dispatch_nt(const GemmBf16Args &a, hipStream_t s) {
  const int nt = (a.N + 31) / 32;
  switch (nt) {
    case 1: return launch_gemm<1, 1, PRO, EPI, XF32, YF32>(a, s);
    case 2: return launch_gemm<2, 1, PRO, EPI, XF32, YF32>(a, s);
    case 3: return launch_gemm<3, 1, PRO, EPI, XF32, YF32>(a, s);
    case 4:
      // resident weights that leave room for ONE workgroup per CU (K = 256 -> N = 128: 68 KB + tile): eight waves on the
      // tile instead of four — the staging / patch / epilogue phases between the barriers get twice the threads and every
      // SIMD a second wave (same column sums: a column's partial sums do not depend on CW)
      if (!XF32 && gemm_lds_need(a.N, a.Kp, PRO, EPI) > 80 * 1024) return launch_gemm<2, 2, PRO, EPI, XF32, YF32>(a, s);
      return launch_gemm<4, 1, PRO, EPI, XF32, YF32>(a, s);
    case 5: case 6: return launch_gemm<3, 2, PRO, EPI, XF32, YF32>(a, s);
    case 7: case 8: return launch_gemm<4, 2, PRO, EPI, XF32, YF32>(a, s);
    case 9: case 10: return launch_gemm<5, 2, PRO, EPI, XF32, YF32>(a, s);
    default: return PN2_EINVAL;
  }
}

template <int NTW, int KTB, int MT, int GMODE, int AMODE, bool XF32>
int launch_wgrad(const WgradBf16Args &a, hipStream_t s) {
  constexpr int MP = MT + 8, NB = 4 * NTW * 32, KB = KTB * 32;
  const size_t lds = (size_t)(NB + KB) * MP * 2 + (size_t)(3 * NB + 2 * KB) * 4;
  auto kfn = mlp_wgrad_bf16_kernel<NTW, KTB, MT, GMODE, AMODE, XF32>;
  static bool big_lds = false;
  if (lds > 64 * 1024 && !big_lds) {
    if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return pn2_check_launch();
    big_lds = true;
  }
  const unsigned kblocks = (unsigned)((a.K + KB - 1) / KB);
  const long long ntiles = ((a.seg ? a.seg_max : a.M) + MT - 1) / MT;
  const int nseg = a.seg ? a.nseg : 1;
  long long gx = 512 / kblocks / nseg;
  if (gx < 1) gx = 1;
  {  // few rows: the N x KB block of float atomics every workgroup ends with outweighs its rows (see pn2_mlp_wgrad)
    long long cap = (a.seg ? a.seg_max : a.M) / 512;
    if (cap < 64) cap = 64;
    if (gx > cap) gx = cap;
  }
  if (gx > ntiles) gx = ntiles;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(kfn, dim3((unsigned)gx, kblocks, (unsigned)nseg), dim3(256), lds, s, a);
  return pn2_check_launch();
}

template <int GMODE, int AMODE, bool XF32>
int dispatch_wgrad(const WgradBf16Args &a, hipStream_t s) {
  if (a.N <= 128) {
    if (a.K <= 32) return launch_wgrad<1, 1, 128, GMODE, AMODE, XF32>(a, s);
    if (a.K <= 64) return launch_wgrad<1, 2, 128, GMODE, AMODE, XF32>(a, s);
    // every K block re-reads the gy / y_l tiles: K = 131 (128 features + xyz) as ONE 160-wide block instead of 128 + 3
    // (0.27 -> 0.22 ms at 1M rows; K = 259 as 160 + 99 instead of 128 + 128 + 3 measured slower next to the sampling kernels)
    if constexpr (GMODE == PRO_GY) {                 // (the pooled-gradient form has no registers for a fifth tile)
      if (a.K > 128 && a.K <= 160) return launch_wgrad<1, 5, 128, GMODE, AMODE, XF32>(a, s);
    }
    // K = 195 (the scene-graph encoders' 192 features + xyz): 128 + 67 read gy / y_l twice; one 224-wide block on 64-row
    // tiles (112 accumulator registers): 18.9M rows 4.86 -> 3.89 ms
    if (a.K > 160 && a.K <= 224 && !XF32) return launch_wgrad<1, 7, 64, GMODE, AMODE, XF32>(a, s);
    return launch_wgrad<1, 4, 128, GMODE, AMODE, XF32>(a, s);
  }
  if (a.N <= 256) {
    if (a.K <= 32) return launch_wgrad<2, 1, 64, GMODE, AMODE, XF32>(a, s);
    // 64 < K <= 128 (the pooled 128 -> 256 layers): ALL of K in one block — with two 64-wide blocks the 256-wide gy / y_l
    // tiles, two thirds of the kernel's bytes, were read once per block (128 accumulator registers per lane instead of 64)
    if (a.K > 64 && a.K <= 128) return launch_wgrad<2, 4, 64, GMODE, AMODE, XF32>(a, s);   // (wider K: small M, the blocks are the parallelism)
    return launch_wgrad<2, 2, 64, GMODE, AMODE, XF32>(a, s);
  }
  if (a.N <= 384) {
    if (a.K <= 32) return launch_wgrad<3, 1, 64, GMODE, AMODE, XF32>(a, s);
    return launch_wgrad<3, 2, 64, GMODE, AMODE, XF32>(a, s);
  }
  return PN2_EINVAL;
}

int pn2_mlp_gemm_bf16_block(GemmBf16Args a, int n0, int nb, int ys, int pro, int epi, int x_f32, int y_f32, hipStream_t s);

}  // namespace

namespace {
struct SegTab { const long long *ptr; int nseg; long long max_rows; int pstride; };

int gemm_bf16_impl(long long M, int K, int N, int pro, int epi, int x_f32, int y_f32, int ldx, int ldy,
                   const void *X, const void *X2, const float *p0, const float *p1, const float *p2,
                   const int *arg, const float *gP, int ns, const float *W, void *Y, double *stats,
                   const void *Yprev, const float *e_fin, const SegTab &sg, void *stream) {
  if (M < 0 || K <= 0 || N <= 0 || K > 4096 || ldx < K || ldy < N) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!W || !Y) return PN2_ENULL;
  if (!x_f32 && (ldx % 8 != 0 || ((uintptr_t)X & 15) || ((uintptr_t)X2 & 15))) return PN2_EINVAL;   // 16-byte row groups
  if (x_f32 && pro != PRO_NONE) return PN2_EINVAL;
  if (pro != PRO_POOLG && !X) return PN2_ENULL;
  if (pro != PRO_NONE && (!p0 || !p1)) return PN2_ENULL;
  if ((pro == PRO_GY || pro == PRO_POOLG) && (!X2 || !p2)) return PN2_ENULL;
  if (pro == PRO_POOLG && (!arg || !gP || ns <= 0)) return PN2_ENULL;
  if (epi != EPI_NONE && !stats) return PN2_ENULL;
  if (epi == EPI_MASK && (!Yprev || !e_fin)) return PN2_ENULL;
  if (y_f32 && epi != EPI_NONE) return PN2_EINVAL;
  GemmBf16Args a;
  a.X = X; a.X2 = (const bf16 *)X2; a.p0 = p0; a.p1 = p1; a.p2 = p2; a.arg = arg; a.gP = gP; a.W = W; a.Y = Y;
  a.stats = stats; a.Yprev = (const bf16 *)Yprev; a.e_fin = e_fin; a.M = M; a.K = K; a.N = N; a.ldx = ldx; a.ldy = ldy;
  a.ns = ns; a.Kp = (K + KC - 1) / KC * KC; a.wres = 0; a.Nfull = N;
  a.seg = sg.ptr; a.nseg = sg.nseg; a.seg_max = sg.max_rows; a.pstride = sg.pstride; a.sstride = 2 * N; a.estride = 4 * N;
  a.vcap = 0;
  hipStream_t s = (hipStream_t)stream;
  // Column blocks (A re-read per block, the second time from L2) in two cases: outputs wider than the 320 columns a
  // workgroup covers (the input gradient of a 512-column FP stack), and weights that do not fit LDS at full width — a
  // non-resident weight tile is re-staged per 64-column K chunk of EVERY row tile, which ran the 256 x 256 layers at the
  // fp32 kernel's speed (M = 524k: 632 us, 90 us of HBM time); the widest multiple of 32 columns whose footprint (resident
  // bf16 weights at pitch K + 8, A chunk / epilogue tile, prologue table) fits the LDS budget is taken.
  int block = N > 320 ? 256 : N;
  while (block > 32 && gemm_lds_need(block, a.Kp, pro, epi) > kLdsBudget) block = (block - 1) / 32 * 32;
  if (block < N) {
    const int ys = y_f32 ? 4 : 2;
    for (int n0 = 0; n0 < N; n0 += block) {
      const int nb = N - n0 < block ? N - n0 : block;
      const int rc = pn2_mlp_gemm_bf16_block(a, n0, nb, ys, pro, epi, x_f32, y_f32, s);
      if (rc != PN2_OK) return rc;
    }
    return PN2_OK;
  }
  return pn2_mlp_gemm_bf16_block(a, 0, N, y_f32 ? 4 : 2, pro, epi, x_f32, y_f32, s);
}
}  // namespace

extern "C" int pn2_mlp_gemm_bf16(long long M, int K, int N, int pro, int epi, int x_f32, int y_f32, int ldx, int ldy,
                                 const void *X, const void *X2, const float *p0, const float *p1, const float *p2,
                                 const int *arg, const float *gP, int ns, const float *W, void *Y, double *stats,
                                 const void *Yprev, const float *e_fin, void *stream) {
  return gemm_bf16_impl(M, K, N, pro, epi, x_f32, y_f32, ldx, ldy, X, X2, p0, p1, p2, arg, gP, ns, W, Y, stats, Yprev, e_fin,
                        SegTab{nullptr, 1, 0, 0}, stream);
}

// The SECOND layer of a bf16 stack with the first one re-formed from its input rows (the bf16 counterpart of pn2_mlp_gemm_first):
// X0 (M, 8) bf16 rows (K0 <= 8 real columns, zero padded), W0 (K, K0) fp32, fin0 (4, K) = the first layer's mean | rstd |
// scale | shift, W (N, K) fp32 -> Y (M, N) bf16 = relu(bn_0(X0 W0^T)) W^T with the column sums of the ROUNDED Y, Y^2 in stats.
// y_0 is never stored (its batch statistics: pn2_rows_gram_bf16 + pn2_first_layer_stats).  K <= 128 a multiple of 8; N in {32, 64, 128}.
extern "C" int pn2_mlp_gemm_first_bf16_supported(int K0, int K, int N) {
  return K0 >= 1 && K0 <= 8 && K >= 8 && K <= 128 && K % 8 == 0 && (N == 32 || N == 64 || N == 128);
}

extern "C" int pn2_mlp_gemm_first_bf16(long long M, int K0, int K, int N, const void *X0, const float *W0, const float *fin0,
                                       const float *W, void *Y, double *stats, void *stream) {
  if (M < 0 || !pn2_mlp_gemm_first_bf16_supported(K0, K, N)) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!X0 || !W0 || !fin0 || !W || !Y || !stats) return PN2_ENULL;
  if (((uintptr_t)X0 & 15) || ((uintptr_t)Y & 15)) return PN2_EINVAL;
  GemmBf16Args a = {};
  a.X = X0; a.p0 = fin0 + 2 * (size_t)K; a.p1 = fin0 + 3 * (size_t)K; a.W = W; a.Y = Y; a.stats = stats; a.M = M; a.K = K; a.N = N;
  a.ldx = 8; a.ldy = N; a.Kp = (K + KC - 1) / KC * KC; a.wres = 0; a.Nfull = N;
  a.seg = nullptr; a.nseg = 1; a.sstride = 2 * N; a.estride = 4 * N; a.W0 = W0; a.K0 = K0;
  hipStream_t s = (hipStream_t)stream;
  if (N == 32) return launch_gemm<1, 1, PRO_FIRST, EPI_STATS, false, false>(a, s);
  if (N == 64) return launch_gemm<2, 1, PRO_FIRST, EPI_STATS, false, false>(a, s);
  return launch_gemm<4, 1, PRO_FIRST, EPI_STATS, false, false>(a, s);
}

// The max-pooled last layer of a stack on the bf16 path WITHOUT its (M, N) output (the bf16 counterpart of pn2_mlp_gemm_pool):
// X (M, ldx) bf16 = y_{L-1}, p0 / p1 its BatchNorm scale / shift, Wf (N, K) fp32 with the rows of negative-gamma columns negated
// (pn2_pool_flip_rows; sgn (N)), ns rows per group (16, 32, 64 or 128; M % ns == 0) -> pmax / parg (M / min(ns, 32), N): maximum
// of the fp32 accumulators over each partial group and its row, stats (2, N) += column sums of y_L and y_L^2 (un-flipped, of
// the fp32 accumulators: nothing is rounded to bf16 because nothing is stored).  pn2_pool_finalize turns pmax / parg into the
// pooled activations; the backward re-forms y_L from y_{L-1} (pn2_mlp_bwd_bf16_pool).  N in {64, 128}, K <= 128.
extern "C" int pn2_mlp_gemm_pool_bf16_supported(int K, int N, int ns) {
  return (N == 64 || N == 128) && K >= 1 && K <= 128 && (ns == 16 || ns == 32 || ns == 64 || ns == 128);
}

extern "C" int pn2_mlp_gemm_pool_bf16(long long M, int K, int N, int ldx, const void *X, const float *p0, const float *p1,
                                      const float *Wf, const float *sgn, int ns, double *stats, float *pmax, int *parg,
                                      void *stream) {
  if (M < 0 || !pn2_mlp_gemm_pool_bf16_supported(K, N, ns) || ldx < K || ldx % 8 != 0) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (M % ns != 0) return PN2_EINVAL;
  if (!X || !p0 || !p1 || !Wf || !sgn || !stats || !pmax || !parg) return PN2_ENULL;
  if ((uintptr_t)X & 15) return PN2_EINVAL;
  GemmBf16Args a = {};
  a.X = X; a.p0 = p0; a.p1 = p1; a.W = Wf; a.Y = nullptr; a.stats = stats; a.M = M; a.K = K; a.N = N; a.ldx = ldx; a.ldy = N;
  a.ns = ns; a.Kp = (K + KC - 1) / KC * KC; a.wres = 0; a.Nfull = N;
  a.seg = nullptr; a.nseg = 1; a.seg_max = 0; a.pstride = 0; a.sstride = 2 * N; a.estride = 4 * N; a.vcap = 0;
  a.pmax = pmax; a.parg = parg; a.sgn = sgn;
  hipStream_t s = (hipStream_t)stream;
  return N == 64 ? launch_gemm<2, 1, PRO_BNRELU, EPI_POOL, false, false>(a, s)
                 : launch_gemm<4, 1, PRO_BNRELU, EPI_POOL, false, false>(a, s);
}

// Batched scans with per-scan BatchNorm statistics in ONE launch: the M rows are nseg scans, scan s = rows
// [seg[s], seg[s+1]) (device array of nseg + 1 offsets, multiples of ns for PRO_POOLG; seg_max = the longest scan, for the
// grid); every per-channel operand is an array of per-scan blocks — p0 / p1 / p2 at a pitch of `pstride` floats (forward:
// the scale / shift rows of the (S,4,K) finalize blocks, pstride = 4K; backward: the (S,3,K) constants, 3K), stats
// (S,2,N), e_fin (S,4,N).  Same arithmetic per scan as nseg separate calls (tiles never straddle two scans).
extern "C" int pn2_mlp_gemm_bf16_seg(long long M, int K, int N, int pro, int epi, int x_f32, int y_f32, int ldx, int ldy,
                                     const void *X, const void *X2, const float *p0, const float *p1, const float *p2,
                                     int pstride, const int *arg, const float *gP, int ns, const float *W, void *Y,
                                     double *stats, const void *Yprev, const float *e_fin, const long long *seg, int nseg,
                                     long long seg_max, void *stream) {
  if (!seg) return PN2_ENULL;
  if (nseg < 1 || nseg > 65535 || seg_max < 0 || seg_max > M) return PN2_EINVAL;
  return gemm_bf16_impl(M, K, N, pro, epi, x_f32, y_f32, ldx, ldy, X, X2, p0, p1, p2, arg, gP, ns, W, Y, stats, Yprev, e_fin,
                        SegTab{seg, nseg, seg_max, pstride}, stream);
}

namespace {
int pn2_mlp_gemm_bf16_block(GemmBf16Args a, int n0, int nb, int ys, int pro, int epi, int x_f32, int y_f32, hipStream_t s) {
  a.W += (size_t)n0 * a.K;
  a.Y = (char *)a.Y + (size_t)n0 * ys;
  if (a.Yprev) a.Yprev += n0;
  if (a.e_fin) a.e_fin += n0;
  if (a.stats) a.stats += n0;
  a.N = nb;
  if (x_f32) {
    if (epi == EPI_STATS) return dispatch_nt<PRO_NONE, EPI_STATS, true, false>(a, s);
    if (epi == EPI_NONE && !y_f32) return dispatch_nt<PRO_NONE, EPI_NONE, true, false>(a, s);
    return PN2_EINVAL;
  }
  switch (pro * 8 + epi * 2 + (y_f32 ? 1 : 0)) {
    case PRO_NONE * 8 + EPI_NONE * 2: return dispatch_nt<PRO_NONE, EPI_NONE, false, false>(a, s);
    case PRO_NONE * 8 + EPI_STATS * 2: return dispatch_nt<PRO_NONE, EPI_STATS, false, false>(a, s);
    case PRO_BNRELU * 8 + EPI_NONE * 2: return dispatch_nt<PRO_BNRELU, EPI_NONE, false, false>(a, s);
    case PRO_BNRELU * 8 + EPI_STATS * 2: return dispatch_nt<PRO_BNRELU, EPI_STATS, false, false>(a, s);
    case PRO_GY * 8 + EPI_MASK * 2: return dispatch_nt<PRO_GY, EPI_MASK, false, false>(a, s);
    case PRO_POOLG * 8 + EPI_MASK * 2: return dispatch_nt<PRO_POOLG, EPI_MASK, false, false>(a, s);
    case PRO_GY * 8 + EPI_NONE * 2: return dispatch_nt<PRO_GY, EPI_NONE, false, false>(a, s);        // bf16 gradient rows of a
    case PRO_POOLG * 8 + EPI_NONE * 2: return dispatch_nt<PRO_POOLG, EPI_NONE, false, false>(a, s);  // grouped first layer
    case PRO_GY * 8 + EPI_NONE * 2 + 1: return dispatch_nt<PRO_GY, EPI_NONE, false, true>(a, s);
    case PRO_POOLG * 8 + EPI_NONE * 2 + 1: return dispatch_nt<PRO_POOLG, EPI_NONE, false, true>(a, s);
    default: return PN2_EINVAL;
  }
}
}  // namespace

namespace {
int wgrad_bf16_impl(long long M, int N, int K, int gmode, int amode, int x_f32, int ldx, const void *G,
                    const void *Yl, const float *consts, const int *arg, const float *gP, int ns,
                    const void *X, const float *a_fin, float *dW, const SegTab &sg, void *stream) {
  if (M < 0 || N <= 0 || K <= 0 || N > 384 || N % 8 != 0 || ldx < K) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!Yl || !consts || !X || !dW) return PN2_ENULL;
  if (gmode == PRO_GY && !G) return PN2_ENULL;
  if (gmode == PRO_POOLG && (!arg || !gP || ns <= 0)) return PN2_ENULL;
  if (gmode != PRO_GY && gmode != PRO_POOLG) return PN2_EINVAL;
  if (amode == PRO_BNRELU && (!a_fin || x_f32)) return PN2_EINVAL;
  if (!x_f32 && (ldx % 8 != 0 || ((uintptr_t)X & 15))) return PN2_EINVAL;
  if (((uintptr_t)Yl & 15) || ((uintptr_t)G & 15)) return PN2_EINVAL;
  WgradBf16Args a;
  a.G = (const bf16 *)G; a.Yl = (const bf16 *)Yl; a.consts = consts; a.arg = arg; a.gP = gP; a.X = X; a.a_fin = a_fin;
  a.dW = dW; a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ns = ns;
  a.seg = sg.ptr; a.nseg = sg.nseg; a.seg_max = sg.max_rows;
  hipStream_t s = (hipStream_t)stream;
  if (gmode == PRO_GY) {
    if (amode == PRO_BNRELU) return dispatch_wgrad<PRO_GY, PRO_BNRELU, false>(a, s);
    return x_f32 ? dispatch_wgrad<PRO_GY, PRO_NONE, true>(a, s) : dispatch_wgrad<PRO_GY, PRO_NONE, false>(a, s);
  }
  if (amode == PRO_BNRELU) return dispatch_wgrad<PRO_POOLG, PRO_BNRELU, false>(a, s);
  return x_f32 ? dispatch_wgrad<PRO_POOLG, PRO_NONE, true>(a, s) : dispatch_wgrad<PRO_POOLG, PRO_NONE, false>(a, s);
}
}  // namespace

extern "C" int pn2_mlp_wgrad_bf16(long long M, int N, int K, int gmode, int amode, int x_f32, int ldx, const void *G,
                                  const void *Yl, const float *consts, const int *arg, const float *gP, int ns,
                                  const void *X, const float *a_fin, float *dW, void *stream) {
  return wgrad_bf16_impl(M, N, K, gmode, amode, x_f32, ldx, G, Yl, consts, arg, gP, ns, X, a_fin, dW, SegTab{nullptr, 1, 0, 0},
                         stream);
}

// pn2_mlp_wgrad_bf16 over nseg scans (see pn2_mlp_gemm_bf16_seg): consts (S,3,N), a_fin (S,4,K); dW [N][K] receives the SUM
// over the scans (the step's loss is the mean of the scans' losses; its 1/S is part of the incoming gradient).
extern "C" int pn2_mlp_wgrad_bf16_seg(long long M, int N, int K, int gmode, int amode, int x_f32, int ldx, const void *G,
                                      const void *Yl, const float *consts, const int *arg, const float *gP, int ns,
                                      const void *X, const float *a_fin, float *dW, const long long *seg, int nseg,
                                      long long seg_max, void *stream) {
  if (!seg) return PN2_ENULL;
  if (nseg < 1 || nseg > 65535 || seg_max < 0 || seg_max > M) return PN2_EINVAL;
  return wgrad_bf16_impl(M, N, K, gmode, amode, x_f32, ldx, G, Yl, consts, arg, gP, ns, X, a_fin, dW,
                         SegTab{seg, nseg, seg_max, 0}, stream);
}

extern "C" int pn2_mlp_bwd_bf16_supported(int N, int K) {
  if (N % 32 || K % 32 || N < 32 || K < 32 || N > 128 || K > 128) return 0;
  const int c = (N / 32) * 8 + K / 32;
  return c == 9 || c == 10 || c == 17 || c == 18 || c == 20 || c == 34 || c == 36 || c == 27 || c == 35 || c == 28;
}

extern "C" int pn2_mlp_bwd_bf16(long long M, int N, int K, int gmode, const void *G, const void *Yl, const float *consts,
                                const int *arg, const float *gP, int ns, const float *Wt, const void *Yprev,
                                const float *a_fin, void *Gout, double *sums, float *dW, void *stream) {
  if (M < 0 || !pn2_mlp_bwd_bf16_supported(N, K)) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!Yl || !consts || !Wt || !Yprev || !a_fin || !Gout || !sums || !dW) return PN2_ENULL;
  if (gmode == PRO_GY && !G) return PN2_ENULL;
  if (gmode == PRO_POOLG && (!arg || !gP || ns <= 0)) return PN2_ENULL;
  if (gmode != PRO_GY && gmode != PRO_POOLG) return PN2_EINVAL;
  if (((uintptr_t)Yl & 15) || ((uintptr_t)G & 15) || ((uintptr_t)Yprev & 15) || ((uintptr_t)Gout & 15)) return PN2_EINVAL;
  BwdBf16Args a;
  a.G = (const bf16 *)G; a.Yl = (const bf16 *)Yl; a.consts = consts; a.arg = arg; a.gP = gP; a.Wt = Wt;
  a.Yprev = (const bf16 *)Yprev; a.a_fin = a_fin; a.Gout = (bf16 *)Gout; a.sums = sums; a.dW = dW; a.M = M; a.N = N; a.K = K;
  a.ns = ns; a.X = nullptr; a.P1 = nullptr; a.K0 = 0;
  a.seg = nullptr; a.nseg = 1; a.seg_max = 0;
  return gmode == PRO_GY ? dispatch_bwd<PRO_GY>(a, (hipStream_t)stream) : dispatch_bwd<PRO_POOLG>(a, (hipStream_t)stream);
}

// ... of a max-pooled LAST layer whose output pn2_mlp_gemm_pool_bf16 never stored: y_l is re-formed from y_{l-1} (Yprev) and
// Wt inside the kernel (RECOMP above); consts / arg / gP / sums / dW as pn2_mlp_bwd_bf16 with gmode PRO_POOLG.  N in {64, 128}.
extern "C" int pn2_mlp_bwd_bf16_pool_supported(int N, int K) { return (N == 64 || N == 128) && (K == 32 || K == 64 || K == 128) && pn2_mlp_bwd_bf16_supported(N, K); }

extern "C" int pn2_mlp_bwd_bf16_pool(long long M, int N, int K, const float *consts, const int *arg, const float *gP, int ns,
                                     const float *Wt, const void *Yprev, const float *a_fin, void *Gout, double *sums,
                                     float *dW, void *stream) {
  if (M < 0 || !pn2_mlp_bwd_bf16_pool_supported(N, K) || ns < 16) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!consts || !arg || !gP || !Wt || !Yprev || !a_fin || !Gout || !sums || !dW) return PN2_ENULL;
  if (((uintptr_t)Yprev & 15) || ((uintptr_t)Gout & 15)) return PN2_EINVAL;
  BwdBf16Args a;
  a.G = nullptr; a.Yl = nullptr; a.consts = consts; a.arg = arg; a.gP = gP; a.Wt = Wt;
  a.Yprev = (const bf16 *)Yprev; a.a_fin = a_fin; a.Gout = (bf16 *)Gout; a.sums = sums; a.dW = dW; a.M = M; a.N = N; a.K = K;
  a.ns = ns; a.X = nullptr; a.P1 = nullptr; a.K0 = 0;
  a.seg = nullptr; a.nseg = 1; a.seg_max = 0;
  hipStream_t s = (hipStream_t)stream;
  switch ((N / 32) * 8 + K / 32) {
    case 2 * 8 + 1: return launch_bwd<2, 1, PRO_POOLG, false, true>(a, s);
    case 2 * 8 + 2: return launch_bwd<2, 2, PRO_POOLG, false, true>(a, s);
    case 2 * 8 + 4: return launch_bwd<2, 4, PRO_POOLG, false, true>(a, s);
    case 4 * 8 + 2: return launch_bwd<4, 2, PRO_POOLG, false, true>(a, s);
    case 4 * 8 + 4: return launch_bwd<4, 4, PRO_POOLG, false, true>(a, s);
    default: return PN2_EINVAL;
  }
}

// pn2_mlp_bwd_bf16 over nseg scans (see pn2_mlp_gemm_bf16_seg): consts (S,3,N), a_fin (S,4,K), sums (S,2,K); dW = the SUM.
extern "C" int pn2_mlp_bwd_bf16_seg(long long M, int N, int K, int gmode, const void *G, const void *Yl, const float *consts,
                                    const int *arg, const float *gP, int ns, const float *Wt, const void *Yprev,
                                    const float *a_fin, void *Gout, double *sums, float *dW, const long long *seg, int nseg,
                                    long long seg_max, void *stream) {
  if (M < 0 || !pn2_mlp_bwd_bf16_supported(N, K)) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!Yl || !consts || !Wt || !Yprev || !a_fin || !Gout || !sums || !dW || !seg) return PN2_ENULL;
  if (nseg < 1 || nseg > 65535 || seg_max < 0 || seg_max > M) return PN2_EINVAL;
  if (gmode == PRO_GY && !G) return PN2_ENULL;
  if (gmode == PRO_POOLG && (!arg || !gP || ns <= 0)) return PN2_ENULL;
  if (gmode != PRO_GY && gmode != PRO_POOLG) return PN2_EINVAL;
  if (((uintptr_t)Yl & 15) || ((uintptr_t)G & 15) || ((uintptr_t)Yprev & 15) || ((uintptr_t)Gout & 15)) return PN2_EINVAL;
  BwdBf16Args a;
  a.G = (const bf16 *)G; a.Yl = (const bf16 *)Yl; a.consts = consts; a.arg = arg; a.gP = gP; a.Wt = Wt;
  a.Yprev = (const bf16 *)Yprev; a.a_fin = a_fin; a.Gout = (bf16 *)Gout; a.sums = sums; a.dW = dW; a.M = M; a.N = N; a.K = K;
  a.ns = ns; a.X = nullptr; a.P1 = nullptr; a.K0 = 0;
  a.seg = seg; a.nseg = nseg; a.seg_max = seg_max;
  return gmode == PRO_GY ? dispatch_bwd<PRO_GY>(a, (hipStream_t)stream) : dispatch_bwd<PRO_POOLG>(a, (hipStream_t)stream);
}

// First-layer fold on the bf16 path: as pn2_mlp_bwd_bf16 for the layer above a stack's FIRST layer whose input rows X
// ([M][8] bf16: K0 <= 8 columns, zero padded) need no gradient — the masked input gradient is not stored, P1 [K][K0] +=
// its product with X is reduced instead (pn2_first_layer_dw then needs no pass over g and y_0).  N, K in {32, 64}.
extern "C" int pn2_mlp_bwd_bf16_fold_supported(int N, int K, int K0) {
  return (N == 32 || N == 64) && (K == 32 || K == 64) && K0 >= 1 && K0 <= 8;
}
extern "C" int pn2_mlp_bwd_bf16_fold(long long M, int N, int K, int gmode, const void *G, const void *Yl, const float *consts,
                                     const int *arg, const float *gP, int ns, const float *Wt, const void *Yprev,
                                     const float *a_fin, const void *X, int K0, double *sums, float *dW, float *P1,
                                     void *stream) {
  if (M < 0 || !pn2_mlp_bwd_bf16_fold_supported(N, K, K0)) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!Yl || !consts || !Wt || !Yprev || !a_fin || !X || !sums || !dW || !P1) return PN2_ENULL;
  if (gmode == PRO_GY && !G) return PN2_ENULL;
  if (gmode == PRO_POOLG && (!arg || !gP || ns <= 0)) return PN2_ENULL;
  if (gmode != PRO_GY && gmode != PRO_POOLG) return PN2_EINVAL;
  if (((uintptr_t)Yl & 15) || ((uintptr_t)G & 15) || ((uintptr_t)Yprev & 15) || ((uintptr_t)X & 15)) return PN2_EINVAL;
  BwdBf16Args a;
  a.G = (const bf16 *)G; a.Yl = (const bf16 *)Yl; a.consts = consts; a.arg = arg; a.gP = gP; a.Wt = Wt;
  a.Yprev = (const bf16 *)Yprev; a.a_fin = a_fin; a.Gout = nullptr; a.sums = sums; a.dW = dW; a.M = M; a.N = N; a.K = K;
  a.ns = ns; a.X = (const bf16 *)X; a.P1 = P1; a.K0 = K0;
  a.seg = nullptr; a.nseg = 1; a.seg_max = 0;
  return gmode == PRO_GY ? dispatch_bwd_fold<PRO_GY>(a, (hipStream_t)stream)
                         : dispatch_bwd_fold<PRO_POOLG>(a, (hipStream_t)stream);
}

// ... when the first layer's output was NOT stored (pn2_mlp_gemm_first_bf16): y_0 = X W0^T is re-formed per tile from the input
// rows the fold reads anyway; W0 (K, K0) fp32 replaces Yprev.
extern "C" int pn2_mlp_bwd_bf16_fold_first(long long M, int N, int K, int gmode, const void *G, const void *Yl, const float *consts,
                                           const int *arg, const float *gP, int ns, const float *Wt, const float *W0,
                                           const float *a_fin, const void *X, int K0, double *sums, float *dW, float *P1,
                                           void *stream) {
  if (M < 0 || !pn2_mlp_bwd_bf16_fold_supported(N, K, K0)) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!Yl || !consts || !Wt || !W0 || !a_fin || !X || !sums || !dW || !P1) return PN2_ENULL;
  if (gmode == PRO_GY && !G) return PN2_ENULL;
  if (gmode == PRO_POOLG && (!arg || !gP || ns <= 0)) return PN2_ENULL;
  if (gmode != PRO_GY && gmode != PRO_POOLG) return PN2_EINVAL;
  if (((uintptr_t)Yl & 15) || ((uintptr_t)G & 15) || ((uintptr_t)X & 15)) return PN2_EINVAL;
  BwdBf16Args a;
  a.G = (const bf16 *)G; a.Yl = (const bf16 *)Yl; a.consts = consts; a.arg = arg; a.gP = gP; a.Wt = Wt;
  a.Yprev = nullptr; a.a_fin = a_fin; a.Gout = nullptr; a.sums = sums; a.dW = dW; a.M = M; a.N = N; a.K = K;
  a.ns = ns; a.X = (const bf16 *)X; a.P1 = P1; a.K0 = K0; a.W0 = W0;
  a.seg = nullptr; a.nseg = 1; a.seg_max = 0;
  hipStream_t s = (hipStream_t)stream;
#define PN2_FY(NTN_, KTK_)                                                                                   \
  (gmode == PRO_GY ? launch_bwd<NTN_, KTK_, PRO_GY, true, false, true>(a, s)                                  \
                   : launch_bwd<NTN_, KTK_, PRO_POOLG, true, false, true>(a, s))
  switch ((N / 32) * 8 + K / 32) {
    case 1 * 8 + 1: return PN2_FY(1, 1);
    case 1 * 8 + 2: return PN2_FY(1, 2);
    case 2 * 8 + 1: return PN2_FY(2, 1);
    case 2 * 8 + 2: return PN2_FY(2, 2);
    default: return PN2_EINVAL;
  }
#undef PN2_FY
}

namespace {
// gram[K0*K0] += X^T X, gram[K0*K0 + k] += column sums of X for bf16 rows of pitch 8 (pn2_rows_gram of the fp32 path)
__global__ __launch_bounds__(256) void rows_gram_bf16_kernel(long long M, int K0, const bf16 *__restrict__ X,
                                                             double *__restrict__ gram) {
  __shared__ double part[4][8 * 8 + 8];
  double ds[8][8], dc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    dc[i] = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) ds[i][j] = 0.0;
  }
  float sf[8][8], cf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    cf[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sf[i][j] = 0.f;
  }
  int n = 0;
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long long)gridDim.x * 256) {
    const u32x4 w = *(const u32x4 *)(X + (size_t)r * 8);
    float x[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[2 * e] = bf_lo(w[e]); x[2 * e + 1] = bf_hi(w[e]); }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      cf[i] += x[i];
#pragma unroll
      for (int j = i; j < 8; ++j) sf[i][j] = fmaf(x[i], x[j], sf[i][j]);
    }
    if (++n == 64) {
      n = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        dc[i] += (double)cf[i]; cf[i] = 0.f;
#pragma unroll
        for (int j = i; j < 8; ++j) { ds[i][j] += (double)sf[i][j]; sf[i][j] = 0.f; }
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    double v = dc[i] + (double)cf[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) part[wave][64 + i] = v;
#pragma unroll
    for (int j = i; j < 8; ++j) {
      double w2 = ds[i][j] + (double)sf[i][j];
      for (int o = 32; o > 0; o >>= 1) w2 += __shfl_xor(w2, o);
      if (lane == 0) part[wave][i * 8 + j] = w2;
    }
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < 72) {
    const int i = t < 64 ? t / 8 : t - 64, j = t < 64 ? t % 8 : 0;
    if (t >= 64) {
      if (i < K0) atomicAdd(gram + K0 * K0 + i, part[0][t] + part[1][t] + part[2][t] + part[3][t]);
    } else if (i < K0 && j < K0) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      const int u = lo * 8 + hi;
      atomicAdd(gram + i * K0 + j, part[0][u] + part[1][u] + part[2][u] + part[3][u]);
    }
  }
}
}  // namespace

extern "C" int pn2_rows_gram_bf16(long long M, int K0, const void *X, double *gram, void *stream) {
  if (M < 0 || K0 < 1 || K0 > 8) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!X || !gram || ((uintptr_t)X & 15)) return PN2_ENULL;
  long long blocks = (M + 256 * 64 - 1) / (256 * 64);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(rows_gram_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, M, K0,
                     (const bf16 *)X, gram);
  return pn2_check_launch();
}

extern "C" int pn2_bn_relu_apply_bf16(long long M, int N, const void *y, const float *fin, float *out, void *stream) {
  if (M < 0 || N <= 0) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!y || !fin || !out) return PN2_ENULL;
  const size_t total = (size_t)M * N;
  hipLaunchKernelGGL(bn_relu_apply_bf16_kernel, dim3(capped_grid(total)), dim3(256), 0, (hipStream_t)stream, total, N,
                     (const bf16 *)y, fin, out);
  return pn2_check_launch();
}

extern "C" int pn2_bn_relu_bwd_prep_bf16(long long M, int N, const void *y, const float *gout, const float *fin,
                                         void *gpre, double *sums, void *stream) {
  if (M < 0 || N <= 0) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!y || !gout || !fin || !gpre || !sums) return PN2_ENULL;
  if (N % 4 == 0 && N / 4 <= 256 && ((((uintptr_t)y) | ((uintptr_t)gpre)) & 7) == 0 && ((((uintptr_t)gout) | ((uintptr_t)fin)) & 15) == 0) {
    const long long rpbv = prep_vec_rpb_bf(M, N);
    hipLaunchKernelGGL(prep_vec_bf16_kernel, dim3((unsigned)((M + rpbv - 1) / rpbv)), dim3(256), 0, (hipStream_t)stream, M, N, rpbv,
                       (const bf16 *)y, gout, fin, (bf16 *)gpre, sums);
    return pn2_check_launch();
  }
  // rows per block as in pn2_bn_relu_bwd_prep (csrc/mlp_gemm.hip): 16 up to 64k rows, then ~4096 blocks
  int rpb = kPrepRowsBf;
  if (M > 65536) rpb = (int)(((M + 4095) / 4096 + kPrepRowsBf - 1) / kPrepRowsBf * kPrepRowsBf);
  const long long blocks = (M + rpb - 1) / rpb;
  hipLaunchKernelGGL(bn_relu_bwd_prep_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, M, N, rpb,
                     (const bf16 *)y, gout, fin, (bf16 *)gpre, sums);
  return pn2_check_launch();
}

namespace {
// Eight channels per thread (one 16-byte load per row, two rows in flight): the pair-per-thread kernel above keeps 4 bytes
// per lane outstanding and ran at 3.1 TB/s on the scene-graph encoders' last layers.
__global__ __launch_bounds__(256) void bn_relu_rows_max_bf16_v8_kernel(size_t total /* R*C/8 */, int ns, int C,
                                                                      const bf16 *__restrict__ y,
                                                                      const float *__restrict__ fin, float *__restrict__ out,
                                                                      int *__restrict__ arg, float *__restrict__ yraw,
                                                                      const long long *__restrict__ seg) {
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  typedef float f4v __attribute__((ext_vector_type(4)));
  typedef int i4v __attribute__((ext_vector_type(4)));
  const int CV = C / 8;
  if (seg) {                                         // blockIdx.y = scan: its groups and its (4,C) finalize block
    const size_t g0 = (size_t)(seg[blockIdx.y] / ns);
    total = ((size_t)(seg[blockIdx.y + 1] / ns) - g0) * CV;
    y += g0 * ns * C; out += g0 * C; arg += g0 * C; yraw += g0 * C;
    fin += (size_t)blockIdx.y * 4 * C;
  }
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t r = e / CV;
    const int c = (int)(e - r * CV) * 8;
    float sc[8], sh[8], best[8], raw[8];
    int bi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = fin[2 * C + c + i]; sh[i] = fin[3 * C + c + i]; best[i] = -1.f; raw[i] = 0.f; bi[i] = 0; }
    const u4v *p = (const u4v *)(y + r * ns * C + c);
    auto take = [&](const u4v w, int s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float q0 = bf_lo(w[i]), q1 = bf_hi(w[i]);
        const float z0 = fmaxf(fmaf(q0, sc[2 * i], sh[2 * i]), 0.f), z1 = fmaxf(fmaf(q1, sc[2 * i + 1], sh[2 * i + 1]), 0.f);
        if (z0 > best[2 * i]) { best[2 * i] = z0; bi[2 * i] = s; raw[2 * i] = q0; }           // first maximum wins (z >= 0 > -1)
        if (z1 > best[2 * i + 1]) { best[2 * i + 1] = z1; bi[2 * i + 1] = s; raw[2 * i + 1] = q1; }
      }
    };
    int s = 0;
    for (; s + 2 <= ns; s += 2) {
      const u4v w0 = p[(size_t)s * CV], w1 = p[(size_t)(s + 1) * CV];
      take(w0, s);
      take(w1, s + 1);
    }
    if (s < ns) take(p[(size_t)s * CV], s);
    const size_t o = r * C + c;
    *(f4v *)(out + o) = f4v{best[0], best[1], best[2], best[3]};
    *(f4v *)(out + o + 4) = f4v{best[4], best[5], best[6], best[7]};
    *(i4v *)(arg + o) = i4v{bi[0], bi[1], bi[2], bi[3]};
    *(i4v *)(arg + o + 4) = i4v{bi[4], bi[5], bi[6], bi[7]};
    *(f4v *)(yraw + o) = f4v{raw[0], raw[1], raw[2], raw[3]};
    *(f4v *)(yraw + o + 4) = f4v{raw[4], raw[5], raw[6], raw[7]};
  }
}
}  // namespace

namespace {
int rows_max_bf16_impl(long long R, int ns, int C, const void *y, const float *fin, float *out, int *arg, float *yraw,
                       const long long *seg, int nseg, long long seg_max, void *stream) {
  if (R < 0 || ns <= 0 || C <= 0 || C % 2 != 0) return PN2_EINVAL;
  if (R == 0) return PN2_OK;
  if (!y || !fin || !out || !arg || !yraw) return PN2_ENULL;
  const size_t Rg = seg ? (size_t)(seg_max / ns) : (size_t)R;           // groups a grid row covers
  const unsigned gy = seg ? (unsigned)nseg : 1u;
  if (C % 8 == 0 && !(((uintptr_t)y | (uintptr_t)out | (uintptr_t)arg | (uintptr_t)yraw) & 15)) {
    const size_t total8 = (size_t)R * C / 8;
    hipLaunchKernelGGL(bn_relu_rows_max_bf16_v8_kernel, dim3(capped_grid(Rg * C / 8, 16384 / gy + 1), gy), dim3(256), 0,
                       (hipStream_t)stream, total8, ns, C, (const bf16 *)y, fin, out, arg, yraw, seg);
    return pn2_check_launch();
  }
  const size_t total = (size_t)R * C / 2;
  hipLaunchKernelGGL(bn_relu_rows_max_bf16_kernel, dim3(capped_grid(Rg * C / 2, 16384 / gy + 1), gy), dim3(256), 0,
                     (hipStream_t)stream, total, ns, C, (const bf16 *)y, fin, out, arg, yraw, seg);
  return pn2_check_launch();
}
}  // namespace

extern "C" int pn2_bn_relu_rows_max_bf16(long long R, int ns, int C, const void *y, const float *fin, float *out,
                                         int *arg, float *yraw, void *stream) {
  return rows_max_bf16_impl(R, ns, C, y, fin, out, arg, yraw, nullptr, 1, 0, stream);
}

// The same over nseg scans with per-scan constants: seg = ROW offsets of the scans in y (multiples of ns), fin (S,4,C).
extern "C" int pn2_bn_relu_rows_max_bf16_seg(long long R, int ns, int C, const void *y, const float *fin, float *out,
                                             int *arg, float *yraw, const long long *seg, int nseg, long long seg_max,
                                             void *stream) {
  if (!seg) return PN2_ENULL;
  if (nseg < 1 || nseg > 65535 || seg_max < 0) return PN2_EINVAL;
  return rows_max_bf16_impl(R, ns, C, y, fin, out, arg, yraw, seg, nseg, seg_max, stream);
}
