// pool_bwd.hip — backward of the max-pooled LAST layer of an SA stack without its (M, N) output.
//
// Forward (mlp_gemm.hip, EPI_POOL): y_L = a W^T is never stored, a = relu(bn(y_{L-1})) the activation of the layer
// below, only pooled = relu(bn(max_s y_L)) and the arg-max rows are kept (OPS/pointnet2_modules.py:58-70).
// Backward of BatchNorm (batch statistics) + max pool: dL/dy_L = c1 gS + c2 y_L + c3 with per-column constants, gS the
// pooled gradient scattered to the arg-max rows (ONE non-zero per group and column).  With y_L = a W^T both products of
// the layer's backward collapse onto K x K matrices (K = width of the layer below, N = width of this one):
//
//   dL/da = dL/dy_L W      = a (W^T diag(c2) W)  +  1 (c3^T W)  +  gS (diag(c1) W)
//                          = a G + 1 v^T + S            G: K x K,  S: row r gets sum_{n: arg[g][n] = r} gPm[g][n] W'[n][:]
//   dW    = dL/dy_L^T a    = diag(c1) T  +  diag(c2) W (a^T a)  +  c3 (1^T a)
//                                                       T[n][:] = sum_g gPm[g][n] a[row arg[g][n] of group g][:]
//
// so one pass over y_{L-1} computes   g = [a > 0] (a G + v + S)   (stored: the gradient the layer below continues from),
// its BatchNorm-backward column sums, the Gram matrix a^T a, the column sums of a and T — 2 K^2 MACs per row on the
// matrix pipe instead of 4 N K, and no read of y_L: the tensor does not exist.  The sparse parts S and T are N K MACs per
// GROUP and run on the vector unit next to the MFMAs: S as an owner-computes scatter through an LDS tile (LDS float
// atomics retire at 0.3 lane-operations per clock: tools/ubench/lds_atomic.hip), T as a gather from the activation tile.
//
// Tiling: 64-row tiles, 2 K/32 waves.  z-tile (activation) in LDS with pitch K+1: the a G product reads A fragments from
// it (B = G lives in registers: K/2 per lane, loaded once), the Gram product reads both fragments from it.  Results go
// through an LDS staging tile so that mask, statistics and the 16-byte stores run in the row-major layout the raw
// y_{L-1} registers already have.
#include "pn2_common.h"
#include "mlp_common.h"
#include "x3_common.h"
#include <stdlib.h>

namespace {

struct PoolBwdArgs {
  const float *Yp;    // [M][K]  raw pre-BN output of the layer below
  const float *finp;  // [4][K]  mean | rstd | scale | shift of the layer below
  const float *G;     // [K][K]  W^T diag(c2) W
  const float *v;     // [K]     W^T c3
  const float *Wp;    // [N][K]  diag(c1) W
  const int *arg;     // [R][N]  arg-max row per (group, column)
  const float *gPm;   // [R][N]  pooled gradient, zero where pooled <= 0
  float *Gout;        // [M][K]  dL/dz of the layer below (ReLU mask applied)
  double *sums;       // [2][K]  += sum g, sum g * yhat
  float *part;        // [grid][K*K + K + N*K]  per-workgroup partials: a^T a | column sums of a | T
  long long M;
  int N, ns;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 64;

template <int KT, int TNW, bool WLDS>
__global__ __launch_bounds__(128 * KT, 2) void pool_bwd_kernel(const PoolBwdArgs a) {
  constexpr int K = 32 * KT, NW = 2 * KT, THREADS = 64 * NW;
  constexpr int LDZ = K + 1;          // activation tile: conflict-free ds_read_b32 for both fragment patterns
  constexpr int LDT = K + 4;          // staging tile: 16-byte aligned rows
  constexpr int CG = K / 4;           // threads per row in the row-major phases
  constexpr int RP = THREADS / CG;    // rows per pass (16)
  constexpr int NPASS = TM / RP;      // 4
  constexpr int RPW = TM / NW;        // staging rows a wave owns in the scatter phase
  constexpr int KH = KT / 2;          // 64-column halves of a row a lane covers
  constexpr int GB = KT / 2;          // Gram blocks per wave
  constexpr int NBC = NW * TNW / 64;  // 64-column batches of a group's N entries
  constexpr int GMAX = (8 / NBC) < 4 ? (8 / NBC) : 4;   // groups per tile this instantiation holds index registers for
  static_assert(RP == 16 && NPASS == 4 && NBC >= 1, "row-major mapping");
  __shared__ float zt[TM * LDZ];                                    // activation a = relu(bn(y))
  __shared__ __attribute__((aligned(16))) float st[TM * LDT];       // S, then a G + v + S
  __shared__ __attribute__((aligned(16))) float prm[4 * K];         // mean | rstd | scale | shift of the layer below
  __shared__ float wl[WLDS ? NW * TNW * K : 1];                     // W' = diag(c1) W, resident (lane = column: conflict-free)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long M = a.M;
  const int N = a.N, ns = a.ns;
  const long long R = M / ns;
  const long long ntiles = (M + TM - 1) / TM;
  const int ngt = ns >= TM ? 1 : TM / ns;               // groups that overlap a tile (<= GMAX by the host's choice)

  // ---- kernel constants ----
  const int c4 = tid % CG, r0 = tid / CG;
  for (int i = tid; i < 4 * K; i += THREADS) prm[i] = a.finp[i];
  if (WLDS)
    for (int i = tid; i < N * K; i += THREADS) wl[WLDS ? i : 0] = a.Wp[i];
  const int rb = wave / KT, cb = wave % KT;             // a G output block of this wave
  float Greg[K / 2];
#pragma unroll
  for (int s = 0; s < K / 2; ++s) Greg[s] = a.G[(2 * s + (lane >> 5)) * K + cb * 32 + (lane & 31)];
  const float vreg = a.v[cb * 32 + (lane & 31)];
  const int ib = wave >> 1, jb0 = (wave & 1) * GB;      // Gram blocks (ib, jb0 .. jb0 + GB - 1)

  f32x16 accG[GB];
#pragma unroll
  for (int b = 0; b < GB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) accG[b][r] = 0.f;
  float tacc[TNW][KH];
#pragma unroll
  for (int j = 0; j < TNW; ++j)
#pragma unroll
    for (int h = 0; h < KH; ++h) tacc[j][h] = 0.f;
  float zsum[4] = {0.f, 0.f, 0.f, 0.f}, cs1[4] = {0.f, 0.f, 0.f, 0.f}, cs2[4] = {0.f, 0.f, 0.f, 0.f};

  const int yoff = (r0 * K + 4 * c4) * 4;               // lane part of a row-major address (bytes), pass i adds RP*K*4*i
  f32x4 ycur[NPASS], ynxt[NPASS];
  auto load_tile = [&](long long tile, f32x4 (&y)[NPASS]) {
    const long long m0 = tile * TM;
    const rsrc_t rs_ = make_rsrc(a.Yp + (size_t)m0 * K, (M - m0) * K * 4);
#pragma unroll
    for (int i = 0; i < NPASS; ++i)
      y[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, yoff, i * RP * K * 4, 0));
  };
  // index registers of a tile: S scans all N entries of each group (lane = entry), T takes this wave's TNW columns.
  // Loaded one phase ahead (behind the matrix products of the previous tile).
  int sa[GMAX][NBC], ta[GMAX];
  float sg[GMAX][NBC], tg[GMAX];
  auto load_idx = [&](long long tile) {
    const long long g_first = tile * TM / ns;
#pragma unroll
    for (int gi = 0; gi < GMAX; ++gi) {
      const long long g = g_first + gi;
      const bool gok = gi < ngt && g < R;
      const int *argg = a.arg + (size_t)(gok ? g : 0) * N;
      const float *gpg = a.gPm + (size_t)(gok ? g : 0) * N;
#pragma unroll
      for (int q = 0; q < NBC; ++q) {
        const int n = q * 64 + lane;
        const bool ok = gok && n < N;
        sa[gi][q] = ok ? argg[ok ? n : 0] : -(1 << 20);
        sg[gi][q] = ok ? gpg[ok ? n : 0] : 0.f;
      }
      const int n = wave * TNW + lane;
      const bool ok = gok && lane < TNW && n < N;
      ta[gi] = ok ? argg[ok ? n : 0] : -(1 << 20);
      tg[gi] = ok ? gpg[ok ? n : 0] : 0.f;
    }
  };

  long long tile = blockIdx.x;
  if (tile < ntiles) {
    load_tile(tile, ycur);
    load_idx(tile);
  }
  __syncthreads();                                       // parameter table, resident W'
  for (; tile < ntiles; tile += gridDim.x) {
    const long long m0 = tile * TM;
    const int mrem = (int)((M - m0) < (long long)TM ? (M - m0) : (long long)TM);
    // ---- (A) activation tile ----
    {
      const f32x4 sc = *reinterpret_cast<const f32x4 *>(&prm[2 * K + 4 * c4]);
      const f32x4 sh = *reinterpret_cast<const f32x4 *>(&prm[3 * K + 4 * c4]);
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int row = r0 + RP * i;
        const bool valid = row < mrem;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float z = fmaxf(__fmaf_rn(ycur[i][j], sc[j], sh[j]), 0.f);
          z = valid ? z : 0.f;
          zsum[j] += z;
          zt[row * LDZ + 4 * c4 + j] = z;
        }
      }
    }
    __syncthreads();                                     // activation tile visible; the previous tile's staging reads are done
    // the staging rows this wave owns start from zero
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
      for (int h = 0; h < KH; ++h) st[(wave * RPW + rr) * LDT + lane + 64 * h] = 0.f;
    // next tile's rows in flight behind everything below
    const long long nt = (tile + gridDim.x) < ntiles ? tile + gridDim.x : tile;
    load_tile(nt, ynxt);

    // ---- (B) sparse parts ----
    const long long g_first = m0 / ns;
    // S: the RPW staging rows this wave owns lie in ONE group (RPW divides ns): of that group's N entries (lane = entry)
    // the wave keeps those whose arg-max row it owns and adds coef * W'[n][:] to the row (lanes = columns; LDS
    // read-modify-write, exclusive rows — LDS float atomics are 20x slower, tools/ubench/lds_atomic.hip)
    {
      const int gw = ns >= TM ? 0 : (wave * RPW) / ns;
#pragma unroll
      for (int gi = 0; gi < GMAX; ++gi) {
        if (gi == gw) {
          const int base = (int)((g_first + gi) * ns - m0);
#pragma unroll
          for (int q = 0; q < NBC; ++q) {
            const int rt = base + sa[gi][q];
            const float cf = sg[gi][q];
            const bool mine = (unsigned)(rt - wave * RPW) < (unsigned)RPW && cf != 0.f;
            u64 mask = __ballot(mine);
#ifdef PB_NO_S
            mask = 0;
#endif
            while (mask) {
              int er[4], en[4];
              float ec[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                if (mask) {
                  const int e = __builtin_ctzll(mask);
                  mask &= mask - 1;
                  er[u] = __builtin_amdgcn_readlane(rt, e);
                  ec[u] = pn2_readlane_f32(cf, e);
                  en[u] = q * 64 + e;
                } else {
                  er[u] = wave * RPW;
                  ec[u] = 0.f;
                  en[u] = 0;
                }
              }
              float w[4][KH];
#pragma unroll
              for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int h = 0; h < KH; ++h)
                  w[u][h] = WLDS ? wl[WLDS ? en[u] * K + lane + 64 * h : 0] : a.Wp[(size_t)en[u] * K + lane + 64 * h];
#pragma unroll
              for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int h = 0; h < KH; ++h) {
                  float *p = &st[er[u] * LDT + lane + 64 * h];
                  *p = __fmaf_rn(ec[u], w[u][h], *p);
                }
            }
          }
        }
      }
    }
    // T: this wave's columns n = wave * TNW + j gather their arg-max activation row of every group in the tile
#ifndef PB_NO_T
#pragma unroll
    for (int gi = 0; gi < GMAX; ++gi) {
      if (gi < ngt) {
        const int base = (int)((g_first + gi) * ns - m0);  // tile row of the group's first row (-64: second half of ns = 128)
        int rt = base + ta[gi];
        const bool ok = (unsigned)rt < (unsigned)TM;
        const float cf = ok ? tg[gi] : 0.f;
        rt = ok ? rt : 0;
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
          const int r = __builtin_amdgcn_readlane(rt, j);
          const float c = pn2_readlane_f32(cf, j);
#pragma unroll
          for (int h = 0; h < KH; ++h) tacc[j][h] = __fmaf_rn(c, zt[r * LDZ + lane + 64 * h], tacc[j][h]);
          if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#endif
    load_idx(nt);                                        // (the index registers are free now)

    // ---- (C) matrix products ----
    f32x16 accZ;
#pragma unroll
    for (int r = 0; r < 16; ++r) accZ[r] = 0.f;
    {
      const float *za = &zt[(rb * 32 + (lane & 31)) * LDZ + (lane >> 5)];
#pragma unroll
      for (int s = 0; s < K / 2; ++s) {
        accZ = __builtin_amdgcn_mfma_f32_32x32x2f32(za[2 * s], Greg[s], accZ, 0, 0, 0);
        if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);     // bounds how many fragment reads are hoisted (registers)
      }
      const float *ga = &zt[(lane >> 5) * LDZ + ib * 32 + (lane & 31)];
      const float *gb = &zt[(lane >> 5) * LDZ + jb0 * 32 + (lane & 31)];
#pragma unroll
      for (int s = 0; s < TM / 2; ++s) {
        const float av = ga[2 * s * LDZ];
#pragma unroll
        for (int b = 0; b < GB; ++b)
          accG[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gb[2 * s * LDZ + b * 32], accG[b], 0, 0, 0);
        if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                                     // S complete, matrix products done with the activation tile
    // ---- (D) a G + v + S into the staging tile ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      float *p = &st[row * LDT + cb * 32 + (lane & 31)];
      *p = (accZ[r] + vreg) + *p;
    }
    __syncthreads();
    // ---- (E) mask, statistics, store (row-major) ----
    {
      const rsrc_t rso = make_rsrc(a.Gout + (size_t)m0 * K, (M - m0) * K * 4);
      const f32x4 mu = *reinterpret_cast<const f32x4 *>(&prm[4 * c4]);
      const f32x4 rs = *reinterpret_cast<const f32x4 *>(&prm[K + 4 * c4]);
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int row = r0 + RP * i;
        const f32x4 q = *reinterpret_cast<const f32x4 *>(&st[row * LDT + 4 * c4]);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // the activation tile is the forward's own ReLU decision (rows past M hold zeros)
          const float g = zt[row * LDZ + 4 * c4 + j] > 0.f ? q[j] : 0.f;
          o[j] = g;
          cs1[j] += g;
          cs2[j] = __fmaf_rn(g, (ycur[i][j] - mu[j]) * rs[j], cs2[j]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), rso,
                                               yoff, i * RP * K * 4, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NPASS; ++i) ycur[i] = ynxt[i];
  }

  // ---- flush ----
  float *prec = a.part + (size_t)blockIdx.x * (K * K + K + (size_t)N * K);
  __syncthreads();
  float *rbuf = zt;                                      // [3][THREADS][4]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rbuf[(0 * THREADS + tid) * 4 + j] = zsum[j];
    rbuf[(1 * THREADS + tid) * 4 + j] = cs1[j];
    rbuf[(2 * THREADS + tid) * 4 + j] = cs2[j];
  }
  __syncthreads();
  if (tid < K) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int q = 0; q < RP; ++q) {
      const int src = ((q * CG + (tid >> 2)) * 4) + (tid & 3);
      t0 += rbuf[0 * THREADS * 4 + src];
      t1 += rbuf[1 * THREADS * 4 + src];
      t2 += rbuf[2 * THREADS * 4 + src];
    }
    prec[K * K + tid] = t0;
    atomicAdd(a.sums + tid, (double)t1);
    atomicAdd(a.sums + K + tid, (double)t2);
  }
  float *pz = prec;
#pragma unroll
  for (int b = 0; b < GB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      pz[i * K + (jb0 + b) * 32 + (lane & 31)] = accG[b][r];
    }
  float *pt = prec + K * K + K;
#pragma unroll
  for (int j = 0; j < TNW; ++j) {
    const int n = wave * TNW + j;
    if (n < N) {
#pragma unroll
      for (int h = 0; h < KH; ++h) pt[(size_t)n * K + lane + 64 * h] = tacc[j][h];
    }
  }
}

// K = 64: eight waves per 64-row tile, one workgroup per CU.  Matrix roles: waves 0-3 own the four 32 x 32 blocks of
// a G (+ v + S), waves 4-7 the four blocks of the Gram matrix a^T a.  The sparse parts are laid out so that the inner loops
// have no scalar bookkeeping at all (per-entry readlane / find-first-bit chains cost 180 cycles per entry):
//   T (waves 4-7): lanes = columns n, registers = a slab of columns k; every lane walks along ITS arg-max row of the
//     activation tile with immediate offsets;
//   S (waves 0-3): lanes = the 64 rows of the tile, registers = a 16-column slab of k.  While the tile is staged, every
//     thread files ONE (group, column) entry under its arg-max row — rank from an LDS integer atomic, up to CAP entries per
//     row, the rest in an overflow list — and in the scatter phase lane r walks the list of row r, reading W'[n][slab]
//     (pitch K + 1: the rows n differ per lane).  The order inside a row's list is the arrival order of the atomics: the
//     fp32 sum of a row is not bit-reproducible from run to run (like the atomic weight gradients of mlp_wgrad).
#ifdef PB_PROF
#define PB_T(i) { const long long t_ = __builtin_readcyclecounter(); prof[i] += t_ - tprev; tprev = t_; }
#else
#define PB_T(i)
#endif
// X3: the two matrix products (a G and the Gram blocks) on the split-bf16 product of x3_common.h: the same LDS operand reads,
// eight per 16-deep chunk, split in registers; 6 bf16 matrix instructions per chunk instead of 8 fp32 ones at half their length,
// and they issue beside the vector work (row lists, S / T walks) of the wave that shares the SIMD — the fp32 ones do not.
template <int NH, bool X3 = false>
__global__ __launch_bounds__(512, X3 ? 1 : 2) void pool_bwd64_kernel(const PoolBwdArgs a) {
#ifdef PB_PROF
  long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = __builtin_readcyclecounter();
#endif
  constexpr int K = 64, THREADS = 512;
  // (LDW = K + 4: rows of W' 16-byte aligned for the S walk's ds_read_b128 — four reads per list slot instead of sixteen and
  // three-way instead of four-to-five-way bank serialisation for random arg-max columns: 1.180 -> 1.150 ms at the SA1 shape)
  constexpr int LDZ = K + 1, LDT = K + 4, LDW = K + 4;
  constexpr int CG = K / 4;           // 16 threads per row
  constexpr int RP = THREADS / CG;    // 32 rows per pass
  constexpr int NPASS = TM / RP;      // 2
  constexpr int NPAD = 64 * NH;       // entries per group, padded (N <= NPAD)
  constexpr int GMAX = THREADS / NPAD < 4 ? THREADS / NPAD : 4;   // groups per tile: one entry per thread
  constexpr int KQ = 16 * NH;         // T: waves 4-7 = (64-column half of N) x (KQ-column slab of K)
  constexpr int KS = K / 4;           // S: 16-column slab of K per wave (waves 0-3; waves 4-7 run T meanwhile)
  constexpr int CAP = 8;              // listed entries per row
  constexpr int OVC = GMAX * NPAD;    // overflow capacity: every entry of the tile
  __shared__ float zt[TM * LDZ];
  __shared__ __attribute__((aligned(16))) float st[TM * LDT];
  __shared__ __attribute__((aligned(16))) float prm[4 * K];
  __shared__ __attribute__((aligned(16))) float wl[NPAD * LDW];
  __shared__ int cnt[TM + 1];         // entries filed per row; [TM] = overflow count
  __shared__ int lst_n[TM * CAP];
  __shared__ float lst_c[TM * CAP];
  __shared__ int ovf_rn[OVC];         // row << 16 | n
  __shared__ float ovf_c[OVC];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool zrole = wave < 4;
  const long long M = a.M;
  const int N = a.N, ns = a.ns;
  const long long R = M / ns;
  const long long ntiles = (M + TM - 1) / TM;
  const int ngt = ns >= TM ? 1 : TM / ns;

  const int c4 = tid % CG, r0 = tid / CG;
  for (int i = tid; i < 4 * K; i += THREADS) prm[i] = a.finp[i];
  for (int i = tid; i < N * K; i += THREADS) wl[(i / K) * LDW + (i % K)] = a.Wp[i];
  if (tid <= TM) cnt[tid] = 0;
  const int blk = wave & 3;
  const int rb = blk >> 1, cb = blk & 1;                 // output block of a G (waves 0-3) / Gram block (waves 4-7)
  // Role registers: waves 0-3 keep the B fragments of their a G block in them, waves 4-7 two of their three running Gram
  // accumulators.  a^T a is symmetric: its blocks (0,0), (1,1), (0,1) are computed — every Gram wave a quarter of the tile's
  // rows (16 wq .. 16 wq + 15) of each of the three: 24 matrix instructions per tile instead of 32, from two LDS operands per
  // step instead of two per instruction — and summed / mirrored at the flush.
  // (X3: ONE register array for both roles — the compiler cannot know that a wave keeps its role, two sets would both stay
  // live: waves 0-3 hold the twelve pieces of their four B fragments of G in it, chunk c = rows k = 16 c + 2 i + h, the fp32
  // loop's order; waves 4-7 the bits of the two running Gram accumulators)
  f32x16 rg0, rg1;
  f32x4 rr[X3 ? 12 : 1];
  auto rr_get = [&](int base) {
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * q + j] = rr[X3 ? base + q : 0][j];
    return v;
  };
  auto rr_put = [&](int base, const f32x16 &v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) rr[X3 ? base + q : 0] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  };
  if (X3) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float gv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) gv[i] = zrole ? a.G[(16 * c + 2 * i + (lane >> 5)) * K + cb * 32 + (lane & 31)] : 0.f;
      x3_frag f;
      x3_split8(gv, f);                                  // (waves 4-7: zeros split into zeros)
#pragma unroll
      for (int q = 0; q < 3; ++q) rr[X3 ? 3 * c + q : 0] = __builtin_bit_cast(f32x4, f.p[q]);
    }
  } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      rg0[s] = zrole ? a.G[(2 * s + (lane >> 5)) * K + cb * 32 + (lane & 31)] : 0.f;
      rg1[s] = zrole ? a.G[(2 * (s + 16) + (lane >> 5)) * K + cb * 32 + (lane & 31)] : 0.f;
    }
  }
  const int wq = wave & 3;
  const float vreg = a.v[cb * 32 + (lane & 31)];
  const int tn0 = ((wave & 3) % NH) * 64, tk0 = ((wave & 3) / NH) * KQ;     // T (waves 4-7): first column n, first column k
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 tacc[KQ / 2];                                       // (pairs of columns: packed FMAs, see the S walk)
#pragma unroll
  for (int kk = 0; kk < KQ / 2; ++kk) tacc[kk] = f2{0.f, 0.f};

  f32x16 acc;                                            // role 1: per-tile a G block; role 2: running Gram block
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float zsum[4] = {0.f, 0.f, 0.f, 0.f}, cs1[4] = {0.f, 0.f, 0.f, 0.f}, cs2[4] = {0.f, 0.f, 0.f, 0.f};

  const int yoff = (r0 * K + 4 * c4) * 4;
  f32x4 ycur[NPASS], ynxt[NPASS];
  auto load_tile = [&](long long tile, f32x4 (&y)[NPASS]) {
    const long long m0 = tile * TM;
    const rsrc_t rs_ = make_rsrc(a.Yp + (size_t)m0 * K, (M - m0) * K * 4);
#pragma unroll
    for (int i = 0; i < NPASS; ++i)
      y[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, yoff, i * RP * K * 4, 0));
  };
  // index registers, loaded one phase ahead: the thread's own entry (group tid / NPAD, column tid % NPAD) for the row
  // lists, and for waves 4-7 the column tn0 + lane of every group for T
  const int egi = tid / NPAD, en = tid % NPAD;
  int ea, ta[GMAX];
  float ec, tg[GMAX];
  auto load_entry = [&](long long tile) {
    const long long g = tile * TM / ns + egi;
    const bool ok = egi < ngt && g < R && en < N;
    ea = ok ? a.arg[(size_t)(ok ? g : 0) * N + (ok ? en : 0)] : -(1 << 20);
    ec = ok ? a.gPm[(size_t)(ok ? g : 0) * N + (ok ? en : 0)] : 0.f;
  };
  auto load_tcols = [&](long long tile) {
    const long long g_first = tile * TM / ns;
#pragma unroll
    for (int gi = 0; gi < GMAX; ++gi) {
      const long long g = g_first + gi;
      const int n = tn0 + lane;
      const bool ok = !zrole && gi < ngt && g < R && n < N;
      ta[gi] = ok ? a.arg[(size_t)(ok ? g : 0) * N + (ok ? n : 0)] : -(1 << 20);
      tg[gi] = ok ? a.gPm[(size_t)(ok ? g : 0) * N + (ok ? n : 0)] : 0.f;
    }
  };
  auto load_idx = [&](long long tile) { load_entry(tile); load_tcols(tile); };     // (before the first tile)

  long long tile = blockIdx.x;
  if (tile < ntiles) {
    load_tile(tile, ycur);
    load_idx(tile);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const long long m0 = tile * TM;
    const int mrem = (int)((M - m0) < (long long)TM ? (M - m0) : (long long)TM);
    const long long g_first = m0 / ns;
    PB_T(9)
    // ---- (A) activation tile; this thread's entry filed under its arg-max row ----
    {
      const f32x4 sc = *reinterpret_cast<const f32x4 *>(&prm[2 * K + 4 * c4]);
      const f32x4 sh = *reinterpret_cast<const f32x4 *>(&prm[3 * K + 4 * c4]);
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int row = r0 + RP * i;
        const bool valid = row < mrem;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float z = fmaxf(__fmaf_rn(ycur[i][j], sc[j], sh[j]), 0.f);
          z = valid ? z : 0.f;
          zsum[j] += z;
          zt[row * LDZ + 4 * c4 + j] = z;
        }
      }
      const int row = (int)((g_first + egi) * ns - m0) + ea;     // tile row of the entry's arg-max (ns = 128: may lie outside)
      if ((unsigned)row < (unsigned)TM && ec != 0.f) {
        const int rank = atomicAdd(&cnt[row], 1);
        if (rank < CAP) {
          lst_n[row * CAP + rank] = en;
          lst_c[row * CAP + rank] = ec;
        } else {
          const int o = atomicAdd(&cnt[TM], 1);
          ovf_rn[o] = (row << 16) | en;
          ovf_c[o] = ec;
        }
      }
    }
    PB_T(0)
    __syncthreads();
    PB_T(1)
    const long long nt = (tile + gridDim.x) < ntiles ? tile + gridDim.x : tile;
    load_tile(nt, ynxt);

    // ---- (B) sparse parts ----
    if (zrole) {
      // S: lane = tile row, registers = this wave's KS columns.  The first CAP list slots of the row are read up front
      // (empty slots: coefficient 0), so the W' reads of all slots are independent and in flight together.
      // (packed fp32 FMAs — v_pk_fma_f32, the IEEE operation on two columns per instruction: fp32 matrix and vector
      // instructions share the SIMD's datapath on this part, so every vector instruction saved is matrix time)
      f2 sreg[KS / 2];
#pragma unroll
      for (int kk = 0; kk < KS / 2; ++kk) sreg[kk] = f2{0.f, 0.f};
      const int filed = cnt[lane];
      const int mine = filed < CAP ? filed : CAP;
      // two list slots per step: their W' reads are independent and in flight together
      for (int c = 0; __ballot(c < mine) != 0ull; c += 2) {
        int n2[2];
        f2 c2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bool act = c + u < mine;
          n2[u] = act ? lst_n[lane * CAP + c + u] : 0;
          const float cf = act ? lst_c[lane * CAP + c + u] : 0.f;
          c2[u] = f2{cf, cf};
        }
        f2 w[2][KS / 2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int kk = 0; kk < KS; kk += 4) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(&wl[n2[u] * LDW + wave * KS + kk]);
            w[u][kk / 2] = f2{q[0], q[1]};
            w[u][kk / 2 + 1] = f2{q[2], q[3]};
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int kk = 0; kk < KS / 2; ++kk) sreg[kk] = __builtin_elementwise_fma(c2[u], w[u][kk], sreg[kk]);
      }
      const int nov = __builtin_amdgcn_readfirstlane(cnt[TM]);
      for (int o = 0; o < nov; ++o) {                      // rows with more than CAP entries: one lane at a time
        const int rn = __builtin_amdgcn_readfirstlane(ovf_rn[o]);
        const float cf = (rn >> 16) == lane ? ovf_c[o] : 0.f;
        const float *wr = &wl[(rn & 0xffff) * LDW + wave * KS];
#pragma unroll
        for (int kk = 0; kk < KS / 2; ++kk) sreg[kk] = __builtin_elementwise_fma(f2{cf, cf}, f2{wr[2 * kk], wr[2 * kk + 1]}, sreg[kk]);
      }
#pragma unroll
      for (int kk = 0; kk < KS; kk += 4) {
        f32x4 o4 = {sreg[kk / 2][0], sreg[kk / 2][1], sreg[kk / 2 + 1][0], sreg[kk / 2 + 1][1]};
        *reinterpret_cast<f32x4 *>(&st[lane * LDT + wave * KS + kk]) = o4;
      }
    }
    // T: lanes = columns n of this wave's half, registers = its KQ columns k; every lane walks along ITS arg-max row of
    // the activation tile (immediate offsets, independent loads)
    auto t_walk = [&]() {
#pragma unroll
      for (int gi = 0; gi < GMAX; ++gi) {
        if (!zrole && gi < ngt) {
          const int base = (int)((g_first + gi) * ns - m0);
          int rt = base + ta[gi];
          const bool ok = (unsigned)rt < (unsigned)TM;
          const float cf = ok ? tg[gi] : 0.f;
          rt = ok ? rt : 0;
          const float *zr = &zt[rt * LDZ + tk0];
#pragma unroll
          for (int kk = 0; kk < KQ / 2; ++kk) tacc[kk] = __builtin_elementwise_fma(f2{cf, cf}, f2{zr[2 * kk], zr[2 * kk + 1]}, tacc[kk]);
        }
      }
    };
    PB_T(2)
    load_entry(nt);

    // ---- (C) matrix products: one 32 x 32 block per wave ----
    if (zrole) {
      // A fragments from LDS, the operands of the next eight steps read while eight MFMAs issue (read-wait-issue per pair
      // of steps made this phase wait for an LDS round trip sixteen times per tile: 3.3k cycles for 2.0k of matrix pipe)
      const float *za = &zt[(rb * 32 + (lane & 31)) * LDZ + (lane >> 5)];
      float zp[2][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) zp[0][u] = za[2 * u];
#pragma unroll
      for (int g8 = 0; g8 < K / 16; ++g8) {
        if (g8 + 1 < K / 16) {
#pragma unroll
          for (int u = 0; u < 8; ++u) zp[(g8 + 1) & 1][u] = za[2 * (8 * (g8 + 1) + u)];
        }
        if (X3) {
          x3_frag fa, fb;
          x3_split8(zp[g8 & 1], fa);
#pragma unroll
          for (int q = 0; q < 3; ++q) fb.p[q] = __builtin_bit_cast(x3_u32x4, rr[X3 ? 3 * g8 + q : 0]);
          x3_mma(fa, fb, acc);
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(zp[g8 & 1][u], g8 < 2 ? rg0[8 * g8 + u] : rg1[8 * (g8 - 2) + u], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // Gram: rows 16 wq .. 16 wq + 15 of the blocks (0,0), (1,1), (0,1) — both column halves of the eight row pairs are read
      // up front (sixteen LDS operands for twenty-four matrix instructions)
      const float *g0 = &zt[((lane >> 5) + 16 * wq) * LDZ + (lane & 31)];
      float p0[8], p1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { p0[u] = g0[2 * u * LDZ]; p1[u] = g0[2 * u * LDZ + 32]; }
      if (X3) {
        x3_frag f0, f1;
        x3_split8(p0, f0);
        x3_split8(p1, f1);
        f32x16 r0 = rr_get(0), r1 = rr_get(4);
        x3_mma(f0, f0, acc);
        x3_mma(f1, f1, r0);
        x3_mma(f0, f1, r1);
        rr_put(0, r0);
        rr_put(4, r1);
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p0[u], p0[u], acc, 0, 0, 0);
          rg0 = __builtin_amdgcn_mfma_f32_32x32x2f32(p1[u], p1[u], rg0, 0, 0, 0);
          rg1 = __builtin_amdgcn_mfma_f32_32x32x2f32(p0[u], p1[u], rg1, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // T AFTER the Gram block (it only needs the activation tile): the matrix pipe starts while waves 0-3 still walk their
      // row lists, and this walk runs in the shadow of their a G products (1.154 -> 1.146 ms at the SA1 shape)
      t_walk();
      load_tcols(nt);
    }
    PB_T(3)
    __syncthreads();
    PB_T(4)
    // ---- (D) a G + v + S into the staging tile; the row lists are free again ----
    if (zrole) {
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = st[(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + cb * 32 + (lane & 31)];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + cb * 32 + (lane & 31)] = (acc[r] + vreg) + t[r];
        acc[r] = 0.f;
      }
    } else if (tid - 256 <= TM) {
      cnt[tid - 256] = 0;
    }
    PB_T(5)
    __syncthreads();
    PB_T(6)
    // ---- (E) mask, statistics, store ----
    {
      const rsrc_t rso = make_rsrc(a.Gout + (size_t)m0 * K, (M - m0) * K * 4);
      const f32x4 mu = *reinterpret_cast<const f32x4 *>(&prm[4 * c4]);
      const f32x4 rs = *reinterpret_cast<const f32x4 *>(&prm[K + 4 * c4]);
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int row = r0 + RP * i;
        const f32x4 q = *reinterpret_cast<const f32x4 *>(&st[row * LDT + 4 * c4]);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g = zt[row * LDZ + 4 * c4 + j] > 0.f ? q[j] : 0.f;
          o[j] = g;
          cs1[j] += g;
          cs2[j] = __fmaf_rn(g, (ycur[i][j] - mu[j]) * rs[j], cs2[j]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), rso,
                                               yoff, i * RP * K * 4, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NPASS; ++i) ycur[i] = ynxt[i];
    PB_T(7)
  }
#ifdef PB_PROF
  if (lane == 0 && blockIdx.x < 64) {
    long long *dst = (long long *)a.Gout + (blockIdx.x * 8 + wave) * 10;      // (experiment build only: clobbers the first rows of the output)
    for (int i = 0; i < 10; ++i) dst[i] = prof[i];
  }
#endif

  // ---- flush ----
  float *prec = a.part + (size_t)blockIdx.x * (K * K + K + (size_t)N * K);
  __syncthreads();
  float *rbuf = st;                                      // [THREADS][4], three rounds (st holds 64 x 68 floats)
  float t3[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int round = 0; round < 3; ++round) {
#pragma unroll
    for (int j = 0; j < 4; ++j) rbuf[tid * 4 + j] = round == 0 ? zsum[j] : (round == 1 ? cs1[j] : cs2[j]);
    __syncthreads();
    if (tid < K) {
      float t = 0.f;
      for (int q = 0; q < RP; ++q) t += rbuf[((q * CG + (tid >> 2)) * 4) + (tid & 3)];
      t3[round] = t;
    }
    __syncthreads();
  }
  if (tid < K) {
    prec[K * K + tid] = t3[0];
    atomicAdd(a.sums + tid, (double)t3[1]);
    atomicAdd(a.sums + K + tid, (double)t3[2]);
  }
  // Gram: the four row-quarter partials of a block meet in LDS (one block per round), wave 4 + b writes block b and, for
  // (0,1), its mirror image
  float *gq = st;                                        // [4 waves][16][64] (16 KB of the staging tile)
  const f32x16 g0f = X3 ? rr_get(0) : rg0, g1f = X3 ? rr_get(4) : rg1;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    if (!zrole) {
#pragma unroll
      for (int r = 0; r < 16; ++r) gq[(wq * 16 + r) * 64 + lane] = b == 0 ? acc[r] : (b == 1 ? g0f[r] : g1f[r]);
    }
    __syncthreads();
    if (wave == 4 + b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 31;
        const float v = (gq[r * 64 + lane] + gq[(16 + r) * 64 + lane]) + (gq[(32 + r) * 64 + lane] + gq[(48 + r) * 64 + lane]);
        if (b == 0) prec[i * K + j] = v;
        else if (b == 1) prec[(32 + i) * K + 32 + j] = v;
        else { prec[i * K + 32 + j] = v; prec[(32 + j) * K + i] = v; }
      }
    }
    __syncthreads();
  }
  float *pt = prec + K * K + K;
  if (!zrole && tn0 + lane < N) {
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) pt[(size_t)(tn0 + lane) * K + tk0 + kk] = tacc[kk / 2][kk & 1];
  }
}

// K = 128: the same organisation (row lists, lanes = rows for S, lanes = columns for T), eight waves, one workgroup per
// CU.  Every wave owns one of the eight 32 x 32 blocks of a G and two of the sixteen Gram blocks.  G (64 KB) lives in
// LDS (B fragments read per step: the registers hold the 64 T accumulators instead); W' (N x 128: up to 128 KB) does not
// fit next to it and is gathered from L2 — lane r reads the 64-byte slab W'[n][16 w .. 16 w + 15] of ITS list entry.
// A thread files two (group, column) entries per tile (four groups x 256 columns at ns = 16).
__global__ __launch_bounds__(512, 2) void pool_bwd128_kernel(const PoolBwdArgs a) {
#ifdef PB_PROF
  long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = __builtin_readcyclecounter();
#endif
  constexpr int K = 128, THREADS = 512;
  constexpr int LDZ = K + 1, LDT = K + 4, LDG = K + 1;
  constexpr int CG = K / 4;           // 32 threads per row
  constexpr int RP = THREADS / CG;    // 16 rows per pass
  constexpr int NPASS = TM / RP;      // 4
  constexpr int NPAD = 256;           // entries per group, padded (N <= 256)
  constexpr int GMAX = 4;             // groups per tile (ns >= 16)
  constexpr int EPT = GMAX * NPAD / THREADS;   // entries a thread files per tile (2)
  constexpr int KQ = 64;              // T: wave = (64-column quarter of N) x (64-column half of K)
  constexpr int CAP = 16;             // N / ns = 8 entries per row on average at the headline shape
  constexpr int OVC = GMAX * NPAD;
  __shared__ float zt[TM * LDZ];
  __shared__ __attribute__((aligned(16))) float st[TM * LDT];
  __shared__ float gl[K * LDG];
  __shared__ __attribute__((aligned(16))) float prm[4 * K];
  __shared__ int cnt[TM + 1];
  __shared__ int lst_n[TM * CAP];
  __shared__ float lst_c[TM * CAP];
  __shared__ int ovf_rn[OVC];
  __shared__ float ovf_c[OVC];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long M = a.M;
  const int N = a.N, ns = a.ns;
  const long long R = M / ns;
  const long long ntiles = (M + TM - 1) / TM;
  const int ngt = ns >= TM ? 1 : TM / ns;

  const int c4 = tid % CG, r0 = tid / CG;
  for (int i = tid; i < 4 * K; i += THREADS) prm[i] = a.finp[i];
  for (int i = tid; i < K * K; i += THREADS) gl[(i / K) * LDG + (i % K)] = a.G[i];
  if (tid <= TM) cnt[tid] = 0;
  const int rb = wave >> 2, cb = wave & 3;               // a G block
  // Gram blocks: a^T a is symmetric — ten of its sixteen 32 x 32 blocks are computed (the flush mirrors the off-diagonal
  // ones).  Every wave owns ONE block (waves 0-3 the diagonal, 4-7: (0,1) (2,3) (0,2) (1,3)) and a quarter of the tile's
  // rows of one of the two remaining blocks ((0,3): waves 0-3, (1,2): waves 4-7; summed through LDS at the flush):
  // 32 + 8 matrix instructions per tile and wave instead of 64.
  const int oi = wave < 4 ? wave : (wave == 4 || wave == 6 ? 0 : (wave == 5 ? 2 : 1));
  const int oj = wave < 4 ? wave : (wave == 4 ? 1 : (wave == 6 ? 2 : 3));
  const int si = wave < 4 ? 0 : 1, sj = wave < 4 ? 3 : 2, sq = wave & 3;
  const float vreg = a.v[cb * 32 + (lane & 31)];
  const int tn0 = (wave & 3) * 64, tk0 = (wave >> 2) * KQ;

  f32x16 accZ, accG[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { accZ[r] = 0.f; accG[0][r] = 0.f; accG[1][r] = 0.f; }
  float tacc[KQ];
#pragma unroll
  for (int kk = 0; kk < KQ; ++kk) tacc[kk] = 0.f;
  float zsum[4] = {0.f, 0.f, 0.f, 0.f}, cs1[4] = {0.f, 0.f, 0.f, 0.f}, cs2[4] = {0.f, 0.f, 0.f, 0.f};

  const int yoff = (r0 * K + 4 * c4) * 4;
  f32x4 ycur[NPASS], ynxt[NPASS];
  auto load_tile = [&](long long tile, f32x4 (&y)[NPASS]) {
    const long long m0 = tile * TM;
    const rsrc_t rs_ = make_rsrc(a.Yp + (size_t)m0 * K, (M - m0) * K * 4);
#pragma unroll
    for (int i = 0; i < NPASS; ++i)
      y[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, yoff, i * RP * K * 4, 0));
  };
  int ea[EPT], ta[GMAX];
  float ec[EPT], tg[GMAX];
  auto load_idx = [&](long long tile) {
    const long long g_first = tile * TM / ns;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      const int e = tid + THREADS * u;
      const int egi = e / NPAD, en = e % NPAD;
      const long long g = g_first + egi;
      const bool ok = egi < ngt && g < R && en < N;
      ea[u] = ok ? a.arg[(size_t)(ok ? g : 0) * N + (ok ? en : 0)] : -(1 << 20);
      ec[u] = ok ? a.gPm[(size_t)(ok ? g : 0) * N + (ok ? en : 0)] : 0.f;
    }
#pragma unroll
    for (int gi = 0; gi < GMAX; ++gi) {
      const long long g = g_first + gi;
      const int n = tn0 + lane;
      const bool ok = gi < ngt && g < R && n < N;
      ta[gi] = ok ? a.arg[(size_t)(ok ? g : 0) * N + (ok ? n : 0)] : -(1 << 20);
      tg[gi] = ok ? a.gPm[(size_t)(ok ? g : 0) * N + (ok ? n : 0)] : 0.f;
    }
  };

  long long tile = blockIdx.x;
  if (tile < ntiles) {
    load_tile(tile, ycur);
    load_idx(tile);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const long long m0 = tile * TM;
    const int mrem = (int)((M - m0) < (long long)TM ? (M - m0) : (long long)TM);
    const long long g_first = m0 / ns;
    // ---- (A) activation tile; this thread's entries filed under their arg-max rows ----
    {
      const f32x4 sc = *reinterpret_cast<const f32x4 *>(&prm[2 * K + 4 * c4]);
      const f32x4 sh = *reinterpret_cast<const f32x4 *>(&prm[3 * K + 4 * c4]);
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int row = r0 + RP * i;
        const bool valid = row < mrem;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float z = fmaxf(__fmaf_rn(ycur[i][j], sc[j], sh[j]), 0.f);
          z = valid ? z : 0.f;
          zsum[j] += z;
          zt[row * LDZ + 4 * c4 + j] = z;
        }
      }
#pragma unroll
      for (int u = 0; u < EPT; ++u) {
        const int e = tid + THREADS * u;
        const int row = (int)((g_first + e / NPAD) * ns - m0) + ea[u];
        if ((unsigned)row < (unsigned)TM && ec[u] != 0.f) {
          const int rank = atomicAdd(&cnt[row], 1);
          if (rank < CAP) {
            lst_n[row * CAP + rank] = e % NPAD;
            lst_c[row * CAP + rank] = ec[u];
          } else {
            const int o = atomicAdd(&cnt[TM], 1);
            ovf_rn[o] = (row << 16) | (e % NPAD);
            ovf_c[o] = ec[u];
          }
        }
      }
    }
    PB_T(0)
    __syncthreads();
    PB_T(1)
    const long long nt = (tile + gridDim.x) < ntiles ? tile + gridDim.x : tile;

    // ---- (B) sparse parts ----
    {
      // S: wave w owns the tile rows 8w .. 8w + 7 and ALL K columns (lane l: columns 2l, 2l + 1).  A list entry is then
      // one coalesced 512-byte row of W' with a wave-uniform coefficient — as lanes = rows (round-3 first version) every
      // wave gathered its own 64-byte slice of the same 64 rows per step: 16k cache-line requests per tile through the one
      // texture-address unit of the CU, which was the kernel's time.  Eight list slots are fetched back to back (an
      // empty slot reads row 0 with coefficient 0: no branches between the loads).
      typedef float f2 __attribute__((ext_vector_type(2)));
      const int rl = lane >> 3, c0 = lane & 7;               // lane -> (row of the wave's eight, list slot)
      const int myrow = wave * 8 + rl;
      const int filed = cnt[myrow];
      const int mine = filed < CAP ? filed : CAP;
      const bool va = c0 < mine, vb = c0 + 8 < mine;
      const int n_a = va ? lst_n[myrow * CAP + c0] : 0, n_b = vb ? lst_n[myrow * CAP + c0 + 8] : 0;
      const float c_a = va ? lst_c[myrow * CAP + c0] : 0.f, c_b = vb ? lst_c[myrow * CAP + c0 + 8] : 0.f;
      const float *wl = a.Wp + 2 * lane;
#pragma unroll
      for (int r = 0; r < 8; r += 2) {
        // two rows per batch; only their filed slots are fetched (the counts are wave-uniform: a scalar branch per slot, no
        // wait in between).  Fetching all sixteen slots of every row — an empty one read row 0 — kept the loads branch-free
        // but moved 512 KB per tile through the CU's one 64-byte-per-clock vector memory path for 256 KB of entries: that
        // path, not the L2 latency, was the phase's time.
        int cr[2];
        f2 acc[2] = {{0.f, 0.f}, {0.f, 0.f}};
        f2 w[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          cr[u] = __builtin_amdgcn_readlane(mine, 8 * (r + u));
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            w[u][q] = f2{0.f, 0.f};
            if (q < cr[u]) {
              const int n = __builtin_amdgcn_readlane(n_a, 8 * (r + u) + q);
              w[u][q] = *reinterpret_cast<const f2 *>(wl + (size_t)n * K);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float cf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c_a), 8 * (r + u) + q));
            acc[u] = __builtin_elementwise_fma(f2{cf, cf}, w[u][q], acc[u]);
          }
        if (cr[0] > 8 || cr[1] > 8) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              w[u][q] = f2{0.f, 0.f};
              if (q + 8 < cr[u]) {
                const int n = __builtin_amdgcn_readlane(n_b, 8 * (r + u) + q);
                w[u][q] = *reinterpret_cast<const f2 *>(wl + (size_t)n * K);
              }
            }
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const float cf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c_b), 8 * (r + u) + q));
              acc[u] = __builtin_elementwise_fma(f2{cf, cf}, w[u][q], acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<f2 *>(&st[(wave * 8 + r + u) * LDT + 2 * lane]) = acc[u];
        __builtin_amdgcn_sched_barrier(0);
      }
      const int nov = __builtin_amdgcn_readfirstlane(cnt[TM]);
      for (int o = 0; o < nov; ++o) {                      // rows with more than CAP entries: their wave, one entry at a time
        const int rn = __builtin_amdgcn_readfirstlane(ovf_rn[o]);
        const int orow = rn >> 16;
        if ((orow >> 3) == wave) {
          const float cf = ovf_c[o];
          const f2 wv = *reinterpret_cast<const f2 *>(wl + (size_t)(rn & 0xffff) * K);
          f2 *dst = reinterpret_cast<f2 *>(&st[orow * LDT + 2 * lane]);
          *dst = __builtin_elementwise_fma(f2{cf, cf}, wv, *dst);
        }
      }
    }
    PB_T(2)
    // (the next tile's rows are requested only now: during the S walk their sixteen registers hold its second row of loads;
    // T and the matrix products still cover the latency)
    load_tile(nt, ynxt);
    // T: lanes = columns n of this wave's quarter, registers = its 64 columns k
#pragma unroll
    for (int gi = 0; gi < GMAX; ++gi) {
      if (gi < ngt) {
        const int base = (int)((g_first + gi) * ns - m0);
        int rt = base + ta[gi];
        const bool ok = (unsigned)rt < (unsigned)TM;
        const float cf = ok ? tg[gi] : 0.f;
        rt = ok ? rt : 0;
        const float *zr = &zt[rt * LDZ + tk0];
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
          tacc[kk] = __fmaf_rn(cf, zr[kk], tacc[kk]);
          if ((kk & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    load_idx(nt);
    PB_T(8)

    // ---- (C) matrix products ----
    {
      const float *za = &zt[(rb * 32 + (lane & 31)) * LDZ + (lane >> 5)];
      const float *gb_ = &gl[(lane >> 5) * LDG + cb * 32 + (lane & 31)];
#pragma unroll
      for (int s = 0; s < K / 2; ++s) {
        accZ = __builtin_amdgcn_mfma_f32_32x32x2f32(za[2 * s], gb_[2 * s * LDG], accZ, 0, 0, 0);
        if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
      const float *ga = &zt[(lane >> 5) * LDZ + oi * 32 + (lane & 31)];
      const float *gb = &zt[(lane >> 5) * LDZ + oj * 32 + (lane & 31)];
#pragma unroll
      for (int s = 0; s < TM / 2; ++s) {
        accG[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[2 * s * LDZ], gb[2 * s * LDZ], accG[0], 0, 0, 0);
        if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
      const float *sa = &zt[((lane >> 5) + 16 * sq) * LDZ + si * 32 + (lane & 31)];     // rows 16 sq .. 16 sq + 15
      const float *sb = &zt[((lane >> 5) + 16 * sq) * LDZ + sj * 32 + (lane & 31)];
#pragma unroll
      for (int s = 0; s < TM / 8; ++s) accG[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[2 * s * LDZ], sb[2 * s * LDZ], accG[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    PB_T(3)
    __syncthreads();
    PB_T(4)
    // ---- (D) a G + v + S into the staging tile; the row lists are free again ----
    {
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = st[(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + cb * 32 + (lane & 31)];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + cb * 32 + (lane & 31)] = (accZ[r] + vreg) + t[r];
        accZ[r] = 0.f;
      }
      if (tid <= TM) cnt[tid] = 0;
    }
    PB_T(5)
    __syncthreads();
    PB_T(6)
    // ---- (E) mask, statistics, store ----
    {
      const rsrc_t rso = make_rsrc(a.Gout + (size_t)m0 * K, (M - m0) * K * 4);
      const f32x4 mu = *reinterpret_cast<const f32x4 *>(&prm[4 * c4]);
      const f32x4 rs = *reinterpret_cast<const f32x4 *>(&prm[K + 4 * c4]);
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int row = r0 + RP * i;
        const f32x4 q = *reinterpret_cast<const f32x4 *>(&st[row * LDT + 4 * c4]);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g = zt[row * LDZ + 4 * c4 + j] > 0.f ? q[j] : 0.f;
          o[j] = g;
          cs1[j] += g;
          cs2[j] = __fmaf_rn(g, (ycur[i][j] - mu[j]) * rs[j], cs2[j]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), rso,
                                               yoff, i * RP * K * 4, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NPASS; ++i) ycur[i] = ynxt[i];
    PB_T(7)
  }
#ifdef PB_PROF
  if (lane == 0 && blockIdx.x < 64) {
    long long *dst = (long long *)a.Gout + (blockIdx.x * 8 + wave) * 10;      // (experiment build only: clobbers the first rows of the output)
    for (int i = 0; i < 10; ++i) dst[i] = prof[i];
  }
#endif

  // ---- flush ----
  float *prec = a.part + (size_t)blockIdx.x * (K * K + K + (size_t)N * K);
  __syncthreads();
  float *rbuf = st;                                      // [THREADS][4], three rounds
  float t3[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int round = 0; round < 3; ++round) {
#pragma unroll
    for (int j = 0; j < 4; ++j) rbuf[tid * 4 + j] = round == 0 ? zsum[j] : (round == 1 ? cs1[j] : cs2[j]);
    __syncthreads();
    if (tid < K) {
      float t = 0.f;
      for (int q = 0; q < RP; ++q) t += rbuf[((q * CG + (tid >> 2)) * 4) + (tid & 3)];
      t3[round] = t;
    }
    __syncthreads();
  }
  if (tid < K) {
    prec[K * K + tid] = t3[0];
    atomicAdd(a.sums + tid, (double)t3[1]);
    atomicAdd(a.sums + K + tid, (double)t3[2]);
  }
  // Gram blocks: the wave's own block and its mirror image; the quarter-partials of the two shared blocks meet in LDS
  float *gq = st;                                        // [8 waves][16][64] (32 KB of the staging tile)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 31;
    prec[(oi * 32 + i) * K + oj * 32 + j] = accG[0][r];
    if (oi != oj) prec[(oj * 32 + j) * K + oi * 32 + i] = accG[0][r];
    gq[(wave * 16 + r) * 64 + lane] = accG[1][r];
  }
  __syncthreads();
  if (sq == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 31;
      const float v = (gq[(wave * 16 + r) * 64 + lane] + gq[((wave + 1) * 16 + r) * 64 + lane]) +
                      (gq[((wave + 2) * 16 + r) * 64 + lane] + gq[((wave + 3) * 16 + r) * 64 + lane]);
      prec[(si * 32 + i) * K + sj * 32 + j] = v;
      prec[(sj * 32 + j) * K + si * 32 + i] = v;
    }
  }
  float *pt = prec + K * K + K;
  if (tn0 + lane < N) {
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) pt[(size_t)(tn0 + lane) * K + tk0 + kk] = tacc[kk];
  }
}

// G = W^T diag(c2) W, v = W^T c3, W' = diag(c1) W  (consts = [c1 | c2 | c3] x N from pn2_bn_bwd_consts)
__global__ __launch_bounds__(128) void pool_bwd_setup_kernel(int N, int K, const float *__restrict__ W,
                                                            const float *__restrict__ consts, float *__restrict__ G,
                                                            float *__restrict__ v, float *__restrict__ Wp) {
  const int j = blockIdx.x, k = threadIdx.x;             // grid K + 1 + N blocks, K threads
  if (k >= K) return;
  // (four independent partial sums: as ONE chain the N dependent fp64 adds behind L2 loads were 60 us for a 128 x 128 matrix)
  if (j < K) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    int n = 0;
    for (; n + 3 < N; n += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        s[u] += (double)consts[N + n + u] * (double)W[(size_t)(n + u) * K + j] * (double)W[(size_t)(n + u) * K + k];
    }
    for (; n < N; ++n) s[0] += (double)consts[N + n] * (double)W[(size_t)n * K + j] * (double)W[(size_t)n * K + k];
    G[j * K + k] = (float)((s[0] + s[1]) + (s[2] + s[3]));
  } else if (j == K) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    int n = 0;
    for (; n + 3 < N; n += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] += (double)consts[2 * N + n + u] * (double)W[(size_t)(n + u) * K + k];
    }
    for (; n < N; ++n) s[0] += (double)consts[2 * N + n] * (double)W[(size_t)n * K + k];
    v[k] = (float)((s[0] + s[1]) + (s[2] + s[3]));
  } else {
    const int n = j - K - 1;
    Wp[(size_t)n * K + k] = consts[n] * W[(size_t)n * K + k];
  }
}

// fp64 sums of the per-workgroup partial records: out[e] += sum over this block's slice of the records (out zeroed by
// the caller; grid.y slices of the records so that the sum is not one thread's serial chain of nblk dependent adds)
__global__ __launch_bounds__(256) void pool_bwd_reduce_kernel(int nblk, int count, const float *__restrict__ part,
                                                             double *__restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= count) return;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int b = b0;
  for (; b + 3 < b1; b += 4) {
    s0 += (double)part[(size_t)b * count + e];
    s1 += (double)part[(size_t)(b + 1) * count + e];
    s2 += (double)part[(size_t)(b + 2) * count + e];
    s3 += (double)part[(size_t)(b + 3) * count + e];
  }
  for (; b < b1; ++b) s0 += (double)part[(size_t)b * count + e];
  atomicAdd(out + e, (s0 + s1) + (s2 + s3));
}

// dW[n][k] = c1[n] T[n][k] + c2[n] sum_j W[n][j] Z[j][k] + c3[n] s[k]   (red = reduced record: Z | s | T)
__global__ __launch_bounds__(128) void pool_bwd_assemble_kernel(int N, int K, const float *__restrict__ W,
                                                               const float *__restrict__ consts,
                                                               const double *__restrict__ red, float *__restrict__ dW) {
  const int n = blockIdx.x, k = threadIdx.x;
  if (k >= K) return;
  const double *Z = red, *S = red + K * K, *T = red + K * K + K;
  double w4[4] = {0.0, 0.0, 0.0, 0.0};                  // (K is 64 or 128: four independent chains, see the setup kernel)
  for (int j = 0; j < K; j += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) w4[u] += (double)W[(size_t)n * K + j + u] * Z[(j + u) * K + k];
  }
  const double wz = (w4[0] + w4[1]) + (w4[2] + w4[3]);
  dW[(size_t)n * K + k] =
      (float)((double)consts[n] * T[(size_t)n * K + k] + (double)consts[N + n] * wz + (double)consts[2 * N + n] * S[k]);
}

int pool_bwd_grid(int K, long long ntiles) {
  const long long g = 256;                               // one workgroup per CU
  return (int)(ntiles < g ? ntiles : g);
}

}  // namespace

extern "C" int pn2_pool_bwd_supported(int N, int K, int ns) {
  if (!((K == 64 || K == 128) && N >= 1 && N <= 256 && (ns == 16 || ns == 32 || ns == 64 || ns == 128))) return 0;
  const int ngt = ns >= 64 ? 1 : 64 / ns;
  if (K == 64 && N <= 128) return ngt * (N <= 64 ? 64 : 128) <= 512;      // one (group, column) entry per thread
  // K = 128: W' (N x 128 floats) is gathered from L2 — N / ns entries per row and 512 bytes each.  At ns = 16 (16 entries
  // per row on average, 512 KB of W' per 64-row tile against 32 KB of y) the gather, not the matrix products, is the
  // kernel's time (measured 1.24 ms vs 0.46 ms for the materialised path at 256k rows): such layers stay on that path
  if (K == 128) return ns >= (getenv("PN2_POOL_K128_NS16") ? 16 : 32) && getenv("PN2_POOL_K128_OFF") == nullptr;
  return ngt * 4 <= 8;                                                     // K = 64, N > 128: the index-register kernel
}

extern "C" size_t pn2_pool_bwd_workspace_bytes(long long M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const long long ntiles = (M + TM - 1) / TM;
  const size_t g = (size_t)pool_bwd_grid(K, ntiles);
  const size_t rec = (size_t)K * K + K + (size_t)N * K;
  // G | v | W' | per-workgroup partial records | reduced record (fp64)
  size_t b = ((size_t)(K * K + K + N * K) * 4 + 255) / 256 * 256;
  b += (g * rec * 4 + 255) / 256 * 256;
  b += rec * 8;
  return b;
}

// Backward of the pooled last layer in Gram form (see the head of this file): replaces group_points_grad-free parts of
// F.max_pool2d + BatchNorm2d + Conv2d backward (autograd of OPS/pointnet2_modules.py:58-70) for that layer.
//   consts [3][N] from pn2_bn_bwd_consts of this layer; arg / gPm [M/ns][N] from pn2_pool_finalize / pn2_pool_bwd_prep;
//   Yp [M][K], fin_p [4][K] of the layer below; Gout [M][K]; sums [2][K] fp64 ACCUMULATES; dW [N][K] is written.
namespace {
int pool_bwd_impl(bool x3, long long M, int N, int K, int ns, const float *Yp, const float *fin_p, const float *W,
                  const float *consts, const int *arg, const float *gPm, float *Gout, double *sums, float *dW, void *workspace,
                  size_t workspace_bytes, void *stream) {
  if (M < 0 || !pn2_pool_bwd_supported(N, K, ns) || M % ns) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!Yp || !fin_p || !W || !consts || !arg || !gPm || !Gout || !sums || !dW || !workspace) return PN2_ENULL;
  if (workspace_bytes < pn2_pool_bwd_workspace_bytes(M, N, K)) return PN2_ENOSPC;
  hipStream_t s = (hipStream_t)stream;
  const long long ntiles = (M + TM - 1) / TM;
  const int grid = pool_bwd_grid(K, ntiles);
  char *ws = (char *)workspace;
  float *G = (float *)ws, *v = G + K * K, *Wp = v + K;
  size_t off = ((size_t)(K * K + K + N * K) * 4 + 255) / 256 * 256;
  const size_t rec = (size_t)K * K + K + (size_t)N * K;
  float *part = (float *)(ws + off);
  off += ((size_t)grid * rec * 4 + 255) / 256 * 256;
  double *red = (double *)(ws + off);
  if (hipMemsetAsync(red, 0, rec * 8, s) != hipSuccess) return PN2_ELAUNCH;

  hipLaunchKernelGGL(pool_bwd_setup_kernel, dim3(K + 1 + N), dim3(128), 0, s, N, K, W, consts, G, v, Wp);
  PoolBwdArgs a;
  a.Yp = Yp; a.finp = fin_p; a.G = G; a.v = v; a.Wp = Wp; a.arg = arg; a.gPm = gPm; a.Gout = Gout; a.sums = sums;
  a.part = part; a.M = M; a.N = N; a.ns = ns;
  if (K == 64) {
    if (N <= 64 && x3) hipLaunchKernelGGL((pool_bwd64_kernel<1, true>), dim3(grid), dim3(512), 0, s, a);
    else if (N <= 64) hipLaunchKernelGGL((pool_bwd64_kernel<1>), dim3(grid), dim3(512), 0, s, a);
    else if (N <= 128 && x3) hipLaunchKernelGGL((pool_bwd64_kernel<2, true>), dim3(grid), dim3(512), 0, s, a);
    else if (N <= 128) hipLaunchKernelGGL((pool_bwd64_kernel<2>), dim3(grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((pool_bwd_kernel<2, 64, false>), dim3(grid), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(pool_bwd128_kernel, dim3(grid), dim3(512), 0, s, a);
  }
  const int slices = grid >= 64 ? 16 : 1;
  hipLaunchKernelGGL(pool_bwd_reduce_kernel, dim3((unsigned)((rec + 255) / 256), slices), dim3(256), 0, s, grid, (int)rec,
                     part, red);
  hipLaunchKernelGGL(pool_bwd_assemble_kernel, dim3(N), dim3(128), 0, s, N, K, W, consts, red, dW);
  return pn2_check_launch();
}
}  // namespace

extern "C" int pn2_pool_bwd(long long M, int N, int K, int ns, const float *Yp, const float *fin_p, const float *W,
                            const float *consts, const int *arg, const float *gPm, float *Gout, double *sums,
                            float *dW, void *workspace, size_t workspace_bytes, void *stream) {
  return pool_bwd_impl(false, M, N, K, ns, Yp, fin_p, W, consts, arg, gPm, Gout, sums, dW, workspace, workspace_bytes, stream);
}

// pn2_pool_bwd with the matrix products of the K = 64, N <= 128 kernel (a G, a^T a) on the f32x3 product (x3_common.h); every
// other shape runs the exact kernels.  Same arguments, workspace and outputs.
extern "C" int pn2_x3_pool_bwd(long long M, int N, int K, int ns, const float *Yp, const float *fin_p, const float *W,
                               const float *consts, const int *arg, const float *gPm, float *Gout, double *sums,
                               float *dW, void *workspace, size_t workspace_bytes, void *stream) {
  return pool_bwd_impl(true, M, N, K, ns, Yp, fin_p, W, consts, arg, gPm, Gout, sums, dW, workspace, workspace_bytes, stream);
}
