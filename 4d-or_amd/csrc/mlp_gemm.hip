// mlp_gemm.hip — fused fp32-MFMA kernels for the shared per-point MLPs.
//
// The reference runs each SA/FP "shared MLP" as nn.Conv2d(1x1, bias=False) +
// BatchNorm2d + ReLU per layer (OPS/pointnet2_modules.py:9-19) followed by
// F.max_pool2d (:67-70): per layer that is one GEMM plus 3-4 full passes over a
// (B*npoint*nsample, C) tensor in the forward and 5 more in the backward.  On
// MI355X those passes, not the FLOPs, dominate (profiles/r01_bench_v1_summary.md:
// 25 ms of BN/ReLU kernels + 30 ms of extremely skinny hipBLASLt GEMMs per step).
//
// Here a layer is ONE tall-skinny GEMM kernel on the f32 MFMA path
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, 157 TF peak):
//
//     out[M, N] = pro(X)[M, K] * W[N, K]^T        M = rows (10^5..10^7), K, N <= 512
//
//   prologue `pro` (applied while staging the A tile into LDS, never materialised):
//     PRO_NONE    x
//     PRO_BNRELU  relu(x * p0[k] + p1[k])            = ReLU(BatchNorm(y_{l-1})) of the previous layer
//     PRO_GY      p0[k]*g + p1[k]*y + p2[k]          = dL/dy_l from dL/dz_l (BatchNorm backward, see below)
//     PRO_POOLG   same, with g gathered from the pooled gradient through the arg-max
//     PRO_FIRST   relu((X0 W0^T) * p0[k] + p1[k])    = the stack's FIRST layer recomputed from its <= 8-column input rows
//                                                      X0 [M][K0] (the grouped xyz / colour rows): y_0 is never stored
//     PRO_LIFT    relu((Pq[gidx[row]][k] - Q[row / ns][k]) * p0[k] + p1[k])
//                                                    = a LIFTED first layer (group_lift.hip) re-formed from its per-point
//                                                      products: y_0 = Pq[point] - Q[centre] is never stored either; the
//                                                      rows of Pq (a cloud's share: 1 MB at the headline's SA2) come out of L2
//   epilogue:
//     EPI_STATS   column sums of out and out^2 (fp64 atomics)  -> BatchNorm batch statistics
//     EPI_MASK    out *= [BN(yprev) > 0]; column sums of out and out * yhat_prev
//                                                             -> ReLU backward + BN-backward reductions
//     EPI_MASKL   EPI_MASK with yprev[row] = Pq[gidx[row]] - Q[row / ns] (the lifted first layer below, not stored)
//     EPI_POOL    column sums as EPI_STATS, and per group of `ns` consecutive rows the maximum of every column
//                 with its row index; `out` is NOT stored (the max-pooled last layer of an SA stack: BatchNorm with
//                 a positive scale and ReLU are monotone, so max(relu(bn(y))) = relu(bn(max y)); columns with a
//                 negative gamma arrive with their weight row negated, `sgn` restores the column sums)
//
// BatchNorm backward in this formulation: with yhat = (y-mean)*rstd, z = gamma*yhat+beta,
// g = dL/dz, dbeta = sum g, dgamma = sum g*yhat (both produced by the previous kernel's
// epilogue), dL/dy = gamma*rstd*(g - dbeta/M - yhat*dgamma/M) = c1*g + c2*y + c3 with
// per-channel constants — so it folds into the next GEMM's prologue.
//
// Tiling: workgroup = 4 waves = 128 rows x (NT*32) columns; each wave owns 32 rows
// and NT accumulators of 32x32 (16 VGPRs each); K is walked in chunks of 16/32 staged
// through double-buffered LDS with +1 padding (conflict-free ds_read_b32 fragment
// reads), global loads of the next chunk in flight behind the current chunk's MFMAs.
#include "pn2_common.h"
#include "mlp_common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {


struct GemmArgs {
  const float *X;    // [M][K]  (PRO_GY: g = dL/dz [M][K]; PRO_POOLG: unused)
  const float *X2;   // PRO_GY / PRO_POOLG: y [M][K]
  const float *p0, *p1, *p2;
  const int *arg;    // PRO_POOLG: [M/ns][K] arg-max sample per (row group, channel)
  const float *gP;   // PRO_POOLG: [M/ns][K] pooled gradient (already masked by pooled > 0)
  const float *W;    // [N][K]
  float *Y;          // [M][N]
  double *stats;     // [2][N]
  const float *Yprev;                              // EPI_MASK: [M][N]
  const float *e_scale, *e_shift, *e_mean, *e_rstd;  // EPI_MASK: per output column
  long long M;
  int K, N, ns, pro, epi;
  // EPI_POOL
  float *pmax;       // [M/psz][N], psz = min(ns, 32): maximum of the (sign-adjusted) raw output over each partial row group
  int *parg;         // [M/psz][N] row of that maximum inside the partial group (first one among equals)
  const float *sgn;  // [N] +-1: sign the weight rows were multiplied with
  // PRO_FIRST: X = X0 [M][K0] (input rows of the stack), W0 [K][K0] (first layer's weight), p0 / p1 = its BatchNorm scale / shift
  const float *W0;
  int K0;
  // PRO_LIFT / EPI_MASKL: X (PRO_LIFT) resp. Yprev (EPI_MASKL) = Pq [lrows][K resp. N] per-point products incl. the coordinate
  // term, lidx [M] row of Pq every grouped row reads (cloud offset included), lQ [M/ns][K resp. N] per-centre term
  const int *lidx;
  const float *lQ;
  long long lrows;
};

constexpr int BM = 128;

// Persistent, software-pipelined workgroups.  A workgroup walks row tiles
// blockIdx.x, blockIdx.x + gridDim.x, ... and the (tile, K-chunk) steps form ONE
// pipeline over a two-deep register ring: the global loads of step s+2 are issued before
// the MFMAs of step s, the registers of step s+1 (loaded one iteration earlier) are
// transformed (prologue) and written to LDS buffer (s+1)&1 after them; one barrier per
// step.  Prologue / epilogue modes are TEMPLATE parameters and the loop body is straight-line
// code: with conditionals around loads hipcc's s_waitcnt insertion degrades to vmcnt(0) in
// front of the MFMA block and nothing overlaps.
// Column sums for the epilogue reductions stay in registers across tiles and are flushed
// once per workgroup.
//
// CW = wave columns: the workgroup is 4 x CW waves; wave (wr, wc) owns rows wr*32.. and the
// NT column tiles wc*NT.. (CW = 2 keeps N = 256/288 at 64-80 accumulator registers per wave).
template <int NT, int KC, int CW, int PRO, int EPI>
__global__ __launch_bounds__(256 * CW, 2) void mlp_gemm_kernel(const GemmArgs a) {
  constexpr int THREADS = 256 * CW;
  constexpr int NTT = NT * CW;                 // column tiles per workgroup
  constexpr int LD = KC + 1;
  constexpr int APT = BM * KC / THREADS;       // A elements per thread per chunk
  constexpr int WPT = NTT * 32 * KC / THREADS; // W elements per thread per chunk
  constexpr int RSTEP = THREADS / KC;          // rows covered by one pass of the workgroup
  constexpr bool MASKL = EPI == EPI_MASKL;     // EPI_MASK whose previous layer is a lifted first layer (gathered, not stored)
  constexpr bool MASKE = EPI == EPI_MASK || MASKL;
  constexpr bool TWO = PRO == PRO_GY;          // second A matrix (y) needed
  constexpr bool LIFT = PRO == PRO_LIFT;       // A rows gathered from the per-point products, minus the per-centre term
  constexpr bool RB = TWO || LIFT;             // a second value per A element rides the ring
  constexpr bool POOL = PRO == PRO_POOLG;      // dense c2*y+c3 from ONE matrix + sparse arg-max patch in LDS
  __shared__ float As[2][BM * LD];
  __shared__ float Ws[2][NTT * 32 * LD];
  __shared__ float red[2][NTT * 32];
  constexpr bool POOLE = EPI == EPI_POOL;
  // PRO_FIRST: the A tile is COMPUTED while it is staged: a[row][k] = relu(bn_0(X0[row] . W0[k])), eight FMAs per element
  // from the 128 x 8 input tile (LDS, two copies by tile parity) and W0's row k (LDS, [Kpad][8], zero padded).
  // The input rows ride the register ring in place of the A registers, one STEP AHEAD of the weights: a tile's rows must
  // be in LDS (and behind a barrier) before its first chunk is staged.
  constexpr bool FIRST = PRO == PRO_FIRST;
  constexpr int XW = 8;
  constexpr int XPT = BM * XW / THREADS;       // input-tile elements per thread
  constexpr int ARN = FIRST ? XPT : APT;       // A registers of a ring entry
  __shared__ __attribute__((aligned(16))) float Xs[FIRST ? 2 * BM * XW : 4];
  // per-input-column prologue parameters (p0, p1, p2), staged once and ZERO beyond K: a padded
  // column then evaluates to relu(0*x + 0) = 0 (v_max_f32 drops a NaN operand) resp. 0*g + 0*y + 0,
  // so ragged K needs no masks.  (A global load inside the step loop would be the youngest entry of
  // the in-order vmcnt queue and waiting for it would drain the whole prefetch ring.)
  extern __shared__ __attribute__((aligned(16))) float prm[];   // [3][Kpad], Kpad = nchunks * KC (+ PRO_FIRST: W0 as [Kpad][8])

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = (tid >> 6) & 3;            // row block of this wave
  const int wcol = tid >> 8;                  // column block of this wave
  const int n0 = blockIdx.y * (NTT * 32);
  const int K = a.K, N = a.N;
  const long long M = a.M;
  const int nchunks = (K + KC - 1) / KC;
  const int Kpad = nchunks * KC;
  const long long ntiles = (M + BM - 1) / BM;
  const long long my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;   // >= 1 (grid <= ntiles)
  const long long total_steps = my_tiles * nchunks;
  const long long last_tile = blockIdx.x + (my_tiles - 1) * gridDim.x;

  if (PRO != PRO_NONE) {
    for (int i = tid; i < Kpad; i += THREADS) {
      const bool in = i < K;
      prm[i] = in ? a.p0[i] : 0.f;
      prm[Kpad + i] = in ? a.p1[i] : 0.f;
      if (TWO || POOL) prm[2 * Kpad + i] = in ? a.p2[i] : 0.f;
    }
  }
  float *w0s = prm + 3 * Kpad;
  const int K0 = FIRST ? a.K0 : 0;
  // PRO_FIRST: thread -> (row tid / 8 + (THREADS / 8) i, column tid % 8) of an input tile; columns past K0 read zeros
  const int xoff = (FIRST && (tid % XW) < K0) ? ((tid / XW) * K0 + (tid % XW)) * 4 : kOobOffset;
  const int xpass = (THREADS / XW) * K0 * 4;
  if (FIRST) {
    for (int i = tid; i < Kpad * XW; i += THREADS) {
      const int k = i / XW, j = i % XW;
      w0s[i] = (k < K && j < K0) ? a.W0[(size_t)k * K0 + j] : 0.f;
    }
    // input rows of the first tile (every later tile arrives through the ring)
    const long long xm0 = (long long)blockIdx.x * BM;
    const rsrc_t rsx = make_rsrc(a.X + (size_t)xm0 * K0, (M - xm0) * K0 * 4);
#pragma unroll
    for (int i = 0; i < XPT; ++i) Xs[tid + THREADS * i] = bload(rsx, xoff, i * xpass);
  }

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float cs1[NT], cs2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { cs1[t] = 0.f; cs2[t] = 0.f; }

  const int kk = tid % KC;
  const int r0 = tid / KC;
  // kernel-constant per-lane byte offsets
  // (the row pass i adds a wave-uniform i * RSTEP * K * 4, carried in the SGPR offset with the chunk)
  const int aoff = (r0 * K + kk) * 4, apass = RSTEP * K * 4;
  const rsrc_t rsW = make_rsrc(a.W + (size_t)n0 * K, (long long)(N - n0) * K * 4);

  // PRO_POOLG: dL/dz of a max-pooled layer has ONE non-zero per (row group, channel) — at the
  // arg-max row.  The tile is staged as the dense part c2*y + c3 and thread (gi, kk) adds
  // c1 * gP[group gi of the tile][k] at LDS row arg (if that row lies in this tile).  The patch
  // operands ride the register ring like everything else.  PGR = patch entries per thread.
  constexpr int PGR = POOL ? (((BM / 16 + 1) * KC + THREADS - 1) / THREADS) : 1;   // supports ns >= 16
  const long long ngroups = POOL ? (M + a.ns - 1) / a.ns : 0;

  float ra0[ARN], rb0[RB ? APT : 1], rw0[WPT], pg0[PGR];
  float ra1[ARN], rb1[RB ? APT : 1], rw1[WPT], pg1[PGR];
  int pa0[PGR], pa1[PGR];

  // PRO_LIFT: pass i of thread (r0, kk) stages row r0 + RSTEP i of the tile: its row of Pq (gi, loaded one step ahead) and
  // its centre's row of Q — group (r0 + RSTEP i) / ns of the tile = r0 / ns + (RSTEP i) / ns (ns and RSTEP powers of two,
  // ns >= 16; BM % ns == 0: checked by the caller), the second term wave-uniform
  int gi[LIFT ? APT : 1];
  int qstep[LIFT ? APT : 1];
  const long long lgroups = LIFT ? (M + a.ns - 1) / a.ns : 0;
  const int qoff = LIFT ? ((r0 / a.ns) * K + kk) * 4 : 0;
  if (LIFT) {
    const long long fm0 = (long long)blockIdx.x * BM;
    const rsrc_t rsi = make_rsrc(a.lidx + fm0, (M - fm0) * 4);
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      gi[LIFT ? i : 0] = bload_i(rsi, r0 * 4, i * RSTEP * 4);
      qstep[LIFT ? i : 0] = ((RSTEP * i) / a.ns) * K * 4;
    }
  }

  // (tile, chunk) cursors: L = next step to LOAD, S = next step to STORE to LDS, C = step computed.
  // Past the end the L/S cursors stay on the last tile: the surplus loads / LDS writes are
  // harmless and keep the loop body free of conditionals.
  long long l_tile = blockIdx.x, s_tile = blockIdx.x, c_tile = blockIdx.x;
  int l_chunk = 0, s_chunk = 0, c_chunk = 0;

  auto load_step = [&](float (&ra)[ARN], float (&rb)[RB ? APT : 1], float (&rw)[WPT], int (&pa)[PGR],
                       float (&pg)[PGR]) {
    const long long m0 = l_tile * BM;
    const long long left = (M - m0) * K * 4;                // bytes from the tile's first row to the end
    const int soff = l_chunk * KC * 4;
    if (POOL) {
      const rsrc_t rs = make_rsrc(a.X2 + (size_t)m0 * K, left);
#pragma unroll
      for (int i = 0; i < APT; ++i) ra[i] = bload(rs, aoff, soff + i * apass);
      const long long g_first = m0 / a.ns;
      const rsrc_t rsa = make_rsrc(a.arg + (size_t)g_first * K, (ngroups - g_first) * K * 4);
      const rsrc_t rsg = make_rsrc(a.gP + (size_t)g_first * K, (ngroups - g_first) * K * 4);
#pragma unroll
      for (int e = 0; e < PGR; ++e) {                       // patch entry (group r0 + RSTEP*e, column kk)
        pa[e] = bload_i(rsa, aoff, soff + e * apass);
        pg[e] = bload(rsg, aoff, soff + e * apass);
      }
    } else if (LIFT) {
      // byte offsets of this step's rows of Pq from the indices that arrived with the PREVIOUS step's loads, then — before
      // this step's loads, so that waiting for them next time does not wait for these — the indices of the NEXT step's tile
      int vo[APT];
#pragma unroll
      for (int i = 0; i < APT; ++i) vo[LIFT ? i : 0] = gi[LIFT ? i : 0] * (K * 4) + kk * 4;
      const bool nwrap = l_chunk + 1 == nchunks;
      const long long nt_ = l_tile + (nwrap ? (long long)gridDim.x : 0ll);
      const long long nm0 = (nt_ < ntiles ? nt_ : last_tile) * BM;
      const rsrc_t rsi = make_rsrc(a.lidx + nm0, (M - nm0) * 4);
#pragma unroll
      for (int i = 0; i < APT; ++i) gi[LIFT ? i : 0] = bload_i(rsi, r0 * 4, i * RSTEP * 4);
      __builtin_amdgcn_sched_barrier(0);
      const rsrc_t rsp = make_rsrc(a.X, a.lrows * K * 4);
#pragma unroll
      for (int i = 0; i < APT; ++i) ra[i] = bload(rsp, vo[LIFT ? i : 0], soff);
      const long long g_first = m0 / a.ns;
      const rsrc_t rsq = make_rsrc(a.lQ + (size_t)g_first * K, (lgroups - g_first) * K * 4);
#pragma unroll
      for (int i = 0; i < APT; ++i) rb[RB ? i : 0] = bload(rsq, qoff, soff + qstep[LIFT ? i : 0]);
    } else if (!FIRST) {
      const rsrc_t rs = make_rsrc(a.X + (size_t)m0 * K, left);
#pragma unroll
      for (int i = 0; i < APT; ++i) ra[i] = bload(rs, aoff, soff + i * apass);
      if (TWO) {
        const rsrc_t rs2 = make_rsrc(a.X2 + (size_t)m0 * K, left);
#pragma unroll
        for (int i = 0; i < APT; ++i) rb[TWO ? i : 0] = bload(rs2, aoff, soff + i * apass);
      }
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) rw[i] = bload(rsW, aoff, soff + i * apass);
    ++l_chunk;
    const bool wrap = l_chunk == nchunks;
    l_chunk = wrap ? 0 : l_chunk;
    const long long nt = l_tile + (wrap ? (long long)gridDim.x : 0ll);
    l_tile = nt < ntiles ? nt : last_tile;
    if (FIRST) {
      // input rows of the tile of the NEXT step (the cursor has just moved there)
      const long long xm0 = l_tile * BM;
      const rsrc_t rsx = make_rsrc(a.X + (size_t)xm0 * K0, (M - xm0) * K0 * 4);
#pragma unroll
      for (int i = 0; i < XPT; ++i) ra[FIRST ? i : 0] = bload(rsx, xoff, i * xpass);
    }
  };

  // prologue transform + LDS write of the OLDEST loaded step.  No masks: rows past M and weight
  // rows past N arrive as zeros (out-of-range loads), ragged K is absorbed by the zero-padded
  // parameter table (PRO_NONE: one select).  Rows past M of a partial last tile do leave the
  // prologue non-zero; the epilogue clears their accumulators before the statistics.
  long long p_tile = blockIdx.x;   // cursor of the step whose sparse patch is pending (PRO_POOLG)
  int p_chunk = 0;
  int s_par = 0;                   // PRO_FIRST: parity of the tile the store cursor is in (selects the input-tile copy)
  auto store_step = [&](float (&ra)[ARN], float (&rb)[RB ? APT : 1], float (&rw)[WPT], int buf) {
    p_tile = s_tile;
    p_chunk = s_chunk;
    const int k = s_chunk * KC + kk;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
    if (PRO != PRO_NONE) {
      q0 = prm[k];
      q1 = prm[Kpad + k];
      if (TWO || POOL) q2 = prm[2 * Kpad + k];
    }
    const bool kin = k < K;
    float *Ad = &As[buf][r0 * LD + kk];
    float4 wa = {0.f, 0.f, 0.f, 0.f}, wb = wa;
    const float *xt = Xs + s_par * (BM * XW) + r0 * XW;
    if (FIRST) {
      const float4 *wq = reinterpret_cast<const float4 *>(w0s + k * XW);
      wa = wq[0]; wb = wq[1];
    }
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      float v = 0.f;
      if (FIRST) {
        // y_0[row][k] as the MFMA would accumulate it: one FMA chain over the input columns, ascending
        const float4 *xr = reinterpret_cast<const float4 *>(xt + RSTEP * i * XW);
        const float4 xa = xr[0], xb = xr[1];
        v = __fmul_rn(xa.x, wa.x);
        v = __fmaf_rn(xa.y, wa.y, v); v = __fmaf_rn(xa.z, wa.z, v); v = __fmaf_rn(xa.w, wa.w, v);
        v = __fmaf_rn(xb.x, wb.x, v); v = __fmaf_rn(xb.y, wb.y, v); v = __fmaf_rn(xb.z, wb.z, v);
        v = __fmaf_rn(xb.w, wb.w, v);
        v = fmaxf(__fmaf_rn(v, q0, q1), 0.f);
      } else {
        v = ra[FIRST ? 0 : i];
      }
      if (PRO == PRO_NONE) v = kin ? v : 0.f;
      if (PRO == PRO_BNRELU) v = fmaxf(__fmaf_rn(v, q0, q1), 0.f);
      if (LIFT) v = fmaxf(__fmaf_rn(__fsub_rn(v, rb[RB ? i : 0]), q0, q1), 0.f);
      if (TWO) v = __fmaf_rn(q0, v, __fmaf_rn(q1, rb[TWO ? i : 0], q2));
      if (POOL) v = __fmaf_rn(q1, v, q2);
      Ad[RSTEP * i * LD] = v;
    }
    float *Wd = &Ws[buf][r0 * LD + kk];
#pragma unroll
    for (int i = 0; i < WPT; ++i) Wd[RSTEP * i * LD] = rw[i];
    ++s_chunk;
    const bool wrap = s_chunk == nchunks;
    s_chunk = wrap ? 0 : s_chunk;
    const long long nt = s_tile + (wrap ? (long long)gridDim.x : 0ll);
    s_tile = nt < ntiles ? nt : last_tile;
    if (FIRST) {
      // input rows of the next step's tile -> their copy (the same tile: the same values over themselves)
      s_par ^= wrap ? 1 : 0;
#pragma unroll
      for (int i = 0; i < XPT; ++i) Xs[s_par * (BM * XW) + tid + THREADS * i] = ra[FIRST ? i : 0];
    }
  };

  // sparse arg-max patch of the step stored last (runs between two barriers)
  auto patch_step = [&](int (&pa)[PGR], float (&pg)[PGR], int buf) {
    const long long m0 = p_tile * BM;
    const int mrem = (int)((M - m0) < (long long)BM ? (M - m0) : (long long)BM);
    const int k = p_chunk * KC + kk;
    const float c1 = prm[k];
    const long long g_first = m0 / a.ns;
    const int ngrp = (int)((m0 + mrem - 1) / a.ns - g_first) + 1;
#pragma unroll
    for (int e = 0; e < PGR; ++e) {
      const int gi = r0 + RSTEP * e;
      const long long row = (g_first + gi) * (long long)a.ns + pa[e] - m0;     // tile-relative arg-max row
      if (gi < ngrp && k < K && row >= 0 && row < mrem) As[buf][(int)row * LD + kk] += c1 * pg[e];
    }
  };

  const int arow = (wave * 32 + (lane & 31)) * LD + (lane >> 5);
  const int brow = (wcol * NT * 32 + (lane & 31)) * LD + (lane >> 5);
  const int cl = lane & 31;
  const int rbase = wave * 32 + 4 * (lane >> 5);
  // output addressing: lane-constant byte offset per column tile (out of range for a column past N),
  // row r of the accumulator adds the wave-uniform (r&3 + 8*(r>>2)) * N * 4
  int yoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = n0 + (wcol * NT + t) * 32 + cl;
    yoff[t] = col < N ? (rbase * N + col) * 4 : kOobOffset;
  }
  const int rowpitch = N * 4;
  // EPI_POOL: partial groups of psz = min(ns, 32) rows; kernel-constant per-lane offsets of the lane's partial result
  const bool psz16 = POOLE && a.ns == 16;
  const int psh = psz16 ? 4 : 5;
  const long long npart = POOLE ? (M >> psh) : 0;
  int poff[POOLE ? NT : 1];
  if (POOLE) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = n0 + (wcol * NT + t) * 32 + cl;
      const int pl = psz16 ? wave * 2 + (lane >> 5) : wave;
      poff[POOLE ? t : 0] = (col < N && (psz16 || lane < 32)) ? (pl * N + col) * 4 : kOobOffset;
    }
  }
  float yp[MASKE ? NT : 1][16];
  // EPI_MASKL: Pq rows of the lane's 16 accumulator rows (four runs of four consecutive rows), the per-centre terms of the
  // wave's two 16-row halves (ns >= 16: a half lies in one group), lane-constant column offsets
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t ge[MASKL ? 4 : 1];
  float qa[MASKL ? NT : 1], qb[MASKL ? NT : 1];
  int ycol[MASKL ? NT : 1];
  const long long egroups = MASKL ? (M + a.ns - 1) / a.ns : 0;
  if (MASKL) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = n0 + (wcol * NT + t) * 32 + cl;
      ycol[MASKL ? t : 0] = col < N ? col * 4 : kOobOffset;
    }
  }
  // EPI_MASK constants of this lane's output columns (fixed for the whole kernel)
  float e_s[MASKE ? NT : 1], e_h[MASKE ? NT : 1], e_m[MASKE ? NT : 1], e_r[MASKE ? NT : 1];
  if (MASKE) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = n0 + (wcol * NT + t) * 32 + cl;
      const int cc = col < N ? col : (N - 1);
      e_s[MASKE ? t : 0] = a.e_scale[cc]; e_h[MASKE ? t : 0] = a.e_shift[cc];
      e_m[MASKE ? t : 0] = a.e_mean[cc];  e_r[MASKE ? t : 0] = a.e_rstd[cc];
    }
  }

  // one pipeline iteration: compute the current step from LDS buffer `buf`; (la, lb, lw) receive
  // the loads of two steps ahead, (sa, sb, sw) hold the next step and are written to buffer buf^1
  auto iteration = [&](int buf, float (&la)[ARN], float (&lb)[RB ? APT : 1], float (&lw)[WPT], int (&lpa)[PGR],
                       float (&lpg)[PGR], float (&sa)[ARN], float (&sb)[RB ? APT : 1], float (&sw)[WPT],
                       int (&spa)[PGR], float (&spg)[PGR]) {
    const bool last_chunk = c_chunk == nchunks - 1;
    const long long m0 = c_tile * BM;
    // Yprev first: vmcnt retires in order, so the epilogue's wait for these loads must not also
    // cover the ring loads issued after them
    if (MASKL) {
      if (c_chunk + 2 == nchunks) {            // (the caller guarantees two chunks or more)
        const rsrc_t rsi = make_rsrc(a.lidx + m0, (M - m0) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          ge[MASKL ? q : 0] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsi, rbase * 4, q * 32, 0));
      }
      if (last_chunk) {
        const rsrc_t rsp = make_rsrc(a.Yprev, a.lrows * N * 4);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            yp[MASKE ? t : 0][r] = bload(rsp, (int)ge[MASKL ? (r >> 2) : 0][r & 3] * (N * 4) + ycol[MASKL ? t : 0], 0);
        long long g0 = (m0 + wave * 32) / a.ns, g1 = (m0 + wave * 32 + 16) / a.ns;
        g0 = g0 < egroups ? g0 : egroups - 1;      // (halves past M of a partial last tile: any valid row, never used)
        g1 = g1 < egroups ? g1 : egroups - 1;
        const rsrc_t rq0 = make_rsrc(a.lQ + (size_t)g0 * N, (egroups - g0) * N * 4);
        const rsrc_t rq1 = make_rsrc(a.lQ + (size_t)g1 * N, (egroups - g1) * N * 4);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          qa[MASKL ? t : 0] = bload(rq0, ycol[MASKL ? t : 0], 0);
          qb[MASKL ? t : 0] = bload(rq1, ycol[MASKL ? t : 0], 0);
        }
      }
    } else if (MASKE) {
      if (last_chunk) {
        const rsrc_t rsp = make_rsrc(a.Yprev + (size_t)m0 * N, (M - m0) * N * 4);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            yp[MASKE ? t : 0][r] = bload(rsp, yoff[t], ((r & 3) + 8 * (r >> 2)) * rowpitch);
      }
    }
    load_step(la, lb, lw, lpa, lpg);
    const float *Ab = As[buf];
    const float *Wb = Ws[buf];
#pragma unroll
    for (int s = 0; s < KC / 2; ++s) {
      const float av = Ab[arow + 2 * s];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float bv = Wb[brow + t * 32 * LD + 2 * s];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
    // registers of the next step -> LDS before the epilogue's global stores (vmcnt counts both)
    store_step(sa, sb, sw, buf ^ 1);
    if (last_chunk) {
      // ---- tile epilogue: mask / statistics / store, straight from the accumulators ----
      if (EPI != EPI_NONE && PRO != PRO_NONE && m0 + BM > M) {
        // partial last tile (wave-uniform, at most once per kernel): rows past M must not reach the sums.
        // The empty asm keeps this a real (never taken) branch; if-converted it is 48 VALU per tile.
        asm volatile("; partial tile");
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            acc[t][r] = (m0 + rbase + (r & 3) + 8 * (r >> 2)) < M ? acc[t][r] : 0.f;
      }
      if (POOLE) {
        // ---- statistics + group maxima; nothing is stored to Y ----
        const long long pfirst = m0 >> psh;                       // first partial group of this tile
        const rsrc_t rspv = make_rsrc(a.pmax + (size_t)pfirst * N, (npart - pfirst) * N * 4);
        const rsrc_t rspr = make_rsrc(a.parg + (size_t)pfirst * N, (npart - pfirst) * N * 4);
        // lane (cl, h = lane >> 5) holds rows 4h + (r & 3) + 8 (r >> 2), r < 16, of the wave's 32-row block:
        // registers 0-7 lie in the first 16-row sub-group (rows ascending in r), registers 8-15 in the second
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float s1 = 0.f, s2 = 0.f;
          float bst[2];
          int bi[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            bst[q] = acc[t][8 * q];
            bi[q] = 0;
            s1 += bst[q];
            s2 = __fmaf_rn(bst[q], bst[q], s2);
            acc[t][8 * q] = 0.f;
#pragma unroll
            for (int r = 1; r < 8; ++r) {
              const float v = acc[t][8 * q + r];
              const bool gt = v > bst[q];
              bst[q] = gt ? v : bst[q];
              bi[q] = gt ? r : bi[q];
              s1 += v;
              s2 = __fmaf_rn(v, v, s2);
              acc[t][8 * q + r] = 0.f;
            }
          }
          cs1[t] += s1;
          cs2[t] += s2;
          // combine the two half-waves (rows interleave in blocks of four): larger value, then smaller row
          float b[2];
          int rw[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int row = (bi[q] & 3) + 8 * (bi[q] >> 2) + 4 * (lane >> 5);
            const float ob = __shfl_xor(bst[q], 32);
            const int orow = __shfl_xor(row, 32);
            const bool take = ob > bst[q] || (ob == bst[q] && orow < row);
            b[q] = take ? ob : bst[q];
            rw[q] = take ? orow : row;
          }
          // partial groups of psz = min(ns, 32) rows: ns = 16 publishes both sub-groups (lanes 0-31 the first, lanes
          // 32-63 the second), otherwise the 32 rows of the wave are one partial (first maximum wins) published by lanes 0-31
          const bool second = b[1] > b[0];
          const float b32 = second ? b[1] : b[0];
          const int r32 = second ? 16 + rw[1] : rw[0];
          const bool hi = (lane >> 5) != 0;
          const float vout = psz16 ? (hi ? b[1] : b[0]) : b32;
          const int rout = psz16 ? (hi ? rw[1] : rw[0]) : r32;
          bstore(vout, rspv, poff[POOLE ? t : 0], 0);
          __builtin_amdgcn_raw_buffer_store_b32((unsigned)rout, rspr, poff[POOLE ? t : 0], 0, 0);
          __builtin_amdgcn_sched_barrier(0);      // one column tile at a time
        }
      }
      const rsrc_t rsy = make_rsrc(POOLE ? (float *)a.pmax : a.Y + (size_t)m0 * N, POOLE ? 4 : (M - m0) * N * 4);
#pragma unroll
      for (int t = 0; t < (POOLE ? 0 : NT); ++t) {
        float es = 0.f, eh = 0.f, em = 0.f, er = 0.f;
        if (MASKE) { es = e_s[MASKE ? t : 0]; eh = e_h[MASKE ? t : 0]; em = e_m[MASKE ? t : 0]; er = e_r[MASKE ? t : 0]; }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[t][r];
          if (MASKE) {
            float y = yp[MASKE ? t : 0][r];
            if (MASKL) y = __fsub_rn(y, r < 8 ? qa[MASKL ? t : 0] : qb[MASKL ? t : 0]);
            v = (__fmaf_rn(y, es, eh) > 0.f) ? v : 0.f;
            s1 += v;
            s2 = __fmaf_rn(v, (y - em) * er, s2);
          } else if (EPI == EPI_STATS) {
            s1 += v;
            s2 = __fmaf_rn(v, v, s2);
          }
          bstore(v, rsy, yoff[t], ((r & 3) + 8 * (r >> 2)) * rowpitch);
          acc[t][r] = 0.f;
        }
        cs1[t] += s1;
        cs2[t] += s2;
      }
      c_chunk = 0;
      c_tile += gridDim.x;
    } else {
      ++c_chunk;
    }
    __syncthreads();
    if (POOL) {
      patch_step(spa, spg, buf ^ 1);
      __syncthreads();
    }
  };

  load_step(ra0, rb0, rw0, pa0, pg0);            // step 0
  load_step(ra1, rb1, rw1, pa1, pg1);            // step 1
  if (PRO != PRO_NONE) __syncthreads();          // parameter table visible
  store_step(ra0, rb0, rw0, 0);
  __syncthreads();
  if (POOL) {
    patch_step(pa0, pg0, 0);
    __syncthreads();
  }
  // Single-exit loop over step PAIRS plus a peeled odd step: there must be no control-flow edge
  // (not even a never-taken one, as the structurizer creates for a mid-loop break) from the end of
  // iteration(0) back to its own start, or hipcc guards the reuse of a ring register with
  // s_waitcnt vmcnt(0) at the loop head and the two-step prefetch distance collapses.
  for (long long pair = total_steps >> 1; pair > 0; --pair) {
    iteration(0, ra0, rb0, rw0, pa0, pg0, ra1, rb1, rw1, pa1, pg1);
    iteration(1, ra1, rb1, rw1, pa1, pg1, ra0, rb0, rw0, pa0, pg0);
  }
  if (total_steps & 1) iteration(0, ra0, rb0, rw0, pa0, pg0, ra1, rb1, rw1, pa1, pg1);

  // ---- flush the column sums once per workgroup ----
  if (EPI != EPI_NONE) {
    for (int i = tid; i < 2 * NTT * 32; i += THREADS) (&red[0][0])[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      atomicAdd(&red[0][(wcol * NT + t) * 32 + cl], cs1[t]);
      atomicAdd(&red[1][(wcol * NT + t) * 32 + cl], cs2[t]);
    }
    __syncthreads();
    for (int i = tid; i < NTT * 32; i += THREADS) {
      const int col = n0 + i;
      if (col < N) {
        const float sg = (POOLE && a.sgn) ? a.sgn[col] : 1.f;
        atomicAdd(a.stats + col, (double)(red[0][i] * sg));
        atomicAdd(a.stats + N + col, (double)red[1][i]);
      }
    }
  }
}

// ---------------------------------------------------------------- wgrad ----
// dW[N][K] += sum_r gy[r][n] * act[r][k]   (reduction over the M rows)
// workgroup = 8 waves; a column block of <= 128 K-columns (4 k-tiles) x all n-tiles
// (<= 10): wave w owns k-tile (w & 3) and the n-tiles of parity (w >> 2).
struct WgradArgs {
  // gy operand [M][N]
  const float *G;    // PRO_GY: g = dL/dz [M][N]
  const float *Yl;   // y_l [M][N]
  const float *c1, *c2, *c3;
  const int *arg;    // PRO_POOLG
  const float *gP;
  // activation operand [M][K]
  const float *X;    // raw input or y_{l-1}
  const float *a_scale, *a_shift;  // PRO_BNRELU
  float *dW;         // [N][K], accumulated with atomics (caller zero-fills)
  long long M, rows_per_wg;
  int N, K, ns, gmode /* PRO_GY | PRO_POOLG */, amode /* PRO_NONE | PRO_BNRELU */;
  int koff;          // leading activation columns (<= 3) reduced on the VALU side, MFMA part = columns [koff, K)
  // amode PRO_LIFT: the activation is relu(bn(Pq[lidx[row]] - lQ[row / ns])) — X = Pq [lrows][K] (mlp_gemm_kernel PRO_LIFT)
  const int *lidx;
  const float *lQ;
  long long lrows;
};


constexpr int WMAXN = 320;    // 10 n-tiles


// Same discipline as mlp_gemm_kernel: modes are template parameters, loads are unconditional
// from clamped addresses, the tile loop is straight-line code over a two-deep register ring
// (tile t+2 is loaded while tile t runs on the MFMAs).
// KT = 32-column k-tiles per workgroup column block (4 -> 128 K-columns; 2 for K <= 64 so that all
// eight waves — and all four SIMDs — own useful output tiles): wave w owns k-tile w % KT and the
// n-tiles (w / KT) + (8/KT)*t, t < NTW.
// LEAD: the first `koff` (1..3) activation columns — the relative xyz in front of 32k feature columns, K = 3 + 128
// or 3 + 256 in the first layer of every SA level — are reduced with plain FMAs by the threads that stage gy, so
// that the MFMA part covers an aligned column range and the ragged K does not cost a second pass over g and y
// (K = 131: 0.63 -> 0.38 ms at M = 1M).
template <int NTW, int GMODE, int AMODE, int KT, bool LEAD = false>  // n-tiles per wave (total n-tiles <= (8/KT)*NTW)
__global__ __launch_bounds__(512, 2) void mlp_wgrad_kernel(const WgradArgs a) {
  // rows per LDS tile: 32; 16 for the wide variants, whose two-deep ring of 32-row gy tiles (2 x 2 x 16..32
  // registers) plus 64-80 accumulators does not fit the register file (64 for the narrow ones: no gain)
  constexpr int WR = NTW >= 3 ? 16 : 32;
  constexpr int WKB = 32 * KT;
  constexpr int NPARS = 8 / KT;
  constexpr int GN = NPARS * NTW * 32;
  __shared__ float Gs[WR * GN];
  __shared__ float Xs[WR * WKB];
  __shared__ __attribute__((aligned(16))) float Xz[LEAD ? WR * 4 : 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int ktile = wave % KT;
  const int npar = wave / KT;
  const int koff = LEAD ? a.koff : 0;
  const int kb0 = koff + blockIdx.y * WKB;
  const bool lead_blk = LEAD && blockIdx.y == 0;   // one column block also reduces the leading columns
  const int N = a.N, K = a.K;
  const long long M = a.M;
  const long long row_begin = (long long)blockIdx.x * a.rows_per_wg;
  long long row_end = row_begin + a.rows_per_wg;
  if (row_end > M) row_end = M;
  const long long ntile = (row_end - row_begin + WR - 1) / WR;     // >= 1 by construction of the grid
  const long long last_rt = row_begin + (ntile - 1) * WR;

  f32x16 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // per-thread element coordinates inside a tile (fixed for the whole kernel): ONE gy column
  // and ONE activation column per thread, so the per-column constants are 5 registers
  constexpr int GRP = (512 / GN) ? (512 / GN) : 1;   // gy rows per pass (GN = 320: 1, 192 threads idle)
  constexpr int GPT = (WR + GRP - 1) / GRP;
  constexpr int XRP = 512 / WKB;                     // activation rows per pass (4)
  constexpr int XPT = WR / XRP;                      // 8
  const int gn = tid % GN, gr0 = tid / GN;
  const bool g_thr = tid < GRP * GN && gn < N;
  const int gnc = gn < N ? gn : (N - 1);
  const float c1 = a.c1[gnc], c2 = a.c2[gnc], c3 = a.c3[gnc];
  const int xr0 = tid / WKB, xk = tid % WKB;
  const int kx = kb0 + xk;
  const int kxc = kx < K ? kx : (K - 1);
  float a_sc = 1.f, a_sh = 0.f;
  constexpr bool LIFT = AMODE == PRO_LIFT;
  if (AMODE == PRO_BNRELU || LIFT) { a_sc = a.a_scale[kxc]; a_sh = a.a_shift[kxc]; }
  // PRO_LIFT: row xr0 + XRP i of a tile lies in its 16-row piece (XRP i) / 16 (xr0 < XRP <= 16), a piece in ONE group (ns >= 16,
  // tiles start at multiples of WR): one per-centre value per piece and thread; the rows' indices arrive one tile ahead
  constexpr int QN = LIFT ? (WR + 15) / 16 : 1;
  const long long lgroups = LIFT ? (M + a.ns - 1) / a.ns : 0;
  // (a wave stages ONE row per pass — xr0 = tid / WKB is wave-uniform for WKB >= 64 —, so the rows' indices are scalar
  // loads and the row offsets ride in the SGPR operand of the gathers: no vector register, no vector memory instruction)
  const int xr0u = __builtin_amdgcn_readfirstlane(xr0);
  int xi[LIFT ? XPT : 1];
  // (read through the constant address space: the index was written by an earlier kernel, a uniform address then loads on the
  // scalar unit)
  typedef const int __attribute__((address_space(4))) *const_int_p;
  const const_int_p lidx_c = (const_int_p)(unsigned long long)a.lidx;
  auto load_indices = [&](long long rt) {
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      long long r = rt + xr0u + XRP * i;
      r = r < M ? r : M - 1;
      xi[LIFT ? i : 0] = lidx_c[r];
    }
  };
  if (LIFT) load_indices(row_begin);
  // Addressing as in mlp_gemm_kernel: buffer descriptors per tile, kernel-constant per-lane byte
  // offsets, no clamps and no masks.  A gy column past N or an activation column past K reads a
  // neighbouring (finite) element or zero and only ever reaches dW entries that are not stored; rows
  // past M read zeros, and the one tile that can contain them clears its activation rows explicitly.
  // (the row pass i adds a wave-uniform i * rows-per-pass * pitch, carried in the SGPR offset)
  const int goff = (gr0 * N + gn) * 4, gpass = GRP * N * 4;
  const int xoff = (xr0 * K + kx) * 4, xpass = XRP * K * 4;
  // PRO_POOLG: dense part c2*y + c3 from ONE matrix; the single non-zero of dL/dz per (row group,
  // column) is patched into the LDS tile at the arg-max row (see mlp_gemm_kernel).  A WR-row tile
  // overlaps at most WR/16 + 1 groups (ns >= 16); patch entry e of thread (gn, gr0) is group gr0 + GRP*e.
  constexpr bool POOL = GMODE == PRO_POOLG;
  constexpr int WPG = POOL ? ((WR / 16 + 1 + GRP - 1) / GRP) : 1;
  constexpr int RGN = POOL ? 1 : GPT;          // the G matrix is only read in PRO_GY mode
  const long long ngroups = POOL ? (M + a.ns - 1) / a.ns : 0;

  float rg0[RGN], ry0[GPT], rx0[XPT], pg0[WPG], rq0[QN];
  float rg1[RGN], ry1[GPT], rx1[XPT], pg1[WPG], rq1[QN];
  int pa0[WPG], pa1[WPG];
  float rz0 = 0.f, rz1 = 0.f;                      // LEAD: one element of the WR x koff leading block per thread
  float lead_acc[LEAD ? 3 : 1];
#pragma unroll
  for (int c = 0; c < (LEAD ? 3 : 1); ++c) lead_acc[c] = 0.f;
  const int zr = tid >> 2, zc = tid & 3;           // (row, column) of that element
  long long l_rt = row_begin;     // next tile to load (clamped to the last tile past the end)
  long long s_rt = row_begin;     // next tile to write to LDS

  auto load_tile = [&](float (&rg)[RGN], float (&ry)[GPT], float (&rx)[XPT], int (&pa)[WPG], float (&pg)[WPG], float &rz,
                       float (&rq)[QN]) {
    const long long rt = l_rt;
    int vo[LIFT ? XPT : 1];
    if (LIFT) {
      // (scalar) offsets of this tile's rows of Pq from the indices loaded with the previous tile, then the next tile's indices
#pragma unroll
      for (int i = 0; i < XPT; ++i) vo[LIFT ? i : 0] = xi[LIFT ? i : 0] * (K * 4);
      load_indices(rt + WR < row_end ? rt + WR : last_rt);
    }
    const rsrc_t rsy = make_rsrc(a.Yl + (size_t)rt * N, (M - rt) * N * 4);
#pragma unroll
    for (int i = 0; i < GPT; ++i) ry[i] = bload(rsy, goff, i * gpass);
    if (POOL) {
      const long long g_first = rt / a.ns;
      const rsrc_t rsa = make_rsrc(a.arg + (size_t)g_first * N, (ngroups - g_first) * N * 4);
      const rsrc_t rsg = make_rsrc(a.gP + (size_t)g_first * N, (ngroups - g_first) * N * 4);
#pragma unroll
      for (int e = 0; e < WPG; ++e) {
        pa[e] = bload_i(rsa, goff, e * gpass);
        pg[e] = bload(rsg, goff, e * gpass);
      }
    } else {
      const rsrc_t rsgy = make_rsrc(a.G + (size_t)rt * N, (M - rt) * N * 4);
#pragma unroll
      for (int i = 0; i < GPT; ++i) rg[POOL ? 0 : i] = bload(rsgy, goff, i * gpass);
    }
    const rsrc_t rsx = LIFT ? make_rsrc(a.X, a.lrows * K * 4) : make_rsrc(a.X + (size_t)rt * K, (M - rt) * K * 4);
#pragma unroll
    for (int i = 0; i < XPT; ++i) rx[i] = LIFT ? bload(rsx, kx * 4, vo[LIFT ? i : 0]) : bload(rsx, xoff, i * xpass);
    if (LIFT) {
#pragma unroll
      for (int h = 0; h < QN; ++h) {
        long long g = (rt + 16 * h) / a.ns;
        g = g < lgroups ? g : lgroups - 1;
        rq[h] = bload(make_rsrc(a.lQ + (size_t)g * K, (lgroups - g) * K * 4), kx * 4, 0);
      }
    }
    if (LEAD) rz = bload(rsx, (zr < WR && zc < koff) ? (zr * K + zc) * 4 : kOobOffset, 0);
    const long long nt = rt + WR;
    l_rt = nt < row_end ? nt : last_rt;
  };

  long long p_rt = row_begin;
  auto store_tile = [&](float (&rg)[RGN], float (&ry)[GPT], float (&rx)[XPT], float rz, float (&rq)[QN]) {
    const long long rt = s_rt;
    p_rt = rt;
    if (tid < GRP * GN) {
#pragma unroll
      for (int i = 0; i < GPT; ++i) {
        const int r = gr0 + GRP * i;
        const float v = POOL ? __fmaf_rn(c2, ry[i], c3) : __fmaf_rn(c1, rg[POOL ? 0 : i], __fmaf_rn(c2, ry[i], c3));
        if (r < WR) Gs[r * GN + gn] = v;
      }
    }
    float xv[XPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      xv[i] = rx[i];
      if (AMODE == PRO_BNRELU) xv[i] = fmaxf(__fmaf_rn(xv[i], a_sc, a_sh), 0.f);
      if (LIFT) xv[i] = fmaxf(__fmaf_rn(__fsub_rn(xv[i], rq[LIFT ? (XRP * i) / 16 : 0]), a_sc, a_sh), 0.f);
    }
    if (rt + WR > M) {
      // the tile that crosses M (wave-uniform, at most one per kernel): its surplus rows carry
      // relu(shift) / c3 instead of zero — clear the activation side.  The empty asm keeps this a branch.
      asm volatile("; partial tile");
#pragma unroll
      for (int i = 0; i < XPT; ++i) xv[i] = (rt + xr0 + XRP * i) < M ? xv[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) Xs[(xr0 + XRP * i) * WKB + xk] = xv[i];
    if (LEAD && tid < WR * 4) Xz[tid] = rz;        // rows past M arrive as zeros (out of range)
    s_rt += WR;
  };

  auto patch_tile = [&](int (&pa)[WPG], float (&pg)[WPG]) {
    const long long rt = p_rt;
    const int rows = (int)((row_end - rt) < (long long)WR ? (row_end - rt) : (long long)WR);
    const long long g_first = rt / a.ns;
    const int ngrp = (int)((rt + rows - 1) / a.ns - g_first) + 1;
#pragma unroll
    for (int e = 0; e < WPG; ++e) {
      const int gi = gr0 + GRP * e;
      const long long row = (g_first + gi) * (long long)a.ns + pa[e] - rt;
      if (g_thr && gi < ngrp && row >= 0 && row < rows) Gs[(int)row * GN + gn] += c1 * pg[e];
    }
  };

  auto iteration = [&](float (&rg)[RGN], float (&ry)[GPT], float (&rx)[XPT], int (&pa)[WPG], float (&pg)[WPG], float &rz,
                       float (&rq)[QN]) {
    store_tile(rg, ry, rx, rz, rq);  // tile t (loaded two iterations ago) -> LDS
    if (POOL) {
      __syncthreads();
      patch_tile(pa, pg);
    }
    __syncthreads();
    load_tile(rg, ry, rx, pa, pg, rz, rq);   // tile t+2 into the registers just freed
    if (lead_blk && tid < GRP * GN) {
      // leading columns: this thread's gy elements (column gn, rows gr0 + GRP*i, patched values from LDS).
      // Deliberately a rolled loop: unrolled, its hoisted LDS reads cost 60 VGPRs and the second workgroup per CU.
#pragma unroll 1
      for (int i = 0; i < GPT; ++i) {
        const int r = gr0 + GRP * i;
        if (r < WR) {
          const float g = Gs[r * GN + gn];
          const float4 xz = *reinterpret_cast<const float4 *>(&Xz[LEAD ? r * 4 : 0]);
          lead_acc[0] = __fmaf_rn(g, xz.x, lead_acc[0]);
          lead_acc[LEAD ? 1 : 0] = __fmaf_rn(g, xz.y, lead_acc[LEAD ? 1 : 0]);
          lead_acc[LEAD ? 2 : 0] = __fmaf_rn(g, xz.z, lead_acc[LEAD ? 2 : 0]);
        }
      }
    }
    // A operand: A[i = n][k = r] = gy[r][n];  B operand: B[k = r][j = kcol] = act[r][kcol]
#pragma unroll
    for (int s = 0; s < WR / 2; ++s) {
      const int rr = 2 * s + (lane >> 5);
      const float bv = Xs[rr * WKB + ktile * 32 + (lane & 31)];
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const float av = Gs[rr * GN + (npar + NPARS * t) * 32 + (lane & 31)];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  };

  load_tile(rg0, ry0, rx0, pa0, pg0, rz0, rq0);
  load_tile(rg1, ry1, rx1, pa1, pg1, rz1, rq1);
  // single-exit pair loop + peeled odd tile (see mlp_gemm_kernel)
  for (long long pair = ntile >> 1; pair > 0; --pair) {
    iteration(rg0, ry0, rx0, pa0, pg0, rz0, rq0);
    iteration(rg1, ry1, rx1, pa1, pg1, rz1, rq1);
  }
  if (ntile & 1) iteration(rg0, ry0, rx0, pa0, pg0, rz0, rq0);
  if (lead_blk && g_thr) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (c < koff) atomicAdd(a.dW + (size_t)gn * K + c, lead_acc[LEAD ? c : 0]);
  }
  // ---- flush: acc[t][reg] = dW[n = ntile*32 + rowmap][k = kb0 + ktile*32 + (lane&31)] ----
  const int kcol = kb0 + ktile * 32 + (lane & 31);
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int nb = (npar + NPARS * t) * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nb + (r & 3) + 8 * (r >> 2);
      if (n < N && kcol < K) atomicAdd(a.dW + (size_t)n * K + kcol, acc[t][r]);
    }
  }
}

// ------------------------------------------------------ small helper kernels ----
// BatchNorm finalisation: batch statistics -> (mean, rstd, scale, shift) and the
// running-stat update of torch's _BatchNorm (momentum; unbiased variance).
__global__ void bn_finalize_kernel(int N, double count, const double *__restrict__ stats,
                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                   float eps, float momentum, float *__restrict__ running_mean,
                                   float *__restrict__ running_var, long long *__restrict__ num_batches_tracked,
                                   float *__restrict__ out /* [4][N] */) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;     // _BatchNorm.forward: num_batches_tracked.add_(1)
  if (c >= N) return;
  const double mean = stats[c] / count;
  double var = stats[N + c] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f;
  const float b = beta ? beta[c] : 0.f;
  const float scale = g * rstd;
  out[c] = (float)mean;
  out[N + c] = rstd;
  out[2 * N + c] = scale;
  out[3 * N + c] = b - (float)mean * scale;
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// BN-backward constants of one layer from the epilogue sums:
//   c1 = gamma*rstd, c2 = -c1*rstd*dgamma/M, c3 = -c1*dbeta/M - c2*mean;
// also emits dgamma / dbeta as fp32 for the optimiser.
__global__ void bn_bwd_consts_kernel(int N, double count, const double *__restrict__ sums /* [2][N]: dbeta, dgamma */,
                                     const float *__restrict__ gamma, const float *__restrict__ fin /* [4][N] */,
                                     int use_batch_stats, float *__restrict__ consts /* [3][N] */,
                                     float *__restrict__ dgamma, float *__restrict__ dbeta,
                                     const float *__restrict__ W /* [N][K] or null */, int K, int k0,
                                     float *__restrict__ Wt /* [K - k0][N] */) {
  // the dgrad GEMM of this layer wants the weight as [K][N] rows (columns k0.. only): transposed here instead of by
  // a separate copy kernel per layer and step
  if (W) {
    const int total = (K - k0) * N;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
      const int k = e / N, n = e - k * N;
      Wt[e] = W[(size_t)n * K + k0 + k];
    }
  }
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  const double db = sums[c], dg = sums[N + c];
  const float mean = fin[c], rstd = fin[N + c];
  const float g = gamma ? gamma[c] : 1.f;
  const float c1 = g * rstd;
  float c2 = 0.f, c3 = 0.f;
  if (use_batch_stats) {
    c2 = (float)(-(double)c1 * (double)rstd * dg / count);
    c3 = (float)(-(double)c1 * db / count - (double)c2 * (double)mean);
  }
  consts[c] = c1;
  consts[N + c] = c2;
  consts[2 * N + c] = c3;
  if (dgamma) dgamma[c] = (float)dg;
  if (dbeta) dbeta[c] = (float)db;
}

// out = relu(y*scale + shift)  (materialised only where a module must return activations: FP)
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(size_t total, int N, const float *__restrict__ y,
                                                           const float *__restrict__ fin, float *__restrict__ out) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % N);
    out[e] = fmaxf(__fmaf_rn(y[e], fin[2 * N + c], fin[3 * N + c]), 0.f);
  }
}

constexpr int kPrepRows = 16;   // minimum rows per block of the two backward-prep kernels

// Rows per block: 16 up to 64k rows (the measured optimum at the headline shapes, see bn_relu_bwd_prep_kernel), then as many
// as keep the grid near 4096 blocks — every block ends with 2 N fp64 atomics onto the SAME 2 N addresses, and with 18 000
// blocks (scene-graph encoders, 295k pooled rows) those serialised atomics, not the rows, were the kernel's time.
__host__ __device__ inline int prep_rows_per_block(long long rows) {
  if (rows <= 65536) return kPrepRows;
  const long long r = (rows + 4095) / 4096;
  return (int)((r + kPrepRows - 1) / kPrepRows * kPrepRows);
}

// g_pre = g_out * [relu(bn(y)) > 0];  sums: dbeta += g_pre, dgamma += g_pre * yhat
__global__ __launch_bounds__(256) void bn_relu_bwd_prep_kernel(long long M, int N, int rpb, const float *__restrict__ y,
                                                              const float *__restrict__ gout,
                                                              const float *__restrict__ fin,
                                                              float *__restrict__ gpre, double *__restrict__ sums) {
  // block = kPrepRows rows x all columns; with N < 256 the 256 threads split into 256/N row groups that are
  // pre-reduced through LDS.  The kernel is latency bound (a thread's rows are a serial chain): measured
  // 64 rows/block 0.28 ms/step, 256 rows 0.79, 16 rows 0.25 — more, smaller blocks win despite 4x the fp64 atomics.
  __shared__ float part[2][256];
  const long long r0 = (long long)blockIdx.x * rpb;
  const int cw = N < 256 ? N : 256;              // columns per pass
  const int groups = 256 / cw;                   // >= 1
  const int grp = threadIdx.x / cw;
  const bool active = grp < groups;
  for (int c0 = 0; c0 < N; c0 += cw) {
    const int c = c0 + threadIdx.x - grp * cw;
    float s1 = 0.f, s2 = 0.f;
    if (active && c < N) {
      const float mean = fin[c], rstd = fin[N + c], sc = fin[2 * N + c], sh = fin[3 * N + c];
#pragma unroll 4
      for (int r = grp; r < rpb; r += groups) {
        const long long row = r0 + r;
        if (row >= M) break;
        const size_t off = (size_t)row * N + c;
        const float yy = y[off];
        const float g = (__fmaf_rn(yy, sc, sh) > 0.f) ? gout[off] : 0.f;
        gpre[off] = g;
        s1 += g;
        s2 = __fmaf_rn(g, (yy - mean) * rstd, s2);
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

// Fused BN + ReLU + max over the ns rows of each group: y (R*ns, C) -> pooled (R, C), arg (R, C)
// and the raw pre-BN value at the arg-max (so the backward reductions need no gather from y).
// VEC = channels per thread (4 -> dwordx4 loads when C % 4 == 0).
template <int VEC>
__global__ __launch_bounds__(256) void bn_relu_rows_max_kernel(size_t total /* R*C/VEC */, int ns, int C,
                                                              const float *__restrict__ y,
                                                              const float *__restrict__ fin,
                                                              float *__restrict__ out, int *__restrict__ arg,
                                                              float *__restrict__ yraw) {
  const int CV = C / VEC;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t r = e / CV;
    const int c = (int)(e - r * CV) * VEC;
    float sc[VEC], sh[VEC], best[VEC], braw[VEC];
    int bi[VEC];
    const float *p = y + r * ns * C + c;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      sc[v] = fin[2 * C + c + v];
      sh[v] = fin[3 * C + c + v];
      braw[v] = p[v];
      best[v] = fmaxf(__fmaf_rn(braw[v], sc[v], sh[v]), 0.f);
      bi[v] = 0;
    }
    for (int s = 1; s < ns; ++s) {
      float raw[VEC];
      if (VEC == 4) {
        const float4 q = *reinterpret_cast<const float4 *>(p + (size_t)s * C);
        raw[0] = q.x; raw[1 % VEC] = q.y; raw[2 % VEC] = q.z; raw[3 % VEC] = q.w;
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) raw[v] = p[(size_t)s * C + v];
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float z = fmaxf(__fmaf_rn(raw[v], sc[v], sh[v]), 0.f);
        if (z > best[v]) { best[v] = z; bi[v] = s; braw[v] = raw[v]; }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      out[r * C + c + v] = best[v];
      arg[r * C + c + v] = bi[v];
      yraw[r * C + c + v] = braw[v];
    }
  }
}

// Pool backward reductions: gPm = gP * [pooled > 0]; dbeta += gPm; dgamma += gPm * yhat[arg-max row]
// (yraw = the pre-BN value at the arg-max, saved by the forward kernel).
__global__ __launch_bounds__(256) void pool_bwd_prep_kernel(long long R, int C, int rpb, const float *__restrict__ yraw,
                                                           const float *__restrict__ pooled,
                                                           const float *__restrict__ gP,
                                                           const float *__restrict__ fin,
                                                           float *__restrict__ gPm, double *__restrict__ sums,
                                                           const long long *__restrict__ seg, int ns) {
  __shared__ float part[2][256];                 // see bn_relu_bwd_prep_kernel
  if (seg) {                                     // blockIdx.y = scan: its pooled rows, its (4,C) finalize block, its sums
    const long long g0 = seg[blockIdx.y] / ns;
    R = seg[blockIdx.y + 1] / ns - g0;
    yraw += (size_t)g0 * C; pooled += (size_t)g0 * C; gP += (size_t)g0 * C; gPm += (size_t)g0 * C;
    fin += (size_t)blockIdx.y * 4 * C;
    sums += (size_t)blockIdx.y * 2 * C;
    rpb = prep_rows_per_block(R);                // the blocks — and with them the fp32 partial sums — of the scan's own call
  }
  const long long r0 = (long long)blockIdx.x * rpb;
  if (r0 >= R) return;
  const int cw = C < 256 ? C : 256;
  const int groups = 256 / cw;
  const int grp = threadIdx.x / cw;
  const bool active = grp < groups;
  for (int c0 = 0; c0 < C; c0 += cw) {
    const int c = c0 + threadIdx.x - grp * cw;
    float s1 = 0.f, s2 = 0.f;
    if (active && c < C) {
      const float mean = fin[c], rstd = fin[C + c];
#pragma unroll 4
      for (int i = grp; i < rpb; i += groups) {
        const long long r = r0 + i;
        if (r >= R) break;
        const size_t off = (size_t)r * C + c;
        const float g = pooled[off] > 0.f ? gP[off] : 0.f;
        gPm[off] = g;
        s1 += g;
        s2 = __fmaf_rn(g, (yraw[off] - mean) * rstd, s2);
      }
    }
    part[0][threadIdx.x] = s1;
    part[1][threadIdx.x] = s2;
    __syncthreads();
    if (grp == 0 && c < C) {
      for (int q = 1; q < groups; ++q) { s1 += part[0][threadIdx.x + q * cw]; s2 += part[1][threadIdx.x + q * cw]; }
      atomicAdd(sums + c, (double)s1);
      atomicAdd(sums + C + c, (double)s2);
    }
    __syncthreads();
  }
}

// ---- vectorised forms of the two prep kernels (round 5) -------------------------------------------------------------
// The kernels above move 4 bytes per lane and end every 16-row block with 2 C fp64 atomics onto the same 2 C addresses:
// at the headline shapes that is 4096 blocks x 256 atomics = 1 M atomics on sixteen cache lines, which — not the 92 MB of
// rows — was their time (58 us = 1.6 TB/s).  Here a thread owns FOUR columns (16-byte accesses) and every fourth row group
// of a block of `rpb` rows, four rows in flight per stream; the grid is capped at kPrepBlocks blocks so that the atomics
// (<= 256 x 2 C, coalesced) disappear behind the rows.  Same arithmetic per element; the per-column sums are fp32 over a thread's
// rows, then fp32 over the block's row groups (LDS), then fp64 across blocks, as before.
constexpr int kPrepBlocks = 256;

// C % 4 == 0 and at most 256 threads per row
__host__ __device__ inline bool prep_vec_ok(int C) { return C % 4 == 0 && C / 4 <= 256; }
// rows per block for R rows of C columns: a multiple of (row groups x 4 rows in flight), at most kPrepBlocks blocks
__host__ __device__ inline long long prep_vec_rpb(long long R, int C) {
  const int groups = 256 / (C / 4);
  const long long unit = (long long)groups * 4;
  const long long want = (R + kPrepBlocks - 1) / kPrepBlocks;
  return (want + unit - 1) / unit * unit;
}

// POOLED: gate = pooled > 0, yv = yraw (pn2_pool_bwd_prep);  else gate = relu(bn(y)) > 0, yv = y (pn2_bn_relu_bwd_prep)
template <bool POOLED>
__global__ __launch_bounds__(256) void prep_vec_kernel(long long R, int C, long long rpb, const float *__restrict__ yv,
                                                       const float *__restrict__ pooled, const float *__restrict__ gin,
                                                       const float *__restrict__ fin, float *__restrict__ gout,
                                                       double *__restrict__ sums, const long long *__restrict__ seg,
                                                       int ns) {
  // per-thread sums: fp32 over at most 16 rows of a column, then folded into fp64 (with <= 256 blocks a thread walks
  // up to 512 rows of a 1M-row call; a plain fp32 running sum over them lost digits the 16-row blocks of round 4 kept)
  __shared__ double dpart[2][256][4];
  if (seg) {                                     // blockIdx.y = scan: its rows, its (4,C) finalize block, its sums
    const long long g0 = seg[blockIdx.y] / ns;
    R = seg[blockIdx.y + 1] / ns - g0;
    const size_t o = (size_t)g0 * C;
    yv += o; gin += o; gout += o;
    if (POOLED) pooled += o;
    fin += (size_t)blockIdx.y * 4 * C;
    sums += (size_t)blockIdx.y * 2 * C;
    rpb = prep_vec_rpb(R, C);                    // the blocks — and with them the fp32 partial sums — of the scan's own call
  }
  const long long r0 = (long long)blockIdx.x * rpb;
  if (r0 >= R) return;                           // (block-uniform)
  const long long r1 = r0 + rpb < R ? r0 + rpb : R;
  const int C4 = C >> 2;
  const int groups = 256 / C4;
  const int grp = threadIdx.x / C4, c4 = threadIdx.x - grp * C4;
  const bool active = grp < groups;
  float4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
  double d1[4] = {0.0, 0.0, 0.0, 0.0}, d2[4] = {0.0, 0.0, 0.0, 0.0};
  int since = 0;
  if (active) {
    const float4 mean = reinterpret_cast<const float4 *>(fin)[c4], rstd = reinterpret_cast<const float4 *>(fin + C)[c4];
    float4 sc = mean, sh = mean;
    if (!POOLED) { sc = reinterpret_cast<const float4 *>(fin + 2 * C)[c4]; sh = reinterpret_cast<const float4 *>(fin + 3 * C)[c4]; }
    const float4 *Y = reinterpret_cast<const float4 *>(yv) + c4;
    const float4 *P = reinterpret_cast<const float4 *>(POOLED ? pooled : yv) + c4;
    const float4 *G = reinterpret_cast<const float4 *>(gin) + c4;
    float4 *O = reinterpret_cast<float4 *>(gout) + c4;
    for (long long r = r0 + grp; r < r1; r += 4 * groups) {
      float4 y[4], p[4], g[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long row = r + (long long)u * groups;
        ok[u] = row < r1;
        const size_t o = (size_t)(ok[u] ? row : r) * C4;
        y[u] = Y[o];
        if (POOLED) p[u] = P[o];
        g[u] = G[o];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) continue;
        float4 o4;
        const float ya[4] = {y[u].x, y[u].y, y[u].z, y[u].w};
        const float pa[4] = {POOLED ? p[u].x : 0.f, POOLED ? p[u].y : 0.f, POOLED ? p[u].z : 0.f, POOLED ? p[u].w : 0.f};
        const float ga[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
        const float sca[4] = {sc.x, sc.y, sc.z, sc.w}, sha[4] = {sh.x, sh.y, sh.z, sh.w};
        const float ma[4] = {mean.x, mean.y, mean.z, mean.w}, ra[4] = {rstd.x, rstd.y, rstd.z, rstd.w};
        float oa[4], a1[4] = {s1.x, s1.y, s1.z, s1.w}, a2[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool gate = POOLED ? pa[j] > 0.f : __fmaf_rn(ya[j], sca[j], sha[j]) > 0.f;
          const float gv = gate ? ga[j] : 0.f;
          oa[j] = gv;
          a1[j] += gv;
          a2[j] = __fmaf_rn(gv, (ya[j] - ma[j]) * ra[j], a2[j]);
        }
        o4 = float4{oa[0], oa[1], oa[2], oa[3]};
        s1 = float4{a1[0], a1[1], a1[2], a1[3]};
        s2 = float4{a2[0], a2[1], a2[2], a2[3]};
        O[(size_t)(r + (long long)u * groups) * C4] = o4;
      }
      if (++since == 4) {                          // 16 rows of fp32, then fp64
        since = 0;
        d1[0] += s1.x; d1[1] += s1.y; d1[2] += s1.z; d1[3] += s1.w;
        d2[0] += s2.x; d2[1] += s2.y; d2[2] += s2.z; d2[3] += s2.w;
        s1 = float4{0.f, 0.f, 0.f, 0.f}; s2 = s1;
      }
    }
  }
  d1[0] += s1.x; d1[1] += s1.y; d1[2] += s1.z; d1[3] += s1.w;
  d2[0] += s2.x; d2[1] += s2.y; d2[2] += s2.z; d2[3] += s2.w;
#pragma unroll
  for (int j = 0; j < 4; ++j) { dpart[0][threadIdx.x][j] = d1[j]; dpart[1][threadIdx.x][j] = d2[j]; }
  __syncthreads();
  if (grp == 0) {
    for (int q = 1; q < groups; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) { d1[j] += dpart[0][threadIdx.x + q * C4][j]; d2[j] += dpart[1][threadIdx.x + q * C4][j]; }
  }
  __syncthreads();
  // the block's 2 C sums leave through LDS so that consecutive lanes add to consecutive doubles: the L2 retires an atomic
  // instruction per 128-byte line it touches (16 doubles) — four columns per lane straight from the registers is a 32-byte
  // lane stride, four times the line visits (measured: 48 instead of 18 us for 512 blocks x 512 sums)
  double *red = &dpart[0][0][0];                                  // [2][C] doubles (C <= 1024: the 16 KB of dpart)
  if (grp == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[4 * c4 + j] = d1[j]; red[C + 4 * c4 + j] = d2[j]; }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * C; t += 256) atomicAdd(sums + t, red[t]);
}

inline bool aligned16(const void *a, const void *b, const void *c, const void *d, const void *e) {
  return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d) | ((uintptr_t)e)) & 15) == 0;
}

inline unsigned capped_grid(size_t work, int block = 256, unsigned cap = 8192) {
  size_t g = (work + block - 1) / block;
  if (g > cap) g = cap;
  return (unsigned)(g ? g : 1);
}

template <int NT, int KC, int CW, int PRO, int EPI>
void launch_one(const GemmArgs &a, hipStream_t s, int grid_override = 0) {
  // persistent: two workgroups per CU (256 CUs), each walking tiles with stride gridDim.x;
  // N wider than the workgroup's NT*CW column tiles is covered by column blocks (grid.y)
  const long long ntiles = (a.M + BM - 1) / BM;
  const unsigned ny = (unsigned)((a.N + NT * CW * 32 - 1) / (NT * CW * 32));
  // 8-wave workgroups: two per CU; 4-wave workgroups: three (their VGPR / LDS budgets admit it and the
  // third hides the barrier and LDS latencies of the other two: 455 -> 418 us on the 64 -> 64 layer of SA1)
  long long gx = (grid_override ? grid_override : (CW == 1 ? 768 : 512)) / ny;
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  dim3 grid((unsigned)gx, ny);
  const size_t prm_bytes = PRO == PRO_NONE ? 0 : (PRO == PRO_FIRST ? 11 : 3) * (size_t)((a.K + KC - 1) / KC * KC) * sizeof(float);
  hipLaunchKernelGGL((mlp_gemm_kernel<NT, KC, CW, PRO, EPI>), grid, dim3(256 * CW), prm_bytes, s, a);
}

// tile configuration by number of 32-column tiles
template <int PRO, int EPI>
void launch_by_width(const GemmArgs &a, int tiles, hipStream_t s) {
  if constexpr (EPI == EPI_MASK) {
    // the Yprev prefetch costs 16 registers per column tile and the GY/POOLG prologues a second
    // register ring: one tile per wave up to 64 columns, two beyond (column blocks past 128)
    if (tiles <= 1) launch_one<1, 32, 1, PRO, EPI>(a, s);
    else if (tiles <= 2) launch_one<1, 32, 2, PRO, EPI>(a, s);
    else launch_one<2, 32, 2, PRO, EPI>(a, s);
  } else {
    if (tiles <= 1) launch_one<1, 32, 1, PRO, EPI>(a, s);
    else if (tiles <= 2) launch_one<2, 32, 1, PRO, EPI>(a, s);
    else if (tiles <= 4) {
      // 16-wide K chunks here: 71 instead of 87 VGPRs and 36 KB of LDS, so
      // THREE of these 8-wave workgroups fit a CU (grid 768), and two still fit next to a resident FPS workgroup of the
      // geometry prefetch: 18.22 -> 18.08 ms/step (6-run means; the 32-wide variant ran with a grid of 512)
      launch_one<2, 16, 2, PRO, EPI>(a, s, 768);
    }
    else if (tiles <= 6) launch_one<3, 16, 2, PRO, EPI>(a, s);
    // 7-8 tiles (N = 256): TWO column blocks of the 128-column variant above instead of one 256-column workgroup (4 tiles
    // per wave, 87+ VGPRs, grid 512) — A is read twice (the second time from L2), but three workgroups per CU stay
    // resident: 17.37 -> 17.14 ms/step, the 256-wide layers of SA3 / SA4 / FP 0.317 -> 0.206, 0.140 -> 0.111, 0.119 -> 0.087 ms
    else if (tiles <= 8) launch_one<2, 16, 2, PRO, EPI>(a, s, 768);
    else if (tiles > 10) launch_one<4, 16, 2, PRO, EPI>(a, s);   // > 10: column blocks of 256
    else launch_one<5, 16, 2, PRO, EPI>(a, s);
  }
}

}  // namespace

// -------------------------------------------------------------------- C ABI ----
extern "C" int pn2_mlp_gemm(long long M, int K, int N, int pro, int epi, const float *X,
                            const float *X2, const float *p0, const float *p1, const float *p2,
                            const int *arg, const float *gP, int ns, const float *W, float *Y,
                            double *stats, const float *Yprev, const float *e_fin, void *stream) {
  if (M < 0 || K <= 0 || N <= 0 || pro < 0 || pro > 3 || epi < 0 || epi > 2) return PN2_EINVAL;
  if (K > 2048) return PN2_EINVAL;               // the prologue parameter table lives in LDS (3 x K floats)
  if (M == 0) return PN2_OK;
  if (!W || !Y) return PN2_ENULL;
  if ((pro == PRO_NONE || pro == PRO_BNRELU || pro == PRO_GY) && !X) return PN2_ENULL;
  if (pro != PRO_NONE && (!p0 || !p1)) return PN2_ENULL;
  if (pro >= PRO_GY && (!X2 || !p2)) return PN2_ENULL;
  if (pro == PRO_GY && !X) return PN2_ENULL;
  if (pro == PRO_POOLG && (!arg || !gP || ns < 16 || M >= 0x7fffffffLL)) return PN2_EINVAL;   // patch sizing assumes ns >= 16
  if (epi != EPI_NONE && !stats) return PN2_ENULL;
  if (epi == EPI_MASK && (!Yprev || !e_fin)) return PN2_ENULL;
  if ((M + BM - 1) / BM > 0x7fffffffLL) return PN2_EINVAL;
  GemmArgs a;
  a.X = X; a.X2 = X2; a.p0 = p0; a.p1 = p1; a.p2 = p2; a.arg = arg; a.gP = gP; a.W = W; a.Y = Y;
  a.stats = stats; a.Yprev = Yprev;
  a.e_mean = e_fin; a.e_rstd = e_fin ? e_fin + N : nullptr;
  a.e_scale = e_fin ? e_fin + 2 * (size_t)N : nullptr;
  a.e_shift = e_fin ? e_fin + 3 * (size_t)N : nullptr;
  a.M = M; a.K = K; a.N = N; a.ns = ns; a.pro = pro; a.epi = epi;
  hipStream_t s = (hipStream_t)stream;
  const int tiles = (N + 31) / 32;
  // KC = 32 while two workgroups still fit a CU's LDS, else 16
  // instantiated combinations: forward = {NONE, BNRELU} x STATS (eval mode passes epi 0 and is
  // served by the same kernels with the reductions compiled out via EPI_NONE on NONE/BNRELU);
  // backward = {GY, POOLG} x {MASK, NONE}
  if (pro == PRO_NONE && epi == EPI_STATS) launch_by_width<PRO_NONE, EPI_STATS>(a, tiles, s);
  else if (pro == PRO_BNRELU && epi == EPI_STATS) launch_by_width<PRO_BNRELU, EPI_STATS>(a, tiles, s);
  else if (pro == PRO_NONE && epi == EPI_NONE) launch_by_width<PRO_NONE, EPI_NONE>(a, tiles, s);
  else if (pro == PRO_BNRELU && epi == EPI_NONE) launch_by_width<PRO_BNRELU, EPI_NONE>(a, tiles, s);
  else if (pro == PRO_GY && epi == EPI_MASK) launch_by_width<PRO_GY, EPI_MASK>(a, tiles, s);
  else if (pro == PRO_POOLG && epi == EPI_MASK) launch_by_width<PRO_POOLG, EPI_MASK>(a, tiles, s);
  else if (pro == PRO_GY && epi == EPI_NONE) launch_by_width<PRO_GY, EPI_NONE>(a, tiles, s);
  else if (pro == PRO_POOLG && epi == EPI_NONE) launch_by_width<PRO_POOLG, EPI_NONE>(a, tiles, s);
  else return PN2_EINVAL;
  return pn2_check_launch();
}

// ---- the layer ABOVE a lifted first layer, with that layer's output re-formed on the fly (PRO_LIFT / EPI_MASKL) ----
// K = width of the lifted layer (columns of Pq / Q), N = width of the layer above.  ns: rows per centre, a power of two in
// [16, 128] (a 128-row tile holds whole groups, a 16-row piece lies in one).
extern "C" int pn2_mlp_lift_supported(int K, int N, int ns) {
  return K >= 64 && K <= 2048 && N >= 1 && N <= 256 && ns >= 16 && ns <= 128 && (ns & (ns - 1)) == 0;
}

static int lift_args_ok(long long M, int K, int N, long long lrows, int ns) {
  if (M < 0 || lrows <= 0 || !pn2_mlp_lift_supported(K, N, ns)) return 0;
  if (M % ns != 0 || M >= 0x7fffffffLL || lrows * (long long)(K > N ? K : N) * 4 >= 0x40000000LL) return 0;
  return 1;
}

// forward: Y[M][N] = relu(bn_0(Pq[gidx] - Q[row / ns])) W^T, column sums of Y and Y^2 into stats
extern "C" int pn2_mlp_gemm_lift(long long M, int K, int N, long long lrows, const float *Pq, const int *gidx, const float *Q,
                                 int ns, const float *fin0 /* [4][K] */, const float *W, float *Y, double *stats,
                                 void *stream) {
  if (!lift_args_ok(M, K, N, lrows, ns)) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!Pq || !gidx || !Q || !fin0 || !W || !Y || !stats) return PN2_ENULL;
  GemmArgs a = {};
  a.X = Pq; a.lidx = gidx; a.lQ = Q; a.lrows = lrows; a.p0 = fin0 + 2 * (size_t)K; a.p1 = fin0 + 3 * (size_t)K;
  a.W = W; a.Y = Y; a.stats = stats; a.M = M; a.K = K; a.N = N; a.ns = ns; a.pro = PRO_LIFT; a.epi = EPI_STATS;
  launch_one<2, 16, 2, PRO_LIFT, EPI_STATS>(a, (hipStream_t)stream, 512);   // (97 VGPRs: two workgroups per CU)
  return pn2_check_launch();
}

// input gradient of that layer: Gout[M][K] = (c1 G + c2 Yl + c3) Wt^T masked by [bn_0(y0) > 0], with the column sums of
// Gout and Gout yhat_0 (BatchNorm backward of the lifted layer), y0 = Pq[gidx] - Q[row / ns]
extern "C" int pn2_mlp_dgrad_lift(long long M, int K, int N, long long lrows, const float *G, const float *Yl,
                                  const float *consts /* [3][N] */, const float *Wt /* [K][N] */, float *Gout, double *sums,
                                  const float *Pq, const int *gidx, const float *Q, int ns, const float *e_fin /* [4][K] */,
                                  void *stream) {
  if (!lift_args_ok(M, K, N, lrows, ns) || N < 64) return PN2_EINVAL;       // (two 32-wide reduction chunks or more)
  if (M == 0) return PN2_OK;
  if (!G || !Yl || !consts || !Wt || !Gout || !sums || !Pq || !gidx || !Q || !e_fin) return PN2_ENULL;
  GemmArgs a = {};
  // the GEMM's reduction runs over the layer's N output channels, its output has the lifted layer's K columns
  a.X = G; a.X2 = Yl; a.p0 = consts; a.p1 = consts + N; a.p2 = consts + 2 * (size_t)N; a.W = Wt; a.Y = Gout; a.stats = sums;
  a.Yprev = Pq; a.lidx = gidx; a.lQ = Q; a.lrows = lrows;
  a.e_mean = e_fin; a.e_rstd = e_fin + K; a.e_scale = e_fin + 2 * (size_t)K; a.e_shift = e_fin + 3 * (size_t)K;
  a.M = M; a.K = N; a.N = K; a.ns = ns; a.pro = PRO_GY; a.epi = EPI_MASKL;
  const int tiles = (K + 31) / 32;
  if (tiles <= 2) launch_one<1, 32, 2, PRO_GY, EPI_MASKL>(a, (hipStream_t)stream);
  else launch_one<2, 32, 2, PRO_GY, EPI_MASKL>(a, (hipStream_t)stream);
  return pn2_check_launch();
}

namespace {
// Batch statistics of the first layer from the Gram matrix of its input: y_0 = X0 W0^T is linear in X0, so
//   sum_r y_0[r][n] = W0[n] . (1^T X0),   sum_r y_0[r][n]^2 = W0[n] (X0^T X0) W0[n]^T   (fp64; gram as rows_gram_kernel
// writes it: [K0 * K0] upper triangle mirrored + [K0] column sums)
__global__ void first_layer_stats_kernel(int N, int K0, const float *__restrict__ W0, const double *__restrict__ gram,
                                         double *__restrict__ stats) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < K0; ++i) {
    const double wi = (double)W0[n * K0 + i];
    s1 += wi * gram[K0 * K0 + i];
    double t = 0.0;
    for (int j = 0; j < K0; ++j) t += gram[i * K0 + j] * (double)W0[n * K0 + j];
    s2 += wi * t;
  }
  stats[n] = s1;
  stats[N + n] = s2 > 0.0 ? s2 : 0.0;
}
}  // namespace

extern "C" int pn2_first_layer_stats(int N, int K0, const float *W0, const double *gram, double *stats, void *stream) {
  if (N <= 0 || K0 < 1 || K0 > 8) return PN2_EINVAL;
  if (!W0 || !gram || !stats) return PN2_ENULL;
  hipLaunchKernelGGL(first_layer_stats_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, (hipStream_t)stream, N, K0, W0,
                     gram, stats);
  return pn2_check_launch();
}

// Second layer of a stack whose first layer is recomputed on the fly (PRO_FIRST): Y = relu(bn_0(X0 W0^T)) W^T.
extern "C" int pn2_mlp_gemm_first_supported(int K0, int K, int N) {
  return K0 >= 1 && K0 <= 8 && K >= 1 && K <= 128 && N >= 1 && N <= 128;
}

extern "C" int pn2_mlp_gemm_first(long long M, int K0, int K, int N, int epi, const float *X0, const float *W0,
                                  const float *scale0, const float *shift0, const float *W, float *Y, double *stats,
                                  void *stream) {
  if (M < 0 || !pn2_mlp_gemm_first_supported(K0, K, N) || (epi != EPI_NONE && epi != EPI_STATS)) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!X0 || !W0 || !scale0 || !shift0 || !W || !Y) return PN2_ENULL;
  if (epi == EPI_STATS && !stats) return PN2_ENULL;
  if ((M + BM - 1) / BM > 0x7fffffffLL) return PN2_EINVAL;
  GemmArgs a = {};
  a.X = X0; a.W0 = W0; a.K0 = K0; a.p0 = scale0; a.p1 = shift0; a.W = W; a.Y = Y; a.stats = stats;
  a.M = M; a.K = K; a.N = N; a.pro = PRO_FIRST; a.epi = epi;
  hipStream_t s = (hipStream_t)stream;
  const int tiles = (N + 31) / 32;
  if (epi == EPI_STATS) {
    // 64-wide outputs: the 4-wave, 16-chunk variant (94 VGPRs, 26 KB of LDS: four workgroups per CU) — 0.461 ms at the SA1
    // shape against 0.477 for <2, 32, 1> (169 VGPRs), 0.463 for <1, 32, 2>, 0.494 for <1, 16, 2> (tools/fold_bench.py)
    if (tiles <= 1) launch_one<1, 32, 1, PRO_FIRST, EPI_STATS>(a, s);
    else if (tiles <= 2) launch_one<2, 16, 1, PRO_FIRST, EPI_STATS>(a, s, 1024);
    else launch_one<2, 16, 2, PRO_FIRST, EPI_STATS>(a, s, 768);
  } else {
    if (tiles <= 1) launch_one<1, 32, 1, PRO_FIRST, EPI_NONE>(a, s);
    else if (tiles <= 2) launch_one<2, 16, 1, PRO_FIRST, EPI_NONE>(a, s, 1024);
    else launch_one<2, 16, 2, PRO_FIRST, EPI_NONE>(a, s, 768);
  }
  return pn2_check_launch();
}

namespace {
// W' = diag(sgn) W, sgn = -1 where gamma < 0 (rows of the pooled layer's weight), and the sign vector itself
__global__ __launch_bounds__(256) void pool_flip_rows_kernel(int N, int K, const float *__restrict__ W,
                                                            const float *__restrict__ gamma, float *__restrict__ Wf,
                                                            float *__restrict__ sgn) {
  const int total = N * K;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int n = e / K;
    const bool neg = gamma && gamma[n] < 0.f;
    Wf[e] = neg ? -W[e] : W[e];
    if (e - n * K == 0) sgn[n] = neg ? -1.f : 1.f;
  }
}

// pooled = relu(bn(max)) from the sign-adjusted raw partial maxima (ns/psz partial groups of psz = min(ns, 32) rows per
// group, first maximum wins): |scale| * pmax + shift == scale * y + shift bit for bit
__global__ __launch_bounds__(256) void pool_finalize_kernel(size_t total, int C, int nsub, int psz,
                                                           const float *__restrict__ pmax, const int *__restrict__ parg,
                                                           const float *__restrict__ fin, const float *__restrict__ sgn,
                                                           float *__restrict__ out, int *__restrict__ arg,
                                                           float *__restrict__ yraw) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t g = e / C;
    const int c = (int)(e - g * C);
    const size_t p0 = g * nsub * C + c;
    float best = pmax[p0];
    int row = parg[p0];
    for (int j = 1; j < nsub; ++j) {
      const float v = pmax[p0 + (size_t)j * C];
      if (v > best) { best = v; row = j * psz + parg[p0 + (size_t)j * C]; }
    }
    const float raw = best * sgn[c];
    yraw[e] = raw;
    arg[e] = row;
    out[e] = fmaxf(__fmaf_rn(raw, fin[2 * C + c], fin[3 * C + c]), 0.f);
  }
}
}  // namespace

// Last layer of a max-pooled stack without materialising its output: column sums for the batch statistics and, per
// group of `ns` rows, the column maxima of pro(X) * Wf^T with their rows (see EPI_POOL above).  Replaces
// Conv2d + BatchNorm2d + ReLU + F.max_pool2d of OPS/pointnet2_modules.py:58-70 together with pn2_pool_finalize.
extern "C" int pn2_mlp_gemm_pool(long long M, int K, int N, int pro, const float *X, const float *p0, const float *p1,
                                 const float *Wf, const float *sgn, int ns, double *stats, float *pmax, int *parg,
                                 void *stream) {
  if (M < 0 || K <= 0 || N <= 0 || K > 2048) return PN2_EINVAL;
  if (pro != PRO_NONE && pro != PRO_BNRELU) return PN2_EINVAL;
  if (ns != 16 && ns != 32 && ns != 64 && ns != 128) return PN2_EINVAL;
  if (M % ns) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!X || !Wf || !sgn || !stats || !pmax || !parg) return PN2_ENULL;
  if (pro == PRO_BNRELU && (!p0 || !p1)) return PN2_ENULL;
  if ((M + BM - 1) / BM > 0x7fffffffLL) return PN2_EINVAL;
  GemmArgs a = {};
  a.X = X; a.p0 = p0; a.p1 = p1; a.W = Wf; a.stats = stats; a.pmax = pmax; a.parg = parg; a.sgn = sgn;
  a.M = M; a.K = K; a.N = N; a.ns = ns; a.pro = pro; a.epi = EPI_POOL;
  const int tiles = (N + 31) / 32;
  if (pro == PRO_NONE) launch_by_width<PRO_NONE, EPI_POOL>(a, tiles, (hipStream_t)stream);
  else launch_by_width<PRO_BNRELU, EPI_POOL>(a, tiles, (hipStream_t)stream);
  return pn2_check_launch();
}

extern "C" int pn2_pool_flip_rows(int N, int K, const float *W, const float *gamma, float *Wf, float *sgn,
                                  void *stream) {
  if (N <= 0 || K <= 0) return PN2_EINVAL;
  if (!W || !Wf || !sgn) return PN2_ENULL;
  hipLaunchKernelGGL(pool_flip_rows_kernel, dim3(capped_grid((size_t)N * K, 256, 256)), dim3(256), 0,
                     (hipStream_t)stream, N, K, W, gamma, Wf, sgn);
  return pn2_check_launch();
}

extern "C" int pn2_pool_finalize(long long R, int C, int ns, const float *pmax, const int *parg, const float *fin,
                                 const float *sgn, float *out, int *arg, float *yraw, void *stream) {
  if (R < 0 || C <= 0) return PN2_EINVAL;
  if (ns != 16 && ns != 32 && ns != 64 && ns != 128) return PN2_EINVAL;
  if (R == 0) return PN2_OK;
  if (!pmax || !parg || !fin || !sgn || !out || !arg || !yraw) return PN2_ENULL;
  const size_t total = (size_t)R * C;
  const int psz = ns < 32 ? ns : 32;
  hipLaunchKernelGGL(pool_finalize_kernel, dim3(capped_grid(total)), dim3(256), 0, (hipStream_t)stream, total, C,
                     ns / psz, psz, pmax, parg, fin, sgn, out, arg, yraw);
  return pn2_check_launch();
}

static int wgrad_impl(long long M, int N, int K, int gmode, int amode, const float *G,
                      const float *Yl, const float *consts /* [3][N] */, const int *arg,
                      const float *gP, int ns, const float *X, const float *a_fin /* [4][K] or NULL */,
                      float *dW, const int *lidx, const float *lQ, long long lrows, void *stream) {
  if (M < 0 || N <= 0 || K <= 0 || N > WMAXN) return PN2_EINVAL;
  if (gmode != PRO_GY && gmode != PRO_POOLG) return PN2_EINVAL;
  if (amode != PRO_NONE && amode != PRO_BNRELU && amode != PRO_LIFT) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!Yl || !consts || !X || !dW) return PN2_ENULL;
  if (gmode == PRO_GY && !G) return PN2_ENULL;
  if (gmode == PRO_POOLG && (!arg || !gP || ns < 16 || M >= 0x7fffffffLL)) return PN2_EINVAL;
  if (amode != PRO_NONE && !a_fin) return PN2_ENULL;
  if (amode == PRO_LIFT && (gmode != PRO_GY || !lidx || !lQ || N > 128 || !lift_args_ok(M, K, N, lrows, ns))) return PN2_EINVAL;
  WgradArgs a;
  a.lidx = lidx; a.lQ = lQ; a.lrows = lrows;
  a.G = G; a.Yl = Yl; a.c1 = consts; a.c2 = consts + N; a.c3 = consts + 2 * (size_t)N;
  a.arg = arg; a.gP = gP; a.X = X;
  a.a_scale = a_fin ? a_fin + 2 * (size_t)K : nullptr;
  a.a_shift = a_fin ? a_fin + 3 * (size_t)K : nullptr;
  a.dW = dW; a.M = M; a.N = N; a.K = K; a.ns = ns; a.gmode = gmode; a.amode = amode;
  // persistent-style: ~2 workgroups per CU, each a contiguous slab of rows (multiple of WR)
  // two k-tiles only pay when there are enough n-tiles to keep four wave groups busy (measured:
  // N=128,K=64 1.48 -> 1.14 ms; N=64,K=64 0.86 -> 1.06 ms because the gy tile would be staged 128 wide)
  // N > 256 (9-10 n-tiles): four wave groups x 3 tiles (139-160 VGPRs) instead of two x 5 (197).  The 5-tile variant
  // cannot share a CU with a workgroup of the cooperative FPS kernel that the geometry prefetch keeps resident on every
  // CU — as the first kernel of the backward it waited 1.6 ms for the FPS to end (kernel-trace timeline) — and capping
  // its registers spills (0.13 -> 1.6 ms).  The price is 64-column K blocks, i.e. more passes over a SMALL gy.
  const int kt = (amode != PRO_LIFT && ((K <= 64 && N > 64) || N > 256)) ? 2 : 4;
  // K = 32j + (1..3) raw-input columns (relative xyz in front of the features): reduce the leading columns on the
  // VALU side and give the MFMA part the aligned rest, instead of a whole extra pass over g and y for 3 columns
  // (measured: K = 131, M = 1M 0.79 -> 0.55 ms; K = 259, M = 256k 0.31 -> 0.28; below that the extra pass is cheaper)
  const int koff = (amode == PRO_NONE && kt == 4 && K > 32 && K % 32 >= 1 && K % 32 <= 3 && M >= (1 << 18)) ? K % 32 : 0;
  a.koff = koff;
  const unsigned kblocks = (unsigned)((K - koff + 32 * kt - 1) / (32 * kt));
  long long wgs = 512 / kblocks;
  if (wgs < 1) wgs = 1;
  // few rows (FP / SA4 layers, M <= 128k): every workgroup ends with an N x 128 block of float atomics onto the same
  // addresses — at 128 rows per workgroup the flush, not the rows, is the kernel's time.  At least 512 rows per workgroup,
  // but not fewer than 64 row slabs (M = 16k, N 256, K 512: 107 -> 70 us; 32k: 147 -> 111; 131k x 256 x 128: 144 -> 129)
  {
    long long cap = M / 512;
    if (cap < 64) cap = 64;
    if (wgs > cap) wgs = cap;
  }
  long long rows = (M + wgs - 1) / wgs;
  rows = ((rows + 63) / 64) * 64;     // multiple of every variant's tile height
  a.rows_per_wg = rows;
  const unsigned gx = (unsigned)((M + rows - 1) / rows);
  hipStream_t s = (hipStream_t)stream;
  const int ntiles = (N + 31) / 32;
  dim3 grid(gx, kblocks);
#define PN2_WGRAD(NTW, KT)                                                                               \
  do {                                                                                                 \
    if (gmode == PRO_GY && amode == PRO_NONE)                                                          \
      hipLaunchKernelGGL((mlp_wgrad_kernel<NTW, PRO_GY, PRO_NONE, KT>), grid, dim3(512), 0, s, a);     \
    else if (gmode == PRO_GY)                                                                          \
      hipLaunchKernelGGL((mlp_wgrad_kernel<NTW, PRO_GY, PRO_BNRELU, KT>), grid, dim3(512), 0, s, a);   \
    else if (amode == PRO_NONE)                                                                        \
      hipLaunchKernelGGL((mlp_wgrad_kernel<NTW, PRO_POOLG, PRO_NONE, KT>), grid, dim3(512), 0, s, a);  \
    else                                                                                               \
      hipLaunchKernelGGL((mlp_wgrad_kernel<NTW, PRO_POOLG, PRO_BNRELU, KT>), grid, dim3(512), 0, s, a);\
  } while (0)
#define PN2_WGRAD_LEAD(NTW)                                                                              \
  do {                                                                                                 \
    if (gmode == PRO_GY)                                                                               \
      hipLaunchKernelGGL((mlp_wgrad_kernel<NTW, PRO_GY, PRO_NONE, 4, true>), grid, dim3(512), 0, s, a);    \
    else                                                                                               \
      hipLaunchKernelGGL((mlp_wgrad_kernel<NTW, PRO_POOLG, PRO_NONE, 4, true>), grid, dim3(512), 0, s, a); \
  } while (0)
  if (amode == PRO_LIFT) {
    hipLaunchKernelGGL((mlp_wgrad_kernel<2, PRO_GY, PRO_LIFT, 4>), grid, dim3(512), 0, s, a);
  } else if (koff) {
    if (ntiles <= 2) PN2_WGRAD_LEAD(1);
    else if (ntiles <= 4) PN2_WGRAD_LEAD(2);
    else if (ntiles <= 8) PN2_WGRAD_LEAD(4);
    else PN2_WGRAD_LEAD(5);
  } else if (kt == 2) {          // four n-groups of waves
    if (ntiles <= 4) PN2_WGRAD(1, 2);
    else if (ntiles <= 8) PN2_WGRAD(2, 2);
    else PN2_WGRAD(3, 2);
  } else {                // two n-groups of waves
    if (ntiles <= 2) PN2_WGRAD(1, 4);
    else if (ntiles <= 4) PN2_WGRAD(2, 4);
    else if (ntiles <= 8) PN2_WGRAD(4, 4);
    else PN2_WGRAD(5, 4);
  }
#undef PN2_WGRAD
#undef PN2_WGRAD_LEAD
  return pn2_check_launch();
}

extern "C" int pn2_mlp_wgrad(long long M, int N, int K, int gmode, int amode, const float *G,
                             const float *Yl, const float *consts /* [3][N] */, const int *arg,
                             const float *gP, int ns, const float *X, const float *a_fin /* [4][K] or NULL */,
                             float *dW, void *stream) {
  if (amode == PRO_LIFT) return PN2_EINVAL;
  return wgrad_impl(M, N, K, gmode, amode, G, Yl, consts, arg, gP, ns, X, a_fin, dW, nullptr, nullptr, 0, stream);
}

// weight gradient of the layer above a lifted first layer: dW[N][K] += (c1 G + c2 Yl + c3)^T relu(bn_0(Pq[gidx] - Q[row / ns]))
extern "C" int pn2_mlp_wgrad_lift(long long M, int N, int K, long long lrows, const float *G, const float *Yl,
                                  const float *consts /* [3][N] */, const float *Pq, const int *gidx, const float *Q, int ns,
                                  const float *a_fin /* [4][K] */, float *dW, void *stream) {
  return wgrad_impl(M, N, K, PRO_GY, PRO_LIFT, G, Yl, consts, nullptr, nullptr, ns, Pq, a_fin, dW, gidx, Q, lrows, stream);
}

extern "C" int pn2_bn_finalize(int N, double count, const double *stats, const float *gamma,
                               const float *beta, float eps, float momentum, float *running_mean,
                               float *running_var, long long *num_batches_tracked, float *fin, void *stream) {
  if (N <= 0 || !(count > 0.0)) return PN2_EINVAL;
  if (!stats || !fin) return PN2_ENULL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, N,
                     count, stats, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, fin);
  return pn2_check_launch();
}

namespace {
// Running statistics after S single-scan training steps (block-diagonal batch with per-scan statistics), in scan order:
// running <- (1 - m) running + m stat_s, s = 0..S-1, closed form with the host-computed weights w[s] = m (1 - m)^(S-1-s),
// wu[s] = w[s] n_s / (n_s - 1) (unbiased variance like torch.nn.functional.batch_norm) and decay = (1 - m)^S.
// fins (S, 4, C): rows 0 / 1 of every scan = batch mean / rstd as bn_finalize_kernel leaves them.
__global__ __launch_bounds__(128) void bn_running_update_kernel(int S, int C, const float *__restrict__ fins, float eps,
                                                               float decay, const float *__restrict__ w,
                                                               const float *__restrict__ wu, float *__restrict__ rm,
                                                               float *__restrict__ rv, long long *__restrict__ nbt) {
  const int c = blockIdx.x * 128 + threadIdx.x;
  if (c < C) {
    float m = 0.f, v = 0.f;
    for (int s = 0; s < S; ++s) {
      const float *f = fins + (size_t)s * 4 * C;
      const float r = f[C + c];
      m = fmaf(w[s], f[c], m);
      v = fmaf(wu[s], fmaxf(1.0f / (r * r) - eps, 0.f), v);
    }
    rm[c] = fmaf(decay, rm[c], m);
    rv[c] = fmaf(decay, rv[c], v);
  }
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += S;
}
}  // namespace

extern "C" int pn2_bn_running_update(int S, int C, const float *fins, float eps, float decay, const float *w,
                                     const float *wu, float *running_mean, float *running_var,
                                     long long *num_batches_tracked, void *stream) {
  if (S <= 0 || C <= 0) return PN2_EINVAL;
  if (!fins || !w || !wu || !running_mean || !running_var) return PN2_ENULL;
  hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, S, C, fins, eps,
                     decay, w, wu, running_mean, running_var, num_batches_tracked);
  return pn2_check_launch();
}

extern "C" int pn2_bn_bwd_consts(int N, double count, const double *sums, const float *gamma,
                                 const float *fin, int use_batch_stats, float *consts, float *dgamma,
                                 float *dbeta, const float *W, int K, int k0, float *Wt, void *stream) {
  if (N <= 0 || !(count > 0.0)) return PN2_EINVAL;
  if (!sums || !fin || !consts) return PN2_ENULL;
  if (W && (!Wt || K <= 0 || k0 < 0 || k0 >= K)) return PN2_EINVAL;
  // with a weight to transpose: enough 256-thread blocks for ~16 elements per thread (N*K <= 320*2048)
  unsigned blocks = (unsigned)((N + 127) / 128), threads = 128;
  if (W) {
    threads = 256;
    blocks = (unsigned)(((size_t)N * (K - k0) + 4095) / 4096);
    if (blocks < (unsigned)((N + 255) / 256)) blocks = (unsigned)((N + 255) / 256);
  }
  hipLaunchKernelGGL(bn_bwd_consts_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, N,
                     count, sums, gamma, fin, use_batch_stats, consts, dgamma, dbeta, W, K, k0, Wt);
  return pn2_check_launch();
}

extern "C" int pn2_bn_relu_apply(long long M, int N, const float *y, const float *fin, float *out,
                                 void *stream) {
  if (M < 0 || N <= 0) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!y || !fin || !out) return PN2_ENULL;
  const size_t total = (size_t)M * N;
  hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(capped_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     total, N, y, fin, out);
  return pn2_check_launch();
}

extern "C" int pn2_bn_relu_bwd_prep(long long M, int N, const float *y, const float *gout,
                                    const float *fin, float *gpre, double *sums, void *stream) {
  if (M < 0 || N <= 0) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!y || !gout || !fin || !gpre || !sums) return PN2_ENULL;
  if (prep_vec_ok(N) && aligned16(y, gout, fin, gpre, nullptr)) {
    const long long rpbv = prep_vec_rpb(M, N);
    hipLaunchKernelGGL(prep_vec_kernel<false>, dim3((unsigned)((M + rpbv - 1) / rpbv)), dim3(256), 0, (hipStream_t)stream, M, N,
                       rpbv, y, (const float *)nullptr, gout, fin, gpre, sums, (const long long *)nullptr, 1);
    return pn2_check_launch();
  }
  const int rpb = prep_rows_per_block(M);
  hipLaunchKernelGGL(bn_relu_bwd_prep_kernel, dim3((unsigned)((M + rpb - 1) / rpb)), dim3(256), 0,
                     (hipStream_t)stream, M, N, rpb, y, gout, fin, gpre, sums);
  return pn2_check_launch();
}

extern "C" int pn2_bn_relu_rows_max(long long R, int ns, int C, const float *y, const float *fin,
                                    float *out, int *arg, float *yraw, void *stream) {
  if (R < 0 || ns <= 0 || C <= 0) return PN2_EINVAL;
  if (R == 0) return PN2_OK;
  if (!y || !fin || !out || !arg || !yraw) return PN2_ENULL;
  if (C % 4 == 0 && ((uintptr_t)y & 15) == 0) {
    const size_t total = (size_t)R * (C / 4);
    hipLaunchKernelGGL(bn_relu_rows_max_kernel<4>, dim3(capped_grid(total, 256, 16384)), dim3(256), 0,
                       (hipStream_t)stream, total, ns, C, y, fin, out, arg, yraw);
  } else {
    const size_t total = (size_t)R * C;
    hipLaunchKernelGGL(bn_relu_rows_max_kernel<1>, dim3(capped_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       total, ns, C, y, fin, out, arg, yraw);
  }
  return pn2_check_launch();
}

extern "C" int pn2_pool_bwd_prep(long long R, int C, const float *yraw, const float *pooled,
                                 const float *gP, const float *fin, float *gPm, double *sums,
                                 void *stream) {
  if (R < 0 || C <= 0) return PN2_EINVAL;
  if (R == 0) return PN2_OK;
  if (!yraw || !pooled || !gP || !fin || !gPm || !sums) return PN2_ENULL;
  if (prep_vec_ok(C) && aligned16(yraw, pooled, gP, fin, gPm)) {
    const long long rpbv = prep_vec_rpb(R, C);
    hipLaunchKernelGGL(prep_vec_kernel<true>, dim3((unsigned)((R + rpbv - 1) / rpbv)), dim3(256), 0, (hipStream_t)stream, R, C,
                       rpbv, yraw, pooled, gP, fin, gPm, sums, (const long long *)nullptr, 1);
    return pn2_check_launch();
  }
  const int rpb = prep_rows_per_block(R);
  hipLaunchKernelGGL(pool_bwd_prep_kernel, dim3((unsigned)((R + rpb - 1) / rpb)), dim3(256), 0,
                     (hipStream_t)stream, R, C, rpb, yraw, pooled, gP, fin, gPm, sums, (const long long *)nullptr, 1);
  return pn2_check_launch();
}

// ------------------------------------------------------------------------------------------------ batched scans
// Per-scan BatchNorm statistics of a block-diagonal batch in ONE launch per kernel (the arithmetic of the reference's
// DataLoader(batch_size=1) steps, SGP/main.py:54-56, at the whole-batch launch count): `seg` = nseg + 1 ROW offsets of the
// scans in the stack's row tensors (device), every per-channel operand an array of per-scan blocks.
extern "C" int pn2_pool_bwd_prep_seg(long long R, int C, const float *yraw, const float *pooled, const float *gP,
                                     const float *fin, float *gPm, double *sums, const long long *seg, int nseg,
                                     long long seg_max, int ns, void *stream) {
  if (R < 0 || C <= 0 || ns <= 0 || nseg < 1 || nseg > 65535 || seg_max < 0) return PN2_EINVAL;
  if (R == 0) return PN2_OK;
  if (!yraw || !pooled || !gP || !fin || !gPm || !sums || !seg) return PN2_ENULL;
  // every scan is cut into the blocks of its own call (rows per block from ITS row count, in the kernel): at most 4096
  // blocks for any count, fewer than rows / 16
  const long long Rs = seg_max / ns;                                  // pooled rows of the longest scan
  if (prep_vec_ok(C) && aligned16(yraw, pooled, gP, fin, gPm)) {
    // (per-scan offsets g0 * C floats keep the 16-byte alignment: C % 4 == 0)
    const long long unit = (long long)(256 / (C / 4)) * 4;
    long long gxv = (Rs + unit - 1) / unit;
    if (gxv > kPrepBlocks) gxv = kPrepBlocks;
    if (gxv < 1) gxv = 1;
    hipLaunchKernelGGL(prep_vec_kernel<true>, dim3((unsigned)gxv, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, R, C,
                       unit, yraw, pooled, gP, fin, gPm, sums, seg, ns);
    return pn2_check_launch();
  }
  long long gx = (Rs + kPrepRows - 1) / kPrepRows;
  if (gx > 4096) gx = 4096;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(pool_bwd_prep_kernel, dim3((unsigned)gx, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, R, C,
                     kPrepRows, yraw, pooled, gP, fin, gPm, sums, seg, ns);
  return pn2_check_launch();
}

namespace {
// pn2_bn_finalize for nseg scans: stats (S,2,N) -> fin (S,4,N); count of scan s = its rows.  A channel's thread walks the
// scans IN ORDER, so the running statistics receive the S momentum updates exactly as S calls of bn_finalize_kernel
// (= S single-scan training steps) apply them; an empty scan leaves them alone.
__global__ __launch_bounds__(128) void bn_finalize_seg_kernel(int S, int N, const long long *__restrict__ seg,
                                                             const double *__restrict__ stats,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             float eps, float momentum, float *__restrict__ running_mean,
                                                             float *__restrict__ running_var,
                                                             long long *__restrict__ num_batches_tracked,
                                                             float *__restrict__ out) {
  const int c = blockIdx.x * 128 + threadIdx.x;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += S;
  if (c >= N) return;
  const float g = gamma ? gamma[c] : 1.f;
  const float b = beta ? beta[c] : 0.f;
  float rm = running_mean ? running_mean[c] : 0.f, rv = running_mean ? running_var[c] : 0.f;
  for (int s = 0; s < S; ++s) {
    const double rows = (double)(seg[s + 1] - seg[s]);
    const double count = rows > 0.0 ? rows : 1.0;
    const double *st = stats + (size_t)s * 2 * N;
    float *o = out + (size_t)s * 4 * N;
    const double mean = st[c] / count;
    double var = st[N + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = g * rstd;
    o[c] = (float)mean;
    o[N + c] = rstd;
    o[2 * N + c] = scale;
    o[3 * N + c] = b - (float)mean * scale;
    if (running_mean && rows > 0.0) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      rm = (1.f - momentum) * rm + momentum * (float)mean;
      rv = (1.f - momentum) * rv + momentum * (float)unbiased;
    }
  }
  if (running_mean) { running_mean[c] = rm; running_var[c] = rv; }
}

// pn2_bn_bwd_consts for nseg scans: sums (S,2,N), fin (S,4,N) -> consts (S,3,N); dgamma / dbeta = the SUM over the scans,
// accumulated in scan order by the channel's thread (deterministic); optional weight transposition as there.
__global__ void bn_bwd_consts_seg_kernel(int S, int N, const long long *__restrict__ seg, const double *__restrict__ sums,
                                         const float *__restrict__ gamma, const float *__restrict__ fin, int use_batch_stats,
                                         float *__restrict__ consts, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                         const float *__restrict__ W, int K, int k0, float *__restrict__ Wt) {
  if (W) {
    const int total = (K - k0) * N;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
      const int k = e / N, n = e - k * N;
      Wt[e] = W[(size_t)n * K + k0 + k];
    }
  }
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  const float g = gamma ? gamma[c] : 1.f;
  float dgs = 0.f, dbs = 0.f;
  for (int s = 0; s < S; ++s) {
    const double rows = (double)(seg[s + 1] - seg[s]);
    const double count = rows > 0.0 ? rows : 1.0;
    const double db = sums[(size_t)s * 2 * N + c], dg = sums[(size_t)s * 2 * N + N + c];
    const float *f = fin + (size_t)s * 4 * N;
    const float mean = f[c], rstd = f[N + c];
    const float c1 = g * rstd;
    float c2 = 0.f, c3 = 0.f;
    if (use_batch_stats) {
      c2 = (float)(-(double)c1 * (double)rstd * dg / count);
      c3 = (float)(-(double)c1 * db / count - (double)c2 * (double)mean);
    }
    float *o = consts + (size_t)s * 3 * N;
    o[c] = c1;
    o[N + c] = c2;
    o[2 * N + c] = c3;
    // per-scan fp32 rounding, then summed in fp32 in scan order: what S single-scan backward passes accumulate into .grad
    dgs = s ? dgs + (float)dg : (float)dg;
    dbs = s ? dbs + (float)db : (float)db;
  }
  if (dgamma) dgamma[c] = dgs;
  if (dbeta) dbeta[c] = dbs;
}
}  // namespace

extern "C" int pn2_bn_finalize_seg(int S, int N, const long long *seg, const double *stats, const float *gamma,
                                   const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                                   long long *num_batches_tracked, float *fin, void *stream) {
  if (S <= 0 || N <= 0) return PN2_EINVAL;
  if (!seg || !stats || !fin) return PN2_ENULL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return PN2_ENULL;
  hipLaunchKernelGGL(bn_finalize_seg_kernel, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, S, N, seg, stats,
                     gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, fin);
  return pn2_check_launch();
}

extern "C" int pn2_bn_bwd_consts_seg(int S, int N, const long long *seg, const double *sums, const float *gamma,
                                     const float *fin, int use_batch_stats, float *consts, float *dgamma, float *dbeta,
                                     const float *W, int K, int k0, float *Wt, void *stream) {
  if (S <= 0 || N <= 0) return PN2_EINVAL;
  if (!seg || !sums || !fin || !consts) return PN2_ENULL;
  if (W && (!Wt || K <= 0 || k0 < 0 || k0 >= K)) return PN2_EINVAL;
  unsigned blocks = (unsigned)((N + 127) / 128), threads = 128;
  if (W) {
    threads = 256;
    blocks = (unsigned)(((size_t)N * (K - k0) + 4095) / 4096);
    if (blocks < (unsigned)((N + 255) / 256)) blocks = (unsigned)((N + 255) / 256);
  }
  hipLaunchKernelGGL(bn_bwd_consts_seg_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, S, N, seg, sums, gamma,
                     fin, use_batch_stats, consts, dgamma, dbeta, W, K, k0, Wt);
  return pn2_check_launch();
}

