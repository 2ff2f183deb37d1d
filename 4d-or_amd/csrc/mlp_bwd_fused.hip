// mlp_bwd_fused.hip — one pass over a layer's backward: dgrad AND wgrad from one read.
//
// For a hidden layer l (input = ReLU(BatchNorm(y_{l-1})), output y_l) the backward needs
//     gy      = dL/dy_l            = c1*g + c2*y_l + c3            [M][N]   (BatchNorm backward, per-column constants)
//     dL/dz_{l-1} = [z_{l-1} > 0] * (gy * W_l)                     [M][K]   (dgrad, + the next BN-backward column sums)
//     dW_l    = gy^T * relu(bn(y_{l-1}))                           [N][K]   (wgrad)
// mlp_gemm_kernel (dgrad) and mlp_wgrad_kernel each stream g, y_l and y_{l-1} from HBM: 2(2MN + MK) floats read
// per layer.  Both products share their operands, so this kernel stages ONE row tile
//     gyT[n][row] (transposed, pitch R+1), act[row][k] = relu(bn(y_{l-1})), W_l[n][k] (resident for the whole kernel)
// in LDS and runs both MFMA products from it: reads 2MN + MK (pooled layer: MN + MK), the memory-bound SA1/SA2
// layers become MFMA-bound.  Both fragment read patterns are conflict-free on the transposed tile:
//     dgrad A[i=row][k=n]:  gyT[(2s + l/32) * (R+1) + row0 + l%32]      (consecutive rows)
//     wgrad A[i=n][k=row]:  gyT[(n0 + l%32) * (R+1) + 2s + l/32]        (stride R+1, odd)
// Supported: 32 < N, K <= 128 with min(N, K) <= 64 (the hidden layers of the SA1 MLP at the bench shapes);
// everything else stays on the two-kernel path.  Same addressing discipline as mlp_gemm.hip (buffer descriptors,
// kernel-constant lane offsets, out-of-range instead of masks), same arithmetic (explicit FMAs).
#include "pn2_common.h"
#include "mlp_common.h"

#include "../../include/pn2_hip.h"

#include <stdlib.h>

namespace {

struct BwdArgs {
  const float *G;      // PRO_GY: g = dL/dz_l [M][N]
  const float *Yl;     // y_l [M][N]
  const float *c1, *c2, *c3;
  const int *arg;      // PRO_POOLG: [M/ns][N]
  const float *gP;     // PRO_POOLG: [M/ns][N]
  const float *W;      // [N][K]
  const float *Yprev;  // y_{l-1} [M][K]
  const float *a_mean, *a_rstd, *a_scale, *a_shift;   // layer l-1 BatchNorm, per column k
  float *Gout;         // dL/dz_{l-1} [M][K]
  double *sums;        // [2][K]: sum g', sum g' * yhat_{l-1}
  float *dW;           // [N][K], accumulated with atomics (caller zero-fills)
  long long M;
  int N, K, ns;
  // first-layer fold (version 2 only): X = input rows of layer l-1 [M][K0], P1 [K][K0] += (dL/dz_{l-1})^T X
  const float *X;
  float *P1;
  int K0;
};

template <int GMODE, int NTN, int KTN, int R>
__global__ __launch_bounds__(512) void mlp_bwd_fused_kernel(const BwdArgs a) {
  constexpr bool POOL = GMODE == PRO_POOLG;
  constexpr int NP = NTN * 32, KP = KTN * 32;
  constexpr int LDT = R + 1;
  constexpr int GROWS = 512 / NP;               // gy rows per pass of the workgroup
  constexpr int GPT = R / GROWS;                // gy elements per thread per tile
  constexpr int AROWS = 512 / KP;
  constexpr int APT = R / AROWS;
  constexpr int RB = R / 32;
  constexpr int DTT = RB * KTN;                 // dgrad output tiles per row tile
  static_assert(DTT % 8 == 0, "every wave owns the same number of dgrad tiles");
  constexpr int DT = DTT / 8;
  constexpr int T = NTN * KTN;                  // dW tiles
  constexpr int TW = T >= 8 ? T / 8 : 1;        // dW tiles per wave
  constexpr int RS = T >= 8 ? 1 : 8 / T;        // row split of a dW tile between waves
  constexpr int WROWS = R / RS;                 // rows a wave reduces per tile
  constexpr int PG = POOL ? ((R / 16 + 1 + GROWS - 1) / GROWS) : 1;   // patch entries per thread (ns >= 16)
  constexpr int RGN = POOL ? 1 : GPT;

  extern __shared__ float lds[];
  float *gyT = lds;                             // [NP][LDT]
  float *act = gyT + NP * LDT;                  // [R][KP]
  float *Wl = act + R * KP;                     // [NP][KP]
  float *red = Wl + NP * KP;                    // [2][KP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int N = a.N, K = a.K;
  const long long M = a.M;
  const long long ntiles = (M + R - 1) / R;

  // ---- resident weights ----
  // (eight loads in flight per thread: as a rolled loop the 32 dependent-latency iterations were ~30 us of every launch)
#pragma unroll 8
  for (int i = tid; i < NP * KP; i += 512) {
    const int n = i / KP, k = i % KP;
    Wl[i] = (n < N && k < K) ? a.W[(size_t)n * K + k] : 0.f;
  }

  // ---- per-thread staging coordinates (fixed for the whole kernel) ----
  const int gn = tid % NP, gr0 = tid / NP;
  const int gnc = gn < N ? gn : (N - 1);
  const float c1 = a.c1[gnc], c2 = a.c2[gnc], c3 = a.c3[gnc];
  const int ak = tid % KP, ar0 = tid / KP;
  const int akc = ak < K ? ak : (K - 1);
  const float a_sc = a.a_scale[akc], a_sh = a.a_shift[akc];
  const int goff = (gr0 * N + gn) * 4, gpass = GROWS * N * 4;
  const int aoff = (ar0 * K + ak) * 4, apass = AROWS * K * 4;
  const long long ngroups = POOL ? (M + a.ns - 1) / a.ns : 0;

  // ---- per-wave MFMA coordinates ----
  const int l31 = lane & 31, lh = lane >> 5;
  int d_rb[DT], d_kb[DT], yoff[DT];
#pragma unroll
  for (int j = 0; j < DT; ++j) {
    const int d = wave + 8 * j;
    d_rb[j] = d % RB;
    d_kb[j] = d / RB;
    const int col = d_kb[j] * 32 + l31;
    yoff[j] = col < K ? ((d_rb[j] * 32 + 4 * lh) * K + col) * 4 : kOobOffset;
  }
  const int rowpitch = K * 4;
  int w_nb[TW], w_kb[TW];
#pragma unroll
  for (int j = 0; j < TW; ++j) {
    const int e = T >= 8 ? wave + 8 * j : wave % T;
    w_nb[j] = e % NTN;
    w_kb[j] = e / NTN;
  }
  const int w_row0 = T >= 8 ? 0 : (wave / T) * WROWS;

  float e_s[DT], e_h[DT], e_m[DT], e_r[DT], cs1[DT], cs2[DT];
#pragma unroll
  for (int j = 0; j < DT; ++j) {
    const int col = d_kb[j] * 32 + l31;
    const int cc = col < K ? col : (K - 1);
    e_s[j] = a.a_scale[cc]; e_h[j] = a.a_shift[cc]; e_m[j] = a.a_mean[cc]; e_r[j] = a.a_rstd[cc];
    cs1[j] = 0.f; cs2[j] = 0.f;
  }

  f32x16 accd[DT], accw[TW];
#pragma unroll
  for (int j = 0; j < DT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accd[j][r] = 0.f;
#pragma unroll
  for (int j = 0; j < TW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[j][r] = 0.f;

  float rg[RGN], ry[GPT], rp[APT], pg[PG];
  int pa[PG];

  // The next tile's operands are prefetched in TWO batches (before the dgrad and before the wgrad MFMAs): a wave
  // can have at most 63 vector-memory operations outstanding (6-bit vmcnt), and one batch of 80-96 dword loads
  // stalled the issue of everything behind it — including the MFMAs — for a memory latency per tile.
  auto load_tile_a = [&](long long tile) {
    const long long m0 = tile * R;
    const rsrc_t rsy = make_rsrc(a.Yl + (size_t)m0 * N, (M - m0) * N * 4);
#pragma unroll
    for (int i = 0; i < GPT; ++i) ry[i] = bload(rsy, goff, i * gpass);
    if (POOL) {
      const long long g_first = m0 / a.ns;
      const rsrc_t rsa = make_rsrc(a.arg + (size_t)g_first * N, (ngroups - g_first) * N * 4);
      const rsrc_t rsg = make_rsrc(a.gP + (size_t)g_first * N, (ngroups - g_first) * N * 4);
#pragma unroll
      for (int e = 0; e < PG; ++e) {
        pa[e] = bload_i(rsa, goff, e * gpass);
        pg[e] = bload(rsg, goff, e * gpass);
      }
    }
  };
  auto load_tile_b = [&](long long tile) {
    const long long m0 = tile * R;
    if (!POOL) {
      const rsrc_t rsg = make_rsrc(a.G + (size_t)m0 * N, (M - m0) * N * 4);
#pragma unroll
      for (int i = 0; i < GPT; ++i) rg[POOL ? 0 : i] = bload(rsg, goff, i * gpass);
    }
    const rsrc_t rsp = make_rsrc(a.Yprev + (size_t)m0 * K, (M - m0) * K * 4);
#pragma unroll
    for (int i = 0; i < APT; ++i) rp[i] = bload(rsp, aoff, i * apass);
  };

  long long tile = blockIdx.x;
  load_tile_a(tile);
  load_tile_b(tile);
  __syncthreads();                               // resident weights visible
  for (; tile < ntiles; tile += gridDim.x) {
    const long long m0 = tile * R;
    // ---- registers of this tile -> LDS (gy transposed, activation row-major) ----
    {
      float gv[GPT];
#pragma unroll
      for (int i = 0; i < GPT; ++i)
        gv[i] = POOL ? __fmaf_rn(c2, ry[i], c3) : __fmaf_rn(c1, rg[POOL ? 0 : i], __fmaf_rn(c2, ry[i], c3));
      if (m0 + R > M) {
        // the one tile that crosses M (wave-uniform): rows past M read zeros, so gy = c3 there — clear them
        asm volatile("; partial tile");
#pragma unroll
        for (int i = 0; i < GPT; ++i) gv[i] = (m0 + gr0 + GROWS * i) < M ? gv[i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < GPT; ++i) gyT[gn * LDT + gr0 + GROWS * i] = gv[i];
#pragma unroll
      for (int i = 0; i < APT; ++i)
        act[(ar0 + AROWS * i) * KP + ak] = fmaxf(__fmaf_rn(rp[i], a_sc, a_sh), 0.f);
    }
    if (POOL) {
      __syncthreads();
      // sparse arg-max patch: the single non-zero of dL/dz per (row group, column)
      const int mrem = (int)((M - m0) < (long long)R ? (M - m0) : (long long)R);
      const long long g_first = m0 / a.ns;
      const int ngrp = (int)((m0 + mrem - 1) / a.ns - g_first) + 1;
#pragma unroll
      for (int e = 0; e < PG; ++e) {
        const int gi = gr0 + GROWS * e;
        const long long row = (g_first + gi) * (long long)a.ns + pa[e] - m0;
        if (gn < N && gi < ngrp && row >= 0 && row < mrem) gyT[gn * LDT + (int)row] += c1 * pg[e];
      }
    }
    __syncthreads();

    // ---- y_{l-1} at the dgrad output positions (mask + sums), then the next tile's operands ----
    float yp[DT][16];
    {
      const rsrc_t rsq = make_rsrc(a.Yprev + (size_t)m0 * K, (M - m0) * K * 4);
#pragma unroll
      for (int j = 0; j < DT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) yp[j][r] = bload(rsq, yoff[j], ((r & 3) + 8 * (r >> 2)) * rowpitch);
    }
    const long long nt = (tile + gridDim.x) < ntiles ? (tile + gridDim.x) : tile;   // past the end: harmless reload
    load_tile_a(nt);

    // ---- dgrad: out[row][k] = sum_n gy[row][n] * W[n][k] ----
#pragma unroll
    for (int s = 0; s < NP / 2; ++s) {
      const int n = 2 * s + lh;
#pragma unroll
      for (int j = 0; j < DT; ++j) {
        const float av = gyT[n * LDT + d_rb[j] * 32 + l31];
        const float bv = Wl[n * KP + d_kb[j] * 32 + l31];
        accd[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accd[j], 0, 0, 0);
      }
    }
    load_tile_b(nt);
    // ---- wgrad: dW[n][k] += sum_row gy[row][n] * act[row][k] ----
#pragma unroll
    for (int s = 0; s < WROWS / 2; ++s) {
      const int row = w_row0 + 2 * s + lh;
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        const float av = gyT[(w_nb[j] * 32 + l31) * LDT + row];
        const float bv = act[row * KP + w_kb[j] * 32 + l31];
        accw[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accw[j], 0, 0, 0);
      }
    }

    // ---- dgrad epilogue: ReLU mask of layer l-1, BN-backward sums, store ----
    const rsrc_t rso = make_rsrc(a.Gout + (size_t)m0 * K, (M - m0) * K * 4);
#pragma unroll
    for (int j = 0; j < DT; ++j) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float y = yp[j][r];
        float v = accd[j][r];
        v = (__fmaf_rn(y, e_s[j], e_h[j]) > 0.f) ? v : 0.f;
        s1 += v;
        s2 = __fmaf_rn(v, (y - e_m[j]) * e_r[j], s2);
        bstore(v, rso, yoff[j], ((r & 3) + 8 * (r >> 2)) * rowpitch);
        accd[j][r] = 0.f;
      }
      cs1[j] += s1;
      cs2[j] += s2;
    }
    __syncthreads();                             // LDS tile free for the next iteration
  }

  // ---- flush the column sums (once per workgroup) ----
  for (int i = tid; i < 2 * KP; i += 512) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < DT; ++j) {
    atomicAdd(&red[d_kb[j] * 32 + l31], cs1[j]);
    atomicAdd(&red[KP + d_kb[j] * 32 + l31], cs2[j]);
  }
  __syncthreads();
  for (int i = tid; i < KP; i += 512) {
    if (i < K) {
      atomicAdd(a.sums + i, (double)red[i]);
      atomicAdd(a.sums + K + i, (double)red[KP + i]);
    }
  }
  // ---- flush dW: accw[j][r] = dW[n = nb*32 + rowmap(r)][k = kb*32 + lane%32] ----
#pragma unroll
  for (int j = 0; j < TW; ++j) {
    const int kcol = w_kb[j] * 32 + l31;
    const int nb = w_nb[j] * 32 + 4 * lh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nb + (r & 3) + 8 * (r >> 2);
      if (n < N && kcol < K) atomicAdd(a.dW + (size_t)n * K + kcol, accw[j][r]);
    }
  }
}

// ---- version 2 (K <= 64): 64-row tiles, double-buffered LDS, role-specialised waves --------------------
// The kernel above runs its eight waves in lockstep (stage -> barrier -> MFMA -> epilogue -> barrier): the MFMA
// pipe idles ~40 % of the time because VALU/LDS work does not overlap with fp32 MFMAs inside a wave and both
// waves of a SIMD are always in the same phase.  Here the row tile is 64 x N, staged into the OTHER half of a
// double-buffered LDS while the current half is consumed, and the waves of a SIMD are given different roles
// with opposite phase order:
//     waves 0-3 ("dgrad"):  stage tile t+1  ->  64 x 64 dgrad tile (one 32x32 block each)  ->  mask / sums / store
//     waves 4-7 ("wgrad"):  dW blocks (n-block w-4, both k-blocks) over the 64 rows        ->  stage tile t+1
// (wave w and w+4 share a SIMD), so one wave's staging and epilogue run under the other's MFMAs.  Both roles
// issue the same number of MFMAs per tile (N/2 resp. 2 * 32 for N = 128; 32 resp. 32 for N = 64).
// One barrier per tile (+ one for the pooled-gradient patch).
//
// FOLD (layer l-1 is the FIRST layer of the stack, its input X needs no gradient and has K0 <= 8 columns): the masked
// input gradient gz = dL/dz_{l-1} is not stored.  The first layer's weight gradient is linear in its BatchNorm-backward
// constants,  dW_{l-1} = (c1 gz + c2 y_{l-1} + c3)^T X = diag(c1) gz^T X + diag(c2) W_{l-1} (X^T X) + c3 (1^T X),
// so this kernel only reduces P1 = gz^T X (16 x K0 FMAs per lane and tile, X tile broadcast from LDS); X^T X and 1^T X
// come from rows_gram_kernel and first_layer_dw_kernel combines them once the constants exist.  Saves the M x K store
// here and the whole first-layer wgrad kernel (which re-read gz and y_{l-1}).
template <int GMODE, int NTN, bool FOLD = false>
__global__ __launch_bounds__(512) void mlp_bwd_fused2_kernel(const BwdArgs a) {
  constexpr bool POOL = GMODE == PRO_POOLG;
  constexpr int XW = 8;                         // padded width of the X tile
  constexpr int R = 64, KTN = 2;
  constexpr int NP = NTN * 32, KP = KTN * 32;
  constexpr int LDT = R + 1;
  constexpr int GROWS = 512 / NP;               // 4 (N = 128) or 8 (N = 64)
  constexpr int GPT = R / GROWS;                // 16 or 8
  constexpr int AROWS = 512 / KP;               // 8
  constexpr int APT = R / AROWS;                // 8
  constexpr int TWN = NTN == 4 ? 2 : 1;         // dW blocks per wgrad wave
  constexpr int PG = POOL ? ((R / 16 + 1 + GROWS - 1) / GROWS) : 1;
  constexpr int RGN = POOL ? 1 : GPT;
  constexpr int GY_SZ = NP * LDT, ACT_SZ = R * KP;

  extern __shared__ float lds[];
  float *gyT0 = lds;                            // [2][NP][LDT]
  float *act0 = gyT0 + 2 * GY_SZ;               // [2][R][KP]
  float *Wl = act0 + 2 * ACT_SZ;                // [NP][KP]
  float *red = Wl + NP * KP;                    // [2][KP]
  float *Xs0 = red + 2 * KP;                    // FOLD: [2][R][XW] (+ [KP][XW] for the final reduction)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const bool dgrad_role = wave < 4;
  const int N = a.N, K = a.K;
  const long long M = a.M;
  const long long ntiles = (M + R - 1) / R;

  // (eight loads in flight per thread: as a rolled loop the 32 dependent-latency iterations were ~30 us of every launch)
#pragma unroll 8
  for (int i = tid; i < NP * KP; i += 512) {
    const int n = i / KP, k = i % KP;
    Wl[i] = (n < N && k < K) ? a.W[(size_t)n * K + k] : 0.f;
  }

  const int gn = tid % NP, gr0 = tid / NP;
  const int gnc = gn < N ? gn : (N - 1);
  const float c1 = a.c1[gnc], c2 = a.c2[gnc], c3 = a.c3[gnc];
  const int ak = tid % KP, ar0 = tid / KP;
  const int akc = ak < K ? ak : (K - 1);
  const float a_sc = a.a_scale[akc], a_sh = a.a_shift[akc];
  const int goff = (gr0 * N + gn) * 4, gpass = GROWS * N * 4;
  const int aoff = (ar0 * K + ak) * 4, apass = AROWS * K * 4;
  const unsigned ns = POOL ? (unsigned)a.ns : 1u;
  const unsigned ngroups = POOL ? (unsigned)((M + ns - 1) / ns) : 0u;      // M < 2^31 in pooled mode

  const int l31 = lane & 31, lh = lane >> 5;
  // dgrad role: block (rb, kb) of the 64 x 64 output tile
  const int d_rb = wave & 1, d_kb = (wave >> 1) & 1;
  const int dcol = d_kb * 32 + l31;
  const int yoff = dcol < K ? ((d_rb * 32 + 4 * lh) * K + dcol) * 4 : kOobOffset;
  const int rowpitch = K * 4;
  const int dcc = dcol < K ? dcol : (K - 1);
  const float e_s = a.a_scale[dcc], e_h = a.a_shift[dcc], e_m = a.a_mean[dcc], e_r = a.a_rstd[dcc];
  float cs1 = 0.f, cs2 = 0.f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 px[FOLD ? XW / 2 : 1];
#pragma unroll
  for (int k = 0; k < (FOLD ? XW / 2 : 1); ++k) px[k] = f2{0.f, 0.f};
  // FOLD: thread (row tid / 8, column tid % 8) of the 64 x 8 X tile; columns past K0 are out of range (read 0)
  const int xoff = (tid % XW) < a.K0 ? ((tid / XW) * a.K0 + (tid % XW)) * 4 : kOobOffset;
  // wgrad role: NTN == 4: n-block (wave-4), k-blocks 0 and 1, all 64 rows; NTN == 2: block (nb, kb) = ((w-4)&1, (w-4)>>1)
  const int w_nb = NTN == 4 ? (wave & 3) : (wave & 1);
  const int w_kb0 = NTN == 4 ? 0 : ((wave >> 1) & 1);

  // one accumulator array for both roles (a wave is either dgrad or wgrad for the whole kernel; two arrays would
  // both be live across the loop and cost 16 VGPRs of a 256-register budget)
  f32x16 accs[TWN];
#pragma unroll
  for (int j = 0; j < TWN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accs[j][r] = 0.f;

  // two register sets: the operands of tiles t+1 and t+2 are in flight / in registers while tile t is computed
  // (a burst of one tile per CU takes ~3 us to stream at full HBM rate plus the latency: with a single set the
  // whole transfer was exposed behind the MFMA phase and the two times simply added up)
  float rg0[RGN], ry0[GPT], rp0[APT], pg0[PG];
  float rg1[RGN], ry1[GPT], rp1[APT], pg1[PG];
  int pa0[PG], pa1[PG];
  float rx0 = 0.f, rx1 = 0.f;

  auto load_tile = [&](long long tile, float (&rg)[RGN], float (&ry)[GPT], float (&rp)[APT], int (&pa)[PG],
                       float (&pg)[PG], float &rx) {
    const long long m0 = tile * R;
    if (FOLD) {
      const rsrc_t rsx = make_rsrc(a.X + (size_t)m0 * a.K0, (M - m0) * a.K0 * 4);
      rx = bload(rsx, xoff, 0);
    }
    const rsrc_t rsy = make_rsrc(a.Yl + (size_t)m0 * N, (M - m0) * N * 4);
#pragma unroll
    for (int i = 0; i < GPT; ++i) ry[i] = bload(rsy, goff, i * gpass);
    if (POOL) {
      const unsigned g_first = (unsigned)m0 / ns;
      const rsrc_t rsa = make_rsrc(a.arg + (size_t)g_first * N, (long long)(ngroups - g_first) * N * 4);
      const rsrc_t rsg = make_rsrc(a.gP + (size_t)g_first * N, (long long)(ngroups - g_first) * N * 4);
#pragma unroll
      for (int e = 0; e < PG; ++e) {
        pa[e] = bload_i(rsa, goff, e * gpass);
        pg[e] = bload(rsg, goff, e * gpass);
      }
    } else {
      const rsrc_t rsg = make_rsrc(a.G + (size_t)m0 * N, (M - m0) * N * 4);
#pragma unroll
      for (int i = 0; i < GPT; ++i) rg[POOL ? 0 : i] = bload(rsg, goff, i * gpass);
    }
    const rsrc_t rsp = make_rsrc(a.Yprev + (size_t)m0 * K, (M - m0) * K * 4);
#pragma unroll
    for (int i = 0; i < APT; ++i) rp[i] = bload(rsp, aoff, i * apass);
  };

  // registers (tile `st`) -> LDS buffer `buf`; keeps the patch operands of that tile in (spa, spg)
  int spa[PG];
  float spg[PG];
  long long p_m0 = 0;
  auto stage = [&](long long st, int buf, float (&rg)[RGN], float (&ry)[GPT], float (&rp)[APT], int (&pa)[PG],
                   float (&pg)[PG], float &rx) {
    const long long m0 = st * R;
    float *gyT = gyT0 + buf * GY_SZ;
    float *act = act0 + buf * ACT_SZ;
    if (FOLD) Xs0[buf * (R * XW) + tid] = rx;
    float gv[GPT];
#pragma unroll
    for (int i = 0; i < GPT; ++i)
      gv[i] = POOL ? __fmaf_rn(c2, ry[i], c3) : __fmaf_rn(c1, rg[POOL ? 0 : i], __fmaf_rn(c2, ry[i], c3));
    if (m0 + R > M) {
      asm volatile("; partial tile");            // rows past M read zeros, so gy = c3 there: clear them (real branch)
#pragma unroll
      for (int i = 0; i < GPT; ++i) gv[i] = (m0 + gr0 + GROWS * i) < M ? gv[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < GPT; ++i) gyT[gn * LDT + gr0 + GROWS * i] = gv[i];
#pragma unroll
    for (int i = 0; i < APT; ++i) act[(ar0 + AROWS * i) * KP + ak] = fmaxf(__fmaf_rn(rp[i], a_sc, a_sh), 0.f);
    if (POOL) {
#pragma unroll
      for (int e = 0; e < PG; ++e) { spa[e] = pa[e]; spg[e] = pg[e]; }
      p_m0 = m0;
    }
  };
  auto patch = [&](int buf) {                    // sparse arg-max patch of the tile staged last
    float *gyT = gyT0 + buf * GY_SZ;
    const long long m0 = p_m0;
    const int mrem = (int)((M - m0) < (long long)R ? (M - m0) : (long long)R);
    const unsigned g_first = (unsigned)m0 / ns;
    const int ngrp = (int)((unsigned)(m0 + mrem - 1) / ns - g_first) + 1;
#pragma unroll
    for (int e = 0; e < PG; ++e) {
      const int gi = gr0 + GROWS * e;
      const long long row = (long long)(g_first + gi) * ns + spa[e] - m0;
      if (gn < N && gi < ngrp && row >= 0 && row < mrem) gyT[gn * LDT + (int)row] += c1 * spg[e];
    }
  };

  const long long stride = gridDim.x;
  const long long my_tiles = (ntiles - blockIdx.x + stride - 1) / stride;          // >= 1 (grid <= ntiles)
  const long long last = blockIdx.x + (my_tiles - 1) * stride;
  auto clampt = [&](long long t) { return t < ntiles ? t : last; };                // past the end: harmless reloads
  long long tile = blockIdx.x;

  // one pipeline iteration: tile `tile` is in LDS buffer `buf`, (rg, ry, ...) hold tile+stride and are staged into
  // buf^1, then refilled with tile+3*stride (the other register set holds tile+2*stride)
  auto iteration = [&](int buf, float (&rg)[RGN], float (&ry)[GPT], float (&rp)[APT], int (&pa)[PG], float (&pg)[PG],
                       float &rx) {
    const long long m0 = tile * R;
    const long long t1 = clampt(tile + stride), t3 = clampt(tile + 3 * stride);
    const float *gyT = gyT0 + buf * GY_SZ;
    const float *act = act0 + buf * ACT_SZ;
    if (dgrad_role) {
      float yp[16];
      const rsrc_t rsq = make_rsrc(a.Yprev + (size_t)m0 * K, (M - m0) * K * 4);
#pragma unroll
      for (int r = 0; r < 16; ++r) yp[r] = bload(rsq, yoff, ((r & 3) + 8 * (r >> 2)) * rowpitch);
      stage(t1, buf ^ 1, rg, ry, rp, pa, pg, rx);
      load_tile(t3, rg, ry, rp, pa, pg, rx);
#pragma unroll
      for (int s = 0; s < NP / 2; ++s) {
        const int n = 2 * s + lh;
        const float av = gyT[n * LDT + d_rb * 32 + l31];
        const float bv = Wl[n * KP + d_kb * 32 + l31];
        accs[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accs[0], 0, 0, 0);
      }
      const rsrc_t rso = make_rsrc(a.Gout + (size_t)m0 * K, (M - m0) * K * 4);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float y = yp[r];
        float v = accs[0][r];
        v = (__fmaf_rn(y, e_s, e_h) > 0.f) ? v : 0.f;
        s1 += v;
        s2 = __fmaf_rn(v, (y - e_m) * e_r, s2);
        if (FOLD) {
          if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);   // keep the X-tile reads from being hoisted en bloc
          const float4 *xr = reinterpret_cast<const float4 *>(Xs0 + buf * (R * XW) +
                                                              (d_rb * 32 + 4 * lh + (r & 3) + 8 * (r >> 2)) * XW);
          const float4 xa = xr[0], xb = xr[1];                    // two broadcast ds_read_b128
          const f2 v2 = {v, v};
          px[0] = __builtin_elementwise_fma(v2, f2{xa.x, xa.y}, px[0]);   // v_pk_fma_f32
          px[FOLD ? 1 : 0] = __builtin_elementwise_fma(v2, f2{xa.z, xa.w}, px[FOLD ? 1 : 0]);
          px[FOLD ? 2 : 0] = __builtin_elementwise_fma(v2, f2{xb.x, xb.y}, px[FOLD ? 2 : 0]);
          px[FOLD ? 3 : 0] = __builtin_elementwise_fma(v2, f2{xb.z, xb.w}, px[FOLD ? 3 : 0]);
        } else {
          bstore(v, rso, yoff, ((r & 3) + 8 * (r >> 2)) * rowpitch);
        }
        accs[0][r] = 0.f;
      }
      cs1 += s1;
      cs2 += s2;
    } else {
#pragma unroll
      for (int s = 0; s < R / 2; ++s) {
        const int row = 2 * s + lh;
        const float av = gyT[(w_nb * 32 + l31) * LDT + row];
#pragma unroll
        for (int j = 0; j < TWN; ++j) {
          const float bv = act[row * KP + (w_kb0 + j) * 32 + l31];
          accs[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accs[j], 0, 0, 0);
        }
      }
      stage(t1, buf ^ 1, rg, ry, rp, pa, pg, rx);
      load_tile(t3, rg, ry, rp, pa, pg, rx);
    }
    __syncthreads();
    if (POOL) {
      patch(buf ^ 1);
      __syncthreads();
    }
    tile += stride;
  };

  load_tile(tile, rg0, ry0, rp0, pa0, pg0, rx0);
  load_tile(clampt(tile + stride), rg1, ry1, rp1, pa1, pg1, rx1);
  __syncthreads();                               // resident weights visible
  stage(tile, 0, rg0, ry0, rp0, pa0, pg0, rx0);
  load_tile(clampt(tile + 2 * stride), rg0, ry0, rp0, pa0, pg0, rx0);
  __syncthreads();
  if (POOL) {
    patch(0);
    __syncthreads();
  }
  // single-exit pair loop + peeled odd iteration (see mlp_gemm_kernel): set 1 holds tile+stride, set 0 tile+2*stride
  for (long long pair = my_tiles >> 1; pair > 0; --pair) {
    iteration(0, rg1, ry1, rp1, pa1, pg1, rx1);
    iteration(1, rg0, ry0, rp0, pa0, pg0, rx0);
  }
  if (my_tiles & 1) iteration(0, rg1, ry1, rp1, pa1, pg1, rx1);

  // ---- flush the column sums (dgrad waves; both row blocks of a column add up in LDS) ----
  float *redP = Xs0 + 2 * R * XW;                // FOLD: [KP][XW]
  for (int i = tid; i < 2 * KP; i += 512) red[i] = 0.f;
  if (FOLD) redP[tid] = 0.f;                     // KP * XW == 512
  __syncthreads();
  if (dgrad_role) {
    atomicAdd(&red[dcol], cs1);
    atomicAdd(&red[KP + dcol], cs2);
    if (FOLD) {
#pragma unroll
      for (int k = 0; k < (FOLD ? XW / 2 : 1); ++k) {
        atomicAdd(&redP[dcol * XW + 2 * k], px[k].x);
        atomicAdd(&redP[dcol * XW + 2 * k + 1], px[k].y);
      }
    }
  }
  __syncthreads();
  if (FOLD) {
    const int c = tid / XW, k = tid % XW;
    if (c < K && k < a.K0) atomicAdd(a.P1 + (size_t)c * a.K0 + k, redP[tid]);
  }
  for (int i = tid; i < KP; i += 512) {
    if (i < K) {
      atomicAdd(a.sums + i, (double)red[i]);
      atomicAdd(a.sums + K + i, (double)red[KP + i]);
    }
  }
  // ---- flush dW (wgrad waves) ----
  if (!dgrad_role) {
#pragma unroll
    for (int j = 0; j < TWN; ++j) {
      const int kcol = (w_kb0 + j) * 32 + l31;
      const int nb = w_nb * 32 + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nb + (r & 3) + 8 * (r >> 2);
        if (n < N && kcol < K) atomicAdd(a.dW + (size_t)n * K + kcol, accs[j][r]);
      }
    }
  }
}

template <int GMODE, int NTN, bool FOLD = false>
int launch_fused2(const BwdArgs &a, hipStream_t s) {
  constexpr int NP = NTN * 32, KP = 64, R = 64;
  constexpr size_t lds_bytes = (size_t)(2 * NP * (R + 1) + 2 * R * KP + NP * KP + 2 * KP +
                                        (FOLD ? 2 * R * 8 + KP * 8 : 0)) * sizeof(float);
  static_assert(lds_bytes <= 160 * 1024, "LDS budget of one CU");
  auto kern = mlp_bwd_fused2_kernel<GMODE, NTN, FOLD>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
      return pn2_check_launch();
    attr_set = true;
  }
  const long long ntiles = (a.M + R - 1) / R;
  long long gx = lds_bytes * 2 <= 160 * 1024 ? 512 : 256;      // workgroups per CU by LDS
  if (gx > ntiles) gx = ntiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(512), lds_bytes, s, a);
  return pn2_check_launch();
}

template <int GMODE, int NTN, int KTN, int R>
int launch_fused(const BwdArgs &a, hipStream_t s) {
  constexpr int NP = NTN * 32, KP = KTN * 32;
  constexpr size_t lds_bytes = (size_t)(NP * (R + 1) + R * KP + NP * KP + 2 * KP) * sizeof(float);
  static_assert(lds_bytes <= 160 * 1024, "LDS budget of one CU");
  auto kern = mlp_bwd_fused_kernel<GMODE, NTN, KTN, R>;
  static bool attr_set = false;                  // per instantiation
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
      return pn2_check_launch();
    attr_set = true;
  }
  const long long ntiles = (a.M + R - 1) / R;
  long long gx = 256;                            // one 130 KB workgroup per CU
  if (gx > ntiles) gx = ntiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(512), lds_bytes, s, a);
  return pn2_check_launch();
}

template <int GMODE>
int dispatch_fused(const BwdArgs &a, hipStream_t s) {
  const int ntn = a.N <= 64 ? 2 : 4, ktn = a.K <= 64 ? 2 : 4;
  if (a.X) return (ntn == 2 && ktn == 2) ? launch_fused2<GMODE, 2, true>(a, s) : PN2_EINVAL;
  // N, K <= 64: role-specialised version 2 (1.16 vs 1.31 ms at M = 4.2M); N = 128: version 1 (1.67 vs 2.44 ms —
  // the two register sets of 64 x 128 gy tiles push version 2 past 256 VGPRs).
  if (ntn == 2 && ktn == 2) return launch_fused2<GMODE, 2>(a, s);
  if (ntn == 4 && ktn == 2) return launch_fused<GMODE, 4, 2, 128>(a, s);
  if (ntn == 2 && ktn == 4) return launch_fused<GMODE, 2, 4, 128>(a, s);
  return PN2_EINVAL;
}

}  // namespace

// N, K in (32, 128] with at least one of them <= 64: the 128 x 128 layer needs 64-row tiles to fit its weights in LDS
// and was measured slower than the two-kernel path (0.94 vs 0.79 ms at M = 1M)
extern "C" int pn2_mlp_bwd_fused_supported(int N, int K) {
  return N > 32 && N <= 128 && K > 32 && K <= 128 && (N <= 64 || K <= 64);
}

extern "C" int pn2_mlp_bwd_fused(long long M, int N, int K, int gmode, const float *G, const float *Yl,
                                 const float *consts, const int *arg, const float *gP, int ns, const float *W,
                                 const float *Yprev, const float *a_fin, float *Gout, double *sums, float *dW,
                                 void *stream) {
  if (M < 0 || !pn2_mlp_bwd_fused_supported(N, K)) return PN2_EINVAL;
  if (gmode != PRO_GY && gmode != PRO_POOLG) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!Yl || !consts || !W || !Yprev || !a_fin || !Gout || !sums || !dW) return PN2_ENULL;
  if (gmode == PRO_GY && !G) return PN2_ENULL;
  if (gmode == PRO_POOLG && (!arg || !gP || ns < 16 || M >= 0x7fffffffLL)) return PN2_EINVAL;
  BwdArgs a;
  a.G = G; a.Yl = Yl; a.c1 = consts; a.c2 = consts + N; a.c3 = consts + 2 * (size_t)N;
  a.arg = arg; a.gP = gP; a.W = W; a.Yprev = Yprev;
  a.a_mean = a_fin; a.a_rstd = a_fin + K; a.a_scale = a_fin + 2 * (size_t)K; a.a_shift = a_fin + 3 * (size_t)K;
  a.Gout = Gout; a.sums = sums; a.dW = dW; a.M = M; a.N = N; a.K = K; a.ns = ns;
  a.X = nullptr; a.P1 = nullptr; a.K0 = 0;
  hipStream_t s = (hipStream_t)stream;
  return gmode == PRO_GY ? dispatch_fused<PRO_GY>(a, s) : dispatch_fused<PRO_POOLG>(a, s);
}

// ---- first-layer fold -------------------------------------------------------------------------------------------
namespace {

// gram[K0*K0] = X^T X, gram[K0*K0 + k] = column sums of X  (fp64 accumulation; X is [M][K0], K0 <= 8)
__global__ __launch_bounds__(256) void rows_gram_kernel(long long M, int K0, const float *__restrict__ X,
                                                        double *__restrict__ gram) {
  __shared__ double part[4][8 * 8 + 8];
  float s[8][8], c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s[i][j] = 0.f;
  }
  double ds[8][8], dc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    dc[i] = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) ds[i][j] = 0.0;
  }
  int n = 0;
  // eight rows in flight per thread, at most 256 workgroups: every workgroup ends with 42 fp64 atomics onto the SAME 42
  // addresses, and same-address atomics retire one per ~35 ns at the L2 (1024 workgroups: 36 us of the kernel's 76 were the
  // queue in front of gram[0]); a row-at-a-time loop on the other hand waits for one memory round trip per row
  constexpr int RU = 8;
  const long long stride = (long long)gridDim.x * 256;
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < M; r += RU * stride) {
    float xs[RU][8];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const long long ru = r + u * stride;
#pragma unroll
      for (int i = 0; i < 8; ++i) xs[u][i] = (i < K0 && ru < M) ? X[(size_t)ru * K0 + i] : 0.f;      // (a row past M adds zeros)
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        c[i] += xs[u][i];
#pragma unroll
        for (int j = i; j < 8; ++j) s[i][j] = __fmaf_rn(xs[u][i], xs[u][j], s[i][j]);
      }
    }
    if (++n == 8) {                                // bound the fp32 partial sums (64 rows)
      n = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        dc[i] += (double)c[i]; c[i] = 0.f;
#pragma unroll
        for (int j = i; j < 8; ++j) { ds[i][j] += (double)s[i][j]; s[i][j] = 0.f; }
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    double v = dc[i] + (double)c[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) part[wave][64 + i] = v;
#pragma unroll
    for (int j = i; j < 8; ++j) {
      double w = ds[i][j] + (double)s[i][j];
      for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o);
      if (lane == 0) part[wave][i * 8 + j] = w;
    }
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < 72) {
    const int i = t < 64 ? t / 8 : t - 64, j = t < 64 ? t % 8 : 0;
    if (t >= 64) {
      if (i < K0) atomicAdd(gram + K0 * K0 + i, part[0][t] + part[1][t] + part[2][t] + part[3][t]);
    } else if (i < K0 && j < K0) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;      // upper triangle holds the sums
      const int u = lo * 8 + hi;
      atomicAdd(gram + i * K0 + j, part[0][u] + part[1][u] + part[2][u] + part[3][u]);
    }
  }
}

// dW0[n][k] = c1[n] P1[n][k] + c2[n] sum_j W0[n][j] S[j][k] + c3[n] colsum[k]
__global__ void first_layer_dw_kernel(int N, int K0, const float *__restrict__ consts, const float *__restrict__ P1,
                                      const float *__restrict__ W0, const double *__restrict__ gram,
                                      float *__restrict__ dW0) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * K0) return;
  const int n = e / K0, k = e % K0;
  double ws = 0.0;
  for (int j = 0; j < K0; ++j) ws += (double)W0[n * K0 + j] * gram[j * K0 + k];
  const double v = (double)consts[n] * (double)P1[e] + (double)consts[N + n] * ws +
                   (double)consts[2 * N + n] * gram[K0 * K0 + k];
  dW0[e] = (float)v;
}

}  // namespace

extern "C" int pn2_mlp_bwd_fused_fold_supported(int N, int K, int K0) {
  return N > 32 && N <= 64 && K > 32 && K <= 64 && K0 >= 1 && K0 <= 8;
}

extern "C" int pn2_mlp_bwd_fused_fold(long long M, int N, int K, int gmode, const float *G, const float *Yl,
                                      const float *consts, const int *arg, const float *gP, int ns, const float *W,
                                      const float *Yprev, const float *a_fin, const float *X, int K0, double *sums,
                                      float *dW, float *P1, void *stream) {
  if (M < 0 || !pn2_mlp_bwd_fused_fold_supported(N, K, K0)) return PN2_EINVAL;
  if (gmode != PRO_GY && gmode != PRO_POOLG) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!Yl || !consts || !W || !Yprev || !a_fin || !X || !sums || !dW || !P1) return PN2_ENULL;
  if (gmode == PRO_GY && !G) return PN2_ENULL;
  if (gmode == PRO_POOLG && (!arg || !gP || ns < 16 || M >= 0x7fffffffLL)) return PN2_EINVAL;
  BwdArgs a;
  a.G = G; a.Yl = Yl; a.c1 = consts; a.c2 = consts + N; a.c3 = consts + 2 * (size_t)N;
  a.arg = arg; a.gP = gP; a.W = W; a.Yprev = Yprev;
  a.a_mean = a_fin; a.a_rstd = a_fin + K; a.a_scale = a_fin + 2 * (size_t)K; a.a_shift = a_fin + 3 * (size_t)K;
  a.Gout = nullptr; a.sums = sums; a.dW = dW; a.M = M; a.N = N; a.K = K; a.ns = ns;
  a.X = X; a.P1 = P1; a.K0 = K0;
  hipStream_t s = (hipStream_t)stream;
  return gmode == PRO_GY ? dispatch_fused<PRO_GY>(a, s) : dispatch_fused<PRO_POOLG>(a, s);
}

extern "C" int pn2_rows_gram(long long M, int K0, const float *X, double *gram, void *stream) {
  if (M < 0 || K0 < 1 || K0 > 8) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!X || !gram) return PN2_ENULL;
  long long blocks = (M + 256 * 16 - 1) / (256 * 16);
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(rows_gram_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, M, K0, X, gram);
  return pn2_check_launch();
}

extern "C" int pn2_first_layer_dw(int N, int K0, const float *consts, const float *P1, const float *W0,
                                  const double *gram, float *dW0, void *stream) {
  if (N <= 0 || K0 < 1 || K0 > 8) return PN2_EINVAL;
  if (!consts || !P1 || !W0 || !gram || !dW0) return PN2_ENULL;
  hipLaunchKernelGGL(first_layer_dw_kernel, dim3((N * K0 + 127) / 128), dim3(128), 0, (hipStream_t)stream, N, K0,
                     consts, P1, W0, gram, dW0);
  return pn2_check_launch();
}
