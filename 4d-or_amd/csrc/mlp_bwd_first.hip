// mlp_bwd_first.hip — backward of the layer above a stack's FIRST layer when that layer's output was never stored.
//
// The forward of such a stack (pn2_mlp_gemm_first, csrc/mlp_gemm.hip PRO_FIRST) recomputes y_0 = X W0^T from the <= 8-column
// input rows while it stages the second layer's A tiles; this is the matching backward: the role-specialised one-pass
// dgrad + wgrad kernel of csrc/mlp_bwd_fused.hip (version 2, first-layer FOLD: gz = dL/dz_0 is not stored, only
// P1 = gz^T X is reduced) with y_0 recomputed as well.  The X tiles run one tile AHEAD of the other operands through a
// ring of three LDS copies, and staging a tile computes y_0[row][k] = X[row] . W0[k] (the forward's FMA chain, bit for
// bit) into a raw copy (ReLU mask and yhat of the epilogue) next to relu(bn(.)) (wgrad operand): 8 FMAs per element
// instead of a 4-byte read — the stored-y_0 kernel is HBM-bound on three M x 64 streams, this one reads two.
//
// A separate translation unit on purpose: inside the shared template the extra branches tipped hipcc's register
// allocation of EVERY instantiation from 219-247 VGPRs without scratch to 256 + 240-420 bytes of scratch (0.87 -> 2.5 ms
// at the SA1 shape).
#include "pn2_common.h"
#include "mlp_common.h"
#include "x3_common.h"

#include "../../include/pn2_hip.h"

#include <stdlib.h>

namespace {

struct FirstArgs {
  const float *G;      // PRO_GY: g = dL/dz_l [M][N]
  const float *Yl;     // y_l [M][N]
  const float *c1, *c2, *c3;
  const int *arg;      // PRO_POOLG: [M/ns][N]
  const float *gP;     // PRO_POOLG: [M/ns][N]
  const float *W;      // [N][K]
  const float *W0;     // [K][K0]: y_{l-1} = X W0^T
  const float *a_mean, *a_rstd, *a_scale, *a_shift;   // layer l-1 BatchNorm, per column k
  double *sums;        // [2][K]: sum g', sum g' * yhat_{l-1}
  float *dW;           // [N][K], accumulated with atomics (caller zero-fills)
  const float *X;      // [M][K0]
  float *P1;           // [K][K0] += (dL/dz_{l-1})^T X
  long long M;
  int N, K, ns, K0;
};

constexpr int XW = 8;                           // padded width of an X tile
constexpr int R = 64, NP = 64, KP = 64;         // rows per tile, padded N and K (32 < N, K <= 64)
constexpr int LDT = R + 1;
constexpr int GROWS = 512 / NP, GPT = R / GROWS;
constexpr int GY_SZ = NP * LDT, ACT_SZ = R * KP;
constexpr int kFirstLds = 2 * GY_SZ + 2 * ACT_SZ + NP * KP + 2 * KP + 3 * R * XW + KP * XW + XW * KP;   // floats

// Waves 0-3 ("dgrad"): stage gy of tile t+1 -> 32 x 32 block of gy W (32 MFMAs) -> ReLU mask / BatchNorm-backward sums /
// P1 from y_{l-1} held in registers -> y_{l-1} block of tile t+1: FOUR MFMAs on the X tile (exactly the products of the
// forward GEMM), kept for the next epilogue and written as relu(bn(.)) into the activation tile of t+1.
// Waves 4-7 ("wgrad"): 32 x 32 block of gy^T act over the 64 rows (32 MFMAs) -> stage gy of tile t+1.
// Wave w and w+4 share a SIMD and run the phases in opposite order; one barrier per tile.
// X3: both 64-deep products (gy W and gy^T act) on the split-bf16 product of x3_common.h — the operands are the same LDS reads
// (eight per 16-deep chunk and lane), split in registers, six bf16 matrix instructions per chunk instead of eight fp32 ones at
// half their length each, and — unlike the fp32 matrix instruction — issued beside the vector work of the wave that shares
// the SIMD.  The recomputation of y_0 (four matrix instructions per tile) stays exact: it decides the ReLU mask.
template <int GMODE, bool X3>
__global__ __launch_bounds__(512) void mlp_bwd_first_kernel(const FirstArgs a) {
  constexpr bool POOL = GMODE == PRO_POOLG;
  constexpr int PG = POOL ? ((R / 16 + 1 + GROWS - 1) / GROWS) : 1;
  constexpr int RGN = POOL ? 1 : GPT;

  extern __shared__ float lds[];
  float *gyT0 = lds;                            // [2][NP][LDT]   gy transposed
  float *act0 = gyT0 + 2 * GY_SZ;               // [2][R][KP]     relu(bn(y_{l-1}))
  float *Wl = act0 + 2 * ACT_SZ;                // [NP][KP]       resident weights
  float *red = Wl + NP * KP;                    // [2][KP]
  float *Xs0 = red + 2 * KP;                    // [3][R][XW]     X tiles: computed | next | the one after
  float *redP = Xs0 + 3 * R * XW;               // [KP][XW]
  float *W0s = redP + KP * XW;                  // [XW][KP]       W0 transposed, zero padded

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const bool dgrad_role = wave < 4;
  const int N = a.N, K = a.K;
  const long long M = a.M;
  const long long ntiles = (M + R - 1) / R;

#pragma unroll 8
  for (int i = tid; i < NP * KP; i += 512) {
    const int n = i / KP, k = i % KP;
    Wl[i] = (n < N && k < K) ? a.W[(size_t)n * K + k] : 0.f;
  }
  {
    const int j = tid / KP, k = tid % KP;       // XW * KP == 512
    W0s[tid] = (k < K && j < a.K0) ? a.W0[(size_t)k * a.K0 + j] : 0.f;
  }

  const int gn = tid % NP, gr0 = tid / NP;
  const int gnc = gn < N ? gn : (N - 1);
  const float c1 = a.c1[gnc], c2 = a.c2[gnc], c3 = a.c3[gnc];
  const int goff = (gr0 * N + gn) * 4, gpass = GROWS * N * 4;
  const unsigned ns = POOL ? (unsigned)a.ns : 1u;
  const unsigned ngroups = POOL ? (unsigned)((M + ns - 1) / ns) : 0u;      // M < 2^31 in pooled mode

  const int l31 = lane & 31, lh = lane >> 5;
  const int d_rb = wave & 1, d_kb = (wave >> 1) & 1;                       // dgrad role: block of the 64 x 64 tile
  const int dcol = d_kb * 32 + l31;
  const int dcc = dcol < K ? dcol : (K - 1);
  const float e_s = a.a_scale[dcc], e_h = a.a_shift[dcc], e_m = a.a_mean[dcc], e_r = a.a_rstd[dcc];
  float cs1 = 0.f, cs2 = 0.f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 px[XW / 2];
#pragma unroll
  for (int k = 0; k < XW / 2; ++k) px[k] = f2{0.f, 0.f};
  // thread (row tid / 8, column tid % 8) of a 64 x 8 X tile; columns past K0 are out of range (read 0)
  const int xoff = (tid % XW) < a.K0 ? ((tid / XW) * a.K0 + (tid % XW)) * 4 : kOobOffset;
  const int w_nb = wave & 1, w_kb = (wave >> 1) & 1;                       // wgrad role: block of dW

  x3_frag wB[X3 ? NP / 16 : 1];                  // X3, dgrad waves: B fragments of the wave's column block of W
  f32x16 acc;                                    // dgrad block or dW block (a wave has one role for the whole kernel)
  f32x16 ypv;                                    // dgrad role: y_{l-1} block of the tile being computed
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; ypv[r] = 0.f; }

  float rg0[RGN], ry0[GPT], pg0[PG];
  float rg1[RGN], ry1[GPT], pg1[PG];
  int pa0[PG], pa1[PG];
  float xring;

  auto load_x = [&](long long tile) {
    const long long m0 = tile * R;
    const rsrc_t rsx = make_rsrc(a.X + (size_t)m0 * a.K0, (M - m0) * a.K0 * 4);
    return bload(rsx, xoff, 0);
  };
  auto load_tile = [&](long long tile, float (&rg)[RGN], float (&ry)[GPT], int (&pa)[PG], float (&pg)[PG]) {
    const long long m0 = tile * R;
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
  };

  int spa[PG];
  float spg[PG];
  long long p_m0 = 0;
  auto stage = [&](long long st, int buf, float (&rg)[RGN], float (&ry)[GPT], int (&pa)[PG], float (&pg)[PG]) {
    const long long m0 = st * R;
    float *gyT = gyT0 + buf * GY_SZ;
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
  // y_{l-1} block (d_rb, d_kb) of the tile whose X rows are ring copy `xpos`: kept in ypv, relu(bn(.)) -> act[buf]
  auto compute_y = [&](int xpos, int buf) {
    const float *xt = Xs0 + xpos * (R * XW) + (d_rb * 32 + l31) * XW + lh;
    f32x16 y;
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = 0.f;
#pragma unroll
    for (int s = 0; s < XW / 2; ++s)
      y = __builtin_amdgcn_mfma_f32_32x32x2f32(xt[2 * s], W0s[(2 * s + lh) * KP + dcol], y, 0, 0, 0);
    float *act = act0 + buf * ACT_SZ + (d_rb * 32 + 4 * lh) * KP + dcol;
#pragma unroll
    for (int r = 0; r < 16; ++r) act[((r & 3) + 8 * (r >> 2)) * KP] = fmaxf(__fmaf_rn(y[r], e_s, e_h), 0.f);
    ypv = y;
  };

  const long long stride = gridDim.x;
  const long long my_tiles = (ntiles - blockIdx.x + stride - 1) / stride;          // >= 1 (grid <= ntiles)
  const long long last = blockIdx.x + (my_tiles - 1) * stride;
  auto clampt = [&](long long t) { return t < ntiles ? t : last; };                // past the end: harmless reloads
  long long tile = blockIdx.x;
  int xq = 0;                                    // ring position of the X copy of the tile being computed

  // one pipeline iteration: tile `tile` is in LDS buffer `buf`, (rg, ry, ...) hold tile+stride and are staged into
  // buf^1, then refilled with tile+3*stride (the other register set holds tile+2*stride)
  auto iteration = [&](int buf, float (&rg)[RGN], float (&ry)[GPT], int (&pa)[PG], float (&pg)[PG]) {
    const long long t1 = clampt(tile + stride), t3 = clampt(tile + 3 * stride);
    const float *gyT = gyT0 + buf * GY_SZ;
    const float *act = act0 + buf * ACT_SZ;
    // X rows of tile + 2 stride (loaded an iteration ago) enter the ring two tiles ahead of their use
    const int xq1 = xq == 2 ? 0 : xq + 1, xq2 = xq1 == 2 ? 0 : xq1 + 1;
    Xs0[xq2 * (R * XW) + tid] = xring;
    xring = load_x(t3);
    if (dgrad_role) {
      stage(t1, buf ^ 1, rg, ry, pa, pg);
      load_tile(t3, rg, ry, pa, pg);
      if (X3) {
        // chunk c: contraction indices n = 16 c + 2 i + lh (i = 0..7) — the order the fp32 loop walks them in
        const float *ga = gyT + lh * LDT + d_rb * 32 + l31;
        float av[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) av[0][i] = ga[2 * i * LDT];
#pragma unroll
        for (int c = 0; c < NP / 16; ++c) {
          if (c + 1 < NP / 16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) av[(c + 1) & 1][i] = ga[(16 * (c + 1) + 2 * i) * LDT];
          }
          x3_frag fa;
          x3_split8(av[c & 1], fa);
          x3_mma(fa, wB[X3 ? c : 0], acc);
        }
      } else {
#pragma unroll
        for (int s = 0; s < NP / 2; ++s) {
          const int n = 2 * s + lh;
          const float av = gyT[n * LDT + d_rb * 32 + l31];
          const float bv = Wl[n * KP + d_kb * 32 + l31];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float y = ypv[r];
        float v = acc[r];
        v = (__fmaf_rn(y, e_s, e_h) > 0.f) ? v : 0.f;
        s1 += v;
        s2 = __fmaf_rn(v, (y - e_m) * e_r, s2);
        if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);   // keep the X-tile reads from being hoisted en bloc
        const float4 *xr = reinterpret_cast<const float4 *>(Xs0 + xq * (R * XW) +
                                                            (d_rb * 32 + 4 * lh + (r & 3) + 8 * (r >> 2)) * XW);
        const float4 xa = xr[0], xb = xr[1];                    // two broadcast ds_read_b128
        const f2 v2 = {v, v};
        px[0] = __builtin_elementwise_fma(v2, f2{xa.x, xa.y}, px[0]);   // v_pk_fma_f32
        px[1] = __builtin_elementwise_fma(v2, f2{xa.z, xa.w}, px[1]);
        px[2] = __builtin_elementwise_fma(v2, f2{xb.x, xb.y}, px[2]);
        px[3] = __builtin_elementwise_fma(v2, f2{xb.z, xb.w}, px[3]);
        acc[r] = 0.f;
      }
      cs1 += s1;
      cs2 += s2;
      compute_y(xq1, buf ^ 1);
    } else {
      if (X3) {
        // chunk c: rows 16 c + 2 i + lh of the tile
        const float *ga = gyT + (w_nb * 32 + l31) * LDT + lh;
        const float *ba = act + lh * KP + w_kb * 32 + l31;
        float av[2][8], bv[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { av[0][i] = ga[2 * i]; bv[0][i] = ba[2 * i * KP]; }
#pragma unroll
        for (int c = 0; c < R / 16; ++c) {
          if (c + 1 < R / 16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              av[(c + 1) & 1][i] = ga[16 * (c + 1) + 2 * i];
              bv[(c + 1) & 1][i] = ba[(16 * (c + 1) + 2 * i) * KP];
            }
          }
          x3_frag fa, fb;
          x3_split8(av[c & 1], fa);
          x3_split8(bv[c & 1], fb);
          x3_mma(fa, fb, acc);
        }
      } else {
#pragma unroll
        for (int s = 0; s < R / 2; ++s) {
          const int row = 2 * s + lh;
          const float av = gyT[(w_nb * 32 + l31) * LDT + row];
          const float bv = act[row * KP + w_kb * 32 + l31];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
      }
      stage(t1, buf ^ 1, rg, ry, pa, pg);
      load_tile(t3, rg, ry, pa, pg);
    }
    __syncthreads();
    if (POOL) {
      patch(buf ^ 1);
      __syncthreads();
    }
    tile += stride;
    xq = xq1;
  };

  load_tile(tile, rg0, ry0, pa0, pg0);
  load_tile(clampt(tile + stride), rg1, ry1, pa1, pg1);
  {
    // X of the first two tiles: ring positions 0 and 1; the third waits in its register
    const float xa_ = load_x(tile), xb_ = load_x(clampt(tile + stride));
    xring = load_x(clampt(tile + 2 * stride));
    Xs0[tid] = xa_;
    Xs0[R * XW + tid] = xb_;
  }
  __syncthreads();                               // resident weights and the first X tiles visible
  // X3, dgrad waves: the B fragments of this wave's column block of W, split once (chunk c: rows n = 16 c + 2 i + lh)
  if (X3 && dgrad_role) {
#pragma unroll
    for (int c = 0; c < NP / 16; ++c) {
      float wv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) wv[i] = Wl[(16 * c + 2 * i + lh) * KP + d_kb * 32 + l31];
      x3_split8(wv, wB[X3 ? c : 0]);
    }
  }
  stage(tile, 0, rg0, ry0, pa0, pg0);
  if (dgrad_role) compute_y(0, 0);
  load_tile(clampt(tile + 2 * stride), rg0, ry0, pa0, pg0);
  __syncthreads();
  if (POOL) {
    patch(0);
    __syncthreads();
  }
  // single-exit pair loop + peeled odd iteration (see mlp_gemm_kernel): set 1 holds tile+stride, set 0 tile+2*stride
  for (long long pair = my_tiles >> 1; pair > 0; --pair) {
    iteration(0, rg1, ry1, pa1, pg1);
    iteration(1, rg0, ry0, pa0, pg0);
  }
  if (my_tiles & 1) iteration(0, rg1, ry1, pa1, pg1);

  // ---- flush the column sums and P1 (dgrad waves; both row blocks of a column add up in LDS) ----
  for (int i = tid; i < 2 * KP; i += 512) red[i] = 0.f;
  redP[tid] = 0.f;                               // KP * XW == 512
  __syncthreads();
  if (dgrad_role) {
    atomicAdd(&red[dcol], cs1);
    atomicAdd(&red[KP + dcol], cs2);
#pragma unroll
    for (int k = 0; k < XW / 2; ++k) {
      atomicAdd(&redP[dcol * XW + 2 * k], px[k].x);
      atomicAdd(&redP[dcol * XW + 2 * k + 1], px[k].y);
    }
  }
  __syncthreads();
  {
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
    const int kcol = w_kb * 32 + l31;
    const int nb = w_nb * 32 + 4 * lh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nb + (r & 3) + 8 * (r >> 2);
      if (n < N && kcol < K) atomicAdd(a.dW + (size_t)n * K + kcol, acc[r]);
    }
  }
}

template <int GMODE, bool X3 = false>
int launch_first(const FirstArgs &a, hipStream_t s) {
  constexpr size_t lds_bytes = (size_t)kFirstLds * sizeof(float);
  static_assert(lds_bytes <= 160 * 1024, "LDS budget of one CU");
  auto kern = mlp_bwd_first_kernel<GMODE, X3>;
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

}  // namespace

// pn2_mlp_bwd_fused_fold when the forward never stored y_{l-1} (pn2_mlp_gemm_first): recomputed from X and W0 [K][K0].
namespace {
int bwd_fold_first(bool x3, long long M, int N, int K, int gmode, const float *G, const float *Yl, const float *consts,
                   const int *arg, const float *gP, int ns, const float *W, const float *W0, const float *a_fin, const float *X,
                   int K0, double *sums, float *dW, float *P1, void *stream) {
  if (M < 0 || !pn2_mlp_bwd_fused_fold_supported(N, K, K0)) return PN2_EINVAL;
  if (gmode != PRO_GY && gmode != PRO_POOLG) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!Yl || !consts || !W || !W0 || !a_fin || !X || !sums || !dW || !P1) return PN2_ENULL;
  if (gmode == PRO_GY && !G) return PN2_ENULL;
  if (gmode == PRO_POOLG && (!arg || !gP || ns < 16 || M >= 0x7fffffffLL)) return PN2_EINVAL;
  FirstArgs a;
  a.G = G; a.Yl = Yl; a.c1 = consts; a.c2 = consts + N; a.c3 = consts + 2 * (size_t)N;
  a.arg = arg; a.gP = gP; a.W = W; a.W0 = W0;
  a.a_mean = a_fin; a.a_rstd = a_fin + K; a.a_scale = a_fin + 2 * (size_t)K; a.a_shift = a_fin + 3 * (size_t)K;
  a.sums = sums; a.dW = dW; a.M = M; a.N = N; a.K = K; a.ns = ns;
  a.X = X; a.P1 = P1; a.K0 = K0;
  hipStream_t s = (hipStream_t)stream;
  if (x3) return gmode == PRO_GY ? launch_first<PRO_GY, true>(a, s) : launch_first<PRO_POOLG, true>(a, s);
  return gmode == PRO_GY ? launch_first<PRO_GY>(a, s) : launch_first<PRO_POOLG>(a, s);
}
}  // namespace

extern "C" int pn2_mlp_bwd_fused_fold_first(long long M, int N, int K, int gmode, const float *G, const float *Yl,
                                            const float *consts, const int *arg, const float *gP, int ns, const float *W,
                                            const float *W0, const float *a_fin, const float *X, int K0, double *sums,
                                            float *dW, float *P1, void *stream) {
  return bwd_fold_first(false, M, N, K, gmode, G, Yl, consts, arg, gP, ns, W, W0, a_fin, X, K0, sums, dW, P1, stream);
}

// The same backward with its two 64-deep products on the split-bf16 ("f32x3") product (x3_common.h); y_0 is still re-formed exactly.
extern "C" int pn2_x3_bwd_fold_first(long long M, int N, int K, int gmode, const float *G, const float *Yl, const float *consts,
                                     const int *arg, const float *gP, int ns, const float *W, const float *W0,
                                     const float *a_fin, const float *X, int K0, double *sums, float *dW, float *P1,
                                     void *stream) {
  return bwd_fold_first(true, M, N, K, gmode, G, Yl, consts, arg, gP, ns, W, W0, a_fin, X, K0, sums, dW, P1, stream);
}
