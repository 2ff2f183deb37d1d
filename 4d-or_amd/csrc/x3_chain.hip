// x3_chain.hip — a set-abstraction level in EVAL mode as ONE kernel (reference OPS/pointnet2_modules.py:29-74:
// group -> [Conv1x1 + BatchNorm + ReLU] x k -> max over the neighbourhood).
//
// In eval mode BatchNorm is an affine map with constant (running) statistics: it folds into the convolution's weight and a
// bias, so nothing between the gather and the pooled maximum depends on the batch.  The kernel keeps every activation in
// REGISTERS from the gathered input rows to the maximum (x3_common.h: hidden layers are computed transposed, so an
// accumulator tile is directly the next layer's operand fragment); no (rows x channels) tensor exists in HBM or LDS.
// Arithmetic: the split-bf16 ("f32x3") product — fp32-grade error at 6/16 of the exact fp32 MFMA's matrix time.
//
//   workgroup = 4 or 8 waves; a wave owns 32 consecutive rows (row = (centre, sample)) of a pass; persistent workgroups
//   walk the passes.  Per pass:
//     input stage   IN_SMALL: gather [xyz[idx] - centre | feats[idx] | 1] (<= 15 columns + the bias column) and apply the
//                             first layer (one 16-wide chunk, weights resident in LDS)
//                   IN_LIFT : the first layer was applied per POINT before the grouping (Pq = W_f f + W_x x / r per point,
//                             Q = W_x c / r - b per centre: csrc/group_lift.hip); gather relu(Pq[idx] - Q[centre])
//     middle layer  (optional) transposed product, bias in the accumulator's initial value, ReLU + split in registers
//     last layer    row-major product (lane = channel, registers = rows): maximum over the rows of a centre in the
//                   epilogue, ReLU, one coalesced 128-byte store per (centre, 32 channels)
//   The weights of the middle and last layer stream through a 2-slot LDS ring (24 KB slots = 8 (chunk, tile) units) filled
//   by global_load_lds (direct global -> LDS, no registers) one step ahead; one barrier per step.
#include "pn2_common.h"
#include "mlp_common.h"
#include "x3_common.h"

namespace {

struct EvalArgs {
  const float *xyz;       // (B, N, 3)
  const float *new_xyz;   // (B, m, 3)
  const int *idx;         // (B, m, ns)
  const float *feats;     // IN_SMALL: (B, N, C) rows | IN_LIFT: Pq (B N, c1)
  const float *Q;         // IN_LIFT: (B m, c1)
  const unsigned char *w0;        // IN_SMALL: first layer's fragments, c1 / 32 units
  const unsigned char *wstream;   // middle layer's units, then the last layer's
  const float *bias_mid;  // (c_mid)
  const float *bias_fin;  // (c_out)
  float *out;             // (B m, ldo), columns [0, c_out)
  unsigned *next_pass;    // [0] the next unclaimed pass (workgroups take passes as they get to run), [1] workgroups done; both zero between launches
  long long Mrows;        // B m ns  (IN_ROWS: rows of X)
  long long ncentres;     // B m
  int N, m, ns_shift, C, c_out, ldo, steps_fin, spp;
  // IN_ROWS (training GEMM, pn2_x3_gemm): A = pro(X) rows [Mrows][16 KA]
  const float *X, *X2;    // PRO_GY: X = g, X2 = y
  const float *p0, *p1, *p2;
  // epilogues of the training GEMM
  float *Y;               // EPI_STATS / EPI_MASK: [Mrows][c_out]
  double *stats;          // [2][c_out] (ACCUMULATES)
  const float *Yprev;     // EPI_MASK: [Mrows][c_out]
  const float *e_fin;     // EPI_MASK: [4][c_out] mean | rstd | scale | shift
  float *pmax;            // EPI_POOL: [Mrows / psz][c_out]
  int *parg;
  const float *sgn;       // EPI_POOL: [c_out] or NULL
};

enum { X3_IN_SMALL = 0, X3_IN_LIFT = 1, X3_IN_ROWS = 2 };
enum { X3_EPI_MAX = 0, X3_EPI_STATS = 1, X3_EPI_MASK = 2, X3_EPI_POOL = 3 };
enum { X3_PRO_NONE = 0, X3_PRO_BNRELU = 1, X3_PRO_GY = 2 };

constexpr int kRing = 2;

// WAVES = waves per workgroup (each owns 32 rows of a pass of 32 WAVES rows).  Two workgroups share a CU (2 x 55 KB of LDS)
// and run out of phase: one's gather / split phases under the other's matrix phases.  16 waves per CU at <= 128 registers
// (IN_SMALL instances, WAVES = 8), 8 waves at <= 256 (IN_LIFT: 96 registers of operand fragments per layer, WAVES = 4).
// IN_ROWS turns the same kernel into the TRAINING GEMM of a shared-MLP layer (pn2_x3_gemm): the input stage reads plain rows
// X [M][K] with the prologue of pn2_mlp_gemm applied before the split, there is no middle layer, and the last layer's
// epilogue is one of pn2_mlp_gemm's: store + column sums (EPI_STATS), ReLU mask + BatchNorm-backward sums (EPI_MASK), or
// group maxima + arg-max without a store (EPI_POOL = pn2_mlp_gemm_pool).
template <int IN, int KA, int KB, int WAVES, int PRO = 0, int EPI = 0>      // KA = c1 / 16; KB = c_mid / 16 (0: no middle layer)
__global__ __launch_bounds__(64 * WAVES, WAVES / 2) void sa_eval_kernel(const EvalArgs a) {
  constexpr int THREADS = 64 * WAVES, PASS = 32 * WAVES;
  constexpr bool SMALL = IN == X3_IN_SMALL, ROWS = IN == X3_IN_ROWS;
  constexpr bool SUMS = EPI != X3_EPI_MAX;
  constexpr int KF = KB ? KB : KA;                      // chunks of the last layer's input
  constexpr int TPS_MID = kX3SlotUnits / KA;            // tiles per ring slot
  constexpr int TPS_FIN = kX3SlotUnits / KF;
  constexpr int STEPS_MID = KB ? (KB / 2) * KA / kX3SlotUnits : 0;
  constexpr int W0_BYTES = SMALL ? (KA / 2) * kX3UnitBytes : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *ring = lds;
  unsigned char *w0s = lds + kRing * kX3SlotBytes;
  float *bmid = reinterpret_cast<float *>(w0s + W0_BYTES);     // permuted: [(tile, half)][16]
  float *bfin = bmid + (KB ? KB * 16 : 16);
  float *prm = bfin + (ROWS ? 0 : a.c_out);                    // IN_ROWS: [3][16 KA] prologue constants (no bias table then)
  float *csum = prm + (ROWS ? 3 * 16 * KA : 0);                // EPI with sums: [2][c_out] fp32 partial column sums of this workgroup
  // EPI_MAX, neighbourhoods of 64 .. 32 WAVES rows: the waves of a centre meet here (two buffers by step parity)
  float *spart = csum + (SUMS ? 2 * a.c_out : 0);              // [2][WAVES][TPS_FIN][32]
  const int wpc = 1 << (a.ns_shift > 5 ? a.ns_shift - 5 : 0);  // waves per centre
  const bool meet = EPI == X3_EPI_MAX && a.ns_shift > 5 && wpc <= WAVES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const long long npass = (a.Mrows + PASS - 1) / PASS;
  // Passes are CLAIMED, not dealt: under a co-running kernel (the sampling chain of the next batch on its side stream holds
  // 64-128 CUs) part of the grid starts late, and a static share per workgroup made the kernel as slow as its last starter
  // (1.83 ms contended against 1.40 alone for the four levels of the headline shape).  Thread 0 claims one pass ahead; the id
  // travels through LDS behind the step barriers the pass has anyway.
  // (the first two passes of a workgroup are its static ones — a burst of 2 x grid same-address atomics at kernel start is
  // serialised by the L2: +35 us on the 1024-pass level; the counter counts the claims BEHIND those 2 x grid passes)
  __shared__ unsigned s_claim[2];
  const unsigned claim0 = 2u * gridDim.x;

  // resident tables
  if (SMALL)
    for (int i = tid; i < W0_BYTES / 16; i += THREADS)
      reinterpret_cast<x3_u32x4 *>(w0s)[i] = reinterpret_cast<const x3_u32x4 *>(a.w0)[i];
  if (KB)
    for (int i = tid; i < KB * 16; i += THREADS) {
      const int r = i & 15, hh = (i >> 4) & 1, T = i >> 5;
      bmid[i] = a.bias_mid[32 * T + (r & 3) + 8 * (r >> 2) + 4 * hh];
    }
  if (!ROWS)
    for (int i = tid; i < a.c_out; i += THREADS) bfin[i] = a.bias_fin ? a.bias_fin[i] : 0.f;
  if (ROWS && PRO != X3_PRO_NONE)
    for (int i = tid; i < 16 * KA; i += THREADS) {
      prm[i] = a.p0[i];
      prm[16 * KA + i] = a.p1[i];
      prm[32 * KA + i] = PRO == X3_PRO_GY ? a.p2[i] : 0.f;
    }
  if (SUMS)
    for (int i = tid; i < 2 * a.c_out; i += THREADS) csum[i] = 0.f;

  // weight stream: global step g reads slot (g mod spp) of the stream into ring slot (g mod 2), one step ahead
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned ring_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ring);
  auto step_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA pieces have landed
    __syncthreads();
  };
  int dma_j = 0, dma_slot = 0;
  auto issue_dma = [&]() {
    // global_load_lds_dwordx4 through inline asm: issued by the builtin, hipcc counts the DMA into lgkmcnt as well and every
    // ds_read of the matrix loops then waits lgkmcnt(0) (measured on the ISA: 65 such waits against fine-grained lgkmcnt(3-5)
    // without it).  The asm form is invisible to the compiler's counters: its completion is awaited by the explicit
    // s_waitcnt vmcnt(0) in front of every step barrier (step_barrier below).
    const unsigned char *src = a.wstream + (size_t)dma_j * kX3SlotBytes + lane * 16;
    const unsigned dst = ring_base + dma_slot * kX3SlotBytes;          // wave-uniform LDS byte address
#pragma unroll
    for (int i = 0; i < kX3SlotBytes / 1024 / WAVES; ++i) {
      const int piece = wave_u + WAVES * i;
      const unsigned char *g = src + piece * 1024;
      const unsigned l = dst + piece * 1024;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(g), "s"(l) : "memory");
    }
    dma_j = dma_j + 1 == a.spp ? 0 : dma_j + 1;
    dma_slot = dma_slot + 1 == kRing ? 0 : dma_slot + 1;
  };
  issue_dma();
  step_barrier();

  int slot = 0;                                          // ring slot of the step being computed
  // neighbourhood index of this lane's row, loaded one pass ahead (the gathers depend on it: one L2 round trip less per pass)
  auto load_idx = [&](long long pass) {
    const long long row = pass * PASS + wave * 32 + l32;
    return (!ROWS && a.idx && pass < npass && row < a.Mrows) ? a.idx[row] : 0;
  };
  long long cur = blockIdx.x, nxt = (long long)blockIdx.x + gridDim.x;
  int parity = 0;
  int p_next = load_idx(cur);
  int fstep = 0;       // last-layer steps so far, over all passes: its parity picks the meeting buffer (consecutive steps alternate
                       // even across a pass boundary with an odd number of steps per pass)
  x3_frag actA[KA];
  x3_frag actB[KB ? KB : 1];

  auto load_w = [&](const unsigned char *base, int unit) {
    x3_frag f;
#pragma unroll
    for (int s = 0; s < 3; ++s)
      f.p[s] = *reinterpret_cast<const x3_u32x4 *>(base + unit * kX3UnitBytes + s * 1024 + lane * 16);
    return f;
  };
  // a transposed accumulator tile -> the two operand fragments it holds (ReLU, split)
  auto acc_to_frags = [&](const x3_f32x16 &acc, x3_frag &f0, x3_frag &f1) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      x3_split_pair(fmaxf(acc[2 * d], 0.f), fmaxf(acc[2 * d + 1], 0.f), f0, d);
      x3_split_pair(fmaxf(acc[8 + 2 * d], 0.f), fmaxf(acc[8 + 2 * d + 1], 0.f), f1, d);
    }
  };

  while (cur < npass) {
    const long long row0 = cur * PASS + wave * 32;     // wave-uniform
    if (tid == 0) s_claim[parity] = claim0 + atomicAdd(a.next_pass, 1u);      // the pass after next (read behind this pass's barriers)
    // ------------------------------------------------------------------ input stage
    {
      const long long row = row0 + l32;
      const bool valid = row < a.Mrows;
      int b = 0;
      long long centre = 0;
      const int p = p_next;
      p_next = load_idx(nxt);
      if (valid) {
        centre = row >> a.ns_shift;
        b = (int)(centre / a.m);
      }
      const size_t pt = (size_t)b * a.N + p;
      if (ROWS) {
        // plain rows with pn2_mlp_gemm's prologue: PRO_BNRELU relu(x p0[k] + p1[k]) = ReLU(BatchNorm(y_{l-1})),
        // PRO_GY p0[k] g + p1[k] y + p2[k] = dL/dy_l from dL/dz_l; rows past M stay zero (they must not reach the sums)
        // (straight-line loads from a clamped row: a branch around them would collect every chunk's loads in one block)
        const long long rsafe = valid ? row : a.Mrows - 1;
        const float *xr = a.X + (size_t)rsafe * (16 * KA) + 8 * h;
        const float *yr = PRO == X3_PRO_GY ? a.X2 + (size_t)rsafe * (16 * KA) + 8 * h : xr;
#pragma unroll
        for (int c = 0; c < KA; ++c) {
          float4 x0 = *reinterpret_cast<const float4 *>(xr + 16 * c);
          float4 x1 = *reinterpret_cast<const float4 *>(xr + 16 * c + 4);
          float4 y0 = x0, y1 = x1;
          if (PRO == X3_PRO_GY) {
            y0 = *reinterpret_cast<const float4 *>(yr + 16 * c);
            y1 = *reinterpret_cast<const float4 *>(yr + 16 * c + 4);
          }
          float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
          if (PRO == X3_PRO_NONE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = valid ? v[i] : 0.f;
          }
          if (PRO != X3_PRO_NONE) {
            const float4 *q0 = reinterpret_cast<const float4 *>(prm + 16 * c + 8 * h);
            const float4 *q1 = reinterpret_cast<const float4 *>(prm + 16 * KA + 16 * c + 8 * h);
            const float4 *q2 = reinterpret_cast<const float4 *>(prm + 32 * KA + 16 * c + 8 * h);
            const float4 a0 = q0[0], a1 = q0[1], b0 = q1[0], b1 = q1[1];
            const float s0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const float s1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            if (PRO == X3_PRO_BNRELU) {
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] = valid ? fmaxf(__fmaf_rn(v[i], s0[i], s1[i]), 0.f) : 0.f;
            } else {
              const float4 c0 = q2[0], c1 = q2[1];
              const float s2[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
              const float w[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] = valid ? __fmaf_rn(s0[i], v[i], __fmaf_rn(s1[i], w[i], s2[i])) : 0.f;
            }
          }
          x3_split8(v, actA[c]);
          // PRO_GY reads two matrices: all of a 128-wide row's loads in flight at once (128 registers) next to the 96 of the
          // fragments spills; in halves it fits
          if (PRO == X3_PRO_GY && KA == 8 && c == 3) __builtin_amdgcn_sched_barrier(0);
        }
      } else if (SMALL) {
        const int K0 = 3 + a.C;                           // bias column
        float v[8];
        if (a.X) {
          // pn2_x3_gemm_first: the grouped rows are given ([rel xyz | features] x K0 <= 8 columns, (M, K0) row-major)
          const float *xr = a.X + (size_t)(valid ? row : 0) * K0;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int col = 8 * h + i;
            v[i] = !valid ? 0.f : (col < K0 ? xr[col < K0 ? col : 0] : (col == K0 ? 1.f : 0.f));
          }
        } else {
          const float *fx = a.xyz + pt * 3, *cx = a.new_xyz + (size_t)centre * 3, *ff = a.feats + pt * a.C;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int col = 8 * h + i;
            float val = 0.f;
            if (valid) {
              if (col < 3) val = fx[col] - cx[col];         // the reference's in-place subtraction (OPS/pointnet2_utils.py:321)
              else if (col < K0) val = ff[col - 3];
              else if (col == K0) val = 1.f;
            }
            v[i] = val;
          }
        }
        x3_frag xin;
        x3_split8(v, xin);
#pragma unroll
        for (int T = 0; T < KA / 2; ++T) {
          x3_f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          const x3_frag w = load_w(w0s, T);
          x3_mma(w, xin, acc);                            // transposed: lane = row, registers = channels
          acc_to_frags(acc, actA[2 * T], actA[2 * T + 1]);
        }
      } else {
        const float *pq = a.feats + pt * (size_t)(KA * 16) + 8 * h;
        const float *qq = a.Q + (size_t)centre * (KA * 16) + 8 * h;
#pragma unroll
        for (int c = 0; c < KA; ++c) {
          float4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0, q0 = x0, q1 = x0;
          if (valid) {
            x0 = *reinterpret_cast<const float4 *>(pq + 16 * c);
            x1 = *reinterpret_cast<const float4 *>(pq + 16 * c + 4);
            q0 = *reinterpret_cast<const float4 *>(qq + 16 * c);
            q1 = *reinterpret_cast<const float4 *>(qq + 16 * c + 4);
          }
          const float v[8] = {fmaxf(x0.x - q0.x, 0.f), fmaxf(x0.y - q0.y, 0.f), fmaxf(x0.z - q0.z, 0.f), fmaxf(x0.w - q0.w, 0.f),
                              fmaxf(x1.x - q1.x, 0.f), fmaxf(x1.y - q1.y, 0.f), fmaxf(x1.z - q1.z, 0.f), fmaxf(x1.w - q1.w, 0.f)};
          x3_split8(v, actA[c]);
        }
      }
    }
    // ------------------------------------------------------------------ middle layer (transposed), compile-time tiles
#pragma unroll
    for (int jm = 0; jm < STEPS_MID; ++jm) {
      issue_dma();
      const unsigned char *sb = ring + slot * kX3SlotBytes;
#pragma unroll
      for (int tl = 0; tl < TPS_MID; ++tl) {
        const int T = jm * TPS_MID + tl;
        x3_f32x16 acc;
        const float4 *bt = reinterpret_cast<const float4 *>(bmid + (T * 2 + h) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 bv = bt[q];
          acc[4 * q] = bv.x; acc[4 * q + 1] = bv.y; acc[4 * q + 2] = bv.z; acc[4 * q + 3] = bv.w;
        }
        // weight fragments one chunk ahead of the matrix instructions that read them
        x3_frag wq[2];
        wq[0] = load_w(sb, tl * KA);
        __builtin_amdgcn_sched_barrier(0);      // (the bias reads above must not be counted into the loop's ds_read groups)
#pragma unroll
        for (int c = 0; c < KA; ++c) {
          if (c + 1 < KA) wq[(c + 1) & 1] = load_w(sb, tl * KA + c + 1);
          x3_mma(wq[c & 1], actA[c], acc);
          // pin the order (hipcc otherwise sinks every ds_read to its first use and waits lgkmcnt(0) three times per chunk)
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        }
        acc_to_frags(acc, actB[KB ? 2 * T : 0], actB[KB ? 2 * T + 1 : 0]);
      }
      step_barrier();
      slot = slot + 1 == kRing ? 0 : slot + 1;
    }
    // ------------------------------------------------------------------ last layer + maximum over each centre's rows
    for (int jf = 0; jf < a.steps_fin; ++jf) {
      issue_dma();
      const unsigned char *sb = ring + slot * kX3SlotBytes;
#pragma unroll
      for (int tl = 0; tl < TPS_FIN; ++tl) {
        const int t = jf * TPS_FIN + tl;
        // the last layer's bias is per channel = per LANE here: max_rows(x + b) = max_rows(x) + b, so it is added once behind
        // the maximum (EPI_MAX) instead of initialising 16 accumulator registers per tile with it; the accumulators start at the
        // matrix instruction's inline zero
        const float bv = (ROWS || EPI == X3_EPI_MAX) ? 0.f : bfin[32 * t + l32];
        const float bmax = (!ROWS && EPI == X3_EPI_MAX) ? bfin[32 * t + l32] : 0.f;
        x3_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bv;
        // EPI_MASK: the previous layer's raw output at this tile's positions and its BatchNorm constants, ahead of the matrix loop
        float yp[EPI == X3_EPI_MASK ? 16 : 1];
        float es = 0.f, eh = 0.f, em = 0.f, er = 0.f;
        if (EPI == X3_EPI_MASK) {
          const int col = 32 * t + l32;
          em = a.e_fin[col]; er = a.e_fin[a.c_out + col]; es = a.e_fin[2 * a.c_out + col]; eh = a.e_fin[3 * a.c_out + col];
          // buffer loads: wave-uniform base (the wave's first row) and row offsets, one lane-constant column offset; rows past M
          // are out of range and read 0
          const long long left = a.Mrows - row0;
          const rsrc_t rsp = make_rsrc(a.Yprev + (size_t)(left > 0 ? row0 : 0) * a.c_out, (left > 0 ? left : 1) * a.c_out * 4);
          const int voff = left > 0 ? (4 * h * a.c_out + col) * 4 : kOobOffset;
#pragma unroll
          for (int r = 0; r < 16; ++r) yp[EPI == X3_EPI_MASK ? r : 0] = bload(rsp, voff, ((r & 3) + 8 * (r >> 2)) * a.c_out * 4);
        }
        x3_frag wq[2];
        wq[0] = load_w(sb, tl * KF);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < KF; ++c) {
          if (c + 1 < KF) wq[(c + 1) & 1] = load_w(sb, tl * KF + c + 1);
          if (KB) x3_mma(actB[KB ? c : 0], wq[c & 1], acc);
          else x3_mma(actA[c < KA ? c : 0], wq[c & 1], acc);
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        }
        if (EPI == X3_EPI_STATS || EPI == X3_EPI_MASK) {
          // lane = column 32 t + l32, register r = row 4h + (r & 3) + 8 (r >> 2) of the wave's 32: one 128-byte row segment per
          // half-wave and register; column sums over the lane's 16 rows, the other half's added, one LDS atomic per column
          const int col = 32 * t + l32;
          float s1 = 0.f, s2 = 0.f;
          const long long left = a.Mrows - row0;
          const rsrc_t rsy = make_rsrc(a.Y + (size_t)(left > 0 ? row0 : 0) * a.c_out, (left > 0 ? left : 1) * a.c_out * 4);
          const int voff = left > 0 ? (4 * h * a.c_out + col) * 4 : kOobOffset;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[r];
            if (EPI == X3_EPI_MASK) {
              const float y = yp[EPI == X3_EPI_MASK ? r : 0];
              v = (__fmaf_rn(y, es, eh) > 0.f) ? v : 0.f;
              s1 += v;
              s2 = __fmaf_rn(v, (y - em) * er, s2);
            } else {
              s1 += v;
              s2 = __fmaf_rn(v, v, s2);
            }
            bstore(v, rsy, voff, ((r & 3) + 8 * (r >> 2)) * a.c_out * 4);     // rows past M: out of range, dropped
          }
          s1 += __shfl_xor(s1, 32);
          s2 += __shfl_xor(s2, 32);
          if (h == 0 && a.stats) {
            atomicAdd(&csum[col], s1);
            atomicAdd(&csum[a.c_out + col], s2);
          }
          continue;
        }
        if (EPI == X3_EPI_POOL) {
          // pn2_mlp_gemm_pool's epilogue: column sums, and per partial group of psz = min(ns, 32) rows the maximum of every
          // column with its row inside the partial group (first one among equals); nothing of the output is stored
          const int col = 32 * t + l32;
          float s1 = 0.f, s2 = 0.f, bst[2];
          int bi[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            bst[q] = acc[8 * q];
            bi[q] = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const float v = acc[8 * q + r];
              if (r > 0 && v > bst[q]) { bst[q] = v; bi[q] = r; }
              s1 += v;
              s2 = __fmaf_rn(v, v, s2);
            }
          }
          s1 += __shfl_xor(s1, 32);
          s2 += __shfl_xor(s2, 32);
          if (h == 0 && a.stats) {
            atomicAdd(&csum[col], s1);
            atomicAdd(&csum[a.c_out + col], s2);
          }
          float b2[2];
          int rw[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {                    // rows of the two halves interleave in blocks of four
            const int row = (bi[q] & 3) + 8 * (bi[q] >> 2) + 4 * h;       // row inside the 16-row sub-group + 8 q .. (r = 8 q + bi)
            const float ob = __shfl_xor(bst[q], 32);
            const int orow = __shfl_xor(row, 32);
            const bool take = ob > bst[q] || (ob == bst[q] && orow < row);
            b2[q] = take ? ob : bst[q];
            rw[q] = take ? orow : row;
          }
          const long long npart = a.Mrows >> (a.ns_shift == 4 ? 4 : 5);
          if (a.ns_shift == 4) {
            const long long part = (row0 >> 4) + h;
            if (part < npart) {
              a.pmax[(size_t)part * a.c_out + col] = h ? b2[1] : b2[0];
              a.parg[(size_t)part * a.c_out + col] = h ? rw[1] : rw[0];
            }
          } else if (h == 0) {
            const long long part = row0 >> 5;
            const bool second = b2[1] > b2[0];
            if (part < npart) {
              a.pmax[(size_t)part * a.c_out + col] = second ? b2[1] : b2[0];
              a.parg[(size_t)part * a.c_out + col] = second ? 16 + rw[1] : rw[0];
            }
          }
          continue;
        }
        // lane = channel 32 t + l32; registers 0-7: rows 0-15 of the wave's 32 (this half's share), 8-15: rows 16-31
        float m0 = acc[0], m1 = acc[8];
#pragma unroll
        for (int r = 1; r < 8; ++r) { m0 = fmaxf(m0, acc[r]); m1 = fmaxf(m1, acc[8 + r]); }
        m0 = fmaxf(m0, __shfl_xor(m0, 32)) + bmax;
        m1 = fmaxf(m1, __shfl_xor(m1, 32)) + bmax;
        const int col = 32 * t + l32;
        if (a.ns_shift == 4) {                             // two centres per wave: lower half stores the first, upper the second
          const long long centre = (row0 >> 4) + h;
          if (centre < a.ncentres) a.out[(size_t)centre * a.ldo + col] = fmaxf(h ? m1 : m0, 0.f);
        } else {
          const long long centre = row0 >> a.ns_shift;
          const float v = fmaxf(fmaxf(m0, m1), 0.f);
          if (meet) {
            if (h == 0) spart[(((fstep & 1) * WAVES + wave) * TPS_FIN + tl) * 32 + l32] = v;      // combined behind the step barrier
          } else if (h == 0 && centre < a.ncentres) {
            float *o = a.out + (size_t)centre * a.ldo + col;
            if (a.ns_shift == 5) *o = v;
            else atomicMax(reinterpret_cast<int *>(o), __builtin_bit_cast(int, v));   // v >= 0: integer order = float order; out zero-filled
          }
        }
      }
      step_barrier();
      slot = slot + 1 == kRing ? 0 : slot + 1;
      if (meet && (wave & (wpc - 1)) == 0 && h == 0) {
        // the first wave of every centre: maximum over the centre's waves, one plain 128-byte store per tile — no zero fill,
        // no atomics (they doubled the level's HBM traffic: 193 MB measured against 89 MB algorithmic at the headline's SA1)
        const long long centre = row0 >> a.ns_shift;
        if (centre < a.ncentres) {
#pragma unroll
          for (int tl = 0; tl < TPS_FIN; ++tl) {
            float v = 0.f;
            for (int w = 0; w < wpc; ++w) v = fmaxf(v, spart[(((fstep & 1) * WAVES + wave + w) * TPS_FIN + tl) * 32 + l32]);
            a.out[(size_t)centre * a.ldo + 32 * (jf * TPS_FIN + tl) + l32] = v;
          }
        }
      }
      ++fstep;
    }
    cur = nxt;
    nxt = s_claim[parity];
    parity ^= 1;
  }
  // the last workgroup to leave re-arms the counters for the stream's next launch (no memset launch per call: it cost the
  // small levels 20-35 us); everybody's claims are behind them when they get here
  if (tid == 0) {
    const unsigned done = atomicAdd(a.next_pass + 1, 1u);
    if (done == gridDim.x - 1) {
      a.next_pass[0] = 0u;
      a.next_pass[1] = 0u;
    }
  }
  if (SUMS && a.stats) {
    // (the last step_barrier made every wave's LDS atomics visible)
    for (int i = tid; i < a.c_out; i += THREADS) {
      const float sg = (EPI == X3_EPI_POOL && a.sgn) ? a.sgn[i] : 1.f;
      atomicAdd(a.stats + i, (double)(csum[i] * sg));
      atomicAdd(a.stats + a.c_out + i, (double)csum[a.c_out + i]);
    }
  }
}

// W [N][ldw] fp32 (columns [0, K) used) -> (N / 32) (K / 16) units, unit (t, c) at index t (K / 16) + c:
// [piece][lane] 16 bytes = the 8 bf16 of weight row 32 t + (lane & 31), contraction indices kmap(c, lane >> 5, 0..7)
__global__ __launch_bounds__(256) void x3_pack_kernel(int N, int K, int ldw, int perm, const float *__restrict__ W,
                                                      x3_u32x4 *__restrict__ out) {
  const int cpt = K / 16;
  const int total = (N / 32) * cpt * 64;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int lane = e & 63, unit = e >> 6, c = unit % cpt, t = unit / cpt;
    const float *row = W + (size_t)(32 * t + (lane & 31)) * ldw;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = row[x3_kmap(c, lane >> 5, i, perm != 0)];
    x3_frag f;
    x3_split8(v, f);
#pragma unroll
    for (int s = 0; s < 3; ++s) out[(size_t)unit * 192 + s * 64 + lane] = f.p[s];
  }
}

// The FIRST layer of a training stack with its BatchNorm folded in: M0 (N0, 16) = [diag(scale) W0 | shift | 0] -> N0 / 32 units
// (one chunk, natural contraction order), the resident first-layer table of the IN_SMALL input stage.
__global__ __launch_bounds__(64) void x3_pack_first_kernel(int N0, int K0, const float *__restrict__ W0,
                                                           const float *__restrict__ scale, const float *__restrict__ shift,
                                                           x3_u32x4 *__restrict__ out) {
  const int t = blockIdx.x, lane = threadIdx.x, n = 32 * t + (lane & 31), h = lane >> 5;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int col = 8 * h + i;
    v[i] = col < K0 ? W0[(size_t)n * K0 + col] * scale[n] : (col == K0 ? shift[n] : 0.f);
  }
  x3_frag f;
  x3_split8(v, f);
#pragma unroll
  for (int s = 0; s < 3; ++s) out[(size_t)t * 192 + s * 64 + lane] = f.p[s];
}

template <int IN, int KA, int KB, int WAVES, int PRO = 0, int EPI = 0>
int launch_eval(const EvalArgs &a, hipStream_t stream) {
  constexpr int W0_BYTES = IN == 0 ? (KA / 2) * kX3UnitBytes : 0;
  const size_t lds = (size_t)kRing * kX3SlotBytes + W0_BYTES +
                     ((KB ? KB * 16 : 16) + (IN == X3_IN_ROWS ? 3 * 16 * KA : a.c_out) + (EPI != X3_EPI_MAX ? 2 * a.c_out : 0) +
                      (EPI == X3_EPI_MAX ? 2 * WAVES * (kX3SlotUnits / (KB ? KB : KA)) * 32 : 0)) * sizeof(float);
  static const bool ok = hipFuncSetAttribute((const void *)sa_eval_kernel<IN, KA, KB, WAVES, PRO, EPI>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
  if (!ok || lds > 80 * 1024) return PN2_ELAUNCH;
  const long long npass = (a.Mrows + 32 * WAVES - 1) / (32 * WAVES);
  if (npass >= 0xffffffffll - 2048) return PN2_EINVAL;                 // (32-bit pass counter, every workgroup over-claims two)
  const unsigned grid = (unsigned)(npass < 512 ? npass : 512);         // two workgroups per CU
  hipLaunchKernelGGL((sa_eval_kernel<IN, KA, KB, WAVES, PRO, EPI>), dim3(grid), dim3(64 * WAVES), lds, stream, a);
  return pn2_check_launch();
}

}  // namespace

extern "C" size_t pn2_x3_weight_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || (N & 31) || (K & 15)) return 0;
  // rounded up to whole ring slots: the kernels' weight stream is read slot by slot
  const size_t units = (size_t)(N / 32) * (K / 16);
  return (units + kX3SlotUnits - 1) / kX3SlotUnits * kX3SlotBytes;
}

extern "C" int pn2_x3_pack_weight(int N, int K, int ldw, int perm, const float *W, void *frags, void *stream) {
  if (N <= 0 || K <= 0 || (N & 31) || (K & 15) || ldw < K) return PN2_EINVAL;
  if (!W || !frags) return PN2_ENULL;
  if ((((size_t)frags) & 15) != 0) return PN2_EINVAL;
  const int total = (N / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(x3_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, K, ldw, perm, W,
                     (x3_u32x4 *)frags);
  return pn2_check_launch();
}

// Shapes the fused eval level covers: neighbourhoods of 16 / 32 / 64 k rows, a first layer of <= 12 feature columns
// (mode 0) or a lifted first layer (mode 1), layer widths 64 or 128 in front of the last layer, any last width that is a
// multiple of 32 and fills whole ring slots.
extern "C" int pn2_sa_eval_x3_supported(int mode, int ns, int C, int c1, int c_mid, int c_out) {
  if (mode != 0 && mode != 1) return 0;
  if (ns < 16 || (ns & (ns - 1)) != 0) return 0;
  if (mode == 0 && (C < 0 || C > 12)) return 0;
  const bool inst = (mode == 0 && c1 == 64 && (c_mid == 0 || c_mid == 64)) || (mode == 1 && c1 == 128 && (c_mid == 0 || c_mid == 128));
  if (!inst) return 0;
  const int kf = (c_mid ? c_mid : c1) / 16;
  if (c_out <= 0 || (c_out & 31) || ((c_out / 32) * kf) % kX3SlotUnits != 0 || c_out > 4096) return 0;
  return 1;
}

extern "C" int pn2_sa_eval_x3(int mode, int B, int N, int m, int ns, int C, const float *xyz, const float *new_xyz,
                              const int *idx, const float *feats, const float *Q, int c1, const void *w0_frags, int c_mid,
                              const void *wstream, const float *bias_mid, int c_out, const float *bias_fin, float *out, int ldo,
                              void *workspace, void *stream) {
  if (B < 0 || N <= 0 || m <= 0 || ldo < c_out) return PN2_EINVAL;
  if (!pn2_sa_eval_x3_supported(mode, ns, C, c1, c_mid, c_out)) return PN2_EINVAL;
  if (B == 0) return PN2_OK;
  if (!new_xyz || !idx || !feats || !wstream || !bias_fin || !out || (mode == 0 && (!xyz || !w0_frags)) || (mode == 1 && !Q) ||
      (c_mid && !bias_mid) || !workspace)
    return PN2_ENULL;
  EvalArgs a{};
  a.next_pass = (unsigned *)workspace;
  a.xyz = xyz; a.new_xyz = new_xyz; a.idx = idx; a.feats = feats; a.Q = Q;
  a.w0 = (const unsigned char *)w0_frags; a.wstream = (const unsigned char *)wstream;
  a.bias_mid = bias_mid; a.bias_fin = bias_fin; a.out = out;
  a.ncentres = (long long)B * m;
  a.Mrows = a.ncentres * ns;
  a.N = N; a.m = m; a.C = C; a.c_out = c_out; a.ldo = ldo;
  int sh = 0;
  while ((1 << sh) < ns) ++sh;
  a.ns_shift = sh;
  const int kf = (c_mid ? c_mid : c1) / 16;
  a.steps_fin = (c_out / 32) * kf / kX3SlotUnits;
  const int steps_mid = c_mid ? (c_mid / 32) * (c1 / 16) / kX3SlotUnits : 0;
  a.spp = steps_mid + a.steps_fin;
  hipStream_t st = (hipStream_t)stream;
  const int waves = mode == 0 ? 8 : 4;
  if (ns > 32 * waves) {
    // more waves per centre than a workgroup has (group-all sized neighbourhoods): they meet in an integer atomic maximum over
    // the zero-filled result; up to 32 x waves rows the waves of a centre meet in LDS and the result is stored once
    if (hipMemset2DAsync(out, (size_t)ldo * 4, 0, (size_t)c_out * 4, (size_t)a.ncentres, st) != hipSuccess) return PN2_ELAUNCH;
  }
  if (mode == 0 && c_mid == 64) return launch_eval<0, 4, 4, 8>(a, st);
  if (mode == 0) return launch_eval<0, 4, 0, 8>(a, st);
  if (c_mid == 128) return launch_eval<1, 8, 8, 4>(a, st);
  return launch_eval<1, 8, 0, 4>(a, st);
}

// ---------------------------------------------------------------------------------------------- training GEMM on the f32x3 product
// Y[M][N] = pro(X)[M][K] W[N][K]^T with pn2_mlp_gemm's prologues / epilogues (and pn2_mlp_gemm_pool's as epi 3), W given as
// the fragments of pn2_x3_pack_weight(N, K, perm 0).  K in {64, 128}, N a multiple of 32 that fills whole ring slots.
extern "C" int pn2_x3_gemm_supported(int K, int N, int pro, int epi, int ns) {
  if (K != 64 && K != 128) return 0;
  if (N <= 0 || (N & 31) || ((N / 32) * (K / 16)) % kX3SlotUnits != 0 || N > 4096) return 0;
  const bool combo = (pro == X3_PRO_BNRELU && (epi == X3_EPI_STATS || epi == X3_EPI_POOL)) || (pro == X3_PRO_GY && epi == X3_EPI_MASK) ||
                     (pro == X3_PRO_NONE && epi == X3_EPI_STATS);
  if (!combo) return 0;
  if (epi == X3_EPI_POOL && (ns < 16 || (ns & (ns - 1)) != 0)) return 0;
  return 1;
}

extern "C" int pn2_x3_gemm(long long M, int K, int N, int pro, int epi, const float *X, const float *X2, const float *p0,
                           const float *p1, const float *p2, const void *wfrags, float *Y, double *stats, const float *Yprev,
                           const float *e_fin, float *pmax, int *parg, const float *sgn, int ns, void *workspace, void *stream) {
  if (M < 0 || !pn2_x3_gemm_supported(K, N, pro, epi, ns)) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!X || !wfrags || (pro != X3_PRO_NONE && (!p0 || !p1)) || (pro == X3_PRO_GY && (!X2 || !p2)) ||
      (epi != X3_EPI_POOL && !Y) || (epi == X3_EPI_MASK && (!Yprev || !e_fin)) || (epi == X3_EPI_POOL && (!pmax || !parg)) || !workspace)
    return PN2_ENULL;
  if (epi == X3_EPI_POOL && M % (ns < 32 ? ns : 32) != 0) return PN2_EINVAL;
  EvalArgs a{};
  a.next_pass = (unsigned *)workspace;
  a.X = X; a.X2 = X2; a.p0 = p0; a.p1 = p1; a.p2 = p2;
  a.wstream = (const unsigned char *)wfrags;
  a.Y = Y; a.stats = stats; a.Yprev = Yprev; a.e_fin = e_fin; a.pmax = pmax; a.parg = parg; a.sgn = sgn;
  a.Mrows = M; a.c_out = N;
  int sh = 0;
  while ((1 << sh) < ns) ++sh;
  a.ns_shift = epi == X3_EPI_POOL ? sh : 5;
  a.steps_fin = (N / 32) * (K / 16) / kX3SlotUnits;
  a.spp = a.steps_fin;
  hipStream_t st = (hipStream_t)stream;
#define PN2_X3G(KA_, W_, PRO_, EPI_) return launch_eval<X3_IN_ROWS, KA_, 0, W_, PRO_, EPI_>(a, st)
  if (K == 64) {
    // 64-wide rows: <= 128 registers -> workgroups of 8 waves, 16 waves per CU (the split and the epilogue are vector work:
    // more waves keep the matrix pipe fed); the two-matrix input-gradient form needs more registers
    if (pro == X3_PRO_BNRELU && epi == X3_EPI_STATS) PN2_X3G(4, 8, X3_PRO_BNRELU, X3_EPI_STATS);
    if (pro == X3_PRO_BNRELU && epi == X3_EPI_POOL) PN2_X3G(4, 8, X3_PRO_BNRELU, X3_EPI_POOL);
    if (pro == X3_PRO_GY) PN2_X3G(4, 4, X3_PRO_GY, X3_EPI_MASK);
    PN2_X3G(4, 8, X3_PRO_NONE, X3_EPI_STATS);
  }
  if (pro == X3_PRO_BNRELU && epi == X3_EPI_STATS) PN2_X3G(8, 4, X3_PRO_BNRELU, X3_EPI_STATS);
  if (pro == X3_PRO_BNRELU && epi == X3_EPI_POOL) PN2_X3G(8, 4, X3_PRO_BNRELU, X3_EPI_POOL);
  if (pro == X3_PRO_GY) PN2_X3G(8, 4, X3_PRO_GY, X3_EPI_MASK);
  PN2_X3G(8, 4, X3_PRO_NONE, X3_EPI_STATS);
#undef PN2_X3G
}

// pn2_mlp_gemm_first on the f32x3 product: Y (M, N) = relu(bn_0(X0 W0^T)) W^T + the column sums of Y, Y^2 — the second layer of a
// training stack with the first one re-formed from the grouped input rows (y_0 is never stored).  bn_0 is folded into the first
// layer's fragments (`w0_frags` of pn2_x3_pack_first: scale_0 W0 | shift_0 in the bias column), the products run through the
// eval level's chain (IN_SMALL input stage on stored rows, no middle layer, store + sums epilogue).
extern "C" int pn2_x3_gemm_first_supported(int K0, int K, int N) {
  if (K0 < 1 || K0 > 8 || K != 64) return 0;
  if (N <= 0 || (N & 31) || ((N / 32) * (K / 16)) % kX3SlotUnits != 0 || N > 4096) return 0;
  return 1;
}

extern "C" int pn2_x3_pack_first(int N0, int K0, const float *W0, const float *scale, const float *shift, void *frags,
                                 void *stream) {
  if (N0 <= 0 || (N0 & 31) || K0 < 1 || K0 > 8) return PN2_EINVAL;
  if (!W0 || !scale || !shift || !frags) return PN2_ENULL;
  if ((((size_t)frags) & 15) != 0) return PN2_EINVAL;
  hipLaunchKernelGGL(x3_pack_first_kernel, dim3(N0 / 32), dim3(64), 0, (hipStream_t)stream, N0, K0, W0, scale, shift,
                     (x3_u32x4 *)frags);
  return pn2_check_launch();
}

extern "C" int pn2_x3_gemm_first(long long M, int K0, int K, int N, const float *X0, const void *w0_frags, const void *wfrags,
                                 float *Y, double *stats, void *workspace, void *stream) {
  if (M < 0 || !pn2_x3_gemm_first_supported(K0, K, N)) return PN2_EINVAL;
  if (M == 0) return PN2_OK;
  if (!X0 || !w0_frags || !wfrags || !Y || !workspace) return PN2_ENULL;
  EvalArgs a{};
  a.next_pass = (unsigned *)workspace;
  a.X = X0;
  a.w0 = (const unsigned char *)w0_frags; a.wstream = (const unsigned char *)wfrags;
  a.Y = Y; a.stats = stats;
  a.Mrows = M; a.ncentres = M; a.m = 1; a.N = 1; a.C = K0 - 3; a.c_out = N; a.ldo = N;
  a.ns_shift = 5;
  a.steps_fin = (N / 32) * (K / 16) / kX3SlotUnits;
  a.spp = a.steps_fin;
  return launch_eval<X3_IN_SMALL, 4, 0, 8, X3_PRO_NONE, X3_EPI_STATS>(a, (hipStream_t)stream);
}
