// group_lift.hip — the FIRST layer of an SA stack applied BEFORE the grouping (round 4).
//
// The reference stages an SA level as  ball query -> group_points (gather the C feature channels of every neighbour:
// EXT/src/group_points_gpu.cu:8-28) -> cat with the relative coordinates (OPS/pointnet2_utils.py:317-328) -> Conv2d 1x1
// over the 3 + C channels of every (centre, sample) position (OPS/pointnet2_modules.py:9-19, 58-63).  A 1x1 convolution is
// linear and the gather only copies, so
//
//     y0[b, j, s, :] = W [rel | f[b, idx]] = Wx rel[b, j, s] + (Wf f[b, :])[idx[b, j, s]]
//
// — the feature part of the product can be taken once per POINT (B N rows) instead of once per (centre, sample) position
// (B m ns rows: 16x as many at the headline's SA2), and the grouped (B, m, ns, 3 + C) tensor never exists:
//   forward : P = f Wf^T (a B N x C x N0 GEMM), then THIS kernel gathers rows of P through idx, adds the three coordinate
//             terms and accumulates the BatchNorm column sums of y0  (pn2_group_lift_rows);
//   backward: dL/dy0 = c1 g + c2 y0 + c3 per row (BatchNorm backward of layer 0, from the masked gradient g the layer above
//             left).  Its sum over the rows that gathered point (b, n) — S[b, n, :], walked through the inverse
//             neighbourhood index of group_csr.hip — is all the feature side needs:  dL/df = S Wf,  dWf = S^T f (GEMMs over
//             B N rows); S and the coordinate columns  dWx = sum_r dL/dy0[r] rel[r]^T  follow from four sums over the rows
//             that do not involve the constants (pn2_group_lift_rows_grad: Sg, SR, Dg, RR, see the kernel).  The atomic scatter of group_points_grad_kernel (EXT/src/group_points_gpu.cu:44-75),
//             the M x (3 + C) input gradient and the M-row weight-gradient GEMM disappear.
// Arithmetic: y0 = fma(Wx2, rz, fma(Wx1, ry, fma(Wx0, rx, P))) in fp32 — the same real number as the reference's 3 + C term
// dot product, summed in another order (tested at 1e-4 against the oracle like every MLP kernel).
#include "pn2_common.h"

namespace {
constexpr int kLiftBlock = 256;

typedef float f4v __attribute__((ext_vector_type(4)));
struct Row3f { float a, b, c; };   // 12 bytes at 4-byte alignment: one global_load_dwordx3

struct LiftFwdArgs {
  const float *xyz;      // (B, N, 3)
  const float *new_xyz;  // (B, m, 3)
  const int *idx;        // (B, m, ns)
  const float *P;        // (B N, N0)   per-point products f Wf^T
  const float *Wx;       // (N0, 3)     coordinate columns of the first conv
  void *Y;               // (B m ns, N0) fp32, or bf16 (the mixed-precision stacks: the next layer's bf16 MFMA reads it)
  double *stats;         // (2, N0) += column sums of y0, y0^2 (or null)
  int N, m, ns, N0, normalize;
  float radius;
  int centres, chunks;   // B m; centres dealt to workgroups in chunks of 4 (one per wave)
  // segment table (grid.y = scan): ROW offsets of the scans (device), rows per cloud; every scan runs as its OWN call would —
  // its clouds, its grid, its block of the statistics — so its sums are bit for bit those of a single-scan launch
  const long long *seg;
  int per;
  long long row0;        // first row of the call in Y (0 without a table)
};

// LPR lanes per row (N0 = 4 LPR columns, 16 bytes per lane), R = 64 / LPR rows per wave instruction.  A wave owns one
// centre at a time: lane s first resolves slot s (index + relative coordinates), the rows then stream R at a time with
// four 16-byte gathers of P in flight per lane.  Workgroups walk the centres in the XCD-aware order of the ball query
// (XCD x takes the x-th eighth: a cloud's P rows — 1 MB at SA2 — stay in one L2).
// bf16 rows: round to nearest even, like the epilogue of mlp_gemm_bf16 (whose statistics are those of the ROUNDED values too)
__device__ __forceinline__ unsigned lift_bf_pack(float lo, float hi) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
}
__device__ __forceinline__ float lift_bf_round(float f) { return (float)(__bf16)f; }

// grid of a forward call over `centres` centres: 8 XCD shares x up to 256 workgroups, at least four chunks (16 centres) per
// workgroup (every workgroup ends with 2 N0 fp64 atomics onto the same addresses)
__host__ __device__ inline int lift_fwd_grid(long long centres) {
  const int chunks = (int)((centres + 3) / 4);
  const int per = (chunks + 7) / 8;
  int nwg = (per + 3) / 4;
  nwg = nwg < 1 ? 1 : (nwg > 256 ? 256 : nwg);
  return nwg * 8;
}

template <int R, bool BF>
__global__ __launch_bounds__(kLiftBlock) void group_lift_rows_kernel(const LiftFwdArgs a_in) {
  LiftFwdArgs a = a_in;
  unsigned gdim = gridDim.x;
  if (a.seg) {
    const long long r0 = a.seg[blockIdx.y], r1 = a.seg[blockIdx.y + 1];
    const long long c0 = r0 / a.per, nc = (r1 - r0) / a.per;
    a.xyz += (size_t)c0 * a.N * 3; a.new_xyz += (size_t)c0 * a.m * 3; a.idx += r0; a.P += (size_t)c0 * a.N * a.N0;
    if (a.stats) a.stats += (size_t)blockIdx.y * 2 * a.N0;
    a.row0 = r0;
    a.centres = (int)(nc * a.m);
    a.chunks = (a.centres + 3) / 4;
    gdim = (unsigned)lift_fwd_grid(a.centres);
    if (blockIdx.x >= gdim || nc == 0) return;                     // (workgroup-uniform)
  }
  constexpr int LPR = 64 / R;
  __shared__ float red[2][4][4 * LPR];
  const int lane = pn2_lane(), wv = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int N0 = a.N0, ns = a.ns;
  const bool live = 4 * l < N0;
  f4v wx0 = f4v{0.f, 0.f, 0.f, 0.f}, wx1 = wx0, wx2 = wx0;
  if (live) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      wx0[c] = a.Wx[(size_t)(4 * l + c) * 3 + 0];
      wx1[c] = a.Wx[(size_t)(4 * l + c) * 3 + 1];
      wx2[c] = a.Wx[(size_t)(4 * l + c) * 3 + 2];
    }
  }
  f4v s1 = f4v{0.f, 0.f, 0.f, 0.f}, s2 = s1;
  const int per = (a.chunks + 7) >> 3;                       // chunks per XCD share
  const int nwg = gdim >> 3;                                 // workgroups per XCD share
  for (int ck = (int)(blockIdx.x >> 3); ck < per; ck += nwg) {
    const int chunk = (int)(blockIdx.x & 7) * per + ck;
    const int g = chunk * 4 + wv;
    if (chunk >= a.chunks || g >= a.centres) continue;       // wave-uniform
    const int b = (int)((unsigned)g / (unsigned)a.m);
    const float qx = a.new_xyz[(size_t)g * 3 + 0], qy = a.new_xyz[(size_t)g * 3 + 1], qz = a.new_xyz[(size_t)g * 3 + 2];
    const float *X = a.xyz + (size_t)b * a.N * 3;
    const float *Pb = a.P + (size_t)b * a.N * N0;
    const int *row_idx = a.idx + (size_t)g * ns;
    float *Yg = (float *)a.Y + ((size_t)a.row0 + (size_t)g * ns) * N0;                  // (fp32 rows)
    uint2 *Yh = (uint2 *)a.Y + ((size_t)a.row0 + (size_t)g * ns) * (N0 >> 2);           // (bf16 rows: four columns = 8 bytes per lane)
    for (int s0 = 0; s0 < ns; s0 += 64) {
      const int cnt = ns - s0 < 64 ? ns - s0 : 64;
      int mi = 0;
      float rx = 0.f, ry = 0.f, rz = 0.f;
      if (lane < cnt) {
        mi = row_idx[s0 + lane];
        const Row3f p = *reinterpret_cast<const Row3f *>(X + (size_t)mi * 3);
        rx = p.a - qx; ry = p.b - qy; rz = p.c - qz;
        if (a.normalize) { rx = __fdiv_rn(rx, a.radius); ry = __fdiv_rn(ry, a.radius); rz = __fdiv_rn(rz, a.radius); }
      }
      for (int t = 0; t * R < cnt; t += 4) {
        f4v v[4];
        float ex[4], ey[4], ez[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = (t + u) * R + sub;
          const int src = i & 63;
          const int pi = __shfl(mi, src);
          ex[u] = __shfl(rx, src); ey[u] = __shfl(ry, src); ez[u] = __shfl(rz, src);
          ok[u] = i < cnt && live;
          v[u] = f4v{0.f, 0.f, 0.f, 0.f};
          if (ok[u]) v[u] = *reinterpret_cast<const f4v *>(Pb + (size_t)pi * N0 + 4 * l);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!ok[u]) continue;
          f4v y;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            y[c] = __fmaf_rn(wx2[c], ez[u], __fmaf_rn(wx1[c], ey[u], __fmaf_rn(wx0[c], ex[u], v[u][c])));
            if constexpr (BF) y[c] = lift_bf_round(y[c]);
            s1[c] = __fadd_rn(s1[c], y[c]);
            s2[c] = __fmaf_rn(y[c], y[c], s2[c]);
          }
          if constexpr (BF)
            Yh[(size_t)(s0 + (t + u) * R + sub) * (N0 >> 2) + l] = uint2{lift_bf_pack(y[0], y[1]), lift_bf_pack(y[2], y[3])};
          else
            *reinterpret_cast<f4v *>(Yg + (size_t)(s0 + (t + u) * R + sub) * N0 + 4 * l) = y;
        }
      }
    }
  }
  if (!a.stats) return;
  // column sums: sub-waves -> wave -> workgroup (LDS) -> one fp64 atomic per column and workgroup
#pragma unroll
  for (int d = 32; d >= LPR; d >>= 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      s1[c] = __fadd_rn(s1[c], __shfl_xor(s1[c], d));
      s2[c] = __fadd_rn(s2[c], __shfl_xor(s2[c], d));
    }
  }
  if (sub == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { red[0][wv][4 * l + c] = s1[c]; red[1][wv][4 * l + c] = s2[c]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * N0; c += kLiftBlock) {
    const int which = c >= N0, col = which ? c - N0 : c;
    const double t = (double)red[which][0][col] + (double)red[which][1][col] + (double)red[which][2][col] +
                     (double)red[which][3][col];
    atomicAdd(a.stats + (size_t)which * N0 + col, t);
  }
}

struct LiftBwdArgs {
  const float *xyz;      // (B, N, 3)
  const float *new_xyz;  // (B m, 3)
  const void *G;         // (M, N0)  masked gradient dL/dz0 the layer above left: fp32, or bf16 (mixed-precision stacks)
  const float *P;        // (B N, N0) per-point products of the forward
  const float *Wx;       // (N0, 3)
  const float *consts;   // (3, N0)  c1 | c2 | c3 of BatchNorm's backward
  const int *ptr;        // (B N + 1)
  const int *refs;       // (M)      row ids sorted by (point, row)
  float *S;              // (B N, N0)  sum_r dL/dy0[r] over the rows that gathered the point
  float *part;           // (workgroups of both passes, 3 N0 + 16) per-workgroup partials of dWx | RR (summed by lift_reduce_kernel:
                         //          same-address atomics from 3 000 workgroups finishing together cost more than the walk)
  int *heavy;            // [0] = number of heavy points (zero on entry), [1 ..] their ids
  int ns, N0, normalize;
  float radius;
  unsigned npoints;
  // segment table (grid.y = scan; see LiftFwdArgs): per-point tensors, the constants, the heavy list and the partials are
  // the scan's own slices; G, new_xyz and refs stay the batch's (row ids index them in place)
  const long long *seg;
  int per, N;
  unsigned heavy_stride, part_stride;      // ints / floats per scan
};

constexpr int kLiftHeavy = 192;     // a point gathered by more rows than this is walked by kLiftSplit waves in a second pass
constexpr int kLiftSplit = 16;

// wave-wide sum on the DPP network; every lane returns the total
__device__ __forceinline__ float lift_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
  return pn2_readlane_f32(v, 63);
}

// The backward behind BatchNorm is AFFINE in the rows: dL/dy0[r] = c1 g[r] + c2 y0[r] + c3 with y0[r] = P[point] + Wx rel[r],
// so the per-channel constants stay OUT of the walk over the rows (the first version carried them, Wx and P through the
// loop: 118 VGPRs, four waves per SIMD).  Per point p with rows r:
//     S[p]  = sum_r dL/dy0[r]          = c1 Sg + n (c2 P[p] + c3) + c2 Wx SR         Sg = sum_r g[r], SR = sum_r rel[r], n = rows
//     dWx   = sum_r dL/dy0[r] rel[r]^T = c1 Dg + sum_p (c2 P[p] + c3) SR[p]^T + c2 Wx RR,   Dg = sum_r g[r] rel[r]^T, RR = sum_r rel rel^T
// The loop sums g, rel and g rel^T; the constants enter once per point (S, the second term of dWx) and once per wave (c1 Dg);
// RR is returned and the caller adds c2 Wx RR.  Both formulas are linear in (Sg, n, SR): a slice of a point's rows can be
// finished on its own and ADDED — which is how heavy points are handled: ball queries return the first nsample hits in
// index order, so with large radii (SA3 / SA4: r 0.8 / 1.2) the lowest indices of a cloud sit in nearly every ball — 500
// rows on one point, none on most — and a wave per point leaves the kernel waiting for a few long serial walks (0.25 ms
// for 77 MB).  Points with more than kLiftHeavy rows are only listed by the first pass (their S row zeroed) and walked by
// kLiftSplit waves each in the second.
template <int R, bool BF>
__device__ __forceinline__ void lift_walk(const LiftBwdArgs &a, unsigned n, int p0, int p1, bool add, f4v &dx, f4v &dy, f4v &dz,
                                          f4v &ex_, f4v &ey_, f4v &ez_, float (&rr)[6]) {
  constexpr int LPR = 64 / R;
  const int lane = pn2_lane();
  const int sub = lane / LPR, l = lane % LPR;
  const int N0 = a.N0, ns = a.ns;
  const bool live = 4 * l < N0;
  const float px = a.xyz[(size_t)n * 3 + 0], py = a.xyz[(size_t)n * 3 + 1], pz = a.xyz[(size_t)n * 3 + 2];
  f4v acc = f4v{0.f, 0.f, 0.f, 0.f};
  float srx = 0.f, sry = 0.f, srz = 0.f;                       // per-lane partials of SR
  for (int base = p0; base < p1; base += 64) {
    const int cnt = p1 - base < 64 ? p1 - base : 64;
    int myref = 0;
    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (lane < cnt) {
      myref = a.refs[base + lane];
      const Row3f q = *reinterpret_cast<const Row3f *>(a.new_xyz + (size_t)((unsigned)myref / (unsigned)ns) * 3);
      rx = px - q.a; ry = py - q.b; rz = pz - q.c;              // the forward's relative coordinates, bit for bit
      if (a.normalize) { rx = __fdiv_rn(rx, a.radius); ry = __fdiv_rn(ry, a.radius); rz = __fdiv_rn(rz, a.radius); }
      srx += rx; sry += ry; srz += rz;
      rr[0] = __fmaf_rn(rx, rx, rr[0]); rr[1] = __fmaf_rn(rx, ry, rr[1]); rr[2] = __fmaf_rn(rx, rz, rr[2]);
      rr[3] = __fmaf_rn(ry, ry, rr[3]); rr[4] = __fmaf_rn(ry, rz, rr[4]); rr[5] = __fmaf_rn(rz, rz, rr[5]);
    }
    for (int t = 0; t * R < cnt; t += 4) {
      f4v g[4];
      float ex[4], ey[4], ez[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = (t + u) * R + sub;
        const int src = i & 63;
        const int r = __shfl(myref, src);
        ex[u] = __shfl(rx, src); ey[u] = __shfl(ry, src); ez[u] = __shfl(rz, src);   // (0 beyond cnt: those rows add nothing)
        g[u] = f4v{0.f, 0.f, 0.f, 0.f};
        if (i < cnt && live) {
          if constexpr (BF) {
            const uint2 w = reinterpret_cast<const uint2 *>(a.G)[(size_t)r * (N0 >> 2) + l];
            g[u] = f4v{__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xFFFF0000u),
                       __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xFFFF0000u)};
          } else {
            g[u] = *reinterpret_cast<const f4v *>((const float *)a.G + (size_t)r * N0 + 4 * l);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[c] = __fadd_rn(acc[c], g[u][c]);
          dx[c] = __fmaf_rn(g[u][c], ex[u], dx[c]);
          dy[c] = __fmaf_rn(g[u][c], ey[u], dy[c]);
          dz[c] = __fmaf_rn(g[u][c], ez[u], dz[c]);
        }
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= LPR; d >>= 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = __fadd_rn(acc[c], __shfl_xor(acc[c], d));
  }
  const float tx = lift_wave_sum(srx), ty = lift_wave_sum(sry), tz = lift_wave_sum(srz);
  const float cnt_f = (float)(p1 - p0);
  // the constants enter here, once per point (short live ranges: the loop above does not carry them)
  if (sub == 0 && live) {
    const f4v c1 = *reinterpret_cast<const f4v *>(a.consts + 4 * l);
    const f4v c2 = *reinterpret_cast<const f4v *>(a.consts + N0 + 4 * l);
    const f4v c3 = *reinterpret_cast<const f4v *>(a.consts + 2 * N0 + 4 * l);
    const f4v pp = *reinterpret_cast<const f4v *>(a.P + (size_t)n * N0 + 4 * l);
    f4v sv;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float A = __fmaf_rn(c2[c], pp[c], c3[c]);                        // c2 P[p] + c3
      const float w0 = a.Wx[(size_t)(4 * l + c) * 3 + 0], w1 = a.Wx[(size_t)(4 * l + c) * 3 + 1],
                  w2 = a.Wx[(size_t)(4 * l + c) * 3 + 2];
      const float wsr = __fmaf_rn(w2, tz, __fmaf_rn(w1, ty, w0 * tx));       // Wx SR
      sv[c] = __fmaf_rn(c1[c], acc[c], __fmaf_rn(cnt_f, A, c2[c] * wsr));
      ex_[c] = __fmaf_rn(A, tx, ex_[c]); ey_[c] = __fmaf_rn(A, ty, ey_[c]); ez_[c] = __fmaf_rn(A, tz, ez_[c]);
    }
    float *dst = a.S + (size_t)n * N0 + 4 * l;
    if (add) {
#pragma unroll
      for (int c = 0; c < 4; ++c) atomicAdd(dst + c, sv[c]);
    } else {
      *reinterpret_cast<f4v *>(dst) = sv;
    }
  }
}

// flush of a wave's coordinate-column accumulators: c1 Dg + E and RR, sub-waves -> waves through LDS -> this workgroup's row
// of the partials buffer
template <int R>
__device__ __forceinline__ void lift_flush(const LiftBwdArgs &a, unsigned row, f4v dx, f4v dy, f4v dz, f4v ex_, f4v ey_, f4v ez_,
                                           const float (&rr)[6], float (*red)[4][256], float (*rrs)[16]) {
  constexpr int LPR = 64 / R;
  const int lane = pn2_lane(), wv = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int N0 = a.N0;
#pragma unroll
  for (int d = 32; d >= LPR; d >>= 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      dx[c] = __fadd_rn(dx[c], __shfl_xor(dx[c], d));
      dy[c] = __fadd_rn(dy[c], __shfl_xor(dy[c], d));
      dz[c] = __fadd_rn(dz[c], __shfl_xor(dz[c], d));
    }
  }
  if (sub == 0) {
    f4v c1 = f4v{0.f, 0.f, 0.f, 0.f};
    if (4 * l < N0) c1 = *reinterpret_cast<const f4v *>(a.consts + 4 * l);
#pragma unroll
    for (int c = 0; c < 4; ++c) {      // (E lives in the sub == 0 lanes only)
      red[0][wv][4 * l + c] = __fmaf_rn(c1[c], dx[c], ex_[c]);
      red[1][wv][4 * l + c] = __fmaf_rn(c1[c], dy[c], ey_[c]);
      red[2][wv][4 * l + c] = __fmaf_rn(c1[c], dz[c], ez_[c]);
    }
  }
  {
    const float t0 = lift_wave_sum(rr[0]), t1 = lift_wave_sum(rr[1]), t2 = lift_wave_sum(rr[2]);
    const float t3 = lift_wave_sum(rr[3]), t4 = lift_wave_sum(rr[4]), t5 = lift_wave_sum(rr[5]);
    // the full symmetric 3 x 3 matrix, row-major: xx xy xz | xy yy yz | xz yz zz
    if (lane < 9)
      rrs[wv][lane] = (lane == 0) ? t0 : (lane == 1 || lane == 3) ? t1 : (lane == 2 || lane == 6) ? t2 : (lane == 4) ? t3
                      : (lane == 5 || lane == 7) ? t4 : t5;
  }
  __syncthreads();
  float *prow = a.part + (size_t)row * (3 * N0 + 16);
  for (int e = threadIdx.x; e < 3 * N0; e += kLiftBlock) {
    const int d = e / N0, col = e % N0;
    prow[col * 3 + d] = (red[d][0][col] + red[d][1][col]) + (red[d][2][col] + red[d][3][col]);
  }
  if (threadIdx.x < 9) prow[3 * N0 + threadIdx.x] = (rrs[0][threadIdx.x] + rrs[1][threadIdx.x]) + (rrs[2][threadIdx.x] + rrs[3][threadIdx.x]);
}

// workgroups of pass 1 over `npoints` points
__host__ __device__ inline unsigned lift_grid1(size_t npoints) {
  const unsigned waves_wanted = 256u * 32u;
  unsigned grid = (unsigned)((npoints < waves_wanted ? npoints : waves_wanted) + 3) / 4;
  return grid == 0 ? 1 : grid;
}

// column sums of the partials: out[0 : 3 N0] = dWx (without c2 Wx RR), out[3 N0 : 3 N0 + 9] = RR; fixed order
__global__ __launch_bounds__(256) void lift_reduce_kernel(const float *__restrict__ part, int rows, int width, int pitch,
                                                         float *__restrict__ out, const long long *__restrict__ seg, int per,
                                                         int N, unsigned part_stride, int tail_rows) {
  __shared__ float red[256];
  if (seg) {                                   // grid.y = scan: its partial rows (pass 1 of ITS point count + pass 2), its output row
    const long long nc = (seg[blockIdx.y + 1] - seg[blockIdx.y]) / per;
    rows = nc ? (int)lift_grid1((size_t)nc * N) + tail_rows : 0;
    part += (size_t)blockIdx.y * part_stride;
    out += (size_t)blockIdx.y * width;
  }
  const int col = blockIdx.x;
  float t = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) t += part[(size_t)r * pitch + col];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0 && col < width) out[col] = red[0];
}

// a scan's view of the arguments (segment table); false: this workgroup has nothing to do.  `grid1`: pass-1 workgroups of
// the scan's own call (= first partial row of its pass 2)
__device__ __forceinline__ bool lift_bwd_scan(LiftBwdArgs &a, unsigned &grid1) {
  grid1 = lift_grid1(a.npoints);
  if (!a.seg) return true;
  const long long r0 = a.seg[blockIdx.y], r1 = a.seg[blockIdx.y + 1];
  const long long c0 = r0 / a.per, nc = (r1 - r0) / a.per;
  const size_t p0 = (size_t)c0 * a.N;
  a.xyz += p0 * 3; a.P += p0 * a.N0; a.S += p0 * a.N0; a.ptr += p0;
  a.consts += (size_t)blockIdx.y * 3 * a.N0;
  a.heavy += (size_t)blockIdx.y * a.heavy_stride;
  a.part += (size_t)blockIdx.y * a.part_stride;
  a.npoints = (unsigned)(nc * a.N);
  grid1 = lift_grid1(a.npoints);
  return nc != 0;
}

// pass 1: a wave per point (like group_rows_grad_csr_kernel); heavy points are listed, their S row zeroed
template <int R, bool BF>
__global__ __launch_bounds__(kLiftBlock) void group_lift_rows_grad_kernel(const LiftBwdArgs a_in) {
  LiftBwdArgs a = a_in;
  unsigned grid1;
  if (!lift_bwd_scan(a, grid1) || blockIdx.x >= grid1) return;
  const unsigned gdim = a.seg ? grid1 : gridDim.x;
  constexpr int LPR = 64 / R;
  __shared__ float red[3][4][256];
  __shared__ float rrs[4][16];
  const int lane = pn2_lane(), wv = threadIdx.x >> 6;
  f4v dx = f4v{0.f, 0.f, 0.f, 0.f}, dy = dx, dz = dx, ex_ = dx, ey_ = dx, ez_ = dx;
  float rr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const unsigned nwaves = gdim * (kLiftBlock / 64);
  for (unsigned n = __builtin_amdgcn_readfirstlane(blockIdx.x * (kLiftBlock / 64) + wv); n < a.npoints; n += nwaves) {
    const int p0 = a.ptr[n], p1 = a.ptr[n + 1];
    if (p1 - p0 > kLiftHeavy) {                                   // wave-uniform
      if (lane == 0) a.heavy[1 + atomicAdd(a.heavy, 1)] = (int)n;
      const int l = lane % LPR;
      if (lane < LPR && 4 * l < a.N0) *reinterpret_cast<f4v *>(a.S + (size_t)n * a.N0 + 4 * l) = f4v{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    lift_walk<R, BF>(a, n, p0, p1, false, dx, dy, dz, ex_, ey_, ez_, rr);
  }
  lift_flush<R>(a, blockIdx.x, dx, dy, dz, ex_, ey_, ez_, rr, red, rrs);
}

// pass 2: kLiftSplit waves per heavy point, each a contiguous slice of its rows, results added
template <int R, bool BF>
__global__ __launch_bounds__(kLiftBlock) void group_lift_rows_grad_heavy_kernel(const LiftBwdArgs a_in, unsigned row0) {
  LiftBwdArgs a = a_in;
  unsigned grid1;
  if (!lift_bwd_scan(a, grid1)) return;
  if (a.seg) row0 = grid1;
  __shared__ float red[3][4][256];
  __shared__ float rrs[4][16];
  const int wv = threadIdx.x >> 6;
  f4v dx = f4v{0.f, 0.f, 0.f, 0.f}, dy = dx, dz = dx, ex_ = dx, ey_ = dx, ez_ = dx;
  float rr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const unsigned nwaves = gridDim.x * (kLiftBlock / 64);
  const unsigned items = (unsigned)a.heavy[0] * kLiftSplit;
  for (unsigned w = __builtin_amdgcn_readfirstlane(blockIdx.x * (kLiftBlock / 64) + wv); w < items; w += nwaves) {
    const unsigned n = (unsigned)a.heavy[1 + w / kLiftSplit];
    const int k = (int)(w % kLiftSplit);
    const int p0 = a.ptr[n], p1 = a.ptr[n + 1];
    const int per = ((p1 - p0 + kLiftSplit - 1) / kLiftSplit + 63) & ~63;     // slices of whole 64-row batches
    const int q0 = p0 + k * per, q1 = q0 + per < p1 ? q0 + per : p1;
    if (q0 >= p1) continue;
    lift_walk<R, BF>(a, n, q0, q1, true, dx, dy, dz, ex_, ey_, ez_, rr);
  }
  lift_flush<R>(a, row0 + blockIdx.x, dx, dy, dz, ex_, ey_, ez_, rr, red, rrs);
}

bool lift_shape_ok(int N0) { return N0 >= 16 && N0 <= 256 && (N0 & 3) == 0; }
}  // namespace

extern "C" int pn2_group_lift_supported(int N0) { return lift_shape_ok(N0) ? 1 : 0; }

namespace {
int lift_rows_launch(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz, const float *new_xyz,
                     const int *idx, const float *P, const float *Wx, void *Y, bool bf, double *stats, void *stream,
                     const long long *seg = nullptr, int nseg = 0, int max_clouds = 0) {
  if (seg && (nseg < 1 || nseg > 65535 || max_clouds < 1 || max_clouds > B)) return PN2_EINVAL;
  if (B < 0 || N < 0 || m < 0 || ns < 0) return PN2_EINVAL;
  if (!lift_shape_ok(N0) || (normalize && !(radius > 0.f))) return PN2_EINVAL;
  const long long centres = (long long)B * m;
  if (centres == 0 || ns == 0) return PN2_OK;
  if (centres > 0x7fffffffLL - 64 || (long long)B * N >= 0x7fffffffLL) return PN2_EINVAL;
  if (!xyz || !new_xyz || !idx || !P || !Wx || !Y) return PN2_ENULL;
  if ((((uintptr_t)P) | ((uintptr_t)Y)) & 15) return PN2_EINVAL;
  LiftFwdArgs a{xyz, new_xyz, idx, P, Wx, Y, stats, N, m, ns, N0, normalize ? 1 : 0, radius, (int)centres,
                (int)((centres + 3) / 4), seg, m * ns, 0};
  // without a table one call; with one: grid.y = scan, grid.x = the grid of the LARGEST scan (a scan's surplus workgroups exit)
  const dim3 grid((unsigned)lift_fwd_grid(seg ? (long long)max_clouds * m : centres), (unsigned)(seg ? nseg : 1)), block(kLiftBlock);
  hipStream_t s = (hipStream_t)stream;
  if (bf) {
    if (N0 <= 64) hipLaunchKernelGGL((group_lift_rows_kernel<4, true>), grid, block, 0, s, a);
    else if (N0 <= 128) hipLaunchKernelGGL((group_lift_rows_kernel<2, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((group_lift_rows_kernel<1, true>), grid, block, 0, s, a);
  } else {
    if (N0 <= 64) hipLaunchKernelGGL((group_lift_rows_kernel<4, false>), grid, block, 0, s, a);
    else if (N0 <= 128) hipLaunchKernelGGL((group_lift_rows_kernel<2, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((group_lift_rows_kernel<1, false>), grid, block, 0, s, a);
  }
  return pn2_check_launch();
}
}  // namespace

extern "C" int pn2_group_lift_rows(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                                   const float *new_xyz, const int *idx, const float *P, const float *Wx, float *Y,
                                   double *stats, void *stream) {
  return lift_rows_launch(B, N, m, ns, N0, normalize, radius, xyz, new_xyz, idx, P, Wx, Y, false, stats, stream);
}

// ... with bf16 rows (the mixed-precision stacks): Y (B m ns, N0) bf16, rounded to nearest even; `stats` are the column sums of
// the ROUNDED values (the convention of pn2_mlp_gemm_bf16)
extern "C" int pn2_group_lift_rows_bf16(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                                        const float *new_xyz, const int *idx, const float *P, const float *Wx, void *Y,
                                        double *stats, void *stream) {
  return lift_rows_launch(B, N, m, ns, N0, normalize, radius, xyz, new_xyz, idx, P, Wx, Y, true, stats, stream);
}

namespace {
constexpr unsigned kLiftGrid2 = 512;      // workgroups of the heavy pass: 2048 waves = 128 heavy points at a time
size_t lift_heavy_bytes(int B, int m, int ns) {
  const size_t M = (size_t)B * m * ns;
  return ((M / kLiftHeavy + 2) * sizeof(int) + 255) & ~(size_t)255;     // a point is heavy with more than kLiftHeavy rows
}
}  // namespace

extern "C" size_t pn2_group_lift_rows_grad_workspace_bytes(int B, int N, int m, int ns, int N0) {
  if (B <= 0 || N <= 0 || m <= 0 || ns <= 0 || !lift_shape_ok(N0)) return 0;
  return lift_heavy_bytes(B, m, ns) + (size_t)(lift_grid1((size_t)B * N) + kLiftGrid2) * (3 * N0 + 16) * sizeof(float);
}

namespace {
// workspace of a segment-table call: nseg x (heavy list | partial rows) of the LARGEST scan
size_t lift_seg_heavy_bytes(int max_clouds, int m, int ns) { return lift_heavy_bytes(max_clouds, m, ns); }
size_t lift_seg_part_floats(int max_clouds, int N, int N0) {
  return (size_t)(lift_grid1((size_t)max_clouds * N) + kLiftGrid2) * (3 * N0 + 16);
}

int lift_rows_grad_launch(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                          const float *new_xyz, const void *G, bool bf, const float *P, const float *Wx, const float *consts,
                          const int *ptr, const int *refs, float *S, float *acc, void *workspace, size_t workspace_bytes,
                          void *stream, const long long *seg = nullptr, int nseg = 0, int max_clouds = 0) {
  if (seg) {
    if (nseg < 1 || nseg > 65535 || max_clouds < 1 || max_clouds > B || B < 0 || N <= 0 || m <= 0 || ns <= 0) return PN2_EINVAL;
    if (!lift_shape_ok(N0) || (normalize && !(radius > 0.f))) return PN2_EINVAL;
    if ((size_t)B * N >= 0x7fffffffull || (long long)B * m * ns >= 0x7fffffffLL) return PN2_EINVAL;
    if (!xyz || !new_xyz || !G || !P || !Wx || !consts || !ptr || !refs || !S || !acc || !workspace) return PN2_ENULL;
    if ((((uintptr_t)G) | ((uintptr_t)P) | ((uintptr_t)S) | ((uintptr_t)consts) | ((uintptr_t)workspace)) & 15) return PN2_EINVAL;
    const size_t hb = lift_seg_heavy_bytes(max_clouds, m, ns), pf = lift_seg_part_floats(max_clouds, N, N0);
    if (workspace_bytes < (size_t)nseg * (hb + pf * sizeof(float))) return PN2_ENOSPC;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, (size_t)nseg * hb, s) != hipSuccess) return PN2_ELAUNCH;      // every scan's heavy count
    float *part = (float *)((char *)workspace + (size_t)nseg * hb);
    LiftBwdArgs a{xyz, new_xyz, G, P, Wx, consts, ptr, refs, S, part, (int *)workspace, ns, N0, normalize ? 1 : 0, radius,
                  (unsigned)((size_t)max_clouds * N), seg, m * ns, N, (unsigned)(hb / sizeof(int)), (unsigned)pf};
    const unsigned grid = lift_grid1((size_t)max_clouds * N);
#define PN2_LIFT_SEG(RR_, BF_)                                                                                                       \
  do {                                                                                                                               \
    hipLaunchKernelGGL((group_lift_rows_grad_kernel<RR_, BF_>), dim3(grid, (unsigned)nseg), dim3(kLiftBlock), 0, s, a);              \
    hipLaunchKernelGGL((group_lift_rows_grad_heavy_kernel<RR_, BF_>), dim3(kLiftGrid2, (unsigned)nseg), dim3(kLiftBlock), 0, s, a, grid); \
  } while (0)
    if (bf) { if (N0 <= 64) PN2_LIFT_SEG(4, true); else if (N0 <= 128) PN2_LIFT_SEG(2, true); else PN2_LIFT_SEG(1, true); }
    else { if (N0 <= 64) PN2_LIFT_SEG(4, false); else if (N0 <= 128) PN2_LIFT_SEG(2, false); else PN2_LIFT_SEG(1, false); }
#undef PN2_LIFT_SEG
    hipLaunchKernelGGL(lift_reduce_kernel, dim3((unsigned)(3 * N0 + 9), (unsigned)nseg), dim3(256), 0, s, part, 0, 3 * N0 + 9,
                       3 * N0 + 16, acc, seg, m * ns, N, (unsigned)pf, (int)kLiftGrid2);
    return pn2_check_launch();
  }
  if (B < 0 || N < 0 || m < 0 || ns <= 0) return PN2_EINVAL;
  if (!lift_shape_ok(N0) || (normalize && !(radius > 0.f))) return PN2_EINVAL;
  const size_t npoints = (size_t)B * N;
  if (npoints == 0) return PN2_OK;
  if (npoints >= 0x7fffffffull || (long long)B * m * ns >= 0x7fffffffLL) return PN2_EINVAL;
  if (!xyz || !new_xyz || !G || !P || !Wx || !consts || !ptr || !refs || !S || !acc || !workspace) return PN2_ENULL;
  if ((((uintptr_t)G) | ((uintptr_t)P) | ((uintptr_t)S) | ((uintptr_t)consts) | ((uintptr_t)workspace)) & 15) return PN2_EINVAL;
  if (workspace_bytes < pn2_group_lift_rows_grad_workspace_bytes(B, N, m, ns, N0)) return PN2_ENOSPC;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(workspace, 0, sizeof(int), s) != hipSuccess) return PN2_ELAUNCH;
  float *part = (float *)((char *)workspace + lift_heavy_bytes(B, m, ns));
  LiftBwdArgs a{xyz, new_xyz, G, P, Wx, consts, ptr, refs, S, part, (int *)workspace, ns, N0, normalize ? 1 : 0, radius,
                (unsigned)npoints, nullptr, m * ns, N, 0u, 0u};
  const unsigned grid = lift_grid1(npoints);
#define PN2_LIFT(RR_, BF_)                                                                                                  \
  do {                                                                                                                      \
    hipLaunchKernelGGL((group_lift_rows_grad_kernel<RR_, BF_>), dim3(grid), dim3(kLiftBlock), 0, s, a);                     \
    hipLaunchKernelGGL((group_lift_rows_grad_heavy_kernel<RR_, BF_>), dim3(kLiftGrid2), dim3(kLiftBlock), 0, s, a, grid);   \
  } while (0)
  if (bf) { if (N0 <= 64) PN2_LIFT(4, true); else if (N0 <= 128) PN2_LIFT(2, true); else PN2_LIFT(1, true); }
  else { if (N0 <= 64) PN2_LIFT(4, false); else if (N0 <= 128) PN2_LIFT(2, false); else PN2_LIFT(1, false); }
#undef PN2_LIFT
  hipLaunchKernelGGL(lift_reduce_kernel, dim3((unsigned)(3 * N0 + 9)), dim3(256), 0, s, part, (int)(grid + kLiftGrid2),
                     3 * N0 + 9, 3 * N0 + 16, acc, (const long long *)nullptr, 1, 0, 0u, 0);
  return pn2_check_launch();
}
}  // namespace

extern "C" int pn2_group_lift_rows_grad(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                                        const float *new_xyz, const float *G, const float *P, const float *Wx,
                                        const float *consts, const int *ptr, const int *refs, float *S, float *acc,
                                        void *workspace, size_t workspace_bytes, void *stream) {
  return lift_rows_grad_launch(B, N, m, ns, N0, normalize, radius, xyz, new_xyz, G, false, P, Wx, consts, ptr, refs, S, acc,
                               workspace, workspace_bytes, stream);
}

// ... with the masked gradient G (M, N0) in bf16 (what pn2_mlp_bwd_bf16 / pn2_mlp_gemm_bf16 leave for the layer below);
// everything that is summed stays fp32
extern "C" int pn2_group_lift_rows_grad_bf16(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                                             const float *new_xyz, const void *G, const float *P, const float *Wx,
                                             const float *consts, const int *ptr, const int *refs, float *S, float *acc,
                                             void *workspace, size_t workspace_bytes, void *stream) {
  return lift_rows_grad_launch(B, N, m, ns, N0, normalize, radius, xyz, new_xyz, G, true, P, Wx, consts, ptr, refs, S, acc,
                               workspace, workspace_bytes, stream);
}

// ---- the S scans of a batch in ONE launch (segment-table stacks of the mixed-precision node, round 4) --------------------
// grid.y = scan: every scan runs exactly as its own single-scan call would — its clouds, the grid its centre / point count
// gives (surplus workgroups of shorter scans exit), its (2, N0) block of the statistics resp. its (3, N0) constants and
// (3 N0 + 9) accumulator row — so its sums are bit for bit those of a single-scan launch, at the launch count of one call.
// `seg`: (nseg + 1) ROW offsets of the scans (device, int64; multiples of m ns), `max_clouds`: clouds of the largest scan.
extern "C" int pn2_group_lift_rows_seg(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                                       const float *new_xyz, const int *idx, const float *P, const float *Wx, void *Y, int y_bf16,
                                       double *stats, const long long *seg, int nseg, int max_clouds, void *stream) {
  if (!seg) return PN2_ENULL;
  return lift_rows_launch(B, N, m, ns, N0, normalize, radius, xyz, new_xyz, idx, P, Wx, Y, y_bf16 != 0, stats, stream, seg, nseg,
                          max_clouds);
}

extern "C" size_t pn2_group_lift_rows_grad_seg_workspace_bytes(int nseg, int max_clouds, int N, int m, int ns, int N0) {
  if (nseg <= 0 || max_clouds <= 0 || N <= 0 || m <= 0 || ns <= 0 || !lift_shape_ok(N0)) return 0;
  return (size_t)nseg * (lift_seg_heavy_bytes(max_clouds, m, ns) + lift_seg_part_floats(max_clouds, N, N0) * sizeof(float));
}

// consts (nseg, 3, N0), acc (nseg, 3 N0 + 9); G, new_xyz, refs: the batch's tensors (row ids index them in place)
extern "C" int pn2_group_lift_rows_grad_seg(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                                            const float *new_xyz, const void *G, int g_bf16, const float *P, const float *Wx,
                                            const float *consts, const int *ptr, const int *refs, float *S_out, float *acc,
                                            const long long *seg, int nseg, int max_clouds, void *workspace,
                                            size_t workspace_bytes, void *stream) {
  if (!seg) return PN2_ENULL;
  return lift_rows_grad_launch(B, N, m, ns, N0, normalize, radius, xyz, new_xyz, G, g_bf16 != 0, P, Wx, consts, ptr, refs, S_out,
                               acc, workspace, workspace_bytes, stream, seg, nseg, max_clouds);
}

// ---- the lifted layer's weight in pieces / its gradient in one piece (round 5) ------------------------------------------------
// The first conv's weight W (N0, 3 + C) is used as Wx = W[:, :3] (coordinate columns), Wf = W[:, 3:] (per-point product) and
// Wf^T (input gradient): three strided torch copies per level and direction became ONE launch in the forward, reused by the
// backward.  The weight gradient dW = [dWx + c2 Wx RR | dWf] was a 3 x 3 vendor GEMM + addcmul + cat: one launch.
namespace {
__global__ __launch_bounds__(256) void lift_split_weight_kernel(int N0, int C, const float *__restrict__ W, float *__restrict__ Wx,
                                                               float *__restrict__ Wf, float *__restrict__ WfT) {
  const int K0 = C + 3;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < N0 * K0; e += gridDim.x * 256) {
    const int n = e / K0, k = e - n * K0;
    const float v = W[e];
    if (k < 3) Wx[n * 3 + k] = v;
    else { Wf[(size_t)n * C + (k - 3)] = v; WfT[(size_t)(k - 3) * N0 + n] = v; }
  }
}

__global__ __launch_bounds__(256) void lift_dw_assemble_kernel(int N0, int C, const float *__restrict__ acc,
                                                              const float *__restrict__ Wx, const float *__restrict__ c2,
                                                              const float *__restrict__ dWf, float *__restrict__ dW) {
  const int K0 = C + 3;
  const float *RR = acc + 3 * N0;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < N0 * K0; e += gridDim.x * 256) {
    const int n = e / K0, k = e - n * K0;
    float v;
    if (k < 3) {
      const float t = __fmaf_rn(Wx[n * 3 + 2], RR[6 + k], __fmaf_rn(Wx[n * 3 + 1], RR[3 + k], __fmul_rn(Wx[n * 3], RR[k])));
      v = __fmaf_rn(c2[n], t, acc[n * 3 + k]);
    } else {
      v = dWf[(size_t)n * C + (k - 3)];
    }
    dW[e] = v;
  }
}
}  // namespace

extern "C" int pn2_lift_split_weight(int N0, int C, const float *W, float *Wx, float *Wf, float *WfT, void *stream) {
  if (N0 <= 0 || C <= 0) return PN2_EINVAL;
  if (!W || !Wx || !Wf || !WfT) return PN2_ENULL;
  const int total = N0 * (C + 3);
  hipLaunchKernelGGL(lift_split_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N0, C, W, Wx,
                     Wf, WfT);
  return pn2_check_launch();
}

extern "C" int pn2_lift_dw_assemble(int N0, int C, const float *acc, const float *Wx, const float *c2, const float *dWf, float *dW,
                                    void *stream) {
  if (N0 <= 0 || C <= 0) return PN2_EINVAL;
  if (!acc || !Wx || !c2 || !dWf || !dW) return PN2_ENULL;
  const int total = N0 * (C + 3);
  hipLaunchKernelGGL(lift_dw_assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N0, C, acc, Wx,
                     c2, dWf, dW);
  return pn2_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// The lifted first layer WITHOUT its output tensor (round 5).  With the coordinate term split between the point and the
// centre,
//     y0[b, j, s, :] = Wx (x[idx] - c_j) / r + (Wf f)[idx] = Pq[b, idx[b, j, s]] - Q[b, j],
//     Pq[b, n] = Wf f[b, n] + Wx x[b, n] / r   (B N rows),      Q[b, j] = Wx c[b, j] / r   (B m rows),
// a row of y0 is ONE gathered row of Pq minus a per-centre row: cheap enough to re-form inside the consumers — the GEMM of the
// layer above (mlp_gemm.hip PRO_LIFT), its weight gradient (pn2_mlp_wgrad_lift) and the mask / BatchNorm-backward sums of
// its input gradient (EPI_MASKL) — so the (B m ns, N0) tensor (537 MB at the headline's SA2, written once and read three
// times) is never stored.  What is left of the forward here is the BatchNorm statistics of y0: pn2_group_lift_stats gathers
// like pn2_group_lift_rows but writes only the row ids (cloud offset included) the consumers index Pq with.
// y0 is the same real number as before, rounded differently (the coordinate products are subtracted after the multiplication,
// not before: |Wx x / r| eps ~ 1e-7 against 1e-4 asked of the features); the backward of the layer itself
// (pn2_group_lift_rows_grad) is linear in y0 and keeps its own form.
namespace {
__global__ __launch_bounds__(256) void lift_points_kernel(int npoints, int ncentres, int N0, int normalize, float radius,
                                                          const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                          const float *__restrict__ P, const float *__restrict__ Wx,
                                                          float *__restrict__ Pq, float *__restrict__ Q) {
  const int q4 = N0 >> 2;
  const long long total = (long long)(npoints + ncentres) * q4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int row = (int)(e / q4), l = (int)(e - (long long)row * q4);
    const bool pt = row < npoints;
    const float *src = pt ? xyz + (size_t)row * 3 : new_xyz + (size_t)(row - npoints) * 3;
    float x = src[0], y = src[1], z = src[2];
    if (normalize) { x = __fdiv_rn(x, radius); y = __fdiv_rn(y, radius); z = __fdiv_rn(z, radius); }
    f4v base = f4v{0.f, 0.f, 0.f, 0.f};
    if (pt) base = *reinterpret_cast<const f4v *>(P + (size_t)row * N0 + 4 * l);
    f4v o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float *w = Wx + (size_t)(4 * l + c) * 3;
      o[c] = __fmaf_rn(w[2], z, __fmaf_rn(w[1], y, __fmaf_rn(w[0], x, base[c])));
    }
    float *dst = pt ? Pq + (size_t)row * N0 + 4 * l : Q + (size_t)(row - npoints) * N0 + 4 * l;
    *reinterpret_cast<f4v *>(dst) = o;
  }
}

struct LiftStatArgs {
  const int *idx;        // (B, m, ns)
  const float *Pq;       // (B N, N0)
  const float *Q;        // (B m, N0)
  int *gidx;             // (B m ns)   b N + idx: the row of Pq every grouped row reads
  double *stats;         // (2, N0) += column sums of y0, y0^2
  int N, m, ns, N0, centres, chunks;
};

// the walk of group_lift_rows_kernel (a wave per centre, XCD-aware order), reading only
template <int R>
__global__ __launch_bounds__(kLiftBlock) void group_lift_stats_kernel(const LiftStatArgs a) {
  constexpr int LPR = 64 / R;
  __shared__ float red[2][4][4 * LPR];
  const int lane = pn2_lane(), wv = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int N0 = a.N0, ns = a.ns;
  const bool live = 4 * l < N0;
  f4v s1 = f4v{0.f, 0.f, 0.f, 0.f}, s2 = s1;
  const int per = (a.chunks + 7) >> 3;
  const int nwg = gridDim.x >> 3;
  for (int ck = (int)(blockIdx.x >> 3); ck < per; ck += nwg) {
    const int chunk = (int)(blockIdx.x & 7) * per + ck;
    const int g = chunk * 4 + wv;
    if (chunk >= a.chunks || g >= a.centres) continue;       // wave-uniform
    const int b = (int)((unsigned)g / (unsigned)a.m);
    const float *Pb = a.Pq + (size_t)b * a.N * N0;
    const int *row_idx = a.idx + (size_t)g * ns;
    f4v q = f4v{0.f, 0.f, 0.f, 0.f};
    if (live) q = *reinterpret_cast<const f4v *>(a.Q + (size_t)g * N0 + 4 * l);
    for (int s0 = 0; s0 < ns; s0 += 64) {
      const int cnt = ns - s0 < 64 ? ns - s0 : 64;
      int mi = 0;
      if (lane < cnt) {
        mi = row_idx[s0 + lane];
        a.gidx[(size_t)g * ns + s0 + lane] = b * a.N + mi;
      }
      for (int t = 0; t * R < cnt; t += 4) {
        f4v v[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = (t + u) * R + sub;
          const int pi = __shfl(mi, i & 63);
          ok[u] = i < cnt && live;
          v[u] = f4v{0.f, 0.f, 0.f, 0.f};
          if (ok[u]) v[u] = *reinterpret_cast<const f4v *>(Pb + (size_t)pi * N0 + 4 * l);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!ok[u]) continue;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float y = __fsub_rn(v[u][c], q[c]);
            s1[c] = __fadd_rn(s1[c], y);
            s2[c] = __fmaf_rn(y, y, s2[c]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= LPR; d >>= 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      s1[c] = __fadd_rn(s1[c], __shfl_xor(s1[c], d));
      s2[c] = __fadd_rn(s2[c], __shfl_xor(s2[c], d));
    }
  }
  if (sub == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { red[0][wv][4 * l + c] = s1[c]; red[1][wv][4 * l + c] = s2[c]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * N0; c += kLiftBlock) {
    const int which = c >= N0, col = which ? c - N0 : c;
    const double t = (double)red[which][0][col] + (double)red[which][1][col] + (double)red[which][2][col] +
                     (double)red[which][3][col];
    atomicAdd(a.stats + (size_t)which * N0 + col, t);
  }
}
}  // namespace

extern "C" int pn2_lift_points(int B, int N, int m, int N0, int normalize, float radius, const float *xyz, const float *new_xyz,
                               const float *P, const float *Wx, float *Pq, float *Q, void *stream) {
  if (B < 0 || N < 0 || m < 0 || !lift_shape_ok(N0) || (normalize && !(radius > 0.f))) return PN2_EINVAL;
  const long long np = (long long)B * N, nc = (long long)B * m;
  if (np + nc == 0) return PN2_OK;
  if (np + nc >= 0x7fffffffLL) return PN2_EINVAL;
  if (!xyz || !new_xyz || !P || !Wx || !Pq || !Q) return PN2_ENULL;
  if ((((uintptr_t)P) | ((uintptr_t)Pq) | ((uintptr_t)Q)) & 15) return PN2_EINVAL;
  const long long total = (np + nc) * (N0 >> 2);
  long long grid = (total + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(lift_points_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (int)np, (int)nc, N0,
                     normalize ? 1 : 0, radius, xyz, new_xyz, P, Wx, Pq, Q);
  return pn2_check_launch();
}

extern "C" int pn2_group_lift_stats(int B, int N, int m, int ns, int N0, const int *idx, const float *Pq, const float *Q,
                                    int *gidx, double *stats, void *stream) {
  if (B < 0 || N < 0 || m < 0 || ns < 0 || !lift_shape_ok(N0)) return PN2_EINVAL;
  const long long centres = (long long)B * m;
  if (centres == 0 || ns == 0) return PN2_OK;
  if (centres > 0x7fffffffLL - 64 || (long long)B * N >= 0x7fffffffLL || centres * ns >= 0x7fffffffLL) return PN2_EINVAL;
  if (!idx || !Pq || !Q || !gidx || !stats) return PN2_ENULL;
  if ((((uintptr_t)Pq) | ((uintptr_t)Q)) & 15) return PN2_EINVAL;
  const LiftStatArgs a{idx, Pq, Q, gidx, stats, N, m, ns, N0, (int)centres, (int)((centres + 3) / 4)};
  const dim3 grid((unsigned)lift_fwd_grid(centres)), block(kLiftBlock);
  hipStream_t s = (hipStream_t)stream;
  if (N0 <= 64) hipLaunchKernelGGL((group_lift_stats_kernel<4>), grid, block, 0, s, a);
  else if (N0 <= 128) hipLaunchKernelGGL((group_lift_stats_kernel<2>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((group_lift_stats_kernel<1>), grid, block, 0, s, a);
  return pn2_check_launch();
}
