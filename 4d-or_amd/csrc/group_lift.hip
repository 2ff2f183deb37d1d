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
//             B N rows), and the coordinate columns  dWx = sum_r dL/dy0[r] rel[r]^T  are accumulated in the same walk
//             (pn2_group_lift_rows_grad).  The atomic scatter of group_points_grad_kernel (EXT/src/group_points_gpu.cu:44-75),
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
  float *Y;              // (B m ns, N0)
  double *stats;         // (2, N0) += column sums of y0, y0^2 (or null)
  int N, m, ns, N0, normalize;
  float radius;
  int centres, chunks;   // B m; centres dealt to workgroups in chunks of 4 (one per wave)
};

// LPR lanes per row (N0 = 4 LPR columns, 16 bytes per lane), R = 64 / LPR rows per wave instruction.  A wave owns one
// centre at a time: lane s first resolves slot s (index + relative coordinates), the rows then stream R at a time with
// four 16-byte gathers of P in flight per lane.  Workgroups walk the centres in the XCD-aware order of the ball query
// (XCD x takes the x-th eighth: a cloud's P rows — 1 MB at SA2 — stay in one L2).
template <int R>
__global__ __launch_bounds__(kLiftBlock) void group_lift_rows_kernel(const LiftFwdArgs a) {
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
  const int nwg = gridDim.x >> 3;                            // workgroups per XCD share
  for (int ck = (int)(blockIdx.x >> 3); ck < per; ck += nwg) {
    const int chunk = (int)(blockIdx.x & 7) * per + ck;
    const int g = chunk * 4 + wv;
    if (chunk >= a.chunks || g >= a.centres) continue;       // wave-uniform
    const int b = (int)((unsigned)g / (unsigned)a.m);
    const float qx = a.new_xyz[(size_t)g * 3 + 0], qy = a.new_xyz[(size_t)g * 3 + 1], qz = a.new_xyz[(size_t)g * 3 + 2];
    const float *X = a.xyz + (size_t)b * a.N * 3;
    const float *Pb = a.P + (size_t)b * a.N * N0;
    const int *row_idx = a.idx + (size_t)g * ns;
    float *Yg = a.Y + (size_t)g * ns * N0;
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
            s1[c] = __fadd_rn(s1[c], y[c]);
            s2[c] = __fmaf_rn(y[c], y[c], s2[c]);
          }
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
  const float *G;        // (M, N0)  masked gradient dL/dz0 the layer above left
  const float *Y0;       // (M, N0)  raw first-layer output
  const float *consts;   // (3, N0)  c1 | c2 | c3 of BatchNorm's backward
  const int *ptr;        // (B N + 1)
  const int *refs;       // (M)      row ids sorted by (point, row)
  float *S;              // (B N, N0)
  float *dWx;            // (N0, 3)  += (fp32 atomics, once per workgroup)
  int ns, N0, normalize;
  float radius;
  unsigned npoints;
};

// A wave per point (like group_rows_grad_csr_kernel): R sub-waves walk the point's rows R at a time, two 16-byte loads
// (g, y0) per row and lane, four rows in flight.
template <int R>
__global__ __launch_bounds__(kLiftBlock) void group_lift_rows_grad_kernel(const LiftBwdArgs a) {
  constexpr int LPR = 64 / R;
  __shared__ float red[3][4][4 * LPR];
  const int lane = pn2_lane(), wv = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int N0 = a.N0, ns = a.ns;
  const bool live = 4 * l < N0;
  f4v c1 = f4v{0.f, 0.f, 0.f, 0.f}, c2 = c1, c3 = c1;
  if (live) {
    c1 = *reinterpret_cast<const f4v *>(a.consts + 4 * l);
    c2 = *reinterpret_cast<const f4v *>(a.consts + N0 + 4 * l);
    c3 = *reinterpret_cast<const f4v *>(a.consts + 2 * N0 + 4 * l);
  }
  f4v dx = f4v{0.f, 0.f, 0.f, 0.f}, dy = dx, dz = dx;            // coordinate columns, summed over this wave's points
  const unsigned nwaves = gridDim.x * (kLiftBlock / 64);
  for (unsigned n = __builtin_amdgcn_readfirstlane(blockIdx.x * (kLiftBlock / 64) + wv); n < a.npoints; n += nwaves) {
    const int p0 = a.ptr[n], p1 = a.ptr[n + 1];
    const float px = a.xyz[(size_t)n * 3 + 0], py = a.xyz[(size_t)n * 3 + 1], pz = a.xyz[(size_t)n * 3 + 2];
    f4v acc = f4v{0.f, 0.f, 0.f, 0.f};
    for (int base = p0; base < p1; base += 64) {
      const int cnt = p1 - base < 64 ? p1 - base : 64;
      int myref = 0;
      float rx = 0.f, ry = 0.f, rz = 0.f;
      if (lane < cnt) {
        myref = a.refs[base + lane];
        const Row3f q = *reinterpret_cast<const Row3f *>(a.new_xyz + (size_t)((unsigned)myref / (unsigned)ns) * 3);
        rx = px - q.a; ry = py - q.b; rz = pz - q.c;              // the forward's relative coordinates, bit for bit
        if (a.normalize) { rx = __fdiv_rn(rx, a.radius); ry = __fdiv_rn(ry, a.radius); rz = __fdiv_rn(rz, a.radius); }
      }
      for (int t = 0; t * R < cnt; t += 4) {
        f4v g[4], y[4];
        float ex[4], ey[4], ez[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = (t + u) * R + sub;
          const int src = i & 63;
          const int r = __shfl(myref, src);
          ex[u] = __shfl(rx, src); ey[u] = __shfl(ry, src); ez[u] = __shfl(rz, src);
          ok[u] = i < cnt && live;
          g[u] = f4v{0.f, 0.f, 0.f, 0.f};
          y[u] = g[u];
          if (ok[u]) {
            g[u] = *reinterpret_cast<const f4v *>(a.G + (size_t)r * N0 + 4 * l);
            y[u] = *reinterpret_cast<const f4v *>(a.Y0 + (size_t)r * N0 + 4 * l);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!ok[u]) continue;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float gy = __fmaf_rn(c1[c], g[u][c], __fmaf_rn(c2[c], y[u][c], c3[c]));
            acc[c] = __fadd_rn(acc[c], gy);
            dx[c] = __fmaf_rn(gy, ex[u], dx[c]);
            dy[c] = __fmaf_rn(gy, ey[u], dy[c]);
            dz[c] = __fmaf_rn(gy, ez[u], dz[c]);
          }
        }
      }
    }
#pragma unroll
    for (int d = 32; d >= LPR; d >>= 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __fadd_rn(acc[c], __shfl_xor(acc[c], d));
    }
    if (sub == 0 && live) *reinterpret_cast<f4v *>(a.S + (size_t)n * N0 + 4 * l) = acc;
  }
  // coordinate columns: sub-waves -> waves (LDS) -> one fp32 atomic per entry and workgroup
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
#pragma unroll
    for (int c = 0; c < 4; ++c) { red[0][wv][4 * l + c] = dx[c]; red[1][wv][4 * l + c] = dy[c]; red[2][wv][4 * l + c] = dz[c]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 3 * N0; e += kLiftBlock) {
    const int d = e / N0, col = e % N0;
    const float t = (red[d][0][col] + red[d][1][col]) + (red[d][2][col] + red[d][3][col]);
    atomicAdd(a.dWx + (size_t)col * 3 + d, t);
  }
}

bool lift_shape_ok(int N0) { return N0 >= 16 && N0 <= 256 && (N0 & 3) == 0; }
}  // namespace

extern "C" int pn2_group_lift_supported(int N0) { return lift_shape_ok(N0) ? 1 : 0; }

extern "C" int pn2_group_lift_rows(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                                   const float *new_xyz, const int *idx, const float *P, const float *Wx, float *Y,
                                   double *stats, void *stream) {
  if (B < 0 || N < 0 || m < 0 || ns < 0) return PN2_EINVAL;
  if (!lift_shape_ok(N0) || (normalize && !(radius > 0.f))) return PN2_EINVAL;
  const long long centres = (long long)B * m;
  if (centres == 0 || ns == 0) return PN2_OK;
  if (centres > 0x7fffffffLL - 64 || (long long)B * N >= 0x7fffffffLL) return PN2_EINVAL;
  if (!xyz || !new_xyz || !idx || !P || !Wx || !Y) return PN2_ENULL;
  if ((((uintptr_t)P) | ((uintptr_t)Y)) & 15) return PN2_EINVAL;
  LiftFwdArgs a{xyz, new_xyz, idx, P, Wx, Y, stats, N, m, ns, N0, normalize ? 1 : 0, radius, (int)centres,
                (int)((centres + 3) / 4)};
  // persistent grid: 8 XCD shares x up to 256 workgroups, every workgroup flushes its column sums once
  const int per = (a.chunks + 7) / 8;
  const int nwg = per < 256 ? per : 256;
  const dim3 grid((unsigned)(nwg * 8)), block(kLiftBlock);
  hipStream_t s = (hipStream_t)stream;
  if (N0 <= 64) hipLaunchKernelGGL(group_lift_rows_kernel<4>, grid, block, 0, s, a);
  else if (N0 <= 128) hipLaunchKernelGGL(group_lift_rows_kernel<2>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(group_lift_rows_kernel<1>, grid, block, 0, s, a);
  return pn2_check_launch();
}

extern "C" int pn2_group_lift_rows_grad(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                                        const float *new_xyz, const float *G, const float *Y0, const float *consts,
                                        const int *ptr, const int *refs, float *S, float *dWx, void *stream) {
  if (B < 0 || N < 0 || m < 0 || ns <= 0) return PN2_EINVAL;
  if (!lift_shape_ok(N0) || (normalize && !(radius > 0.f))) return PN2_EINVAL;
  const size_t npoints = (size_t)B * N;
  if (npoints == 0) return PN2_OK;
  if (npoints >= 0x7fffffffull || (long long)B * m * ns >= 0x7fffffffLL) return PN2_EINVAL;
  if (!xyz || !new_xyz || !G || !Y0 || !consts || !ptr || !refs || !S || !dWx) return PN2_ENULL;
  if ((((uintptr_t)G) | ((uintptr_t)Y0) | ((uintptr_t)S) | ((uintptr_t)consts)) & 15) return PN2_EINVAL;
  LiftBwdArgs a{xyz, new_xyz, G, Y0, consts, ptr, refs, S, dWx, ns, N0, normalize ? 1 : 0, radius, (unsigned)npoints};
  const unsigned waves_wanted = 256u * 32u;
  unsigned grid = (unsigned)((npoints < waves_wanted ? npoints : waves_wanted) + 3) / 4;
  if (grid == 0) grid = 1;
  hipStream_t s = (hipStream_t)stream;
  if (N0 <= 64) hipLaunchKernelGGL(group_lift_rows_grad_kernel<4>, dim3(grid), dim3(kLiftBlock), 0, s, a);
  else if (N0 <= 128) hipLaunchKernelGGL(group_lift_rows_grad_kernel<2>, dim3(grid), dim3(kLiftBlock), 0, s, a);
  else hipLaunchKernelGGL(group_lift_rows_grad_kernel<1>, dim3(grid), dim3(kLiftBlock), 0, s, a);
  return pn2_check_launch();
}
