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

inline unsigned grid1d(size_t work, int block = kBlock) {
  size_t g = (work + block - 1) / block;
  return (unsigned)(g ? g : 1);
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
__global__ __launch_bounds__(kBlock) void group_concat_rows_kernel(
    int N, int m, int ns, int C, int Cx, int normalize, float radius,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ feats, const int *__restrict__ idx, float *__restrict__ out,
    size_t total /* rows * W */) {
  const int W = Cx + C;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (size_t)gridDim.x * kBlock) {
    const size_t row = e / W;
    const int w = (int)(e - row * W);
    const size_t bj = row / ns;       // b*m + j
    const size_t b = bj / m;
    const int ii = idx[row];
    float v;
    if (w < Cx) {
      v = xyz[(b * N + ii) * 3 + w] - new_xyz[bj * 3 + w];
      if (normalize) v = __fdiv_rn(v, radius);
    } else {
      v = feats[(b * N + ii) * C + (w - Cx)];
    }
    out[e] = v;
  }
}

__global__ __launch_bounds__(kBlock) void group_rows_grad_kernel(
    int N, int m, int ns, int C, int ldg, int col0, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_feats, size_t total /* rows * C */) {
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (size_t)gridDim.x * kBlock) {
    const size_t row = e / C;
    const int c = (int)(e - row * C);
    const size_t b = row / ((size_t)m * ns);
    const int ii = idx[row];
    atomicAdd(grad_feats + (b * N + ii) * C + c, grad_out[row * ldg + col0 + c]);
  }
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
  hipLaunchKernelGGL(group_concat_rows_kernel, dim3(capped(total)), dim3(kBlock), 0,
                     (hipStream_t)stream, N, m, ns, C, Cx, normalize, radius, xyz, new_xyz, feats,
                     idx, out, total);
  return pn2_check_launch();
}

extern "C" int pn2_group_rows_grad(int B, int N, int m, int ns, int C, int ldg, int col0,
                                   const float *grad_out, const int *idx,
                                   float *grad_feats, void *stream) {
  if (B < 0 || N < 0 || m < 0 || ns < 0 || C < 0 || col0 < 0 || ldg < col0 + C) return PN2_EINVAL;
  const size_t total = (size_t)B * m * ns * (size_t)C;
  if (total == 0) return PN2_OK;
  if (!grad_out || !idx || !grad_feats) return PN2_ENULL;
  hipLaunchKernelGGL(group_rows_grad_kernel, dim3(capped(total)), dim3(kBlock), 0,
                     (hipStream_t)stream, N, m, ns, C, ldg, col0, grad_out, idx, grad_feats, total);
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
