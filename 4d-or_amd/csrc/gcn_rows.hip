// gcn_rows.hip — TripletGCN edge primitives for gfx950.
//
// Replace torch_geometric 2.0.2 MessagePassing.__lift__ (index_select) and
// torch_scatter 2.0.9 scatter(reduce='add') at
// scene_graph_prediction/scene_graph_helpers/model/gcns/network_TripletGCN.py:41,57, and the
// BatchNorm1d(track_running_stats=False) + ReLU of build_mlp (:11-27) when several scans are
// batched block-diagonally (statistics per scan, like the reference's batch-of-one loop).
//
// Mapping: rows are H contiguous floats (256-1280).  One wavefront owns one row: the row index
// (edge endpoint, CSR entry) is a scalar load, the row moves as 16-byte accesses per lane
// (64 lanes x float4 = 256 floats per pass), so a gather / scatter is one coalesced stream per
// row instead of a 64-bit division and an index load per element.  The CSR segment sum keeps its
// accumulators in registers (up to 8 passes = H <= 2048) and adds the rows of a node in edge
// order, which makes it bit-identical to a sequential CPU scatter_add_.  The strided ldo / col0
// form lets the caller assemble cat[x_i, e, x_j] (:46) without a separate torch.cat.
#include "pn2_common.h"

namespace {
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kMaxPasses = 8;   // register accumulators of the segment sum: H <= 2048

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long long wave_row() {
  return (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
}

// ---- gather: out[r, col0 : col0+H] = x[index[r]] ----------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(long long E, int H, int ldo, int col0,
                                                            const float *__restrict__ x,
                                                            const int64_t *__restrict__ index,
                                                            float *__restrict__ out) {
  const int lane = pn2_lane();
  for (long long r = wave_row(); r < E; r += (long long)gridDim.x * kWavesPerBlock) {
    const float *src = x + (size_t)index[r] * H;
    float *dst = out + (size_t)r * ldo + col0;
    if constexpr (VEC) {
      for (int h = lane * 4; h < H; h += 256) *(f4 *)(dst + h) = *(const f4 *)(src + h);
    } else {
      for (int h = lane; h < H; h += 64) dst[h] = src[h];
    }
  }
}

// ---- q[e, :] += pa[ia[e], cola : cola+H] + pb[ib[e], colb : colb+H]  (first Linear of the triplet MLP applied to the
//      NODES before the lift: W [x_i | e | x_j] = Wa x_i + Wb e + Wc x_j, network_TripletGCN.py:46-47)
__global__ __launch_bounds__(kBlock) void gather2_add_rows_kernel(long long E, int H, int ldp, int cola, int colb,
                                                                 const float *__restrict__ p,
                                                                 const int64_t *__restrict__ ia,
                                                                 const int64_t *__restrict__ ib, float *__restrict__ q) {
  const int lane = pn2_lane();
  for (long long r = wave_row(); r < E; r += (long long)gridDim.x * kWavesPerBlock) {
    const float *a = p + (size_t)ia[r] * ldp + cola, *b = p + (size_t)ib[r] * ldp + colb;
    float *d = q + (size_t)r * H;
    for (int h = lane * 4; h < H; h += 256) {
      const f4 va = *(const f4 *)(a + h), vb = *(const f4 *)(b + h);
      f4 v = *(const f4 *)(d + h);
      v.x = __fadd_rn(v.x, __fadd_rn(va.x, vb.x)); v.y = __fadd_rn(v.y, __fadd_rn(va.y, vb.y));
      v.z = __fadd_rn(v.z, __fadd_rn(va.z, vb.z)); v.w = __fadd_rn(v.w, __fadd_rn(va.w, vb.w));
      *(f4 *)(d + h) = v;
    }
  }
}

// ---- atomic scatter-add: out[index[r]] += src[r, col0 : col0+H] --------------------------------
__global__ __launch_bounds__(kBlock) void scatter_add_rows_kernel(long long E, int H, int lds, int col0,
                                                                 const float *__restrict__ src,
                                                                 const int64_t *__restrict__ index,
                                                                 float *__restrict__ out) {
  const int lane = pn2_lane();
  for (long long r = wave_row(); r < E; r += (long long)gridDim.x * kWavesPerBlock) {
    const float *s = src + (size_t)r * lds + col0;
    float *d = out + (size_t)index[r] * H;
    for (int h = lane; h < H; h += 64) atomicAdd(d + h, s[h]);
  }
}

// ---- deterministic CSR segment sum -----------------------------------------------------------
// out[n] = sum over p in [rowptr[n], rowptr[n+1]) of src[order[p], col0 : col0+H], added in that order.
// col1 >= 0: the row contribution is src[e, col0 : col0+H] + src[e, col1 : col1+H] (the TripletGCN node message is the
// sum of the first and the last block of nn1's output, network_TripletGCN.py:50), added as ONE value per element
template <bool VEC>
__global__ __launch_bounds__(kBlock) void segment_sum_rows_kernel(long long N, int H, int lds, int col0, int col1,
                                                                 const float *__restrict__ src,
                                                                 const int64_t *__restrict__ order,
                                                                 const int64_t *__restrict__ rowptr,
                                                                 float *__restrict__ out) {
  const int lane = pn2_lane();
  for (long long n = wave_row(); n < N; n += (long long)gridDim.x * kWavesPerBlock) {
    const int64_t p0 = rowptr[n], p1 = rowptr[n + 1];
    float *dst = out + (size_t)n * H;
    if constexpr (VEC) {
      f4 acc[kMaxPasses];
#pragma unroll
      for (int i = 0; i < kMaxPasses; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
      for (int64_t p = p0; p < p1; ++p) {
        const float *s = src + (size_t)order[p] * lds + col0;
#pragma unroll
        for (int i = 0; i < kMaxPasses; ++i) {
          const int h = lane * 4 + i * 256;
          if (h < H) {
            f4 v = *(const f4 *)(s + h);
            if (col1 >= 0) {
              const f4 w = *(const f4 *)(s + (col1 - col0) + h);
              v.x = __fadd_rn(v.x, w.x); v.y = __fadd_rn(v.y, w.y); v.z = __fadd_rn(v.z, w.z); v.w = __fadd_rn(v.w, w.w);
            }
            acc[i].x = __fadd_rn(acc[i].x, v.x); acc[i].y = __fadd_rn(acc[i].y, v.y);
            acc[i].z = __fadd_rn(acc[i].z, v.z); acc[i].w = __fadd_rn(acc[i].w, v.w);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < kMaxPasses; ++i) {
        const int h = lane * 4 + i * 256;
        if (h < H) *(f4 *)(dst + h) = acc[i];
      }
    } else {
      for (int h = lane; h < H; h += 64) {
        float acc = 0.f;
        for (int64_t p = p0; p < p1; ++p) {
          const float *r = src + (size_t)order[p] * lds;
          acc = __fadd_rn(acc, col1 >= 0 ? __fadd_rn(r[col0 + h], r[col1 + h]) : r[col0 + h]);
        }
        dst[h] = acc;
      }
    }
  }
}

// ---- per-segment BatchNorm1d (+ ReLU) ----------------------------------------------------------
// x (R, ldx) columns [col0, col0 + C); ptr (S + 1) row offsets of the segments (scans).  Thread (s, c) walks the rows
// of segment s for channel c (coalesced across the 256 channels of a workgroup): mean, biased variance around that
// mean (two passes, like torch's CPU kernel), then y = [relu]((x - mean) * rstd * gamma + beta).
__global__ __launch_bounds__(kBlock) void segment_bn_fwd_kernel(int C, int ldx, int col0, int relu, float eps,
                                                               const float *__restrict__ x,
                                                               const int64_t *__restrict__ ptr,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               float *__restrict__ y, float *__restrict__ mean_out,
                                                               float *__restrict__ rstd_out) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  const int s = blockIdx.y;
  if (c >= C) return;
  const int64_t p0 = ptr[s], p1 = ptr[s + 1];
  const float n = (float)(p1 - p0);
  const float *xc = x + col0 + c;
  float sum = 0.f;
  for (int64_t r = p0; r < p1; ++r) sum += xc[(size_t)r * ldx];
  const float mean = p1 > p0 ? sum / n : 0.f;
  float ss = 0.f;
  for (int64_t r = p0; r < p1; ++r) {
    const float d = xc[(size_t)r * ldx] - mean;
    ss = fmaf(d, d, ss);
  }
  const float rstd = 1.0f / sqrtf((p1 > p0 ? ss / n : 0.f) + eps);
  mean_out[(size_t)s * C + c] = mean;
  rstd_out[(size_t)s * C + c] = rstd;
  const float a = rstd * gamma[c], b = beta[c];
  for (int64_t r = p0; r < p1; ++r) {
    float v = fmaf((xc[(size_t)r * ldx] - mean), a, b);
    if (relu) v = fmaxf(v, 0.f);
    y[(size_t)r * C + c] = v;
  }
}

// backward: gm = relu ? g * [z > 0] : g;  dx = gamma * rstd * (gm - mean(gm) - xhat * mean(gm * xhat));
// dgamma_part[s][c] = sum gm * xhat, dbeta_part[s][c] = sum gm  (summed over the segments by the caller).
__global__ __launch_bounds__(kBlock) void segment_bn_bwd_kernel(int C, int ldx, int col0, int relu,
                                                               const float *__restrict__ g,
                                                               const float *__restrict__ x,
                                                               const int64_t *__restrict__ ptr,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               const float *__restrict__ mean_in,
                                                               const float *__restrict__ rstd_in,
                                                               float *__restrict__ dx, float *__restrict__ dgamma_part,
                                                               float *__restrict__ dbeta_part) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  const int s = blockIdx.y;
  if (c >= C) return;
  const int64_t p0 = ptr[s], p1 = ptr[s + 1];
  const float n = (float)(p1 - p0);
  const float mean = mean_in[(size_t)s * C + c], rstd = rstd_in[(size_t)s * C + c];
  const float ga = gamma[c], be = beta[c];
  const float *xc = x + col0 + c;
  float s1 = 0.f, s2 = 0.f;
  for (int64_t r = p0; r < p1; ++r) {
    const float xh = (xc[(size_t)r * ldx] - mean) * rstd;
    float gv = g[(size_t)r * C + c];
    if (relu && !(fmaf(xh, ga, be) > 0.f)) gv = 0.f;
    s1 += gv;
    s2 = fmaf(gv, xh, s2);
  }
  dgamma_part[(size_t)s * C + c] = s2;
  dbeta_part[(size_t)s * C + c] = s1;
  const float m1 = p1 > p0 ? s1 / n : 0.f, m2 = p1 > p0 ? s2 / n : 0.f;
  const float k = ga * rstd;
  for (int64_t r = p0; r < p1; ++r) {
    const float xh = (xc[(size_t)r * ldx] - mean) * rstd;
    float gv = g[(size_t)r * C + c];
    if (relu && !(fmaf(xh, ga, be) > 0.f)) gv = 0.f;
    dx[(size_t)r * C + c] = k * (gv - m1 - xh * m2);
  }
}

// Running statistics of a BatchNorm1d after the training-mode steps of S scans, in scan order: running <- (1 - m) running
// + m stat_s with the unbiased variance n/(n-1) var, as torch.nn.functional.batch_norm updates them once per call
// (EXT network_PointNet.py heads: nn.BatchNorm1d(512) / (256) with running statistics, :198-203).  mean / rstd (S,C) as
// segment_bn_fwd_kernel leaves them; a scan of fewer than 1 row is skipped.
__global__ __launch_bounds__(kBlock) void segment_bn_running_kernel(int S, int C, const float *__restrict__ mean,
                                                                   const float *__restrict__ rstd,
                                                                   const int64_t *__restrict__ ptr, float eps, float momentum,
                                                                   float *__restrict__ rm, float *__restrict__ rv,
                                                                   long long *__restrict__ nbt) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c == 0 && nbt) *nbt += S;
  if (c >= C) return;
  float m = rm[c], v = rv[c];
  for (int s = 0; s < S; ++s) {
    const float n = (float)(ptr[s + 1] - ptr[s]);
    if (!(n > 0.f)) continue;
    const float r = rstd[(size_t)s * C + c];
    float var = fmaxf(1.0f / (r * r) - eps, 0.f);
    var = var * n / fmaxf(n - 1.f, 1.f);
    m = (1.f - momentum) * m + momentum * mean[(size_t)s * C + c];
    v = (1.f - momentum) * v + momentum * var;
  }
  rm[c] = m;
  rv[c] = v;
}

inline unsigned row_grid(long long rows) {
  long long g = (rows + kWavesPerBlock - 1) / kWavesPerBlock;
  if (g > 16384) g = 16384;
  return (unsigned)(g ? g : 1);
}
inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }
}  // namespace

extern "C" int pn2_gather_rows(int64_t E, int H, int64_t N, int ldo, int col0, const float *x,
                               const int64_t *index, float *out, void *stream) {
  if (E < 0 || H < 0 || N < 0 || col0 < 0 || ldo < col0 + H) return PN2_EINVAL;
  if (E == 0 || H == 0) return PN2_OK;
  if (!x || !index || !out) return PN2_ENULL;
  const bool vec = H % 4 == 0 && ldo % 4 == 0 && col0 % 4 == 0 && aligned16(x) && aligned16(out);
  if (vec)
    hipLaunchKernelGGL(gather_rows_kernel<true>, dim3(row_grid(E)), dim3(kBlock), 0, (hipStream_t)stream, (long long)E,
                       H, ldo, col0, x, index, out);
  else
    hipLaunchKernelGGL(gather_rows_kernel<false>, dim3(row_grid(E)), dim3(kBlock), 0, (hipStream_t)stream, (long long)E,
                       H, ldo, col0, x, index, out);
  return pn2_check_launch();
}

extern "C" int pn2_scatter_add_rows(int64_t E, int H, int64_t N, int lds, int col0,
                                    const float *src, const int64_t *index, float *out,
                                    void *stream) {
  if (E < 0 || H < 0 || N < 0 || col0 < 0 || lds < col0 + H) return PN2_EINVAL;
  if (E == 0 || H == 0) return PN2_OK;
  if (!src || !index || !out) return PN2_ENULL;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(row_grid(E)), dim3(kBlock), 0, (hipStream_t)stream, (long long)E, H,
                     lds, col0, src, index, out);
  return pn2_check_launch();
}

extern "C" int pn2_segment_sum_rows(int64_t E, int H, int64_t N, int lds, int col0,
                                    const float *src, const int64_t *order,
                                    const int64_t *rowptr, float *out, void *stream) {
  if (E < 0 || H < 0 || N < 0 || col0 < 0 || lds < col0 + H) return PN2_EINVAL;
  if (N == 0 || H == 0) return PN2_OK;
  if (!rowptr || !out || (E > 0 && (!src || !order))) return PN2_ENULL;
  return pn2_segment_sum2_rows(E, H, N, lds, col0, -1, src, order, rowptr, out, stream);
}

extern "C" int pn2_segment_sum2_rows(int64_t E, int H, int64_t N, int lds, int col0, int col1, const float *src,
                                     const int64_t *order, const int64_t *rowptr, float *out, void *stream) {
  if (E < 0 || H < 0 || N < 0 || col0 < 0 || lds < col0 + H || (col1 >= 0 && lds < col1 + H)) return PN2_EINVAL;
  if (N == 0 || H == 0) return PN2_OK;
  if (!rowptr || !out || (E > 0 && (!src || !order))) return PN2_ENULL;
  const bool vec = H % 4 == 0 && H <= 256 * kMaxPasses && lds % 4 == 0 && col0 % 4 == 0 && (col1 < 0 || col1 % 4 == 0) &&
                   aligned16(src) && aligned16(out);
  if (vec)
    hipLaunchKernelGGL(segment_sum_rows_kernel<true>, dim3(row_grid(N)), dim3(kBlock), 0, (hipStream_t)stream,
                       (long long)N, H, lds, col0, col1, src, order, rowptr, out);
  else
    hipLaunchKernelGGL(segment_sum_rows_kernel<false>, dim3(row_grid(N)), dim3(kBlock), 0, (hipStream_t)stream,
                       (long long)N, H, lds, col0, col1, src, order, rowptr, out);
  return pn2_check_launch();
}

extern "C" int pn2_gather2_add_rows(int64_t E, int H, int64_t N, int ldp, int cola, int colb, const float *p,
                                    const int64_t *ia, const int64_t *ib, float *q, void *stream) {
  if (E < 0 || H < 0 || N < 0 || cola < 0 || colb < 0 || ldp < cola + H || ldp < colb + H) return PN2_EINVAL;
  if (H % 4 != 0 || ldp % 4 != 0 || cola % 4 != 0 || colb % 4 != 0) return PN2_EINVAL;
  if (E == 0 || H == 0) return PN2_OK;
  if (!p || !ia || !ib || !q) return PN2_ENULL;
  if (!aligned16(p) || !aligned16(q)) return PN2_EINVAL;
  hipLaunchKernelGGL(gather2_add_rows_kernel, dim3(row_grid(E)), dim3(kBlock), 0, (hipStream_t)stream, (long long)E, H, ldp,
                     cola, colb, p, ia, ib, q);
  return pn2_check_launch();
}

extern "C" int pn2_segment_bn_rows(int64_t R, int C, int ldx, int col0, int64_t S, const float *x, const int64_t *ptr,
                                   const float *gamma, const float *beta, float eps, int relu, float *y, float *mean,
                                   float *rstd, void *stream) {
  if (R < 0 || C < 0 || S < 0 || col0 < 0 || ldx < col0 + C || S > 65535) return PN2_EINVAL;
  if (S == 0 || C == 0) return PN2_OK;
  if (!ptr || !gamma || !beta || !mean || !rstd || (R > 0 && (!x || !y))) return PN2_ENULL;
  hipLaunchKernelGGL(segment_bn_fwd_kernel, dim3((unsigned)((C + kBlock - 1) / kBlock), (unsigned)S), dim3(kBlock), 0,
                     (hipStream_t)stream, C, ldx, col0, relu, eps, x, ptr, gamma, beta, y, mean, rstd);
  return pn2_check_launch();
}

extern "C" int pn2_segment_bn_rows_grad(int64_t R, int C, int ldx, int col0, int64_t S, const float *grad_out,
                                        const float *x, const int64_t *ptr, const float *gamma, const float *beta,
                                        const float *mean, const float *rstd, int relu, float *grad_x,
                                        float *dgamma_part, float *dbeta_part, void *stream) {
  if (R < 0 || C < 0 || S < 0 || col0 < 0 || ldx < col0 + C || S > 65535) return PN2_EINVAL;
  if (S == 0 || C == 0) return PN2_OK;
  if (!ptr || !gamma || !beta || !mean || !rstd || !dgamma_part || !dbeta_part || (R > 0 && (!x || !grad_out || !grad_x)))
    return PN2_ENULL;
  hipLaunchKernelGGL(segment_bn_bwd_kernel, dim3((unsigned)((C + kBlock - 1) / kBlock), (unsigned)S), dim3(kBlock), 0,
                     (hipStream_t)stream, C, ldx, col0, relu, grad_out, x, ptr, gamma, beta, mean, rstd, grad_x,
                     dgamma_part, dbeta_part);
  return pn2_check_launch();
}

extern "C" int pn2_segment_bn_running_update(int64_t S, int C, const float *mean, const float *rstd, const int64_t *ptr,
                                             float eps, float momentum, float *running_mean, float *running_var,
                                             long long *num_batches_tracked, void *stream) {
  if (S < 0 || C < 0 || S > 0x7fffffff) return PN2_EINVAL;
  if (S == 0 || C == 0) return PN2_OK;
  if (!mean || !rstd || !ptr || !running_mean || !running_var) return PN2_ENULL;
  hipLaunchKernelGGL(segment_bn_running_kernel, dim3((unsigned)((C + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, (int)S, C, mean, rstd, ptr, eps, momentum, running_mean, running_var,
                     num_batches_tracked);
  return pn2_check_launch();
}
