// gcn_rows.hip — TripletGCN edge primitives for gfx950.
//
// Replace torch_geometric 2.0.2 MessagePassing.__lift__ (index_select) and
// torch_scatter 2.0.9 scatter(reduce='add') at
// scene_graph_prediction/scene_graph_helpers/model/gcns/network_TripletGCN.py:41,57.
// Rows are H contiguous floats (256-1280): a wavefront moves one row with
// coalesced (vectorised when aligned) accesses; the strided ldo/col0 form lets
// the caller assemble cat[x_i, e, x_j] (:46) without a separate torch.cat.
#include "pn2_common.h"

namespace {
constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void gather_rows_kernel(int H, int ldo, int col0,
                                                            const float *__restrict__ x,
                                                            const int64_t *__restrict__ index,
                                                            float *__restrict__ out, size_t total) {
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (size_t)gridDim.x * kBlock) {
    const size_t r = e / H;
    const int h = (int)(e - r * H);
    out[r * ldo + col0 + h] = x[(size_t)index[r] * H + h];
  }
}

__global__ __launch_bounds__(kBlock) void scatter_add_rows_kernel(int H, int lds, int col0,
                                                                 const float *__restrict__ src,
                                                                 const int64_t *__restrict__ index,
                                                                 float *__restrict__ out, size_t total) {
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (size_t)gridDim.x * kBlock) {
    const size_t r = e / H;
    const int h = (int)(e - r * H);
    atomicAdd(out + (size_t)index[r] * H + h, src[r * lds + col0 + h]);
  }
}

// Deterministic CSR segment sum: thread (n,h) adds its segment's rows in the
// stable-sorted (= original edge) order => bitwise equal to a sequential
// scatter_add_ on the CPU.
__global__ __launch_bounds__(kBlock) void segment_sum_rows_kernel(int H, int lds, int col0,
                                                                 const float *__restrict__ src,
                                                                 const int64_t *__restrict__ order,
                                                                 const int64_t *__restrict__ rowptr,
                                                                 float *__restrict__ out, size_t total) {
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (size_t)gridDim.x * kBlock) {
    const size_t n = e / H;
    const int h = (int)(e - n * H);
    const int64_t p0 = rowptr[n], p1 = rowptr[n + 1];
    float acc = 0.f;
    for (int64_t p = p0; p < p1; ++p) acc = __fadd_rn(acc, src[(size_t)order[p] * lds + col0 + h]);
    out[e] = acc;
  }
}

inline unsigned capped(size_t work) {
  size_t g = (work + kBlock - 1) / kBlock;
  if (g > 8192) g = 8192;
  return (unsigned)(g ? g : 1);
}
}  // namespace

extern "C" int pn2_gather_rows(int64_t E, int H, int64_t N, int ldo, int col0, const float *x,
                               const int64_t *index, float *out, void *stream) {
  if (E < 0 || H < 0 || N < 0 || col0 < 0 || ldo < col0 + H) return PN2_EINVAL;
  const size_t total = (size_t)E * H;
  if (total == 0) return PN2_OK;
  if (!x || !index || !out) return PN2_ENULL;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(capped(total)), dim3(kBlock), 0, (hipStream_t)stream,
                     H, ldo, col0, x, index, out, total);
  return pn2_check_launch();
}

extern "C" int pn2_scatter_add_rows(int64_t E, int H, int64_t N, int lds, int col0,
                                    const float *src, const int64_t *index, float *out,
                                    void *stream) {
  if (E < 0 || H < 0 || N < 0 || col0 < 0 || lds < col0 + H) return PN2_EINVAL;
  const size_t total = (size_t)E * H;
  if (total == 0) return PN2_OK;
  if (!src || !index || !out) return PN2_ENULL;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(capped(total)), dim3(kBlock), 0,
                     (hipStream_t)stream, H, lds, col0, src, index, out, total);
  return pn2_check_launch();
}

extern "C" int pn2_segment_sum_rows(int64_t E, int H, int64_t N, int lds, int col0,
                                    const float *src, const int64_t *order,
                                    const int64_t *rowptr, float *out, void *stream) {
  if (E < 0 || H < 0 || N < 0 || col0 < 0 || lds < col0 + H) return PN2_EINVAL;
  const size_t total = (size_t)N * H;
  if (total == 0) return PN2_OK;
  if (!rowptr || !out || (E > 0 && (!src || !order))) return PN2_ENULL;
  hipLaunchKernelGGL(segment_sum_rows_kernel, dim3(capped(total)), dim3(kBlock), 0,
                     (hipStream_t)stream, H, lds, col0, src, order, rowptr, out, total);
  return pn2_check_launch();
}
