// graph_algos.hip — all-pairs shortest paths + edge-feature paths of the Graphormer role-prediction pre-processing.
//
// Replaces role_prediction/graphormer/algos.pyx:11-89 (Cython, one graph at a time on the host; called from
// role_prediction/graphormer/wrapper.py:39-41): `floyd_warshall` (hop distances with unreachable = 12 and the
// intermediate-vertex matrix `path`) and `gen_edge_input` (edge features along every reconstructed shortest path).
// The graphs are tiny (objects + people of one OR scene: <= ~30 nodes), so the win is batching: one workgroup per
// graph with the distance matrix in LDS, one thread per (i, j) pair for the path reconstruction.
//
// Bit-for-bit the reference, including its quirks: MAX_DIST = 12 doubles as "unreachable" AND as a legal vertex id in
// `path` (a pair routed through vertex 12 is skipped by gen_edge_input exactly like an unreachable one), and
// get_all_edges treats path == 0 as "direct edge", so vertex 0 is never expanded as an intermediate.
// Checked against the compiled reference module (oracle/_ref, built from the .pyx where it lies).
#include "pn2_common.h"

namespace {
constexpr long long kMaxDist = 12;     // algos.pyx:9
constexpr int kMaxNodes = 128;         // distance matrix in LDS: 128 x 128 x 4 bytes

__global__ __launch_bounds__(1024) void floyd_warshall_kernel(int n, const long long *__restrict__ adj,
                                                             long long *__restrict__ Mout, long long *__restrict__ Pout) {
  extern __shared__ int lds[];                       // M[n*n] then P[n*n]
  int *M = lds, *P = lds + n * n;
  const size_t g = (size_t)blockIdx.x * n * n;
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    const int i = e / n, j = e - i * n;
    const long long a = adj[g + e];
    M[e] = i == j ? 0 : (a == 0 ? (int)kMaxDist : (int)a);     // :28-33
    P[e] = 0;
  }
  __syncthreads();
  for (int k = 0; k < n; ++k) {                               // :36-45 (k outermost; (i, j) independent for fixed k)
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
      const int i = e / n, j = e - i * n;
      const int cost = M[i * n + k] + M[k * n + j];
      // row k and column k cannot change in round k (M[k][k] = 0), so the reads above race with no write
      if (M[e] > cost && i != k && j != k) { M[e] = cost; P[e] = k; }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {     // :48-52
    const bool far = M[e] >= (int)kMaxDist;
    Mout[g + e] = far ? kMaxDist : (long long)M[e];
    Pout[g + e] = far ? kMaxDist : (long long)P[e];
  }
}

// out[b][i][j][k][:] = edge_feat[b][v_k][v_{k+1}][:] along v = [i] + get_all_edges(path, i, j) + [j]   (:78-87); the
// caller pre-fills out with -1 (:73)
__global__ __launch_bounds__(256) void gen_edge_input_kernel(int n, int max_dist, int F, const long long *__restrict__ path,
                                                            const long long *__restrict__ feat, long long *__restrict__ out,
                                                            long long pairs) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= pairs) return;
  const int j = (int)(e % n), i = (int)((e / n) % n);
  const long long b = e / ((long long)n * n);
  const long long *Pm = path + b * n * n;
  if (i == j || Pm[i * n + j] == kMaxDist) return;
  // in-order expansion of get_all_edges with an explicit stack of (a, b) segments
  int sa[2 * kMaxNodes], sb[2 * kMaxNodes], top = 0;
  int prev = i, hop = 0;
  sa[0] = i; sb[0] = j; top = 1;
  long long *o = out + e * (long long)max_dist * F;
  const long long *Fm = feat + b * (long long)n * n * F;
  while (top > 0) {
    --top;
    const int a = sa[top], c = sb[top];
    const long long k = Pm[a * n + c];
    if (k == 0) {                                             // :56-57: direct edge a -> c
      if (hop < max_dist)
        for (int f = 0; f < F; ++f) o[(long long)hop * F + f] = Fm[((long long)prev * n + c) * F + f];
      ++hop;
      prev = c;
    } else if (top + 2 <= 2 * kMaxNodes) {
      sa[top] = (int)k; sb[top] = c; ++top;                   // right half later
      sa[top] = a; sb[top] = (int)k; ++top;                   // left half first
    } else {
      return;                                                 // cannot happen for n <= kMaxNodes
    }
  }
}
}  // namespace

extern "C" int pn2_floyd_warshall(int B, int n, const long long *adjacency, long long *dist, long long *path, void *stream) {
  if (B < 0 || n < 0 || n > kMaxNodes) return PN2_EINVAL;
  if (B == 0 || n == 0) return PN2_OK;
  if (!adjacency || !dist || !path) return PN2_ENULL;
  const size_t lds = (size_t)2 * n * n * sizeof(int);
  auto kfn = floyd_warshall_kernel;
  static bool big = false;
  if (lds > 64 * 1024 && !big) {
    if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return pn2_check_launch();
    big = true;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)B), dim3(1024), lds, (hipStream_t)stream, n, adjacency, dist, path);
  return pn2_check_launch();
}

extern "C" int pn2_gen_edge_input(int B, int n, int max_dist, int F, const long long *path, const long long *edge_feat,
                                  long long *out /* (B,n,n,max_dist,F) pre-filled with -1 */, void *stream) {
  if (B < 0 || n < 0 || n > kMaxNodes || max_dist < 0 || F < 0) return PN2_EINVAL;
  const long long pairs = (long long)B * n * n;
  if (pairs == 0 || max_dist == 0 || F == 0) return PN2_OK;
  if (!path || !edge_feat || !out) return PN2_ENULL;
  const long long blocks = (pairs + 255) / 256;
  if (blocks > 0x7fffffffLL) return PN2_EINVAL;
  hipLaunchKernelGGL(gen_edge_input_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, max_dist, F, path,
                     edge_feat, out, pairs);
  return pn2_check_launch();
}
