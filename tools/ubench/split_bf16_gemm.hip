// split_bf16_gemm.hip — is an fp32-ACCURATE product on the bf16 matrix cores faster than the exact fp32 MFMA?
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (157 TF/s, /opt/skills/guides/MI355X_MICROARCH.md:41): every shared-MLP
// kernel of the headline step is bound by it.  v_mfma_f32_32x32x16_bf16 runs 16x faster.  Writing each fp32 operand as
// hi + mid + lo (three bf16 values, round to nearest: |mid| <= 2^-9 |x|, |lo| <= 2^-17 |x|, remainder <= 2^-26 |x|) and
// keeping the six products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid (the dropped ones are <= 2^-24 of the term) costs
// six bf16 matrix instructions per fp32 one: 6/16 of the time, with fp32 accumulation inside the matrix core.
//
// This program measures, at the SA2 128 -> 256 shape (M = 1 048 576 rows) and three others:
//   exact  : the library's pn2_mlp_gemm (fp32 MFMA; dlopen of 4d-or_amd/libpn2_hip.so)
//   split3 : three pieces, six products          (the candidate)
//   split2 : two pieces (hi, lo), three products (~2^-17: for the accuracy / cost trade-off)
//   bf16   : one piece, one product              (rate reference: what the staging alone sustains)
// and for each: time, TF/s on 2 M N K, GB/s on 4 (M K + N K + M N), and the error against an fp64 product on 256 sampled
// rows (max |err|, and err relative to sum_k |x||w|, the natural scale of a dot product's rounding error).
//
//   hipcc --offload-arch=gfx950 -O3 split_bf16_gemm.hip -o split_bf16_gemm -ldl && ./split_bf16_gemm [path/to/libpn2_hip.so]
// Output: one JSON line per (shape, flavour).  The operand split runs on the VALU while the matrix pipe works (bf16 MFMA and
// VALU overlap on a SIMD, unlike fp32 MFMA: tools/ubench/mfma_valu_overlap.hip).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned bf_pack(float lo, float hi) {
  return (unsigned)__builtin_bit_cast(unsigned short, (bf16)lo) | ((unsigned)__builtin_bit_cast(unsigned short, (bf16)hi) << 16);
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// x (two floats) -> P packed bf16 pairs: piece p = RNE(x - sum of the pieces before it)
template <int P>
__device__ __forceinline__ void split2(float a, float b, unsigned (&out)[P]) {
#pragma unroll
  for (int p = 0; p < P; ++p) {
    out[p] = bf_pack(a, b);
    if (p + 1 < P) { a -= bf_lo(out[p]); b -= bf_hi(out[p]); }     // exact: the piece is the leading bits of the operand
  }
}

// W [N][K] fp32 -> planes Wp [P][N][K] bf16
template <int P>
__global__ void split_w_kernel(int total2, const float *__restrict__ W, unsigned *__restrict__ Wp) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total2; e += gridDim.x * 256) {
    unsigned o[P];
    split2<P>(W[2 * e], W[2 * e + 1], o);
#pragma unroll
    for (int p = 0; p < P; ++p) Wp[(size_t)p * total2 + e] = o[p];
  }
}

constexpr int BM = 128, BN = 128, KC = 32, AP = KC + 8;      // bf16 row pitch 80 bytes: 16-byte aligned, conflict-free b128 reads

// out[M][N] = X[M][K] W[N][K]^T; workgroup = 4 waves = 128 rows x 128 columns (blockIdx.y = column block), K in chunks of 32:
// registers hold the next chunk (raw fp32 A, pre-split W planes) while the matrix core works on the current one from LDS.
template <int P>
__global__ __launch_bounds__(256, 2) void split_gemm_kernel(long long M, int K, int N, const float *__restrict__ X,
                                                           const unsigned *__restrict__ Wp, float *__restrict__ Y) {
  __shared__ __attribute__((aligned(16))) bf16 sA[P][BM * AP];
  __shared__ __attribute__((aligned(16))) bf16 sW[P][BN * AP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.y * BN;
  const long long ntiles = (M + BM - 1) / BM;
  const int nchunks = K / KC;
  const size_t plane = (size_t)N * K / 2;                      // dwords per W plane
  // A: thread -> (row tid/8 + 32 i, columns 4 (tid%8) ..+3), i < 4;  W: thread -> (row tid/4 + 64 i, columns 8 (tid%4) ..+7), i < 2
  const int ar = tid >> 3, ak = (tid & 7) * 4;
  const int wr = tid >> 2, wk = (tid & 3) * 8;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long m0 = tile * BM;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float4 ra[4];
    u32x4 rw[P][2];
    auto issue = [&](int kc) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long row = m0 + ar + 32 * i;
        ra[i] = row < M ? *reinterpret_cast<const float4 *>(X + (size_t)row * K + kc * KC + ak) : float4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          rw[p][i] = *reinterpret_cast<const u32x4 *>(Wp + p * plane + ((size_t)(n0 + wr + 64 * i) * K + kc * KC + wk) / 2);
    };
    auto commit = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned lo2[P], hi2[P];
        split2<P>(ra[i].x, ra[i].y, lo2);
        split2<P>(ra[i].z, ra[i].w, hi2);
#pragma unroll
        for (int p = 0; p < P; ++p) *reinterpret_cast<u32x2 *>(&sA[p][(ar + 32 * i) * AP + ak]) = u32x2{lo2[p], hi2[p]};
      }
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4 *>(&sW[p][(wr + 64 * i) * AP + wk]) = rw[p][i];
    };
    issue(0);
    for (int kc = 0; kc < nchunks; ++kc) {
      __syncthreads();                               // the previous chunk's fragment reads are done
      commit();
      if (kc + 1 < nchunks) issue(kc + 1);
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        bf16x8 af[P];
#pragma unroll
        for (int p = 0; p < P; ++p) af[p] = *(const bf16x8 *)&sA[p][(wave * 32 + (lane & 31)) * AP + ks * 16 + (lane >> 5) * 8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          bf16x8 bfr[P];
#pragma unroll
          for (int p = 0; p < P; ++p) bfr[p] = *(const bf16x8 *)&sW[p][(t * 32 + (lane & 31)) * AP + ks * 16 + (lane >> 5) * 8];
          // small products first (their sum is formed before it meets the large one inside the accumulator chain)
#pragma unroll
          for (int d = P - 1; d >= 0; --d)
#pragma unroll
            for (int i = 0; i <= d; ++i)
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[d - i], acc[t], 0, 0, 0);
        }
      }
    }
    // C layout: lane holds column t*32 + (lane & 31), rows (r&3) + 8 (r>>2) + 4 (lane>>5) of its wave's 32
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M) Y[(size_t)row * N + n0 + t * 32 + (lane & 31)] = acc[t][r];
      }
  }
}

// ---- v2: the same product software-pipelined like the library's exact kernel -------------------------------------------
// 512 threads = 8 waves (4 row blocks x 2 column blocks: a wave owns 32 rows x 64 columns), tile 128 x 128, K chunks of 32,
// TWO LDS buffers: the registers of step s+1 (loaded while step s-1 ran) are split and written to the other buffer behind
// the matrix instructions of step s, the loads of step s+2 are issued in front of them; one barrier per step; the
// (tile, chunk) steps of a persistent workgroup form ONE pipeline (no drain between tiles).
template <int P>
__global__ __launch_bounds__(512, 2) void split_gemm_v2_kernel(long long M, int K, int N, const float *__restrict__ X,
                                                              const unsigned *__restrict__ Wp, float *__restrict__ Y) {
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];           // [2][P][(BM + BN) * AP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr_ = wave & 3, wc_ = wave >> 2;                           // row block / column block of this wave
  const int n0 = blockIdx.y * BN;
  const long long ntiles = (M + BM - 1) / BM;
  const int nchunks = K / KC;
  const size_t plane = (size_t)N * K / 2;
  const int ar = tid >> 3, ak = (tid & 7) * 4;                         // A: rows ar, ar + 64; columns ak .. ak + 3
  const int wn = tid >> 2, wk = (tid & 3) * 8;                         // W: row wn, columns wk .. wk + 7 (one 16-byte piece per plane)
  auto sA = [&](int buf, int p) { return lds + ((size_t)(buf * P + p)) * ((BM + BN) * AP); };
  auto sW = [&](int buf, int p) { return lds + ((size_t)(buf * P + p)) * ((BM + BN) * AP) + BM * AP; };
  const long long my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
  const long long total = my_tiles * nchunks;
  long long l_tile = blockIdx.x;
  int l_chunk = 0;
  float4 ra[2];
  u32x4 rw[P];
  auto issue = [&]() {
    const long long m0 = l_tile * BM;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long row = m0 + ar + 64 * i;
      ra[i] = row < M ? *reinterpret_cast<const float4 *>(X + (size_t)row * K + l_chunk * KC + ak) : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
      rw[p] = *reinterpret_cast<const u32x4 *>(Wp + p * plane + ((size_t)(n0 + wn) * K + l_chunk * KC + wk) / 2);
    if (++l_chunk == nchunks) { l_chunk = 0; l_tile += gridDim.x; if (l_tile >= ntiles) l_tile = blockIdx.x; }   // (surplus loads: harmless)
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned lo2[P], hi2[P];
      split2<P>(ra[i].x, ra[i].y, lo2);
      split2<P>(ra[i].z, ra[i].w, hi2);
#pragma unroll
      for (int p = 0; p < P; ++p) *reinterpret_cast<u32x2 *>(&sA(buf, p)[(ar + 64 * i) * AP + ak]) = u32x2{lo2[p], hi2[p]};
    }
#pragma unroll
    for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4 *>(&sW(buf, p)[wn * AP + wk]) = rw[p];
  };
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  issue();
  commit(0);
  issue();
  __syncthreads();
  long long c_tile = blockIdx.x;
  int c_chunk = 0;
  for (long long s = 0; s < total; ++s) {
    const int buf = (int)(s & 1);
    float4 na[2];
    u32x4 nw[P];
    // registers of step s+1 are in ra / rw; move them aside and issue step s+2
#pragma unroll
    for (int i = 0; i < 2; ++i) na[i] = ra[i];
#pragma unroll
    for (int p = 0; p < P; ++p) nw[p] = rw[p];
    issue();
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      bf16x8 af[P];
#pragma unroll
      for (int p = 0; p < P; ++p) af[p] = *(const bf16x8 *)&sA(buf, p)[(wr_ * 32 + (lane & 31)) * AP + ks * 16 + (lane >> 5) * 8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8 bfr[P];
#pragma unroll
        for (int p = 0; p < P; ++p) bfr[p] = *(const bf16x8 *)&sW(buf, p)[((wc_ * 2 + t) * 32 + (lane & 31)) * AP + ks * 16 + (lane >> 5) * 8];
#pragma unroll
        for (int d = P - 1; d >= 0; --d)
#pragma unroll
          for (int i = 0; i <= d; ++i)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[d - i], acc[t], 0, 0, 0);
      }
    }
    // step s+1 -> the other buffer (its readers finished at the previous barrier)
    {
      float4 keep[2];
      u32x4 keepw[P];
#pragma unroll
      for (int i = 0; i < 2; ++i) { keep[i] = ra[i]; ra[i] = na[i]; }
#pragma unroll
      for (int p = 0; p < P; ++p) { keepw[p] = rw[p]; rw[p] = nw[p]; }
      commit(buf ^ 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) ra[i] = keep[i];
#pragma unroll
      for (int p = 0; p < P; ++p) rw[p] = keepw[p];
    }
    if (++c_chunk == nchunks) {
      const long long m0 = c_tile * BM;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long row = m0 + wr_ * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < M) Y[(size_t)row * N + n0 + (wc_ * 2 + t) * 32 + (lane & 31)] = acc[t][r];
          acc[t][r] = 0.f;
        }
      c_chunk = 0;
      c_tile += gridDim.x;
    }
    __syncthreads();
  }
}

typedef int (*gemm_fn)(long long, int, int, int, int, const float *, const float *, const float *, const float *, const float *,
                       const int *, const float *, int, const float *, float *, double *, const float *, const float *, void *);

struct Err { double max_abs, max_rel; };

static Err check(const std::vector<float> &X, const std::vector<float> &W, const float *Yh, long long M, int K, int N, int nsample) {
  Err e = {0, 0};
  for (int si = 0; si < nsample; ++si) {
    const long long row = (M / nsample) * si + (si * 37) % 128;
    if (row >= M) continue;
    for (int n = 0; n < N; ++n) {
      double s = 0, sc = 0;
      for (int k = 0; k < K; ++k) {
        const double p = (double)X[(size_t)row * K + k] * (double)W[(size_t)n * K + k];
        s += p;
        sc += fabs(p);
      }
      const double d = fabs((double)Yh[(size_t)si * N + n] - s);
      if (d > e.max_abs) e.max_abs = d;
      if (sc > 0 && d / sc > e.max_rel) e.max_rel = d / sc;
    }
  }
  return e;
}

template <typename F>
static float time_ms(F f, int reps) {
  hipEvent_t s, e;
  CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
  f(); f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(s));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e));
  CK(hipEventSynchronize(e));
  float ms;
  CK(hipEventElapsedTime(&ms, s, e));
  return ms / reps;
}

template <int P, bool V2 = false>
static void run_split(const char *name, long long M, int K, int N, const float *dX, const float *dW, float *dY, unsigned *dWp,
                      const std::vector<float> &X, const std::vector<float> &W, int nsample) {
  const int total2 = N * K / 2;
  hipLaunchKernelGGL(split_w_kernel<P>, dim3(256), dim3(256), 0, 0, total2, dW, dWp);
  const long long ntiles = (M + BM - 1) / BM;
  const unsigned ny = (unsigned)(N / BN);
  unsigned gx = 512 / ny;                                    // two workgroups per CU
  if (gx > ntiles) gx = (unsigned)ntiles;
  const size_t lds2 = (size_t)2 * P * (BM + BN) * AP * sizeof(bf16);
  unsigned gx2 = 256 / ny;                                   // v2: one 8-wave workgroup per CU (two LDS buffers)
  if (gx2 > ntiles) gx2 = (unsigned)ntiles;
  if (V2) CK(hipFuncSetAttribute((const void *)split_gemm_v2_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  auto f = [&]() {
    if (V2) hipLaunchKernelGGL(split_gemm_v2_kernel<P>, dim3(gx2, ny), dim3(512), lds2, 0, M, K, N, dX, dWp, dY);
    else hipLaunchKernelGGL(split_gemm_kernel<P>, dim3(gx, ny), dim3(256), 0, 0, M, K, N, dX, dWp, dY);
  };
  const float ms = time_ms(f, 20);
  CK(hipGetLastError());
  std::vector<float> Yh((size_t)nsample * N);
  for (int si = 0; si < nsample; ++si) {
    const long long row = (M / nsample) * si + (si * 37) % 128;
    if (row < M) CK(hipMemcpy(&Yh[(size_t)si * N], dY + (size_t)row * N, (size_t)N * 4, hipMemcpyDeviceToHost));
  }
  const Err e = check(X, W, Yh.data(), M, K, N, nsample);
  printf("{\"kernel\": \"%s\", \"M\": %lld, \"K\": %d, \"N\": %d, \"ms\": %.4f, \"TFLOPps\": %.1f, \"GBps\": %.0f, "
         "\"max_abs_err_vs_f64\": %.3e, \"max_err_rel_to_sum_abs_terms\": %.3e, \"products\": %d}\n",
         name, M, K, N, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12, 4.0 * ((double)M * K + (double)N * K + (double)M * N) / (ms * 1e-3) / 1e9,
         e.max_abs, e.max_rel, P * (P + 1) / 2);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const char *libpath = argc > 1 ? argv[1] : "4d-or_amd/libpn2_hip.so";
  void *h = dlopen(libpath, RTLD_NOW);
  gemm_fn exact = h ? (gemm_fn)dlsym(h, "pn2_mlp_gemm") : nullptr;
  if (!exact) fprintf(stderr, "note: %s not loadable (%s): the exact row is skipped\n", libpath, dlerror());
  struct Shape { long long M; int K, N; };
  const Shape shapes[] = {{1048576, 128, 256}, {1048576, 128, 128}, {4194304, 64, 128}, {262144, 256, 128}, {32768, 256, 256}};
  const int nsample = 256;
  for (const Shape &sh : shapes) {
    const long long M = sh.M;
    const int K = sh.K, N = sh.N;
    std::vector<float> X((size_t)M * K), W((size_t)N * K);
    unsigned long long st = 0x9E3779B97F4A7C15ull ^ (unsigned long long)(M * 31 + K * 7 + N);
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
    // (sum of four uniforms: near-normal, unit variance, a fraction of Box-Muller's host time for 600 M values)
    auto gauss = [&]() { return ((rnd() + rnd() + rnd() + rnd()) - 2.0) * 1.7320508075688772; };
    for (auto &x : X) { const double g = gauss(); x = (float)(g > 0 ? g : 0.0); }       // relu(bn(.)) activations
    for (auto &w : W) w = (float)(gauss() / sqrt((double)K));                            // kaiming-scale weights
    float *dX, *dW, *dY;
    unsigned *dWp;
    CK(hipMalloc(&dX, (size_t)M * K * 4)); CK(hipMalloc(&dW, (size_t)N * K * 4)); CK(hipMalloc(&dY, (size_t)M * N * 4));
    CK(hipMalloc(&dWp, (size_t)3 * N * K * 2));
    CK(hipMemcpy(dX, X.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, W.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
    if (exact) {
      auto f = [&]() { exact(M, K, N, 0, 0, dX, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, dW, dY, nullptr, nullptr, nullptr, nullptr); };
      const float ms = time_ms(f, 20);
      std::vector<float> Yh((size_t)nsample * N);
      for (int si = 0; si < nsample; ++si) {
        const long long row = (M / nsample) * si + (si * 37) % 128;
        if (row < M) CK(hipMemcpy(&Yh[(size_t)si * N], dY + (size_t)row * N, (size_t)N * 4, hipMemcpyDeviceToHost));
      }
      const Err e = check(X, W, Yh.data(), M, K, N, nsample);
      printf("{\"kernel\": \"exact_fp32_mfma (pn2_mlp_gemm)\", \"M\": %lld, \"K\": %d, \"N\": %d, \"ms\": %.4f, \"TFLOPps\": %.1f, \"GBps\": %.0f, "
             "\"max_abs_err_vs_f64\": %.3e, \"max_err_rel_to_sum_abs_terms\": %.3e, \"products\": 1}\n",
             M, K, N, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12, 4.0 * ((double)M * K + (double)N * K + (double)M * N) / (ms * 1e-3) / 1e9,
             e.max_abs, e.max_rel);
      fflush(stdout);
    }
    run_split<3>("split3_bf16x6 (hi+mid+lo, six products)", M, K, N, dX, dW, dY, dWp, X, W, nsample);
    run_split<3, true>("split3_bf16x6 v2 (pipelined, two LDS buffers)", M, K, N, dX, dW, dY, dWp, X, W, nsample);
    run_split<1, true>("bf16x1 v2 (rate reference)", M, K, N, dX, dW, dY, dWp, X, W, nsample);
    run_split<2>("split2_bf16x3 (hi+lo, three products)", M, K, N, dX, dW, dY, dWp, X, W, nsample);
    run_split<1>("bf16x1 (one product: rate reference)", M, K, N, dX, dW, dY, dWp, X, W, nsample);
    CK(hipFree(dX)); CK(hipFree(dW)); CK(hipFree(dY)); CK(hipFree(dWp));
  }
  return 0;
}
