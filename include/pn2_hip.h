/*
 * pn2_hip.h — C ABI of libpn2_hip.so, the MI355X (gfx950) implementation of
 * 4D-OR's scene-graph-prediction hot path.
 *
 * This is the drop-in boundary: one extern "C" entry point per
 * function of the reference's pybind11 module `pointnet2_ops._ext`
 *   EXT = scene_graph_prediction/pointnet2_dir/pointnet2_ops_lib/pointnet2_ops/_ext-src
 *   EXT/src/bindings.cpp:6-19  (nine m.def's)
 * plus the fused / TripletGCN extras that have no native counterpart in the
 * reference (they replace python-level compositions; cited per function).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer on the
 *     current HIP device; tensors are dense row-major in the stated shape;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); work
 *     is enqueued asynchronously, no host synchronisation, no allocation, no
 *     global state => re-entrant per stream (same threading contract as the
 *     reference, which launches on at::cuda::getCurrentCUDAStream());
 *   - outputs are fully written by the kernels unless stated "ACCUMULATES"
 *     (then the caller zero-fills first, like the reference's torch::zeros);
 *   - return 0 on success, a negative PN2_E* code otherwise.  Never exits the
 *     process (the reference's CUDA_CHECK_ERRORS does exit(-1),
 *     EXT/include/cuda_utils.h:30-39 — deliberately not reproduced).
 */
#ifndef PN2_HIP_H
#define PN2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden; only this header's API is exported */
#pragma GCC visibility push(default)

#define PN2_OK 0
#define PN2_EINVAL (-1)   /* bad size / argument combination            */
#define PN2_ENULL (-2)    /* required pointer is NULL                   */
#define PN2_ELAUNCH (-3)  /* hipGetLastError() != hipSuccess after launch */
#define PN2_ENOSPC (-4)   /* workspace too small                        */

/* Library identification; pn2_strerror never returns NULL. */
int pn2_abi_version(void);
const char *pn2_strerror(int code);
/* Last hipError_t observed by a failing launch on this thread (0 if none). */
int pn2_last_hip_error(void);
/* Timing events without the system-scope fence of a default HIP event (timestamps only; measurement code such as bench.py's
 * per-kernel table): create / record on a stream / elapsed milliseconds between two recorded events / destroy. */
void *pn2_event_create(void);
int pn2_event_record(void *ev, void *stream);
int pn2_event_elapsed_ms(void *start, void *stop, float *ms);
int pn2_event_destroy(void *ev);

/* ------------------------------------------------------------------ A5 ---
 * furthest_point_sampling   (EXT/include/sampling.h:6,
 *   EXT/src/sampling.cpp:66-87, EXT/src/sampling_gpu.cu:69-229)
 * xyz (B,N,3) f32 -> idxs (B,m) i32.  Bit-exact with the reference kernel's
 * selection rule: start at 0; points with |p|^2 <= 1e-3 never selected;
 * arg-max of the running min distance with the reference's tree tie-break.
 * `workspace` replaces the reference's internal `tmp(B,N)=1e10` scratch: it
 * must hold pn2_fps_workspace_bytes(B,N,m) bytes (may be 0 -> NULL allowed);
 * it needs no initialisation.
 */
size_t pn2_fps_workspace_bytes(int B, int N, int m);
int pn2_furthest_point_sampling(int B, int N, int m, const float *xyz,
                                void *workspace, size_t workspace_bytes,
                                int *idxs, void *stream);

/* Same operator with a scheduling hint (identical results).  PN2_FPS_FEW_CUS: the call is enqueued on a side stream
 * next to compute kernels (geometry of the next batch prefetched during a training step): pick the cooperative shape
 * that occupies half as many CUs (1024-thread cluster workgroups) instead of the one with the shortest latency. */
#define PN2_FPS_FEW_CUS 1
/* PN2_FPS_FEWEST_CUS (with or without PN2_FPS_FEW_CUS): as few CUs as the register file allows — up to 26 points per
 * lane in 1024-thread workgroups, e.g. TWO workgroups per 50k-point cloud: 64 CUs for 32 clouds instead of 128 (256 for
 * the latency-optimal shape).  The sampling itself takes longer (3.0 vs 2.6 us per round); worth it when the step it
 * runs next to is longer than the sampling chain anyway (fp32 training step: 16.8 -> 16.3 ms; not the bf16 one). */
#define PN2_FPS_FEWEST_CUS 2
int pn2_furthest_point_sampling_ex(int B, int N, int m, const float *xyz,
                                   void *workspace, size_t workspace_bytes,
                                   int *idxs, int flags, void *stream);
/* Sampling of clouds the caller believes to be in farthest-point ORDER already — the centres of the SA level above, which
 * pointnet2_modules.py:38-48 stores in the order its own sampling picked them.  Sampling m points from such a cloud returns
 * 0 .. m-1 (point k was the farthest of ALL original points from the first k, so it is the farthest of the subset) unless a
 * tie or a degenerate round (no candidate, NaN centre, duplicates) intervenes.  The entry point VERIFIES the order per
 * cloud on the device with the kernel's own arithmetic (every point replays its running distance against the first m - 1
 * centres and must lose every round k to point k — smaller distance, or equal with the larger rank of the kernel's tie
 * order: no barriers, 0.1 ms at 32 x 2048 -> 1024 against 0.7 ms of sampling rounds); a cloud gets 0 .. r-1 for the rounds
 * that verify and runs the sampling rounds from the first unverified round r on (none when the whole cloud verifies) —
 * results identical to pn2_furthest_point_sampling_ex for ANY input.  Only shapes whose plan is one workgroup per cloud
 * with the points in registers, and at least 256 samples (below, the rounds are cheaper than the verification launches),
 * take the shortcut; the rest run the plain call.  workspace: 256-byte aligned,
 * pn2_fps_ordered_workspace_bytes(B, N, m) bytes. */
size_t pn2_fps_ordered_workspace_bytes(int B, int N, int m);
int pn2_furthest_point_sampling_ordered(int B, int N, int m, const float *xyz, void *workspace, size_t workspace_bytes,
                                        int *idxs, int flags, void *stream);

/* Multi-workgroup ("cluster") FPS variants spin on their peers with BOUNDED waits.  A launch is only admitted when
 * the occupancy the runtime reports for the kernel covers the whole grid on the current device (otherwise the
 * streaming kernel runs), and cluster launches of one process are chained so that two never overlap.  Should a wait
 * still expire (CU-masked stream, another process on the GPU) the kernel stops, leaves the remaining indices 0 and
 * sets the int32 at pn2_fps_status_offset() bytes into `workspace` to 1 (-1: this shape has no such word); callers
 * must check it (the python binding asserts on the device, asynchronously). */
long long pn2_fps_status_offset(int B, int N, int m);
/* ... for a call made with `flags` (pn2_furthest_point_sampling_ex / _ordered): the flags take part in the choice of the
 * kernel variant, so the word's presence has to be asked for with the flags the call was made with. */
long long pn2_fps_status_offset_ex(int B, int N, int m, int flags);
/* Test hook: force a kernel variant (mode: -1 heuristic | 0 resident | 1 cluster | 2 streaming | 3 cluster with a
 * streamed tail | 4 one workgroup over the binned cloud | 5 cluster with several samples per hand-off) and cluster
 * shape (0 = heuristic).  Process-global; results never depend on it. */
int pn2_fps_set_plan_override(int mode, int G, int NC, int coop_bs, int bs);
/* Cluster variants with >= 8 point slots per lane first bin the cloud into spatially compact groups of 64 x slots points
 * (one extra launch, records in the workspace) so that a wave can skip a round's distance updates when the new sample is
 * farther from its bounding box than its largest running distance (profiles/HISTORY.md 4c).  Test / measurement hook: 0 switches
 * that off.  Process-global; results never depend on it. */
int pn2_fps_set_bucketing(int on);
int pn2_fps_get_bucketing(void);   /* the current value (1 / 0), so that a scoped override can restore it */
/* Clouds of cluster size (16k < N <= 106k points) run on two or four 1024-thread workgroups that exchange the arg-max
 * candidates of all 64 sub-blobs of the binned cloud per hand-off and accept SEVERAL samples from them whenever the next
 * ones are provably the reference's (profiles/HISTORY.md 4c, round 4).  Measurement hook: 0 restores one sample per hand-off.
 * Process-global; results never depend on it. */
int pn2_fps_set_multi(int on);
int pn2_fps_get_multi(void);
/* Test hook for the multi-workgroup FPS variant: returns the status word a
 * launch left in `workspace` (0 ok, 1 a bounded inter-workgroup wait expired,
 * <0 query failed).  Synchronises `stream`; never used on the hot path. */
int pn2_fps_coop_status(int B, const void *workspace, void *stream);

/* ------------------------------------------------------------------ A6 ---
 * gather_points / gather_points_grad  (EXT/include/sampling.h:4-5,
 *   EXT/src/sampling.cpp:15-65, EXT/src/sampling_gpu.cu:8-57)
 * points (B,C,N), idx (B,m) -> out (B,C,m).
 * grad: grad_out (B,C,m), idx (B,m) -> grad_points (B,C,N)  ACCUMULATES.
 */
int pn2_gather_points(int B, int C, int N, int m, const float *points,
                      const int *idx, float *out, void *stream);
int pn2_gather_points_grad(int B, int C, int N, int m, const float *grad_out,
                           const int *idx, float *grad_points, void *stream);

/* ------------------------------------------------------------------ A7 ---
 * ball_query  (EXT/include/ball_query.h:4, EXT/src/ball_query.cpp:8-32,
 *   EXT/src/ball_query_gpu.cu:9-54).  Argument order follows the C++ function:
 * new_xyz (B,m,3), xyz (B,N,3) -> idx (B,m,nsample) i32: the first `nsample`
 * points in ascending index with d^2 < radius^2 (strict, fp32), padded with
 * the first hit; an empty ball yields a zero row.  Every slot is written.
 */
int pn2_ball_query(int B, int N, int m, float radius, int nsample,
                   const float *new_xyz, const float *xyz, int *idx,
                   void *stream);

/* Accelerated variants of the same operator (identical results, bit for bit).  Three algorithms:
 *   PN2_BQ_SCAN  — the index-order scan of pn2_ball_query (early exit after nsample hits); no workspace;
 *   PN2_BQ_CELLS — per-cloud cell list (cells of edge >= radius, 27 neighbouring cells, hits rank-sorted by index): pays
 *                  when balls are SPARSE (the scene-graph encoders: radius 0.1 / 0.2 in 4000 / 8000-point clouds, where
 *                  the scan never exits early); nsample <= 256;
 *   PN2_BQ_SLABS — one hash-grid cell list per slab of 2048 (or, where a ball needs more than ~1.5 such slabs for its
 *                  nsample hits, 8192) consecutive indices; hits set bits of a per-slab mask, which yields them in
 *                  ascending index without a sort; the walk over the slabs stops after nsample hits: pays when balls
 *                  are CROWDED (the SA levels of the backbone).
 * pn2_ball_query_auto() is the library's choice for a shape (estimated hits per ball N r^3 against 4 nsample; small
 * clouds scan), pn2_ball_query_workspace_bytes() the workspace of that choice (0: scan), pn2_ball_query_algo_bytes() the
 * workspace of a given algorithm (0: shape not covered).  `workspace`: 16-byte aligned, no initialisation needed.
 * pn2_ball_query_ws runs the automatic choice when the workspace holds it, else the per-cloud cell list when it holds
 * pn2_ball_query_grid_bytes(), else the scan; pn2_ball_query_algo runs the named algorithm (scan when the shape or the
 * workspace does not allow it, or when radius is not positive and finite).
 */
enum { PN2_BQ_SCAN = 0, PN2_BQ_CELLS = 1, PN2_BQ_SLABS = 2 };
int pn2_ball_query_auto(int B, int N, int m, float radius, int nsample);
size_t pn2_ball_query_workspace_bytes(int B, int N, int m, float radius, int nsample);
size_t pn2_ball_query_algo_bytes(int algo, int B, int N, int m, float radius, int nsample);
size_t pn2_ball_query_grid_bytes(int B, int N, int nsample);   /* = algo_bytes(PN2_BQ_CELLS, ...) */
int pn2_ball_query_ws(int B, int N, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                      int *idx, void *workspace, size_t workspace_bytes, void *stream);
int pn2_ball_query_algo(int algo, int B, int N, int m, float radius, int nsample, const float *new_xyz,
                        const float *xyz, int *idx, void *workspace, size_t workspace_bytes, void *stream);

/* Ball query + grouping in ONE pass (round 4; SURVEY.md 8d "fused ball_query+group").  Replaces, for one SA scale,
 *   query_ball_point_kernel      EXT/src/ball_query_gpu.cu:9-44
 *   group_points_kernel (x 2)    EXT/src/group_points_gpu.cu:8-28
 *   QueryAndGroup.forward tail   OPS/pointnet2_utils.py:317-328 (centre subtraction in place, concat with the features;
 *                                with `normalize` != 0 the `grouped_xyz /= radius` of GF3D/pointnet2/pointnet2_utils.py:343-344)
 * The wave that finds a centre's first `nsample` hits (slab cell lists: ascending index without a sort) also emits the
 * neighbourhood's grouped rows:  rows[b,j,s,0:Cx] = (xyz[b,idx[b,j,s]] - new_xyz[b,j]) (/ radius),
 * rows[b,j,s,Cx:Cx+C] = feats[b,idx[b,j,s],0:C]  (Cx = use_xyz ? 3 : 0) — and idx (B,m,nsample), kept for the backward.
 * Bit-identical to pn2_ball_query + pn2_group_concat_rows.  xyz (B,N,3); new_xyz (B,m,3); feats (B,N,C) point-major or
 * NULL (C = 0).  Covers nsample <= 256 and Cx + C <= 16 (pn2_ball_query_group_supported); `workspace`: 16-byte aligned,
 * pn2_ball_query_group_workspace_bytes(B, N), no initialisation.  `slab_w`: 0 = the library's choice of slab width,
 * 1 / 4 = slabs of 2048 / 8192 consecutive indices (test and measurement hook; results never depend on it).
 * Algorithmic bytes: B (12 N + 12 m + 4 C N + 4 (Cx + C) m nsample + 4 m nsample). */
int pn2_ball_query_group_supported(int B, int N, int m, float radius, int nsample, int C, int use_xyz);
size_t pn2_ball_query_group_workspace_bytes(int B, int N);
int pn2_ball_query_group(int B, int N, int m, float radius, int nsample, int C, int use_xyz, int normalize,
                         const float *new_xyz, const float *xyz, const float *feats, int *idx, float *rows,
                         void *workspace, size_t workspace_bytes, int slab_w, void *stream);

/* sample_uniformly / ret_unique_cnt of the Group-Free-3D QueryAndGroup
 *   (GF3D/pointnet2/pointnet2_utils.py:327-336: a host loop of torch.unique + torch.randint per region).
 * idx (rows, nsample) ball-query rows, IN PLACE: the padded tail of every row (everything behind its strictly
 * ascending prefix = the unique hits) is refilled with members of that prefix drawn uniformly by a counter-based
 * generator keyed on (seed, row, slot); unique_cnt (rows) f32 (may be NULL) receives the prefix length.
 * Same distribution as the reference, not the same random stream (the reference consumes torch's host generator).
 */
int pn2_ball_query_unique_resample(long long rows, int nsample, unsigned seed, int *idx, float *unique_cnt,
                                   void *stream);

/* ------------------------------------------------------------------ A8 ---
 * group_points / group_points_grad  (EXT/include/group_points.h:4-5,
 *   EXT/src/group_points.cpp:12-62, EXT/src/group_points_gpu.cu:8-75)
 * points (B,C,N), idx (B,npoints,nsample) -> out (B,C,npoints,nsample).
 * grad: grad_out (B,C,npoints,nsample) -> grad_points (B,C,N)  ACCUMULATES.
 */
int pn2_group_points(int B, int C, int N, int npoints, int nsample,
                     const float *points, const int *idx, float *out,
                     void *stream);
int pn2_group_points_grad(int B, int C, int N, int npoints, int nsample,
                          const float *grad_out, const int *idx,
                          float *grad_points, void *stream);

/* ----------------------------------------------------------------- A11 ---
 * three_nn / three_interpolate / three_interpolate_grad
 *   (EXT/include/interpolate.h:6-10, EXT/src/interpolate.cpp:14-99,
 *    EXT/src/interpolate_gpu.cu:9-154)
 * three_nn: unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3) f32 (SQUARED
 *   distances, ascending), idx (B,n,3) i32; strict '<' => earliest index wins.
 * three_interpolate: points (B,C,m), idx (B,n,3), weight (B,n,3) -> out (B,C,n).
 * grad: grad_out (B,C,n) -> grad_points (B,C,m)  ACCUMULATES.
 */
int pn2_three_nn(int B, int n, int m, const float *unknown, const float *known,
                 float *dist2, int *idx, void *stream);
int pn2_three_interpolate(int B, int C, int m, int n, const float *points,
                          const int *idx, const float *weight, float *out,
                          void *stream);
int pn2_three_interpolate_grad(int B, int C, int n, int m,
                               const float *grad_out, const int *idx,
                               const float *weight, float *grad_points,
                               void *stream);

/* ---------------------------------------------------------- fused extras ---
 * Point-major ("channels-last") layout extras used by the fast path.  They
 * replace python-level compositions of the reference and have no native
 * counterpart there.
 *
 * pn2_group_concat_rows: the whole of QueryAndGroup.forward after the ball
 *   query (OPS/pointnet2_utils.py:317-335 and, with `normalize` != 0, the
 *   `grouped_xyz /= radius` of GF3D/pointnet2/pointnet2_utils.py:343-344) in
 *   one pass:  out[b,j,s,0:3]   = (xyz[b,idx[b,j,s]] - new_xyz[b,j]) (/ radius)
 *              out[b,j,s,3:3+C] = feats[b,idx[b,j,s],0:C]
 *   xyz (B,N,3); new_xyz (B,m,3); feats (B,N,C) point-major or NULL (C=0);
 *   idx (B,m,ns); out (B,m,ns,Cx+C) with Cx = use_xyz ? 3 : 0.
 *   `normalize` != 0 divides the relative xyz by `radius` (IEEE division).
 * pn2_group_rows_grad: gradient of the feature part:
 *   grad_out (B,m,ns,ldg) rows, columns [col0, col0+C) -> grad_feats (B,N,C)
 *   ACCUMULATES (fp32 atomics; tolerance-tested, not bitwise).
 */
int pn2_group_concat_rows(int B, int N, int m, int ns, int C, int use_xyz,
                          int normalize, float radius, const float *xyz,
                          const float *new_xyz, const float *feats,
                          const int *idx, float *out, void *stream);
int pn2_group_rows_grad(int B, int N, int m, int ns, int C, int ldg, int col0,
                        const float *grad_out, const int *idx,
                        float *grad_feats, void *stream);

/* The FIRST conv of an SA stack applied before the grouping (round 4, csrc/group_lift.hip).  A 1x1 convolution is linear
 * and the grouping only copies:  W [rel | f[idx]] = Wx rel + (Wf f)[idx]  — the feature part of the product is taken once
 * per point (B N rows) instead of once per (centre, sample) position (B m ns rows) and the grouped (B,m,ns,3+C) tensor of
 * QueryAndGroup (OPS/pointnet2_utils.py:317-328; group_points_kernel EXT/src/group_points_gpu.cu:8-28) never exists.
 *   pn2_group_lift_rows:      Y[(b m + j) ns + s][0:N0] = P[b, idx[b,j,s]][0:N0] + Wx rel[b,j,s],
 *       rel = (xyz[b, idx] - new_xyz[b, j]) (/ radius when `normalize`), P (B N, N0) = f Wf^T, Wx (N0, 3);
 *       stats (2, N0) f64 (may be NULL) += column sums of Y and Y^2 (the layer's BatchNorm batch statistics).
 *   pn2_group_lift_rows_grad: behind BatchNorm dL/dy0[r] = c1 g[r] + c2 y0[r] + c3 (consts (3, N0); G = the masked gradient
 *       the layer above left; y0[r] = P[point] + Wx rel[r] is affine in the row, so it is never read or recomputed: the walk
 *       sums g, rel and g rel^T and the constants enter once per point).  Through the inverse index (ptr / refs of
 *       pn2_group_inverse_index):  S (B N, N0) = sum of dL/dy0 over the rows that gathered each point (every row written);
 *       acc (3 N0 + 9) = [ dWx (N0, 3) = sum_r dL/dy0[r] rel[r]^T MINUS its last term c2 Wx RR | RR (3, 3) = sum_r rel[r] rel[r]^T ]
 *       so that the caller can add that term (written, not accumulated: per-workgroup partials summed in a fixed order).  The caller finishes with GEMMs over B N rows:
 *       dL/df = S Wf, dWf = S^T f.  Points gathered by more than 192 rows (low indices under large radii) are walked by 16 waves
 *       each in a second launch and ADDED (their S is not bit-reproducible).  `workspace`: pn2_group_lift_rows_grad_workspace_bytes.
 *       Replaces the first layer's M-row dgrad / wgrad and group_points_grad_kernel (EXT/src/group_points_gpu.cu:44-75).
 * N0 a multiple of 4 in [16, 256] (pn2_group_lift_supported); P, Y, G, S, consts 16-byte aligned.
 * Algorithmic bytes: forward B (4 m ns + 12 N + 12 m + 4 N0 N + 4 N0 m ns); backward 8 M + 4 M N0 + 4 B N (2 N0 + 4). */
int pn2_group_lift_supported(int N0);
int pn2_group_lift_rows(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                        const float *new_xyz, const int *idx, const float *P, const float *Wx, float *Y, double *stats,
                        void *stream);
int pn2_group_lift_rows_grad(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                             const float *new_xyz, const float *G, const float *P, const float *Wx, const float *consts,
                             const int *ptr, const int *refs, float *S, float *acc, void *workspace,
                             size_t workspace_bytes, void *stream);
/* The lifted layer's weight W (N0, 3 + C) in the pieces the kernels take — Wx (N0, 3) = W[:, :3], Wf (N0, C) = W[:, 3:],
 * WfT (C, N0) = Wf^T (the input-gradient GEMM's weight) — in one launch, and its gradient in one piece:
 * dW (N0, 3 + C) = [acc[:3 N0] + diag(c2) Wx RR | dWf] with acc = pn2_group_lift_rows_grad's (3 N0 + 9) result (RR = its last
 * nine floats) and c2 = row 1 of the BatchNorm-backward constants.  Both replace strided torch copies / a 3 x 3 vendor GEMM +
 * addcmul + cat around the first Conv2d of OPS/pointnet2_modules.py:9-19. */
int pn2_lift_split_weight(int N0, int C, const float *W, float *Wx, float *Wf, float *WfT, void *stream);
int pn2_lift_dw_assemble(int N0, int C, const float *acc, const float *Wx, const float *c2, const float *dWf, float *dW,
                         void *stream);

/* The lifted first layer WITHOUT its output tensor (round 5; csrc/group_lift.hip, csrc/mlp_gemm.hip PRO_LIFT / EPI_MASKL).
 * With the coordinate term of the first Conv2d (OPS/pointnet2_modules.py:9-19 over OPS/pointnet2_utils.py:317-328's
 * [rel | features]) split between the point and the centre, y0[b, j, s] = Pq[b, idx[b, j, s]] - Q[b, j]:
 *   pn2_lift_points       Pq (B N, N0) = P + Wx x / r (r = radius when `normalize`, else 1), Q (B m, N0) = Wx c / r;
 *   pn2_group_lift_stats  stats (2, N0) += column sums of y0 and y0^2 (BatchNorm batch statistics), gidx (B m ns) = b N + idx
 *                         (the row of Pq every grouped row reads);
 *   pn2_mlp_gemm_lift     Y (M, N) = relu(bn_0(y0)) W^T with the column sums of Y, Y^2 — the layer ABOVE, fin0 (4, K) the lifted
 *                         layer's mean | rstd | scale | shift; K = N0;
 *   pn2_mlp_wgrad_lift    dW (N, K) += (c1 G + c2 Yl + c3)^T relu(bn_0(y0))     (pn2_mlp_wgrad with the activation re-formed);
 *   pn2_mlp_dgrad_lift    Gout (M, K) = [(c1 G + c2 Yl + c3) Wt^T] masked by bn_0(y0) > 0, sums (2, K) += column sums of Gout and
 *                         Gout yhat_0 (pn2_mlp_gemm PRO_GY / EPI_MASK with Yprev re-formed); Wt (K, N) rows.
 * The (M, N0) tensor y0 is never stored.  Preconditions (pn2_mlp_lift_supported): 64 <= K <= 2048, N <= 256 (wgrad: <= 128),
 * ns a power of two in [16, 128], M % ns == 0, Pq below 1 GiB.  Same real numbers as pn2_group_lift_rows, rounded in another
 * order (tests: 1e-4 against the oracle like every MLP kernel). */
int pn2_mlp_lift_supported(int K, int N, int ns);
int pn2_lift_points(int B, int N, int m, int N0, int normalize, float radius, const float *xyz, const float *new_xyz,
                    const float *P, const float *Wx, float *Pq, float *Q, void *stream);
int pn2_group_lift_stats(int B, int N, int m, int ns, int N0, const int *idx, const float *Pq, const float *Q, int *gidx,
                         double *stats, void *stream);
int pn2_mlp_gemm_lift(long long M, int K, int N, long long lrows, const float *Pq, const int *gidx, const float *Q, int ns,
                      const float *fin0, const float *W, float *Y, double *stats, void *stream);
int pn2_mlp_wgrad_lift(long long M, int N, int K, long long lrows, const float *G, const float *Yl, const float *consts,
                       const float *Pq, const int *gidx, const float *Q, int ns, const float *a_fin, float *dW, void *stream);
int pn2_mlp_dgrad_lift(long long M, int K, int N, long long lrows, const float *G, const float *Yl, const float *consts,
                       const float *Wt, float *Gout, double *sums, const float *Pq, const int *gidx, const float *Q, int ns,
                       const float *e_fin, void *stream);

size_t pn2_group_lift_rows_grad_workspace_bytes(int B, int N, int m, int ns, int N0);
/* The same pair for the mixed-precision stacks (round 4): Y (B m ns, N0) bf16 (rounded to nearest even; `stats` are the
 * column sums of the rounded values, the convention of pn2_mlp_gemm_bf16) and G (M, N0) bf16 (what pn2_mlp_bwd_bf16 /
 * pn2_mlp_gemm_bf16 leave for the layer below); P, S, every sum and the weight-gradient terms stay fp32. */
int pn2_group_lift_rows_bf16(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                             const float *new_xyz, const int *idx, const float *P, const float *Wx, void *Y, double *stats,
                             void *stream);
int pn2_group_lift_rows_grad_bf16(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                                  const float *new_xyz, const void *G, const float *P, const float *Wx, const float *consts,
                                  const int *ptr, const int *refs, float *S, float *acc, void *workspace,
                                  size_t workspace_bytes, void *stream);
/* ... and for the S scans of a batch in ONE launch (segment-table stacks): grid.y = scan, every scan runs exactly as its own
 * single-scan call would (its clouds, the grid its centre / point count gives, its (2, N0) block of `stats` resp. its (3, N0)
 * `consts` and (3 N0 + 9) row of `acc`), so its sums are bit for bit those of a single-scan launch.  `seg`: (nseg + 1) int64
 * ROW offsets of the scans on the device (multiples of m ns), `max_clouds`: clouds of the largest scan.  G, new_xyz and refs of
 * the backward are the whole batch's tensors (row ids index them in place); `workspace`:
 * pn2_group_lift_rows_grad_seg_workspace_bytes.  y_bf16 / g_bf16: bf16 rows. */
int pn2_group_lift_rows_seg(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                            const float *new_xyz, const int *idx, const float *P, const float *Wx, void *Y, int y_bf16,
                            double *stats, const long long *seg, int nseg, int max_clouds, void *stream);
size_t pn2_group_lift_rows_grad_seg_workspace_bytes(int nseg, int max_clouds, int N, int m, int ns, int N0);
int pn2_group_lift_rows_grad_seg(int B, int N, int m, int ns, int N0, int normalize, float radius, const float *xyz,
                                 const float *new_xyz, const void *G, int g_bf16, const float *P, const float *Wx,
                                 const float *consts, const int *ptr, const int *refs, float *S_out, float *acc,
                                 const long long *seg, int nseg, int max_clouds, void *workspace, size_t workspace_bytes,
                                 void *stream);

/* pn2_rows_max / pn2_rows_max_grad: F.max_pool2d(kernel=[1,ns]) of
 *   OPS/pointnet2_modules.py:67-70 in point-major layout.
 *   x (R,ns,C) -> out (R,C), arg (R,C) i32 (first maximal s, like torch's
 *   max_pool2d argmax);  grad: grad_out (R,C), arg -> grad_x (R,ns,C), every
 *   element written (zeros elsewhere).
 */
int pn2_rows_max(int64_t R, int ns, int C, const float *x, float *out,
                 int *arg, void *stream);
int pn2_rows_max_grad(int64_t R, int ns, int C, const float *grad_out,
                      const int *arg, float *grad_x, void *stream);

/* pn2_three_interpolate_rows / _grad: three_interpolate in point-major layout.
 *   feats (B,m,C), idx (B,n,3), weight (B,n,3) -> out (B,n,ldo) columns
 *   [col0, col0+C) (so the caller can write straight into the concat buffer of
 *   OPS/pointnet2_modules.py:195-198).  grad ACCUMULATES into (B,m,C).
 */
int pn2_three_interpolate_rows(int B, int C, int m, int n, int ldo, int col0,
                               const float *feats, const int *idx,
                               const float *weight, float *out, void *stream);
int pn2_three_interpolate_rows_grad(int B, int C, int m, int n, int ldg,
                                    int col0, const float *grad_out,
                                    const int *idx, const float *weight,
                                    float *grad_feats, void *stream);
/* The same gradient (EXT/src/interpolate_gpu.cu:120-154: three atomicAdds per gradient element) as a gather through the
 * inverse of idx: (ptr, refs) = pn2_group_inverse_index(B, N = m, m' = n, ns = 3, idx) — flat slots (b n + j) 3 + t sorted
 * by (known point, slot).  grad_feats (B, m, C) = sum over a point's slots, in slot order, of weight[slot] *
 * grad_out[slot / 3][col0 : col0 + C]; EVERY output row is written (no zero fill), no atomics, bit-reproducible.
 * C % 4 == 0, grad_feats 16-byte aligned. */
int pn2_three_interpolate_rows_grad_csr(int B, int C, int m, int n, int ldg, int col0, const float *grad_out,
                                        const float *weight, const int *ptr, const int *refs, float *grad_feats,
                                        void *stream);


/* ------------------------------------------------------------------ A10 ---
 * Shared per-point MLP (nn.Conv2d 1x1 bias=False + BatchNorm2d + ReLU per layer,
 * OPS/pointnet2_modules.py:9-19, then F.max_pool2d :67-70) as fused fp32-MFMA
 * kernels on point-major rows.  One layer's forward is ONE kernel:
 *
 *   pn2_mlp_gemm:  Y[M][N] = pro(X)[M][K] * W[N][K]^T, optional epilogue reductions.
 *     pro: 0 none | 1 relu(X*p0[k]+p1[k]) | 2 p0[k]*X + p1[k]*X2 + p2[k]
 *          | 3 like 2 with X gathered from the pooled gradient: (arg[row/ns][k]==row%ns) ? gP[row/ns][k] : 0
 *     epi: 0 none | 1 stats[0][n] += sum Y, stats[1][n] += sum Y^2  (fp64, ACCUMULATES)
 *          | 2 Y *= [Yprev*scale+shift > 0]; stats[0][n] += sum Y; stats[1][n] += sum Y*(Yprev-mean)*rstd
 *     e_fin = [mean | rstd | scale | shift] x N of the layer that produced Yprev.
 *   pn2_mlp_wgrad: dW[N][K] += sum_r gy[r][n] * act[r][k] (ACCUMULATES, fp32 atomics), with
 *     gy = c1[n]*g + c2[n]*Yl + c3[n] (g = G, or gathered from gP/arg when gmode == 3) and
 *     act = X (amode 0) or relu(X*scale[k]+shift[k]) (amode 1, a_fin like e_fin for K columns).  N <= 320.
 *   pn2_bn_finalize: batch sums -> fin = [mean|rstd|gamma*rstd|beta-mean*gamma*rstd]; updates the
 *     running statistics exactly like torch's _BatchNorm (momentum, unbiased variance) when non-NULL.
 *   pn2_bn_bwd_consts: epilogue sums (dbeta, dgamma) -> consts [c1|c2|c3] x N (+ fp32 dgamma, dbeta).
 *   pn2_bn_relu_apply / pn2_bn_relu_bwd_prep: materialised ReLU(BN(y)) and its backward prep
 *     (gpre = gout*[z>0], sums ACCUMULATE) for stacks that must return activations (FP modules).
 *   pn2_bn_relu_rows_max / pn2_pool_bwd_prep: ReLU(BN(y)) fused into the neighbourhood max
 *     (+ first arg-max, + yraw = pre-BN value at the arg-max) and the matching backward
 *     reductions (gPm = gP*[pooled>0]; sums ACCUMULATE).
 */
int pn2_mlp_gemm(long long M, int K, int N, int pro, int epi, const float *X, const float *X2,
                 const float *p0, const float *p1, const float *p2, const int *arg,
                 const float *gP, int ns, const float *W, float *Y, double *stats,
                 const float *Yprev, const float *e_fin, void *stream);
/* One-pass backward of a hidden layer (32 < N,K <= 128): what pn2_mlp_gemm(pro 2|3, epi 2) and
 * pn2_mlp_wgrad(amode 1) compute for the same operands, from ONE read of g / y_l / y_{l-1}:
 *   Gout[M][K] = [y_{l-1}*scale+shift > 0] * (gy * W),  sums += column sums (as epi 2),  dW[N][K] += gy^T * relu(bn(y_{l-1})).
 * a_fin = [mean | rstd | scale | shift] x K of layer l-1.  Replaces the same reference lines as those two
 * (autograd of Conv2d/BatchNorm2d/ReLU, OPS/pointnet2_modules.py:9-19).  pn2_mlp_bwd_fused_supported(N, K) != 0
 * tells whether the shape is covered; otherwise the call returns PN2_EINVAL. */
int pn2_mlp_bwd_fused_supported(int N, int K);
int pn2_mlp_bwd_fused(long long M, int N, int K, int gmode, const float *G, const float *Yl,
                      const float *consts, const int *arg, const float *gP, int ns, const float *W,
                      const float *Yprev, const float *a_fin, float *Gout, double *sums, float *dW,
                      void *stream);
/* First-layer fold (same reference lines): when layer l-1 is the FIRST layer of the stack, its input rows X [M][K0]
 * (K0 <= 8: relative xyz + a few feature columns) need no gradient, and 32 < N, K <= 64, the masked input gradient
 * gz = dL/dz_{l-1} is not stored at all.  The first layer's weight gradient is linear in its BatchNorm-backward
 * constants:  dW_{l-1} = (c1 gz + c2 y_{l-1} + c3)^T X = diag(c1) gz^T X + diag(c2) W_{l-1} (X^T X) + c3 (1^T X).
 *   pn2_mlp_bwd_fused_fold : as pn2_mlp_bwd_fused without Gout, plus P1[K][K0] += gz^T X (caller zero-fills);
 *   pn2_rows_gram          : gram[K0*K0] += X^T X, gram[K0*K0 + k] += column sums of X (fp64, caller zero-fills);
 *   pn2_first_layer_dw     : dW0[N0][K0] from the constants of layer l-1 (N0 = K above), P1, W_{l-1} and gram. */
int pn2_mlp_bwd_fused_fold_supported(int N, int K, int K0);
int pn2_mlp_bwd_fused_fold(long long M, int N, int K, int gmode, const float *G, const float *Yl,
                           const float *consts, const int *arg, const float *gP, int ns, const float *W,
                           const float *Yprev, const float *a_fin, const float *X, int K0, double *sums,
                           float *dW, float *P1, void *stream);
int pn2_rows_gram(long long M, int K0, const float *X, double *gram, void *stream);
int pn2_first_layer_dw(int N, int K0, const float *consts, const float *P1, const float *W0,
                       const double *gram, float *dW0, void *stream);

/* The first layer of a stack WITHOUT its output tensor (round 3).  y_0 = X0 W0^T has K0 <= 8 input columns (the grouped
 * relative xyz + colours of OPS/pointnet2_utils.py:317-328): writing it and reading it back twice costs 3 x 4 M N0 bytes,
 * recomputing an element costs 8 FMAs.
 *   pn2_first_layer_stats : stats[2][N0] (fp64, overwritten) = column sums of y_0 and y_0^2 from gram = pn2_rows_gram(X0)
 *                           (y_0 is linear in X0: W0 (1^T X0) and W0 (X0^T X0) W0^T) -> pn2_bn_finalize as usual;
 *   pn2_mlp_gemm_first    : Y[M][N] = relu(X0 W0^T * scale0 + shift0) W^T, epi = 0 (none) | 1 (column sums for the
 *                           batch statistics of Y); K (= N0) and N <= 128;
 *   pn2_mlp_bwd_fused_fold_first : pn2_mlp_bwd_fused_fold with y_{l-1} recomputed from X and W0 [K][K0] (ReLU mask,
 *                           BatchNorm-backward sums and the wgrad operand); same outputs. */
int pn2_first_layer_stats(int N0, int K0, const float *W0, const double *gram, double *stats, void *stream);
int pn2_mlp_gemm_first_supported(int K0, int K, int N);
int pn2_mlp_gemm_first(long long M, int K0, int K, int N, int epi, const float *X0, const float *W0,
                       const float *scale0, const float *shift0, const float *W, float *Y, double *stats,
                       void *stream);
int pn2_mlp_bwd_fused_fold_first(long long M, int N, int K, int gmode, const float *G, const float *Yl,
                                 const float *consts, const int *arg, const float *gP, int ns, const float *W,
                                 const float *W0, const float *a_fin, const float *X, int K0, double *sums,
                                 float *dW, float *P1, void *stream);
int pn2_mlp_wgrad(long long M, int N, int K, int gmode, int amode, const float *G,
                  const float *Yl, const float *consts, const int *arg, const float *gP, int ns,
                  const float *X, const float *a_fin, float *dW, void *stream);
int pn2_bn_finalize(int N, double count, const double *stats, const float *gamma,
                    const float *beta, float eps, float momentum, float *running_mean,
                    float *running_var, long long *num_batches_tracked /* += 1 if not NULL */, float *fin,
                    void *stream);
/* Running statistics of a block-diagonal batch of S scans trained with per-scan BatchNorm statistics: the S momentum
 * updates of S single-scan steps of the reference (main.py:54-56), in scan order, in one launch.  fins (S,4,C): the
 * (mean | rstd | scale | shift) blocks pn2_bn_finalize wrote for the scans (called with running_mean = NULL);
 * w[s] = m (1-m)^(S-1-s), wu[s] = w[s] n_s / (n_s - 1), decay = (1-m)^S; num_batches_tracked (optional) += S. */
int pn2_bn_running_update(int S, int C, const float *fins, float eps, float decay, const float *w, const float *wu,
                          float *running_mean, float *running_var, long long *num_batches_tracked, void *stream);

/* W != NULL: additionally Wt[K - k0][N] = W[N][K0 + ..]^T (the row-major weight the dgrad call of this layer takes) */
int pn2_bn_bwd_consts(int N, double count, const double *sums, const float *gamma,
                      const float *fin, int use_batch_stats, float *consts, float *dgamma,
                      float *dbeta, const float *W, int K, int k0, float *Wt, void *stream);
int pn2_bn_relu_apply(long long M, int N, const float *y, const float *fin, float *out,
                      void *stream);
int pn2_bn_relu_bwd_prep(long long M, int N, const float *y, const float *gout,
                         const float *fin, float *gpre, double *sums, void *stream);
int pn2_bn_relu_rows_max(long long R, int ns, int C, const float *y, const float *fin,
                         float *out, int *arg, float *yraw, void *stream);
int pn2_pool_bwd_prep(long long R, int C, const float *yraw, const float *pooled,
                      const float *gP, const float *fin, float *gPm, double *sums,
                      void *stream);

/* Max-pooled LAST layer of an SA stack without materialising its (B*npoint*nsample, C_out) output — the grouped
 * per-neighbourhood MLP + max of OPS/pointnet2_modules.py:58-70 (Conv2d 1x1 + BatchNorm2d + ReLU, F.max_pool2d) fused:
 * BatchNorm with a positive scale and ReLU are monotone, so max_s relu(bn(y_s)) = relu(bn(max_s y_s)).
 *   pn2_pool_flip_rows: Wf = diag(sgn) W, sgn[n] = -1 where gamma[n] < 0 (then the MINIMUM of y is wanted) else +1.
 *   pn2_mlp_gemm_pool:  y' = pro(X) Wf^T (pro 0 | 1 as pn2_mlp_gemm), never stored; stats[0][n] += sgn[n] sum y',
 *     stats[1][n] += sum y'^2 (fp64, ACCUMULATES); per partial group of psz = min(ns, 32) consecutive rows and column:
 *     pmax[M/psz][N] = max y', parg[M/psz][N] = row of the first maximum.  ns in {16, 32, 64, 128}, M % ns == 0.
 *   pn2_pool_finalize:  combines the ns/psz partial groups (first maximum wins) -> arg[R][C]; yraw = sgn * max
 *     (= the raw pre-BN value at the arg-max), out = relu(yraw * scale + shift) with fin = [mean|rstd|scale|shift].
 *   Outputs are what pn2_bn_relu_rows_max produces from a materialised y (equal values; among rows that tie AFTER
 *   BatchNorm's rounding the arg-max is the row of the largest raw value instead of the first of them). */
int pn2_pool_flip_rows(int N, int K, const float *W, const float *gamma, float *Wf, float *sgn, void *stream);
int pn2_mlp_gemm_pool(long long M, int K, int N, int pro, const float *X, const float *p0, const float *p1,
                      const float *Wf, const float *sgn, int ns, double *stats, float *pmax, int *parg,
                      void *stream);
int pn2_pool_finalize(long long R, int C, int ns, const float *pmax, const int *parg, const float *fin,
                      const float *sgn, float *out, int *arg, float *yraw, void *stream);

/* Backward of that layer in Gram form (csrc/pool_bwd.hip): with y_L = a W^T never stored, a = relu(bn(y_{L-1})),
 *   dL/da = a (W^T diag(c2) W) + 1 (W^T c3)^T + S,   S[row] = sum over the columns n whose arg-max is `row` of
 *           gPm[g][n] c1[n] W[n][:]
 *   dW    = diag(c1) T + diag(c2) W (a^T a) + c3 (1^T a),   T[n][:] = sum_g gPm[g][n] a[arg-max row of (g, n)][:]
 * — the autograd of Conv2d 1x1 + BatchNorm2d (batch statistics) + ReLU + F.max_pool2d (OPS/pointnet2_modules.py:58-70)
 * for the pooled layer from ONE pass over y_{L-1}: 2 K^2 MACs per row instead of 4 N K, no (M, N) tensor.
 *   consts = [c1|c2|c3] x N of this layer (pn2_bn_bwd_consts on the sums of pn2_pool_bwd_prep); arg, gPm [M/ns][N];
 *   Yp [M][K] and fin_p = [mean|rstd|scale|shift] x K of the layer below.  Writes Gout [M][K] = dL/dz_{L-1} (ReLU
 *   mask applied) and dW [N][K]; sums [2][K] += (sum Gout, sum Gout * yhat_{L-1}) (fp64, ACCUMULATES).
 *   K in {64, 128}, N <= 256, ns in {16, 32, 64, 128} (pn2_pool_bwd_supported); workspace >= pn2_pool_bwd_workspace_bytes. */
int pn2_pool_bwd_supported(int N, int K, int ns);
size_t pn2_pool_bwd_workspace_bytes(long long M, int N, int K);
int pn2_pool_bwd(long long M, int N, int K, int ns, const float *Yp, const float *fin_p, const float *W,
                 const float *consts, const int *arg, const float *gPm, float *Gout, double *sums, float *dW,
                 void *workspace, size_t workspace_bytes, void *stream);

/* ----------------------------------------------------- A10, mixed precision ---
 * bf16 variants of the shared-MLP kernels.  The reference trains under 16-bit AMP (scene_graph_prediction/main.py:64
 * `precision=16`; GroupingOperation forces fp32, OPS/pointnet2_utils.py:198): 1x1 convolutions in half precision,
 * fp32 master weights, fp32 BatchNorm statistics.  Here the activations BETWEEN the layers of a stack (raw pre-BN
 * outputs y_l, gradients dL/dz_l) are stored as bf16, the products run on v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation, statistics / constants / weights / weight gradients stay fp32 (fp64 sums).  bf16 tensors are passed as
 * `void *`, row pitches (ldx / ldy, in elements) must be multiples of 8 and the base pointers 16-byte aligned.
 *
 *   pn2_mlp_gemm_bf16: Y[M][N] (pitch ldy) = pro(X)[M][K] (pitch ldx) * W[N][K]^T, pro / epi as pn2_mlp_gemm.
 *     x_f32: X is fp32 rows of any pitch (pro must be 0);  y_f32: Y is fp32 (epi must be 0: the input gradient that
 *     leaves the stack).  Columns K..ldx-1 of a bf16 X must be finite (they meet zero weights).  N <= 320.
 *   pn2_mlp_wgrad_bf16: dW[N][K] fp32 += gy^T * act, gy / act formed on the fly as in pn2_mlp_wgrad; G, Yl bf16 [M][N]
 *     (N % 8 == 0), X = bf16 y_{l-1} (amode 1) or the stack's input rows (amode 0; fp32 of any pitch when x_f32).
 *   pn2_bn_relu_apply_bf16 / pn2_bn_relu_bwd_prep_bf16 / pn2_bn_relu_rows_max_bf16: as the fp32 helpers with y bf16
 *     (gpre bf16); pooled outputs, arg-max and raw arg-max values stay fp32 / int32 (C % 2 == 0).
 *   pn2_group_concat_rows_bf16: pn2_group_concat_rows writing bf16 rows of pitch ldo (% 8 == 0), pad columns zeroed.
 */
/* First-layer fold on the bf16 path (round 3): pn2_mlp_bwd_bf16 for the layer above a stack's FIRST layer whose input rows
 * X ([M][8] bf16: K0 <= 8 columns, zero padded — what pn2_group_concat_rows_bf16 writes) need no gradient: the masked input
 * gradient is not stored, P1 [K][K0] += its product with X is reduced instead; pn2_rows_gram_bf16 + pn2_first_layer_dw then
 * give the first layer's weight gradient without a pass over g and y_0.  N, K in {32, 64}. */
int pn2_mlp_bwd_bf16_fold_supported(int N, int K, int K0);
int pn2_mlp_bwd_bf16_fold(long long M, int N, int K, int gmode, const void *G, const void *Yl, const float *consts,
                          const int *arg, const float *gP, int ns, const float *Wt, const void *Yprev,
                          const float *a_fin, const void *X, int K0, double *sums, float *dW, float *P1, void *stream);
int pn2_rows_gram_bf16(long long M, int K0, const void *X, double *gram, void *stream);
int pn2_mlp_gemm_bf16(long long M, int K, int N, int pro, int epi, int x_f32, int y_f32, int ldx, int ldy,
                      const void *X, const void *X2, const float *p0, const float *p1, const float *p2,
                      const int *arg, const float *gP, int ns, const float *W, void *Y, double *stats,
                      const void *Yprev, const float *e_fin, void *stream);
int pn2_mlp_wgrad_bf16(long long M, int N, int K, int gmode, int amode, int x_f32, int ldx, const void *G,
                       const void *Yl, const float *consts, const int *arg, const float *gP, int ns,
                       const void *X, const float *a_fin, float *dW, void *stream);
/* One-pass backward of a hidden layer in bf16 (N, K multiples of 32 up to 128; pn2_mlp_bwd_bf16_supported): what
 * pn2_mlp_gemm_bf16(pro 2|3, epi 2) + pn2_mlp_wgrad_bf16(amode 1) compute, from ONE read of g / y_l / y_{l-1}.
 * Wt = the layer's weights transposed, fp32 [K][N] (as pn2_bn_bwd_consts emits them); dW and sums ACCUMULATE. */
int pn2_mlp_bwd_bf16_supported(int N, int K);
int pn2_mlp_bwd_bf16(long long M, int N, int K, int gmode, const void *G, const void *Yl, const float *consts,
                     const int *arg, const float *gP, int ns, const float *Wt, const void *Yprev,
                     const float *a_fin, void *Gout, double *sums, float *dW, void *stream);
/* The max-pooled LAST layer of a bf16 stack without its (M, N) output (round 5; the bf16 counterpart of pn2_mlp_gemm_pool /
 * pn2_pool_bwd — Conv2d 1x1 + BatchNorm2d + ReLU + F.max_pool2d of OPS/pointnet2_modules.py:9-19,67-70 under the reference's
 * 16-bit AMP, scene_graph_prediction/main.py:64):
 *   pn2_mlp_gemm_pool_bf16  X (M, ldx) bf16 = y_{L-1}, p0 / p1 its BatchNorm scale / shift, Wf (N, K) fp32 with the rows of
 *                           negative-gamma columns negated (pn2_pool_flip_rows, sgn (N)) -> pmax / parg (M / min(ns, 32), N):
 *                           maximum of the fp32 accumulators per partial group and its row; stats (2, N) += column sums of
 *                           y_L, y_L^2 (of the accumulators: nothing is stored, so nothing is rounded).  pn2_pool_finalize
 *                           turns pmax / parg into the pooled activations, the arg-max rows and their raw values;
 *   pn2_mlp_bwd_bf16_pool   pn2_mlp_bwd_bf16 with gmode 3 for that layer: y_L is RE-FORMED from y_{L-1} (Yprev) and Wt on the
 *                           matrix pipe inside the kernel — the forward's own product, bit for bit — instead of read.
 * N in {64, 128}; K <= 128 (backward: 32, 64 or 128); ns in {16, 32, 64, 128}, M % ns == 0. */
/* ... and the FIRST layer of a bf16 stack without its output (the counterpart of pn2_mlp_gemm_first / pn2_mlp_bwd_fused_fold_first):
 *   pn2_mlp_gemm_first_bf16     Y (M, N) bf16 = relu(bn_0(X0 W0^T)) W^T with the column sums of the rounded Y, Y^2: the second layer
 *                               with the first one re-formed from its input rows X0 (M, 8) bf16 (K0 <= 8 real columns), W0 (K, K0),
 *                               fin0 (4, K); y_0 is never stored (statistics: pn2_rows_gram_bf16 + pn2_first_layer_stats);
 *   pn2_mlp_bwd_bf16_fold_first pn2_mlp_bwd_bf16_fold with y_0 re-formed per tile from X and W0 instead of read (Yprev). */
int pn2_mlp_gemm_first_bf16_supported(int K0, int K, int N);
int pn2_mlp_gemm_first_bf16(long long M, int K0, int K, int N, const void *X0, const float *W0, const float *fin0,
                            const float *W, void *Y, double *stats, void *stream);
int pn2_mlp_bwd_bf16_fold_first(long long M, int N, int K, int gmode, const void *G, const void *Yl, const float *consts,
                                const int *arg, const float *gP, int ns, const float *Wt, const float *W0, const float *a_fin,
                                const void *X, int K0, double *sums, float *dW, float *P1, void *stream);
int pn2_mlp_gemm_pool_bf16_supported(int K, int N, int ns);
int pn2_mlp_gemm_pool_bf16(long long M, int K, int N, int ldx, const void *X, const float *p0, const float *p1,
                           const float *Wf, const float *sgn, int ns, double *stats, float *pmax, int *parg, void *stream);
int pn2_mlp_bwd_bf16_pool_supported(int N, int K);
int pn2_mlp_bwd_bf16_pool(long long M, int N, int K, const float *consts, const int *arg, const float *gP, int ns,
                          const float *Wt, const void *Yprev, const float *a_fin, void *Gout, double *sums, float *dW,
                          void *stream);
int pn2_bn_relu_apply_bf16(long long M, int N, const void *y, const float *fin, float *out, void *stream);
int pn2_bn_relu_bwd_prep_bf16(long long M, int N, const void *y, const float *gout, const float *fin,
                              void *gpre, double *sums, void *stream);
int pn2_bn_relu_rows_max_bf16(long long R, int ns, int C, const void *y, const float *fin, float *out,
                              int *arg, float *yraw, void *stream);
int pn2_group_concat_rows_bf16(int B, int N, int m, int ns, int C, int use_xyz, int normalize, float radius,
                               int ldo, const float *xyz, const float *new_xyz, const float *feats,
                               const int *idx, void *out, void *stream);

/* ------------------------------------------------ batched scans, per-scan BatchNorm statistics (segment table) ---
 * The reference trains on ONE scan per step (scene_graph_prediction/main.py:54-56, DataLoader(batch_size=1)): every
 * training-mode BatchNorm of the encoders (OPS/pointnet2_modules.py:9-19 shared MLPs) sees the clouds of one scan.  A
 * block-diagonal batch of S scans keeps that arithmetic when the statistics are taken PER SCAN; these entry points do so
 * at the launch count of ONE call instead of S: the M rows of a stack are S scans, scan s = rows [seg[s], seg[s+1])
 * (`seg`: S + 1 row offsets on the device, multiples of `ns` wherever a pooled operand is involved; `seg_max` = rows of the
 * longest scan, sizes the grid), grid.y (z for the weight gradient) walks the scans, and every per-channel operand is an
 * array of per-scan blocks: statistics / sums (S,2,C) fp64, finalize blocks fin (S,4,C), backward constants (S,3,C).  Row
 * tiles never straddle two scans, so each scan's sums, outputs and gradients are those of its own call; weight-gradient
 * accumulators (dW, and dgamma / dbeta of pn2_bn_bwd_consts_seg) receive the SUM over the scans — the 1/S of the mean loss
 * arrives with the incoming gradient.
 *   pn2_mlp_gemm_bf16_seg: p0 / p1 / p2 point at scan 0's vector, scan s's lies `pstride` floats further (4K for the
 *                          scale / shift rows of fin, 3K for the rows of the constants); stats (S,2,N), e_fin (S,4,N).
 *   pn2_mlp_wgrad_bf16_seg / pn2_mlp_bwd_bf16_seg: consts (S,3,N), a_fin (S,4,K), sums (S,2,K).
 *   pn2_bn_relu_rows_max_bf16_seg / pn2_pool_bwd_prep_seg: seg in rows of the UN-pooled tensor, fin (S,4,C), sums (S,2,C).
 *   pn2_bn_finalize_seg: count of scan s = seg[s+1] - seg[s]; running_mean / running_var (optional) receive the S momentum
 *                        updates in scan order — what S calls of pn2_bn_finalize do —, num_batches_tracked += S.
 *   pn2_bn_bwd_consts_seg: counts likewise; W / Wt as in pn2_bn_bwd_consts. */
int pn2_mlp_gemm_bf16_seg(long long M, int K, int N, int pro, int epi, int x_f32, int y_f32, int ldx, int ldy,
                          const void *X, const void *X2, const float *p0, const float *p1, const float *p2, int pstride,
                          const int *arg, const float *gP, int ns, const float *W, void *Y, double *stats,
                          const void *Yprev, const float *e_fin, const long long *seg, int nseg, long long seg_max,
                          void *stream);
int pn2_mlp_wgrad_bf16_seg(long long M, int N, int K, int gmode, int amode, int x_f32, int ldx, const void *G,
                           const void *Yl, const float *consts, const int *arg, const float *gP, int ns, const void *X,
                           const float *a_fin, float *dW, const long long *seg, int nseg, long long seg_max, void *stream);
int pn2_mlp_bwd_bf16_seg(long long M, int N, int K, int gmode, const void *G, const void *Yl, const float *consts,
                         const int *arg, const float *gP, int ns, const float *Wt, const void *Yprev, const float *a_fin,
                         void *Gout, double *sums, float *dW, const long long *seg, int nseg, long long seg_max, void *stream);
int pn2_bn_relu_rows_max_bf16_seg(long long R, int ns, int C, const void *y, const float *fin, float *out, int *arg,
                                  float *yraw, const long long *seg, int nseg, long long seg_max, void *stream);
int pn2_pool_bwd_prep_seg(long long R, int C, const float *yraw, const float *pooled, const float *gP, const float *fin,
                          float *gPm, double *sums, const long long *seg, int nseg, long long seg_max, int ns, void *stream);
int pn2_bn_finalize_seg(int S, int N, const long long *seg, const double *stats, const float *gamma, const float *beta,
                        float eps, float momentum, float *running_mean, float *running_var,
                        long long *num_batches_tracked, float *fin, void *stream);
int pn2_bn_bwd_consts_seg(int S, int N, const long long *seg, const double *sums, const float *gamma, const float *fin,
                          int use_batch_stats, float *consts, float *dgamma, float *dbeta, const float *W, int K, int k0,
                          float *Wt, void *stream);

/* ----------------------------------------------------------------- (f)3 ---
 * Per-object / per-pair crops of a fused scan on the GPU: the step in front of the hot path
 *   (SGH/dataset/data_preparation_utils.py:110-125 object crops, :173-224 pair crops, :37-49 re-sampling, :12-18
 *   zero_mean; 81 crops per scan on the host with open3d in the reference).
 * points (P, ld) fp32 rows (xyz first, ld >= 3), masks (P) i32 object id 1..n_obj (0 = context),
 * edges (2, E) i32 ordered pairs of object indices (0-based).  Crops 0..n_obj-1 are the objects, then the E pairs.
 *   pn2_prep_object_boxes     : boxes (n_obj, 6) = [min xyz - padding | max xyz + padding]; keys = n_obj*6 u32 scratch.
 *   pn2_prep_chunk_counts     : counts (crops, pn2_prep_num_chunks(P)) i32 members per 1024-point chunk
 *                               (object: masks == id; pair: strictly inside the union of the two boxes, all points).
 *   [caller: prefix (crops, chunks+1) i64 = exclusive prefix sums of counts along the chunks]
 *   pn2_prep_select           : sel (n_obj*t_obj + E*t_rel) i32 scan indices: members >= target -> one distinct member
 *                               per stratum of the member order; fewer -> draws with replacement; counter-based
 *                               generator keyed on (seed, crop, slot); -1 for an empty crop.
 *   pn2_prep_gather_normalise : obj_out (n_obj, t_obj, ld), rel_out (E, t_rel, ld + 1) with the mask channel
 *                               1 = subject / 2 = object / 0 = context last, xyz centred on the mean of the crop and
 *                               divided by its largest norm (zero_mean).
 * Deterministic parts restate the reference exactly; the sampler replaces open3d's voxel trace + numpy's global
 * generator (not reproducible) by a seeded one with the same two regimes.
 */
/* One rung of the reference's voxel ladder (data_preparation_utils.py:37-49, open3d voxel_down_sample_and_trace):
 * keys[i] = voxel (13 bits per axis) << 3 | octant of point i for voxels of edge `size` anchored at min_bound - size/2,
 * double-precision arithmetic like open3d; points with equal keys share a slot of the trace matrix (the slot keeps the
 * LAST of them).  pts (n, ld) fp32 rows, min_bound (3) fp32 = per-axis minimum of the same rows. */
int pn2_prep_voxel_keys(int n, int ld, const float *pts, const float *min_bound, double size, long long *keys,
                        void *stream);
int pn2_prep_num_chunks(int P);
int pn2_prep_object_boxes(int P, int ld, int n_obj, float padding, const float *points, const int *masks,
                          unsigned *keys, float *boxes, void *stream);
int pn2_prep_chunk_counts(int P, int ld, int n_obj, int E, const float *points, const int *masks,
                          const float *boxes, const int *edges, int *counts, void *stream);
int pn2_prep_select(int P, int ld, int n_obj, int E, int t_obj, int t_rel, unsigned seed, const float *points,
                    const int *masks, const float *boxes, const int *edges, const long long *prefix,
                    int *sel, void *stream);
int pn2_prep_gather_normalise(int ld, int n_obj, int E, int t_obj, int t_rel, const float *points,
                              const int *masks, const int *edges, const int *sel, float *obj_out,
                              float *rel_out, void *stream);

/* ------------------------------------------------------------------ A12 ---
 * TripletGCN edge primitives.  Replace torch_geometric 2.0.2
 * MessagePassing.__lift__ (x.index_select(-2, edge_index[i])) and
 * torch_scatter 2.0.9 scatter(reduce='add'); call sites
 * scene_graph_prediction/scene_graph_helpers/model/gcns/network_TripletGCN.py:41,57.
 *
 * pn2_gather_rows: x (N,H), index (E) i64 -> out (E,ldo) columns
 *   [col0,col0+H)  (lets the caller build cat[x_i, e, x_j] in place, :46).
 * pn2_scatter_add_rows: src (E,lds) columns [col0,col0+H), index (E) i64 ->
 *   out (N,H)  ACCUMULATES with fp32 atomics (order-nondeterministic).
 * pn2_segment_sum_rows: deterministic alternative: `order` (E) i64 is a STABLE
 *   arg-sort of `index`, `rowptr` (N+1) i64 its CSR offsets;
 *   out[n] = sum over e in order[rowptr[n]:rowptr[n+1]] of src[e] taken in that
 *   (= original edge) order, i.e. bit-identical to a sequential CPU
 *   scatter_add_.  Every row of out is written.
 * Indices out of [0,N) are the caller's error (checked on the python side).
 */
int pn2_gather_rows(int64_t E, int H, int64_t N, int ldo, int col0,
                    const float *x, const int64_t *index, float *out,
                    void *stream);
int pn2_scatter_add_rows(int64_t E, int H, int64_t N, int lds, int col0,
                         const float *src, const int64_t *index, float *out,
                         void *stream);
int pn2_segment_sum_rows(int64_t E, int H, int64_t N, int lds, int col0,
                         const float *src, const int64_t *order,
                         const int64_t *rowptr, float *out, void *stream);

/* The feature-gradient scatter of QueryAndGroup as a gather (csrc/group_csr.hip; replaces the atomic form of
 * group_points_grad_kernel, src/group_points_gpu.cu:44-75, where a prefetched neighbourhood index is available).
 *   pn2_group_inverse_index : idx (B,m,ns) int32 -> refs (B*m*ns) = row ids (b*m + j)*ns + s sorted by
 *       (b*N + idx[row], row) [ONE launch: a stable counting sort in LDS, one workgroup per (cloud, slice of its points), idx
 *       values clamped to [0, N); a stable radix sort when B N m ns > 1.2e11 or the 144 KB of LDS are refused — the workspace
 *       query tells which] and ptr (B*N + 1): refs[ptr[p] : ptr[p+1]] are the rows that gathered
 *       point p = b*N + n.  `workspace`: 256-byte aligned device scratch of at least
 *       pn2_group_inverse_index_workspace_bytes(B, N, m, ns) bytes (PN2_ENOSPC if smaller); B*m*ns and B*N < 2^31.
 *   pn2_group_rows_grad_csr : grad_feats (B,N,C) = sum over refs of grad_out[row, col0 : col0+C]  (grad_out (rows, ldg)
 *       fp32); every output row is WRITTEN (zeros for unreferenced points), the summation order is fixed by the sort, so
 *       the result is bit-reproducible.
 */
/* bf16 gradient rows (mixed-precision stack): pn2_group_rows_grad_bf16 is pn2_group_rows_grad over bf16 `grad_out`
 * (C, ldg, col0 even, 4-byte aligned base), pn2_group_rows_grad_csr_bf16 the per-point sum; the accumulation and
 * grad_feats stay fp32. */
int pn2_group_rows_grad_bf16(int B, int N, int m, int ns, int C, int ldg, int col0, const void *grad_out,
                             const int *idx, float *grad_feats, void *stream);
int pn2_group_rows_grad_csr_bf16(int B, int N, int C, int ldg, int col0, int64_t rows, const void *grad_out,
                                 const int *ptr, const int *refs, float *grad_feats, void *stream);
size_t pn2_group_inverse_index_workspace_bytes(int B, int N, int m, int ns);
int pn2_group_inverse_index(int B, int N, int m, int ns, const int *idx, int *ptr, int *refs, void *workspace,
                            size_t workspace_bytes, void *stream);
int pn2_group_rows_grad_csr(int B, int N, int C, int ldg, int col0, int64_t rows, const float *grad_out,
                            const int *ptr, const int *refs, float *grad_feats, void *stream);

/* Triplet message without the (E, 2*dn+de) concatenation (network_TripletGCN.py:45-52):
 *   pn2_gather2_add_rows : q (E,H) += p[ia[e], cola:cola+H] + p[ib[e], colb:colb+H].  nn1's first Linear is applied to the
 *     NODES once (p = x [Wa | Wc]^T, (N, 2H)) and to the edge features (q = e Wb^T + b) and only the products are lifted:
 *     W [x_i | e | x_j] = Wa x_i + Wb e + Wc x_j — E/N times fewer FLOPs in the node part, no concat buffer.
 *   pn2_segment_sum2_rows: pn2_segment_sum_rows over src[e, col0:col0+H] + src[e, col1:col1+H] (node message = first +
 *     last block of nn1's output, :50) without materialising the sum; col1 < 0 = plain segment sum.
 * H and the column offsets must be multiples of 4 for pn2_gather2_add_rows (16-byte lanes).
 */
int pn2_gather2_add_rows(int64_t E, int H, int64_t N, int ldp, int cola, int colb, const float *p,
                         const int64_t *ia, const int64_t *ib, float *q, void *stream);
int pn2_segment_sum2_rows(int64_t E, int H, int64_t N, int lds, int col0, int col1, const float *src,
                          const int64_t *order, const int64_t *rowptr, float *out, void *stream);

/* Per-segment BatchNorm1d (+ optional ReLU) for block-diagonally batched scans.  The reference trains and evaluates
 * one scan per step, and its GCN BatchNorm1d layers are built with track_running_stats=False
 * (network_TripletGCN.py:20), i.e. they ALWAYS normalise with the statistics of the current scan's rows.  When S scans
 * are batched, rows [ptr[s], ptr[s+1]) of x belong to scan s and get their own mean / biased variance:
 *   y[r][c] = [relu]((x[r][col0+c] - mean[s][c]) * rstd[s][c] * gamma[c] + beta[c]),  rstd = 1/sqrt(var + eps).
 * x (R, ldx) columns [col0, col0+C); ptr (S+1) i64; y (R, C); mean / rstd (S, C) are outputs kept for the backward.
 * _grad: grad_out (R, C) w.r.t. y -> grad_x (R, C) and the per-segment partial sums dgamma_part / dbeta_part (S, C)
 * (the caller sums them over S: deterministic).  Replaces S calls of torch's batch_norm on S tiny tensors.
 */
int pn2_segment_bn_rows(int64_t R, int C, int ldx, int col0, int64_t S, const float *x, const int64_t *ptr,
                        const float *gamma, const float *beta, float eps, int relu, float *y, float *mean,
                        float *rstd, void *stream);
int pn2_segment_bn_rows_grad(int64_t R, int C, int ldx, int col0, int64_t S, const float *grad_out,
                             const float *x, const int64_t *ptr, const float *gamma, const float *beta,
                             const float *mean, const float *rstd, int relu, float *grad_x,
                             float *dgamma_part, float *dbeta_part, void *stream);
/* ---- fused TripletGCN blocks (round 4; csrc/gcn_fused.hip) ----------------------------------------------------
 * network_TripletGCN.py:11-58 for scans of <= 128 rows batched block-diagonally: `ptr` (S + 1) int64 row offsets of the
 * scans, BatchNorm1d(track_running_stats=False) statistics per scan (:20), everything a BatchNorm needs local to the
 * workgroup that owns a 32-column tile of a Linear's output of ONE scan.  K, N (and dn, de, dh) multiples of 32;
 * pn2_gcn_fused_supported says whether a layer's dimensions / the longest scan fit.
 *   pn2_gcn_linear         Out (R, N) = [ReLU][BN_scan](A W^T + bias); A rows (R, lda), or (x != NULL) the virtual
 *                          cat[x[dst], e, x[src]] of :46 (K = 2 dn + de) — the concatenation is never built.  With gamma:
 *                          Ypre (R, N) pre-BN values, mean / rstd (S, N) for the backward.
 *   pn2_gcn_linear_grad_w  backward up to the weights: G (R, N) gradient of Out — or (gagg != NULL) the adjoint of split +
 *                          aggregate (:48-58) read in place: [gagg[dst] | gedge | gagg[dst]], N = 2 dh + dE — through the
 *                          ReLU mask and BatchNorm's backward -> Gz (R, N); dW (N, K) += Gz^T A over all R rows (one
 *                          workgroup per 32 x 32 tile, plain read-modify-write: deterministic — do not run two calls on
 *                          the same dW concurrently), dbias (N) += column sums of Gz, dgamma / dbeta (N) += (fp32 atomics
 *                          across scans); all four zeroed by the caller once per step.  Two launches (BatchNorm backward
 *                          per scan, then the product).  gamma NULL: no BatchNorm (relu: Ypre = Out).
 *   pn2_gcn_linear_grad_x  input gradient Gz W: Gin (R, K), or (gx != NULL) scattered through the adjoint of the triplet
 *                          gather: gx (nodes, dn) += columns [0, dn) at dst and [dn + de, K) at src, ge (R, de) = the middle.
 *   pn2_gcn_edge_slice     out (R, de) = [ReLU] h[:, off : off + de]  (the new edge feature, :51).
 * FLOPs 2 R K N each; every operand is read once per 32-column tile (L2-resident at these sizes).
 * PRECONDITION: every scan has at most 128 rows.  `ptr` lives on the device, so the entry points cannot return an error for
 * it: a scan with more rows comes back with EVERY result row NaN (Out / Ypre of pn2_gcn_linear, Gz — and through it the
 * weight gradients — of pn2_gcn_linear_grad_w), never with statistics over its first 128 rows.
 * pn2_gcn_fused_supported(dn, de, dh, longest scan) is the host-side check; the python layer routes longer scans (and
 * batches of more than 32 scans) to the unfused kernels. */
int pn2_gcn_fused_supported(int dn, int de, int dh, int max_rows_per_scan);
int pn2_gcn_linear(long long R, int S, int K, int N, const float *A, int lda, const float *x, const float *e,
                   const long long *dst, const long long *src, int dn, int de, const float *W, const float *bias,
                   const long long *ptr, const float *gamma, const float *beta, float eps, int relu, float *Ypre,
                   float *Out, float *mean, float *rstd, void *stream);
int pn2_gcn_linear_grad_w(long long R, int S, int K, int N, const float *G, const float *gagg, const float *gedge, int dh,
                          int dE, const float *Ypre, const float *mean, const float *rstd, const float *gamma,
                          const float *beta, int relu, const long long *ptr, const float *A, int lda, const float *x,
                          const float *e, const long long *dst, const long long *src, int dn, int de, float *Gz, float *dW,
                          float *dbias, float *dgamma, float *dbeta, void *stream);
int pn2_gcn_linear_grad_x(long long R, int S, int K, int N, const float *Gz, const float *W, const long long *ptr, float *Gin,
                          float *gx, float *ge, const long long *dst, const long long *src, int dn, int de, void *stream);
int pn2_gcn_edge_slice(long long R, int ld, int off, int de, int relu, const float *hrows, float *out, void *stream);

/* One TripletGCN layer per C call (round 4): the launch sequence of network_TripletGCN.py:40-58 issued from C, because a scan-
 * sized layer is bound by the host thread — six (forward) / nine (backward) python -> C round trips cost more than the kernels.
 *   forward : h1 = ReLU BN (cat[x[dst], e, x[src]] W1^T + b1)           (edges, dh)          nn1[0..2]   (:36, 45-47)
 *             h2 = ReLU BN (h1 W2^T + b2)                               (edges, 2 dh + de)   nn1[3..5]
 *             agg[n] = sum over the edges of `order[rowptr[n] : rowptr[n+1]]` of h2[:, :dh] + h2[:, dh+de:]    (:50, 54-58)
 *             e_out = h2[:, dh : dh+de]  [ReLU: relu_out]                                    (:51)
 *             t  = ReLU BN (agg W3^T + b3)                              (nodes, dh)          nn2[0..2]   (:38, 42-43)
 *             out = [ReLU: relu_out] (t W4^T + b4)                      (nodes, dn)          nn2[3]
 *   backward: g_out (nodes, dn), g_e (edges, de) -> gx (nodes, dn; ZERO on entry), ge (edges, de); the 14 parameter gradients
 *             dW1 .. db4 += (zero on entry, like pn2_gcn_linear_grad_w).  `work`: pn2_gcn_layer_backward_workspace_bytes.
 * `dst` = edge_index[1] (x_i: the target row, also the aggregation index, :46, 57), `src` = edge_index[0] (x_j); order /
 * rowptr = the CSR of `dst` in edge order (stable); node_ptr / edge_ptr (S + 1): row offsets of the scans.  h1p / h2p / tp: pre-BatchNorm values,
 * m* / r* (S, width): mean and 1/sqrt(var + eps) per scan — written by the forward, read by the backward.
 * The same preconditions as the block entry points (pn2_gcn_fused_supported; every scan <= 128 rows). */
typedef struct pn2_gcn_layer {
  long long nodes, edges;
  int S, dn, de, dh, relu_out;
  float eps1, eps2, eps3;
  const float *x, *e;
  const long long *dst, *src, *order, *rowptr, *node_ptr, *edge_ptr;
  const float *W1, *b1, *g1, *be1, *W2, *b2, *g2, *be2, *W3, *b3, *g3, *be3, *W4, *b4;
  float *h1, *h1p, *m1, *r1, *h2, *h2p, *m2, *r2, *agg, *e_out, *t, *tp, *m3, *r3, *out;
  const float *g_out, *g_e;
  float *dW1, *db1, *dg1, *dbe1, *dW2, *db2, *dg2, *dbe2, *dW3, *db3, *dg3, *dbe3, *dW4, *db4;
  float *gx, *ge;
  void *work;
} pn2_gcn_layer;
int pn2_gcn_layer_forward(const pn2_gcn_layer *layer, void *stream);
int pn2_gcn_layer_backward(const pn2_gcn_layer *layer, void *stream);
size_t pn2_gcn_layer_backward_workspace_bytes(long long nodes, long long edges, int dn, int de, int dh);

/* Running statistics of a BatchNorm1d WITH running statistics (the classification heads, network_PointNet.py:198-203) after
 * the S per-scan batches of pn2_segment_bn_rows, applied in scan order like S calls of F.batch_norm(training=True):
 * running <- (1 - momentum) running + momentum stat_s, variance unbiased (n_s / (n_s - 1)), num_batches_tracked += S. */
int pn2_segment_bn_running_update(int64_t S, int C, const float *mean, const float *rstd, const int64_t *ptr, float eps,
                                  float momentum, float *running_mean, float *running_var, long long *num_batches_tracked,
                                  void *stream);

/* ----------------------------------------------------------------- (f)4 ---
 * Graphormer pre-processing of the role-prediction task (role_prediction/graphormer/algos.pyx:11-89, called from
 * wrapper.py:39-41), batched over B graphs of n <= 128 nodes, int64 like the reference's numpy arrays:
 *   pn2_floyd_warshall: adjacency (B,n,n) -> dist (B,n,n) hop counts (unreachable = 12) and path (B,n,n) intermediate
 *     vertices (12 where unreachable);
 *   pn2_gen_edge_input: path (B,n,n), edge_feat (B,n,n,F) -> out (B,n,n,max_dist,F), which the CALLER pre-fills with
 *     -1: edge features along [i] + get_all_edges(path, i, j) + [j].
 * Bit-for-bit the reference, quirks included (see csrc/graph_algos.hip); parity is checked against the reference's
 * own Cython module compiled into oracle/_ref.
 */
int pn2_floyd_warshall(int B, int n, const long long *adjacency, long long *dist, long long *path, void *stream);
int pn2_gen_edge_input(int B, int n, int max_dist, int F, const long long *path, const long long *edge_feat,
                       long long *out, void *stream);

/* ------------------------------------------------------------- round 6: eval-mode set-abstraction level as ONE kernel ---
 * The "f32x3" product (csrc/x3_common.h): every fp32 operand = hi + mid + lo (three bf16, round to nearest), six partial
 * products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation — fp32-grade error (1.0-1.5x the exact fp32 MFMA kernel's
 * error against fp64, profiles/r05_split_bf16_gemm.jsonl) at 6/16 of its matrix time.
 *
 *   pn2_x3_pack_weight : W (N, ldw) fp32, columns [0, K) -> operand fragments, (N / 32)(K / 16) units of 3 KB, unit
 *       (tile t, chunk c) at index t (K / 16) + c = [piece][lane] 16 bytes: the 8 bf16 of weight row 32 t + (lane & 31) at
 *       the contraction indices of (c, lane >> 5) — natural order (perm = 0: the layer reads gathered rows) or the order in
 *       which a transposed accumulator tile holds them (perm = 1: the layer reads the previous layer's registers).
 *       N % 32 == 0, K % 16 == 0; `frags` holds pn2_x3_weight_bytes(N, K) bytes (whole 24 KB ring slots), 16-byte aligned.
 *   pn2_sa_eval_x3 : one scale of a set-abstraction level in EVAL mode (OPS/pointnet2_modules.py:29-74 with every
 *       BatchNorm2d folded into its Conv2d as W' = diag(gamma / sqrt(var + eps)) W, b' = beta - gamma mean / sqrt(var + eps)):
 *         out[b m + j][0:c_out] = max_s relu(L_last(... relu(L_1(row(b, j, s))))),  row = grouped input of sample s of centre j
 *       mode 0: the first layer sees [xyz[idx] - centre | feats[idx]] (3 + C <= 15 columns; feats (B, N, C) point-major rows);
 *               `w0_frags` = pn2_x3_pack_weight of the (c1, 16) matrix [W'_x (/ radius when the grouper normalises) | W'_f |
 *               b' | 0], perm 0 — the bias rides in the padding column 3 + C.
 *       mode 1: the first layer was applied per point (pn2_lift_points on folded weights): feats = Pq (B N, c1), Q (B m, c1)
 *               with the bias already SUBTRACTED from Q: first activation = relu(Pq[idx] - Q[centre]).
 *       then an optional middle layer c1 -> c_mid (bias_mid) and the last layer -> c_out (bias_fin); `wstream` = the middle
 *       layer's fragments followed by the last layer's (perm 1 wherever the layer's input is the previous layer's
 *       accumulators, i.e. everywhere except the last layer of a mode-1 stack without a middle layer).
 *       Nothing of size rows x channels is written: the activations stay in registers from the gather to the maximum.
 *       `out` rows have pitch ldo >= c_out (a multi-scale level writes its scales side by side).  ns in {16, 32, 64 k};
 *       `workspace`: 256 bytes of device memory owned by the call's stream (the kernels' pass counter: persistent workgroups
 *       CLAIM passes, so a grid that starts late under a co-running kernel does not hold the others back).  ZERO before its
 *       first use; every launch leaves it zero again (the last workgroup re-arms it).
 *       covered widths: pn2_sa_eval_x3_supported.  idx as written by pn2_ball_query.  Indices bit-exact by construction
 *       (they are inputs); features within 1e-4 of the fp32 reference (tests/test_gpu_round6.py, against the oracle).
 * Algorithmic bytes: B (4 m ns + 12 N + 12 m + 4 C N + 4 c_out m)  (mode 1: 4 c1 (N + m) instead of 12 N + 4 C N). */
size_t pn2_x3_weight_bytes(int N, int K);
int pn2_x3_pack_weight(int N, int K, int ldw, int perm, const float *W, void *frags, void *stream);
int pn2_sa_eval_x3_supported(int mode, int ns, int C, int c1, int c_mid, int c_out);
int pn2_sa_eval_x3(int mode, int B, int N, int m, int ns, int C, const float *xyz, const float *new_xyz, const int *idx,
                   const float *feats, const float *Q, int c1, const void *w0_frags, int c_mid, const void *wstream,
                   const float *bias_mid, int c_out, const float *bias_fin, float *out, int ldo, void *workspace, void *stream);

/* The same product as the TRAINING GEMM of a shared-MLP layer (opt-in arithmetic "f32x3": bench.py --dtype f32x3, PN2_X3=1;
 * the exact fp32 MFMA kernels stay the default).  pn2_x3_gemm = pn2_mlp_gemm / pn2_mlp_gemm_pool with W given as
 * pn2_x3_pack_weight(N, K, perm 0) fragments:
 *     pro 0 none | 1 relu(X p0[k] + p1[k]) | 2 p0[k] X + p1[k] X2 + p2[k]
 *     epi 1 store Y, stats (2, N) f64 += column sums of Y, Y^2 (stats may be NULL: plain store)
 *         2 Y *= [Yprev scale + shift > 0], stats += column sums of Y, Y (Yprev - mean) rstd      (e_fin (4, N) as pn2_mlp_gemm)
 *         3 nothing stored: stats as epi 1 (x sgn), pmax / parg (M / min(ns, 32), N) partial maxima + rows (pn2_mlp_gemm_pool)
 * covered: K in {64, 128}; N % 32 == 0 filling whole ring slots; (pro, epi) in {(0,1), (1,1), (1,3), (2,2)}
 * (pn2_x3_gemm_supported).  A is read straight into registers in operand layout (each row once), split there; one wave owns
 * 32 rows x all N columns.  Error: fp32-grade (tests/test_gpu_round6.py: <= 1e-4 against the exact kernels and the oracle). */
int pn2_x3_gemm_supported(int K, int N, int pro, int epi, int ns);
int pn2_x3_gemm(long long M, int K, int N, int pro, int epi, const float *X, const float *X2, const float *p0, const float *p1,
                const float *p2, const void *wfrags, float *Y, double *stats, const float *Yprev, const float *e_fin, float *pmax,
                int *parg, const float *sgn, int ns, void *workspace, void *stream);

/* The literal op's gradient (EXT/src/group_points_gpu.cu:43-64: one fp32 atomicAdd per gradient element) as a gather through
 * the inverse of idx: (ptr, refs) = pn2_group_inverse_index(B, N, npoints, nsample, idx).  grad_out (B, C, npoints, nsample),
 * grad_points (B, C, N): EVERY element written (no zero fill), a point's rows summed in ascending row order — no atomics,
 * bit-reproducible.  What `pointnet2_ops._ext.group_points_grad` runs (the atomic form stays exported as pn2_group_points_grad). */
int pn2_group_points_grad_csr(int B, int C, int N, int npoints, int nsample, const float *grad_out, const int *ptr,
                              const int *refs, float *grad_points, void *stream);
/* ... and three_interpolate's (EXT/src/interpolate_gpu.cu:116-143, three atomicAdds per gradient element): (ptr, refs) =
 * pn2_group_inverse_index(B, m, n, 3, idx); grad_out (B, C, n), weight (B, n, 3) -> grad_points (B, C, m), every element written.
 * `_ext.gather_points_grad` is pn2_group_points_grad_csr with nsample = 1. */
int pn2_three_interpolate_grad_csr(int B, int C, int n, int m, const float *grad_out, const float *weight, const int *ptr,
                                   const int *refs, float *grad_points, void *stream);

/* pn2_mlp_gemm_first on the f32x3 product (csrc/x3_chain.hip): Y (M, N) = relu(bn_0(X0 W0^T)) W^T and stats (2, N) += the column
 * sums of Y, Y^2 (NULL: none).  X0 (M, K0 <= 8) the grouped input rows; w0_frags = pn2_x3_pack_first(N0 = K, K0, W0, scale_0,
 * shift_0): the first layer with its BatchNorm folded in, (K / 32) x 3072 bytes; wfrags = pn2_x3_pack_weight(N, K, K, perm 1, W).
 * K = 64, N a multiple of 32 filling whole ring slots (pn2_x3_gemm_first_supported).  `workspace` as pn2_sa_eval_x3. */
int pn2_x3_gemm_first_supported(int K0, int K, int N);
int pn2_x3_pack_first(int N0, int K0, const float *W0, const float *scale, const float *shift, void *frags, void *stream);
int pn2_x3_gemm_first(long long M, int K0, int K, int N, const float *X0, const void *w0_frags, const void *wfrags, float *Y,
                      double *stats, void *workspace, void *stream);

/* pn2_mlp_bwd_fused_fold_first with its two 64-deep products (gy W and gy^T act) on the f32x3 product (csrc/mlp_bwd_first.hip,
 * template parameter X3); same arguments, preconditions and outputs; y_0 is re-formed exactly (it decides the ReLU mask). */
int pn2_x3_bwd_fold_first(long long M, int N, int K, int gmode, const float *G, const float *Yl, const float *consts,
                          const int *arg, const float *gP, int ns, const float *W, const float *W0, const float *a_fin,
                          const float *X, int K0, double *sums, float *dW, float *P1, void *stream);

/* pn2_pool_bwd with the matrix products of its K = 64, N <= 128 kernel (a G and the Gram blocks of a^T a) on the f32x3 product
 * (csrc/pool_bwd.hip, template parameter X3); every other supported shape runs the exact kernels.  Same arguments, workspace
 * (pn2_pool_bwd_workspace_bytes) and outputs. */
int pn2_x3_pool_bwd(long long M, int N, int K, int ns, const float *Yp, const float *fin_p, const float *W, const float *consts,
                    const int *arg, const float *gPm, float *Gout, double *sums, float *dW, void *workspace,
                    size_t workspace_bytes, void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PN2_HIP_H */
