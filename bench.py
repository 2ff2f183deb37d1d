#!/usr/bin/env python
"""bench.py — headline benchmark of the 4D-OR scene-graph hot path on MI355X.

Metric (BASELINE.json): OR scenes/sec, forward + backward, 50k points per scene,
batch 32 per GPU, fp32, through the full SA/FP stack (BASELINE configs[1]:
Group-Free `Pointnet2Backbone` shapes: SA 2048/0.2/64, 1024/0.4/32, 512/0.8/16,
256/1.2/16 + 2 FP levels).  A "step" = forward, loss, backward (gradient
all-reduce over RCCL when N > 1) and the AdamW update on one resident synthetic
batch.  Weak scaling: every rank processes its own 32 scenes.

Geometry pipeline (default; `--no-geometry-pipeline` turns it off): the parameter-free,
coordinate-only part of a batch — the FPS chain, the ball queries and the 3-NN weights — is what a
data loader can prepare ahead of the optimisation step.  Step i enqueues the geometry of batch i+1
on a side HIP stream before its own forward, so the latency-bound cooperative FPS kernel (2.6 us per
dependent round, < 10 % of the machine busy) co-runs with the MFMA kernels of step i instead of
serialising in front of them.  Every timed step still computes exactly one full geometry and
consumes the one computed during the previous step (the first one during warm-up); nothing is cached
or skipped.  The JSON also carries `ms_per_step_without_geometry_pipeline` and the uncontended
per-kernel table `kernels_without_geometry_pipeline` (same build, everything on one stream).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) that also carries
  "roofline":     the dominant hand-written HIP kernel of the timed region, measured
                  live with HIP events on the launch stream (algorithmic bytes from
                  SURVEY.md §8d / DESIGN.md), plus the per-kernel table in "kernels";
  "cpu_baseline": the same workload on the host CPU through the oracle port (N=1 only,
                  bounded sample).
"""
import argparse
import gc
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tests")]
# multi-process GPU jobs on this pool: the host driver only supports dmabuf IPC (RCCL's hipIpcGetMemHandle fails otherwise);
# normally exported already, set here too so that a bare torchrun environment works
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from runtime.affinity import pin_to_gpu_numa  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md chip table (spec; ~6300 achievable)
F32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32-input MFMA (= fp32 vector peak), same table
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (32x32x16), same table (the 5 PF headline figure is 2:1 sparse)


def synthetic_scenes(batch, points, seed, device):
    """xyz uniform in the unit ball, zero-mean, max-norm 1 (zero_mean of
    data_preparation_utils.py:12-18), rgb ~ U[0,1]  (BASELINE.md §3)."""
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(batch, points, 3, generator=g)
    p = p / p.norm(dim=2, keepdim=True) * torch.rand(batch, points, 1, generator=g).pow(1.0 / 3.0)
    p = p - p.mean(dim=1, keepdim=True)
    p = p / p.norm(dim=2).amax(dim=1).view(batch, 1, 1)
    rgb = torch.rand(batch, points, 3, generator=g)
    return torch.cat([p, rgb], dim=2).contiguous().to(device)


class ObjectEncoderWorkload(torch.nn.Module):
    """Stack 2a of SURVEY.md §8d: the scene-graph model's MSG object encoder (`PointNetfeat`: SA 512/[0.1,0.2]/[16,32],
    128/[0.2,0.4]/[32,64], group-all) on the same 32 x 50k x (3+3) batch, behind the backbone's bench interface."""

    def __init__(self):
        super().__init__()
        from scene_graph_prediction.scene_graph_helpers.model.pointnets.network_PointNet2 import PointNetfeat
        self.enc = PointNetfeat(input_dim=6, out_size=256)
        for n, p in self.enc.named_parameters():
            if n.startswith("backbone.fc_layer."):     # inherited classification head, never used
                p.requires_grad_(False)

    def precompute_geometry(self, pc):
        return self.enc.precompute_geometry(pc.transpose(1, 2))

    def forward(self, pc, geometry=None):
        return {"fp2_features": self.enc(pc.transpose(1, 2), geometry=geometry)}


def build_model(device, workload="backbone"):
    torch.manual_seed(0)
    if workload == "encoder":
        return ObjectEncoderWorkload().to(device)
    from external_src.group_free_3D.models.backbone_module import Pointnet2Backbone
    return Pointnet2Backbone(input_feature_dim=3).to(device)


def record_stream_tree(obj, stream):
    if torch.is_tensor(obj):
        obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            record_stream_tree(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            record_stream_tree(v, stream)


class FlatGradSync:
    """Data parallelism without DistributedDataParallel: every parameter's .grad is a view of ONE flat fp32 buffer
    (runtime.FlatGrads), zeroed once per step and averaged with ONE RCCL all-reduce between backward and the optimizer.
    The models on this path have 0.65 M (backbone) / 3.9 M (scene-graph model) live parameters: a single 2.6 / 15.6 MB
    collective is latency-bound over xGMI and there is nothing to overlap, while DDP's per-step bookkeeping (bucket
    rebuild, autograd hooks on ~100 parameters, reducer traversal) costs the host thread that is this step's bottleneck
    (measured at world size 1 under torchrun: 22.1 ms/step with DDP vs 17.9 without)."""

    def __init__(self, params, world):
        from runtime.graphed_step import FlatGrads
        self.world = world
        self.flat = FlatGrads(params)
        if world > 1:                                     # same initial weights everywhere (all ranks seed alike; belt and braces)
            for p in self.flat.params:
                dist.broadcast(p.data, src=0)

    def zero(self):
        self.flat.zero_()

    def sync(self):
        if self.world > 1:
            self.flat.all_reduce_mean()


class _MeanSquare(torch.autograd.Function):
    """mean(x^2) of the synthetic loss as ONE node: the composite `x.square().mean()` costs the backward three full-size
    element-wise launches (expand / divide, times two, times x) for what is x * (2 g / n)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x.square().mean()

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return x * (g * (2.0 / x.numel()))


def train_step(model, opt, pc, geometry=None, between=None, sync=None):
    if sync is not None:
        sync.zero()
    else:
        opt.zero_grad(set_to_none=True)
    feats = model(pc, geometry=geometry)["fp2_features"]
    # mean of squares over every element of the (B, C, n) result, taken through its point-major (B, n, C) view — the
    # layout the rows path produced it in (`feats` is a transposed view): the same number, and the gradient comes back
    # in that layout instead of through torch's strided element-wise kernels + a 33 MB transposing copy (round 4 trace:
    # 53 + 52 us of the step were this synthetic loss's layout, not the path under test)
    loss = _MeanSquare.apply(feats.transpose(1, 2) if feats.dim() == 3 else feats)
    nxt = between() if between is not None else None     # hook between forward and backward
    loss.backward()
    if sync is not None:
        sync.sync()
    opt.step()
    return nxt


def side_stream(device):
    """The geometry-prefetch stream.  HIP maps streams onto a few hardware queues round-robin, and two streams on one
    queue run strictly one after the other: under torchrun (RCCL's own streams shift the mapping) a default-priority
    side stream landed on the main stream's queue and the whole overlap was gone (23.8 instead of 18.2 ms/step,
    profiles/r02_stream_queue_aliasing.md).  A high-priority stream gets a queue of its own."""
    return torch.cuda.Stream(device=device, priority=-1)


class GeometryPrefetcher:
    """Data-loader-style pipelining of the coordinate-only work (FPS chain, ball queries, 3-NN
    weights: no parameters involved).  The geometry of the NEXT batch is enqueued on a side HIP
    stream before the current batch's forward/backward is enqueued on the main stream; the
    latency-bound cooperative FPS kernel (8 waves per CU) co-runs with the MFMA kernels.
    Every timed step still computes one full geometry (for the following step)."""

    def __init__(self, backbone, device):
        self.backbone = backbone
        self.side = side_stream(device)
        self.main = torch.cuda.current_stream(device)
        self.pending = None           # geometry enqueued for the NEXT step (of the same resident batch)

    def launch(self, pc):
        self.side.wait_stream(self.main)          # pc (and everything it depends on) is ready
        with torch.cuda.stream(self.side):
            geo = self.backbone.precompute_geometry(pc)
        return geo

    def acquire(self, geo):
        self.main.wait_stream(self.side)
        record_stream_tree(geo, self.main)
        return geo


class _NoCollectorPauses:
    """The timed region runs without the cyclic garbage collector (collected and frozen right before, re-enabled after):
    a generation-2 pass over this process's heap (modules, autograd nodes, cached tables) stops the enqueueing thread for
    30-90 ms, more than the host's lead over the GPU — seen in about half of the 20-step runs around step 10 as an idle
    GPU (the kernel brackets of a step sampled there: 3.7 ms of GEMMs measured as 10-90 ms; 20-step bf16 lines of
    10.2-14.4 ms against 9.9-10.0 clean).  Reference counting still frees every tensor; training loops that care about
    step-time jitter schedule collections themselves the same way.  PN2_BENCH_GC=auto restores the automatic collector."""

    def __enter__(self):
        self.on = os.environ.get("PN2_BENCH_GC", "manual") != "auto"
        if self.on:
            gc.collect()
            gc.freeze()
            gc.disable()
        return self

    def __exit__(self, *exc):
        if self.on:
            gc.enable()
            gc.unfreeze()
        return False


def run_steps(net, backbone, opt, pc, steps, prefetcher, on_step=None, sync=None):
    """`steps` training steps on the resident batch; with a prefetcher, step i consumes the geometry
    enqueued during step i-1 and enqueues the one for step i+1."""
    if prefetcher is None:
        for i in range(steps):
            if on_step is not None:
                on_step(i)
            train_step(net, opt, pc, sync=sync)
        return
    # steady state across calls: the geometry enqueued by the last step of the previous call (warm-up) feeds the first
    # step of this one, exactly as it does between two steps
    geo = prefetcher.pending if prefetcher.pending is not None else prefetcher.launch(pc)
    prefetcher.pending = None
    for i in range(steps):
        if on_step is not None:
            on_step(i)
        cur = prefetcher.acquire(geo)
        # the next batch's geometry is enqueued BEFORE this batch's forward: it co-runs with the whole step
        # (measured launch positions with 512-thread FPS clusters: start of the step 20.2 ms, behind sa1's forward 20.7,
        # behind sa2 21.5, behind the whole forward 22.2-23.3; with PN2_FPS_FEW_CUS: 18.2 at the start, 18.2 behind the forward)
        nxt = prefetcher.launch(pc)
        train_step(net, opt, pc, cur, sync=sync)
        geo = nxt
    prefetcher.pending = geo


def forward_only(net, backbone, pc, steps, prefetcher):
    """Forward-only figure asked for beside the headline: the same train-mode forward (batch
    statistics) without autograd, geometry prefetched like the timed steps when the pipeline is on."""
    def run(k):
        with torch.no_grad():
            if prefetcher is None:
                for _ in range(k):
                    net(pc)
                return
            geo = prefetcher.pending if prefetcher.pending is not None else prefetcher.launch(pc)
            for _ in range(k):
                cur = prefetcher.acquire(geo)
                geo = prefetcher.launch(pc)
                net(pc, geometry=cur)
            prefetcher.pending = geo
    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def kernel_table(table, steps):
    """Per-entry-point rows from a KernelTimer summary, sorted by time."""
    rows = []
    for name, d in table.items():
        # (f32x3 kernels run six bf16 partial products per fp32-grade product: their ceiling is a sixth of the bf16 peak)
        peak = BF16_MFMA_PEAK_TFLOPS if "bf16" in name else (BF16_MFMA_PEAK_TFLOPS / 6.0 if "x3" in name else F32_MFMA_PEAK_TFLOPS)
        ridge = peak * 1e12 / (HBM_PEAK_GBPS * 1e9)                   # flop/byte where the rooflines cross
        per_launch_ms = d["ms"] / d["calls"]
        per_launch_bytes = d["alg_bytes"] / d["calls"]
        per_launch_flops = d["alg_flops"] / d["calls"]
        gbps = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        tfps = per_launch_flops / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0
        mfma_bound = per_launch_bytes > 0 and per_launch_flops / per_launch_bytes >= ridge
        rows.append({"kernel": name, "calls_per_step": d["calls"] / steps,
                     "ms_per_step": round(d["ms"] / steps, 4),
                     "avg_launch_us": round(per_launch_ms * 1e3, 2),
                     "alg_MB_per_launch": round(per_launch_bytes / 1e6, 3),
                     "alg_GFLOP_per_launch": round(per_launch_flops / 1e9, 3),
                     "GBps": round(gbps, 1), "TFLOPps": round(tfps, 2),
                     "bound": "mfma" if mfma_bound else "hbm", "mfma_peak_TFLOPps": peak,
                     "frac": round(tfps / peak if mfma_bound else gbps / HBM_PEAK_GBPS, 5)})
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def roofline_of(top, pmc_applies=True, pmc_files=None):
    """`roofline` object for the dominant critical-path kernel row (`pmc_applies`: the committed PMC traffic was
    collected on the default command — backbone workload, 32 x 50k — and is quoted for that command only; `pmc_files`:
    the counter summaries of ANOTHER command that has its own committed passes, e.g. the 8-scan bf16 scene-graph step)."""
    if top["bound"] == "mfma":
        roof = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["TFLOPps"],
                "peak": top.get("mfma_peak_TFLOPps", F32_MFMA_PEAK_TFLOPS), "unit": "TFLOP/s", "frac": top["frac"]}
    else:
        roof = {"bound": "hbm", "kernel": top["kernel"], "achieved": top["GBps"],
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": top["frac"]}
    traffic, traffic_src = None, None
    # PMC records are keyed by C-ABI entry point (`entry:<name>`: tools/kernel_keys.py maps every traced instantiation to the
    # entry point that launches it), so `traffic` and `alg_bytes_per_launch` describe the same launches; counter files
    # written before round 4 only have kernel-family keys (second element, mixes e.g. the pooled GEMM into pn2_mlp_gemm)
    legacy = {"pn2_mlp_gemm": "mlp_gemm_kernel", "pn2_mlp_gemm_pool": "mlp_gemm_kernel", "pn2_mlp_wgrad": "mlp_wgrad_kernel",
              "pn2_mlp_bwd_fused": "mlp_bwd_fused_kernel", "pn2_mlp_bwd_fused_fold": "mlp_bwd_fused2_kernel",
              "pn2_pool_bwd": "pool_bwd64_kernel",
              "pn2_bn_relu_rows_max": "bn_relu_rows_max_kernel",
              "pn2_group_concat_rows": "group_concat_rows_wide4_kernel",
              "pn2_group_rows_grad": "group_rows_grad_csr_kernel",
              "pn2_mlp_gemm_bf16": "mlp_gemm_bf16_kernel", "pn2_mlp_wgrad_bf16": "mlp_wgrad_bf16_kernel",
              "pn2_mlp_bwd_bf16": "mlp_bwd_bf16_kernel", "pn2_bn_relu_rows_max_bf16": "bn_relu_rows_max_bf16_v8_kernel",
              "pn2_group_concat_rows_bf16": "group_concat_rows_bf16_wide8_kernel"}.get(top["kernel"])
    kname = ("entry:" + top["kernel"], legacy)
    bf16 = "bf16" in top["kernel"]
    # newest committed counter summary of the command first (tools/profile_round.sh + tools/summarise_profile.py); the
    # fp32 and the bf16 default commands have their own files
    files = ((("r06_backbone_bf16_counters.json", "hbm_MB_per_launch", 1e6), ("r05_backbone_bf16_counters.json", "hbm_MB_per_launch", 1e6), ("r04_backbone_bf16_counters.json", "hbm_MB_per_launch", 1e6), ("r03_backbone_bf16_counters.json", "hbm_MB_per_launch", 1e6),
              ("r02_backbone_bf16_counters.json", "hbm_MB_per_launch", 1e6))
             if bf16 else
             (("r06_backbone_counters.json", "hbm_MB_per_launch", 1e6), ("r05_backbone_counters.json", "hbm_MB_per_launch", 1e6), ("r04_backbone_counters.json", "hbm_MB_per_launch", 1e6), ("r03_backbone_counters.json", "hbm_MB_per_launch", 1e6),
              ("r02_backbone_counters.json", "hbm_MB_per_launch", 1e6),
              ("r01_hbm_traffic_per_kernel.json", "hbm_bytes_per_launch", 1.0)))
    if pmc_files is not None:
        files, pmc_applies = tuple((f, "hbm_MB_per_launch", 1e6) for f in pmc_files), True
    for fname, key, scale in files:
        tf = os.path.join(REPO, "profiles", fname)
        if not (pmc_applies and os.path.exists(tf)) or traffic is not None:
            continue
        try:
            recs = json.load(open(tf))["kernels"]
            hit = next((k for k in kname if k and recs.get(k, {}).get(key)), None)
            if hit:
                traffic = int(recs[hit][key] * scale)
                traffic_src = ("PMC FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE per launch, separate rocprofv3 "
                               f"passes over this command: profiles/{fname}, record `{hit}`"
                               + ("" if hit.startswith("entry:") else " (kernel family: all instantiations)"))
        except Exception:
            pass
    roof.update({"traffic": traffic, "traffic_source": traffic_src,
                 "avg_launch_us": top["avg_launch_us"],
                 "alg_bytes_per_launch": int(top["alg_MB_per_launch"] * 1e6),
                 "alg_flops_per_launch": int(top["alg_GFLOP_per_launch"] * 1e9),
                 "note": "dominant hand-written kernel of the main (critical-path) stream; entry point aggregated over "
                         "its launches in the timed steps (shapes differ per layer; per-shape table: "
                         "PN2_TIMER_DETAIL=1 python bench.py)"})
    return roof


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(points, sample_scenes, threads, workload="backbone", repeats=5):
    """Same stack, same step, on the host: oracle port (C/OpenMP restatement of the nine
    native ops) + stock-torch CPU MLPs.  Protocol of SURVEY.md 8d: one warm-up, then the MEDIAN of `repeats` runs,
    forward-only and forward + backward + AdamW, all host cores.  Bounded sample; reported, never optimised."""
    from pointnet2_ops import pointnet2_utils as pu
    import oracle_ext
    import statistics
    torch.set_num_threads(threads)
    saved = pu._ext
    pu._ext = oracle_ext.OracleRowsExt
    try:
        model = build_model("cpu", workload)
        opt = torch.optim.AdamW(model.parameters(), lr=3e-5, weight_decay=1e-3, fused=True)
        pc = synthetic_scenes(sample_scenes, points, seed=1234, device="cpu")
        t_all = time.perf_counter()
        train_step(model, opt, pc)                                   # warm-up (allocator, OpenMP pool, oneDNN primitives)
        step_s, fwd_s = [], []
        for _ in range(repeats):
            t0 = time.perf_counter()
            train_step(model, opt, pc)
            step_s.append(time.perf_counter() - t0)
        with torch.no_grad():
            for _ in range(repeats):
                t0 = time.perf_counter()
                model(pc)
                fwd_s.append(time.perf_counter() - t0)
        wall = time.perf_counter() - t_all
    finally:
        pu._ext = saved
    dt, df = statistics.median(step_s), statistics.median(fwd_s)
    # `value` is a per-scene rate measured on `sample_scenes` scenes (the GPU line runs 32 per step): an extrapolation, said so
    return {"value": round(sample_scenes / dt, 4), "unit": "scenes/s", "cores": threads, "kind": "port",
            "sample_scenes": int(sample_scenes), "extrapolated": True,
            "forward_only_value": round(sample_scenes / df, 4), "cpu_model": cpu_model_name(),
            "median_step_s": round(dt, 3), "median_forward_s": round(df, 3), "repeats": repeats,
            "sample": f"{sample_scenes} scenes x {points} pts; 1 warm-up + median of {repeats} fwd+bwd+AdamW steps and of "
                      f"{repeats} train-mode forwards, {wall:.1f} s of CPU work in total (oracle ops are OpenMP-parallel "
                      f"over scenes; MLPs torch CPU with {threads} threads)"}


def emit_json(out, args):
    sys.stdout.flush()
    os.dup2(args._real_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)


def bench_sgp(args, device, rank, world, distributed, _ext):
    """BASELINE configs[2] shape on one or more GPUs: SGPNModelWrapper (2 MSG encoders + 2-layer
    TripletGCN + heads), `--scans-per-step` synthetic scans per step and rank (block-diagonal batch, per-scan GCN
    BatchNorm statistics and loss average; 1 = the reference's DataLoader(batch_size=1)), fwd + loss + bwd + AdamW."""
    from scene_graph_prediction.main import RELATION_NAMES, config_loader
    from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import collate_scans, synthetic_scan, to_device
    from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
    cfg = config_loader("no_gt.json")
    torch.manual_seed(0)
    model = SGPNModelWrapper(cfg, 12, len(RELATION_NAMES), torch.ones(12), torch.ones(len(RELATION_NAMES)),
                             RELATION_NAMES).to(device).train()
    # the two inherited `backbone.fc_layer` heads never receive a gradient: freeze them
    for n, p in model.named_parameters():
        if ".backbone.fc_layer." in n:
            p.requires_grad_(False)
    trainable = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(trainable, lr=float(cfg["LR"]), weight_decay=float(cfg["W_DECAY"]), capturable=args.graphs, fused=True)
    S = max(1, int(args.scans_per_step))
    # this step is bound by the ONE host thread that enqueues it: keep that thread next to the rank's GPU
    affinity = pin_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    model.per_scan_statistics = not args.whole_batch_statistics
    if args.segment_streams is not None:
        from pointnet2_ops import fused_mlp
        fused_mlp.SEGMENT_STREAMS = max(1, int(args.segment_streams))
    fused = None
    if args.with_prep:
        # end to end: every step first cuts this step's scans out of resident fused room clouds (300k points, 9
        # instances each) with the GPU preparation kernels (dataset/gpu_preparation.py), then trains on them
        from scene_graph_prediction.scene_graph_helpers.dataset import gpu_preparation as gp
        fused = [gp.synthetic_fused_scan(9, 300000, seed=100 + rank * S + i, device=device) for i in range(S)]
        names = ["Patient", "operating_table", "human_0", "human_1", "instrument", "secondary_table", "instrument_table",
                 "anesthesia_equipment", "human_2"]
        g = torch.Generator().manual_seed(7)
        labels = [(torch.randint(0, 12, (9,), generator=g).to(device), torch.randint(0, 15, (72,), generator=g).to(device))
                  for _ in range(S)]

        def prepare(step_idx):
            scans = [gp.prepare_scan(p, m, 9, 4000, 8000, seed=step_idx * S + i, object_names=names, gt_class=labels[i][0],
                                     gt_rels=labels[i][1], scan_id=f"prep_{i:06d}") for i, (p, m) in enumerate(fused)]
            for sc in scans:
                sc.pop("prep")
            return scans[0] if S == 1 else to_device(collate_scans(scans), device)
        scan = prepare(0)
    elif S == 1:
        scan = to_device(synthetic_scan(9, 4000, 8000, seed=100 + rank), device)
    else:
        scan = to_device(collate_scans([synthetic_scan(9, 4000, 8000, seed=100 + rank * S + i, scan_id=f"synthetic_{i:06d}")
                                        for i in range(S)]), device)
    # geometry of the NEXT scan (FPS chains + ball queries of both encoders) on a side stream during this step,
    # like the backbone workload; the scan-at-a-time loop of the reference knows its next scan from the data loader
    side = side_stream(device) if args.geometry_pipeline else None
    main = torch.cuda.current_stream(device)
    state = {"geo": None}

    counter = {"step": 0}

    def launch_geometry():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            return model.precompute_geometry(scan)

    def with_prefetched_geometry():
        """This step's batch (geometry enqueued during the previous step) + enqueue the next scan's geometry."""
        if fused is not None:
            counter["step"] += 1
            return prepare(counter["step"])                  # crops change every step: preparation + geometry in the step
        if side is None:
            return scan
        if state["geo"] is None:
            state["geo"] = launch_geometry()
        main.wait_stream(side)
        batch = dict(scan, geometry=state["geo"])
        record_stream_tree(batch["geometry"], main)
        state["geo"] = None if late_geometry else launch_geometry()
        return batch

    # Where the next scan's geometry is enqueued: at the start of the step it shares the chip with the encoders' forward
    # GEMMs; between forward and backward ("late") it runs beside the backward of the heads and the GCN — a few hundred
    # launch-bound kernels of microseconds each, during which the chip is otherwise ~70 % idle (profiles/r03_sgp_gaps.md)
    late_geometry = side is not None and fused is None and args.geometry_launch == "backward" and not args.graphs

    def launch_late_geometry():
        if late_geometry and state["geo"] is None:
            state["geo"] = launch_geometry()

    if args.graphs:
        # the step (with the geometry as an INPUT: its tensors are copied into the graph's static buffers) is one replay;
        # the host only enqueues the ~40 geometry kernels of the next scan, the copies and the replay
        from runtime import GraphedTrainStep
        graphed = GraphedTrainStep(model.pure_training_step, trainable, opt)
        args.no_kernel_timing = True          # HIP events cannot be recorded around kernels inside a replay

        def step():
            graphed(with_prefetched_geometry())
    else:
        net, sync = model, None
        if distributed and args.grad_sync == "ddp":
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index], bucket_cap_mb=64,
                                                            gradient_as_bucket_view=True, broadcast_buffers=False)
        elif distributed:
            sync = FlatGradSync(trainable, world)

        def step():
            batch = with_prefetched_geometry()
            if sync is not None:
                sync.zero()
            else:
                opt.zero_grad(set_to_none=True)
            obj, rel = net(batch)
            loss = model.loss(obj, rel, batch)
            launch_late_geometry()
            loss.backward()
            if sync is not None:
                sync.sync()
            opt.step()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # per-kernel HIP events on ONE sampled step only: two event records per launch double the host cost of a step that
    # is host-bound at one scan per step (~400 launches)
    timer, sampled = None, -1
    if not args.no_kernel_timing:
        timer = _ext.KernelTimer(torch.cuda.current_stream(device).cuda_stream)
        timer.enabled = False
        _ext.TIMER = timer
        sampled = args.steps // 2
    with _NoCollectorPauses():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if timer is not None:
                timer.enabled = i == sampled
            step()
        enqueue_ms = (time.perf_counter() - t0) / args.steps * 1e3      # host time to enqueue a step (no GPU wait)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    _ext.TIMER = None
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        out = {"metric": "OR scans/sec fwd+bwd (9 objects x 4000 pts + 72 pairs x 8000 pts per scan)",
               "value": round(world * S * args.steps / elapsed, 3), "unit": "scans/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"BASELINE configs[2] shape: SGPNModelWrapper(no_gt.json), {S} synthetic scan(s) per step "
                                      "and rank (block-diagonal batch, loss = mean of the per-scan losses; BatchNorm batch "
                                      "statistics " + ("per scan in encoders, GCN and heads = the arithmetic of single-scan steps"
                                                       if model.per_scan_statistics or S == 1 else
                                                       "per scan in the GCN, over the whole step's clouds in encoders and heads") +
                                      "), train mode, fwd + weighted NLL + bwd + AdamW",
                          "scans_per_step": S, "per_scan_statistics": bool(model.per_scan_statistics or S == 1),
                          "with_gpu_preparation": bool(args.with_prep),
                          "parallelism": f"dp{world}", "hip_graphs": bool(args.graphs),
                          "host_enqueue_ms_per_step": round(enqueue_ms, 3), "host_affinity": affinity,
                          "geometry_pipeline": bool(args.geometry_pipeline and not args.graphs)}}
        if timer is not None:
            rows = kernel_table(timer.summary(), 1)
            out["kernels"] = rows
            out["kernel_timing"] = {"method": "HIP events on the launch stream around every C-ABI call", "sampled_steps": 1,
                                    "of_steps": args.steps}
            main_rows = [r for r in rows if not r["kernel"].endswith("@side")]
            out["hip_kernel_ms_per_step"] = round(sum(r["ms_per_step"] for r in main_rows), 3)
            if main_rows:
                # the one scene-graph command with committed counter passes (tools/profile_round.sh r04_sgp8_bf16 ...)
                own = None
                if S == 8 and model.per_scan_statistics and not args.with_prep and world == 1:
                    own = (["r05_sgp8_bf16_counters.json", "r04_sgp8_bf16_counters.json", "r03_sgp8_bf16_counters.json"]
                           if args.dtype == "bf16" else ["r05_sgp8_f32_counters.json"])
                out["roofline"] = roofline_of(main_rows[0], False, own)
        emit_json(out, args)
    if distributed:
        dist.destroy_process_group()


X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0     # six bf16 partial products per fp32-grade product (csrc/x3_common.h)


def bench_forward_eval(args, device, rank, world, distributed, _ext):
    """Inference: eval-mode forward (running-statistic BatchNorm, no autograd) of the workload's model on a resident batch,
    one full geometry (FPS chain, ball queries, 3-NN) per step — prefetched for the NEXT batch on the side stream like the
    training steps, or inside the step with --no-geometry-pipeline.  `value` = scenes/s through the one-kernel SA levels
    (pn2_sa_eval_x3); the same model through the layer-by-layer exact-fp32 kernels is timed beside it."""
    from pointnet2_ops import eval_fused
    sgp = args.workload == "sgp"
    if sgp:
        from scene_graph_prediction.main import RELATION_NAMES, config_loader
        from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import collate_scans, synthetic_scan, to_device
        from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
        torch.manual_seed(0)
        model = SGPNModelWrapper(config_loader("no_gt.json"), 12, len(RELATION_NAMES), torch.ones(12),
                                 torch.ones(len(RELATION_NAMES)), RELATION_NAMES).to(device).eval()
        S = max(1, int(args.scans_per_step))
        batch = (to_device(synthetic_scan(9, 4000, 8000, seed=100 + rank), device) if S == 1 else
                 to_device(collate_scans([synthetic_scan(9, 4000, 8000, seed=100 + rank * S + i, scan_id=f"synthetic_{i:06d}")
                                          for i in range(S)]), device))
        units, unit_name = S, "scans"

        def geometry():
            return model.precompute_geometry(batch)

        def forward(geo):
            b = batch if geo is None else dict(batch, geometry=geo)
            return model(b)
    else:
        model = build_model(device, args.workload).eval()
        batch = synthetic_scenes(args.batch, args.points, seed=1000 + rank, device=device)
        units, unit_name = args.batch, "scenes"

        def geometry():
            return model.precompute_geometry(batch)

        def forward(geo):
            return model(batch, geometry=geo)

    affinity = pin_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    # inference pipeline: the sampling chain (latency-bound, a fraction of the CUs) of the NEXT `depth` batches runs on `depth`
    # side streams while the current batch's forward runs on the main stream; depth 2 keeps two sampling chains in flight —
    # at the headline shape one chain (2.7 ms) is longer than the forward (2.0 ms) and would bound the rate
    depth = max(1, int(args.eval_prefetch_depth)) if args.geometry_pipeline else 0
    sides = [side_stream(device) for _ in range(depth)]
    side = sides[0] if sides else None
    main = torch.cuda.current_stream(device)
    state = {"queue": [], "n": 0}

    def launch_geometry():
        st = sides[state["n"] % depth]
        state["n"] += 1
        st.wait_stream(main)
        with torch.cuda.stream(st):
            return st, geometry()

    def run(k, on_step=None):
        with torch.no_grad():
            for i in range(k):
                if on_step is not None:
                    on_step(i)
                if side is None:
                    forward(None)
                    continue
                while len(state["queue"]) < depth:
                    state["queue"].append(launch_geometry())
                st, geo = state["queue"].pop(0)
                main.wait_stream(st)
                record_stream_tree(geo, main)
                state["queue"].append(launch_geometry())          # the geometry `depth` batches ahead co-runs with this forward
                forward(geo)

    def timed(k):
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(k)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # the layer-by-layer exact route first (same model, same batch), then the one-kernel route = `value`
    prev = eval_fused.set_eval_fused(False)
    run(args.warmup)
    layers_s = timed(max(3, args.steps // 2)) / max(3, args.steps // 2)
    eval_fused.set_eval_fused(True)
    run(args.warmup)
    with _NoCollectorPauses():
        elapsed = timed(args.steps)
    rows = None
    if not args.no_kernel_timing and rank == 0:
        timer = _ext.KernelTimer(main.cuda_stream)
        _ext.TIMER = timer
        run(2)
        _ext.TIMER = None
        rows = kernel_table(timer.summary(), 2)
        for r in rows:
            if r["kernel"].startswith("pn2_sa_eval_x3"):      # priced against what six bf16 products per product allow
                r["mfma_peak_TFLOPps"] = round(X3_PEAK_TFLOPS, 1)
                r["bound"] = "mfma"
                r["frac"] = round(r["TFLOPps"] / X3_PEAK_TFLOPS, 5)
    eval_fused.set_eval_fused(prev)
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": f"OR {unit_name}/sec forward, eval mode" + ("" if sgp else f" ({args.points // 1000}k pts, batch {args.batch})"),
            "value": round(units * world * args.steps / elapsed, 3), "unit": f"{unit_name}/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32x3", "data": "synthetic",
            "config": {
                "workload": ("BASELINE configs[2] shape, inference: SGPNModelWrapper.eval() on synthetic scans (9 objects x 4000 pts + "
                             f"72 pairs x 8000 pts), {units} scan(s) per step" if sgp else
                             f"BASELINE configs[1] shape, inference: {args.batch} scenes/GPU x {args.points} pts x (3 xyz + 3 rgb), "
                             + ("Pointnet2Backbone" if args.workload == "backbone" else "MSG object encoder (PointNetfeat)")
                             + ".eval() forward under no_grad"),
                "arithmetic": "SA levels: split-bf16 (f32x3) product, fp32 accumulation, fp32-grade error (1e-6 against the oracle, "
                              "tests/test_gpu_round6.py); sampling / grouping geometry and FP levels: fp32",
                "global_batch": units * world, "parallelism": f"dp{world}",
                "geometry_pipeline": "off: geometry inside the step" if side is None else
                f"on: FPS / ball-query / 3-NN of batches i+1 .. i+{depth} on {depth} side stream(s) during forward i (one geometry per "
                "timed step)",
                "host_affinity": affinity,
                "layer_by_layer_exact_fp32_ms_per_step": round(layers_s * 1e3, 3),
                "layer_by_layer_exact_fp32_per_s": round(units * world / layers_s, 1),
            },
        }
        if rows is not None:
            out["kernels"] = rows
            main_rows = [r for r in rows if not r["kernel"].endswith("@side")]
            out["hip_kernel_ms_per_step"] = round(sum(r["ms_per_step"] for r in main_rows), 3)
            top = next((r for r in main_rows if r["kernel"].startswith("pn2_sa_eval_x3")), main_rows[0] if main_rows else None)
            if top is not None:
                pmc = ["r06_eval_counters.json"] if (not sgp and args.workload == "backbone" and args.batch == 32 and args.points == 50000) else None
                out["roofline"] = roofline_of(top, False, pmc_files=pmc)
                out["roofline"]["note"] = ("pn2_sa_eval_x3 aggregated over the step's launches; peak = dense bf16 MFMA / 6 (the f32x3 "
                                           "product issues six bf16 matrix instructions per fp32-grade product)")
        emit_json(out, args)
    if distributed:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 40 steps: run-to-run spread of the mean below 0.5 %
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="scenes per GPU")
    ap.add_argument("--points", type=int, default=50000)
    ap.add_argument("--geometry-launch", choices=["start", "backward"], default="backward",
                    help="sgp workload: enqueue the next scan's geometry at the start of the step or between forward and backward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-scenes", type=int, default=2)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-forward-only", action="store_true", help="skip the forward-only figure reported beside the headline")
    ap.add_argument("--no-serial-reference", action="store_true",
                    help="skip the short un-pipelined pass behind `ms_per_step_without_geometry_pipeline`")
    ap.add_argument("--workload", choices=["backbone", "encoder", "sgp"], default="backbone",
                    help="backbone = BASELINE configs[1] (the headline metric); encoder = the same batch through the "
                         "scene-graph model's MSG object encoder (SURVEY 8d stack 2a); sgp = BASELINE configs[2] shape: the full "
                         "scene-graph model on synthetic scans (9 objects x 4000 pts + 72 pairs x 8000 pts, one scan per "
                         "step like the reference's DataLoader(batch_size=1)), fp32")
    ap.add_argument("--grad-sync", choices=["flat", "ddp"], default="flat",
                    help="multi-GPU gradient exchange: flat = one RCCL all-reduce of a flat gradient buffer per step (default); "
                         "ddp = torch DistributedDataParallel")
    ap.add_argument("--with-prep", action="store_true",
                    help="sgp workload: include the GPU data preparation (object / pair crops of 300k-point fused scans, "
                         "re-sampling, zero_mean) of every step's scans in the timed region")
    ap.add_argument("--dtype", choices=["f32", "bf16", "f32x3"], default="f32",
                    help="arithmetic of the shared-MLP stacks: f32 = exact fp32 MFMA (the headline / parity path); bf16 = the "
                         "counterpart of the reference's 16-bit AMP (bf16 activations and MFMA, fp32 weights and statistics); "
                         "f32x3 = fp32 tensors, the GEMMs the split-bf16 product covers on the bf16 matrix cores as hi/mid/lo "
                         "pieces (fp32-grade error, NOT the exact fp32 arithmetic: reported under its own dtype label)")
    ap.add_argument("--scans-per-step", type=int, default=1,
                    help="sgp workload: scans per step and rank, collated block-diagonally (BASELINE configs[2] names 32)")
    ap.add_argument("--whole-batch-statistics", action="store_true",
                    help="sgp workload with several scans per step: encoders and heads normalise over all the step's clouds "
                         "(a larger BatchNorm batch, fewer launches) instead of per scan (default: per scan = the arithmetic "
                         "of single-scan steps of the reference with their gradients averaged)")
    ap.add_argument("--segment-streams", type=int, default=None,
                    help="sgp workload with per-scan statistics: streams the scans' shared-MLP chains are spread over "
                         "(pointnet2_ops.fused_mlp.SEGMENT_STREAMS; default: the library's setting)")
    ap.add_argument("--graphs", action="store_true",
                    help="sgp workload: replay the whole step as one hipGraph (runtime.GraphedTrainStep); gradients are "
                         "averaged with one flat all-reduce between the backward and the optimizer graph when N > 1")
    ap.add_argument("--forward-eval", action="store_true",
                    help="inference instead of training: model.eval() forward under no_grad (running-statistic BatchNorm), every "
                         "SA scale as ONE kernel (pn2_sa_eval_x3: gather -> folded-BatchNorm MLP chain in registers on the bf16 matrix "
                         "cores, f32x3 split product -> max); reports scenes/s forward, the layer-by-layer route beside it")
    ap.add_argument("--eval-prefetch-depth", type=int, default=2,
                    help="--forward-eval: batches whose sampling / grouping geometry is in flight on side streams ahead of the forward")
    ap.add_argument("--no-geometry-pipeline", dest="geometry_pipeline", action="store_false",
                    help="run the sampling/grouping geometry inside the step on the main stream instead of prefetching "
                         "the NEXT batch's geometry on a side stream during the step (25.0 vs 20.7 ms/step on MI355X)")
    ap.add_argument("--geometry-pipeline", dest="geometry_pipeline", action="store_true", help=argparse.SUPPRESS)
    ap.set_defaults(geometry_pipeline=True)
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line: anything a library prints there (RCCL's version banner at communicator
    # creation) is routed to stderr; the real stdout is restored for the final print
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    args._real_stdout = real_stdout

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = world > 1 or "RANK" in os.environ     # launched by torch.distributed.run (even with 1 rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)   # "nccl" is RCCL on ROCm
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    from pointnet2_ops import _ext, fused_mlp
    fused_mlp.set_mlp_dtype("f32" if args.dtype == "f32x3" else args.dtype)
    fused_mlp.set_x3(args.dtype == "f32x3")

    if args.forward_eval:
        return bench_forward_eval(args, device, rank, world, distributed, _ext)
    if args.workload == "sgp":
        return bench_sgp(args, device, rank, world, distributed, _ext)

    model = build_model(device, args.workload)
    net, sync = model, None
    if distributed and args.grad_sync == "ddp":
        # gradients only: one flat bucket (650k params = 2.6 MB, latency-bound over xGMI)
        # BatchNorm statistics stay per rank (no SyncBN: the reference is single-GPU), so the running
        # buffers are not re-broadcast from rank 0 before every forward either
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], bucket_cap_mb=64,
                                                        gradient_as_bucket_view=True, broadcast_buffers=False)
    elif distributed:
        sync = FlatGradSync(model.parameters(), world)
    opt = torch.optim.AdamW(model.parameters(), lr=3e-5, weight_decay=1e-3, fused=True)   # one kernel (the foreach form: ~12 launches; 1.1 ms per step on slow-host boxes)
    pc = synthetic_scenes(args.batch, args.points, seed=1000 + rank, device=device)   # resident in HBM
    # the enqueueing thread (and the autograd thread it spawns at the first backward) next to this rank's GPU
    # (runtime/affinity.py); the host threads torch already created keep the whole machine for the CPU baseline
    host_cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    affinity = pin_to_gpu_numa(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))

    prefetcher = GeometryPrefetcher(model, device) if args.geometry_pipeline else None
    run_steps(net, model, opt, pc, args.warmup, prefetcher, sync=sync)
    torch.cuda.synchronize()

    main_stream = torch.cuda.current_stream(device).cuda_stream
    # reference pass without the geometry pipeline (same build, same batch): wall time and an
    # uncontended per-kernel table, reported alongside
    serial_ms, serial_rows = None, None
    if prefetcher is not None and rank == 0 and world == 1 and not args.no_serial_reference:
        ks = max(2, min(args.steps, 5))
        run_steps(net, model, opt, pc, 1, None, sync=sync)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        run_steps(net, model, opt, pc, ks, None, sync=sync)
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - ts) / ks * 1e3
        if not args.no_kernel_timing:
            st = _ext.KernelTimer(main_stream)
            _ext.TIMER = st
            run_steps(net, model, opt, pc, 2, None, sync=sync)
            _ext.TIMER = None
            serial_rows = kernel_table(st.summary(), 2)
        run_steps(net, model, opt, pc, 2, prefetcher, sync=sync)       # back to the pipelined steady state
        torch.cuda.synchronize()

    # per-kernel HIP events are sampled on a few of the timed steps; the table is normalised by the number of sampled steps
    timer, sampled_steps, on_step = None, 0, None
    if not args.no_kernel_timing:
        timer = _ext.KernelTimer(main_stream)
        _ext.TIMER = timer
        # K < 4: every step; K < 24: one step in the middle of the timed region; longer runs: two steps.  (The events are the
        # library's fence-free ones, pn2_event_*: with torch.cuda.Event — a system-scope release per record — a sampled step
        # was ~11 ms longer, 0.55 ms per step of a 20-step run; now ~2 ms)
        if args.steps < 4:
            sampled = set(range(args.steps))
        elif args.steps < 24:
            sampled = {args.steps // 2}
        else:
            sampled = {args.steps // 3, 2 * args.steps // 3}
        sampled_steps = len(sampled)

    # host time of a step = the time the enqueue loop takes BEFORE the hardware queue fills up (a step is ~250 packets: a
    # host that runs ahead of the GPU is throttled by the queue after ~15 steps, and the average over all steps then
    # mostly measures the GPU again): the first `head` steps after the synchronize, none of them a sampled one
    head = max(1, min(8, args.steps))
    stamps = []
    user_on_step = None
    if timer is not None:
        sampled = {i for i in sampled if i >= head} or sampled        # (very short runs keep their sampled step)

        def user_on_step(i):
            timer.enabled = i in sampled

    def on_step(i):
        stamps.append(time.perf_counter())
        if user_on_step is not None:
            user_on_step(i)
    with _NoCollectorPauses():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(net, model, opt, pc, args.steps, prefetcher, on_step, sync=sync)
        t_loop = time.perf_counter()
        enqueue_ms = ((stamps[head] if head < len(stamps) else t_loop) - stamps[0]) / head * 1e3
        enqueue_all_ms = (t_loop - t0) / args.steps * 1e3
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    _ext.TIMER = None

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    fwd_ms = None
    if world == 1 and not args.no_forward_only:
        fwd_ms = forward_only(model, model, pc, max(3, min(args.steps, 10)), prefetcher)

    pmc_default = args.workload == "backbone" and args.batch == 32 and args.points == 50000
    if rank == 0:
        scenes = args.batch * world * args.steps
        out = {
            "metric": "OR scenes/sec fwd+bwd (50k pts, batch 32)",
            "value": round(scenes / elapsed, 3),
            "unit": "scenes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[1]: {args.batch} scenes/GPU x {args.points} pts x (3 xyz + 3 rgb) fp32, " + (
                    "full SA/FP stack (Pointnet2Backbone: SA 2048/0.2/64, 1024/0.4/32, 512/0.8/16, "
                    "256/1.2/16, FP x2), train mode, fwd + bwd + AdamW" if args.workload == "backbone" else
                    "stack 2a = MSG object encoder (PointNetfeat: SA 512/[0.1,0.2]/[16,32], 128/[0.2,0.4]/[32,64], "
                    "group-all), train mode, fwd + bwd + AdamW"),
                "global_batch": args.batch * world,
                "points_per_scene": args.points,
                "parallelism": f"dp{world}",
                "host_enqueue_ms_per_step": round(enqueue_ms, 3),
                "host_enqueue_note": f"first {head} timed steps (queue still empty); all {args.steps} steps incl. queue "
                                     f"back-pressure: {enqueue_all_ms:.3f} ms",
                "geometry_pipeline": "off" if prefetcher is None else
                "on: FPS / ball-query / 3-NN of batch i+1 run on a side stream during step i (one geometry per timed step)",
                "host_affinity": affinity,
            },
        }
        if serial_ms is not None:
            out["ms_per_step_without_geometry_pipeline"] = round(serial_ms, 3)
            out["config"]["ms_per_step_without_geometry_pipeline"] = round(serial_ms, 3)   # = the step LATENCY
        if fwd_ms is not None:
            out["config"]["forward_only_ms_per_step"] = round(fwd_ms, 3)
            out["config"]["forward_only_scenes_per_s"] = round(args.batch / fwd_ms * 1e3, 1)
        if timer is not None:
            rows = kernel_table(timer.summary(), sampled_steps)
            out["kernels"] = rows
            out["kernel_timing"] = {
                "method": "HIP events on the launch stream around every C-ABI call, inside the timed region",
                "sampled_steps": sampled_steps, "of_steps": args.steps,
                "note": None if prefetcher is None else
                "entries ending in @side were enqueued on the geometry-prefetch stream: they co-run with the main-stream "
                "kernels (their durations include waiting for CUs), are off the critical path and are not candidates "
                "for `roofline`; `kernels_without_geometry_pipeline` is the uncontended table"}
            main_rows = [r for r in rows if not r["kernel"].endswith("@side")]
            out["hip_kernel_ms_per_step"] = round(sum(r["ms_per_step"] for r in main_rows), 3)
            if main_rows:
                out["roofline"] = roofline_of(main_rows[0], pmc_default,
                                              pmc_files=["r06_backbone_f32x3_counters.json"] if (pmc_default and args.dtype == "f32x3") else None)
        if serial_rows is not None:
            keep = ("kernel", "calls_per_step", "ms_per_step", "avg_launch_us", "GBps", "TFLOPps", "bound", "frac")
            out["kernels_without_geometry_pipeline"] = [{k: r[k] for k in keep} for r in serial_rows]
            mlp_rows = [r for r in serial_rows if r["kernel"].startswith("pn2_mlp_")]
            if mlp_rows:
                # the same dominant kernel measured without the co-running prefetch stream: what the kernel itself achieves
                top = max(mlp_rows, key=lambda r: r["ms_per_step"])
                out["roofline_without_geometry_pipeline"] = {k: v for k, v in roofline_of(top, pmc_default).items() if k != "note"}
        if fwd_ms is not None:
            out["forward_only"] = {"ms_per_step": round(fwd_ms, 3), "scenes_per_s": round(args.batch / fwd_ms * 1e3, 1),
                                   "note": "train-mode forward (batch statistics) under no_grad, measured after the timed region; "
                                           "not part of `value`"}
        if world == 1 and not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, 64)
            if host_cpus and affinity.get("pinned"):
                os.sched_setaffinity(0, host_cpus)        # the baseline runs on all host cores again
            try:
                out["cpu_baseline"] = cpu_baseline(args.points, args.cpu_sample_scenes, threads, args.workload)
            except Exception as e:  # the baseline is informational; never lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "scenes/s", "cores": threads, "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        emit_json(out, args)

    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
