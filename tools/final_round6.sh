#!/bin/bash
# Round 6: everything DESIGN §5 quotes, in ONE gpurun call (same box).  Outputs: gpurun_out/r06_*.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r06_driver_runs.jsonl
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/r06_driver_runs.jsonl; done
O=gpurun_out/r06_other_workloads.jsonl
: > $O
run() { echo "# $*" >> $O; python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 >> $O; }
run --dtype f32x3
run --dtype f32x3
run --dtype bf16
run --workload encoder
run --workload sgp
run --workload sgp --scans-per-step 8
run --workload sgp --scans-per-step 8 --dtype bf16
run --batch 64 --points 200000 --no-geometry-pipeline
E=gpurun_out/r06_eval_runs.jsonl
: > $E
rune() { echo "# $*" >> $E; python bench.py --forward-eval "$@" 2>/dev/null | tail -1 >> $E; }
rune
rune
rune --eval-prefetch-depth 1
rune --no-geometry-pipeline
rune --workload encoder
rune --workload sgp
rune --workload sgp --scans-per-step 8
python tools/microbench.py 2>/dev/null | grep '^{' > gpurun_out/r06_microbench.jsonl
python tools/sa_eval_bench.py gpurun_out/r06_sa_eval.jsonl > /dev/null 2>&1
python tools/x3_gemm_bench.py gpurun_out/r06_x3_gemm.jsonl > /dev/null 2>&1
# the two backward kernels with an f32x3 form (exact | f32x3), and the inverse index by slice count
{ python tools/diag/fold_first_time.py 2>/dev/null | tail -1; python tools/pool_bwd_bench.py 2>/dev/null | grep '^{' | head -1; python tools/pool_bwd_bench.py x3 2>/dev/null | grep '^{' | head -1; } > gpurun_out/r06_x3_backward.jsonl
python tools/diag/inv_slices_time.py 2>/dev/null | grep '^(' > gpurun_out/r06_inverse_index_slices.txt
python - <<'PY'
import json
for f in ("r06_driver_runs.jsonl", "r06_other_workloads.jsonl", "r06_eval_runs.jsonl"):
    for line in open("gpurun_out/" + f):
        if line.startswith("#"):
            print(line.strip()); continue
        try:
            d = json.loads(line)
        except Exception:
            print("??", line[:80]); continue
        c = d.get("config", {})
        print("  ", d.get("dtype"), d.get("value"), d.get("unit"), "ms", d.get("ms_per_step"), "unpipelined", d.get("ms_per_step_without_geometry_pipeline"),
              "fwd", c.get("forward_only_ms_per_step"), "layers-route", c.get("layer_by_layer_exact_fp32_ms_per_step"),
              "roof", (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
