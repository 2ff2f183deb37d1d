#!/bin/bash
# Everything DESIGN §5 quotes for a round, in ONE gpurun call (same box): the driver's command three times, the other
# workloads (all_benches.sh), the micro-benchmarks, the FPS variants and the GCN timings.  Outputs: gpurun_out/final_*.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/final_driver_runs.jsonl
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/final_driver_runs.jsonl; done
python bench.py --no-geometry-pipeline --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final_unpipelined.json
bash tools/all_benches.sh
python tools/microbench.py 2>/dev/null | grep '^{' > gpurun_out/final_microbench.jsonl
python tools/fps_multi_time.py 2>/dev/null | grep '^{' > gpurun_out/final_fps_multi_time.jsonl
python tools/gcn_time.py 2>/dev/null | grep '^{' > gpurun_out/final_gcn_time.jsonl
python tools/gcn_kernel_time.py 2>/dev/null | grep '^{' > gpurun_out/final_gcn_kernel_time.jsonl
python - <<'PY'
import json
for f in ("final_driver_runs.jsonl", "final_unpipelined.json", "final_benches.jsonl"):
    for line in open("gpurun_out/" + f):
        if line.startswith("#"):
            print(line.strip()); continue
        try:
            d = json.loads(line)
        except Exception:
            print("??", line[:80]); continue
        c = d.get("config", {})
        print(f, d.get("value"), d.get("unit"), d.get("ms_per_step"), "fwd", c.get("forward_only_ms"), "host", c.get("host_enqueue_ms_per_step"),
              "roof", (d.get("roofline") or {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
