# end-of-round check on one box: smoke(), the driver's command three times, the torchrun form once
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
: > gpurun_out/final_driver_runs.jsonl
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/final_driver_runs.jsonl; done
python bench.py 2>/dev/null | tail -1 >> gpurun_out/final_driver_runs.jsonl
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/final_torchrun.json
python - <<'PY'
import json
for f in ("final_driver_runs.jsonl", "final_torchrun.json"):
    for l in open("gpurun_out/" + f):
        d = json.loads(l); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
PY
