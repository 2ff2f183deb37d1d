#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_benches.jsonl
: > $O
run() { echo "# $*" >> $O; python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 >> $O; }
run --dtype bf16
run --workload encoder
run --workload sgp
run --workload sgp --scans-per-step 8
run --workload sgp --scans-per-step 8 --whole-batch-statistics
run --workload sgp --scans-per-step 8 --dtype bf16
run --workload sgp --scans-per-step 8 --dtype bf16 --whole-batch-statistics
run --workload sgp --scans-per-step 32 --dtype bf16
run --workload sgp --scans-per-step 32 --dtype bf16 --whole-batch-statistics
run --workload sgp --scans-per-step 32 --whole-batch-statistics
run --workload sgp --scans-per-step 8 --dtype bf16 --whole-batch-statistics --with-prep
run --batch 64 --points 200000 --no-geometry-pipeline
