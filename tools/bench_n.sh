#!/bin/bash
# The driver's scaling sequence on one node: N = 1, 2, 4, 8 ranks (one per GPU, RCCL over xGMI), one JSON line each.
#   bash tools/bench_n.sh [extra bench.py arguments]        -> gpurun_out/scale_n{1,2,4,8}.json
# N = 1 runs bench.py directly; N > 1 through torch.distributed.run exactly like the driver does.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "skip N=$N: $NGPU GPU(s) visible" >&2; continue; fi
  if [ "$N" -eq 1 ]; then
    python $R/bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $R/gpurun_out/scale_n$N.json 2> $R/gpurun_out/scale_n$N.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29400 + N)) \
      $R/bench.py --gpus $N --steps 20 --warmup 5 --no-cpu-baseline "$@" > $R/gpurun_out/scale_n$N.json 2> $R/gpurun_out/scale_n$N.err
  fi
  tail -c 400 $R/gpurun_out/scale_n$N.json; echo
done
