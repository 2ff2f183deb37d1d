#!/bin/bash
# bash tools/bench_n.sh <n> [bench args...]: run bench.py n times, print ms_per_step of each
n=$1; shift
for i in $(seq 1 $n); do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-serial-reference --no-forward-only "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done
