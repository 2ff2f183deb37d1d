#!/bin/bash
# Sum of kernel durations per step from a rocprofv3 kernel trace of a bench command (is a step GPU-bound or host-bound?)
# usage: bash tools/kernel_time_sum.sh <tag> <bench args...>
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ktrace_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing "$@" > $O/log.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
sel = rows[int(n * 0.35):int(n * 0.75)]                    # the timed steps (middle of the launch sequence)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
u, cs, ce = 0, iv[0][0], iv[0][1]
for a, b in iv[1:]:
    if a > ce: u += ce - cs; cs, ce = a, b
    else: ce = max(ce, b)
u += ce - cs
span = max(e for _, e in iv) - iv[0][0]
print("window %.1f ms: %d kernels, sum of durations %.1f ms, GPU busy (union) %.1f ms = %.0f%% of the window" % (span / 1e6, len(sel), busy / 1e6, u / 1e6, 100.0 * u / span))
PY
tail -2 $O/log.txt | cut -c1-200
