#!/bin/bash
# Kernel trace of the 1-scan-per-step scene-graph training step: GPU time per step by kernel + the timeline's gaps.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sgp_trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload sgp --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-serial-reference --no-forward-only $@"
timeout 600 rocprofv3 --kernel-trace -d $O/kt --output-format csv -- $CMD > $O/kt.log 2>&1
tail -1 $O/kt.log | cut -c1-300
python - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$O/kt/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
print("kernels", n)
# last 3 steps: take the last half of the launches
t0 = int(rows[0]["Start_Timestamp"])
# find step period by locating a marker kernel (multi_tensor_apply = AdamW) occurrences
marks = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
# group AdamW launches into steps (gaps > 1 ms)
steps = []
for i in marks:
    t = int(rows[i]["Start_Timestamp"])
    if not steps or t - steps[-1][-1][1] > 1_000_000: steps.append([])
    steps[-1].append((i, t))
print("optimizer bursts", len(steps))
if len(steps) >= 3:
    a, b = steps[-3][-1][0] + 1, steps[-1][-1][0] + 1
    sel = rows[a:b]
    span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
    # union of intervals (streams overlap)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
    u = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: u += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    u += ce - cs
    print(f"2 steps: span {span/2e6:.3f} ms/step, launches {len(sel)/2:.0f}/step, kernel time sum {busy/2e6:.3f}, union {u/2e6:.3f} ms/step, idle {(span-u)/2e6:.3f}")
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        k = r["Kernel_Name"][:90]
        agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
    for k, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:40]:
        print(f"{t/2e3:9.1f} us/step {c/2:6.1f} x  {k}")
    # gaps > 20 us on the union timeline
    gaps = []
    cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce:
            gaps.append(s - ce); cs, ce = s, e
        else: ce = max(ce, e)
    big = [g for g in gaps if g > 20000]
    print("gaps >20us:", len(big) / 2, "per step, total", sum(big) / 2e6, "ms/step; all gaps", sum(gaps) / 2e6)
PY
