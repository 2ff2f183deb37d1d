#!/bin/bash
# kernel timeline of the pipelined step: bash tools/timeline.sh  (GPU box) -> gpurun_out/timeline/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/timeline
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/trace --output-format csv -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-serial-reference --no-forward-only "$@" > $O/run.log 2>&1
python - <<PY
import csv, glob, re
f = glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"([A-Za-z_0-9]+_kernel)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]
# find the last fps_coop launch and print the timeline around the last full step
coop = [i for i, r in enumerate(rows) if "fps_coop" in r["Kernel_Name"]]
i0 = coop[-2]
t0 = int(rows[i0]["Start_Timestamp"])
i1 = coop[-1]
out = []
for r in rows[i0:i1 + 1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    if e - s > 40 or "fps" in r["Kernel_Name"]:
        out.append("%9.1f %9.1f %8.1f  q%s  %s" % (s, e, e - s, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
open("$O/summary.txt", "w").write("\n".join(out))
print("\n".join(out[:120]))
PY
rm -rf $O/trace
