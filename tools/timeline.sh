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
# main-queue occupancy of that step: busy time, idle time between kernels, launches by duration class
main_q = rows[i0 + 1].get("Queue_Id", "?")
mk = [r for r in rows[i0:i1] if r.get("Queue_Id", "?") == main_q]
if mk:
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mk) / 1e3
    span = (int(mk[-1]["End_Timestamp"]) - int(mk[0]["Start_Timestamp"])) / 1e3
    gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(mk, mk[1:])]
    pos = [g for g in gaps if g > 0]
    small = [r for r in mk if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 15000]
    out.append("# main queue q%s: %d kernels (%d shorter than 15 us, %.0f us in total), span %.0f us, busy %.0f us, idle between "
               "kernels %.0f us (%d gaps, median %.1f us)" % (main_q, len(mk), len(small),
               sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in small) / 1e3, span, busy, sum(pos), len(pos),
               sorted(pos)[len(pos) // 2] if pos else 0.0))
    import collections
    cnt = collections.Counter(); tim = collections.Counter()
    for r in small:
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[-70:]
        cnt[n] += 1; tim[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for n, c in cnt.most_common(30):
        out.append("#   %3d x %-70s %6.1f us" % (c, n, tim[n]))
open("$O/summary.txt", "w").write("\n".join(out))
print("\n".join(out[:120]))
PY
rm -rf $O/trace
