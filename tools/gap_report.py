"""Gaps on the main stream of a pipelined step from a rocprofv3 kernel trace (csv): which kernels start after idle time.
usage: python tools/gap_report.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
# main queue = the one with the most mlp_gemm kernels
mainq = max(byq, key=lambda q: sum("mlp_gemm_kernel" in k[2] for k in byq[q]))
import re
ks = sorted(byq[mainq])
t_lo = ks[0][0] + 0.6 * (ks[-1][1] - ks[0][0])      # the timed steps are at the end of the run
ks = [k for k in ks if k[0] >= t_lo]
def short(n):
    m = re.search(r"([A-Za-z_0-9]+_kernel\w*|rocclr_\w+|[A-Za-z_]+Functor\w*|multi_tensor\w+|reduce_kernel|CatArray\w+)", n)
    return m.group(1) if m else n[:50]
gaps = collections.defaultdict(lambda: [0, 0])
tot_gap = tot_busy = 0
for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
    g = s1 - e0
    if 0 < g < 200000:          # ignore the pauses between phases of the bench
        key = short(n1)
        gaps[key][0] += 1; gaps[key][1] += g
        tot_gap += g
    tot_busy += e1 - s1
print("main queue", mainq, "kernels", len(ks), "busy ms", tot_busy / 1e6, "gap ms", tot_gap / 1e6)
for k, (n, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{n:5d} {g/1e3:9.1f} us total {g/n/1e3:7.2f} us avg  before {k}")
