#!/bin/bash
# Round profile of `python bench.py` on the GPU box: kernel-trace statistics + HBM traffic counters
# (separate rocprofv3 passes, as MI355X_MICROARCH.md prescribes).  Usage: bash tools/profile_round.sh <tag>
# Outputs land in gpurun_out/prof_<tag>/; copy the summaries you keep into profiles/.
# Optional further arguments are appended to the bench command (e.g. `--workload sgp --scans-per-step 8 --dtype bf16`).
TAG=${1:-r02}
shift
EXTRA="$@"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-serial-reference --no-forward-only $EXTRA"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- $CMD > $O/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- $CMD > $O/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- $CMD > $O/write.log 2>&1
# matrix-core utilisation: SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD with the MFMA pipe busy, GRBM_GUI_ACTIVE the
# kernel's cycles (MI355X_MICROARCH.md, rocprofv3 PMC slots): busy fraction = MFMA_BUSY / (GUI_ACTIVE x 1024 SIMDs)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/mfma --output-format csv -- $CMD > $O/mfma.log 2>&1
python - <<PY
import csv, glob, json, re, collections, sys
O = "$O"
sys.path.insert(0, "$R/tools")
from kernel_keys import keys      # family | instantiation (mlp_gemm_kernel<PRO,EPI>) | entry:<C-ABI entry point>
def agg(sub, counter):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            for k in keys(r["Kernel_Name"]):
                tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    return tot, cnt
fe, fc = agg("fetch", "FETCH_SIZE"); wr, wc = agg("write", "WRITE_SIZE")
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- " + "$CMD".replace("$R/", "") + " (separate passes)",
       "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section); counters in KiB", "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    n = max(fc[k], wc[k], 1)
    f = fe[k] * 1024 / max(fc[k], 1); w = wr[k] * 1024 / max(wc[k], 1)
    out["kernels"][k] = {"launches_traced": n, "fetch_bytes_per_launch_raw": f, "fetch_bytes_per_launch_corrected_x2": 2 * f,
                         "write_bytes_per_launch": w, "hbm_bytes_per_launch": 2 * f + w}
json.dump(out, open(f"{O}/hbm_traffic_per_kernel.json", "w"), indent=1)
# MFMA pass
mf = {}
for name in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"):
    tot, cnt = agg("mfma", name)
    for k in tot:
        mf.setdefault(k, {})[name] = tot[k] / max(cnt[k], 1)
for k, d in mf.items():
    if d.get("GRBM_GUI_ACTIVE"):
        d["mfma_busy_frac_of_1024_simds"] = round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (d["GRBM_GUI_ACTIVE"] * 1024.0), 4)
json.dump({"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- " + "$CMD".replace("$R/", ""),
           "note": "per-launch averages; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)", "kernels": mf},
          open(f"{O}/mfma_util_per_kernel.json", "w"), indent=1)
# kernel stats: name, calls, total ns, avg ns, pct
for f in glob.glob(f"{O}/stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(f"{O}/kernel_stats.md", "w") as g:
        g.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:40]:
            g.write("| %s | %s | %.3f | %.1f | %s |\n" % (r["Name"][:90].replace("|", "/"), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                         float(r["AverageNs"]) / 1e3, r["Percentage"]))
    import shutil; shutil.copy(f, f"{O}/kernel_stats.csv")
print(open(f"{O}/kernel_stats.md").read()[:3000])
PY
