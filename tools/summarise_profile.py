"""gpurun_out/prof_<tag>/ (tools/profile_round.sh) -> profiles/<tag>_kernel_stats.{csv,md} + profiles/<tag>_counters.json.

Counter normalisation on gfx950 (checked against launches with a known MFMA count): GRBM_GUI_ACTIVE comes back summed
over the 8 XCDs, so a kernel's cycle count is GUI_ACTIVE / 8 (which, divided by the traced duration, gives the clock
the chip actually ran at: 1.7-1.8 GHz under these kernels, not the 2.4 GHz the peak figures assume);
SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs.  FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section)."""
import csv
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(repo, "gpurun_out", f"prof_{tag}")
dst = os.path.join(repo, "profiles")


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_keys import keys  # noqa: E402  (family | instantiation | entry:<C-ABI entry point>)


shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "kernel_stats.md"), os.path.join(dst, f"{tag}_kernel_stats.md"))
dur = {}
for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))):
    for k in keys(r["Name"]):
        d = dur.setdefault(k, [0, 0.0])
        d[0] += int(r["Calls"])
        d[1] += float(r["TotalDurationNs"])
hbm = json.load(open(os.path.join(src, "hbm_traffic_per_kernel.json")))
mf = json.load(open(os.path.join(src, "mfma_util_per_kernel.json")))
out = {"commands": {"kernel_stats": "rocprofv3 --kernel-trace --stats -- <bench command>", "hbm": hbm["command"], "mfma": mf["command"]},
       "corrections": "FETCH_SIZE x2 (gfx950); GRBM_GUI_ACTIVE / 8 XCDs = kernel cycles; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                      "(kernel cycles x 1024 SIMDs); effective clock = kernel cycles / traced average duration",
       "keys": "kernel family (all instantiations) | instantiation `mlp_gemm_kernel<PRO,EPI>` | `entry:<C-ABI entry point>` = the "
               "launches bench.py's per-entry-point rows (alg_bytes_per_launch) describe (tools/kernel_keys.py)",
       "kernels": {}}
for k in sorted(set(hbm["kernels"]) | set(mf["kernels"])):
    e = {}
    if k in dur and dur[k][0]:
        e["calls_traced"] = dur[k][0]
        e["avg_us"] = round(dur[k][1] / dur[k][0] / 1e3, 2)
    if k in hbm["kernels"]:
        h = hbm["kernels"][k]
        e["hbm_MB_per_launch"] = round(h["hbm_bytes_per_launch"] / 1e6, 2)
        if "avg_us" in e and e["avg_us"] > 0:
            e["hbm_GBps"] = round(h["hbm_bytes_per_launch"] / (e["avg_us"] * 1e-6) / 1e9, 1)
    if k in mf["kernels"] and mf["kernels"][k].get("GRBM_GUI_ACTIVE"):
        m = mf["kernels"][k]
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        e["kernel_cycles"] = round(cyc)
        e["mfma_busy_frac"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0), 4)
        if "avg_us" in e and e["avg_us"] > 0:
            e["effective_clock_GHz"] = round(cyc / (e["avg_us"] * 1e-6) / 1e9, 3)
        e["valu_insts_per_mfma_busy_cycle"] = (round(m.get("SQ_INSTS_VALU", 0.0) / m["SQ_VALU_MFMA_BUSY_CYCLES"], 3)
                                                if m.get("SQ_VALU_MFMA_BUSY_CYCLES") else None)
    out["kernels"][k] = e
json.dump(out, open(os.path.join(dst, f"{tag}_counters.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    if e.get("avg_us", 0) > 50:
        print(f"{k:42s} {e}")
