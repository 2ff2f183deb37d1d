#!/bin/bash
# HBM traffic of the kernels one command launches: bash tools/pmc_hbm.sh <tag> <kernel-name-substring> -- <command...>
# Two separate rocprofv3 passes (FETCH_SIZE, WRITE_SIZE: KiB per launch; FETCH_SIZE x2 on gfx950 — MI355X_MICROARCH.md, HBM
# section); prints and stores the per-launch average of the kernels whose name contains the substring.
TAG=$1; PAT=$2; shift 3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/hbm_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- "$@" > $O/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- "$@" > $O/write.log 2>&1
python - <<PY
import csv, glob, collections, json
O="$O"; PAT="$PAT"
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.Counter())
for sub, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if PAT not in r["Kernel_Name"] or r["Counter_Name"] != name: continue
            k=r["Kernel_Name"][:90]
            tot[k][name]+=float(r["Counter_Value"]); cnt[k][name]+=1
out={}
for k in tot:
    f=tot[k]["FETCH_SIZE"]*1024/max(cnt[k]["FETCH_SIZE"],1); w=tot[k]["WRITE_SIZE"]*1024/max(cnt[k]["WRITE_SIZE"],1)
    out[k]={"launches": max(cnt[k].values()), "fetch_bytes_raw": f, "fetch_bytes_x2": 2*f, "write_bytes": w, "hbm_bytes_per_launch": 2*f+w}
json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- " + " ".join("$@".split()),
           "correction": "FETCH_SIZE x2 on gfx950; counters in KiB", "kernels": out}, open(f"{O}/summary.json","w"), indent=1)
for k,d in out.items(): print(k, {a: round(b/1e6,2) if a!="launches" else b for a,b in d.items()})
PY
