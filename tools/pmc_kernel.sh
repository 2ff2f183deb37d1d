#!/bin/bash
# PMC counters of the kernels one command launches: bash tools/pmc_kernel.sh <tag> <kernel-name-substring> -- <command...>
# Two SQ passes (8 SQ slots each); prints per-launch averages for kernels whose name contains the substring.
TAG=$1; PAT=$2; shift 3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $O/p1 --output-format csv -- "$@" > $O/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/p2 --output-format csv -- "$@" > $O/p2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVES -d $O/p3 --output-format csv -- "$@" > $O/p3.log 2>&1
python - <<PY
import csv, glob, collections, json
O="$O"; PAT="$PAT"
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.Counter())
for f in glob.glob(f"{O}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if PAT not in r["Kernel_Name"]: continue
        k=r["Kernel_Name"][:70]+"|grid"+r.get("Grid_Size","")
        tot[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
out={}
for k in tot:
    out[k]={c: tot[k][c]/cnt[k][c] for c in tot[k]}
json.dump(out, open(f"{O}/summary.json","w"), indent=1)
for k,d in out.items():
    print(k)
    for c,v in sorted(d.items()): print("   %-28s %14.0f" % (c,v))
PY
tail -3 $O/p1.log
