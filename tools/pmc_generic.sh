#!/bin/bash
# SQ counter passes over a python script: bash tools/pmc_generic.sh <script.py> <kernel-substring> <outtag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$3
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $O/p1 --output-format csv -- python $R/$1 > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA -d $O/p2 --output-format csv -- python $R/$1 > $O/p2.log 2>&1
python - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % p, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "$2" not in k: continue
            key = (k[30:95], r["Grid_Size"])
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(key, r["Counter_Name"])] += 1
    for key, d in agg.items():
        print(p, key, {c: round(v / cnt[(key, c)] / 1e6, 2) for c, v in d.items()}, "(millions)")
PY
tail -2 $O/p1.log
