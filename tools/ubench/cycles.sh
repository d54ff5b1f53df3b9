#!/bin/bash
# tools/ubench/cycles.sh: op_rates2 under rocprofv3 --pmc GRBM_GUI_ACTIVE -> shader cycles per wave-instruction per SIMD
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_ub3; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/a -- ./tools/ubench/op_rates2 > $O/a.log 2>&1
python3 - <<PY
import csv,glob,re,collections
acc=collections.OrderedDict()
for f in glob.glob("$O/a/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m=re.match(r"k_(\w+)\(",r["Kernel_Name"])
        if m: acc.setdefault(m.group(1),[]).append(float(r["Counter_Value"]))
# 256 CUs x 4 SIMDs, 8 waves each, iters from the log; GUI_ACTIVE is summed over 8 XCDs
import os
log=open("$O/a.log").read()
for k,v in acc.items():
    best=min(v)/8.0
    print("%-18s %7.2f cycles per slot (one instruction of a 1-op kernel, the pair of a 2-op mix)"%(k,best/640000.0))
PY
