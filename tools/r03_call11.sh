#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_small; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --lean --no-stats --no-other --size 1080p --frames 1 --variant vec_lds --steps 200 --warmup 20 > $O/kt.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$O/kt/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:4]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
for f in glob.glob("$O/kt/*/*_kernel_trace.csv"):
    rows=[r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    last=rows[-12:]
    t0=int(last[0]["Start_Timestamp"])
    for r in last: print("%-40s start %8.1f us  dur %6.1f us  LDS %s" % (r["Kernel_Name"][:40], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("LDS_Block_Size")))
PY
tail -1 $O/kt.log | cut -c1-200
