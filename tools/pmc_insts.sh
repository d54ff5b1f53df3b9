#!/bin/bash
# tools/pmc_insts.sh NAME...: instruction counts per pixel of the headline kernel under each experimental library
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for n in "$@"; do
  O=$R/gpurun_out/prof_insts_$n; rm -rf $O; mkdir -p $O
  lib=$R/lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=$R/lut_renderer_amd/lib/liblutr.so
  export LUTR_LIBRARY=$lib
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O -- python3 bench.py --steps 4 --warmup 2 --lean --no-stats --no-other > $O/log.txt 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$O/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_yuv_tile2" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
px=256*3840*2160/64.0
print("%-6s"%"$n", "  ".join("%s %.2f"%(k.replace("SQ_INSTS_",""), (sum(v)/len(v))/px) for k,v in sorted(acc.items()) if k.startswith("SQ_INSTS")), " cycles/px-wave/SIMD %.1f"%(sum(acc["GRBM_GUI_ACTIVE"])/len(acc["GRBM_GUI_ACTIVE"])/8/(px/1024)))
PY
done
