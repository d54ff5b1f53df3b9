#!/bin/bash
# tools/ubench/cycles_sq.sh: the same op_rates2 kernels measured with the SQ's own counters -- cycles per VALU wave-instruction and SIMD =
# (SQ_WAVE_CYCLES x 4 / SQ_WAVES: the waves' lifetime, all resident at once) x 1024 SIMDs / SQ_INSTS_VALU.  This is the unit
# tools/summarize_profile.py's valu_issue uses for the product kernels, so the 2.5-cycle floor can be checked in that unit.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_ub4; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O/a -- ./tools/ubench/op_rates2 > $O/a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/b -- ./tools/ubench/op_rates2 > $O/b.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/c -- ./tools/ubench/op_rates2 > $O/c.log 2>&1
python3 - <<PY
import csv,glob,re,collections
acc=collections.OrderedDict()
for f in glob.glob("$O/a/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m=re.match(r"k_(\w+)\(",r["Kernel_Name"])
        if m: acc.setdefault(m.group(1),collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
gui=collections.OrderedDict()
for f in glob.glob("$O/b/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m=re.match(r"k_(\w+)\(",r["Kernel_Name"])
        if m: gui.setdefault(m.group(1),[]).append(float(r["Counter_Value"]))
dur=collections.OrderedDict()
for f in glob.glob("$O/c/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        m=re.match(r"k_(\w+)\(",r["Kernel_Name"])
        if m: dur.setdefault(m.group(1),[]).append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("%-18s %10s %10s %10s %10s" % ("kernel", "SQ cyc/inst", "GUI cyc/slot", "SQ GHz", "GUI GHz"))
for k,c in acc.items():
    w=min(c["SQ_WAVES"]); life=4.0*min(c["SQ_WAVE_CYCLES"])/w; insts=min(c["SQ_INSTS_VALU"])
    g=min(gui.get(k,[0]))/8.0; t=min(dur.get(k,[1]))
    print("%-18s %10.2f %10.2f %10.2f %10.2f" % (k, life*1024/insts if insts else 0, g/640000.0, life/t, g/t))
PY
