#!/bin/bash
# tools/pmc_tcc_ea.sh TAG [bench args]: the L2 -> fabric side of a launch (request / stall counters of the planar-RGB winner):
# read and write requests to the memory side, their credit stalls and average levels, in separate --pmc passes.
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tccea_$TAG; rm -rf $O; mkdir -p $O; cd $R
ARGS="--steps 12 --warmup 4 --lean --no-stats --no-other $@"
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUBBLE_sum" \
           "TCC_REQ_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 bench.py $ARGS > $O/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.OrderedDict()
for f in sorted(glob.glob("$O/p*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "lutr::" in r["Kernel_Name"] and "make_lat16" not in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print("$TAG: per launch (mean over the launches of the pass)")
for k,v in acc.items(): print("  %-40s %16.0f" % (k, sum(v)/len(v)))
PY
