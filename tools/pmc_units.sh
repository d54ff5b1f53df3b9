cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_pmc2; rm -rf $O; mkdir -p $O; cd $R
ARGS="--steps 6 --warmup 2 --lean --no-stats --no-other"
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python3 bench.py $ARGS > $O/a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/b -- python3 bench.py $ARGS > $O/b.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("a","b"):
    acc=collections.defaultdict(list)
    for f in glob.glob("$O/%s/*/*_counter_collection.csv"%d):
        for r in csv.DictReader(open(f)):
            if "k_yuv_tile2" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(k, sum(v)/len(v), len(v))
PY
tail -2 $O/a.log | cut -c1-300
