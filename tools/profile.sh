#!/bin/bash
# tools/profile.sh TAG [bench args...] -- rocprofv3 passes for bench.py on the GPU box (run through gpurun).
# Pass 1: --kernel-trace --stats (per-kernel durations).  Pass 2/3: --pmc FETCH_SIZE / WRITE_SIZE alone
# (TCC slots do not hold both; counters are never combined with trace domains other than kernel-trace).
# Pass 4: SQ counters.  Output: gpurun_out/prof_$TAG/...
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O; cd $R
ARGS="--steps 12 --warmup 4 --lean --no-stats --no-other $@"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py $ARGS > $O/kt.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py $ARGS > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py $ARGS > $O/write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- python3 bench.py $ARGS > $O/sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -- python3 bench.py $ARGS > $O/tcc.log 2>&1
# the shader clock's own count of the launch (summed over the 8 XCDs): the unit tools/ubench/cycles.sh calibrated the VALU issue rate in
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/gui -- python3 bench.py $ARGS > $O/gui.log 2>&1
grep -h '"metric"' $O/kt.log | tail -1 > $O/bench_line.json
ls $O/*/*/ | head -40
