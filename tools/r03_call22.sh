#!/bin/bash
O=gpurun_out; mkdir -p $O
{
echo "== s_setprio around the phases of a tile: base / prio1 (memory phase high) / prio2 (body high)"
tools/exp_run.sh base prio1 prio2 base prio1 prio2
} > $O/r03_exp22.txt 2>&1
cat $O/r03_exp22.txt
