#!/bin/bash
O=gpurun_out; mkdir -p $O
{
echo "== this round's headline kernels (base) against round 2's lutr_tile2.hip (round2), same box, same call"
tools/exp_run.sh base round2 base round2
tools/ab_dist.sh base round2
} > $O/r03_exp14.txt 2>&1
cat $O/r03_exp14.txt
