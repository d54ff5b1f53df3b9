#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/r03_gputest_c.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_gputest_c.log
tail -3 $O/r03_gputest_c.log
{
echo "== padded tube planes (base) vs unpadded (LUTR_TUBE_NOPAD), strict, 64 frames"
for cfg in "6 256 70 0" "6 256 70 1" "7 128 85 0" "7 128 85 1"; do set -- $cfg
  for d in natural noise16 vivid; do
    if [ "$4" = 1 ]; then export LUTR_TUBE_NOPAD=1; else unset LUTR_TUBE_NOPAD; fi
    LUTR_TUBE_H=$1 LUTR_MIN_WIN=$2 LUTR_TUBE_PCT=$3 timeout -k 10 100 python bench.py --lean --no-other --dist $d --frames 64 --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['config'].get('lds_window') or {}
print('H=$1 minwin=$2 nopad=$4 %-8s strict %6.1f Gpx/s  tube %s level2 %s restage %s gather %s of %s tiles' % ('$d', d['value']/1e3, w.get('tube_tiles'), w.get('level2_tiles'), w.get('misses'), w.get('global_tiles'), w.get('tiles')))"
  done
done
unset LUTR_TUBE_NOPAD
echo "== 256 frames, both precisions"
tools/exp_run.sh base br
echo "== content, base (b-g, padded) vs br (b-r tube of round 2, padded)"
tools/ab_dist.sh base br
} > $O/r03_exp3.txt 2>&1
cat $O/r03_exp3.txt
