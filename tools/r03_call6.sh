#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_gputest_f.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_gputest_f.log
tail -25 $O/r03_gputest_f.log | cut -c1-300
{
for f in "gbrp10le 128" "gbrp 128" "gbrp16le 64" "rgb24 64" "rgba 64" "rgb48le 32" "rgba64le 32"; do set -- $f
  for m in tetrahedral trilinear; do
  for nr in 0 1 2; do
    unset LUTR_NO_RGB2 LUTR_RGB2; if [ $nr = 1 ]; then export LUTR_NO_RGB2=1; fi; if [ $nr = 2 ]; then export LUTR_RGB2=all; fi
    timeout -k 10 100 python bench.py --lean --no-other --fmt $1 --frames $2 --interp $m --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['config'].get('lds_window') or {}
print('%-9s %-11s %6.1f Gpx/s %5.0f GB/s %.3f  %s  tube %s gather %s / %s' % ('$1', '$m', d['value']/1e3, d['roofline']['achieved'], d['roofline']['frac'], d['config']['kernel'], w.get('tube_tiles'), w.get('global_tiles'), w.get('tiles')))"
  done; done
done
unset LUTR_NO_RGB2 LUTR_RGB2
} > $O/r03_exp6.txt 2>&1
cat $O/r03_exp6.txt
