#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/r03_gputest_q.log 2>&1 || { echo "pytest failed"; tail -15 $O/r03_gputest_q.log; exit 1; }
tail -2 $O/r03_gputest_q.log
{
echo "== trilinear with strides that respect the 16-byte nodes' banking (mod 16): fused YUV (LUTR_TUBE_NOPAD=1 = unpadded planes), planar / packed RGB"
for np in 0 1; do
  if [ $np = 1 ]; then export LUTR_TUBE_NOPAD=1; else unset LUTR_TUBE_NOPAD; fi
  timeout -k 10 100 python bench.py --lean --interp trilinear 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('yuv420p10le trilinear nopad=$np  strict %6.1f  fast %6.1f  %s' % (d['value']/1e3, o.get('Mpx_s',0)/1e3, d['config']['kernel']))"
done
unset LUTR_TUBE_NOPAD
for cfg in "gbrp 0" "gbrp all" "gbrp10le 0" "gbrp10le all" "rgb24 all" "rgba all" "rgb48le all"; do set -- $cfg
  LUTR_RGB2=$2 timeout -k 10 100 python bench.py --lean --no-other --fmt $1 --frames 128 --interp trilinear --variant vec_lds --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-9s trilinear LUTR_RGB2=%-3s %6.1f Gpx/s %.3f  %s' % ('$1', '$2', d['value']/1e3, d['roofline']['frac'], d['config']['kernel']))"
done
} > $O/r03_exp23.txt 2>&1
cat $O/r03_exp23.txt
