#!/bin/bash
O=gpurun_out; mkdir -p $O
{
echo "== planar RGB: round-1 kernel (LUTR_RGB2=0) vs tube kernel with the two-level queue (LUTR_RGB2=all)"
for fmt in gbrp gbrp10le gbrp16le; do for m in tetrahedral trilinear nearest; do for f in 8 16 128; do for pol in 0 all; do
  LUTR_RGB2=$pol timeout -k 10 100 python bench.py --lean --no-other --fmt $fmt --frames $f --interp $m --variant vec_lds --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-9s %-11s frames %3d  LUTR_RGB2=%-3s %6.1f Gpx/s %.3f  %s' % ('$fmt', '$m', $f, '$pol', d['value']/1e3, d['roofline']['frac'], d['config']['kernel']))"
done; done; done; done
} > $O/r03_exp21.txt 2>&1
cat $O/r03_exp21.txt
