#!/bin/bash
# fast trilinear: node + r-difference records (12 bytes, Node::rec) -- parity, tube width, and the trilinear / tetrahedral A/B of config 4
O=gpurun_out; mkdir -p $O
{
echo "== fast trilinear with records: UHD yuv420p10le 256 frames, Gpx/s strict | fast, by tube width (LUTR_TUBE_H) and content"
for dist in natural vivid noise16; do for h in auto 7 6 8; do
  env=""; [ "$h" != auto ] && env="LUTR_TUBE_H=$h"
  env $env LUTR_TUBE_PCT=90 timeout -k 10 100 python bench.py --lean --interp trilinear --dist $dist --frames 128 --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('trilinear   %-8s H=%-4s strict %6.1f  fast %6.1f  %s' % ('$dist', '$h', d['value']/1e3, o.get('Mpx_s',0)/1e3, o.get('kernel')))"
done; done
for size in uhd 8k; do for m in tetrahedral trilinear; do
  f=256; [ $size = 8k ] && f=64
  timeout -k 10 100 python bench.py --lean --size $size --interp $m --frames $f --steps 20 --warmup 6 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('%-4s %-11s frames %3d strict %6.1f (%.3f)  fast %6.1f (%.3f)  %s' % ('$size', '$m', $f, d['value']/1e3, d['roofline']['frac'], o.get('Mpx_s',0)/1e3, o.get('frac',0), o.get('kernel')))"
done; done
} > $O/r03_exp28.txt 2>&1
cat $O/r03_exp28.txt
