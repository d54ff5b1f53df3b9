#!/bin/bash
# 65^3 lattices (BASELINE config 3): a tube after all?  (round 2 said no: old axes, no mixed tiles, unpadded planes)
O=gpurun_out; mkdir -p $O
{
echo "== UHD yuv420p10le tetrahedral, 65^3, 128 frames: no tube (default) vs LUTR_TUBE_H = 3..7; strict | fast Gpx/s"
for dist in natural noise8 vivid; do for h in none 3 4 5 6 7; do
  env=""; [ "$h" != none ] && env="LUTR_TUBE_H=$h LUTR_TUBE_PCT=90"
  env $env timeout -k 10 100 python bench.py --lean --lut 65 --dist $dist --frames 128 --steps 20 --warmup 6 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}; w=d['config'].get('lds_window') or {}
print('%-8s H=%-4s strict %6.1f  fast %6.1f  %s tube %s mixed %s level2 %s restage %s gather %s of %s' % ('$dist', '$h', d['value']/1e3, o.get('Mpx_s',0)/1e3, d['config']['kernel'], w.get('tube_tiles'), w.get('mixed_tiles'), w.get('level2_tiles'), w.get('misses'), w.get('global_tiles'), w.get('tiles')))"
done; done
} > $O/r03_exp31.txt 2>&1
cat $O/r03_exp31.txt
