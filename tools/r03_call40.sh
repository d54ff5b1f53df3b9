#!/bin/bash
# machine-scheduler options for the headline translation unit
O=gpurun_out; mkdir -p $O
{
echo "== lutr_tile2.hip (10-bit 4:2:0 unit) compiled with other scheduler options; strict | fast Gpx/s, 256 UHD frames, two rounds"
for rep in 1 2; do for n in base relax mclause maxilp; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('%-8s strict %6.1f  fast %6.1f' % ('$n', d['value']/1e3, o.get('Mpx_s',0)/1e3))"
done; done
} > $O/r03_exp40.txt 2>&1
cat $O/r03_exp40.txt
