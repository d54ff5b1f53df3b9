#!/bin/bash
O=gpurun_out; mkdir -p $O
{
echo "== mixed tiles again, now that a restage is cheap (closed-form stride): base (mix_max 63) / LUTR_MIX_MAX / nomix / round2"
for d in natural vivid noise8 noise16; do
 for cfg in "base 63" "base 32" "base 16" "base 8" "nomix 0" "round2 0"; do set -- $cfg
  lib=lut_renderer_amd/lib/liblutr_$1.so; [ "$1" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_MIX_MAX=$2 LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --dist $d --frames 64 --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}; w=d['config'].get('lds_window') or {}
print('%-8s %-7s mix_max %2d  strict %6.1f fast %6.1f  strict tiles: tube %s mixed %s level2 %s restage %s gather %s' % ('$d', '$1', $2, d['value']/1e3, o.get('Mpx_s',0)/1e3, w.get('tube_tiles'), w.get('mixed_tiles'), w.get('level2_tiles'), w.get('misses'), w.get('global_tiles')))"
 done
done
} > $O/r03_exp16.txt 2>&1
cat $O/r03_exp16.txt
