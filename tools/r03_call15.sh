#!/bin/bash
O=gpurun_out; mkdir -p $O
{
echo "== windows: (b-g) padded [base] / (b-r) padded [winbr] / (b-g) unpadded [winnopad] / (b-r) unpadded [winbrnopad] / round 2 kernel"
for d in natural vivid noise16; do for n in base winbr winnopad winbrnopad round2; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --dist $d --frames 64 --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}; w=d['config'].get('lds_window') or {}
print('%-8s %-11s strict %6.1f fast %6.1f  strict tiles: tube %s mixed %s level2 %s restage %s gather %s' % ('$d', '$n', d['value']/1e3, o.get('Mpx_s',0)/1e3, w.get('tube_tiles'), w.get('mixed_tiles'), w.get('level2_tiles'), w.get('misses'), w.get('global_tiles')))"
done; done
echo "== strict H=6 / 277-node windows with the same builds (vivid)"
for n in base winbr; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_TUBE_H=6 LUTR_MIN_WIN=256 LUTR_TUBE_PCT=70 LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --no-other --dist vivid --frames 64 --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['config'].get('lds_window') or {}
print('vivid H6 %-8s strict %6.1f  tube %s mixed %s level2 %s restage %s gather %s' % ('$n', d['value']/1e3, w.get('tube_tiles'), w.get('mixed_tiles'), w.get('level2_tiles'), w.get('misses'), w.get('global_tiles')))"
done
} > $O/r03_exp15.txt 2>&1
cat $O/r03_exp15.txt
