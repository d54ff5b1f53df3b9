#!/bin/bash
# strict kernel: how many outlier lanes a mixed tile may have (LUTR_MIX_MAX), final build
O=gpurun_out; mkdir -p $O
{
echo "== strict, UHD yuv420p10le tetrahedral: LUTR_MIX_MAX (lanes outside the tube a mixed tile may have; 0 = no mixed tiles)"
for cfg in "natural 256" "noise8 64" "noise16 64" "vivid 64"; do set -- $cfg; for mm in 0 2 4 8 16 63; do
  LUTR_MIX_MAX=$mm timeout -k 10 100 python bench.py --lean --no-other --dist $1 --frames $2 --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['config'].get('lds_window') or {}
print('%-8s frames %3d mix_max %2d  %6.1f Gpx/s  tube %s mixed %s level2 %s restage %s gather %s of %s' % ('$1', $2, $mm, d['value']/1e3, w.get('tube_tiles'), w.get('mixed_tiles'), w.get('level2_tiles'), w.get('misses'), w.get('global_tiles'), w.get('tiles')))"
done; done
} > $O/r03_exp29.txt 2>&1
cat $O/r03_exp29.txt
