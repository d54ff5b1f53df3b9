#!/bin/bash
# fast kernels with mixed tiles (mixfast build) by the number of outlier lanes allowed, vs without (base)
O=gpurun_out; mkdir -p $O
{
echo "== fast kernels: mixed tiles off (base) vs on (mixfast) with LUTR_MIX_MAX; UHD yuv420p10le tetrahedral, fast Gpx/s (strict beside it)"
for cfg in "natural 256" "noise8 64" "noise16 64" "vivid 64"; do set -- $cfg
  for n in base mixfast:4 mixfast:8 mixfast:16 mixfast:63; do
    lib=lut_renderer_amd/lib/liblutr_${n%%:*}.so; [ "${n%%:*}" = base ] && lib=lut_renderer_amd/lib/liblutr.so
    mm=${n##*:}; [ "$n" = base ] && mm=63
    LUTR_MIX_MAX=$mm LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --precision fast --no-other --dist $1 --frames $2 --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['config'].get('lds_window') or {}
print('%-8s frames %3d %-10s fast %6.1f Gpx/s  tube %s mixed %s level2 %s restage %s gather %s' % ('$1', $2, '$n', d['value']/1e3, w.get('tube_tiles'), w.get('mixed_tiles'), w.get('level2_tiles'), w.get('misses'), w.get('global_tiles')))"
  done
done
} > $O/r03_exp34.txt 2>&1
cat $O/r03_exp34.txt
