#!/bin/bash
# row strides re-read from the kernel-argument segment per tile (kstride) vs kept in SGPRs / VGPR lanes (base)
O=gpurun_out; mkdir -p $O
{
echo "== LUTR_T2_KSTRIDE: strict | fast Gpx/s, 256 UHD frames, three rounds; then 8 frames"
for rep in 1 2 3; do for n in base kstride; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('%-8s strict %6.1f  fast %6.1f' % ('$n', d['value']/1e3, o.get('Mpx_s',0)/1e3))"
done; done
for n in base kstride; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --frames 8 --variant vec_lds --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('%-8s 8 frames strict %6.1f  fast %6.1f' % ('$n', d['value']/1e3, o.get('Mpx_s',0)/1e3))"
done
} > $O/r03_exp39.txt 2>&1
cat $O/r03_exp39.txt
