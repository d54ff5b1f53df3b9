#!/bin/bash
# HEAD against the library of commit 03d6384 (before the shared prelut, the self-resetting queue and the padded whole-lattice strides put
# more fields into the kernels' argument structs): did the extra SGPR pressure cost the headline anything?
O=gpurun_out; mkdir -p $O
{
echo "== HEAD (base) vs commit 03d6384 (old): strict | fast Gpx/s, three rounds of 256 UHD frames; then trilinear, 1080p 8-bit, 8 frames"
for rep in 1 2 3; do for n in base old; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('%-5s tetrahedral 256  strict %6.1f  fast %6.1f' % ('$n', d['value']/1e3, o.get('Mpx_s',0)/1e3))"
done; done
for cfg in "--interp trilinear" "--size 1080p --fmt yuv420p --frames 512" "--frames 8 --variant vec_lds" "--range-src pc"; do for n in base old; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --steps 30 --warmup 8 $cfg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('%-5s %-40s strict %6.1f  fast %6.1f' % ('$n', '$cfg', d['value']/1e3, o.get('Mpx_s',0)/1e3))"
done; done
} > $O/r03_exp41.txt 2>&1
cat $O/r03_exp41.txt
