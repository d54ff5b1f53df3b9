#!/bin/bash
O=gpurun_out; mkdir -p $O
{
echo "== small launches under four builds of the headline kernel (vec_lds forced): strict / fast Gpx/s and us per launch (strict)"
for cfg in "uhd 1" "uhd 4" "uhd 8" "uhd 16" "uhd 32" "1080p 64"; do set -- $cfg
for n in base head0 premix round2; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --no-stats --size $1 --frames $2 --variant vec_lds --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('%-6s frames %3d %-7s strict %6.1f Gpx/s (%6.1f us)  fast %6.1f (%6.1f us)' % ('$1', $2, '$n', d['value']/1e3, d['ms_per_step']*1e3, o.get('Mpx_s',0)/1e3, o.get('kernel_ms',0)*1e3))"
done; done
} > $O/r03_exp9.txt 2>&1
cat $O/r03_exp9.txt
