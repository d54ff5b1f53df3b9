#!/bin/bash
# the queue words reset by the last wave of a launch (base) vs a memset node before every launch (prev)
O=gpurun_out; mkdir -p $O
{
echo "== self-resetting chunk queue (base) vs hipMemsetD32Async per launch (prev): strict | fast Gpx/s, us per launch"
for cfg in "uhd yuv420p10le 256" "uhd yuv420p10le 16" "uhd yuv420p10le 8" "uhd yuv420p10le 4" "1080p yuv420p 64" "1080p yuv420p 16" "uhd rgb24 8"; do set -- $cfg; for n in base prev base prev; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --size $1 --fmt $2 --frames $3 --variant vec_lds --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('%-6s %-12s frames %3d %-5s strict %6.1f (%7.1f us)  fast %6.1f' % ('$1', '$2', $3, '$n', d['value']/1e3, d['ms_per_step']*1e3, o.get('Mpx_s',0)/1e3))"
done; done
for n in base prev; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --pipeline hbm 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$n', 'host_us_per_apply', d.get('host_us_per_apply'))"
done
} > $O/r03_exp33.txt 2>&1
cat $O/r03_exp33.txt
