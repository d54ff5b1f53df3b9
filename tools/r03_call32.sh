#!/bin/bash
# short launches: chunk of 1 tile (LUTR_CHUNK=1) vs the default 2 (2048 px) vs 4
O=gpurun_out; mkdir -p $O
{
echo "== short launches, strict | fast Gpx/s by chunk height in tiles (default: 2 below 64 tiles per wave, else 4)"
for cfg in "uhd 4" "uhd 8" "uhd 16" "uhd 32" "uhd 64" "1080p 64" "1080p 128"; do set -- $cfg; for c in default 1 2 4; do
  env=""; [ "$c" != default ] && env="LUTR_CHUNK=$c"
  fmt=yuv420p10le; [ $1 = 1080p ] && fmt=yuv420p
  env $env timeout -k 10 100 python bench.py --lean --size $1 --fmt $fmt --frames $2 --variant vec_lds --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('%-6s frames %3d chunk %-7s strict %6.1f (%6.1f us)  fast %6.1f' % ('$1', $2, '$c', d['value']/1e3, d['ms_per_step']*1e3, o.get('Mpx_s',0)/1e3))"
done; done
} > $O/r03_exp32.txt 2>&1
cat $O/r03_exp32.txt
