#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 5 90 python -c "import __graft_entry__ as g; g.smoke()" > $O/r03_smoke_l.log 2>&1 || { echo "smoke failed or hung"; tail -3 $O/r03_smoke_l.log; exit 1; }
{
echo "== small launches with the two-level queue and small chunks: vec_global vs tile kernels (strict = value, fast = other)"
for size in uhd 1080p; do for f in 1 2 4 8 16 32 64; do for v in vec_global vec_lds; do
  timeout -k 10 100 python bench.py --lean --no-stats --size $size --frames $f --variant $v --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('%-6s frames %3d %-10s strict %6.1f Gpx/s (%6.1f us)  fast %6.1f   %s' % ('$size', $f, '$v', d['value']/1e3, d['ms_per_step']*1e3, o.get('Mpx_s',0)/1e3, d['config']['kernel']))"
done; done; done
echo "== 256 frames and content"
tools/exp_run.sh base
tools/ab_dist.sh base
} > $O/r03_exp19.txt 2>&1
cat $O/r03_exp19.txt
