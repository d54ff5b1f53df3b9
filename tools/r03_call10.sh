#!/bin/bash
O=gpurun_out; mkdir -p $O
{
echo "== where does a short launch's time go?  1080p x 1 frame and UHD x 1 frame on the tile kernel (strict), us per launch"
for size in 1080p uhd; do
for env in "A=0" "LUTR_TUBE_H=0" "LUTR_NO_TAB=1" "LUTR_WAVES_PER_CU=16 LUTR_CHUNK=1" "LUTR_CHUNK=4"; do
  env $env timeout -k 10 100 python bench.py --lean --no-stats --no-other --size $size --frames 1 --variant vec_lds --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-6s %-40s %7.1f us  %s' % ('$size', '$env', d['ms_per_step']*1e3, d['config']['kernel']))"
done; done
} > $O/r03_exp10.txt 2>&1
cat $O/r03_exp10.txt
