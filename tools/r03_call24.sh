#!/bin/bash
O=gpurun_out; mkdir -p $O
{
echo "== trilinear on the RGB tube kernels: 16-byte nodes (base) vs 12-byte nodes and the wider tube (tri12); round-1 kernel for reference"
for cfg in "gbrp" "gbrp10le" "rgb24" "rgba"; do for n in base tri12; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib LUTR_RGB2=all timeout -k 10 100 python bench.py --lean --no-other --fmt $cfg --frames 128 --interp trilinear --variant vec_lds --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['config'].get('lds_window') or {}
print('%-9s trilinear %-6s %6.1f Gpx/s %.3f  %s  tube %s / %s' % ('$cfg', '$n', d['value']/1e3, d['roofline']['frac'], d['config']['kernel'], w.get('tube_tiles'), w.get('tiles')))"
done; done
for cfg in gbrp gbrp10le; do
  LUTR_RGB2=0 timeout -k 10 100 python bench.py --lean --no-other --fmt $cfg --frames 128 --interp trilinear --variant vec_lds --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-9s trilinear round1 %6.1f Gpx/s %.3f  %s' % ('$cfg', d['value']/1e3, d['roofline']['frac'], d['config']['kernel']))"
done
} > $O/r03_exp24.txt 2>&1
cat $O/r03_exp24.txt
